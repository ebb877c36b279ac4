"""gpurun_out/parity_stats.json (written by the -m gpu parity tests, tests/util.py check_parity_stats) -> tests/golden/parity_measured.json:
the MEASURED p50 / p90 / p99 of every resynced comparison, which the tests then hold themselves to within PARITY_MARGIN (3 x).
Run after a GPU test run whose numbers are to become the reference; prints the table DESIGN.md section 2 quotes."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "parity_stats.json")
cur = json.load(open(src))
dst = os.path.join(ROOT, "tests", "golden", "parity_measured.json")
json.dump(cur, open(dst, "w"), indent=1, sort_keys=True)
print("| comparison | samples | p50 | p90 | p99 | max |\n|---|---|---|---|---|---|")
for k in sorted(cur):
    q = cur[k]
    print("| %s | %d | %.1e | %.1e | %.1e | %.1e |" % (k, q["n"], q["p50"], q["p90"], q["p99"], q["max"]))
