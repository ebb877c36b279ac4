"""Copy one GPU round's evidence from gpurun_out/<tag>_* into profiles/<round>_* and rebuild profiles/<round>_traffic.json
(the PMC-derived numbers bench.py attaches to its roofline object).  Dev tool, runs here (no GPU).
usage: collect_profiles.py <tag> <round>      e.g.  collect_profiles.py r02n r02"""
import hashlib, json, os, re, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tag, rnd = sys.argv[1], sys.argv[2]
G = lambda n: os.path.join(ROOT, "gpurun_out", "%s_%s" % (tag, n))
P = lambda n: os.path.join(ROOT, "profiles", "%s_%s" % (rnd, n))


def counters(*names):
    out = {}
    for n in names:
        for ln in open(G(n)):
            m = re.match(r"(.*?)\s{2,}(\S+)\s+avg/dispatch\s+([0-9.]+)", ln)
            if m: out[m.group(2)] = float(m.group(3))
    return out


def cat(dst, *srcs):
    with open(dst, "w") as f:
        for s in srcs: f.write(open(G(s)).read())


cat(P("pmc_hbm_traffic.txt"), "pmc_fetch.txt", "pmc_write.txt")
cat(P("pmc_issue.txt"), "pmc_issue.txt", "pmc_issue2.txt")
cat(P("pmc_fp32.txt"), "pmc_fp.txt", "pmc_fp2.txt")
shutil.copy(G("kernel_stats.csv"), P("kernel_stats.csv"))
shutil.copy(G("bench_g2.json"), P("bench_2rank_rehearsal_one_gpu.json"))
for extra, dst in (("bench_g4.json", "bench_4rank_rehearsal_one_gpu.json"), ("bench.json", "bench.json"), ("sizes.txt", "sizes.txt"),
                   ("wave_hist.txt", "wave_hist.txt"), ("wave_hist_8192.txt", "wave_hist_8192.txt")):
    if os.path.exists(G(extra)): shutil.copy(G(extra), P(dst))
c = counters("pmc_fetch.txt", "pmc_write.txt", "pmc_issue.txt", "pmc_issue2.txt", "pmc_fp.txt")
lanes = c["SQ_THREAD_CYCLES_VALU"] / c["SQ_ACTIVE_INST_VALU"]   # active lanes per VALU instruction (about 45 of 64 for this kernel)
flops = (c["SQ_INSTS_VALU_ADD_F32"] + c["SQ_INSTS_VALU_MUL_F32"] + 2.0 * c["SQ_INSTS_VALU_FMA_F32"] + c["SQ_INSTS_VALU_TRANS_F32"]) * lanes
old = json.load(open(P("traffic.json"))) if os.path.exists(P("traffic.json")) else {}
t = {
    "bytes_per_launch": (c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0,
    "FETCH_SIZE_KB": c["FETCH_SIZE"], "WRITE_SIZE_KB": c["WRITE_SIZE"],
    "valu_insts_per_launch": c["SQ_INSTS_VALU"], "salu_insts_per_launch": c["SQ_INSTS_SALU"],
    "lds_insts_per_launch": c["SQ_INSTS_LDS"], "vmem_insts_per_launch": c["SQ_INSTS_VMEM"],
    "wave_cycles_quads_per_launch": c["SQ_WAVE_CYCLES"],
    "avg_active_lanes_per_valu_inst": lanes,
    "fp32_flops_per_launch": flops,
}
t["fp32_flops_note"] = ("instruction-derived from PMC: (SQ_INSTS_VALU_ADD_F32 + MUL_F32 + 2 x FMA_F32 + TRANS_F32) wave-instructions x average "
                        "active lanes (SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU = %.1f of 64); a packed v_pk_fma_f32 is counted as one FMA, so "
                        "this is a slight under-count; redundant lanes (both halves of a team hold the accumulators) are counted as executed" % lanes)
# the library these counters were measured on: bench.py attaches them to its roofline object only while it loads the SAME file
lib = os.path.join(ROOT, "solorl_amd", "_lib", "libsolorl_hip.so")
t["lib_sha256_16"] = hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16]
t["wait_any_frac"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
t["source"] = ("rocprofv3 --pmc passes (one counter group per pass, tools/dev/pmc_round.sh) on tools/dev/prof_step.py: step_kernel_team<float,1>, 4096 envs, random "
               "policy, 450 burn-in steps, average over the last 40 full-size dispatches: profiles/%s_pmc_hbm_traffic.txt, %s_pmc_issue.txt, %s_pmc_fp32.txt" % (rnd, rnd, rnd))
t["note"] = ("raw FETCH_SIZE / WRITE_SIZE (KB = 1024 B).  The gfx950 x2 FETCH_SIZE correction is documented for 16-B/lane streaming reads; these are 4-B/lane "
             "accesses of 256 B per wave instruction (uncalibrated, left raw).  Of the bytes written ~3.4 MB are what the algorithm stores; the rest is the write-back of "
             "dirty scratch lines (callee-saved register saves of the sweep variants of five and more contacts) -- not partial sectors: writing the observation rows as one "
             "sector-aligned block changed WRITE_SIZE by 0.03 % (profiles/r03_notes.md)")
json.dump(t, open(P("traffic.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in t.items() if not isinstance(v, str)}, indent=1))
