#!/usr/bin/env python3
"""[K] sensitivity ledger (VERDICT r03 "Next" #1): what would it cost if an adopted Bullet default were wrong?

The reference leaves every Bullet / PyBullet parameter at its default (`simulation.py:13-35` never calls
`setPhysicsEngineParameter`, `solo.py:72-73` loads the URDF with `flags=0`), PyBullet is absent here, and every one of those defaults is a
[K] item of SURVEY.md Appendix B: restated from knowledge of the Bullet SDK, unverifiable in this container.  This script varies them ONE AT
A TIME on the fp64 oracle (CPU only; the oracle is test infrastructure) and prints, per variant and robot, the change of

  terminations / falls, mean base height, reward quantiles of the non-terminal steps        512 envs x 300 random-policy steps x 3 seeds
  PGS sweeps per solve: mean, median and the share of solves that never meet the residual    (same rollouts; only solves with rows)
  the 1000-step PD stance (tests/golden/make_golden.py stand_cfg): z range, mean sweeps

each beside the seed-to-seed scatter of the baseline, as a markdown table (DESIGN.md section 3 holds the committed copy) and as JSON.

    python tools/k_ledger.py [--envs 512] [--steps 300] [--seeds 1 2 3] [--threads 8] [--json profiles/r04_k_ledger.json] [--only name ...]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle.oracle_py import Oracle                                                           # noqa: E402
from solorl_amd.config import (default_config, ROBOT_SOLO8, ROBOT_SOLO12, TASK_WALK, TASK_STAND, CONTROL_PD,   # noqa: E402
                               FRICTION_CONE, FRICTION_PYRAMID)

# name -> (config overrides, oracle options, contact model, what it restates)
VARIANTS = [
    ("baseline", {}, {}, 0, "the defaults of solorl_default_config: friction cone, contact ERP 0.08, collision margin 1 mm, K1-K11 as adopted"),
    ("friction pyramid", {"friction_model": FRICTION_PYRAMID}, {}, 0, "rounds 1-3: each friction row clamped on its own, Gauss-Seidel between the two"),
    ("contact erp 0.2", {"contact_erp": 0.2}, {}, 0, "btContactSolverInfo's own m_erp2 default (rounds 1-3)"),
    ("round 3: pyramid, erp 0.2, margin 0", {"friction_model": FRICTION_PYRAMID, "contact_erp": 0.2, "collision_margin": 0.0}, {}, 0, "round 3's defaults"),
    ("pyramid: skip friction at zero normal", {"friction_model": FRICTION_PYRAMID}, {"friction_skip_zero_normal": 1}, 0, "`if (totalImpulse > 0)` of the pyramid branch"),
    ("collision margin 0", {"collision_margin": 0.0}, {}, 0, "the bare primitives (rounds 1-3) instead of Bullet's 1 mm margin around URDF hulls"),
    ("collision margin 2 mm", {"collision_margin": 0.002}, {}, 0, ""),
    ("breaking threshold x 0.5", {}, {"breaking_scale": 0.5}, 0, "gContactBreakingThreshold 0.01 instead of 0.02"),
    ("breaking threshold x 2", {}, {"breaking_scale": 2.0}, 0, "0.04"),
    ("damping 0", {"damping": 0.0}, {}, 0, "btMultiBody linear / angular damping 0 instead of 0.04 (K3)"),
    ("damping 0.08", {"damping": 0.08}, {}, 0, "twice K3's value"),
    ("URDF inertia (K2)", {"use_urdf_inertia": 1}, {}, 0, "URDF_USE_INERTIA_FROM_FILE instead of the AABB box rule"),
    ("held torque (K8)", {"hold_torque": 1}, {}, 0, "torque acting on all four sub-steps instead of the first"),
    ("no gyroscopic term", {}, {"gyro": 0}, 0, "btMultiBody::m_useGyroTerm off"),
    ("limit rows: split-impulse rule", {}, {"limit_split": 1}, 0, "no positional term beyond 0.04 rad (btMultiBodyJointLimitConstraint)"),
    ("linear slop 0", {"linear_slop": 0.0}, {}, 0, "PyBullet contactSlop 0 instead of 1e-5"),
    ("linear slop 1e-4", {"linear_slop": 1e-4}, {}, 0, ""),
    ("200 solver sweeps", {"solver_iterations": 200}, {}, 0, "numSolverIterations 200 instead of 50"),
    ("residual exit off", {"solver_residual_threshold": 0.0}, {}, 0, "always 50 sweeps (rounds 1-2)"),
    ("hull manifolds (K6)", {}, {}, 1, "Bullet's hull-vs-plane persistent manifolds instead of the primitives (oracle contact_model 1)"),
    ("hull manifolds, margin 0", {"collision_margin": 0.0}, {}, 1, "the same without the 1 mm hull margin"),
]


def make_cfg(robot, task, over):
    c = default_config(robot, task)
    c.num_history_stack = 1
    for k, v in over.items():
        setattr(c, k, v)
    return c


def rollout(robot, over, opts, model, seed, N, T, threads):
    n = 12 if robot == ROBOT_SOLO12 else 8
    o = Oracle(make_cfg(robot, TASK_WALK, over), N, seed=seed, threads=threads)
    if model:
        o.set_contact_model(model)
    for k, v in opts.items():
        o.set_option(k, v)
    o.reset()
    o.iteration_histogram(clear=True)
    rng = np.random.default_rng(seed + 100)
    acts = rng.uniform(-1, 1, size=(32, N, n))
    done = falls = 0
    zsum = 0.0
    rew = []
    for t in range(T):
        obs, r, d, info = o.step(acts[t % 32])
        done += int(d.sum()); falls += int((d.astype(bool) & ~info["timeout"].astype(bool)).sum())
        zsum += float(obs[:, 0].mean()); rew.append(r[d == 0])
    h = o.iteration_histogram()
    rew = np.concatenate(rew)
    tot = max(int(h.sum()), 1)
    cum = np.cumsum(h)
    cap = over.get("solver_iterations", 50)
    return dict(done=done, falls=falls, z=zsum / T, r10=float(np.percentile(rew, 10)), r50=float(np.median(rew)), r90=float(np.percentile(rew, 90)),
                sweeps_mean=float((h * np.arange(h.size)).sum() / tot), sweeps_median=int(np.searchsorted(cum, 0.5 * tot)),
                never=float(h[min(cap, h.size - 1):].sum() / tot), solves=tot)


def stance(over, opts, model, steps=1000):
    """1000-step PD stance of tests/golden/make_golden.py (stand, kp 5, kd 0.08, held torque, crouch + 0.005 sin): z range after the
    first 100 steps, mean sweeps of the last sub-step."""
    c = make_cfg(ROBOT_SOLO12, TASK_STAND, {})
    c.settle_min = c.settle_max = 8; c.disable_termination = 1
    c.control = CONTROL_PD; c.kp = 5.0; c.kd = 0.08; c.hold_torque = 1
    for k, v in over.items():
        setattr(c, k, v)
    o = Oracle(c, 1, seed=1)
    if model:
        o.set_contact_model(model)
    for k, v in opts.items():
        o.set_option(k, v)
    o.reset()
    crouch = np.array([0.0, 0.8, -1.6] * 4) / 10.0
    zs, its = [], []
    for t in range(steps + 400):
        a = crouch if t < 400 else crouch + 0.005 * np.sin(2 * np.pi * (t - 400) / 60 + np.arange(12) * np.pi / 6)
        o.step(a[None])
        if t >= 500:
            zs.append(o.get_state(0).pos[2]); its.append(o.last_iterations(0))
    return dict(zmin=float(np.min(zs)), zmax=float(np.max(zs)), sweeps=float(np.mean(its)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=512)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--seeds", type=int, nargs="+", default=[1, 2, 3])
    ap.add_argument("--threads", type=int, default=min(8, os.cpu_count() or 1))
    ap.add_argument("--json", type=str, default=None)
    ap.add_argument("--only", type=str, nargs="*", default=None, help="substrings of variant names (baseline always runs)")
    ap.add_argument("--no-stance", action="store_true")
    a = ap.parse_args()
    res = {}
    t0 = time.time()
    for name, over, opts, model, what in VARIANTS:
        if a.only and name != "baseline" and not any(s in name for s in a.only):
            continue
        res[name] = dict(what=what, robots={})
        for rname, robot in (("solo12", ROBOT_SOLO12), ("solo8", ROBOT_SOLO8)):
            runs = [rollout(robot, over, opts, model, s, a.envs, a.steps, a.threads) for s in a.seeds]
            agg = {k: float(np.mean([r[k] for r in runs])) for k in runs[0]}
            agg["done_seeds"] = [r["done"] for r in runs]
            agg["done_sd"] = float(np.std([r["done"] for r in runs], ddof=1)) if len(runs) > 1 else 0.0
            agg["z_sd"] = float(np.std([r["z"] for r in runs], ddof=1)) if len(runs) > 1 else 0.0
            agg["r50_sd"] = float(np.std([r["r50"] for r in runs], ddof=1)) if len(runs) > 1 else 0.0
            res[name]["robots"][rname] = agg
        if not a.no_stance:
            res[name]["stance"] = stance(over, opts, model)
        print("# %-40s done (%.0f s)" % (name, time.time() - t0), file=sys.stderr, flush=True)
    base = res["baseline"]
    print("| variant | robot | terminations (vs baseline, in baseline seed-sd) | falls | mean base z | reward p10 / p50 / p90 | sweeps mean / median | never converged | PD stance z range, sweeps |")
    print("|---|---|---|---|---|---|---|---|---|")
    for name, r in res.items():
        for rname in ("solo12", "solo8"):
            x, b = r["robots"][rname], base["robots"][rname]
            sd = max(b["done_sd"], 1.0)
            rel = 100.0 * (x["done"] - b["done"]) / b["done"]
            st = r.get("stance")
            stx = "%.3f-%.3f, %.1f" % (st["zmin"], st["zmax"], st["sweeps"]) if (st and rname == "solo12") else ""
            print("| %s | %s | %.0f (%+.1f %%, %+.1f sd) | %.0f | %.4f (%+.1f %%) | %.2f / %.2f / %.2f | %.1f / %d | %.2f %% | %s |" % (
                name, rname, x["done"], rel, (x["done"] - b["done"]) / sd, x["falls"], x["z"], 100.0 * (x["z"] - b["z"]) / b["z"],
                x["r10"], x["r50"], x["r90"], x["sweeps_mean"], x["sweeps_median"], 100.0 * x["never"], stx))
    print("\nbaseline seed scatter (sd over seeds %s): solo12 terminations %.0f, z %.4f, reward median %.3f; solo8 %.0f, %.4f, %.3f" % (
        a.seeds, base["robots"]["solo12"]["done_sd"], base["robots"]["solo12"]["z_sd"], base["robots"]["solo12"]["r50_sd"],
        base["robots"]["solo8"]["done_sd"], base["robots"]["solo8"]["z_sd"], base["robots"]["solo8"]["r50_sd"]))
    if a.json:
        json.dump(dict(envs=a.envs, steps=a.steps, seeds=a.seeds, variants=res), open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
