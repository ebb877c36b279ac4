"""Per-wavefront cycle histogram of the team-mode step kernel at steady state (dev tool; rebuilds the engine with
-DSOLO_WAVE_TIMING on the GPU box -- four time stamps per wavefront, plain stores at the end -- and restores the normal
build afterwards).  One wavefront = 4 envs = one workgroup; with <= 1024 workgroups every wavefront has a SIMD to itself
and the launch lasts as long as its slowest wavefront.    usage: wave_hist.py [N] > profiles/rNN_wave_hist.txt"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if "SOLO_WAVE_TIMING" not in os.environ.get("SOLORL_BUILD_DEFINES", ""):
    env = dict(os.environ, SOLORL_BUILD_DEFINES=(os.environ.get("SOLORL_BUILD_DEFINES", "") + " SOLO_WAVE_TIMING " + os.environ.get("WH_DEFINES", "")).strip())   # WH_DEFINES: SOLO_SWEEP_STATS | SOLO_DEAD_STATS
    subprocess.check_call([sys.executable, "-m", "solorl_amd.build", "-f"], cwd=ROOT, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    rc = subprocess.call([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env)
    subprocess.check_call([sys.executable, "-m", "solorl_amd.build", "-f"], cwd=ROOT, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    sys.exit(rc)
import numpy as np
import torch
from solorl_amd import _native
from solorl_amd.config import *
from solorl_amd.vec_env import SoloVecEnv
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
c = default_config(ROBOT_SOLO12, TASK_WALK); c.num_history_stack = 1
if os.environ.get("WH_THR") is not None: c.solver_residual_threshold = float(os.environ["WH_THR"])
if os.environ.get("WH_WARM") is not None: c.warmstart = float(os.environ["WH_WARM"])
print("solver_residual_threshold %g, warmstart %g" % (c.solver_residual_threshold, c.warmstart))
env = SoloVecEnv(c, N, device="cuda:0", seed=1); env.reset()
g = torch.Generator(device="cuda:0"); g.manual_seed(1234)
a = torch.rand(64, N, 12, device="cuda:0", generator=g) * 2 - 1
for t in range(450): env.step_inplace(a[t % 64])
L = _native.lib()
nw = ((N + 3) // 4 + 7) & ~7
F = 28
recs, wall = [], []
buf = (C.c_ulonglong * (nw * F))()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for t in range(40):
    L.solorl_debug_wave_times(None, nw, 1)
    ev[0].record(); env.step_inplace(a[t % 64]); ev[1].record()
    L.solorl_debug_wave_times(buf, nw, 0)
    wall.append(ev[0].elapsed_time(ev[1]) * 1e3)
    recs.append(np.frombuffer(buf, dtype=np.uint64).reshape(nw, F).astype(np.float64).copy())
R = np.stack(recs)                      # [launch][wave][field]
tot = R[:, :, 0] + R[:, :, 1] + R[:, :, 2]
tick_ns = R[:, :, 3].sum() * 10.0 / tot.sum()           # ns per clock64() tick (s_memrealtime: 100 MHz); measured 1.0 on gfx950:
us = lambda ticks: ticks * tick_ns * 1e-3               # clock64() is a constant-rate counter here, not the 2.4 GHz shader clock
print("step_kernel_team<float, solo12>, %d envs, %d wavefronts, 40 launches at steady state; launch (HIP events, eager) %.1f us; clock64 tick %.3f ns" % (
    N, nw, np.mean(wall), tick_ns))
print("wavefront duration: mean %.1f us  median %.1f  p90 %.1f  p99 %.1f  slowest of a launch: mean %.1f (min %.1f max %.1f)" % (
    us(tot.mean()), us(np.median(tot)), us(np.percentile(tot, 90)), us(np.percentile(tot, 99)), us(tot.max(axis=1).mean()),
    us(tot.max(axis=1).min()), us(tot.max(axis=1).max())))
print("slowest / median wavefront = %.2f ; slowest / mean = %.2f" % (tot.max(axis=1).mean() / np.median(tot), tot.max(axis=1).mean() / tot.mean()))
print("parts of a wavefront (mean): load + action + history push %.1f us, the %d sub-steps %.1f us, reward / termination / reset / observation / store %.1f us" % (
    us(R[:, :, 0].mean()), c.frame_skip, us(R[:, :, 1].mean()), us(R[:, :, 2].mean())))
segs = ["load env", "torques", "history push", "sub-steps", "state back", "reward/termination/outputs", "auto-reset", "obs state", "store env", "obs write + tail"]
print("intervals between stamps (us; mean over all wavefronts | over the slowest 1 %):")
slow_ = tot >= np.percentile(tot, 99)
for i, n_ in enumerate(segs):
    print("  %-28s %7.2f | %7.2f" % (n_, us(R[:, :, 6 + i].mean()), us(R[:, :, 6 + i][slow_].mean())))
fr = R[:, :, 26].sum(); frs = R[:, :, 26][slow_].sum()
if "SOLO_SWEEP_STATS" in os.environ.get("SOLORL_BUILD_DEFINES", "") and fr > 0:
    its = R[:, :, 26] / np.maximum(R[:, :, 27], 1)          # mean sweeps per solve of each wavefront and launch
    print("sweeps per solve as the WAVEFRONT runs them (until its four envs are done): mean %.1f  median %.1f  p90 %.1f; share of wavefronts whose solves all ran the 50: %.1f %%; slowest 1 %% of the wavefronts: mean %.1f" % (
        its[R[:, :, 27] > 0].mean(), np.median(its[R[:, :, 27] > 0]), np.percentile(its[R[:, :, 27] > 0], 90), 100.0 * (its >= 49.99).mean(), its[slow_].mean()))
elif fr > 0: print("friction-slot visits with zero bound and zero impulse in every lane of the wavefront (a skip would be value-exact): %.1f %% of all, %.1f %% in the slowest 1 %% of the wavefronts" % (
    100.0 * R[:, :, 27].sum() / max(fr, 1), 100.0 * R[:, :, 27][slow_].sum() / max(frs, 1)))
phases = ["sin/cos", "collision front", "legs (4 lanes)", "leg sum", "base solve (leader)", "leg rates", "row finish", "PGS sweep", "integrate"]
print("phases of the sub-steps, summed over a step's %d sub-steps (us; mean over all wavefronts | over the slowest 1 %%; timing build drains the memory counters at every stamp):" % c.frame_skip)
for i, n_ in enumerate(phases):
    print("  %-28s %7.2f | %7.2f" % (n_, us(R[:, :, 16 + i].mean()), us(R[:, :, 16 + i][slow_].mean())))
bw = 10.0
edges = np.arange(0, us(tot.max()) + bw, bw)
h, _ = np.histogram(us(tot), bins=edges)
print("histogram of wavefront durations (bin = %.0f us):" % bw)
for lo, n_ in zip(edges[:-1], h):
    if n_:
        print("  %5.0f-%5.0f us  %7.3f %%  %s" % (lo, lo + bw, 100.0 * n_ / tot.size, "#" * int(60 * n_ / h.max())))
# least squares: sub-step time of a wavefront = a + b * (slots swept, summed over its sub-steps)
X = np.stack([np.ones(tot.size), R[:, :, 4].ravel()], axis=1)
coef, *_ = np.linalg.lstsq(X, us(R[:, :, 1].ravel()), rcond=None)
print("sub-steps of a wavefront = %.1f us + %.2f us per swept slot (x 50 if every solve ran 50 sweeps) => %.1f ns = %.0f cycles at 2.4 GHz per slot and sweep; "
      "mean slots per sub-step %.2f" % (coef[0], coef[1], coef[1] * 1e3 / 50, coef[1] * 1e3 / 50 * 2.4, R[:, :, 4].mean() / c.frame_skip))
print("by the wavefront's largest contact count in the step (max over its sub-steps): share, mean duration (before / sub-steps / after)")
for k in range(9):
    m = R[:, :, 5] == k
    if m.any():
        print("  ncmax %d: %6.2f %%  %6.1f us  (%5.1f / %6.1f / %5.1f)" % (k, 100 * m.mean(), us(tot[m].mean()), us(R[:, :, 0][m].mean()), us(R[:, :, 1][m].mean()), us(R[:, :, 2][m].mean())))
slow = tot >= np.percentile(tot, 99)
print("slowest 1 %% of the wavefronts: %.1f us (%.1f / %.1f / %.1f), mean slots per sub-step %.2f (all wavefronts %.2f)" % (
    us(tot[slow].mean()), us(R[:, :, 0][slow].mean()), us(R[:, :, 1][slow].mean()), us(R[:, :, 2][slow].mean()), R[:, :, 4][slow].mean() / c.frame_skip, R[:, :, 4].mean() / c.frame_skip))
