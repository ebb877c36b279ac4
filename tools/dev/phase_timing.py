"""Per-phase cycle breakdown of the team-mode step kernel (dev tool; rebuilds the engine with
-DSOLO_PHASE_TIMING on the GPU box, restores the normal build afterwards)."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if os.environ.get("SOLORL_BUILD_DEFINES") != "SOLO_PHASE_TIMING":
    env = dict(os.environ, SOLORL_BUILD_DEFINES="SOLO_PHASE_TIMING")
    subprocess.check_call([sys.executable, "-m", "solorl_amd.build", "-f"], cwd=ROOT, env=env)
    rc = subprocess.call([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env)
    subprocess.check_call([sys.executable, "-m", "solorl_amd.build", "-f"], cwd=ROOT)
    sys.exit(rc)
import torch
from solorl_amd import _native
from solorl_amd.config import *
from solorl_amd.vec_env import SoloVecEnv
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
c = default_config(ROBOT_SOLO12, TASK_WALK); c.num_history_stack = 1
env = SoloVecEnv(c, N, device="cuda:0", seed=1); env.reset()
a = torch.rand(16, N, 12, device="cuda:0") * 2 - 1
for t in range(100): env.step_inplace(a[t % 16])
torch.cuda.synchronize()
L = _native.lib()
buf = (C.c_ulonglong * 48)()
L.solorl_debug_phase_cycles(buf, 1)
pg = (C.c_ulonglong * 20)()
L.solorl_debug_pgs_cycles(pg, 1)
K = 100
for t in range(K): env.step_inplace(a[t % 16])
torch.cuda.synchronize()
L.solorl_debug_phase_cycles(buf, 0)
nw = (N + 3) // 4
names = ["detect", "legs", "base_lead", "finish", "(unused)", "pgs", "readback+integrate", "kernel prologue (to first substep)", "kernel to end of substeps", "kernel to after reward/termination/outputs", "kernel to after auto-reset", "kernel to after write_obs", "kernel to after store_env", "write_obs: current_state (euler)", "write_obs: history loads issued", "write_obs: stores issued"]
tot = 0
for i, n in enumerate(names):
    per = buf[i] / K / nw
    if i < 7: tot += per
    print("%-38s %10.0f cycles/step/wave  (%.1f us at 2.4 GHz)" % (n, per, per / 2400.0))
print("sum of substep phases %.0f cycles/step/wave" % tot)
L.solorl_debug_pgs_cycles(pg, 0)
print("PGS per call by the wave's largest contact count (50 sweeps):")
for c in range(10):
    if pg[10 + c]:
        slots = (c + 1) // 2 + c
        print("  ncmax %d: %6.2f %% of calls, %8.0f cycles/call, %6.1f cycles/sweep, %5.1f cycles/slot (%d slots, no limit slot)" % (
            c, 100.0 * pg[10 + c] / sum(pg[10:20]), pg[c] / pg[10 + c], pg[c] / pg[10 + c] / 50, pg[c] / pg[10 + c] / 50 / max(slots, 1), slots))
