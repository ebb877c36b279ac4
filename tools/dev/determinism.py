"""Two runs must agree bitwise; the second one runs with the dynamic LDS pre-filled with NaNs (SOLORL_POISON_LDS),
so anything the kernel reads before writing it shows up.  Reports the first step / env / obs element that differs."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from solorl_amd.config import *
from solorl_amd.vec_env import SoloVecEnv
N, K = 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 300
c = default_config(ROBOT_SOLO12, TASK_WALK); c.num_history_stack = 1
g = torch.Generator(device="cuda:0"); g.manual_seed(5)
acts = torch.rand((32, N, 12), device="cuda:0", generator=g) * 2 - 1
traj = []
for rep in range(2):
    os.environ["SOLORL_POISON_LDS"] = "0" if rep == 0 else "0x7fc00000"
    env = SoloVecEnv(c, N, device="cuda:0", seed=9); env.reset()
    o_all = []
    for t in range(K):
        o, r, d, info = env.step_inplace(acts[t % 32])
        o_all.append(o.clone())
    traj.append(torch.stack(o_all))
diff = (traj[0] != traj[1])
if not diff.any():
    print("bitwise reproducible over %d steps" % K)
else:
    idx = diff.nonzero()[0].tolist()
    t, e, k = idx
    print("first difference at step %d env %d obs[%d]: %r vs %r; envs differing at that step: %d" % (t, e, k, traj[0][t, e, k].item(), traj[1][t, e, k].item(), int(diff[t].any(dim=1).sum())))
    print("NaN count in runs:", int(torch.isnan(traj[0]).sum()), int(torch.isnan(traj[1]).sum()))
