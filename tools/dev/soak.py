"""Soak: many steps with aggressive random actions; counts NaN resets, checks bitwise reproducibility."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from solorl_amd.config import *
from solorl_amd.vec_env import SoloVecEnv
N, K = 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 10000
outs = []
for rep in range(2):
    for robot, task in ((ROBOT_SOLO12, TASK_WALK), (ROBOT_SOLO8, TASK_POINTGOAL)):
        c = default_config(robot, task); c.num_history_stack = 1
        env = SoloVecEnv(c, N, device="cuda:0", seed=5); env.reset()
        g = torch.Generator(device="cuda:0"); g.manual_seed(1)
        a = torch.rand(64, N, env.act_dim, device="cuda:0", generator=g) * 3 - 1.5       # beyond the +-1 clip on purpose
        nan = torch.zeros((), device="cuda:0"); dones = torch.zeros((), device="cuda:0"); acc = torch.zeros(N, device="cuda:0")
        t0 = time.time()
        for t in range(K):
            o, r, d, info = env.step_inplace(a[t % 64])
            nan += info["nan_reset"].sum(); dones += d.sum(); acc += r
        torch.cuda.synchronize()
        assert torch.isfinite(o).all() and torch.isfinite(acc).all()
        outs.append((robot, task, float(nan), float(dones), float(acc.sum()), o.clone()))
        print("robot %d task %d: %d steps x %d envs in %.1f s, episodes ended %d, nan resets %d, reward sum %.3f" % (robot, task, K, N, time.time() - t0, dones.item(), nan.item(), acc.sum().item()), flush=True)
for a, b in zip(outs[:2], outs[2:]):
    assert a[:5] == b[:5] and torch.equal(a[5], b[5]), "not reproducible"
print("reproducible: yes")
