import sys, numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
from solorl_amd.config import *
from solorl_amd.vec_env import SoloVecEnv
N = 4096
c = default_config(ROBOT_SOLO12, TASK_WALK); c.num_history_stack = 1
env = SoloVecEnv(c, N, device="cuda:0", seed=1); env.reset()
a = torch.rand(16, N, 12, device="cuda:0") * 2 - 1
for t in range(300): env.step_inplace(a[t % 16])
cnt = []
for rep in range(3):
    for t in range(20): env.step_inplace(a[t % 16])
    torch.cuda.synchronize()
    for i in range(0, 1024):
        s = env.get_state(i)
        nl = sum((abs(q) > 9.5) for q in list(s.q))
        cnt.append((bin(s.contact_mask).count("1"), bin(s.contact_mask & 0xFFF).count("1"), min(nl, 2), s.timestep))
cnt = np.array(cnt)
rows = 3 * cnt[:, 0] + cnt[:, 2]
print("contacts hist", np.bincount(cnt[:, 0], minlength=9))
print("base contacts hist", np.bincount(cnt[:, 1], minlength=9))
print("rows mean %.1f median %d p90 %d max %d" % (rows.mean(), np.median(rows), np.percentile(rows, 90), rows.max()))
g = rows[:3072].reshape(-1, 64).max(1); print("max over groups of 64: mean %.1f" % g.mean())
g = rows[:3072].reshape(-1, 16).max(1); print("max over groups of 16: mean %.1f" % g.mean())
g = rows[:3072].reshape(-1, 4).max(1); print("max over groups of 4: mean %.1f" % g.mean())
srt = np.sort(rows[:3072]); print("sorted groups of 64: mean of max %.1f" % srt.reshape(-1, 64).max(1).mean())
print("timestep mean", cnt[:, 3].mean())
