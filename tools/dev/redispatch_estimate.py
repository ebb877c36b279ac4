"""What would it save if a wavefront switched to a smaller sweep variant whenever one of its four envs has converged?  CPU estimate on the oracle
(4096 Solo12-walk envs, random policy, steady state; last sub-step of each control step): slot-sweeps of the slowest wavefront of a launch,
as the engine does it (largest slot set x most sweeps) and with an ideal re-dispatch at every completion.  Result (round 4): 572 -> 559 for the
slowest wavefront (2 %: the env that does not converge IS the one with the most rows), 106 -> 95 for the mean wavefront (which nobody waits for)."""
import sys, numpy as np, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle.oracle_py import Oracle
from solorl_amd.config import default_config, ROBOT_SOLO12, TASK_WALK
c = default_config(ROBOT_SOLO12, TASK_WALK); c.num_history_stack = 1
N = 4096
o = Oracle(c, N, seed=1, threads=8); o.reset()
rng = np.random.default_rng(1)
acts = rng.uniform(-1, 1, size=(32, N, 12))
t0 = time.time()
for t in range(200): o.step(acts[t % 32])
print("burn-in %.0f s" % (time.time() - t0), flush=True)
now, ideal, mean_now, mean_ideal = [], [], [], []
for t in range(40):
    o.step(acts[t % 32])
    it = np.array([o.last_iterations(i) for i in range(N)])
    cnt = np.array([o.last_counts(i) for i in range(N)])
    nc, nl = np.minimum(cnt[:, 1], 8), cnt[:, 3]
    slots = (nl > 0).astype(int) + (nc + 1) // 2 + nc
    it = np.where(slots > 0, it, 0)
    it4, s4 = it.reshape(-1, 4), slots.reshape(-1, 4)
    cost_now = s4.max(1) * it4.max(1)
    order = np.argsort(it4, axis=1)
    its = np.take_along_axis(it4, order, 1); ss = np.take_along_axis(s4, order, 1)
    # active-set max slots from position k on
    suf = np.maximum.accumulate(ss[:, ::-1], axis=1)[:, ::-1]
    prev = np.concatenate([np.zeros((its.shape[0], 1), int), its[:, :-1]], 1)
    cost_ideal = ((its - prev) * suf).sum(1)
    now.append(cost_now.max()); ideal.append(cost_ideal.max()); mean_now.append(cost_now.mean()); mean_ideal.append(cost_ideal.mean())
    if t < 3:
        w = cost_now.argmax(); print("slowest wave: iters", it4[w], "slots", s4[w], "cost now", cost_now[w], "ideal", cost_ideal[w])
print("slot-sweeps of the slowest wave per sub-step: now %.0f, with re-dispatch %.0f (ratio %.3f); mean wave: %.1f -> %.1f" % (
    np.mean(now), np.mean(ideal), np.mean(ideal) / np.mean(now), np.mean(mean_now), np.mean(mean_ideal)))
