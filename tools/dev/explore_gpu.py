"""Exploratory GPU-vs-oracle comparison (not a pytest file)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
from solorl_amd.config import *
from solorl_amd.vec_env import SoloVecEnv
from oracle.oracle_py import Oracle

def run(prec, robot, task, steps, N=4, control=CONTROL_TORQUE):
    c = default_config(robot, task); c.num_history_stack = 1; c.settle_min = c.settle_max = 8
    c.disable_termination = 1; c.precision = prec; c.control = control
    env = SoloVecEnv(c, N, device="cuda:0", seed=1)
    orc = Oracle(c, N, seed=1)
    og = env.reset().cpu().numpy().astype(np.float64); oo = orc.reset()
    print("reset obs diff", np.abs(og - oo).max())
    A = env.act_dim
    maxq = 0
    for t in range(steps):
        a = np.stack([0.5 * np.sin(2 * np.pi * t / 60 + np.arange(A) * np.pi / 6 + 0.3 * i) for i in range(N)])
        og, rg, dg, _ = env.step(torch.tensor(a, dtype=torch.float32, device="cuda:0"))
        oo, ro, do, _ = orc.step(a.astype(np.float32).astype(np.float64))
        sg = [env.get_state(i) for i in range(N)]; so = [orc.get_state(i) for i in range(N)]
        dq = max(np.abs(np.array(sg[i].q) - np.array(so[i].q)).max() for i in range(N))
        dz = max(abs(sg[i].pos[2] - so[i].pos[2]) for i in range(N))
        maxq = max(maxq, dq)
        if t < 5 or t % 50 == 0 or t == steps - 1:
            print(t, "dq %.3e dz %.3e drew %.3e dobs %.3e  z=%.3f mask=%s/%s" % (dq, dz, np.abs(rg.cpu().numpy()[:, 0] - ro).max(),
                  np.abs(og.cpu().numpy() - oo).max(), so[0].pos[2], bin(sg[0].contact_mask), bin(so[0].contact_mask)))
    print("prec", prec, "robot", robot, "task", task, "max dq over run", maxq)

if __name__ == "__main__":
    run(PRECISION_F32, ROBOT_SOLO12, TASK_WALK, 300)
    run(PRECISION_F32, ROBOT_SOLO8, TASK_STAND, 100)
    run(PRECISION_F32, ROBOT_SOLO12, TASK_POINTGOAL, 100)
    run(PRECISION_F32, ROBOT_SOLO12, TASK_STAND, 300, control=CONTROL_PD)
    # timing
    c = default_config(ROBOT_SOLO12, TASK_WALK); c.num_history_stack = 1
    env = SoloVecEnv(c, 4096, device="cuda:0", seed=1); env.reset()
    a = torch.rand(4096, 12, device="cuda:0") * 2 - 1
    for _ in range(10): env.step_inplace(a)
    torch.cuda.synchronize(); t = time.time()
    for _ in range(100): env.step_inplace(a)
    torch.cuda.synchronize(); dt = time.time() - t
    print("4096 envs: %.3f ms/step, %.2f M env-steps/s" % (dt * 10, 4096 * 100 / dt / 1e6))
