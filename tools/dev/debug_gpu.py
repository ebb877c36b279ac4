import sys, time
import numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
from solorl_amd.config import *
from solorl_amd.vec_env import SoloVecEnv
from oracle.oracle_py import Oracle
np.set_printoptions(precision=5, suppress=True, linewidth=200)
def show(tag, s, n):
    print(tag, "pos", np.array(s.pos), "quat", np.array(s.quat), "v", np.array(s.lin_vel), "w", np.array(s.ang_vel))
    print("   q", np.array(s.q)[:n], "qd", np.array(s.qd)[:n])
    print("   lam", np.array(s.lambda_prev), "mask", bin(s.contact_mask), "t", s.timestep, "rng", s.rng_counter, flush=True)
for prec in (PRECISION_F64, PRECISION_F32):
    c = default_config(ROBOT_SOLO12, TASK_WALK); c.num_history_stack = 1; c.settle_min = c.settle_max = 8
    c.disable_termination = 1; c.precision = prec
    t0 = time.time()
    env = SoloVecEnv(c, 2, device="cuda:0", seed=1)
    print("create", time.time() - t0, flush=True)
    orc = Oracle(c, 2, seed=1)
    show("gpu before reset", env.get_state(0), 12)
    # set/get roundtrip
    s = orc.get_state(0); s.pos[0] = 1.25; s.q[3] = 0.5; s.lambda_prev[13] = 0.125; s.timestep = 7
    env.set_state(1, s); show("roundtrip", env.get_state(1), 12)
    og = env.reset().cpu().numpy(); oo = orc.reset()
    show("gpu after reset", env.get_state(0), 12); show("orc after reset", orc.get_state(0), 12)
    print("obs gpu", og[0]); print("obs orc", oo[0])
    a = np.zeros((2, 12)); a[:, 2] = 0.5
    for t in range(3):
        t0 = time.time()
        og, rg, dg, info = env.step(torch.tensor(a, dtype=torch.float32, device="cuda:0")); torch.cuda.synchronize()
        print("step time", time.time() - t0)
        oo, ro, do, _ = orc.step(a)
        show("gpu step%d" % t, env.get_state(0), 12); show("orc step%d" % t, orc.get_state(0), 12)
        print("rew", rg.cpu().numpy().ravel(), ro, flush=True)
