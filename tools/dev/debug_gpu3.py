import sys, time
t00 = time.time()
def log(*a): print("[%.1f]" % (time.time() - t00), *a, flush=True)
import numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
from solorl_amd.config import *
from solorl_amd.vec_env import SoloVecEnv
robot = int(sys.argv[1]); iters = int(sys.argv[2]); settle = int(sys.argv[3]); N = int(sys.argv[4])
c = default_config(robot, TASK_WALK); c.num_history_stack = 1; c.settle_min = c.settle_max = settle
c.disable_termination = 1; c.precision = PRECISION_F64; c.solver_iterations = iters
log("creating", robot, iters, settle, N)
env = SoloVecEnv(c, N, device="cuda:0", seed=1); log("created")
og = env.reset(); torch.cuda.synchronize(); log("reset done", og[0, :6].cpu().numpy())
s = env.get_state(0); log("after reset pos", list(s.pos), "quat", list(s.quat), "mask", bin(s.contact_mask))
a = torch.zeros(N, env.act_dim, device="cuda:0")
for t in range(3):
    o, r, d, i = env.step(a); torch.cuda.synchronize(); log("step", t, o[0, :4].cpu().numpy(), r.cpu().numpy().ravel()[:2])
