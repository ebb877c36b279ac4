import sys, time
t00 = time.time()
def log(*a): print("[%.1f]" % (time.time() - t00), *a, flush=True)
log("start")
import numpy as np, torch
log("torch imported", torch.cuda.is_available())
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
from solorl_amd.config import *
from solorl_amd.vec_env import SoloVecEnv
from solorl_amd import _native
L = _native.lib(); log("lib loaded")
x = torch.zeros(4, device="cuda:0"); torch.cuda.synchronize(); log("torch cuda init")
for prec in (PRECISION_F32, PRECISION_F64):
    c = default_config(ROBOT_SOLO12, TASK_WALK); c.num_history_stack = 1; c.settle_min = c.settle_max = 2
    c.disable_termination = 1; c.precision = prec
    env = SoloVecEnv(c, 2, device="cuda:0", seed=1); log("created prec", prec)
    s = env.get_state(0); log("state0 pos", list(s.pos), "quat", list(s.quat))
    og = env.reset(); torch.cuda.synchronize(); log("reset done", og[0, :6].cpu().numpy())
    s = env.get_state(0); log("after reset pos", list(s.pos), "quat", list(s.quat), "mask", bin(s.contact_mask))
    a = torch.zeros(2, 12, device="cuda:0")
    for t in range(3):
        o, r, d, i = env.step(a); torch.cuda.synchronize(); log("step", t, o[0, :4].cpu().numpy(), r.cpu().numpy().ravel())
