"""A/B check of two engine builds (SOLORL_LIB=...): dump state after settle + K steps from a fixed seed."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from solorl_amd.config import *
from solorl_amd.vec_env import SoloVecEnv
np.set_printoptions(precision=6, suppress=True, linewidth=220)
c = default_config(ROBOT_SOLO8, TASK_STAND); c.settle_min = c.settle_max = int(sys.argv[1]) if len(sys.argv) > 1 else 1
c.disable_termination = 1
env = SoloVecEnv(c, 4, device="cuda:0", seed=7)
env.reset()
s = env.get_state(0)
print("q", np.array(s.q)[:8]); print("pos", np.array(s.pos), "lam", np.array(s.lambda_prev)); print("mask", bin(s.contact_mask))
