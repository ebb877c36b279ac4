import ctypes as C
import json
import os

import numpy as np

from solorl_amd.config import EnvState

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def clone(s):
    t = EnvState()
    C.memmove(C.byref(t), C.byref(s), C.sizeof(EnvState))
    return t


def state_vec(s, n):
    return np.concatenate([s.pos, s.quat, s.lin_vel, s.ang_vel, np.array(s.q)[:n], np.array(s.qd)[:n]])


def load_state(name):
    from tests.golden.make_golden import state_from_dict
    return state_from_dict(json.load(open(os.path.join(GOLDEN, name))))
