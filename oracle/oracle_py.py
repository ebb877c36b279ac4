"""ctypes binding of oracle/build/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; nothing
under solorl_amd/ does.  See oracle/solo_oracle.h ("PARITY UNPINNED").
"""
import ctypes as C
import os
import subprocess

import numpy as np

from solorl_amd.config import SoloConfig, EnvState

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "build", "liboracle.so")
    if force or not os.path.exists(so) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(so)
            for f in ("solo_oracle.c", "solo_oracle.h", "hull_data.h", "../include/solorl.h", "../include/solorl_model_data.h")):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s", "build/liboracle.so"],
                              stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.oracle_create.restype = C.c_void_p
        L.oracle_create.argtypes = [C.POINTER(SoloConfig), C.c_int, C.c_uint64, C.c_int64]
        for name in ("oracle_destroy", "oracle_dims", "oracle_reset", "oracle_step", "oracle_get_observation",
                     "oracle_increment_curriculum", "oracle_get_state", "oracle_set_state", "oracle_set_threads",
                     "oracle_substep", "oracle_mass_matrix", "oracle_forward_dynamics", "oracle_energy_momentum",
                     "oracle_prim_points", "oracle_last_lambda", "oracle_philox", "oracle_euler_from_quat", "oracle_set_caps",
                     "oracle_set_contact_model", "oracle_last_counts", "oracle_iteration_histogram"):
            getattr(L, name).restype = None
        L.oracle_increment_curriculum.argtypes = [C.c_void_p, C.c_double]
        L.oracle_set_option.restype = C.c_int
        L.oracle_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_double]
        L.oracle_last_residual.restype = C.c_double
        L.oracle_last_residual.argtypes = [C.c_void_p, C.c_int]
        L.oracle_last_iterations.restype = C.c_int
        L.oracle_last_iterations.argtypes = [C.c_void_p, C.c_int]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Oracle:
    """Batched fp64 CPU env with the same call shape as the HIP engine (numpy in/out)."""

    def __init__(self, cfg, num_envs, seed=1, env_id_offset=0, threads=1):
        self.L = lib()
        self.cfg = cfg.copy()
        self.h = C.c_void_p(self.L.oracle_create(C.byref(self.cfg), num_envs, seed, env_id_offset))
        o, a, n = C.c_int(), C.c_int(), C.c_int()
        self.L.oracle_dims(self.h, C.byref(o), C.byref(a), C.byref(n))
        self.O, self.A, self.N = o.value, a.value, n.value
        self.nv = 6 + self.A
        self.np = 24          # SOLORL_MAX_PRIMS (buffers for the low-level hooks)
        self.L.oracle_set_threads(self.h, threads)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.oracle_destroy(self.h)
            self.h = None

    def reset(self):
        obs = np.zeros((self.N, self.O))
        self.L.oracle_reset(self.h, _p(obs))
        return obs

    def step(self, actions):
        a = np.ascontiguousarray(actions, dtype=np.float64).reshape(self.N, self.A)
        obs = np.zeros((self.N, self.O)); rew = np.zeros(self.N); done = np.zeros(self.N, np.uint8)
        info = dict(timeout=np.zeros(self.N, np.uint8), success=np.zeros(self.N, np.uint8),
                    episode_length=np.zeros(self.N, np.int32), episode_reward=np.zeros(self.N),
                    goals_reached=np.zeros(self.N), dr=np.zeros((self.N, 5)))
        self.L.oracle_step(self.h, _p(a), _p(obs), _p(rew), _p(done), _p(info["timeout"]), _p(info["success"]),
                           _p(info["episode_length"]), _p(info["episode_reward"]), _p(info["goals_reached"]),
                           _p(info["dr"]))
        return obs, rew, done, info

    def get_observation(self):
        obs = np.zeros((self.N, self.O))
        self.L.oracle_get_observation(self.h, _p(obs))
        return obs

    def increment_curriculum(self, v=1.0):
        self.L.oracle_increment_curriculum(self.h, float(v))

    def get_state(self, i=0):
        s = EnvState()
        self.L.oracle_get_state(self.h, int(i), C.byref(s))
        return s

    def set_state(self, i, s):
        self.L.oracle_set_state(self.h, int(i), C.byref(s))

    def substep(self, i=0):
        self.L.oracle_substep(self.h, i)

    def mass_matrix(self, i=0):
        M = np.zeros((self.nv, self.nv)); h = np.zeros(self.nv)
        self.L.oracle_mass_matrix(self.h, i, _p(M), _p(h))
        return M, h

    def forward_dynamics(self, i=0):
        u = np.zeros(self.nv)
        self.L.oracle_forward_dynamics(self.h, i, _p(u))
        return u

    def energy_momentum(self, i=0):
        o = np.zeros(8)
        self.L.oracle_energy_momentum(self.h, i, _p(o))
        return dict(T=o[0], V=o[1], p=o[2:5].copy(), L=o[5:8].copy())

    def prim_points(self, i=0):
        o = np.zeros((self.np, 4))
        self.L.oracle_prim_points(self.h, i, _p(o))
        return o

    def last_iterations(self, i=0):
        return int(self.L.oracle_last_iterations(self.h, int(i)))

    def last_residual(self, i=0):
        return float(self.L.oracle_last_residual(self.h, int(i)))

    def set_caps(self, max_contacts=0, max_limits=0):
        """engine emulation (0 = uncapped, the default): see oracle_set_caps"""
        self.L.oracle_set_caps(self.h, int(max_contacts), int(max_limits))

    def set_contact_model(self, model):
        """0 = analytic primitives (the engine's model), 1 = hull manifolds (Bullet's scheme [K6])"""
        self.L.oracle_set_contact_model(self.h, int(model))

    def set_option(self, name, value):
        """[K] ledger switches that are not solorl_config fields (oracle/solo_oracle.h oracle_set_option)"""
        if self.L.oracle_set_option(self.h, name.encode(), float(value)) != 0:
            raise KeyError(name)

    def iteration_histogram(self, clear=False):
        """sub-steps with constraint rows, by the number of PGS sweeps they ran (bin = sweeps, 128 bins), summed over envs"""
        o = np.zeros(128, np.int64)
        self.L.oracle_iteration_histogram(self.h, _p(o), int(bool(clear)))
        return o

    def last_counts(self, i=0):
        """(contact points found, solved, joint-limit candidates, solved) of env i's last sub-step"""
        o = (C.c_int * 4)()
        self.L.oracle_last_counts(self.h, int(i), o)
        return tuple(o)

    def last_lambda(self, i=0):
        o = np.zeros(self.np)
        self.L.oracle_last_lambda(self.h, i, _p(o))
        return o


def philox(k0, k1, c0, c1, c2, c3):
    out = (C.c_uint32 * 4)()
    lib().oracle_philox(C.c_uint32(k0), C.c_uint32(k1), C.c_uint32(c0), C.c_uint32(c1), C.c_uint32(c2),
                        C.c_uint32(c3), out)
    return list(out)


def euler_from_quat(q):
    qq = (C.c_double * 4)(*q); r = (C.c_double * 3)()
    lib().oracle_euler_from_quat(qq, r)
    return list(r)
