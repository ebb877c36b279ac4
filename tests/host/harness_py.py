"""TEST ONLY: ctypes wrapper of tests/host/libharness.so (kernel math compiled for the CPU)."""
import ctypes as C, os, subprocess
from solorl_amd.config import SoloConfig, EnvState
HERE = os.path.dirname(os.path.abspath(__file__))
_L = None
def lib():
    global _L
    if _L is None:
        so = os.path.join(HERE, "libharness.so")
        srcs = [os.path.join(HERE, "host_harness.cpp"), os.path.join(HERE, "host_shim.hpp"),
                os.path.join(HERE, "../../solorl_amd/csrc/dynamics.hpp"), os.path.join(HERE, "../../solorl_amd/csrc/spatial.hpp"),
                os.path.join(HERE, "../../include/solorl_model_data.h"), os.path.join(HERE, "../../include/solorl.h")]
        if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
            subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-I" + HERE, "-o", so, srcs[0]])
        _L = C.CDLL(so)
        _L.harness_substep.argtypes = [C.POINTER(EnvState), C.POINTER(SoloConfig), C.c_int]
        _L.harness_substep.restype = None
    return _L
def substep(state, cfg, use_float=False):
    lib().harness_substep(C.byref(state), C.byref(cfg), int(use_float))
