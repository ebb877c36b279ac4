"""ctypes binding of the C ABI (include/solorl.h).  No CPU fallback: if the HIP library is
missing or no GPU is present every entry point raises (the product path never touches oracle/)."""
import ctypes as C
import os

# torch must be imported BEFORE the engine is dlopen'ed: both depend on libamdhip64.so.7 and the
# first copy loaded wins; torch's bundled HIP runtime has to be that copy or device init fails.
import torch  # noqa: F401

from .config import SoloConfig, EnvState, InfoSoA

_LIB = None
LIB_PATH = os.environ.get("SOLORL_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "_lib", "libsolorl_hip.so")   # SOLORL_LIB: dev A/B builds

# every symbol include/solorl.h declares
SYMBOLS = ("solorl_default_config", "solorl_create", "solorl_destroy", "solorl_dims", "solorl_reset", "solorl_step",
           "solorl_get_observation", "solorl_increment_curriculum", "solorl_get_state", "solorl_set_state",
           "solorl_compute_returns", "solorl_ppo_loss", "solorl_last_error", "solorl_version")


class SoloRLError(RuntimeError):
    pass


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise SoloRLError("HIP engine not built: %s missing (run `python -m solorl_amd.build`); "
                              "there is no CPU fallback" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.solorl_last_error.restype = C.c_char_p
        L.solorl_version.restype = C.c_char_p
        L.solorl_default_config.argtypes = [C.POINTER(SoloConfig), C.c_int, C.c_int]
        L.solorl_create.argtypes = [C.POINTER(SoloConfig), C.c_int, C.c_int, C.c_uint64, C.c_int64, C.POINTER(C.c_void_p)]
        L.solorl_destroy.argtypes = [C.c_void_p]
        L.solorl_dims.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.solorl_reset.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.solorl_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(InfoSoA), C.c_void_p]
        L.solorl_get_observation.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.solorl_increment_curriculum.argtypes = [C.c_void_p, C.c_double]
        L.solorl_get_state.argtypes = [C.c_void_p, C.c_int, C.POINTER(EnvState)]
        L.solorl_set_state.argtypes = [C.c_void_p, C.c_int, C.POINTER(EnvState)]
        L.solorl_ppo_loss.argtypes = [C.c_void_p] * 8 + [C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_int, C.c_void_p]
        L.solorl_compute_returns.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                             C.c_float, C.c_float, C.c_int, C.c_void_p]
        for s in SYMBOLS:
            getattr(L, s)
        _LIB = L
    return _LIB


def check(rc):
    if rc != 0:
        raise SoloRLError("solorl error %d: %s" % (rc, lib().solorl_last_error().decode()))
