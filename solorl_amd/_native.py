"""ctypes binding of the C ABI (include/solorl.h).  No CPU fallback: if the HIP library is
missing or no GPU is present every entry point raises (the product path never touches oracle/)."""
import ctypes as C
import os

# torch must be imported BEFORE the engine is dlopen'ed: both depend on libamdhip64.so.7 and the
# first copy loaded wins; torch's bundled HIP runtime has to be that copy or device init fails.
import torch  # noqa: F401

from .config import SoloConfig, EnvState, InfoSoA, ABI_VERSION

_LIB = None
LIB_PATH = os.environ.get("SOLORL_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "_lib", "libsolorl_hip.so")   # SOLORL_LIB: dev A/B builds

# every symbol include/solorl.h declares
SYMBOLS = ("solorl_default_config", "solorl_create", "solorl_destroy", "solorl_dims", "solorl_reset", "solorl_step",
           "solorl_get_observation", "solorl_increment_curriculum", "solorl_get_state", "solorl_set_state", "solorl_get_property",
           "solorl_compute_returns", "solorl_ppo_loss", "solorl_policy_act", "solorl_ppo_grad_stage1", "solorl_ppo_grad_stage2", "solorl_ppo_grad_count", "solorl_ppo_clip_adam", "solorl_last_error", "solorl_version",
           "solorl_abi_version", "solorl_step_act", "solorl_ppo_scratch_count")


class PolicyParams(C.Structure):            # solorl_policy_params
    _fields_ = [("obs_dim", C.c_int), ("act_dim", C.c_int), ("hidden", C.c_int), ("reserved0", C.c_int)] + [
        (n, C.c_void_p) for n in ("critic_w0", "critic_b0", "critic_w1", "critic_b1", "critic_w2", "critic_b2", "actor_w0", "actor_b0",
                                  "actor_w1", "actor_b1", "mean_w", "mean_b", "logstd")]


class PpoBatch(C.Structure):                # solorl_ppo_batch
    _fields_ = [(n, C.c_void_p) for n in ("obs", "actions", "old_logp", "adv", "vpred", "ret", "perm", "offset")] + [
        ("m", C.c_int), ("clipped_value", C.c_int), ("clip", C.c_float), ("value_coef", C.c_float)]


class PpoGrads(C.Structure):                # solorl_ppo_grads
    _fields_ = [(n, C.c_void_p) for n in ("critic_w0", "critic_b0", "critic_w1", "critic_b1", "critic_w2", "critic_b2", "actor_w0", "actor_b0",
                                          "actor_w1", "actor_b1", "mean_w", "mean_b", "logstd", "loss_sums", "logstd_sum", "scratch")] + [
        ("entropy_coef", C.c_float), ("reserved0", C.c_float)]


class AdamState(C.Structure):               # solorl_adam_state
    _fields_ = [("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p), ("step", C.c_void_p), ("lr", C.c_void_p), ("offset", C.c_void_p),
                ("offset_increment", C.c_int64), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float), ("weight_decay", C.c_float),
                ("max_grad_norm", C.c_float), ("grad_scale", C.c_float)]


class PpoStage1(C.Structure):               # solorl_ppo_stage1
    _fields_ = [(n, C.c_void_p) for n in ("xt0", "c_xt1", "c_xt2", "c_g1", "c_g2", "c_gh", "a_xt1", "a_xt2", "a_g1", "a_g2", "a_gh", "partials")]


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
        L.solorl_step_act.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(InfoSoA), C.POINTER(PolicyParams), C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.solorl_increment_curriculum.argtypes = [C.c_void_p, C.c_double]
        L.solorl_get_state.argtypes = [C.c_void_p, C.c_int, C.POINTER(EnvState)]
        L.solorl_set_state.argtypes = [C.c_void_p, C.c_int, C.POINTER(EnvState)]
        L.solorl_get_property.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_double)]
        L.solorl_ppo_loss.argtypes = [C.c_void_p] * 8 + [C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_int, C.c_void_p]
        L.solorl_compute_returns.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                             C.c_float, C.c_float, C.c_int, C.c_void_p]
        L.solorl_policy_act.argtypes = [C.POINTER(PolicyParams), C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.solorl_ppo_grad_stage1.argtypes = [C.POINTER(PolicyParams), C.POINTER(PpoBatch), C.POINTER(PpoStage1), C.c_int, C.c_void_p]
        L.solorl_ppo_grad_stage2.argtypes = [C.POINTER(PolicyParams), C.POINTER(PpoStage1), C.c_int, C.POINTER(PpoGrads), C.c_int, C.c_void_p]
        L.solorl_ppo_grad_count.argtypes = [C.c_int, C.c_int]
        L.solorl_ppo_scratch_count.argtypes = [C.c_int, C.c_int, C.c_int]
        L.solorl_ppo_scratch_count.restype = C.c_int
        L.solorl_ppo_clip_adam.argtypes = [C.POINTER(PolicyParams), C.POINTER(PpoGrads), C.POINTER(AdamState), C.c_int, C.c_void_p]
        for s in SYMBOLS:
            getattr(L, s)
        # the ctypes mirrors in config.py / this file are written against one layout of include/solorl.h's structs
        if L.solorl_abi_version() != ABI_VERSION:
            raise SoloRLError("%s was built for ABI version %d, this binding is written for %d: rebuild (`python -m solorl_amd.build`)"
                              % (LIB_PATH, L.solorl_abi_version(), ABI_VERSION))
        _LIB = L
    return _LIB


def check(rc):
    if rc != 0:
        raise SoloRLError("solorl error %d: %s" % (rc, lib().solorl_last_error().decode()))
