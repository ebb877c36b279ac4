"""Gym-style vec-env surface over the HIP engine -- the drop-in for the reference's
``make_vec_envs`` -> ``PyTorchEnvWrapper(VecNormalize(VecEnvWrapper(...)))`` stack
(agents/ppo/envs.py:14-30, :183-222).  Same attribute names, argument meaning and error behaviour:

  reset() -> Tensor[N, O] f32                                  (envs.py:198-200)
  step(actions Tensor[N, A]) -> (obs[N,O], reward[N,1], done[N] f32, infos)   (envs.py:189-196)
  get_observation(), increment_curriculum(), close(), observation_space, action_space, nenvs

torch is plumbing only: it owns the device buffers and the stream; all arithmetic happens in
``libsolorl_hip.so`` on the caller's current HIP stream (no host synchronisation in step()).
"""
import ctypes as C

import numpy as np
import torch

from . import _native
from .config import SoloConfig, EnvState, InfoSoA, config_from_dict, EPSTAT_FIELDS, EPSTAT_NAMES


class Box:
    """Stand-in for gym.spaces.Box: the policy dispatches on the class NAME
    (agents/ppo/policy.py:22-31) and reads .shape / .high / .low."""

    def __init__(self, low, high):
        self.low = np.asarray(low, dtype=np.float32)
        self.high = np.asarray(high, dtype=np.float32)
        self.shape = self.low.shape
        self.dtype = np.float32

    def sample(self):
        lo = np.where(np.isfinite(self.low), self.low, -1.0)
        hi = np.where(np.isfinite(self.high), self.high, 1.0)
        return np.random.uniform(lo, hi).astype(np.float32)

    def __repr__(self):
        return "Box%s" % (self.shape,)


_INFO_KEYS = ("timeout", "success", "nan_reset", "episode_length", "episode_reward", "goals_reached",
              "dr_stand", "dr_joint_pose", "dr_torque", "dr_balance", "dr_progress")
_DR_NAMES = {"dr_stand": "dr/stand_rew", "dr_joint_pose": "dr/joint_pose_rew", "dr_torque": "dr/torque_rew",
             "dr_balance": "dr/roll_pitch_balance_rew", "dr_progress": "dr/progress_rew"}


class LazyInfos:
    """Sequence of per-env info dicts (reference: tuple of N dicts, envs.py:94-95) backed by SoA
    device tensors; dicts are only materialised for the envs a caller actually indexes."""

    def __init__(self, tensors, done):
        self._t = tensors
        self._done = done
        self._host = None

    def _fetch(self):
        if self._host is None:
            self._host = {k: v.cpu().numpy() for k, v in self._t.items()}
        return self._host

    def __len__(self):
        return self._done.shape[0]

    def __getitem__(self, i):
        h = self._fetch()
        d = {"episode_length": int(h["episode_length"][i]), "episode_reward": float(h["episode_reward"][i]),
             "goals_reached": float(h["goals_reached"][i]),
             # keys PPO reads at done but SoloBaseEnv never sets (SURVEY 8b [BUG]) -> 0.0
             "max_velocity": 0.0, "min_force": 0.0, "max_force": 0.0, "nan_reset": bool(h["nan_reset"][i])}
        for k, name in _DR_NAMES.items():
            d[name] = float(h[k][i])
        if h["done"][i]:
            d["timeout"] = bool(h["timeout"][i])
            d["success"] = bool(h["success"][i])
        return d

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    @property
    def tensors(self):
        """SoA device view: dict of [N] tensors (fast path for on-device episode statistics)."""
        return self._t


class SoloVecEnv:
    """Batched SoloBaseEnv on one MI355X.  ``config`` is a reference-style dict (configs/*.yaml)
    or a ``SoloConfig``."""

    def __init__(self, config, num_envs, device=None, seed=1, env_id_offset=0, applied_torque=False, **overrides):
        self.cfg = config.copy() if isinstance(config, SoloConfig) else config_from_dict(config, **overrides)
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None
        device = torch.device(device) if device is not None else None
        if device is not None and device.type == "cuda" and device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        if device is None or device.type != "cuda":
            raise _native.SoloRLError("SoloVecEnv needs a HIP device (got %r); the engine has no CPU fallback" % (device,))
        self.device = device
        self.L = _native.lib()
        self.nenvs = int(num_envs)
        h = C.c_void_p()
        _native.check(self.L.solorl_create(C.byref(self.cfg), self.nenvs, device.index or 0, int(seed),
                                           int(env_id_offset), C.byref(h)))
        self._h = h
        o, a, n = C.c_int(), C.c_int(), C.c_int()
        _native.check(self.L.solorl_dims(self._h, C.byref(o), C.byref(a), C.byref(n)))
        self.obs_dim, self.act_dim = o.value, a.value
        self.observation_space = Box(-np.inf * np.ones(self.obs_dim), np.inf * np.ones(self.obs_dim))  # baseEnv.py:27-28
        self.action_space = Box(-np.ones(self.act_dim), np.ones(self.act_dim))                          # baseEnv.py:23-25
        N = self.nenvs
        kw = dict(device=device)
        self._obs = torch.empty((N, self.obs_dim), dtype=torch.float32, **kw)
        self._rew = torch.empty((N,), dtype=torch.float32, **kw)
        self._done = torch.empty((N,), dtype=torch.uint8, **kw)
        self._info = {k: torch.zeros((N,), dtype=(torch.uint8 if k in ("timeout", "success", "nan_reset") else
                                                   torch.int32 if k == "episode_length" else torch.float32), **kw)
                      for k in _INFO_KEYS}
        # finished-episode accumulators [fields][N], updated by the step kernel itself (include/solorl.h ep_stats)
        self._ep_stats = torch.zeros((EPSTAT_FIELDS, N), dtype=torch.float32, **kw)
        # the torques the step applied (include/solorl.h applied_torque): allocated on request only (record_applied_torque())
        # (constructor flag `applied_torque`, or record_applied_torque() before the first step: the pointer is a kernel argument, and a HIP
        # graph captured earlier would keep replaying the NULL it was captured with)
        self._tau = torch.zeros((N, self.act_dim), dtype=torch.float32, **kw) if applied_torque else None
        self._steps_issued = 0
        self._info_c = InfoSoA(ep_stats=self._ep_stats.data_ptr(), applied_torque=None if self._tau is None else self._tau.data_ptr(),
                               **{k: self._info[k].data_ptr() for k in _INFO_KEYS})
        self.ob_rms = None          # VecNormalize(ob=False): agents/ppo/envs.py:26, read at train.py:126
        self.closed = False

    # reference call path: PyTorchEnvWrapper.envs (VecNormalize) .venv ...
    @property
    def envs(self):
        return self

    @property
    def venv(self):
        return self

    def __len__(self):
        return self.nenvs

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def reset(self):
        with torch.cuda.device(self.device):
            _native.check(self.L.solorl_reset(self._h, C.c_void_p(self._obs.data_ptr()), self._stream()))
        return self._obs.clone()

    def step(self, actions):
        if actions.shape != (self.nenvs, self.act_dim):
            raise AssertionError("actions must be [%d, %d]" % (self.nenvs, self.act_dim))   # solo.py:226
        a = actions.detach().to(device=self.device, dtype=torch.float32).contiguous()
        self._steps_issued += 1
        with torch.cuda.device(self.device):
            _native.check(self.L.solorl_step(self._h, C.c_void_p(a.data_ptr()), C.c_void_p(self._obs.data_ptr()),
                                             C.c_void_p(self._rew.data_ptr()), C.c_void_p(self._done.data_ptr()),
                                             C.byref(self._info_c), self._stream()))
        t = {k: v.clone() for k, v in self._info.items()}
        t["done"] = self._done.clone()
        return self._obs.clone(), self._rew.clone().unsqueeze(-1), t["done"].float(), LazyInfos(t, t["done"])

    def _raw(self, name, x, shape):
        """The C ABI takes raw pointers: `x` must already be float32, contiguous, of `shape`, on this env's device."""
        if not (x.is_cuda and x.device == self.device and x.dtype == torch.float32 and x.is_contiguous() and tuple(x.shape) == shape):
            raise AssertionError("%s must be a contiguous float32 %s tensor on %s, got %s %s on %s" % (
                name, list(shape), self.device, x.dtype, tuple(x.shape), x.device))
        return C.c_void_p(x.data_ptr())

    def step_inplace(self, actions, obs_out=None, rew_out=None, done_out=None):
        """Zero-copy variant for rollout loops: returns views of the engine-owned output buffers (overwritten by the next
        step) -- or writes observations [N, O] / rewards [N] or [N, 1] / done flags (uint8 [N]) straight into caller-owned tensors
        such as rollout-storage rows (the C ABI takes the caller's pointers anyway).  Host-side checks only, capturable in a HIP graph."""
        a = self._raw("actions", actions, (self.nenvs, self.act_dim))
        o = self._obs if obs_out is None else obs_out
        r = self._rew if rew_out is None else rew_out
        po = self._raw("obs_out", o, (self.nenvs, self.obs_dim))
        pr = self._raw("rew_out", r, tuple(r.shape) if tuple(r.shape) in ((self.nenvs,), (self.nenvs, 1)) else (self.nenvs,))
        d = self._done if done_out is None else done_out
        if not (d.is_cuda and d.device == self.device and d.dtype == torch.uint8 and d.is_contiguous() and tuple(d.shape) == (self.nenvs,)):
            raise AssertionError("done_out must be a contiguous uint8 [%d] tensor on %s" % (self.nenvs, self.device))
        self._steps_issued += 1
        with torch.cuda.device(self.device):
            _native.check(self.L.solorl_step(self._h, a, po, pr, C.c_void_p(d.data_ptr()), C.byref(self._info_c), self._stream()))
        return o, r, d, self._info

    def step_act_supported(self, params):
        """solorl_step_act (include/solorl.h): the engine's defaults (fp32, team mode, no sorting) and an observation size that is a
        multiple of 4 floats and at most 88 (zero or one history level) with the reference's hidden-64 MLP on it."""
        import os
        return (self.cfg.precision == 0 and os.environ.get("SOLORL_TEAM", "1") != "0" and os.environ.get("SOLORL_SORT", "0") == "0"
                and self.obs_dim % 4 == 0 and self.obs_dim <= 88 and params.obs_dim == self.obs_dim and params.act_dim == self.act_dim and params.hidden == 64)

    def step_act_inplace(self, actions, params, noise, value_out, action_out, logp_out, obs_out=None, rew_out=None, done_out=None):
        """step_inplace + Policy.act on the new observations in ONE launch (solorl_step_act): value_out [N] or [N,1], action_out [N,A] =
        mean + exp(logstd) * noise (noise [N,A] or None), logp_out [N] or [N,1] -- e.g. the NEXT step's rollout-storage rows."""
        a = self._raw("actions", actions, (self.nenvs, self.act_dim))
        o = self._obs if obs_out is None else obs_out
        r = self._rew if rew_out is None else rew_out
        po = self._raw("obs_out", o, (self.nenvs, self.obs_dim))
        pr = self._raw("rew_out", r, tuple(r.shape) if tuple(r.shape) in ((self.nenvs,), (self.nenvs, 1)) else (self.nenvs,))
        d = self._done if done_out is None else done_out
        if not (d.is_cuda and d.device == self.device and d.dtype == torch.uint8 and d.is_contiguous() and tuple(d.shape) == (self.nenvs,)):
            raise AssertionError("done_out must be a contiguous uint8 [%d] tensor on %s" % (self.nenvs, self.device))
        pv = self._raw("value_out", value_out, tuple(value_out.shape) if tuple(value_out.shape) in ((self.nenvs,), (self.nenvs, 1)) else (self.nenvs,))
        pa = self._raw("action_out", action_out, (self.nenvs, self.act_dim))
        pl = self._raw("logp_out", logp_out, tuple(logp_out.shape) if tuple(logp_out.shape) in ((self.nenvs,), (self.nenvs, 1)) else (self.nenvs,))
        pn = C.c_void_p(0) if noise is None else self._raw("noise", noise, (self.nenvs, self.act_dim))
        self._steps_issued += 1
        with torch.cuda.device(self.device):
            _native.check(self.L.solorl_step_act(self._h, a, po, pr, C.c_void_p(d.data_ptr()), C.byref(self._info_c), C.byref(params), pn, pv, pa, pl,
                                                 self._stream()))
        return o, r, d, self._info

    def get_observation(self):
        with torch.cuda.device(self.device):
            _native.check(self.L.solorl_get_observation(self._h, C.c_void_p(self._obs.data_ptr()), self._stream()))
        return self._obs.clone()

    def episode_stat_sums(self, reset=True):
        """Device tensor [SOLORL_EPSTAT_FIELDS]: the accumulators summed over this handle's envs (all-reduce it across
        ranks before dividing by field 0)."""
        tot = self._ep_stats.sum(dim=1)
        if reset:
            self._ep_stats.zero_()
        return tot

    def episode_stats(self, reset=True):
        """Statistics of the episodes finished since the last call, reduced on the device from the engine's
        per-env accumulators (every step counts, nothing is sampled): the quantities the reference collects by
        iterating N info dicts per step (agents/ppo/train.py:90-100) -- means of info['episode_reward'] (the LAST
        step's reward, baseEnv.py:65), info['episode_length'], info['success'] and the info['dr/*'] sums over the
        finished episodes -- plus their count.  One small device->host copy; ``reset`` zeroes the accumulators."""
        tot = self.episode_stat_sums(reset).tolist()
        n = tot[0]
        out = {"episodes": int(n), "nan_resets": int(tot[9])}
        for k in range(1, 9):
            out[EPSTAT_NAMES[k]] = (tot[k] / n) if n > 0 else float("nan")
        return out

    def record_applied_torque(self):
        """Every step also writes the joint torques it applied (after the clip / PD law, solo.py:224-259) into the returned [N, A]
        tensor (overwritten by the next step).  Must be enabled before the first step (or with the constructor's
        ``applied_torque=True``): the destination is a kernel argument, so a rollout or bench graph captured earlier would keep
        replaying without it -- enabling it late raises instead of being silently ignored by those graphs."""
        if self._tau is None:
            if self._steps_issued:
                raise _native.SoloRLError("record_applied_torque() after %d steps: step launches (and any HIP graph captured from them) already "
                                          "carry a NULL applied_torque pointer; construct the env with applied_torque=True" % self._steps_issued)
            self._tau = torch.zeros((self.nenvs, self.act_dim), dtype=torch.float32, device=self.device)
            self._info_c.applied_torque = self._tau.data_ptr()
        return self._tau

    def increment_curriculum(self, value=1.0):
        _native.check(self.L.solorl_increment_curriculum(self._h, float(value)))

    def get_property(self, name):
        """Read-only handle properties (include/solorl.h solorl_get_property): lanes_per_env, sweep_variant, max_contacts, ..."""
        v = C.c_double()
        _native.check(self.L.solorl_get_property(self._h, name.encode(), C.byref(v)))
        return v.value

    def get_state(self, i=0):
        s = EnvState()
        _native.check(self.L.solorl_get_state(self._h, int(i), C.byref(s)))
        return s

    def set_state(self, i, s):
        _native.check(self.L.solorl_set_state(self._h, int(i), C.byref(s)))

    def close(self):
        if not self.closed and getattr(self, "_h", None):
            self.L.solorl_destroy(self._h)
            self._h = None
            self.closed = True

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def make_vec_envs(config, num_envs, env_constructor=None, gamma=0.99, device=None, training=True, seed=1,
                  env_id_offset=0):
    """Signature of agents/ppo/envs.py:14.  ``env_constructor`` (the per-process gym env class of the
    reference) and ``gamma`` (VecNormalize's unused return accumulator, ob=False, ret=False) are
    accepted for call-site compatibility and ignored."""
    return SoloVecEnv(config, num_envs, device=device, seed=seed, env_id_offset=env_id_offset)
