"""Fused PPO mini-batch loss (HIP kernel `solorl_ppo_loss`, include/solorl.h): the Gaussian log-prob, ratio,
clipped surrogate, clipped value loss and their gradients w.r.t. the policy heads in ONE launch instead of ~60
elementwise torch kernels (forward + autograd backward) per mini-batch.  Arithmetic of agents/ppo/ppo.py:52-74 and
agents/ppo/policy.py:51-58,171-173; torch's sub-gradient conventions for min/max/clamp.  GPU only."""
import ctypes as C
import math

import torch

from .. import _native

_HALF_LOG_2PI = 0.5 * math.log(2.0 * math.pi)


class _FusedPPOLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mean, logstd, values, action, old_logp, adv, vpred, ret, clip, value_coef, entropy_coef, clipped_value):
        m, A = mean.shape
        dev = mean.device
        f = lambda x: x.detach().contiguous().float()
        mean_, logstd_, values_, action_ = f(mean), f(logstd), f(values).view(-1), f(action)
        old_, adv_, vpred_, ret_ = f(old_logp).view(-1), f(adv).view(-1), f(vpred).view(-1), f(ret).view(-1)
        nb = (m + 255) // 256
        g_mean = torch.empty_like(mean_)
        g_values = torch.empty_like(values_)
        part = torch.empty((nb, 3 + A), device=dev, dtype=torch.float32)
        p = lambda x: C.c_void_p(x.data_ptr())
        with torch.cuda.device(dev):
            _native.check(_native.lib().solorl_ppo_loss(p(mean_), p(logstd_), p(values_), p(action_), p(old_), p(adv_), p(vpred_), p(ret_),
                                                        m, A, float(clip), float(value_coef), int(bool(clipped_value)), p(g_mean), p(g_values),
                                                        p(part), dev.index or 0, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        s = part.sum(0)
        value_loss, action_loss = s[0] / m, s[1] / m
        entropy = (0.5 + _HALF_LOG_2PI + logstd_).mean()
        loss = value_loss * value_coef + action_loss - entropy * entropy_coef
        ctx.save_for_backward(g_mean, g_values, s[3:] - entropy_coef / A)
        ctx.shapes = (mean.shape, logstd.shape, values.shape)
        ctx.mark_non_differentiable(value_loss, action_loss, entropy)
        return loss, value_loss, action_loss, entropy

    @staticmethod
    def backward(ctx, g, *_unused):
        g_mean, g_values, g_logstd = ctx.saved_tensors
        ms, ls, vs = ctx.shapes
        return (g_mean.view(ms) * g, g_logstd.view(ls) * g, g_values.view(vs) * g) + (None,) * 9


def fused_ppo_loss(mean, logstd, values, action, old_logp, adv, vpred, ret, clip, value_coef, entropy_coef, clipped_value=True):
    """-> (loss, value_loss, action_loss, entropy); loss = value_loss*value_coef + action_loss - entropy*entropy_coef."""
    return _FusedPPOLoss.apply(mean, logstd, values, action, old_logp, adv, vpred, ret, clip, value_coef, entropy_coef, clipped_value)


# ---------------------------------------------------------------------------------------------------------------------
# Hand-written policy kernels (csrc/solorl_ppo.hip): the whole forward of Policy.act in one launch, and one PPO mini-batch's
# gather + forward + losses + back-propagation in one launch (the weight gradients stay GEMMs).

_SUPPORTED = {(76, 12), (84, 12), (38, 12), (42, 12), (60, 8), (68, 8), (30, 8), (34, 8)}


def policy_kernels_supported(actor_critic):
    """The kernels are built for the reference's default MLP (hidden 64, agents/ppo/policy.py:62-81) on zero or one history level."""
    b = actor_critic.base
    try:
        o, h, a = b.features[0].in_features, b.features[0].out_features, actor_critic.pi_dist.mean.out_features
    except AttributeError:
        return False
    p = next(actor_critic.parameters())
    return h == 64 and (o, a) in _SUPPORTED and p.is_cuda and p.dtype == torch.float32


def policy_params(actor_critic):
    """solorl_policy_params over the module's own parameter storage (PyTorch layout is the kernels' layout: no copies).
    The pointers stay valid as long as the parameters are updated in place (optimizers do)."""
    b, d = actor_critic.base, actor_critic.pi_dist
    P = _native.PolicyParams()
    P.obs_dim, P.hidden, P.act_dim = b.features[0].in_features, b.features[0].out_features, d.mean.out_features
    named = {"critic_w0": b.critic[0].weight, "critic_b0": b.critic[0].bias, "critic_w1": b.critic[2].weight, "critic_b1": b.critic[2].bias,
             "critic_w2": b.critic[4].weight, "critic_b2": b.critic[4].bias, "actor_w0": b.features[0].weight, "actor_b0": b.features[0].bias,
             "actor_w1": b.features[2].weight, "actor_b1": b.features[2].bias, "mean_w": d.mean.weight, "mean_b": d.mean.bias,
             "logstd": d.logstd}
    for k, t in named.items():
        assert t.is_contiguous() and t.dtype == torch.float32 and t.is_cuda
        setattr(P, k, t.data_ptr())
    P._tensors = named            # keeps the storages alive and lets check_params() notice a re-allocated parameter
    return P


def check_params(P):
    """The kernels read the parameters through raw pointers taken once: a parameter that was re-allocated since (module.to(),
    load_state_dict(assign=True), p.data = ...) would be read from freed memory.  Host-side check, so it guards eager calls and
    graph CAPTURES; a graph replayed after such a change cannot be guarded and must be rebuilt."""
    for k, t in P._tensors.items():
        if getattr(P, k) != t.data_ptr():
            raise RuntimeError("policy parameter %s moved since solorl_policy_params was built; rebuild it (and any HIP graph)" % k)


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def policy_act(params, obs, noise, value_out, action_out, logp_out):
    """Policy.act (agents/ppo/policy.py:33-49) in one launch: value_out [N,1], action_out [N,A] = mean + exp(logstd) * noise
    (noise [N,A] standard normal, or None for the deterministic action), logp_out [N,1]."""
    check_params(params)
    n = obs.shape[0]
    dev = obs.device
    for t in (obs, value_out, action_out, logp_out) + (() if noise is None else (noise,)):
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
    assert obs.shape == (n, params.obs_dim) and action_out.shape == (n, params.act_dim) and value_out.numel() == n and logp_out.numel() == n
    with torch.cuda.device(dev):
        _native.check(_native.lib().solorl_policy_act(C.byref(params), C.c_void_p(obs.data_ptr()),
                                                      C.c_void_p(0 if noise is None else noise.data_ptr()), n, C.c_void_p(value_out.data_ptr()),
                                                      C.c_void_p(action_out.data_ptr()), C.c_void_p(logp_out.data_ptr()), dev.index or 0, _stream(dev)))


class MiniBatchGrad:
    """One PPO mini-batch step's gradients without autograd, three launches: solorl_ppo_grad_stage1 (gather, forward, losses,
    back-propagation to every layer's pre-activation gradient, written in tiles [row / 32][unit][32]), then solorl_ppo_grad_stage2 (the weight
    gradients G^T X of the six layers over row chunks, and their fixed-order sum written straight into the parameters' .grad --
    views of the flat bucket -- with the log-std gradient and the running loss sums).  Same arithmetic as PPO.update's
    loss.backward() (agents/ppo/ppo.py:46-74) in a different summation order."""
    SLICE = 64                        # mini-batch rows must fill whole wavefronts

    def __init__(self, actor_critic, storage, mini_batch, clip, value_coef, entropy_coef, clipped_value, perm, offset, adv):
        n, m = storage.num_samples, mini_batch
        assert m % self.SLICE == 0, "mini-batch must be a multiple of %d rows" % self.SLICE
        dev = storage.device
        self.ac, self.m = actor_critic, m
        self.P = policy_params(actor_critic)
        O, A, H = self.P.obs_dim, self.P.act_dim, 64
        self.A = A
        flat = lambda x: x.reshape(n, *x.shape[2:])
        B = self.B = _native.PpoBatch()
        self._keep = (flat(storage.obs[:-1]), flat(storage.actions), flat(storage.action_log_probs), adv, flat(storage.value_preds[:-1]),
                      flat(storage.returns[:-1]), perm, offset)
        for k, t in zip(("obs", "actions", "old_logp", "adv", "vpred", "ret", "perm", "offset"), self._keep):
            assert t.is_contiguous() and t.is_cuda
            setattr(B, k, t.data_ptr())
        assert self._keep[0].data_ptr() == storage.obs.data_ptr() and perm.dtype == torch.long and offset.dtype == torch.long
        B.m, B.clipped_value, B.clip, B.value_coef = m, int(bool(clipped_value)), float(clip), float(value_coef)

        z = lambda u: torch.zeros(u, m, device=dev)
        xt = z                                # (activations [units][m]; the bias gradients are row sums of g, no row of ones)
        self.buf = dict(xt0=xt(O), c_xt1=xt(H), c_xt2=xt(H), c_g1=z(H), c_g2=z(H), c_gh=z(1), a_xt1=xt(H), a_xt2=xt(H), a_g1=z(H), a_g2=z(H),
                        a_gh=z(A), partials=torch.zeros((m + 31) // 32, 3 + A, device=dev))
        W = self.W = _native.PpoStage1()
        for k, t in self.buf.items():
            setattr(W, k, t.data_ptr())
        b, d = actor_critic.base, actor_critic.pi_dist
        self.psum = torch.zeros(3 + A, device=dev)          # running sums of the partials over the steps of an update
        self.ent = torch.zeros(1, device=dev)               # running sum of sum_a logstd_a
        L = _native.lib()
        self.scratch = torch.zeros(L.solorl_ppo_scratch_count(O, A, m), device=dev)
        G = self.G = _native.PpoGrads()
        grads = {"critic_w0": b.critic[0].weight, "critic_b0": b.critic[0].bias, "critic_w1": b.critic[2].weight, "critic_b1": b.critic[2].bias,
                 "critic_w2": b.critic[4].weight, "critic_b2": b.critic[4].bias, "actor_w0": b.features[0].weight, "actor_b0": b.features[0].bias,
                 "actor_w1": b.features[2].weight, "actor_b1": b.features[2].bias, "mean_w": d.mean.weight, "mean_b": d.mean.bias,
                 "logstd": d.logstd}
        for k, prm in grads.items():
            g = prm.grad
            assert g is not None and g.is_contiguous() and g.shape == prm.shape, "parameters need dense .grad tensors (FlatGradBucket)"
            setattr(G, k, g.data_ptr())
        G.loss_sums, G.logstd_sum, G.scratch, G.entropy_coef = self.psum.data_ptr(), self.ent.data_ptr(), self.scratch.data_ptr(), float(entropy_coef)

    def __call__(self):
        check_params(self.P)
        dev = self.psum.device
        L = _native.lib()
        with torch.cuda.device(dev):
            st = _stream(dev)
            _native.check(L.solorl_ppo_grad_stage1(C.byref(self.P), C.byref(self.B), C.byref(self.W), dev.index or 0, st))
            _native.check(L.solorl_ppo_grad_stage2(C.byref(self.P), C.byref(self.W), self.m, C.byref(self.G), dev.index or 0, st))

    def losses(self, steps):
        """(value loss, action loss, entropy) averaged over `steps` mini-batch steps since reset()."""
        v, a = (self.psum[:2] / (self.m * max(steps, 1))).tolist()
        e = 0.5 + _HALF_LOG_2PI + self.ent.item() / (self.A * max(steps, 1))
        return v, a, e

    def reset(self):
        self.psum.zero_(); self.ent.zero_()


class ClipAdam:
    """clip_grad_norm_ + Adam (torch.optim.Adam's arithmetic, agents/ppo/ppo.py:32,75-77) for the MLP policy in ONE launch
    (solorl_ppo_clip_adam) instead of nine; also advances the mini-batch cursor.  The learning rate is read from `lr`, a device
    tensor (the linear schedule rewrites it between updates); moments and the step count live here."""

    def __init__(self, mini_batch_grad, lr, max_grad_norm, weight_decay=0.0, betas=(0.9, 0.999), eps=1e-8, offset=None, offset_increment=0, grad_scale=1.0):
        mb = mini_batch_grad
        dev = mb.psum.device
        n = _native.lib().solorl_ppo_grad_count(mb.P.obs_dim, mb.P.act_dim) + mb.P.act_dim
        assert n == sum(p.numel() for p in mb.ac.parameters())
        self.P, self.G, self.dev = mb.P, mb.G, dev
        self.exp_avg, self.exp_avg_sq = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
        self.step = torch.zeros(1, device=dev)
        assert torch.is_tensor(lr) and lr.is_cuda and lr.dtype == torch.float32
        self.lr = lr
        S = self.S = _native.AdamState()
        S.exp_avg, S.exp_avg_sq, S.step, S.lr = self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), self.step.data_ptr(), lr.data_ptr()
        S.offset, S.offset_increment = (offset.data_ptr() if offset is not None else 0), int(offset_increment)
        S.beta1, S.beta2, S.eps, S.weight_decay = float(betas[0]), float(betas[1]), float(eps), float(weight_decay)
        S.max_grad_norm = float(max_grad_norm) if max_grad_norm is not None else 0.0
        S.grad_scale = float(grad_scale)     # 1 / world: the gradients arrive SUMMED over the ranks (FlatGradBucket.all_reduce_sum)
        self._offset = offset

    def __call__(self):
        with torch.cuda.device(self.dev):
            _native.check(_native.lib().solorl_ppo_clip_adam(C.byref(self.P), C.byref(self.G), C.byref(self.S), self.dev.index or 0, _stream(self.dev)))

    def zero_state(self):
        self.exp_avg.zero_(); self.exp_avg_sq.zero_(); self.step.zero_()
