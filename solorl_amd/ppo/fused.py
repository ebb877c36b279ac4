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
