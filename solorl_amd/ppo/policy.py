"""Actor-critic policy -- interface of the reference's agents/ppo/policy.py (Policy.act /
get_value / evaluate_actions, :33-58) with parameter names kept identical so `solo.pt` checkpoints
(agents/ppo/train.py:121-131: {'update','state_dict','ob_rms'}) load either way:

    base.features.{0,2}.{weight,bias}   tanh MLP trunk of the actor          (policy.py:66-68)
    base.critic.{0,2,4}.{weight,bias}   separate tanh MLP critic             (policy.py:70-73)
    pi_dist.mean.{weight,bias}, pi_dist.logstd   state-independent diagonal Gaussian (:138-148)

Only Box action spaces exist on the hot path (baseEnv.py:23-25); the reference's Discrete /
MultiDiscrete heads cannot even be constructed at that commit.
"""
import math

import torch
import torch.nn as nn


class _LinearSplitK(torch.autograd.Function):
    """y = x W^T + b whose weight gradient is reduced in independent slices of the batch.

    A PPO mini-batch is tens of thousands of rows through 64-wide layers: dW = g^T x is a (64 x 76) product with a
    32 768-long reduction -- two or three output tiles, i.e. two or three workgroups on a 256-CU device (measured on an
    MI355X: 133 us per layer, two thirds of a mini-batch step).  Cut into S slices it is one batched GEMM over S x as many
    workgroups plus a small sum; the arithmetic is the same sum in a different order."""
    SLICE_ROWS = 512

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        return torch.addmm(b, x, w.t())

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        gx = g.mm(w) if ctx.needs_input_grad[0] else None
        m, r = x.shape[0], _LinearSplitK.SLICE_ROWS
        s = m // r
        if s >= 2:
            m0 = s * r
            gw = torch.bmm(g[:m0].reshape(s, r, -1).transpose(1, 2), x[:m0].reshape(s, r, -1)).sum(0)
            if m0 < m:
                gw = gw + g[m0:].t().mm(x[m0:])
        else:
            gw = g.t().mm(x)
        return gx, gw, g.sum(0)


class Linear(nn.Linear):
    """nn.Linear (same parameter names, same forward) with the split reduction above for training batches."""
    SPLIT_MIN_ROWS = 4096

    def forward(self, x):
        if x.dim() == 2 and x.shape[0] >= Linear.SPLIT_MIN_ROWS and torch.is_grad_enabled() and self.bias is not None:
            return _LinearSplitK.apply(x, self.weight, self.bias)
        return super().forward(x)


def ortho_(linear, gain=math.sqrt(2.0)):
    """agents/utils.py:120-131 (init_layer): orthogonal weights with gain sqrt(2), zero bias."""
    nn.init.orthogonal_(linear.weight.data, gain=gain)
    if linear.bias is not None:
        nn.init.zeros_(linear.bias.data)
    return linear


class MLPBase(nn.Module):
    def __init__(self, num_inputs, hidden_size=64):
        super().__init__()
        self.output_size = hidden_size
        h = hidden_size
        self.features = nn.Sequential(ortho_(Linear(num_inputs, h)), nn.Tanh(), ortho_(Linear(h, h)), nn.Tanh())
        self.critic = nn.Sequential(ortho_(Linear(num_inputs, h)), nn.Tanh(), ortho_(Linear(h, h)), nn.Tanh(),
                                    ortho_(Linear(h, 1)))

    def forward(self, x):
        return self.critic(x), self.features(x)


class DiagGaussianHead(nn.Module):
    def __init__(self, num_inputs, num_outputs):
        super().__init__()
        self.mean = ortho_(Linear(num_inputs, num_outputs))
        self.logstd = nn.Parameter(torch.zeros(num_outputs))

    def forward(self, feat):
        return self.mean(feat), self.logstd


_HALF_LOG_2PI = 0.5 * math.log(2.0 * math.pi)


def gaussian_log_prob(action, mean, logstd):
    """sum_j log N(a_j; mu_j, exp(logstd_j)), keepdim -- ModNormal.log_probs (policy.py:171-173)."""
    z = (action - mean) * torch.exp(-logstd)
    return (-0.5 * z * z - logstd - _HALF_LOG_2PI).sum(-1, keepdim=True)


def gaussian_entropy_mean(mean, logstd):
    """`dist.entropy().mean()` of the reference (policy.py:56): mean over batch AND action dims."""
    return (0.5 + _HALF_LOG_2PI + logstd).expand_as(mean).mean()


class Policy(nn.Module):
    def __init__(self, obs_shape, action_space, base=None, base_kwargs=None):
        super().__init__()
        if len(obs_shape) != 1:
            raise NotImplementedError("only flat observations (the TransformerBase of policy.py:83-120 serves the "
                                      "out-of-scope Timings envs)")
        if action_space.__class__.__name__ != "Box":
            raise NotImplementedError("only Box action spaces (baseEnv.py:23-25)")
        self.base = MLPBase(obs_shape[0], **(base_kwargs or {}))
        if base is not None:                       # --base-checkpoint (agents/ppo/train.py:44-45)
            self.base.load_state_dict(base)
        self.pi_dist = DiagGaussianHead(self.base.output_size, action_space.shape[0])

    def act(self, inputs, deterministic=False):
        value, feat = self.base(inputs)
        mean, logstd = self.pi_dist(feat)
        action = mean if deterministic else mean + torch.exp(logstd) * torch.randn_like(mean)
        return value, action, gaussian_log_prob(action, mean, logstd)

    @torch.no_grad()
    def act_into(self, inputs, value_out, action_out, logp_out, noise=None):
        """`act` for rollout loops that own their buffers (ppo/storage.py): the same arithmetic written straight into
        value_out [N,1], action_out [N,A] and logp_out [N,1] -- about half the launches of act() + three copies, which is what
        a captured rollout step costs besides the env kernel.  `noise` (optional, [N,A]) replaces the standard-normal draw."""
        c, f = self.base.critic, self.base.features
        lin = lambda l, x, out=None: torch.addmm(l.bias, x, l.weight.t(), out=out)
        lin(c[4], torch.tanh_(lin(c[2], torch.tanh_(lin(c[0], inputs)))), out=value_out)
        mean = lin(self.pi_dist.mean, torch.tanh_(lin(f[2], torch.tanh_(lin(f[0], inputs)))))
        logstd = self.pi_dist.logstd
        if noise is None:
            noise = torch.randn_like(mean)
        torch.addcmul(mean, torch.exp(logstd), noise, out=action_out)
        z = (action_out - mean).mul_(torch.exp(-logstd))
        torch.sum(z.mul_(z).mul_(-0.5).sub_(logstd).sub_(_HALF_LOG_2PI), -1, keepdim=True, out=logp_out)

    def heads(self, inputs):
        """(value, mean, logstd): what the fused loss kernel (ppo/fused.py) consumes."""
        value, feat = self.base(inputs)
        mean, logstd = self.pi_dist(feat)
        return value, mean, logstd

    def get_value(self, inputs):
        return self.base.critic(inputs)

    def evaluate_actions(self, inputs, action):
        value, feat = self.base(inputs)
        mean, logstd = self.pi_dist(feat)
        return value, gaussian_log_prob(action, mean, logstd), gaussian_entropy_mean(mean, logstd)
