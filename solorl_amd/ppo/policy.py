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
        self.features = nn.Sequential(ortho_(nn.Linear(num_inputs, h)), nn.Tanh(), ortho_(nn.Linear(h, h)), nn.Tanh())
        self.critic = nn.Sequential(ortho_(nn.Linear(num_inputs, h)), nn.Tanh(), ortho_(nn.Linear(h, h)), nn.Tanh(),
                                    ortho_(nn.Linear(h, 1)))

    def forward(self, x):
        return self.critic(x), self.features(x)


class DiagGaussianHead(nn.Module):
    def __init__(self, num_inputs, num_outputs):
        super().__init__()
        self.mean = ortho_(nn.Linear(num_inputs, num_outputs))
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
