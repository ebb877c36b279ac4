"""On-device rollout storage -- surface of the reference's OPBuffer (agents/ppo/storage.py:5-71):
tensors obs[T+1,N,O], rewards/value_preds/returns/action_log_probs[.,N,1], actions[T,N,A],
masks[T+1,N,1]; append / reset / compute_returns (GAE and plain discounted) / batch_generator.

Everything lives on the rollout device, so a 4096-env x 400-step buffer (~0.55 GB for Solo12
pointGoal) never crosses PCIe; mini-batches are gathered with one index_select per tensor."""
import torch


class RolloutStorage:
    def __init__(self, num_steps, num_agents, obs_shape, action_dim, device):
        z = lambda *s: torch.zeros(*s, device=device)
        self.obs = z(num_steps + 1, num_agents, *obs_shape)
        self.rewards = z(num_steps, num_agents, 1)
        self.value_preds = z(num_steps + 1, num_agents, 1)
        self.returns = z(num_steps + 1, num_agents, 1)
        self.action_log_probs = z(num_steps, num_agents, 1)
        self.actions = z(num_steps, num_agents, action_dim)
        self.masks = torch.ones(num_steps + 1, num_agents, 1, device=device)
        self.num_steps, self.num_agents = num_steps, num_agents
        self.num_samples = num_steps * num_agents
        self.device = device
        self.step = 0

    def append(self, obs, actions, action_log_probs, value_preds, rewards, masks):
        t = self.step
        self.obs[t + 1].copy_(obs)
        self.actions[t].copy_(actions)
        self.action_log_probs[t].copy_(action_log_probs)
        self.value_preds[t].copy_(value_preds)
        self.rewards[t].copy_(rewards)
        self.masks[t + 1].copy_(masks)
        self.step = (t + 1) % self.num_steps

    def reset(self):
        self.obs[0].copy_(self.obs[-1])
        self.masks[0].copy_(self.masks[-1])

    @torch.no_grad()
    def compute_returns(self, next_value, use_gae=True, gamma=0.99, gae_lambda=0.95):
        """storage.py:35-55.  GAE: delta_t = r_t + g V_{t+1} m_{t+1} - V_t,
        A_t = delta_t + g l m_{t+1} A_{t+1}, R_t = A_t + V_t;  else R_t = r_t + g m_{t+1} R_{t+1}."""
        T = self.num_steps
        if self.rewards.is_cuda:      # fused HIP kernel; raises if the engine library is missing
            import ctypes as C
            from .. import _native
            dev = self.rewards.device
            nv = next_value.detach().to(torch.float32).contiguous()
            with torch.cuda.device(dev):
                _native.check(_native.lib().solorl_compute_returns(
                    C.c_void_p(self.rewards.data_ptr()), C.c_void_p(self.value_preds.data_ptr()),
                    C.c_void_p(self.masks.data_ptr()), C.c_void_p(nv.data_ptr()), C.c_void_p(self.returns.data_ptr()),
                    T, self.num_agents, int(bool(use_gae)), float(gamma), float(gae_lambda), dev.index or 0,
                    C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
            return
        if use_gae:
            self.value_preds[-1].copy_(next_value)
            v, m = self.value_preds, self.masks
            delta = self.rewards + gamma * v[1:] * m[1:] - v[:-1]       # one fused elementwise pass
            coef = (gamma * gae_lambda) * m[1:]
            adv = torch.zeros_like(next_value)
            for t in range(T - 1, -1, -1):                                # the recurrence itself is sequential in t
                adv = torch.addcmul(delta[t], coef[t], adv)
                self.returns[t] = adv + v[t]
        else:
            self.returns[-1].copy_(next_value)
            gm = gamma * self.masks[1:]
            for t in range(T - 1, -1, -1):
                self.returns[t] = torch.addcmul(self.rewards[t], gm[t], self.returns[t + 1])

    def batch_generator(self, advantages, mini_batch_size, generator=None):
        """storage.py:57-71: random mini-batches without replacement, incomplete last batch dropped."""
        n = self.num_samples
        perm = torch.randperm(n, device=self.device, generator=generator)
        flat = lambda x: x.reshape(n, *x.shape[2:])
        obs, act, val = flat(self.obs[:-1]), flat(self.actions), flat(self.value_preds[:-1])
        ret, msk, olp, adv = flat(self.returns[:-1]), flat(self.masks[:-1]), flat(self.action_log_probs), advantages.reshape(n, 1)
        for s in range(0, n - mini_batch_size + 1, mini_batch_size):
            idx = perm[s:s + mini_batch_size]
            yield obs[idx], act[idx], val[idx], ret[idx], msk[idx], olp[idx], adv[idx]


OPBuffer = RolloutStorage   # the reference's class name
