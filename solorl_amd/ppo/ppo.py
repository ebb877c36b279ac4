"""Clipped-surrogate PPO update -- arithmetic of the reference's agents/ppo/ppo.py:34-89:
whole-batch advantage normalisation (:35-37), ratio clipping (:54-59), clipped value loss (:61-68),
loss = value*c_v + action - entropy*c_e (:73-74), grad-norm clip (:75-76), Adam (:32, :77).

Multi-GPU: each rank holds N_local envs and a policy replica; gradients are summed across ranks
in ONE flat bucket between backward() and clip_grad_norm_, and the advantage
mean/std are computed over the global T x N batch with a 3-float all-reduce, so W ranks x N envs
reproduce the reference's single-process update over W*N envs."""
import torch
import torch.nn as nn

from . import dist as D


class PPO:
    def __init__(self, actor_critic, clip_param, ppo_epoch, mini_batch_size, value_loss_coef, entropy_coef, lr=None,
                 l2_coef=0.0, max_grad_norm=None, use_clipped_value_loss=True):
        self.actor_critic = actor_critic
        self.clip_param, self.ppo_epoch, self.mini_batch_size = clip_param, ppo_epoch, mini_batch_size
        self.value_loss_coef, self.entropy_coef = value_loss_coef, entropy_coef
        self.max_grad_norm, self.use_clipped_value_loss = max_grad_norm, use_clipped_value_loss
        self.optimizer = torch.optim.Adam(actor_critic.parameters(), lr=lr, weight_decay=l2_coef)
        self.bucket = D.FlatGradBucket(actor_critic.parameters())

    def update(self, storage):
        adv = storage.returns[:-1] - storage.value_preds[:-1]
        mean, std = D.global_mean_std(adv, unbiased=True)
        adv = (adv - mean) / (std + 1e-5)

        stats = torch.zeros(3, device=adv.device)
        n_updates = 0
        clip = self.clip_param
        for _ in range(self.ppo_epoch):
            for obs_b, act_b, vpred_b, ret_b, _mask_b, old_lp_b, adv_b in storage.batch_generator(adv, self.mini_batch_size):
                values, logp, entropy = self.actor_critic.evaluate_actions(obs_b, act_b)
                ratio = torch.exp(logp - old_lp_b)
                action_loss = -torch.min(ratio * adv_b, torch.clamp(ratio, 1.0 - clip, 1.0 + clip) * adv_b).mean()
                if self.use_clipped_value_loss:
                    v_clipped = vpred_b + (values - vpred_b).clamp(-clip, clip)
                    value_loss = 0.5 * torch.max((values - ret_b).pow(2), (v_clipped - ret_b).pow(2)).mean()
                else:
                    value_loss = 0.5 * (ret_b - values).pow(2).mean()
                self.bucket.zero()
                (value_loss * self.value_loss_coef + action_loss - entropy * self.entropy_coef).backward()
                self.bucket.all_reduce_mean()                  # RCCL: one 80 KB collective per optimizer step
                if self.max_grad_norm is not None:
                    nn.utils.clip_grad_norm_(self.actor_critic.parameters(), self.max_grad_norm)
                self.optimizer.step()
                stats += torch.stack([value_loss.detach(), action_loss.detach(), entropy.detach()])   # no host sync
                n_updates += 1
        # the reference divides by ppo_epoch * (num_samples // mini_batch_size) (:83) -- identical count
        v, a, e = (stats / max(n_updates, 1)).tolist()
        return v, a, e
