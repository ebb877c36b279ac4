"""PPO training loop over the HIP rollout engine -- `train(args, config, env_constructor, writer)`
with the argument meaning of the reference's agents/ppo/train.py:21-162 and the flags of
training/train_ppo.py:9-45.

Differences that come with the GPU engine (documented, not behavioural):
  * rollout, policy forward, storage and update all stay on one HIP device: no device->host->pickle
    ->Pipe round trip per step (agents/ppo/envs.py:189-196);
  * episode statistics are reduced on the device from the SoA info tensors instead of iterating N
    Python dicts per step (train.py:90-100) -- same quantities: episode_reward (last-step reward,
    baseEnv.py:65), episode_length, success, dr/* sums;
  * multi-GPU: one process per GPU (`python -m torch.distributed.run --nproc-per-node W ...`), envs
    sharded by rank (global env id = rank*N + i), flat-bucket RCCL gradient all-reduce (ppo.py).
"""
import os
import time
from collections import deque

import torch

from . import dist as D
from .policy import Policy
from .ppo import PPO
from .graphs import GraphedPPO, GraphedRollout
from .storage import RolloutStorage


def update_linear_schedule(optimizer, epoch, total_num_epochs, initial_lr):
    """agents/utils.py:14-18: lr decays linearly to zero."""
    lr = initial_lr - (initial_lr * (epoch / float(total_num_epochs)))
    for g in optimizer.param_groups:
        if torch.is_tensor(g["lr"]):      # capturable Adam (GraphedPPO): the graph reads the device tensor
            g["lr"].fill_(lr)
        else:
            g["lr"] = lr


class EpisodeStats:
    """Rolling episode statistics (the reference keeps deques of the last 32 finished episodes,
    train.py:66-70); finished-episode values are gathered on the device, one host copy per log."""

    def __init__(self, maxlen=32):
        self.reward, self.length, self.success = deque(maxlen=maxlen), deque(maxlen=maxlen), deque(maxlen=maxlen)
        self.dr = {}
        self._pending = []
        self.finished = 0

    def push(self, done, info):
        self._pending.append((done.bool(), {k: v.clone() for k, v in info.items()}))

    def flush(self, limit=32):
        for d, info in self._pending:
            idx = d.nonzero().flatten()
            self.finished += int(idx.numel())
            if idx.numel() == 0:
                continue
            idx = idx[-limit:]
            self.reward.extend(info["episode_reward"][idx].tolist())
            self.length.extend(info["episode_length"][idx].tolist())
            self.success.extend(info["success"][idx].float().tolist())
            for k, name in (("dr_stand", "dr/stand_rew"), ("dr_joint_pose", "dr/joint_pose_rew"), ("dr_torque", "dr/torque_rew"),
                            ("dr_balance", "dr/roll_pitch_balance_rew"), ("dr_progress", "dr/progress_rew")):
                self.dr.setdefault(name, deque(maxlen=32)).extend(info[k][idx].tolist())
        self._pending = []


def rollout(envs, actor_critic, storage, num_steps, stats=None):
    """train.py:82-103: T steps of act -> env.step -> storage.append, entirely on the device."""
    for step in range(num_steps):
        with torch.no_grad():
            value, action, logp = actor_critic.act(storage.obs[step])
        obs, reward, done, info = envs.step_inplace(action.contiguous())
        if stats is not None and step % 8 == 7:       # sample the episode statistics sparsely
            stats.push(done, info)
        storage.append(obs, action, logp, value, reward.unsqueeze(-1), (1.0 - done.float()).unsqueeze(-1))


def train(args, config, env_constructor=None, writer=None):
    from ..vec_env import make_vec_envs
    rank, world = D.init_from_env("cuda" if args.cuda else "cpu")
    torch.manual_seed(args.seed)                      # identical initial weights on every rank (train.py:23)
    if args.cuda:
        torch.cuda.manual_seed_all(args.seed)
    device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0"))) if args.cuda else torch.device("cpu")
    N = args.num_agents
    envs = make_vec_envs(config, N, env_constructor, args.gamma, device, seed=args.seed, env_id_offset=rank * N)
    action_dim = envs.action_space.shape[0]
    base = torch.load(args.base_checkpoint) if getattr(args, "base_checkpoint", None) else None
    actor_critic = Policy(envs.observation_space.shape, envs.action_space, base, {"hidden_size": args.hidden_size}).to(device)
    D.broadcast_parameters(actor_critic)
    use_graphs = bool(args.cuda) and not getattr(args, "no_hip_graphs", False)
    agent = (GraphedPPO if use_graphs else PPO)(actor_critic, args.clip_param, args.ppo_epoch, args.mini_batch_size,
                                                 args.value_loss_coef, args.entropy_coef, lr=args.lr, l2_coef=args.l2_coef,
                                                 max_grad_norm=args.max_grad_norm)
    storage = RolloutStorage(args.num_steps, N, envs.observation_space.shape, action_dim, device)
    storage.obs[0].copy_(envs.reset())
    torch.manual_seed(args.seed + 1000 * rank)        # but different action noise per rank
    stats = EpisodeStats()
    if use_graphs:
        with torch.no_grad():
            actor_critic.act(storage.obs[0]); actor_critic.get_value(storage.obs[0])    # library workspaces before capture
        torch.manual_seed(args.seed + 1000 * rank)
        graphed_rollout = GraphedRollout(envs, actor_critic, storage, args.num_steps, stats)
    start = time.time()
    num_updates = int(args.num_env_steps) // args.num_steps // (N * world)
    history = []
    for j in range(num_updates):
        if args.use_linear_lr_decay:
            update_linear_schedule(agent.optimizer, j, num_updates, args.lr)
        if use_graphs:
            graphed_rollout()
        else:
            rollout(envs, actor_critic, storage, args.num_steps, stats)
        with torch.no_grad():
            next_value = actor_critic.get_value(storage.obs[-1])
        storage.compute_returns(next_value, args.use_gae, args.gamma, args.tau)
        value_loss, action_loss, entropy = agent.update(storage)
        storage.reset()
        if getattr(args, "curriculum_schedule", 0) and (j + 1) % args.curriculum_schedule == 0:
            envs.increment_curriculum()
        total = (j + 1) * N * world * args.num_steps
        if rank == 0 and args.logdir is not None and (j % args.save_interval == 0 or j == num_updates - 1):
            os.makedirs(args.logdir, exist_ok=True)
            ckpt = {"update": j, "state_dict": actor_critic.state_dict(), "ob_rms": getattr(envs.envs, "ob_rms", None)}
            torch.save(ckpt, os.path.join(args.logdir, "solo_{}.pt".format(total)))
            torch.save(ckpt, os.path.join(args.logdir, "solo.pt"))
        if j % args.log_interval == 0:
            stats.flush()
            fps = int(total / (time.time() - start))
            rec = dict(update=j, timesteps=total, fps=fps, value_loss=value_loss, action_loss=action_loss, entropy=entropy,
                       ep_reward=(sum(stats.reward) / len(stats.reward)) if stats.reward else float("nan"),
                       ep_length=(sum(stats.length) / len(stats.length)) if stats.length else float("nan"),
                       success=(sum(stats.success) / len(stats.success)) if stats.success else float("nan"))
            history.append(rec)
            if rank == 0:
                print("Updates {update}, num timesteps {timesteps}, FPS {fps}\n mean reward {ep_reward:.2f} mean length "
                      "{ep_length:.0f} mean success {success:.2f}\n entropy {entropy:.2f} value loss {value_loss:.3f} "
                      "action loss {action_loss:.3f}".format(**rec), flush=True)
                if writer is not None:
                    for k in ("value_loss", "action_loss", "entropy", "ep_reward", "ep_length", "success"):
                        writer.add_scalar(k, rec[k], total)
    return actor_critic, history
