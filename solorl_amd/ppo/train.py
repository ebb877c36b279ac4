"""PPO training loop over the HIP rollout engine -- `train(args, config, env_constructor, writer)`
with the argument meaning of the reference's agents/ppo/train.py:21-162 and the flags of
training/train_ppo.py:9-45.

Differences that come with the GPU engine (documented, not behavioural):
  * rollout, policy forward, storage and update all stay on one HIP device: no device->host->pickle
    ->Pipe round trip per step (agents/ppo/envs.py:189-196);
  * episode statistics are accumulated by the step kernel on the device at EVERY step (no sampling) instead of
    iterating N Python dicts per step (train.py:90-100) -- same quantities: episode_reward (last-step reward,
    baseEnv.py:65), episode_length, success, dr/* sums, as means over the episodes finished per log interval;
  * multi-GPU: one process per GPU (`python -m torch.distributed.run --nproc-per-node W ...`), envs
    sharded by rank (global env id = rank*N + i), flat-bucket RCCL gradient all-reduce (ppo.py).
"""
import os
import time

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


def episode_stats(envs, device):
    """Means over ALL episodes finished since the last call (every step counts), from the engine's on-device
    accumulators (vec_env.SoloVecEnv.episode_stat_sums; include/solorl.h `ep_stats`), summed over ranks.  The
    reference keeps deques of the last 32 finished episodes (agents/ppo/train.py:66-70,90-100) and prints their
    mean; with thousands of envs the per-interval mean is the same quantity without the 32-episode window."""
    from ..config import EPSTAT_NAMES
    sums = envs.episode_stat_sums(reset=True)
    if D.world() > 1:
        import torch.distributed as dist
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
    tot = sums.tolist()
    n = tot[0]
    out = {"episodes": int(n), "nan_resets": int(tot[9])}
    for k in range(1, 9):
        out[EPSTAT_NAMES[k]] = (tot[k] / n) if n > 0 else float("nan")
    return out


def rollout(envs, actor_critic, storage, num_steps):
    """train.py:82-103: T steps of act -> env.step -> storage.append, entirely on the device (the episode statistics
    of train.py:90-100 are accumulated by the step kernel itself)."""
    for step in range(num_steps):
        with torch.no_grad():
            value, action, logp = actor_critic.act(storage.obs[step])
        obs, reward, done, info = envs.step_inplace(action.contiguous())
        storage.append(obs, action, logp, value, reward.unsqueeze(-1), (1.0 - done.float()).unsqueeze(-1))


def train(args, config, env_constructor=None, writer=None):
    from ..vec_env import make_vec_envs
    rank, world = D.init_from_env("cuda" if args.cuda else "cpu")
    torch.set_num_threads(1)                          # as the reference's learner (train.py:29); with one process per GPU the host work is a few scalars,
                                                      # and torch's default pool (one thread per host core) only competes with the other ranks
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
    if use_graphs:
        with torch.no_grad():
            actor_critic.act(storage.obs[0]); actor_critic.get_value(storage.obs[0])    # library workspaces before capture
        torch.manual_seed(args.seed + 1000 * rank)
        graphed_rollout = GraphedRollout(envs, actor_critic, storage, args.num_steps)
    start = time.time()
    num_updates = int(args.num_env_steps) // args.num_steps // (N * world)
    history = []
    for j in range(num_updates):
        if args.use_linear_lr_decay:
            update_linear_schedule(agent.optimizer, j, num_updates, args.lr)
        if use_graphs:
            graphed_rollout()
        else:
            rollout(envs, actor_critic, storage, args.num_steps)
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
            es = episode_stats(envs, device)
            fps = int(total / (time.time() - start))
            rec = dict(update=j, timesteps=total, fps=fps, value_loss=value_loss, action_loss=action_loss, entropy=entropy,
                       ep_reward=es["episode_reward"], ep_length=es["episode_length"], success=es["success"],
                       episodes=es["episodes"], **{k: v for k, v in es.items() if k.startswith("dr/")})
            history.append(rec)
            if rank == 0:
                print("Updates {update}, num timesteps {timesteps}, FPS {fps}\n mean reward {ep_reward:.2f} mean length "
                      "{ep_length:.0f} mean success {success:.2f}\n entropy {entropy:.2f} value loss {value_loss:.3f} "
                      "action loss {action_loss:.3f}".format(**rec), flush=True)
                if writer is not None:
                    for k in ("value_loss", "action_loss", "entropy", "ep_reward", "ep_length", "success"):
                        writer.add_scalar(k, rec[k], total)
                    for k in rec:
                        if k.startswith("dr/"):          # agents/ppo/train.py:98-100, utils.log of the dr/ deques
                            writer.add_scalar(k, rec[k], total)
    return actor_critic, history
