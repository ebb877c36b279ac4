#!/usr/bin/env python3
"""Launcher with the reference's flag names (training/train_ppo.py:9-45) and YAML task definitions
(configs/*.yaml).  Single GPU:  python train_ppo.py --config-file configs/basic12.yaml --task walk
--num-agents 4096 --use-gae --use-linear-lr-decay ...   Multi GPU (one process per GPU, RCCL):
python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 train_ppo.py ..."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def get_ppo_args(argv=None):
    p = argparse.ArgumentParser("PPO args")
    p.add_argument("--num-agents", type=int, default=32)          # envs PER GPU
    p.add_argument("--output-size", type=int, default=64)         # accepted and unused, as in the reference (train_ppo.py:12)
    p.add_argument("--hidden-size", type=int, default=64)
    p.add_argument("--no-cuda", action="store_true", default=False)
    p.add_argument("--no-hip-graphs", action="store_true", default=False, help="run rollout and update eagerly instead of replaying HIP graphs")
    p.add_argument("--env-name", type=str, default="base")
    p.add_argument("--gamma", type=float, default=0.99)
    p.add_argument("--tau", type=float, default=0.95)
    p.add_argument("--clip-param", type=float, default=0.1)
    p.add_argument("--ppo-epoch", type=int, default=10)
    p.add_argument("--mini-batch-size", type=int, default=32)
    p.add_argument("--lr", type=float, default=1e-3)
    p.add_argument("--l2-coef", type=float, default=0.0)
    p.add_argument("--value-loss-coef", type=float, default=0.5)
    p.add_argument("--entropy-coef", type=float, default=0.01)
    p.add_argument("--max-grad-norm", type=float, default=0.5)
    p.add_argument("--clip-value-loss", action="store_true", default=False)   # accepted; the reference's PPO always clips the value
                                                                               # loss (ppo.py:18 default, the flag is never forwarded)
    p.add_argument("--use-linear-lr-decay", action="store_true", default=False)
    p.add_argument("--use-gae", action="store_true", default=False)
    p.add_argument("--num-env-steps", type=float, default=1e6)
    p.add_argument("--seed", type=int, default=2301)
    p.add_argument("--curriculum-schedule", type=int, default=0)
    p.add_argument("--log-interval", type=int, default=10)
    p.add_argument("--logdir", type=str, default=None)
    p.add_argument("--base-checkpoint", type=str, default=None)
    p.add_argument("--timestamp", type=str, default=None)
    p.add_argument("--save-interval", type=int, default=20)
    p.add_argument("--config-file", type=str, default="configs/basic.yaml")
    p.add_argument("--task", type=str, default=None)
    p.add_argument("--num-steps", type=int, default=None, help="rollout length (default: episode_length, train_ppo.py:62-63)")
    return p.parse_args(argv)


def main(argv=None):
    from solorl_amd.config import load_yaml
    from solorl_amd.ppo.train import train
    args = get_ppo_args(argv)
    config = load_yaml(args.config_file)
    if args.task is not None:
        config["task"] = args.task      # the override the reference left commented out (train_ppo.py:57-60)
    if args.env_name != "base":
        raise SystemExit("--env-name %s needs the absent `scripts.Controller` (SURVEY.md section 0.6); only 'base' is provided" % args.env_name)
    args.cuda = not args.no_cuda
    if not args.cuda:
        raise SystemExit("the rollout engine is GPU-only (no CPU fallback)")
    if args.num_steps is None:
        args.num_steps = config["episode_length"]
    writer = None
    rank = int(os.environ.get("RANK", "0"))
    if args.logdir is not None:      # run directory named as training/train_ppo.py:64-72
        from datetime import datetime
        # under torch.distributed.run every rank runs main(): the stamp comes from the launcher's start time when it exports one
        # (TORCHELASTIC_RUN_ID does not carry it), else from this rank's clock -- only rank 0's directory is ever written to
        stamp = datetime.now().strftime("%Y%m%d-%H%M%S") if args.timestamp is None else datetime.now().strftime("%Y%m%d-") + args.timestamp
        task = args.task + "_" if args.task is not None else ""
        args.logdir = os.path.join(args.logdir, "Solo" + args.env_name.capitalize() + "_" + task + stamp)
        if rank == 0:                # the reference is single-process; with N ranks only rank 0 logs and saves
            try:                     # tensorboard is optional here (not installed in the build image)
                from torch.utils.tensorboard import SummaryWriter
                writer = SummaryWriter(args.logdir)
            except Exception:
                writer = None
    return train(args, config, None, writer)


if __name__ == "__main__":
    main()
