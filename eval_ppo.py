#!/usr/bin/env python3
"""Headless evaluation of a trained policy (the role of the reference's GUI script testing/test_ppo.py:17-151):
loads `<checkpoint-dir>/solo.pt` (same dict keys as the reference's checkpoint, agents/ppo/train.py:121-131),
rolls it out on the HIP engine until --num-runs episodes have ended and prints
`mean length / mean reward / mean success` (test_ppo.py:147).  In place of the GUI, --dump writes env 0's
trajectory (base pose, joint angles, action, reward, done) to an .npz."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main(argv=None):
    p = argparse.ArgumentParser("PPO eval")
    p.add_argument("--checkpoint-dir", required=True)
    p.add_argument("--config-file", default="configs/basic.yaml")
    p.add_argument("--task", default=None)
    p.add_argument("--num-runs", type=int, default=10)
    p.add_argument("--num-agents", type=int, default=64)
    p.add_argument("--hidden-size", type=int, default=64)
    p.add_argument("--deterministic", action="store_true")
    p.add_argument("--seed", type=int, default=7)
    p.add_argument("--dump", default=None, help="write env 0's trajectory to this .npz")
    a = p.parse_args(argv)

    import numpy as np
    import torch
    from solorl_amd.config import load_yaml
    from solorl_amd.ppo import Policy
    from solorl_amd.vec_env import make_vec_envs

    config = load_yaml(a.config_file)
    if a.task is not None:
        config["task"] = a.task
    ckpt = torch.load(os.path.join(a.checkpoint_dir, "solo.pt"), map_location="cpu", weights_only=False)
    dev = torch.device("cuda:0")
    env = make_vec_envs(config, a.num_agents, None, device=dev, training=False, seed=a.seed)
    policy = Policy(env.observation_space.shape, env.action_space, None, {"hidden_size": a.hidden_size}).to(dev)
    policy.load_state_dict(ckpt["state_dict"])
    policy.eval()
    torch.manual_seed(a.seed)

    ep_len, ep_R, ep_succ = [], [], []
    ret = torch.zeros(a.num_agents, device=dev)
    traj = dict(pos=[], quat=[], q=[], action=[], reward=[], done=[])
    obs = env.reset()
    while len(ep_len) < a.num_runs:
        with torch.no_grad():
            _, action, _ = policy.act(obs, deterministic=a.deterministic)
        if a.dump:
            s = env.get_state(0)
            traj["pos"].append(list(s.pos)); traj["quat"].append(list(s.quat)); traj["q"].append(list(s.q))
            traj["action"].append(action[0].cpu().numpy())
        obs, reward, done, infos = env.step(action)
        ret += reward.view(-1)
        if a.dump:
            traj["reward"].append(float(reward[0])); traj["done"].append(float(done[0]))
        idx = torch.nonzero(done).view(-1).tolist()
        if idx:
            T = infos.tensors
            for i in idx:
                ep_len.append(int(T["episode_length"][i])); ep_succ.append(float(T["success"][i])); ep_R.append(float(ret[i]))
            ret[done.bool()] = 0
    n = len(ep_len)
    print("episodes {} mean length {:.1f} mean reward {:.3f} mean success {:.2f}".format(
        n, sum(ep_len) / n, sum(ep_R) / n, sum(ep_succ) / n))
    if a.dump:
        np.savez(a.dump, **{k: np.asarray(v) for k, v in traj.items()})
        print("trajectory of env 0 ({} steps) -> {}".format(len(traj["reward"]), a.dump))
    env.close()
    return dict(episodes=n, mean_length=sum(ep_len) / n, mean_reward=sum(ep_R) / n, mean_success=sum(ep_succ) / n)


if __name__ == "__main__":
    main()
