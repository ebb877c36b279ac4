"""Short PPO training on two configs, printing (value loss, action loss, entropy) per update -- run twice, with and without
SOLORL_PPO_KERNELS=0, to compare the policy-kernel path with the PyTorch path end to end (dev tool)."""
import os, sys, types, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from solorl_amd.config import load_yaml
from solorl_amd.ppo.train import train
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for cfg, task in (("basic.yaml", "walk"), ("basic12.yaml", "walk")):
    config = load_yaml(os.path.join(root, "configs", cfg)); config["task"] = task
    args = types.SimpleNamespace(num_agents=512, hidden_size=64, cuda=True, gamma=0.99, tau=0.95, clip_param=0.1, ppo_epoch=2, mini_batch_size=2048,
        lr=2.5e-4, l2_coef=0.0, value_loss_coef=0.5, entropy_coef=0.01, max_grad_norm=0.5, use_linear_lr_decay=True,
        use_gae=True, num_env_steps=512 * 64 * 2, seed=1, curriculum_schedule=0, log_interval=1, logdir="/tmp/ab_%s" % cfg,
        base_checkpoint=None, save_interval=100, num_steps=64)
    pol, hist = train(args, config)
    print(cfg, os.environ.get("SOLORL_PPO_KERNELS", "1"), [(round(h["value_loss"], 3), round(h["action_loss"], 5), round(h["entropy"], 4)) for h in hist], flush=True)
