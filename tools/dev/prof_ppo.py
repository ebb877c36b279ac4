"""PPO-loop leg only (for rocprofv3 --kernel-trace --stats): rollout graph + update graphs, two iterations."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from solorl_amd.config import default_config, ROBOT_SOLO12, TASK_WALK, TASK_POINTGOAL
from solorl_amd.vec_env import SoloVecEnv
dev = torch.device("cuda:0")
cfg = default_config(ROBOT_SOLO12, TASK_POINTGOAL if os.environ.get("PPO_TASK") == "pointgoal" else TASK_WALK); cfg.num_history_stack = 1     # PPO_TASK=pointgoal: BASELINE config 3's shape (obs 84)
env = SoloVecEnv(cfg, int(os.environ.get("PPO_N", "4096")), device=dev, seed=1); env.reset()
print(bench.ppo_leg(env, dev, 1, int(os.environ.get("PPO_T", "400")), int(os.environ.get("PPO_EPOCHS", "1"))))
