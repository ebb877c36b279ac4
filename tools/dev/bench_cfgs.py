"""ms/step for the other robot / task / control combinations at 4096 envs (not a pytest file)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from solorl_amd.config import *
from solorl_amd.vec_env import SoloVecEnv
N = 4096
for robot, task, control, hist in [(ROBOT_SOLO12, TASK_WALK, CONTROL_TORQUE, 1), (ROBOT_SOLO12, TASK_POINTGOAL, CONTROL_TORQUE, 1), (ROBOT_SOLO12, TASK_STAND, CONTROL_PD, 0),
                                   (ROBOT_SOLO8, TASK_WALK, CONTROL_TORQUE, 1), (ROBOT_SOLO8, TASK_STAND, CONTROL_PD, 2)]:
    c = default_config(robot, task); c.control = control; c.num_history_stack = hist
    if control == CONTROL_PD: c.kp, c.kd = 5.0, 0.2
    env = SoloVecEnv(c, N, device="cuda:0", seed=1); env.reset()
    a = torch.rand(16, N, env.act_dim, device="cuda:0") * 2 - 1
    for t in range(30): env.step_inplace(a[t % 16])
    torch.cuda.synchronize(); t0 = time.time(); K = 200
    for t in range(K): env.step_inplace(a[t % 16])
    torch.cuda.synchronize(); dt = time.time() - t0
    print("robot %d task %d control %d hist %d: %.3f ms/step  %.1f M env-steps/s" % (robot, task, control, hist, dt / K * 1e3, N * K / dt / 1e6), flush=True)
