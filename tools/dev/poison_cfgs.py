"""LDS-poison reproducibility check (see tests/test_parity_gpu.py::test_no_lds_read_before_write) over the other
robot / task / control / history configurations."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from solorl_amd.config import *
from solorl_amd.vec_env import SoloVecEnv
N, K = 4096, 450
for robot, task, control, hist in [(ROBOT_SOLO12, TASK_POINTGOAL, CONTROL_TORQUE, 2), (ROBOT_SOLO12, TASK_STAND, CONTROL_PD, 0), (ROBOT_SOLO8, TASK_WALK, CONTROL_TORQUE, 1),
                                   (ROBOT_SOLO8, TASK_POINTGOAL, CONTROL_PD, 2), (ROBOT_SOLO12, TASK_WALK, CONTROL_TORQUE, 1)]:
    c = default_config(robot, task); c.control = control; c.num_history_stack = hist
    if control == CONTROL_PD: c.kp, c.kd = 5.0, 0.2
    outs = []
    for word in ("0", "0x7fc00000", "0xffffffff"):
        os.environ["SOLORL_POISON_LDS"] = word
        env = SoloVecEnv(c, N, device="cuda:0", seed=3); env.reset()
        g = torch.Generator(device="cuda:0"); g.manual_seed(2)
        a = torch.rand(32, N, env.act_dim, device="cuda:0", generator=g) * 2.4 - 1.2
        acc = torch.zeros(N, device="cuda:0")
        for t in range(K):
            o, r, d, info = env.step_inplace(a[t % 32]); acc += r
        outs.append((o.clone(), acc.clone()))
    ok = all(torch.equal(outs[0][0], x[0]) and torch.equal(outs[0][1], x[1]) for x in outs[1:])
    print("robot %d task %d control %d hist %d: %s" % (robot, task, control, hist, "reproducible" if ok else "DIFFERS"), flush=True)
