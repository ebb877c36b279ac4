"""A/B of engine builds, fp64 instantiation (4096 Solo12-walk envs, steady state, eager launches, median of 5 x 50 steps):
usage: ab_f64.py lib1.so lib2.so ...   (each in its own process, SOLORL_LIB)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("AB_CHILD"):
    import time, torch
    sys.path.insert(0, ROOT)
    from solorl_amd.config import *
    from solorl_amd.vec_env import SoloVecEnv
    N = 4096
    c = default_config(ROBOT_SOLO12, TASK_WALK); c.num_history_stack = 1; c.precision = PRECISION_F64
    env = SoloVecEnv(c, N, device="cuda:0", seed=1); env.reset()
    g0 = torch.Generator(device="cuda:0"); g0.manual_seed(1234)
    a = torch.rand(64, N, env.act_dim, device="cuda:0", generator=g0) * 2 - 1
    for t in range(450): env.step_inplace(a[t % 64])
    ts = []
    for r in range(5):
        torch.cuda.synchronize(); t0 = time.time()
        for t in range(50): env.step_inplace(a[t % 64])
        torch.cuda.synchronize(); ts.append(time.time() - t0)
    dt = sorted(ts)[2]
    print("%-28s f64 N %5d: %.4f ms/step  %.2f M env-steps/s" % (os.path.basename(os.environ.get("SOLORL_LIB", "default")), N, dt / 50 * 1e3, N * 50 / dt / 1e6), flush=True)
else:
    for lib in sys.argv[1:]:
        env = dict(os.environ, AB_CHILD="1")
        if lib != "default": env["SOLORL_LIB"] = os.path.abspath(lib)
        subprocess.call([sys.executable, os.path.abspath(__file__)], env=env)
