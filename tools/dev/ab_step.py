"""A/B of engine builds on the headline workload (4096 / 8192 Solo12-walk envs, steady state, 200-step graph replays, median of 5):
usage: ab_step.py lib1.so lib2.so ...   (each measured in its own process: the library is chosen at import, SOLORL_LIB)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("AB_CHILD"):
    import time, torch
    sys.path.insert(0, ROOT)
    from solorl_amd.config import *
    from solorl_amd.vec_env import SoloVecEnv
    for N in (4096, 8192):
        c = default_config(ROBOT_SOLO12, TASK_WALK); c.num_history_stack = 1
        env = SoloVecEnv(c, N, device="cuda:0", seed=1); env.reset()
        g0 = torch.Generator(device="cuda:0"); g0.manual_seed(1234)
        a = torch.rand(64, N, env.act_dim, device="cuda:0", generator=g0) * 2 - 1
        for t in range(450): env.step_inplace(a[t % 64])
        K = 200
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for t in range(K): env.step_inplace(a[t % 64])
        ts = []
        for r in range(5):
            torch.cuda.synchronize(); t0 = time.time(); g.replay(); torch.cuda.synchronize(); ts.append(time.time() - t0)
        dt = sorted(ts)[2]
        print("%-28s N %5d: %.4f ms/step  %.2f M env-steps/s (min %.4f max %.4f)" % (os.path.basename(os.environ.get("SOLORL_LIB", "default")), N, dt / K * 1e3, N * K / dt / 1e6, min(ts) / K * 1e3, max(ts) / K * 1e3), flush=True)
        env.close()
else:
    for lib in sys.argv[1:]:
        env = dict(os.environ, AB_CHILD="1")
        if lib != "default":
            env["SOLORL_LIB"] = os.path.abspath(lib)
        subprocess.call([sys.executable, os.path.abspath(__file__)], env=env)
