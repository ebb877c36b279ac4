"""Sweep envs-per-wave (not a pytest file)."""
import os, sys, time, subprocess
if len(sys.argv) > 1:
    import torch
    sys.path.insert(0, ".")
    from solorl_amd.config import *
    from solorl_amd.vec_env import SoloVecEnv
    N = int(sys.argv[1])
    c = default_config(ROBOT_SOLO12, TASK_WALK); c.num_history_stack = 1
    env = SoloVecEnv(c, N, device="cuda:0", seed=1); env.reset()
    a = torch.rand(16, N, 12, device="cuda:0") * 2 - 1
    for t in range(30): env.step_inplace(a[t % 16])
    torch.cuda.synchronize(); t0 = time.time(); K = 200
    for t in range(K): env.step_inplace(a[t % 16])
    torch.cuda.synchronize(); dt = time.time() - t0
    print("N %6d epw %3s: %.3f ms/step  %.2f M env-steps/s" % (N, os.environ.get("SOLORL_ENVS_PER_WAVE", "auto"), dt / K * 1e3, N * K / dt / 1e6), flush=True)
else:
    for N in (4096, 16384, 65536):
        for epw in ("1", "2", "4", "8", "16", "64", None):
            env = dict(os.environ)
            if epw: env["SOLORL_ENVS_PER_WAVE"] = epw
            else: env.pop("SOLORL_ENVS_PER_WAVE", None)
            subprocess.call([sys.executable, "-u", __file__, str(N)], env=env)
