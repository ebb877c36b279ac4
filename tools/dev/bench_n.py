"""Throughput vs batch size and lane mapping (not a pytest file)."""
import os, sys, time, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import torch
    sys.path.insert(0, ROOT)
    from solorl_amd.config import *
    from solorl_amd.vec_env import SoloVecEnv
    N = int(sys.argv[1])
    c = default_config(ROBOT_SOLO12, TASK_WALK); c.num_history_stack = 1
    env = SoloVecEnv(c, N, device="cuda:0", seed=1); env.reset()
    a = torch.rand(16, N, 12, device="cuda:0") * 2 - 1
    for t in range(30): env.step_inplace(a[t % 16])
    torch.cuda.synchronize(); t0 = time.time(); K = 150
    for t in range(K): env.step_inplace(a[t % 16])
    torch.cuda.synchronize(); dt = time.time() - t0
    print("N %6d team %s: %.3f ms/step  %.2f M env-steps/s" % (N, os.environ.get("SOLORL_TEAM", "auto"), dt / K * 1e3, N * K / dt / 1e6), flush=True)
else:
    for N in (1024, 4096, 8192, 16384, 32768, 65536, 262144):
        for tm in ("0", "1"):
            env = dict(os.environ); env["SOLORL_TEAM"] = tm
            subprocess.call([sys.executable, "-u", __file__, str(N)], env=env)
