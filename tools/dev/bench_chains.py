"""N envs as S independent engines (sub-batches) free-running on S HIP streams, joined only at the end of the K steps -- the
launch tail of one sub-batch overlaps the next steps of the others (not a pytest file).  usage: bench_chains.py [N]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from solorl_amd.config import *
from solorl_amd.vec_env import SoloVecEnv
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for thr, warm in ((1e-7, 0.0), (0.0, 0.85)):
    for S in (1, 2, 4, 8, 16):
        c = default_config(ROBOT_SOLO12, TASK_WALK); c.num_history_stack = 1
        c.solver_residual_threshold, c.warmstart = thr, warm
        n = N // S
        envs = [SoloVecEnv(c, n, device="cuda:0", seed=1, env_id_offset=i * n) for i in range(S)]
        streams = [torch.cuda.Stream() for _ in range(S)]
        acts = [torch.rand(64, n, 12, device="cuda:0") * 2 - 1 for _ in range(S)]
        for e in envs: e.reset()
        torch.cuda.synchronize()
        def run(K):
            for t in range(K):
                for e, s, a in zip(envs, streams, acts):
                    with torch.cuda.stream(s):
                        e.step_inplace(a[t % 64])
        run(450); torch.cuda.synchronize()
        ts = []
        K = 300
        for r in range(3):
            t0 = time.time(); run(K); torch.cuda.synchronize(); ts.append(time.time() - t0)
        dt = sorted(ts)[1]
        print("thr %g warm %.2f: N %6d as %2d x %5d: %.4f ms per step of all  %.2f M env-steps/s" % (thr, warm, N, S, n, dt / K * 1e3, N * K / dt / 1e6), flush=True)
        for e in envs: e.close()
