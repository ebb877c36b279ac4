"""One launch of N envs vs the same N envs as S independent engines stepped on S streams (not a pytest file)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from solorl_amd.config import *
from solorl_amd.vec_env import SoloVecEnv
c = default_config(ROBOT_SOLO12, TASK_WALK); c.num_history_stack = 1
for N, S in [(8192, 1), (8192, 2), (16384, 1), (16384, 4), (4096, 1), (4096, 2), (4096, 4)]:
    n = N // S
    envs = [SoloVecEnv(c, n, device="cuda:0", seed=1, env_id_offset=i * n) for i in range(S)]
    streams = [torch.cuda.Stream() for _ in range(S)]
    acts = [torch.rand(16, n, 12, device="cuda:0") * 2 - 1 for _ in range(S)]
    for e in envs: e.reset()
    torch.cuda.synchronize()
    JOIN = os.environ.get("JOIN", "0") == "1"       # 1: every step waits for all sub-batches of the previous one (synchronous vec-env semantics)
    def run(K):
        for t in range(K):
            for e, s, a in zip(envs, streams, acts):
                with torch.cuda.stream(s):
                    e.step_inplace(a[t % 16])
            if JOIN:
                evs = []
                for s in streams:
                    ev = torch.cuda.Event(); ev.record(s); evs.append(ev)
                for s in streams:
                    for ev in evs: s.wait_event(ev)
    run(30); torch.cuda.synchronize(); t0 = time.time(); K = 200
    run(K); torch.cuda.synchronize(); dt = time.time() - t0
    print("N %6d as %d x %5d: %.3f ms per step of all  %.1f M env-steps/s" % (N, S, n, dt / K * 1e3, N * K / dt / 1e6), flush=True)
