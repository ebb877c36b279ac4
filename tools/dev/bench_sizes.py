"""Steady-state step time vs batch size (not a pytest file): burn-in of 450 steps, then the median of 3 graph replays of
K steps.  usage: bench_sizes.py [N ...]      env: BENCH_CFG=walk|pd50 (config 5: PD path, episode length 50), BENCH_THR=1e-7 (K7 residual threshold)"""
import os, sys, time, statistics
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from solorl_amd.config import *
from solorl_amd.vec_env import SoloVecEnv
sizes = [int(x) for x in sys.argv[1:]] or [4096, 8192, 16384, 65536]
which = os.environ.get("BENCH_CFG", "walk")
for N in sizes:
    c = default_config(ROBOT_SOLO12, TASK_WALK); c.num_history_stack = 1
    amp = 1.0
    if which == "pd50":
        c.control = CONTROL_PD; c.episode_length = 50; amp = 0.3
    if "BENCH_THR" in os.environ: c.solver_residual_threshold = float(os.environ["BENCH_THR"])      # (default: PyBullet's 1e-7)
    env = SoloVecEnv(c, N, device="cuda:0", seed=1); env.reset()
    g = torch.Generator(device="cuda:0"); g.manual_seed(1234)
    a = (torch.rand(64, N, 12, device="cuda:0", generator=g) * 2 - 1) * amp
    for t in range(450): env.step_inplace(a[t % 64])
    K = 200 if N <= 16384 else 60
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for t in range(K): env.step_inplace(a[t % 64])
    ts = []
    for r in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); gr.replay(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    dt = statistics.median(ts)
    print("%s N %6d: %.4f ms/step  %.2f M env-steps/s" % (which, N, dt / K * 1e3, N * K / dt / 1e6), flush=True)
    del env, gr
