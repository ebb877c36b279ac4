#!/usr/bin/env python3
"""bench.py -- env-steps/s of the hot path on N GPUs of one node.

Contract: `python bench.py --gpus N --steps K --warmup W` (N>1 under torch.distributed.run, one
rank per GPU).  A "step" = one pass of the hot path over one batch: solorl_step() on 4096
Solo12 'walk' envs per GPU (configs/basic12.yaml with task overridden to walk: frame_skip 4,
episode_length 400, num_history_stack 1, torque control) driven by a random policy
(a ~ U(-1,1)^12, synthetic, resident in HBM before the timed region).  Envs shard across ranks with no
data-path collective ("weak" scaling); the only collectives are the barrier and the max-reduce
of the elapsed time the contract asks for.

Prints ONE JSON line with value = whole-job env-steps/s plus `roofline` (algorithmic bytes of
SURVEY.md 8d / kernel time from HIP events on the launch stream) and `cpu_baseline` (the fp64
oracle timed on the host cores on a bounded sample; rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ENVS_PER_GPU = 4096
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes_per_env_step(A, S, C, O, hist_state, n_dr=5, n_counters=2):
    """B = 4*(A + 2S + 2C + O + H_rw + 2) + 1."""
    h_rw = 2 * (hist_state + n_dr + n_counters)
    return 4 * (A + 2 * S + 2 * C + O + h_rw + 2) + 1


def cpu_baseline(cfg, budget_s=12.0):
    """Oracle (fp64 CPU restatement, kind 'port') on a bounded sample of the same workload."""
    import numpy as np
    from oracle.oracle_py import Oracle
    try:
        cores = len(os.sched_getaffinity(0))      # the GPU box grants a CPU share, not the whole host
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = min(cores, int(os.environ.get("SOLORL_CPU_THREADS", "64")))
    N = 16 * cores
    orc = Oracle(cfg, N, seed=1, threads=cores)
    orc.reset()
    rng = np.random.default_rng(0)
    acts = rng.uniform(-1, 1, size=(8, N, orc.A))
    orc.step(acts[0])
    t0 = time.perf_counter(); steps = 0
    while True:
        orc.step(acts[steps % 8]); steps += 1
        el = time.perf_counter() - t0
        if el > budget_s or steps >= 2000:
            break
    return {"value": N * steps / el, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": "%d Solo12-walk envs x %d control steps, random policy, OpenMP over envs (%.1f s)" % (N, steps, el)}


def ppo_leg(env, dev, world, T):
    """One PPO iteration with the README hyper-parameters (lr 2.5e-4, entropy 0.01, clip 0.1, GAE,
    ppo-epoch 5 scaled to 1 epoch here, 50 mini-batches): rollout of T steps + returns + update, both loops
    replayed from HIP graphs (solorl_amd/ppo/graphs.py) that were captured in an untimed first iteration."""
    import torch
    from solorl_amd.ppo import Policy, RolloutStorage
    from solorl_amd.ppo.graphs import GraphedPPO, GraphedRollout
    N = env.nenvs
    torch.manual_seed(1)
    pol = Policy(env.observation_space.shape, env.action_space, None, {"hidden_size": 64}).to(dev)
    agent = GraphedPPO(pol, 0.1, 1, max(T * N // 50, 1), 0.5, 0.01, lr=2.5e-4, max_grad_norm=0.5)
    st = RolloutStorage(T, N, env.observation_space.shape, env.act_dim, dev)
    st.obs[0].copy_(env.get_observation())
    with torch.no_grad():
        pol.act(st.obs[0]); pol.get_value(st.obs[0])          # library workspaces before capture
    roll = GraphedRollout(env, pol, st, T)

    def iteration():
        torch.cuda.synchronize(); t0 = time.perf_counter()
        roll()
        torch.cuda.synchronize(); t_roll = time.perf_counter() - t0
        with torch.no_grad():
            nv = pol.get_value(st.obs[-1])
        st.compute_returns(nv, True, 0.99, 0.95)
        agent.update(st)
        st.reset()
        torch.cuda.synchronize()
        return t_roll, time.perf_counter() - t0

    iteration()                     # captures both graphs (and trains one step)
    t_roll, t_all = iteration()
    return {"env_steps_per_s": world * N * T / t_all, "rollout_env_steps_per_s": world * N * T / t_roll, "rollout_steps": T,
            "ppo_epoch": 1, "mini_batches": 50,
            "note": "policy forward + env.step + storage per step, then GAE + one PPO epoch; rollout and mini-batch step replayed from HIP graphs"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--envs-per-gpu", type=int, default=ENVS_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--ppo-steps", type=int, default=64, help="rollout length of the auxiliary PPO-loop leg (0 = skip)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="multi-rank rehearsal on a single-GPU box: all ranks use cuda:0 and the gloo backend (numbers are meaningless)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from solorl_amd.config import default_config, ROBOT_SOLO12, TASK_WALK
    from solorl_amd.vec_env import SoloVecEnv

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the engine has no CPU fallback)")
    if args.rehearse_on_one_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    N = args.envs_per_gpu
    cfg = default_config(ROBOT_SOLO12, TASK_WALK)
    cfg.num_history_stack = 1                      # configs/basic12.yaml
    env = SoloVecEnv(cfg, N, device=dev, seed=1, env_id_offset=rank * N)
    env.reset()
    g = torch.Generator(device=dev); g.manual_seed(1234 + rank)
    R = 64
    acts = torch.rand((R, N, env.act_dim), device=dev, generator=g) * 2 - 1

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for t in range(args.warmup):
        env.step_inplace(acts[t % R])
    # the K timed steps are one HIP-graph replay (K step-kernel nodes): the C-ABI launch is capturable and a graph
    # removes the ~5 us dispatch gap between consecutive launches; falls back to eager launches if capture fails
    launch = "hip_graph"
    try:
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for t in range(args.steps):
                env.step_inplace(acts[t % R])
    except Exception:
        graph, launch = None, "eager"

    def all_ranks(flag):            # collective AND, so that every rank takes the same path (same number of barriers)
        if world > 1:
            f = torch.tensor([1 if flag else 0], device=dev)
            dist.all_reduce(f, op=dist.ReduceOp.MIN)
            return bool(f.item())
        return bool(flag)

    if not all_ranks(graph is not None):
        graph, launch = None, "eager"

    def timed(run):
        barrier()
        t0 = time.perf_counter()
        run()
        barrier()
        return time.perf_counter() - t0

    def eager_steps():
        for t in range(args.steps):
            env.step_inplace(acts[t % R])

    if graph is not None:
        probe = env.get_state(0).rng_counter, env.get_state(0).timestep, float(env._obs.double().sum())
        elapsed = timed(graph.replay)
        after = env.get_state(0).rng_counter, env.get_state(0).timestep, float(env._obs.double().sum())
        if not all_ranks(after != probe):   # a replay did nothing (capture did not see the launches): do not report it
            launch = "eager"
            elapsed = timed(eager_steps)
    else:
        elapsed = timed(eager_steps)
    if world > 1:
        el = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        elapsed = float(el.item())

    # kernel time from HIP events on the launch stream (the engine launches on torch's current stream)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(min(args.steps, 200))]
    for t, (a, b) in enumerate(ev):
        a.record(); env.step_inplace(acts[t % R]); b.record()
    torch.cuda.synchronize()
    kms = sorted(a.elapsed_time(b) for a, b in ev)
    k_avg = sum(kms) / len(kms)      # one step = sort + gather + step_kernel launches; the step kernel is >95 % of it

    ppo = None
    if args.ppo_steps > 0 and world == 1:   # (auxiliary leg, single GPU only: the scaling runs measure the headline metric)
        # BASELINE config 3: the full PPO loop (rollout + GAE + clipped update) on the same engine
        try:
            ppo = ppo_leg(env, dev, world, args.ppo_steps)
        except Exception as ex:      # the auxiliary leg must never take the headline measurement down with it
            ppo = {"error": "%s: %s" % (type(ex).__name__, ex)}

    if rank == 0:
        A, S, O, D = env.act_dim, 37, env.obs_dim, cfg.state_dim
        bytes_step = algorithmic_bytes_per_env_step(A=A, S=S, C=48, O=O, hist_state=D)
        achieved = bytes_step * N / (k_avg * 1e-3) / 1e9
        out = {
            "metric": "env-steps/s (whole node) Solo12 walk, 4096 envs/GPU",
            "value": world * N * args.steps / elapsed, "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "solo12_walk_%denvs_per_gpu_random_policy_sim_only" % N, "robot": "solo12",
                       "task": "walk", "envs_per_gpu": N, "frame_skip": 4, "episode_length": 400,
                       "num_history_stack": 1, "control": "torque", "parallelism": "env-sharded x%d" % world, "launch": launch},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "step_kernel_team<float,solo12>", "kernel_ms_avg": k_avg, "kernel_ms_min": kms[0],
                         "algorithmic_bytes_per_env_step": bytes_step,
                         "note": "path is FP32-VALU issue / dependency bound (SURVEY 8d), see valu_issue; HBM fraction is small by construction"},
        }
        traffic_file = os.path.join(ROOT, "profiles", "r01_traffic.json")
        if os.path.exists(traffic_file):   # HBM bytes per launch from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
            prof = json.load(open(traffic_file))
            out["roofline"]["traffic"] = prof["bytes_per_launch"]
            if "valu_insts_per_launch" in prof:
                # what actually bounds the kernel: VALU issue.  A SIMD issues one wave64 VALU instruction per 4 cycles
                # (MI355X_MICROARCH.md: 256 CUs x 4 SIMD16); instruction count from the committed SQ_INSTS_VALU pass.
                peak = 256 * 4 * 2.4e9 / 4.0
                rate = prof["valu_insts_per_launch"] / (k_avg * 1e-3)
                out["roofline"]["valu_issue"] = {"achieved_ginst_s": rate / 1e9, "peak_ginst_s": peak / 1e9, "frac": rate / peak,
                                                 "valu_insts_per_launch": prof["valu_insts_per_launch"]}
        if ppo is not None:
            out["ppo_loop"] = ppo
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
