#!/usr/bin/env python3
"""bench.py -- env-steps/s of the hot path on N GPUs of one node.

Contract: `python bench.py --gpus N --steps K --warmup W`.  Under torch.distributed.run (WORLD_SIZE set) this process
is one rank; started directly with --gpus N > 1 it first spawns the N ranks itself (before touching the GPU), relays
rank 0's JSON line and exits with the children's status.

A "step" = one pass of the hot path over one batch: solorl_step() on 4096 Solo12 'walk' envs per GPU
(configs/basic12.yaml with the task overridden to walk: frame_skip 4, episode_length 400, num_history_stack 1, torque
control) driven by a random policy (a ~ U(-1,1)^12, synthetic, resident in HBM before the timed region).  Envs shard
across ranks with no data-path collective ("weak" scaling); the only collectives of the headline measurement are the
barrier and the max-reduce of the elapsed time.

Protocol, independent of --warmup: an untimed burn-in of episode_length + 50 control steps brings the
fall/reset distribution to its stationary regime, then W warm-up steps, then the K timed steps -- ONE HIP-graph
replay bracketed by barrier + synchronize -- are repeated 5 times and the MEDIAN repeat is reported (each repeat is
exactly K steps; max over ranks per repeat).  `roofline.kernel_ms_avg` comes from HIP events recorded on the launch
stream around the SAME replays, so it can never exceed ms_per_step.

Prints ONE JSON line: value = whole-job env-steps/s, `roofline` (algorithmic bytes of SURVEY.md 8d / that kernel time;
HBM traffic, VALU issue rate and FP32 rate from the committed rocprofv3 PMC passes -- only while the loaded library IS the
one they were measured on, by hash), `f64` (the same workload on the engine's double-precision instantiation: the reference's
arithmetic is fp64 until agents/ppo/envs.py:192), `cpu_baseline` (the fp64 oracle timed on the host cores per BASELINE.md
section 3: 64 and 4096 envs, median of 5, all granted cores and one thread; rank 0, N=1 only) and `ppo_loop` (the full PPO
iteration of BASELINE config 3 with the README recipe, median of 3 iterations; with N > 1 every optimizer step carries the
80 KB RCCL all-reduce of the flat gradient bucket).
"""
import argparse
import hashlib
import json
import math
import os
import socket
import statistics
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ENVS_PER_GPU = 4096
EPISODE_LENGTH = 400
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_VECTOR_PEAK_TF = 157.3     # MI355X_MICROARCH.md: peak FP32 (vector)
VALU_ISSUE_PEAK = 256 * 4 * 2.4e9 / 2.0   # wave64 VALU instructions/s: 256 CUs x 4 SIMD-32, 2 cycles each (a LONE wave sustains 4)
REPEATS = 5
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r04_traffic.json")
EXIT_WATCHDOG = 3               # --strict: exit code when the watchdog cut the auxiliary legs short (the headline line is out by then)


def algorithmic_bytes_per_env_step(A, S, C, O, hist_state, n_dr=5, n_counters=2):
    """B = 4*(A + 2S + 2C + O + H_rw + 2) + 1."""
    h_rw = 2 * (hist_state + n_dr + n_counters)
    return 4 * (A + 2 * S + 2 * C + O + h_rw + 2) + 1


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--envs-per-gpu", type=int, default=ENVS_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-f64", action="store_true", help="skip the fp64-instantiation leg")
    ap.add_argument("--leg-timeout", type=int, default=420, help="watchdog (s) over the auxiliary legs (PPO loop, fp64, CPU baseline): the headline line is printed without them if they hang")
    ap.add_argument("--ppo-steps", type=int, default=400, help="rollout length of the PPO-loop leg (0 = skip); README: num_steps = episode_length")
    ap.add_argument("--ppo-epoch", type=int, default=5, help="PPO epochs of that leg (README: 5)")
    ap.add_argument("--strict", action="store_true", help="exit with code %d (instead of 0) when the watchdog had to cut an auxiliary leg short; the JSON line "
                    "says so either way (aux_legs_complete)" % EXIT_WATCHDOG)
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="multi-rank rehearsal on a single-GPU box: all ranks use cuda:0 and the gloo backend (numbers are meaningless)")
    return ap.parse_args(argv)


def spawn_ranks(args, argv):
    """`python bench.py --gpus N` started directly: launch the N ranks as children (torch.distributed.run) BEFORE this
    process touches the GPU, relay rank 0's JSON line, fail if any child fails."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    p = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in p.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
    if p.returncode != 0 or line is None:
        sys.stdout.write(p.stdout)
        sys.stderr.write("bench.py: %d-rank run failed (exit code %d, %s)\n" % (args.gpus, p.returncode, "no result line" if line is None else "result line present"))
        return p.returncode or 1
    print(line, flush=True)
    return 0


def granted_cores():
    """CPU share of this process: the affinity mask, cut down to the cgroup's CPU quota where one is set (the GPU box grants a
    share of its host: more runnable threads than quota are throttled, which is what made 64 OpenMP threads only 7 x one)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(math.ceil(float(txt[0]) / float(txt[1])))))
            else:
                q = float(txt[0]); per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, int(math.ceil(q / per))))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, min(n, int(os.environ.get("SOLORL_CPU_THREADS", "256"))))


def cpu_baseline(cfg, budget_s=12.0):
    """Oracle (fp64 CPU restatement, kind 'port') on the same workload, protocol of BASELINE.md section 3: 64 and 4096 envs,
    50 warm-up env-steps, >= 2000 measured env-steps per repetition, median of 5 repetitions; all granted cores (OpenMP over
    envs) and a single thread.  The budget is kept by the number of control steps per repetition, never by the env count."""
    import numpy as np
    from oracle.oracle_py import Oracle
    cores = granted_cores()

    def run(threads, N, budget):
        orc = Oracle(cfg, N, seed=1, threads=cores)          # (the untimed reset = 5-11 settle steps per env runs on all cores)
        orc.reset()
        orc.L.oracle_set_threads(orc.h, threads)
        rng = np.random.default_rng(0)
        acts = rng.uniform(-1, 1, size=(8, N, orc.A))
        t0 = time.perf_counter()
        for k in range(max(1, -(-50 // N))):                 # >= 50 warm-up env-steps
            orc.step(acts[k % 8])
        per_step = (time.perf_counter() - t0) / max(1, -(-50 // N))
        steps = max(-(-2000 // N), min(200, int(budget / 5.0 / max(per_step, 1e-6))))      # >= 2000 env-steps per repetition
        reps, t = [], 0
        for r in range(5):
            t0 = time.perf_counter()
            for k in range(steps):
                orc.step(acts[t % 8]); t += 1
            reps.append(N * steps / (time.perf_counter() - t0))
        return {"value": statistics.median(reps), "unit": "env-steps/s", "cores": threads, "envs": N, "steps_per_repetition": steps,
                "env_steps_per_repetition": N * steps, "repetitions": 5, "min": min(reps), "max": max(reps)}

    all_4096 = run(cores, 4096, budget_s * 0.35)
    all_64 = run(min(cores, 64), 64, budget_s * 0.1)
    one_64 = run(1, 64, budget_s * 0.2)
    one_4096 = run(1, 4096, budget_s * 0.35)
    return {"value": all_4096["value"], "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": "fp64 oracle, %d Solo12-walk envs x %d control steps x 5 repetitions (median), random policy, OpenMP over envs, %d threads; "
                      "every reset inside the sample runs the reference's LIVE settle (5-11 zero-torque control steps, baseEnv.py:79-80), which the GPU "
                      "engine replaces by a snapshot copy: at this workload's stationary episode length (~23 steps) that is about a third more "
                      "physics per env-step than the GPU line does" % (4096, all_4096["steps_per_repetition"], cores),
            "protocol": "BASELINE.md section 3: 50 warm-up env-steps, >= 2000 measured env-steps, median of 5",
            "all_cores": {"envs_4096": all_4096, "envs_64": all_64},
            "single_thread": {"envs_4096": one_4096, "envs_64": one_64},
            "thread_scaling_4096": all_4096["value"] / one_4096["value"]}


class LineGuard:
    """Guarantees the ONE JSON line.  Every leg after the headline measurement is auxiliary; a leg that HANGS (e.g. a collective of
    the PPO leg on a multi-GPU node) cannot be caught by try/except, so a timer prints the headline with the legs finished so far
    (`aux_legs_complete: false`, the unfinished leg carries an error entry) and every rank leaves through os._exit -- no collective
    is waited for.  Exit code then: 0, or EXIT_WATCHDOG with --strict (the caller decides whether a partial line is acceptable)."""

    def __init__(self, rank, timeout_s, strict, build_line, ppo_requested, exit_fn=os._exit):
        self.rank, self.timeout_s, self.strict, self.build_line, self.exit_fn = rank, timeout_s, strict, build_line, exit_fn
        self.state = {"f64": None, "ppo": {"error": "not run"} if ppo_requested else None, "cpu": None, "printed": False}
        self.timer = threading.Timer(timeout_s, self.on_timeout)
        self.timer.daemon = True

    def start(self):
        self.timer.start()

    def rearm(self, timeout_s):
        """A shorter fuse for the section that follows (the first captured RCCL all-reduce this project ever runs on two GPUs is the last
        thing a multi-GPU run does: if it hangs, the line goes out after `timeout_s`, not after the legs' whole budget)."""
        self.timer.cancel()
        self.timeout_s = timeout_s
        self.timer = threading.Timer(timeout_s, self.on_timeout)
        self.timer.daemon = True
        self.timer.start()

    def emit(self, extra_note=None):
        if self.state["printed"] or self.rank != 0:
            return
        self.state["printed"] = True
        line = self.build_line(self.state)
        line["aux_legs_complete"] = extra_note is None
        if extra_note:
            line["note"] = extra_note
        print(json.dumps(line), flush=True)

    def on_timeout(self):
        ppo = self.state["ppo"]
        if ppo is not None and ppo.get("error") == "not run":
            self.state["ppo"] = {"error": "the PPO leg did not finish within %d s (watchdog); headline unaffected" % self.timeout_s}
        self.emit("auxiliary legs cut short by the watchdog")
        self.exit_fn(EXIT_WATCHDOG if self.strict else 0)

    def finish(self):
        self.timer.cancel()
        self.emit()


def ppo_leg(env, dev, world, T, epochs):
    """One PPO iteration with the README recipe (lr 2.5e-4, entropy 0.01, clip 0.1, GAE, `epochs` PPO epochs, 50
    mini-batches per epoch, rollout length T): rollout + returns + update, both loops replayed from HIP graphs
    (solorl_amd/ppo/graphs.py) captured in an untimed first iteration.  world > 1: one flat-bucket all-reduce per
    optimizer step (agents/ppo/ppo.py:72-77 is where the reference would need it)."""
    import torch
    import torch.distributed as dist
    from solorl_amd.ppo import Policy, RolloutStorage
    from solorl_amd.ppo.graphs import GraphedPPO, GraphedRollout
    N = env.nenvs
    torch.manual_seed(1)
    pol = Policy(env.observation_space.shape, env.action_space, None, {"hidden_size": 64}).to(dev)
    mb = max((T * N // 50) // 64 * 64, 64)      # 50 mini-batches per epoch, rounded down to whole wavefronts (the policy kernels' granularity)
    agent = GraphedPPO(pol, 0.1, epochs, mb, 0.5, 0.01, lr=2.5e-4, max_grad_norm=0.5)
    st = RolloutStorage(T, N, env.observation_space.shape, env.act_dim, dev)
    st.obs[0].copy_(env.get_observation())
    with torch.no_grad():
        pol.act(st.obs[0]); pol.get_value(st.obs[0])          # library workspaces before capture
    roll = GraphedRollout(env, pol, st, T)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def iteration():
        sync(); t0 = time.perf_counter()
        roll()
        torch.cuda.synchronize(); t1 = time.perf_counter()
        with torch.no_grad():
            nv = pol.get_value(st.obs[-1])
        st.compute_returns(nv, True, 0.99, 0.95)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        agent.update(st)
        st.reset()
        sync()
        t3 = time.perf_counter()
        return t1 - t0, t3 - t0, t2 - t1, t3 - t2

    iteration()                     # captures both graphs (and trains one step)
    runs = []
    for _ in range(3):              # median of three timed iterations (max over ranks each)
        t_roll, t_all, t_ret, t_upd = iteration()
        if world > 1:
            tt = torch.tensor([t_roll, t_all, t_ret, t_upd], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            t_roll, t_all, t_ret, t_upd = tt.tolist()
        runs.append((t_all, t_roll, t_ret, t_upd))
    runs.sort()
    t_all, t_roll, t_ret, t_upd = runs[1]
    out = {"env_steps_per_s": world * N * T / t_all, "rollout_env_steps_per_s": world * N * T / t_roll, "rollout_steps": T,
           "rollout_ms": 1e3 * t_roll, "returns_ms": 1e3 * t_ret, "update_ms": 1e3 * t_upd,
           "iterations_timed": 3, "statistic": "median iteration", "iteration_ms": [1e3 * r[0] for r in runs],
           "allreduce_in_graph": bool(getattr(agent, "allreduce_in_graph", False)) if world > 1 else None,
           "ppo_epoch": epochs, "mini_batches_per_epoch": (T * N) // mb, "mini_batch": mb, "optimizer_steps": epochs * ((T * N) // mb),
           "note": "policy act (solorl_policy_act) + env.step writing into the rollout storage per step, then GAE + PPO epochs (solorl_ppo_grad_stage1/2 + clip + Adam per mini-batch); rollout and mini-batch step replayed from HIP graphs"}
    out["bucket_numel"] = agent.bucket.flat.numel()
    return out


def allreduce_leg(dev, world, numel, optimizer_steps, sync):
    """The collective of the data-parallel PPO step, timed on its own -- on SCRATCH tensors of the bucket's size (zeros: 50 summed
    all-reduces of live gradients would overflow).  Returns the eager part; `captured_allreduce_probe` is a separate call so that a
    captured collective that hangs on first contact with real xGMI costs only its own entry, not the PPO leg's numbers."""
    import torch
    import torch.distributed as dist
    from solorl_amd.ppo import dist as D
    reps = 50

    def eager_us(n):
        x = torch.zeros(n, device=dev)
        dist.all_reduce(x)
        sync(); t0 = time.perf_counter()
        for _ in range(reps):
            dist.all_reduce(x, op=dist.ReduceOp.SUM)
        sync()
        return 1e6 * (time.perf_counter() - t0) / reps

    return {"bytes": numel * 4, "us_per_call": eager_us(numel), "us_per_call_8_bytes": eager_us(2), "calls_per_update": optimizer_steps,
            "backend": dist.get_backend(), "rccl_env": D.rccl_env(),
            "note": "eager all_reduce(sum) of a scratch tensor of the bucket's size; beside it the same call on 8 bytes: "
                    "equal latencies = the bucket is latency-bound, whatever algorithm RCCL's tuner picked"}


def captured_allreduce_probe(dev, numel):
    """One execution of a CAPTURED all-reduce of the real bucket size (opt-in for training: SOLORL_CAPTURE_ALLREDUCE=1), nccl only."""
    from solorl_amd.ppo.graphs import probe_captured_allreduce
    try:
        ok, sec = probe_captured_allreduce(dev, numel, replays=8, time_it=True)
        return {"ok": ok, "us_per_replay": None if sec is None else 1e6 * sec, "replays_checked": 8}
    except Exception as ex:
        return {"error": "%s: %s" % (type(ex).__name__, ex)}


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args, argv))        # nothing GPU-related has been imported or called in this process

    import torch
    import torch.distributed as dist
    from solorl_amd.config import default_config, ROBOT_SOLO12, TASK_WALK
    from solorl_amd.vec_env import SoloVecEnv

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (start it as `python bench.py --gpus N` or under "
                         "torch.distributed.run with --nproc-per-node N --gpus N)" % (args.gpus, world))
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
            from solorl_amd.ppo.dist import pin_rccl_from_env
            pin_rccl_from_env()             # SOLORL_RCCL_ALGO / SOLORL_RCCL_PROTO -> NCCL_ALGO / NCCL_PROTO, before the communicator exists
            dist.init_process_group("nccl", device_id=dev)

    N, K = args.envs_per_gpu, args.steps
    cfg = default_config(ROBOT_SOLO12, TASK_WALK)
    cfg.num_history_stack = 1                      # configs/basic12.yaml
    assert cfg.episode_length == EPISODE_LENGTH
    env = SoloVecEnv(cfg, N, device=dev, seed=1, env_id_offset=rank * N)
    env.reset()
    g = torch.Generator(device=dev); g.manual_seed(1234 + rank)
    R = 64
    acts = torch.rand((R, N, env.act_dim), device=dev, generator=g) * 2 - 1

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def all_ranks(flag):            # collective AND, so that every rank takes the same path (same number of barriers)
        if world > 1:
            f = torch.tensor([1 if flag else 0], device=dev)
            dist.all_reduce(f, op=dist.ReduceOp.MIN)
            return bool(f.item())
        return bool(flag)

    burn_in = EPISODE_LENGTH + 50   # to the stationary fall/reset regime, whatever --warmup says
    for t in range(burn_in + args.warmup):
        env.step_inplace(acts[t % R])
    # the K timed steps are one HIP-graph replay (K step-kernel nodes): the C-ABI launch is capturable and a graph
    # removes the ~5 us dispatch gap between consecutive launches; falls back to eager launches if capture fails
    launch = "hip_graph"
    try:
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, **({"capture_error_mode": "thread_local"} if world > 1 else {})):   # (the process group's watchdog thread
            for t in range(K):                                                                             #  must not invalidate the capture)
                env.step_inplace(acts[t % R])
    except Exception:
        graph, launch = None, "eager"
    if not all_ranks(graph is not None):
        graph, launch = None, "eager"

    def eager_steps():
        for t in range(K):
            env.step_inplace(acts[t % R])

    def timed(run):
        """exactly K steps: wall clock between barrier+synchronize on both sides, and HIP events on the launch stream
        around the same work"""
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        barrier()
        t0 = time.perf_counter()
        e0.record(); run(); e1.record()
        barrier()
        el = time.perf_counter() - t0
        return el, e0.elapsed_time(e1) * 1e-3

    if graph is not None:
        probe = env.get_state(0).rng_counter, env.get_state(0).timestep, float(env._obs.double().sum())
        first = timed(graph.replay)
        after = env.get_state(0).rng_counter, env.get_state(0).timestep, float(env._obs.double().sum())
        if not all_ranks(after != probe):   # a replay did nothing (capture did not see the launches): do not report it
            graph, launch = None, "eager"
    run = graph.replay if graph is not None else eager_steps
    walls, devs = [], []
    for r in range(REPEATS):
        el, dv = timed(run)
        if world > 1:
            t_ = torch.tensor([el, dv], device=dev, dtype=torch.float64)
            dist.all_reduce(t_, op=dist.ReduceOp.MAX)
            el, dv = t_.tolist()
        walls.append(el); devs.append(dv)
    order = sorted(range(REPEATS), key=lambda i: walls[i])
    mid = order[REPEATS // 2]
    elapsed, dev_elapsed = walls[mid], devs[mid]
    k_avg = 1e3 * dev_elapsed / K                 # ms per launch inside the timed window (graph: includes the ~1 us node gaps)

    def build_line(state):
        f64, ppo = state["f64"], state["ppo"]
        A, S, O, D = env.act_dim, 37, env.obs_dim, cfg.state_dim
        bytes_step = algorithmic_bytes_per_env_step(A=A, S=S, C=48, O=O, hist_state=D)
        achieved = bytes_step * N / (k_avg * 1e-3) / 1e9
        out = {
            "metric": "env-steps/s (whole node) Solo12 walk, 4096 envs/GPU",
            "value": world * N * K / elapsed, "unit": "env-steps/s",
            "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "solo12_walk_%denvs_per_gpu_random_policy_sim_only" % N, "robot": "solo12",
                       "task": "walk", "envs_per_gpu": N, "frame_skip": 4, "episode_length": EPISODE_LENGTH,
                       "num_history_stack": 1, "control": "torque", "parallelism": "env-sharded x%d" % world, "launch": launch,
                       "solver": {"iterations_max": cfg.solver_iterations, "residual_threshold": cfg.solver_residual_threshold,
                                  "warmstart": cfg.warmstart, "friction_model": "cone" if cfg.friction_model else "pyramid", "erp": cfg.erp, "contact_erp": cfg.contact_erp,
                                  "note": "PyBullet defaults [K]: 50 iterations, solverResidualThreshold 1e-7, no multibody warm start, implicit friction cone, erp 0.2 / contactERP 0.08"},
                       "engine": {k: env.get_property(k) for k in ("lanes_per_env", "sweep_variant", "max_contacts", "max_limit_rows")},
                       "burn_in_steps": burn_in, "repeats": REPEATS, "statistic": "median of %d repeats of exactly %d steps" % (REPEATS, K)},
            "repeats_ms_per_step": [1e3 * w / K for w in walls],
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "step_kernel_team<float,solo12>", "kernel_ms_avg": k_avg,
                         "kernel_ms_source": "HIP events around the median timed replay / steps",
                         "algorithmic_bytes_per_env_step": bytes_step,
                         "note": "path is FP32-VALU issue / dependency bound (SURVEY 8d), see valu_issue and fp32; HBM fraction is small by construction"},
        }
        prof = json.load(open(TRAFFIC_FILE)) if os.path.exists(TRAFFIC_FILE) else None
        from solorl_amd import _native
        lib_hash = hashlib.sha256(open(_native.LIB_PATH, "rb").read()).hexdigest()[:16]
        out["roofline"]["lib_sha256_16"] = lib_hash
        if prof is not None and (prof.get("lib_sha256_16") != lib_hash or N != ENVS_PER_GPU):
            # the counters belong to another build of the engine (or another batch size): do not carry them
            out["roofline"]["pmc_note"] = "PMC-derived fields omitted: %s was measured on library %s, loaded library is %s" % (
                os.path.basename(TRAFFIC_FILE), prof.get("lib_sha256_16"), lib_hash)
            prof = None
        if prof is not None:
            # per-launch figures from the committed rocprofv3 --pmc passes (profiles/r03_pmc_*.txt), steady state, 4096 envs, SAME library
            rf = out["roofline"]
            rf["pmc_source"] = os.path.relpath(TRAFFIC_FILE, ROOT)
            rf["traffic"] = prof.get("bytes_per_launch")
            if "valu_insts_per_launch" in prof:
                rate = prof["valu_insts_per_launch"] / (k_avg * 1e-3)
                rf["valu_issue"] = {"achieved_ginst_s": rate / 1e9, "peak_ginst_s": VALU_ISSUE_PEAK / 1e9, "frac": rate / VALU_ISSUE_PEAK,
                                    "valu_insts_per_launch": prof["valu_insts_per_launch"],
                                    "peak_note": "256 CUs x 4 SIMD-32 x 2.4 GHz / 2 cycles per wave64 VALU (a lone wave sustains one per 4)"}
            if "fp32_flops_per_launch" in prof:
                fl = prof["fp32_flops_per_launch"] / (k_avg * 1e-3)
                rf["fp32"] = {"achieved_tflops": fl / 1e12, "peak_tflops": FP32_VECTOR_PEAK_TF, "frac": fl / 1e12 / FP32_VECTOR_PEAK_TF,
                              "flops_per_launch": prof["fp32_flops_per_launch"], "flops_per_env_step": prof["fp32_flops_per_launch"] / N,
                              "count": prof.get("fp32_flops_note", "")}
        if f64 is not None:
            out["f64"] = f64
        if ppo is not None:
            out["ppo_loop"] = ppo
        if state.get("cpu") is not None:
            out["cpu_baseline"] = state["cpu"]
        return out

    ppo = None
    # From here on every leg is auxiliary to the headline measurement above (LineGuard: the line is printed even if one hangs).
    guard = LineGuard(rank, args.leg_timeout, args.strict, build_line, args.ppo_steps > 0)
    state = guard.state
    guard.start()
    # the same workload on the engine's fp64 instantiation (the reference's arithmetic type), single repeat of <= 100 steps
    f64 = None
    if not args.no_f64:
        try:
            from solorl_amd.config import PRECISION_F64
            c64 = cfg.copy(); c64.precision = PRECISION_F64
            e64 = SoloVecEnv(c64, N, device=dev, seed=1, env_id_offset=rank * N)
            e64.reset()
            for t in range(burn_in):
                e64.step_inplace(acts[t % R])
            K64 = 100               # fixed, whatever --steps says: the documented fp64 figure is the one the driver's command prints
            launch64 = "hip_graph"
            try:
                torch.cuda.synchronize()
                g64 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g64, **({"capture_error_mode": "thread_local"} if world > 1 else {})):
                    for t in range(K64):
                        e64.step_inplace(acts[t % R])
            except Exception:
                g64, launch64 = None, "eager"
            if not all_ranks(g64 is not None):
                g64, launch64 = None, "eager"

            def run64():
                if g64 is not None:
                    g64.replay()
                else:
                    for t in range(K64):
                        e64.step_inplace(acts[t % R])

            run64()                 # first replay untimed
            els = []
            for _ in range(3):
                barrier(); t0 = time.perf_counter()
                run64()
                barrier(); el = time.perf_counter() - t0
                if world > 1:
                    t_ = torch.tensor([el], device=dev, dtype=torch.float64); dist.all_reduce(t_, op=dist.ReduceOp.MAX); el = float(t_.item())
                els.append(el)
            el64 = sorted(els)[1]
            f64 = {"value": world * N * K64 / el64, "unit": "env-steps/s", "ms_per_step": 1e3 * el64 / K64, "steps": K64, "repeats": 3, "dtype": "f64",
                   "launch": launch64, "statistic": "median of 3 repeats of exactly 100 steps after the same burn-in",
                   "note": "step_kernel_team<double,solo12>, same workload and burn-in; parity-tested against the oracle to rounding"}
            e64.close()
        except Exception as ex:
            f64 = {"error": "%s: %s" % (type(ex).__name__, ex)}
    state["f64"] = f64

    if args.ppo_steps > 0:
        # BASELINE config 3 / 4: the full PPO loop (rollout + GAE + clipped update) on the same engine; never fatal
        try:
            ppo = ppo_leg(env, dev, world, args.ppo_steps, args.ppo_epoch)
        except Exception as ex:     # never fatal for the headline line.  A failure on ONE rank leaves the others inside a collective
            ppo = {"error": "%s: %s" % (type(ex).__name__, ex)}     # of the leg: they time out with the process group (NCCL watchdog),
        if world > 1 and not all_ranks("error" not in (ppo or {})):   # and every rank that does return agrees on the outcome before
            ppo = ppo if (ppo and "error" in ppo) else {"error": "the PPO leg failed on another rank"}     # any further collective
    state["ppo"] = ppo
    if world > 1 and ppo and "error" not in ppo:          # after the PPO leg's numbers are safe in `state` (a hang below costs only its own entry)
        try:
            ppo["grad_allreduce"] = allreduce_leg(dev, world, ppo["bucket_numel"], ppo["optimizer_steps"], barrier)
            if dist.get_backend() == "nccl":
                guard.rearm(60)
                ppo["grad_allreduce"]["captured_probe"] = {"error": "did not finish (watchdog)"}
                ppo["grad_allreduce"]["captured_probe"] = captured_allreduce_probe(dev, ppo["bucket_numel"])
        except Exception as ex:
            ppo["grad_allreduce"] = {"error": "%s: %s" % (type(ex).__name__, ex)}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        state["cpu"] = cpu_baseline(cfg)
    guard.finish()
    if world > 1:                         # the line is out: a rank whose process group is unhealthy must not keep the job from ending
        bye = threading.Timer(60, lambda: os._exit(EXIT_WATCHDOG if args.strict else 0))
        bye.daemon = True
        bye.start()
        try:
            dist.barrier()
            dist.destroy_process_group()
        except Exception:
            pass


if __name__ == "__main__":
    main()
