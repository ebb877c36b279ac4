"""GPU parity tests, third file (round 4; all through the C ABI, ctypes -> libsolorl_hip.so):

  * TRAINED policies (tests/golden/policies/*.pt: README recipe on this engine, tools/dev/train_fixture_policies.sh) rolled out
    deterministically on the fp32 engine, the fp64 engine, the oracle and the oracle with Bullet-style hull manifolds: the states an RL
    user reaches, not random flailing -- success rate, episode length, reward quantiles within sampling error;
  * the error TAIL of every resynced workload explained or bounded: a sample above the north-star tolerance must be a state the fp64
    engine amplifies too (or a contact flipping at its threshold), quiet states are held to a maximum;
  * BASELINE config 2 verbatim at full size (configs/basic.yaml, task stand, 4096 envs);
  * the friction PYRAMID and contact ERP 0.2 of rounds 1-3, now options, in team mode, lane mode and fp64;
  * `num_history_stack` 3 and 4.
"""
import os
import time

import numpy as np
import pytest
import torch

from solorl_amd.config import (default_config, config_from_dict, load_yaml, ROBOT_SOLO8, ROBOT_SOLO12, TASK_STAND, TASK_WALK,
                               TASK_POINTGOAL, CONTROL_PD, PRECISION_F64, FRICTION_PYRAMID, FRICTION_CONE)
from tests.util import check_parity_stats, GOLDEN
from tests.test_parity_gpu import make, obs_diff, cfg_for

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
POLICY_DIR = os.path.join(GOLDEN, "policies")


# ------------------------------------------------------------------------------------------------ trained policies
def _policy(name, obs_dim, act_dim, device):
    from solorl_amd.ppo import Policy
    from solorl_amd.vec_env import Box
    pol = Policy((obs_dim,), Box(-np.ones(act_dim), np.ones(act_dim)), None, {"hidden_size": 64})
    pol.load_state_dict(torch.load(os.path.join(POLICY_DIR, name + ".pt"), map_location="cpu", weights_only=True))
    return pol.to(device).eval()


def _rollout_stats(step, reset, act, T):
    """Closed loop for T steps: obs -> act(obs) -> step.  Returns per-episode and per-step statistics."""
    obs = reset()
    ep_len, ep_succ, rew_live, z = [], [], [], []
    for t in range(T):
        obs, rew, done, length, success = step(act(obs))
        d = done != 0
        ep_len += list(length[d]); ep_succ += list(success[d])
        rew_live.append(rew[~d]); z.append(float(obs[:, 0].mean()))
    rew_live = np.concatenate(rew_live)
    return dict(episodes=len(ep_len), success=float(np.mean(ep_succ)) if ep_len else float("nan"), length=float(np.mean(ep_len)) if ep_len else float("nan"),
                length_sd=float(np.std(ep_len)) if ep_len else 0.0, z=float(np.mean(z)),
                r10=float(np.percentile(rew_live, 10)), r50=float(np.median(rew_live)), r90=float(np.percentile(rew_live, 90)))


@pytest.mark.parametrize("stochastic", [False, True])
@pytest.mark.parametrize("name,cfg_file,task", [("stand8", "basic.yaml", "stand"), ("pointgoal12", "basic12.yaml", "pointgoal"),
                                                 ("walk8", "basic.yaml", "walk"), ("walk12", "basic12.yaml", "walk")])
def test_trained_policy_statistics_engine_vs_oracle(gpu_device, name, cfg_file, task, stochastic):
    """VERDICT r03 "Next" #2.  The same deterministic policy (action = mean, agents/ppo/policy.py:33-45 `deterministic=True`) -- and the
    same policy as it was trained, action = mean + exp(logstd) * noise with ONE host-side noise sequence shared by all simulators (the
    mean action alone is brittle for some tasks: `stand8` succeeds 0.65 without its exploration noise, 0.99 with it, on engine and oracle
    alike) -- for 450 steps x 256 envs on four simulators from the same seeds.  Trajectories separate (chaos, fp32), the statistics an RL user sees
    must not: success rate within the binomial error of the two samples (+ 0.03), mean episode length within 3 standard errors (+ 3 %),
    reward quantiles of the non-terminal steps within 10 % (+ 0.05), mean base height within 3 %.
    The hull-manifold oracle (`contact_model = 1`: Bullet's collision scheme [K6] -- every link's hull, up to four PERSISTENT points per
    link -- which the engine's one-point primitives approximate) is held to the same success / length bounds for the
    stand and pointgoal policies.  For the WALKING gait it is not, and the test records that instead of hiding it: a gait trained on
    one-point contacts succeeds 0.9 here and 0.2-0.5 on the manifold model (measured round 4; feet alone account for it: their hull SHAPE
    is exact since round 4, the persistent multi-point patch under each foot is what the engine does not have).  How much of that is the
    gait's own brittleness: on the primitives it was trained on, 400 solver sweeps instead of 50 alone take it from 0.89 to 0.17 (DESIGN.md
    section 3 K6 iii).  That is the open K6 item."""
    from oracle.oracle_py import Oracle
    from solorl_amd.vec_env import SoloVecEnv
    d = load_yaml(os.path.join(ROOT, "configs", cfg_file)); d["task"] = task
    c = config_from_dict(d)
    N, T = 256, 450
    dev = torch.device("cuda:0")
    pol_gpu = _policy(name, c.obs_dim, c.n_joints, dev)
    pol_cpu = _policy(name, c.obs_dim, c.n_joints, torch.device("cpu"))
    noise = np.random.default_rng(77).standard_normal((T, N, c.n_joints)).astype(np.float32) if stochastic else None
    std = pol_cpu.pi_dist.logstd.detach().exp().numpy().reshape(1, -1)
    res = {}
    for label, prec in (("engine_f32", 0), ("engine_f64", PRECISION_F64)):
        cc = c.copy(); cc.precision = prec
        env = SoloVecEnv(cc, N, device=dev, seed=5)

        def step(a, env=env):
            o, r, dn, info = env.step(a)
            ti = info.tensors
            return (o.cpu().numpy().astype(np.float64), r.cpu().numpy()[:, 0], dn.cpu().numpy(), ti["episode_length"].cpu().numpy(),
                    ti["success"].cpu().numpy())

        clock = [0]

        def act(obs, clock=clock):
            with torch.no_grad():
                a = pol_gpu.act(torch.from_numpy(obs.astype(np.float32)).to(dev), deterministic=True)[1]
            if stochastic:
                a = a + torch.from_numpy(std * noise[clock[0]]).to(dev)
            clock[0] += 1
            return a.contiguous()

        t0 = time.time()
        res[label] = _rollout_stats(step, lambda env=env: env.reset().cpu().numpy().astype(np.float64), act, T)
        res[label]["seconds"] = time.time() - t0
        env.close()
    try:
        threads = min(8, len(os.sched_getaffinity(0)))
    except AttributeError:
        threads = 4
    torch_threads = torch.get_num_threads()
    torch.set_num_threads(1)              # the CPU policy is a 76 x 64 x 64 MLP on 256 rows: torch's default (one thread per host core, 64 on the GPU box
                                          # against a 16-core quota) spends its time in its own thread pool -- 13 of a rollout's 15 s
    for label, model in (("oracle", 0), ("oracle_hull_manifolds", 1)):
        Nm = N
        orc = Oracle(c, Nm, seed=5, threads=threads)
        if model:
            orc.set_contact_model(1)

        def ostep(a, orc=orc):
            o, r, dn, info = orc.step(a)
            return o, r, dn, info["episode_length"], info["success"]

        oclock = [0]

        def oact(obs, oclock=oclock):          # observations cross the boundary as float32 (agents/ppo/envs.py:192), here too
            with torch.no_grad():
                a = pol_cpu.act(torch.from_numpy(obs.astype(np.float32)), deterministic=True)[1].numpy()
            if stochastic:
                a = a + std * noise[oclock[0]][:a.shape[0]]
            oclock[0] += 1
            return a.astype(np.float64)

        t0 = time.time()
        res[label] = _rollout_stats(ostep, orc.reset, oact, T)
        res[label]["seconds"] = time.time() - t0
    torch.set_num_threads(torch_threads)
    for k, v in res.items():
        print("trained[%s%s] %-22s episodes %4d success %.3f length %6.1f z %.4f reward p10/p50/p90 %.3f / %.3f / %.3f  (%.1f s)" % (
            name, " + noise" if stochastic else "", k, v["episodes"], v["success"], v["length"], v["z"], v["r10"], v["r50"], v["r90"], v["seconds"]))
    out = os.path.join(ROOT, "gpurun_out")
    try:
        import json
        os.makedirs(out, exist_ok=True)
        path = os.path.join(out, "trained_policy_stats.json")
        cur = json.load(open(path)) if os.path.exists(path) else {}
        cur[name + ("_stochastic" if stochastic else "_deterministic")] = res
        json.dump(cur, open(path, "w"), indent=1, sort_keys=True)
    except OSError:
        pass
    ref = res["oracle"]
    assert ref["episodes"] >= N          # every env finished at least one episode in 450 steps
    assert ref["success"] > 0.5          # the fixture really is a trained policy on the oracle too (a random one never succeeds at walk / pointgoal)

    # independent samples: with the noise every env is its own trajectory; WITHOUT it envs that share their reset draws (settle count 5..11,
    # treadmill side) follow the same trajectory on a deterministic simulator -- 7 or 14 distinct ones per reset (pointgoal: the goal is per env)
    distinct = N if (stochastic or task == "pointgoal") else 7 * (2 if c.use_treadmill else 1)

    def close(a, b, full=True):
        pa, pb = a["success"], b["success"]
        na = min(a["episodes"], distinct * max(1, round(a["episodes"] / N))); nb = min(b["episodes"], distinct * max(1, round(b["episodes"] / N)))
        p = (pa * na + pb * nb) / (na + nb)
        assert abs(pa - pb) <= 3.0 * np.sqrt(max(p * (1 - p), 1e-4) * (1.0 / na + 1.0 / nb)) + 0.03, ("success", pa, pb)
        se = np.sqrt(a["length_sd"] ** 2 / na + b["length_sd"] ** 2 / nb)
        assert abs(a["length"] - b["length"]) <= 3.0 * se + 0.03 * b["length"], ("length", a["length"], b["length"])
        if full:
            assert abs(a["z"] - b["z"]) <= 0.03 * b["z"], ("z", a["z"], b["z"])
            rtol = 0.1 if distinct == N else 0.25         # (few distinct trajectories: the quantiles are not those of independent samples either)
            for q in ("r10", "r50", "r90"):
                assert abs(a[q] - b[q]) <= rtol * abs(b[q]) + 0.05, (q, a[q], b[q])

    close(res["engine_f32"], ref)
    close(res["engine_f64"], ref)
    close(res["engine_f32"], res["engine_f64"])
    hm = res["oracle_hull_manifolds"]
    if task != "walk":
        close(hm, ref, full=False)       # K6: success and episode length only
    else:                                # K6, open: recorded, not asserted (see the docstring); sanity only
        print("trained[%s] K6 gap: success %.3f on the one-point primitives vs %.3f on Bullet-style hull manifolds" % (name, ref["success"], hm["success"]))
        assert hm["episodes"] >= N and 0.0 <= hm["success"] <= 1.0


# ------------------------------------------------------------------------------------------------ the error tail, every workload
def _three_way_census(name, c32, N, T, action, min_samples):
    """fp32 engine, fp64 engine and oracle step from the SAME (float32-representable) states.  Returns (e32, e64, flip): per-sample joint
    angle errors of the two engines against the oracle and whether the fp32 engine's contact set differs from the oracle's."""
    c64 = c32.copy(); c64.precision = PRECISION_F64
    env32, orc = make(c32, N, seed=3)
    env64, _ = make(c64, N, seed=3)
    env32.reset(); env64.reset(); orc.reset()
    n = env32.act_dim
    e32, e64, flip = [], [], []
    for t in range(T):
        for i in range(N):
            s = env32.get_state(i)
            orc.set_state(i, s); env64.set_state(i, s)
        a = action(t, N, n)
        ta = torch.from_numpy(a).cuda()
        _, _, d32, _ = env32.step(ta); _, _, d64, _ = env64.step(ta); _, _, od, _ = orc.step(a.astype(np.float64))
        d32 = d32.cpu().numpy(); d64 = d64.cpu().numpy()
        for i in range(N):
            if d32[i] or d64[i] or od[i]:
                continue
            so, s32 = orc.get_state(i), env32.get_state(i)
            qo = np.array(so.q)[:n]
            e32.append(np.abs(np.array(s32.q)[:n] - qo).max())
            e64.append(np.abs(np.array(env64.get_state(i).q)[:n] - qo).max())
            flip.append((s32.contact_mask & 0xFFFFFF) != (so.contact_mask & 0xFFFFFF))
    e32, e64, flip = np.array(e32), np.array(e64), np.array(flip)
    assert len(e32) >= min_samples, len(e32)
    med64 = np.median(e64)
    out32 = e32 > 1e-3
    amplified = e64 > 1e2 * med64
    quiet = (e64 < 1e1 * med64) & ~flip
    unexplained = out32 & ~amplified & ~flip
    print("tail[%s]: samples %d, median e64 %.1e, max e64 %.1e; fp32 > 1e-3 rad: %d (%.3f %%), of which the fp64 engine amplifies (>= 100 x its median) %d, "
          "contact set flipped in fp32 %d, unexplained %d; amplified states %.2f %% of all; quiet states %d, their max e32 %.1e" % (
              name, len(e32), med64, e64.max(), out32.sum(), 100 * out32.mean(), (out32 & amplified).sum(), (out32 & flip & ~amplified).sum(), unexplained.sum(),
              100 * amplified.mean(), quiet.sum(), e32[quiet].max()))
    check_parity_stats("tail/" + name, e32)
    assert med64 < 1e-13, med64
    assert out32.sum() <= 0.02 * len(e32)
    assert unexplained.sum() == 0, "fp32 samples above the north-star tolerance that the fp64 engine does not amplify: %s" % e32[unexplained][:8]
    assert flip.mean() <= 0.02
    assert quiet.sum() > 0.8 * len(e32) and e32[quiet].max() < 1e-3, e32[quiet].max()
    return e32, e64


def _uniform(scale_early, scale_late, switch, seed=0):
    rng = np.random.default_rng(seed)
    return lambda t, N, n: (rng.uniform(-1.2, 1.2, size=(N, n)) * (scale_early if t < switch else scale_late)).astype(np.float32)


def test_error_tail_walk_torque(gpu_device):
    """Solo12 walk, torque (the headline workload's physics): round 3's census, now strict."""
    _three_way_census("solo12_walk_torque", cfg_for(ROBOT_SOLO12, TASK_WALK), 128, 40, _uniform(0.3, 1.0, 15), 3000)


def test_error_tail_config5_pd(gpu_device):
    """BASELINE config 5: configs/basic_contact.yaml (Solo12 walk, PD gains [5, 0.2], episode length 50)."""
    c = config_from_dict(load_yaml(os.path.join(ROOT, "configs", "basic_contact.yaml")))
    _three_way_census("config5_pd", c, 128, 45, _uniform(0.15, 0.15, 0, seed=50), 3000)


def test_error_tail_solo8_stand(gpu_device):
    """BASELINE configs 1-2: Solo8 stand, torque."""
    _three_way_census("solo8_stand_torque", cfg_for(ROBOT_SOLO8, TASK_STAND), 128, 40, _uniform(0.3, 1.0, 15, seed=2), 3000)


def test_error_tail_treadmill(gpu_device):
    """configs/basic.yaml unmodified: Solo8 walk over the treadmill strip."""
    c = config_from_dict(load_yaml(os.path.join(ROOT, "configs", "basic.yaml")))
    _three_way_census("treadmill_basic_yaml", c, 128, 40, _uniform(0.3, 1.0, 20, seed=3), 3000)


def test_error_tail_pointgoal(gpu_device):
    """BASELINE config 4's task: configs/basic12.yaml unmodified (Solo12 pointgoal)."""
    c = config_from_dict(load_yaml(os.path.join(ROOT, "configs", "basic12.yaml")))
    _three_way_census("pointgoal_basic12_yaml", c, 128, 40, _uniform(0.3, 1.0, 15, seed=4), 3000)


# ------------------------------------------------------------------------------------------------ BASELINE config 2 verbatim
def test_config2_solo8_stand_full_size(gpu_device):
    """BASELINE config 2: "Solo8 'stand', 4096 envs on 1 x MI355X, random policy rollout (sim-only correctness)" -- configs/basic.yaml with
    the task set to stand (config 1's file; the treadmill strip is on in it).  Size-independent properties at the full size: bitwise
    reproducible, finite, episode accounting, a shard equals its slice of the big batch, robots stay on the ground; and the first 128
    envs of the 4096 against the oracle (same Philox streams: global env id), resynced, per control step."""
    from solorl_amd.vec_env import SoloVecEnv
    from oracle.oracle_py import Oracle
    d = load_yaml(os.path.join(ROOT, "configs", "basic.yaml")); d["task"] = "stand"
    c = config_from_dict(d)
    assert (c.robot, c.task, c.use_treadmill, c.episode_length, c.num_history_stack) == (ROBOT_SOLO8, TASK_STAND, 1, 400, 1)
    N = 4096
    g = torch.Generator(device="cuda:0"); g.manual_seed(6)
    acts = torch.rand((32, N, 8), device="cuda:0", generator=g) * 2 - 1
    outs = []
    for rep in range(2):
        env = SoloVecEnv(c, N, device="cuda:0", seed=9)
        o = env.reset()
        n_done = torch.zeros(N, device="cuda:0"); ok_len = True
        zmax = torch.zeros((), device="cuda:0"); vmax = torch.zeros((), device="cuda:0")
        for t in range(450):
            o, r, dn, info = env.step_inplace(acts[t % 32])
            assert torch.isfinite(o).all() and torch.isfinite(r).all()
            n_done += dn.float()
            live = dn == 0
            zmax = torch.maximum(zmax, (o[:, 0] * live).max()); vmax = torch.maximum(vmax, (o[:, 4:7].norm(dim=1) * live).max())
            if dn.any():
                el = info["episode_length"][dn.bool()]
                ok_len &= bool(((el >= 1) & (el <= 400)).all())
        assert ok_len and (n_done >= 1).all() and info["nan_reset"].sum().item() == 0
        assert zmax.item() < 0.6 and vmax.item() < 6.0, (zmax.item(), vmax.item())
        st = env.episode_stats()
        assert st["episodes"] == int(n_done.sum())
        outs.append((o.clone(), r.clone(), n_done.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])
    env_a = SoloVecEnv(c, N, device="cuda:0", seed=9)
    env_b = SoloVecEnv(c, N // 2, device="cuda:0", seed=9, env_id_offset=N // 2)
    oa, ob = env_a.reset(), env_b.reset()
    for t in range(40):
        oa, _, _, _ = env_a.step_inplace(acts[t % 32]); ob, _, _, _ = env_b.step_inplace(acts[t % 32][N // 2:].contiguous())
    assert torch.equal(oa[N // 2:], ob)
    # the first 128 envs of the full-size batch vs the oracle
    M = 128
    env = SoloVecEnv(c, N, device="cuda:0", seed=9)
    orc = Oracle(c, M, seed=9, threads=8)
    og = env.reset().cpu().numpy().astype(np.float64); oo = orc.reset()
    assert obs_diff(og[:M], oo, c.state_dim).max() < 2e-3
    dq, mism = [], 0
    for t in range(25):
        for i in range(M):
            orc.set_state(i, env.get_state(i))
        a = acts[t % 32]
        _, _, dn, _ = env.step(a)
        _, _, od, _ = orc.step(a[:M].cpu().numpy().astype(np.float64))
        dn = dn.cpu().numpy()
        for i in range(M):
            if dn[i] or od[i]:
                continue
            sg, so = env.get_state(i), orc.get_state(i)
            dq.append(np.abs(np.array(sg.q)[:8] - np.array(so.q)[:8]).max()); mism += sg.contact_mask != so.contact_mask
    check_parity_stats("config2_solo8_stand_4096", dq)
    assert np.median(dq) < 1e-4 and np.percentile(dq, 90) < 1e-3 and mism <= 0.02 * len(dq)


# ------------------------------------------------------------------------------------------------ rounds 1-3's friction pyramid as an option
@pytest.mark.parametrize("mode", ["team", "lane", "f64"])
@pytest.mark.parametrize("fm,cerp", [(FRICTION_PYRAMID, 0.2), (FRICTION_CONE, 0.2), (FRICTION_PYRAMID, 0.08)])
def test_friction_model_and_contact_erp_options_vs_oracle(gpu_device, monkeypatch, mode, fm, cerp):
    """solorl_config friction_model / contact_erp ([K] ledger): the values that are NOT the defaults -- round 3's pair
    (pyramid, 0.2) and each change alone -- resynced against the oracle running the same rule, in the team-mode sweep, the lane-mode row
    loop and the fp64 instantiation.  (The defaults -- cone, 0.08 -- are what every other test of the suite runs.)"""
    if mode == "lane":
        monkeypatch.setenv("SOLORL_TEAM", "0")
    c = cfg_for(ROBOT_SOLO12, TASK_WALK, friction_model=fm, contact_erp=cerp, precision=PRECISION_F64 if mode == "f64" else 0)
    N = 128
    env, orc = make(c, N, seed=3)
    env.reset(); orc.reset()
    rng = np.random.default_rng(0)
    dq, mism = [], 0
    for t in range(30):
        for i in range(N):
            orc.set_state(i, env.get_state(i))
        a = rng.uniform(-1.2, 1.2, size=(N, 12)).astype(np.float32) * (0.3 if t < 15 else 1.0)
        _, _, done, _ = env.step(torch.from_numpy(a).cuda())
        _, _, odone, _ = orc.step(a.astype(np.float64))
        done = done.cpu().numpy()
        for i in range(N):
            if done[i] or odone[i]:
                continue
            sg, so = env.get_state(i), orc.get_state(i)
            dq.append(np.abs(np.array(sg.q) - np.array(so.q)).max()); mism += sg.contact_mask != so.contact_mask
    dq = np.array(dq)
    check_parity_stats("friction_option/%s_%s_erp%.2f" % (mode, "cone" if fm else "pyramid", cerp), dq, floor=1e-13 if mode == "f64" else 1e-7)
    if mode == "f64":
        assert np.median(dq) < 1e-11 and np.percentile(dq, 90) < 1e-7
    else:
        assert np.median(dq) < 1e-4 and np.percentile(dq, 90) < 1e-3, (np.median(dq), np.percentile(dq, 90))
    assert mism <= 0.02 * len(dq)


# ------------------------------------------------------------------------------------------------ history depth beyond two levels
@pytest.mark.parametrize("team", ["1", "0"])
@pytest.mark.parametrize("robot,task,H", [(ROBOT_SOLO12, TASK_POINTGOAL, 3), (ROBOT_SOLO8, TASK_WALK, 4)])
def test_history_depth_beyond_two_levels(gpu_device, monkeypatch, team, robot, task, H):
    """`num_history_stack` 3 and 4 (the reference takes any depth: solo.py:48 `deque(maxlen=num_history_stack)`; VERDICT r03 Missing #5).
    Levels 0 and 1 ride in registers, deeper ones shift in HBM.  Unsynchronised comparison of the WHOLE observation -- [s, s - h0, ... ,
    s - h(H-1)] -- with the oracle over 12 steps from reset (resynchronised state: the history is part of it), including auto-resets
    (short episodes) and, for pointgoal, the new goal patched into every level."""
    monkeypatch.setenv("SOLORL_TEAM", team)
    c = cfg_for(robot, task, num_history_stack=H, episode_length=7)
    N = 64
    env, orc = make(c, N, seed=8)
    og = env.reset().cpu().numpy().astype(np.float64); oo = orc.reset()
    assert og.shape == oo.shape == (N, c.state_dim * (1 + H))
    assert obs_diff(og, oo, c.state_dim).max() < 2e-3
    rng = np.random.default_rng(1)
    n = env.act_dim
    worst = 0.0
    for t in range(16):
        for i in range(N):
            orc.set_state(i, env.get_state(i))
        a = (0.3 * rng.uniform(-1, 1, size=(N, n))).astype(np.float32)
        obs, rew, done, _ = env.step(torch.from_numpy(a).cuda())
        oobs, orew, odone, _ = orc.step(a.astype(np.float64))
        done = done.cpu().numpy()
        assert np.array_equal(done != 0, odone != 0)
        d = obs_diff(obs.cpu().numpy(), oobs, c.state_dim)
        # the deltas of the DEEPER levels are differences of states several steps apart: fp32 rounding of each, nothing more
        worst = max(worst, float(np.median(d.max(axis=1))))
        assert np.percentile(d.max(axis=1), 90) < 2e-3, (t, np.percentile(d.max(axis=1), 90))
        if t in (6, 13):
            assert done.sum() == N           # timeout at episode_length = 7: every env resets, history refilled from the snapshot
        sg, so = env.get_state(3), orc.get_state(3)
        for lv in range(H):
            assert np.abs(np.array(sg.hist[lv])[:c.state_dim] - np.array(so.hist[lv])[:c.state_dim]).max() < 2e-3, (t, lv)
    assert worst < 1e-4, worst
