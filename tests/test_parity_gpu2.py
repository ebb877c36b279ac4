"""GPU parity tests, second file: the workloads and branches the first round left to the oracle alone.

  * BASELINE config 5 -- configs/basic_contact.yaml as the in-repo PD path (Solo12 walk, PD gains [5, 0.2],
    episode length 50): resynced vs the oracle, and the full-size properties at 8192 envs/GPU;
  * the pointgoal goal-reached / bonus / resample branch and pointgoal-timeout-is-failure on the HIP path;
  * SURVEY.md 8(d)'s parity input verbatim, with the divergence horizon of the fp32 and fp64 engines read against
    the oracle's own horizon under a 1e-12 perturbation (fixture walk_torque_traj.npz);
  * curriculum under HIP-graph replay;
  * the treadmill strip (configs/basic.yaml unmodified);
  * the engine's finished-episode accumulators vs the per-step info tensors and vs the oracle.
All through the C ABI (ctypes -> libsolorl_hip.so)."""
import os

import numpy as np
import pytest
import torch

from solorl_amd.config import (default_config, config_from_dict, load_yaml, ROBOT_SOLO8, ROBOT_SOLO12, TASK_WALK,
                               TASK_POINTGOAL, CONTROL_PD, PRECISION_F64, EPSTAT_NAMES)
from tests.util import check_parity_stats, GOLDEN
from tests.golden.make_golden import walk_cfg, walk_action, divergence_horizon
from tests.test_parity_gpu import make, obs_diff, cfg_for

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def resync(orc, env, N):
    for i in range(N):
        orc.set_state(i, env.get_state(i))


# ------------------------------------------------------------------------------------------------ config 5
def config5():
    c = config_from_dict(load_yaml(os.path.join(ROOT, "configs", "basic_contact.yaml")))
    assert (c.robot, c.task, c.control, c.episode_length, c.num_history_stack) == (ROBOT_SOLO12, TASK_WALK, CONTROL_PD, 50, 1)
    assert (c.kp, c.kd, c.frame_skip) == (5.0, 0.2, 4)
    return c


def test_config5_pd_path_resynced_vs_oracle(gpu_device):
    """60 control steps (one episode end at t = 50 included) of a random PD policy on 128 envs, oracle reloaded with
    the engine's state before every step."""
    c = config5()
    N = 128
    env, orc = make(c, N, seed=5)
    env.reset(); orc.reset()
    rng = np.random.default_rng(50)
    dq, drew, dobs, timeouts = [], [], [], 0
    for t in range(60):
        resync(orc, env, N)
        a = rng.uniform(-1, 1, size=(N, 12)).astype(np.float32) * 0.15       # q_ref = 10 a: +-1.5 rad targets
        obs, rew, done, infos = env.step(torch.from_numpy(a).cuda())
        oobs, orew, odone, oinfo = orc.step(a.astype(np.float64))
        done = done.cpu().numpy(); rew = rew.cpu().numpy()[:, 0]
        assert np.array_equal(done != 0, odone != 0)
        ti = infos.tensors
        assert np.array_equal(ti["episode_length"].cpu().numpy(), oinfo["episode_length"])
        assert np.array_equal(ti["timeout"].cpu().numpy()[done != 0], oinfo["timeout"][odone != 0])
        timeouts += int(ti["timeout"].cpu().numpy()[done != 0].sum())
        ok = (done == 0)
        for i in np.nonzero(ok)[0]:
            dq.append(np.abs(np.array(env.get_state(i).q) - np.array(orc.get_state(i).q)).max())
        drew += list(np.abs(rew - orew)[ok]); dobs += list(obs_diff(obs.cpu().numpy(), oobs, c.state_dim)[ok].max(axis=1))
        if t == 49:
            assert done.sum() >= 0.4 * N                      # everybody still up times out at exactly episode_length = 50 (the others fell)
            assert obs_diff(obs.cpu().numpy(), oobs, c.state_dim)[done != 0].max() < 2e-3     # post-reset observations of the envs that ended
    dq = np.array(dq)
    assert timeouts >= 0.4 * N
    check_parity_stats("config5_pd_path", dq)
    assert np.median(dq) < 1e-4 and np.percentile(dq, 90) < 1e-3, (np.median(dq), np.percentile(dq, 90))      # (measured p90 2e-6)
    assert np.median(drew) < 1e-3 and np.median(dobs) < 1e-3


def test_config5_full_size_8192_envs(gpu_device):
    """BASELINE config 5 size: 8192 envs per GPU, 450 steps of a random PD policy.  Episodes never exceed 50 steps,
    every env is reset at least 9 times (450 / 50), the rollout is bit-reproducible, and a shard created with
    env_id_offset equals the matching slice of the big batch."""
    from solorl_amd.vec_env import SoloVecEnv
    c = config5()
    N = 8192
    g = torch.Generator(device="cuda:0"); g.manual_seed(15)
    acts = (torch.rand((32, N, 12), device="cuda:0", generator=g) * 2 - 1) * 0.3
    outs = []
    for rep in range(2):
        env = SoloVecEnv(c, N, device="cuda:0", seed=4)
        env.reset()
        n_done = torch.zeros(N, device="cuda:0"); max_len = 0; tout = 0
        for t in range(450):
            o, r, d, info = env.step_inplace(acts[t % 32])
            n_done += d.float()
            if d.any():
                max_len = max(max_len, int(info["episode_length"][d.bool()].max()))
                tout += int(info["timeout"][d.bool()].sum())
        assert torch.isfinite(o).all() and info["nan_reset"].sum().item() == 0
        assert max_len == 50 and int(n_done.min()) >= 9 and tout > 0
        st = env.episode_stats()
        assert st["episodes"] == int(n_done.sum()) and st["episode_length"] <= 50.0
        outs.append((o.clone(), r.clone(), n_done.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])
    # (the default solve -- PyBullet's residual exit -- is ONE sweep variant at every batch size, so a shard equals its slice bitwise
    # whatever the two sizes; only with solver_residual_threshold = 0 does the variant depend on the grid, see test_fixed_sweep_variants_vs_oracle)
    off = N // 4
    env_a = SoloVecEnv(c, N, device="cuda:0", seed=4)
    env_b = SoloVecEnv(c, N - off, device="cuda:0", seed=4, env_id_offset=off)
    env_a.reset(); env_b.reset()
    for t in range(60):
        oa, _, _, _ = env_a.step_inplace(acts[t % 32]); ob, _, _, _ = env_b.step_inplace(acts[t % 32][off:].contiguous())
    assert torch.equal(oa[off:], ob)


# ------------------------------------------------------------------------------------------------ pointgoal branches
def test_pointgoal_goal_reached_on_the_hip_path(gpu_device):
    """baseEnv.py:55-57,174-178 + solo.py:269-272: potential < 0.5 => goals_reached += 1, a new goal is drawn, the env
    reports done + success with reward 0.1 (T - t) and auto-resets -- engine vs oracle from the same injected state."""
    c = cfg_for(ROBOT_SOLO12, TASK_POINTGOAL)
    N = 16
    env, orc = make(c, N, seed=21)
    env.reset(); orc.reset()
    z = torch.zeros(N, 12, device="cuda:0")
    for t in range(3):
        env.step(z); orc.step(np.zeros((N, 12)))
    near = [1, 4, 7, 12]
    for i in near:                                         # goal 0.3 m from the base: inside the 0.5 m radius
        s = env.get_state(i); s.goal[0] = s.pos[0] + 0.3; s.goal[1] = s.pos[1]
        s.potential = 0.3
        env.set_state(i, s)
    resync(orc, env, N)
    rng_before = [env.get_state(i).rng_counter for i in range(N)]
    obs, rew, done, infos = env.step(z)
    oobs, orew, odone, oinfo = orc.step(np.zeros((N, 12)))
    done = done.cpu().numpy(); rew = rew.cpu().numpy()[:, 0]
    assert [int(d) for d in done] == [1 if i in near else 0 for i in range(N)] == [int(d) for d in odone]
    ti = infos.tensors
    for i in near:
        assert infos[i]["success"] is True and infos[i]["timeout"] is False
        assert rew[i] == pytest.approx(0.1 * (400 - 4), rel=1e-6) and orew[i] == pytest.approx(0.1 * (400 - 4))
        assert float(ti["goals_reached"][i]) == 1.0 == oinfo["goals_reached"][i]
        assert int(ti["episode_length"][i]) == 4 == oinfo["episode_length"][i]
        sg, so = env.get_state(i), orc.get_state(i)
        # draws: new goal at the goal-reached event (solo.py:272), then the reset's goal + settle count
        assert sg.rng_counter == so.rng_counter == rng_before[i] + 3
        assert np.allclose(sg.goal, so.goal, atol=1e-6) and sg.timestep == 0 and sg.goals_reached == 0 == sg.env_goals_reached
        assert abs(sg.potential - so.potential) < 1e-5
    for i in set(range(N)) - set(near):
        assert abs(rew[i] - orew[i]) < 1e-3 and "success" not in infos[i]
    assert obs_diff(obs.cpu().numpy(), oobs, c.state_dim).max() < 2e-3
    st = env.episode_stats()
    assert st["episodes"] == len(near) and st["success"] == 1.0 and st["episode_reward"] == pytest.approx(39.6, rel=1e-5)


def test_pointgoal_timeout_is_a_failure(gpu_device):
    c = cfg_for(ROBOT_SOLO12, TASK_POINTGOAL, episode_length=3)
    env, orc = make(c, 8, seed=2)
    env.reset(); orc.reset()
    z = torch.zeros(8, 12, device="cuda:0")
    for t in range(3):
        obs, rew, done, infos = env.step(z); _, orew, odone, oinfo = orc.step(np.zeros((8, 12)))
    assert done.sum().item() == 8 and odone.sum() == 8
    assert all(infos[i]["timeout"] is True and infos[i]["success"] is False for i in range(8))      # baseEnv.py:166
    assert np.abs(rew.cpu().numpy()[:, 0] - orew).max() < 1e-3                                        # no -10, no bonus


# ------------------------------------------------------------------------------------------------ 8(d) parity input
def test_walk_torque_parity_input_divergence_horizon(gpu_device):
    """SURVEY.md 8(d) parity run verbatim on the HIP engine (see tests/test_host_harness.py for the regime: the robot
    is on the ground after 20 steps and the fp64 oracle itself, perturbed by 1e-12 rad, leaves the 1e-3 rad band after
    `oracle_self_horizon` = 106 steps with round 4's model (friction cone, hull-profile feet) -- 20 with round 3's, 61 with round 2's: one e-fold every ~5 steps).  Divergence horizon = first control step with max |dq| > 1e-3 rad.
    Actions cross the boundary as float32 (agents/ppo/envs.py:190-192 in the reference as well), so the engines are
    compared with the oracle driven by the SAME float32-rounded actions; that rounding alone (3e-8 relative on the
    torques) moves the oracle off its own float64-action fixture within a couple of dozen steps."""
    from oracle.oracle_py import Oracle
    g = np.load(os.path.join(GOLDEN, "walk_torque_traj.npz"))
    self_h = int(g["oracle_self_horizon"])
    T = 150
    acts32 = np.stack([walk_action(t).astype(np.float32) for t in range(T)])
    o64 = Oracle(walk_cfg(), 1, seed=1); o64.reset()
    o32 = Oracle(walk_cfg(), 1, seed=1); o32.reset()
    q64, q32 = [], []
    for t in range(T):
        o64.step(walk_action(t)[None]); o32.step(acts32[t].astype(np.float64)[None])
        q64.append(np.array(o64.get_state(0).q)); q32.append(np.array(o32.get_state(0).q))
    q64, q32 = np.array(q64), np.array(q32)
    assert np.abs(q64[:40] - g["q"][:40]).max() < 1e-6            # the committed fixture is what the oracle computes here
    h_round = divergence_horizon(np.abs(q32 - q64).max(axis=1))
    hor = {}
    for name, prec in (("f32", 0), ("f64", PRECISION_F64)):
        c = walk_cfg(); c.precision = prec
        env, _ = make(c, 4, seed=1)
        env.reset()
        dq = []
        for t in range(T):
            env.step(torch.tensor(np.tile(acts32[t], (4, 1)), device="cuda:0"))
            dq.append(max(np.abs(np.array(env.get_state(i).q) - q32[t]).max() for i in (0, 3)))
        hor[name] = divergence_horizon(dq)
        assert max(dq[:10]) < (1e-9 if name == "f64" else 5e-4)
    print("divergence horizons (control steps): oracle self (1e-12 perturbation) %d, oracle under float32 action rounding %d, "
          "engine fp64 %d, engine fp32 %d" % (self_h, h_round, hor["f64"], hor["f32"]))
    assert hor["f64"] >= self_h - 25, hor  # as long as the oracle's own horizon (106), within the scatter of a chaotic run
    assert hor["f32"] >= 20, hor           # the same growth rate from fp32 rounding instead of 1e-12 (g++ build of the kernel math: 35 steps)


# (the fp32-outlier census lives in tests/test_parity_gpu3.py: test_error_tail_* -- every resynced workload, strict)


# ------------------------------------------------------------------------------------------------ curriculum + graphs
def test_curriculum_is_seen_by_a_replayed_graph(gpu_device):
    """solo.py:332-334 / agents/ppo/train.py:116-117 on the default (HIP graph) training path: goal_radius lives in a
    device block the kernel reads, so a rollout graph captured BEFORE increment_curriculum() samples wider goals when
    replayed after it."""
    from solorl_amd.vec_env import SoloVecEnv
    from solorl_amd.ppo import Policy, RolloutStorage
    from solorl_amd.ppo.graphs import GraphedRollout
    c = cfg_for(ROBOT_SOLO12, TASK_POINTGOAL, episode_length=20)
    N, T = 1024, 25
    env = SoloVecEnv(c, N, device="cuda:0", seed=3)
    torch.manual_seed(0)
    pol = Policy(env.observation_space.shape, env.action_space, None, {"hidden_size": 32}).to("cuda:0")
    st = RolloutStorage(T, N, env.observation_space.shape, env.act_dim, torch.device("cuda:0"))
    st.obs[0].copy_(env.reset())
    with torch.no_grad():
        pol.act(st.obs[0])
    roll = GraphedRollout(env, pol, st, T)
    roll()                                                 # captures + replays once: every env resets (T > episode_length)
    torch.cuda.synchronize()
    assert roll.graph is not None
    goal = st.obs[-1][:, 40:42] * 2
    assert (goal.abs() < 2.0).all() and (goal.abs() >= 1.0).all()
    env.increment_curriculum()                             # radius 2 -> 3
    st.reset(); roll()
    torch.cuda.synchronize()
    goal = st.obs[-1][:, 40:42] * 2
    assert (goal.abs() < 3.0).all() and (goal.abs() >= 2.0).any(), float(goal.abs().max())


# ------------------------------------------------------------------------------------------------ treadmill
def test_treadmill_configs_basic_yaml_vs_oracle(gpu_device):
    """configs/basic.yaml unmodified (Solo8 walk, use_treadmill: True): reset (strip side from the env's Philox stream,
    snapshot per side), 40 resynced control steps; feet over the strip are reported by the sensor (the plane lies under it)."""
    c = config_from_dict(load_yaml(os.path.join(ROOT, "configs", "basic.yaml")))
    assert c.use_treadmill == 1 and c.robot == ROBOT_SOLO8 and c.task == TASK_WALK
    N = 128
    env, orc = make(c, N, seed=13)
    og = env.reset().cpu().numpy().astype(np.float64); oo = orc.reset()
    assert obs_diff(og, oo, c.state_dim).max() < 2e-3
    ys = np.array([env.get_state(i).treadmill_y for i in range(N)])
    assert np.allclose(ys, [orc.get_state(i).treadmill_y for i in range(N)], atol=1e-6) and set(np.round(ys, 4)) == {-0.49, 0.49}
    for i in range(N):
        assert env.get_state(i).contact_mask == orc.get_state(i).contact_mask and env.get_state(i).rng_counter == orc.get_state(i).rng_counter == 2
    feet = og[:, 26:30]                                     # Solo8: 10 + 2*8 = 26 .. 29: FL FR HL HR
    assert np.array_equal(feet, np.ones_like(feet))          # all four feet down and reported, strip or not
    strip_bits = np.array([(env.get_state(i).contact_mask >> 24) & 0xF for i in range(N)])
    assert np.array_equal(strip_bits[ys > 0], np.full((ys > 0).sum(), 0b0101)) and np.array_equal(strip_bits[ys < 0], np.full((ys < 0).sum(), 0b1010))
    rng = np.random.default_rng(3)
    dq, mism, strip_seen = [], 0, 0
    for t in range(40):
        resync(orc, env, N)
        a = rng.uniform(-1, 1, size=(N, 8)).astype(np.float32) * (0.3 if t < 20 else 1.0)
        obs, rew, done, _ = env.step(torch.from_numpy(a).cuda())
        oobs, orew, odone, _ = orc.step(a.astype(np.float64))
        done = done.cpu().numpy()
        for i in range(N):
            if done[i] or odone[i]:
                continue
            sg, so = env.get_state(i), orc.get_state(i)
            dq.append(np.abs(np.array(sg.q)[:8] - np.array(so.q)[:8]).max())
            mism += sg.contact_mask != so.contact_mask
            strip_seen += (sg.contact_mask >> 24) != 0
    dq = np.array(dq)
    check_parity_stats("treadmill_basic_yaml", dq)
    assert np.median(dq) < 1e-4 and np.percentile(dq, 90) < 1e-3, (np.median(dq), np.percentile(dq, 90))
    assert mism <= 0.02 * len(dq) and strip_seen > 0.3 * len(dq)
    # the strip matters: the same rollout without it separates from this one
    c0 = c.copy(); c0.use_treadmill = 0
    env0, _ = make(c0, N, seed=13)
    env0.reset()
    envt, _ = make(c, N, seed=13)
    envt.reset()
    a = torch.zeros(N, 8, device="cuda:0"); a[:, 0::2] = 0.5
    for t in range(30):
        o0, _, _, _ = env0.step(a); ot, _, _, _ = envt.step(a)
    assert (o0[:, :10] - ot[:, :10]).abs().max().item() > 1e-3


# ------------------------------------------------------------------------------------------------ episode statistics
def test_episode_stat_accumulators(gpu_device):
    """finished-episode statistics reduced on the device over EVERY step.  (1) The engine's accumulators
    equal, exactly, the same sums formed from its per-step info tensors; (2) over a 450-step unsynchronised rollout
    they agree with the oracle's per-env infos within sampling noise."""
    c = cfg_for(ROBOT_SOLO12, TASK_WALK, episode_length=100)
    N, T = 512, 450
    env, orc = make(c, N, seed=31)
    env.reset(); orc.reset()
    rng = np.random.default_rng(5)
    acts = rng.uniform(-1, 1, size=(32, N, 12)).astype(np.float32)
    keys = ("episode_reward", "episode_length", "success", "dr_stand", "dr_joint_pose", "dr_torque", "dr_balance", "dr_progress")
    acc = torch.zeros(9, N, device="cuda:0")
    oacc = np.zeros(9)
    for t in range(T):
        obs, rew, done, infos = env.step(torch.from_numpy(acts[t % 32]).cuda())
        ti = infos.tensors
        d = done.bool()
        acc[0] += d.float()
        for k, name in enumerate(keys):
            acc[1 + k] += torch.where(d, ti[name].float(), torch.zeros_like(acc[0]))
        _, orew, odone, oinfo = orc.step(acts[t % 32].astype(np.float64))
        od = odone != 0
        oacc[0] += od.sum(); oacc[1] += oinfo["episode_reward"][od].sum(); oacc[2] += oinfo["episode_length"][od].sum()
        oacc[3] += oinfo["success"][od].sum(); oacc[4:9] += oinfo["dr"][od].sum(axis=0)
    raw = env._ep_stats.clone()
    assert torch.equal(raw[:9], acc) and raw[9].sum().item() == 0
    st = env.episode_stats()
    assert env._ep_stats.abs().sum().item() == 0                                  # zeroed by the read
    n = acc[0].sum().item()
    assert st["episodes"] == int(n) and n > 2000
    assert st["episode_length"] == pytest.approx(acc[2].sum().item() / n, rel=1e-5)
    names = EPSTAT_NAMES[1:9]
    assert abs(n - oacc[0]) < 0.05 * oacc[0], (n, oacc[0])
    for k, name in enumerate(names):
        mine, theirs = st[name], oacc[1 + k] / oacc[0]
        if name in ("dr/progress_rew", "episode_reward"):       # 2 sign(vx) vx^2 is heavy-tailed: its mean over ~2500 unsynchronised
            assert abs(mine - theirs) < 2.0, (name, mine, theirs)   # chaotic episodes is dominated by a few events (quantiles: test_rollout_statistics_match_oracle)
        else:
            assert abs(mine - theirs) < 0.1 * abs(theirs) + 0.05, (name, mine, theirs)


# ------------------------------------------------------------------------------------------------ K7: fixed sweeps (option) and the residual exit (default)
@pytest.mark.parametrize("warmstart", [0.0, 0.85])
def test_fixed_sweep_variants_vs_oracle(gpu_device, warmstart):
    """solver_residual_threshold = 0 (fixed 50 sweeps, round 2's default): above one wavefront per SIMD (more than 4096 envs on an
    MI355X) the team-mode sweep then runs without the software pipelining (3 instructions fewer per slot).  Both variants forced here at a
    size the oracle can follow (SOLORL_PGS_PIPE=0 / 1): same per-step bounds as the default (residual-exit) variant, and the two agree with
    each other to rounding.  With and without the multibody warm start (the sweep's set-up skips the warm-start accumulation when the
    config has none: both paths are exercised)."""
    from solorl_amd.vec_env import SoloVecEnv
    from oracle.oracle_py import Oracle
    c = cfg_for(ROBOT_SOLO12, TASK_WALK, solver_residual_threshold=0.0, warmstart=warmstart)
    N = 128
    envs = {}
    for pipe in (0, 1):
        os.environ["SOLORL_PGS_PIPE"] = str(pipe)
        try:
            envs[pipe] = SoloVecEnv(c, N, device="cuda:0", seed=3)
        finally:
            del os.environ["SOLORL_PGS_PIPE"]
        envs[pipe].reset()
        assert envs[pipe].get_property("sweep_variant") == pipe        # 0 plain, 1 pipelined, 2 residual exit
    orc = Oracle(c, N, seed=3, threads=8); orc.reset()
    rng = np.random.default_rng(0)
    dq, dv = [], []
    for t in range(30):
        for i in range(N):
            s = envs[0].get_state(i)
            orc.set_state(i, s); envs[1].set_state(i, s)
        a = rng.uniform(-1.2, 1.2, size=(N, 12)).astype(np.float32) * (0.3 if t < 15 else 1.0)
        ta = torch.from_numpy(a).cuda()
        _, _, d0, _ = envs[0].step(ta); _, _, d1, _ = envs[1].step(ta); _, _, od, _ = orc.step(a.astype(np.float64))
        d0 = d0.cpu().numpy(); d1 = d1.cpu().numpy()
        for i in range(N):
            if d0[i] or d1[i] or od[i]:
                continue
            q0 = np.array(envs[0].get_state(i).q)
            dq.append(np.abs(q0 - np.array(orc.get_state(i).q)).max())
            dv.append(np.abs(q0 - np.array(envs[1].get_state(i).q)).max())
    assert np.median(dq) < 1e-4 and np.percentile(dq, 90) < 1e-3, (np.median(dq), np.percentile(dq, 90))
    assert np.median(dv) < 1e-5, np.median(dv)


@pytest.mark.parametrize("team", [1, 0])
def test_residual_threshold_early_exit_vs_oracle(gpu_device, team):
    """solver_residual_threshold = 1e-7 (PyBullet's solverResidualThreshold; the default since round 3): every env's
    solve ends after the first sweep within the threshold -- per env, also when the four envs of a team-mode wavefront
    finish after different sweeps.  Resynced against the oracle running the same rule; team mode (default) and lane mode."""
    from solorl_amd.vec_env import SoloVecEnv
    from oracle.oracle_py import Oracle
    c = cfg_for(ROBOT_SOLO12, TASK_WALK, solver_residual_threshold=1e-7)
    N = 128
    os.environ["SOLORL_TEAM"] = str(team)
    try:
        env = SoloVecEnv(c, N, device="cuda:0", seed=3)
    finally:
        del os.environ["SOLORL_TEAM"]
    orc = Oracle(c, N, seed=3, threads=8)
    env.reset(); orc.reset()
    rng = np.random.default_rng(0)
    dq, its, mism = [], [], 0
    for t in range(30):
        resync(orc, env, N)
        a = rng.uniform(-1.2, 1.2, size=(N, 12)).astype(np.float32) * (0.3 if t < 15 else 1.0)
        _, _, done, _ = env.step(torch.from_numpy(a).cuda())
        _, _, odone, _ = orc.step(a.astype(np.float64))
        done = done.cpu().numpy()
        for i in range(N):
            if done[i] or odone[i]:
                continue
            sg, so = env.get_state(i), orc.get_state(i)
            dq.append(np.abs(np.array(sg.q) - np.array(so.q)).max()); mism += sg.contact_mask != so.contact_mask
            if so.contact_mask:
                its.append(orc.last_iterations(i))
    dq, its = np.array(dq), np.array(its)
    print("early exit (%s mode): median |dq| %.1e, p90 %.1e; oracle sweeps per solve: median %d, %.0f %% below 50" % (
        "team" if team else "lane", np.median(dq), np.percentile(dq, 90), np.median(its), 100 * (its < 50).mean()))
    check_parity_stats("residual_exit_%s" % ("team" if team else "lane"), dq)
    assert np.median(its) < 20 and (its < 50).mean() > 0.6
    assert np.median(dq) < 1e-4 and np.percentile(dq, 90) < 2e-3, (np.median(dq), np.percentile(dq, 90))
    assert mism <= 0.02 * len(dq)


# ------------------------------------------------------------------------------------------------ K2: URDF inertia
@pytest.mark.parametrize("robot", [ROBOT_SOLO8, ROBOT_SOLO12])
def test_urdf_inertia_vs_oracle(gpu_device, robot):
    """use_urdf_inertia = 1 (SURVEY Appendix B K2 "implement both"): the URDF tensors incl. their products of inertia, whose
    signs mirror left/right, in the team-mode leg phase (per-lane sign patterns) -- fp64 engine to rounding, fp32 engine to
    the usual per-step bound; and it is a different robot from the box-inertia default."""
    n = 8 if robot == ROBOT_SOLO8 else 12
    outs = {}
    for prec, tol in ((PRECISION_F64, 1e-10), (0, 1e-4)):
        c = cfg_for(robot, TASK_WALK, use_urdf_inertia=1, precision=prec)
        N = 32
        env, orc = make(c, N, seed=4)
        og = env.reset().cpu().numpy().astype(np.float64); oo = orc.reset()
        assert obs_diff(og, oo, c.state_dim).max() < 2e-3
        rng = np.random.default_rng(2)
        errs = []
        for t in range(12):
            resync(orc, env, N)
            a = (0.4 * rng.uniform(-1, 1, size=(N, n))).astype(np.float32)
            _, _, d, _ = env.step(torch.from_numpy(a).cuda()); _, _, od, _ = orc.step(a.astype(np.float64))
            d = d.cpu().numpy()
            for i in range(N):
                if not (d[i] or od[i]):
                    errs.append(np.abs(np.array(env.get_state(i).q) - np.array(orc.get_state(i).q)).max())
        assert np.median(errs) < tol, (prec, np.median(errs))
        outs[prec] = np.array(env.get_state(0).q)
    c0 = cfg_for(robot, TASK_WALK, precision=PRECISION_F64)
    env0, _ = make(c0, 4, seed=4)
    env0.reset()
    rng = np.random.default_rng(2)
    for t in range(12):
        env0.step(torch.from_numpy((0.4 * rng.uniform(-1, 1, size=(32, n))).astype(np.float32)[:4]).cuda())
    assert np.abs(np.array(env0.get_state(0).q) - outs[PRECISION_F64]).max() > 1e-4      # the inertia model matters


@pytest.mark.parametrize("robot", [ROBOT_SOLO8, ROBOT_SOLO12])
@pytest.mark.parametrize("team", ["1", "0"])
def test_joint_limit_rows_vs_oracle(gpu_device, monkeypatch, robot, team):
    """K5: up to four joint-limit rows (one per joint at or beyond its limit; the most violated four when more compete), rows 3 and
    4 at the first normal positions of the team-mode sweep, contacts shifted behind them (dynamics.hpp MAX_LIMITS).  States
    injected with 1-6 joints beyond their limits, the robot standing on the ground (contact rows behind the limit rows) or lying on
    it (many contacts: the cap 8 - (limit rows - 2) binds); the oracle solves the same rows (oracle_set_caps(8, 4): engine
    emulation, an algebra check); one control step, fp64 engine to rounding, fp32 engine to the usual per-step bound."""
    monkeypatch.setenv("SOLORL_TEAM", team)
    n = 8 if robot == ROBOT_SOLO8 else 12
    for prec, tol in ((PRECISION_F64, 1e-9), (0, 2e-3)):
        c = cfg_for(robot, TASK_WALK, precision=prec)
        N = 96
        env, orc = make(c, N, seed=5)
        orc.set_caps(8, 4)
        env.reset(); orc.reset()
        rng = np.random.default_rng(7)
        nbeyond = []
        for i in range(N):
            s = env.get_state(i)
            k = int(rng.integers(1, 7))
            for j in rng.choice(n, size=k, replace=False):
                s.q[j] = float(rng.choice([-1, 1]) * (10.0 + rng.uniform(0.0, 0.3)))
                s.qd[j] = float(rng.uniform(-10, 10))
            if i % 3 == 2:                                   # lying on the belly: base points + knees + feet in contact
                s.pos[2] = 0.03
            nbeyond.append(k)
            env.set_state(i, s)
        resync(orc, env, N)
        a = rng.uniform(-1, 1, size=(N, n)).astype(np.float32)
        _, _, d, _ = env.step(torch.from_numpy(a).cuda()); _, _, od, _ = orc.step(a.astype(np.float64))
        d = d.cpu().numpy()
        counts = np.array([orc.last_counts(i) for i in range(N)])
        assert (np.array(nbeyond) >= 3).sum() > N // 3 and counts[:, 0].max() >= 6
        errs = [np.abs(np.array(env.get_state(i).q)[:n] - np.array(orc.get_state(i).q)[:n]).max() for i in range(N) if not (d[i] or od[i])]
        masks = [env.get_state(i).contact_mask == orc.get_state(i).contact_mask for i in range(N) if not (d[i] or od[i])]
        assert len(errs) > N // 2 and np.mean(masks) > 0.97
        assert np.median(errs) < tol and np.percentile(errs, 90) < 50 * tol, (prec, np.median(errs), np.percentile(errs, 90))


@pytest.mark.parametrize("cfg_file", ["basic12.yaml", "basic.yaml"])
def test_random_policy_keeps_robots_on_the_ground_full_size(gpu_device, cfg_file):
    """Full-size physical-plausibility property (no oracle needed): under a Gaussian random policy, torque-limited 2.5 kg robots
    neither fly nor exceed a few m/s.  Round 2 found 0.42 % of such steps with |reward| > 20 -- robots at z = 5-23 m and 50 m/s,
    launched by a joint thrown back from radians past its limit (joint-limit rows) -- in engine and oracle
    alike, which no engine-vs-oracle comparison can see."""
    from solorl_amd.vec_env import SoloVecEnv
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = load_yaml(os.path.join(root, "configs", cfg_file)); d["task"] = "walk"
    c = config_from_dict(d)
    N = 4096
    env = SoloVecEnv(c, N, device="cuda:0", seed=1); env.reset()
    g = torch.Generator(device="cuda:0"); g.manual_seed(0)
    vmax = torch.zeros((), device="cuda:0"); zmax = torch.zeros((), device="cuda:0"); rmax = torch.zeros((), device="cuda:0")
    qmax = torch.zeros((), device="cuda:0")
    n = env.act_dim
    for t in range(500):
        o, r, dn, _ = env.step_inplace(torch.randn(N, n, device="cuda:0", generator=g))
        live = dn == 0
        vmax = torch.maximum(vmax, (o[:, 4:7].norm(dim=1) * live).max()); zmax = torch.maximum(zmax, (o[:, 0] * live).max())
        rmax = torch.maximum(rmax, (r.abs() * live).max()); qmax = torch.maximum(qmax, (o[:, 10:10 + n].abs() * live[:, None]).max() * 10.0)
    env.close()
    assert vmax.item() < 6.0 and zmax.item() < 0.6 and rmax.item() < 15.0, (vmax.item(), zmax.item(), rmax.item())
    assert qmax.item() < 10.7, qmax.item()          # joints stay within a fraction of a radian of their +-10 rad range
