"""GPU parity tests proper: the HIP engine, called through the C ABI (ctypes -> libsolorl_hip.so),
against the fp64 oracle on the same seeded inputs, against the committed golden fixtures, and --
at BASELINE.json's full size -- through size-independent properties.

Tolerances (north_star: joint angles within 1e-3 rad):
  * per control step from identical states (fp32 vs fp64): median joint error < 1e-4 rad, contact
    sets equal, reward/obs within 1e-3 for the median env;
  * 1000-step standing trajectory (golden fixture): max |dq| < 1e-3 rad;
  * violent / frictionally jammed states are chaotic in BOTH implementations (DESIGN.md "Solver
    sensitivity") and are judged statistically, never bit-for-bit.
"""
import os

import numpy as np
import pytest
import torch

from solorl_amd.config import (default_config, ROBOT_SOLO8, ROBOT_SOLO12, TASK_STAND, TASK_WALK, TASK_POINTGOAL,
                               CONTROL_PD, CONTROL_TORQUE)
from tests.util import GOLDEN, load_state, check_parity_stats
from tests.golden.make_golden import stand_cfg, stand_action

pytestmark = pytest.mark.gpu


def make(cfg, N, seed=1, off=0):
    from solorl_amd.vec_env import SoloVecEnv
    from oracle.oracle_py import Oracle
    try:
        threads = min(16, len(os.sched_getaffinity(0)))
    except AttributeError:
        threads = 4
    return (SoloVecEnv(cfg, N, device="cuda:0", seed=seed, env_id_offset=off),
            Oracle(cfg, N, seed=seed, env_id_offset=off, threads=threads))      # (OpenMP over envs: results do not depend on it)


def obs_diff(a, b, D):
    """|a-b| with the euler entries compared modulo the wrap of the reference's (euler % 2)/2 quirk
    (solo.py:206): an angle of -1e-9 reads 0.9999999995, +1e-9 reads 0 -- same pose."""
    d = np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))
    for base in range(0, d.shape[-1], D):
        e = d[..., base + 1:base + 4]
        d[..., base + 1:base + 4] = np.minimum(e, np.abs(1.0 - e))
    return d


def cfg_for(robot, task, control=CONTROL_TORQUE, **kw):
    c = default_config(robot, task); c.num_history_stack = 1; c.control = control
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def test_engine_library_is_loaded(gpu_device):
    from solorl_amd import _native
    assert os.path.exists(_native.LIB_PATH)
    maps = open("/proc/self/maps").read()
    _native.lib()
    assert "libsolorl_hip.so" in open("/proc/self/maps").read() or "libsolorl_hip.so" in maps


def test_handle_properties(gpu_device, monkeypatch):
    """solorl_get_property: what a handle runs (the sweep variant used to be a silent function of the grid size)."""
    from solorl_amd import _native
    c = cfg_for(ROBOT_SOLO12, TASK_WALK)
    env, _ = make(c, 8)
    assert env.get_property("lanes_per_env") == 16 and env.get_property("max_contacts") == 8 and env.get_property("max_limit_rows") == 4
    assert env.get_property("sweep_variant") == 2 and env.get_property("f64") == 0          # PyBullet's residual exit: one variant at every size
    with pytest.raises(_native.SoloRLError):
        env.get_property("no_such_property")
    c0 = cfg_for(ROBOT_SOLO12, TASK_WALK, solver_residual_threshold=0.0)
    small, _ = make(c0, 8)
    big, _ = make(c0, 8192)
    assert small.get_property("sweep_variant") == 1 and big.get_property("sweep_variant") == 0   # fixed sweeps: pipelined / plain by grid size
    monkeypatch.setenv("SOLORL_TEAM", "0")
    lane, _ = make(c, 8)
    assert lane.get_property("lanes_per_env") == 1


@pytest.mark.parametrize("robot,task", [(ROBOT_SOLO8, TASK_STAND), (ROBOT_SOLO12, TASK_WALK), (ROBOT_SOLO12, TASK_POINTGOAL)])
def test_reset_matches_oracle(gpu_device, robot, task):
    """reset = snapshot[K] on the GPU vs K live settle steps in the oracle; K and goals from Philox."""
    c = cfg_for(robot, task)
    env, orc = make(c, 256, seed=7)
    og = env.reset().cpu().numpy().astype(np.float64); oo = orc.reset()
    assert og.shape == oo.shape == (256, c.obs_dim)
    assert obs_diff(og, oo, c.state_dim).max() < 2e-3
    for i in (0, 17, 255):
        sg, so = env.get_state(i), orc.get_state(i)
        assert sg.rng_counter == so.rng_counter and sg.timestep == 0 and sg.contact_mask == so.contact_mask
        assert np.abs(np.array(sg.q) - np.array(so.q)).max() < 1e-4
        if task == TASK_POINTGOAL:
            assert np.abs(np.array(sg.goal) - np.array(so.goal)).max() < 1e-6    # same Philox draws
            assert abs(sg.potential - so.potential) < 1e-5


@pytest.mark.parametrize("robot,task,control", [
    (ROBOT_SOLO8, TASK_STAND, CONTROL_TORQUE), (ROBOT_SOLO8, TASK_WALK, CONTROL_PD),
    (ROBOT_SOLO12, TASK_WALK, CONTROL_TORQUE), (ROBOT_SOLO12, TASK_POINTGOAL, CONTROL_TORQUE),
    (ROBOT_SOLO12, TASK_STAND, CONTROL_PD)])
def test_step_matches_oracle_resynced(gpu_device, robot, task, control):
    """30 control steps of a random policy on 128 envs; before every step the oracle is reloaded with
    the engine's state, so each comparison is one control step (4 sub-steps) deep."""
    c = cfg_for(robot, task, control)
    N = 128
    env, orc = make(c, N, seed=3)
    env.reset(); orc.reset()
    rng = np.random.default_rng(0)
    n = env.act_dim
    dq_all, dr_all, dobs_all, done_mismatch, mask_mismatch = [], [], [], 0, 0
    for t in range(30):
        for i in range(N):
            orc.set_state(i, env.get_state(i))
        a = rng.uniform(-1.2, 1.2, size=(N, n)).astype(np.float32) * (0.3 if t < 15 else 1.0)
        obs, rew, done, infos = env.step(torch.from_numpy(a).cuda())
        oobs, orew, odone, oinfo = orc.step(a.astype(np.float64))
        done = done.cpu().numpy(); rew = rew.cpu().numpy()[:, 0]; obs = obs.cpu().numpy()
        done_mismatch += int((done != odone).sum())
        for i in range(N):
            sg, so = env.get_state(i), orc.get_state(i)
            if bool(done[i]) != bool(odone[i]):
                # the two may only disagree AT a termination threshold: the side that lives on sits within 1e-4 of the fall height
                # z = 0.05 (baseEnv.py:169) or of the goal radius 0.5 (solo.py:270)
                live = so if done[i] else sg
                assert abs(live.pos[2] - 0.05) < 1e-4 or (task == TASK_POINTGOAL and abs(live.potential - 0.5) < 1e-4), (i, t, live.pos[2], live.potential)
            if done[i] or odone[i]:
                continue
            dq_all.append(np.abs(np.array(sg.q)[:n] - np.array(so.q)[:n]).max())
            mask_mismatch += sg.contact_mask != so.contact_mask
        ok = (done == 0) & (odone == 0)
        dr_all += list(np.abs(rew - orew)[ok]); dobs_all += list(obs_diff(obs, oobs, c.state_dim)[ok].max(axis=1))
        t_info = infos.tensors
        assert np.array_equal(t_info["episode_length"].cpu().numpy()[ok], oinfo["episode_length"][ok])
    dq_all = np.array(dq_all)
    check_parity_stats("step_resynced/robot%d_task%d_control%d" % (robot, task, control), dq_all)
    assert np.median(dq_all) < 1e-4, np.median(dq_all)
    assert np.percentile(dq_all, 90) < 1e-3, np.percentile(dq_all, 90)         # (measured p90: 1e-6 .. 2e-6, tests/golden/parity_measured.json)
    assert np.median(dr_all) < 1e-3 and np.median(dobs_all) < 1e-3
    assert done_mismatch <= 2 and mask_mismatch <= 0.02 * len(dq_all)


def test_standing_trajectory_1000_steps_within_1e3_rad(gpu_device):
    """North-star tolerance: joint angles within 1e-3 rad of the fp64 oracle over 1000 control steps
    on fixed actions (golden fixture; settled crouch start, PD hold, sinusoidal references)."""
    c = stand_cfg()
    env, _ = make(c, 64)
    env.reset()
    s = load_state("stand_pd_start.json")
    for i in range(64):
        env.set_state(i, s)
    gold = np.load(os.path.join(GOLDEN, "stand_pd_traj.npz"))
    worst = 0.0
    for t in range(1000):
        a = torch.tensor(np.tile(stand_action(t), (64, 1)), dtype=torch.float32, device="cuda:0")
        env.step(a)
        if (t + 1) % 50 == 0:
            k = (t + 1) // 50 - 1
            for i in (0, 31, 63):
                sg = env.get_state(i)
                worst = max(worst, np.abs(np.array(sg.q) - gold["q"][k]).max())
                assert abs(sg.pos[2] - gold["z"][k]) < 1e-3
    assert worst < 1e-3, worst


def test_termination_and_autoreset_semantics(gpu_device):
    c = cfg_for(ROBOT_SOLO8, TASK_STAND, episode_length=5)
    env, orc = make(c, 64, seed=11)
    env.reset(); orc.reset()
    z = torch.zeros(64, 8, device="cuda:0")
    for t in range(5):
        obs, rew, done, infos = env.step(z)
        oobs, orew, odone, oinfo = orc.step(np.zeros((64, 8)))
    assert done.sum().item() == 64 and odone.sum() == 64                      # timeout at exactly episode_length
    d0 = infos[0]
    assert d0["timeout"] is True and d0["success"] is True and d0["episode_length"] == 5
    for k in ("episode_reward", "max_velocity", "min_force", "max_force", "dr/stand_rew", "dr/joint_pose_rew",
              "dr/torque_rew", "dr/roll_pitch_balance_rew", "dr/progress_rew", "goals_reached"):
        assert k in d0                                                         # keys read by agents/ppo/train.py:90-100
    assert abs(d0["dr/stand_rew"] - oinfo["dr"][0][0]) < 1e-5
    assert rew.shape == (64, 1) and done.dtype == torch.float32 and obs.dtype == torch.float32
    s = env.get_state(0)
    assert s.timestep == 0 and all(v == 0 for v in s.dr)                      # obs/state are post-reset
    assert obs_diff(obs.cpu().numpy(), oobs, c.state_dim).max() < 2e-3
    # fall -> -10, not timeout, not success
    c = cfg_for(ROBOT_SOLO12, TASK_WALK)
    env, _ = make(c, 4)
    env.reset()
    st = env.get_state(1); st.pos[2] = 0.04
    for leg in range(4):
        st.q[3 * leg + 1] = np.pi / 2; st.q[3 * leg + 2] = 0.0
    env.set_state(1, st)
    obs, rew, done, infos = env.step(torch.zeros(4, 12, device="cuda:0"))
    assert done[1].item() == 1.0 and rew[1, 0].item() == -10.0
    assert infos[1]["timeout"] is False and infos[1]["success"] is False and done[0].item() == 0.0


def test_step_before_reset_is_an_error(gpu_device):
    from solorl_amd.vec_env import SoloVecEnv
    from solorl_amd._native import SoloRLError
    env = SoloVecEnv(cfg_for(ROBOT_SOLO12, TASK_WALK), 8, device="cuda:0")
    with pytest.raises(SoloRLError, match="reset"):                            # baseEnv.py:43
        env.step(torch.zeros(8, 12, device="cuda:0"))
    with pytest.raises(AssertionError):                                        # solo.py:226
        env.reset(); env.step(torch.zeros(8, 11, device="cuda:0"))


def test_full_size_properties(gpu_device):
    """BASELINE size (4096 envs, Solo12 walk): determinism, finiteness, episode accounting and env
    independence -- properties that do not need the oracle to finish 4096 x 450 steps."""
    from solorl_amd.vec_env import SoloVecEnv
    c = cfg_for(ROBOT_SOLO12, TASK_WALK)
    N = 4096
    g = torch.Generator(device="cuda:0"); g.manual_seed(5)
    acts = torch.rand((32, N, 12), device="cuda:0", generator=g) * 2 - 1
    outs = []
    for rep in range(2):
        env = SoloVecEnv(c, N, device="cuda:0", seed=9)
        o = env.reset()
        n_done = torch.zeros(N, device="cuda:0"); ep_len_ok = True
        for t in range(450):
            o, r, d, info = env.step_inplace(acts[t % 32])
            assert torch.isfinite(o).all() and torch.isfinite(r).all()
            n_done += d.float()
            if d.any():
                el = info["episode_length"][d.bool()]
                ep_len_ok &= bool(((el >= 1) & (el <= 400)).all())
        outs.append((o.clone(), r.clone(), n_done.clone()))
        assert ep_len_ok and (n_done >= 1).all()            # every env ends at least once within 450 steps (T=400)
        assert info["nan_reset"].sum().item() == 0
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])   # bitwise reproducible
    # env independence / sharding: envs 2048.. of the big batch == a shard created with env_id_offset
    env_a = SoloVecEnv(c, N, device="cuda:0", seed=9)
    env_b = SoloVecEnv(c, N // 2, device="cuda:0", seed=9, env_id_offset=N // 2)
    oa, ob = env_a.reset(), env_b.reset()
    for t in range(40):
        oa, _, _, _ = env_a.step_inplace(acts[t % 32]); ob, _, _, _ = env_b.step_inplace(acts[t % 32][N // 2:].contiguous())
    assert torch.equal(oa[N // 2:], ob)


def test_pointgoal_bookkeeping_full_size(gpu_device):
    from solorl_amd.vec_env import SoloVecEnv
    c = cfg_for(ROBOT_SOLO12, TASK_POINTGOAL)
    env = SoloVecEnv(c, 4096, device="cuda:0", seed=2)
    o = env.reset()
    goal = o[:, 40:42] * 2
    assert ((goal.abs() >= 1.0) & (goal.abs() < 2.0)).all()                   # solo.py:327-330
    assert (goal > 0).any() and (goal < 0).any()
    env.increment_curriculum()
    a = torch.zeros(4096, 12, device="cuda:0")
    for t in range(60):
        o, r, d, info = env.step_inplace(a)
    goal = o[:, 40:42] * 2
    assert (goal.abs() < 3.0).all() and (goal.abs() >= 2.0).any()             # goal_radius 2 -> 3 after curriculum


def test_fp64_engine_matches_oracle_tightly(gpu_device):
    """precision = f64: the SAME kernels instantiated in double agree with the oracle to rounding,
    i.e. the HIP code computes the oracle's algorithm; fp32 differences are then pure precision."""
    from solorl_amd.config import PRECISION_F64
    c = cfg_for(ROBOT_SOLO12, TASK_WALK, precision=PRECISION_F64)
    N = 32
    env, orc = make(c, N, seed=4)
    og = env.reset().cpu().numpy(); oo = orc.reset()
    assert obs_diff(og, oo, c.state_dim).max() < 1e-5          # (obs are emitted as float32)
    rng = np.random.default_rng(2)
    errs = []
    for t in range(12):
        for i in range(N):
            orc.set_state(i, env.get_state(i))
        a = (0.3 * rng.uniform(-1, 1, size=(N, 12))).astype(np.float32)
        env.step(torch.from_numpy(a).cuda()); orc.step(a.astype(np.float64))
        for i in range(N):
            sg, so = env.get_state(i), orc.get_state(i)
            assert sg.contact_mask == so.contact_mask
            errs.append(np.abs(np.array(sg.q) - np.array(so.q)).max())
    errs = np.array(errs)
    check_parity_stats("fp64_engine_resynced", errs, floor=1e-13)
    assert np.median(errs) < 1e-11 and np.percentile(errs, 90) < 1e-7


def test_fp64_standing_trajectory(gpu_device):
    from solorl_amd.config import PRECISION_F64
    c = stand_cfg(); c.precision = PRECISION_F64
    env, _ = make(c, 4)
    env.reset()
    s = load_state("stand_pd_start.json")
    for i in range(4):
        env.set_state(i, s)
    gold = np.load(os.path.join(GOLDEN, "stand_pd_traj.npz"))
    worst = 0.0
    for t in range(1000):
        env.step(torch.tensor(np.tile(stand_action(t), (4, 1)), dtype=torch.float32, device="cuda:0"))
        if (t + 1) % 50 == 0:
            worst = max(worst, np.abs(np.array(env.get_state(0).q) - gold["q"][(t + 1) // 50 - 1]).max())
    # actions pass through float32 at the boundary (agents/ppo/envs.py:192), the oracle fixture used float64 actions
    assert worst < 2e-5, worst


def test_fused_returns_kernel_matches_reference_golden(gpu_device):
    """solorl_compute_returns (HIP) vs OPBuffer.compute_returns golden vectors of the reference."""
    from solorl_amd.ppo import RolloutStorage
    G = torch.load(os.path.join(GOLDEN, "ppo_golden.pt"), weights_only=False)
    T, N, O, A = 12, 5, 84, 12
    s = RolloutStorage(T, N, (O,), A, torch.device("cuda:0"))
    for k, v in G["buf"].items():
        getattr(s, k).copy_(v)
    s.compute_returns(G["next_value"].cuda(), True, 0.99, 0.95)
    assert torch.allclose(s.returns[:-1].cpu(), G["returns_gae"][:-1], atol=1e-5)
    assert torch.allclose(s.value_preds.cpu(), G["value_preds_after_gae"])
    s.compute_returns(G["next_value"].cuda(), False, 0.99, 0.95)
    assert torch.allclose(s.returns.cpu(), G["returns_disc"], atol=1e-5)
    # full size vs the torch formulation
    T, N = 400, 4096
    big = RolloutStorage(T, N, (4,), 2, torch.device("cuda:0"))
    g = torch.Generator(device="cuda:0"); g.manual_seed(0)
    big.rewards.copy_(torch.randn(big.rewards.shape, device="cuda:0", generator=g))
    big.value_preds.copy_(torch.randn(big.value_preds.shape, device="cuda:0", generator=g))
    big.masks.copy_((torch.rand(big.masks.shape, device="cuda:0", generator=g) > 0.02).float())
    nv = torch.randn(N, 1, device="cuda:0", generator=g)
    big.compute_returns(nv, True, 0.99, 0.95)
    cpu = RolloutStorage(T, N, (4,), 2, torch.device("cpu"))
    for k in ("rewards", "value_preds", "masks"):
        getattr(cpu, k).copy_(getattr(big, k).cpu())
    cpu.compute_returns(nv.cpu(), True, 0.99, 0.95)
    assert torch.allclose(big.returns[:-1].cpu(), cpu.returns[:-1], atol=2e-4, rtol=1e-4)


def test_lane_mode_sorting_and_team_mode_agree(gpu_device):
    """The three execution layouts of the same physics: team mode (16 lanes per env, default), lane mode
    (one env per lane, SOLORL_TEAM=0) and contact-count-sorted storage (SOLORL_SORT=1).  Sorting only permutes
    storage: bitwise neutral in lane mode; in team mode an env's sweep is specialised on the slot set of its
    wavefront (null rows, other predecessor slots), so like lane vs team it is a re-association of the same sums."""
    from solorl_amd.vec_env import SoloVecEnv
    c = cfg_for(ROBOT_SOLO12, TASK_WALK)
    N = 512
    g = torch.Generator(device="cuda:0"); g.manual_seed(11)
    acts = 0.5 * (torch.rand((8, N, 12), device="cuda:0", generator=g) * 2 - 1)

    def run(**envvars):
        old = {k: os.environ.get(k) for k in envvars}
        os.environ.update({k: str(v) for k, v in envvars.items()})
        try:
            env = SoloVecEnv(c, N, device="cuda:0", seed=3)
        finally:
            for k, v in old.items():
                if v is None: os.environ.pop(k, None)
                else: os.environ[k] = v
        o = env.reset()
        qs = []
        for t in range(8):
            o, r, d, _ = env.step_inplace(acts[t])
            qs.append(o.clone())
        return torch.stack(qs)

    team = run(SOLORL_TEAM=1, SOLORL_SORT=0)
    team_sorted = run(SOLORL_TEAM=1, SOLORL_SORT=1)
    lane = run(SOLORL_TEAM=0, SOLORL_SORT=0)
    lane_sorted = run(SOLORL_TEAM=0, SOLORL_SORT=1)
    assert torch.equal(lane, lane_sorted)
    for other in (team_sorted, lane):
        d = (team - other).abs()[:, :, 16:28]     # joint angles / 10 rad: the north-star quantity
        assert d.median().item() * 10 < 1e-5 and torch.quantile(d.flatten(), 0.9).item() * 10 < 1e-3


@pytest.mark.parametrize("n", [1, 3, 5, 63, 65])
def test_ragged_batch_sizes(gpu_device, n):
    """Batch sizes that do not fill a team wavefront (4 envs) or a workgroup round (multiples of 8 workgroups):
    idle teams must neither disturb their neighbours nor write anything."""
    c = cfg_for(ROBOT_SOLO12, TASK_WALK)
    env, orc = make(c, n, seed=6)
    og = env.reset().cpu().numpy().astype(np.float64); oo = orc.reset()
    assert og.shape == oo.shape == (n, c.obs_dim)
    assert obs_diff(og, oo, c.state_dim).max() < 2e-3
    rng = np.random.default_rng(n)
    guard = torch.full((n + 8, 12), 7.0, device="cuda:0")          # actions live in a larger buffer: nothing beyond n is read as an env
    dq = []
    for t in range(6):
        a = (0.4 * rng.uniform(-1, 1, size=(n, 12))).astype(np.float32)
        guard[:n] = torch.from_numpy(a).cuda()
        for i in range(n):
            orc.set_state(i, env.get_state(i))
        o, r, d, _ = env.step(guard[:n])
        orc.step(a.astype(np.float64))
        assert o.shape == (n, c.obs_dim) and torch.isfinite(o).all()
        dq += [np.abs(np.array(env.get_state(i).q) - np.array(orc.get_state(i).q)).max() for i in range(n)]
    check_parity_stats("ragged_batch_n%d" % n, dq)
    assert np.median(dq) < 1e-4 and np.max(dq) < 5e-2


def test_full_state_matches_oracle_resynced(gpu_device):
    """Every state field after one resynced control step, not only the joint angles: base pose and twist, joint
    rates and the cached normal impulses of the 20 collision primitives (medians: single envs may hit a
    non-convergent friction configuration of the 50-sweep PGS)."""
    c = cfg_for(ROBOT_SOLO12, TASK_WALK)
    N = 64
    env, orc = make(c, N, seed=8)
    env.reset(); orc.reset()
    rng = np.random.default_rng(4)
    err = {k: [] for k in ("pos", "quat", "lin_vel", "ang_vel", "qd", "lambda_prev")}
    for t in range(20):
        for i in range(N):
            orc.set_state(i, env.get_state(i))
        a = (0.5 * rng.uniform(-1, 1, size=(N, 12))).astype(np.float32)
        _, _, done, _ = env.step(torch.from_numpy(a).cuda())
        _, _, odone, _ = orc.step(a.astype(np.float64))
        done = done.cpu().numpy()
        for i in range(N):
            if done[i] or odone[i]:
                continue
            sg, so = env.get_state(i), orc.get_state(i)
            for k in err:
                err[k].append(np.abs(np.array(getattr(sg, k)) - np.array(getattr(so, k))).max())
    tol = dict(pos=2e-7, quat=1e-6, lin_vel=5e-6, ang_vel=5e-5, qd=5e-4, lambda_prev=5e-6)   # measured medians: 7e-9, 5e-8, 1e-7, 1e-6, 2e-5, 8e-8
    for k, v in err.items():
        assert len(v) > 1000 and np.median(v) < tol[k], (k, np.median(v))


def test_no_lds_read_before_write(gpu_device):
    """The same 450-step rollout (resets, joint limits, the 8-contact cap all occur) with the kernel's LDS pre-filled
    with zeros and with NaNs (SOLORL_POISON_LDS test hook) must agree bitwise: nothing may be read before it is
    written in the launch, and lane hand-offs through LDS must be ordered.  (A hand-off in the contact-cap path was
    not: the compiler had sunk the readers' load into the branch opposite the leader's store.)"""
    from solorl_amd.vec_env import SoloVecEnv
    c = cfg_for(ROBOT_SOLO12, TASK_WALK)
    N = 4096
    g = torch.Generator(device="cuda:0"); g.manual_seed(5)
    acts = torch.rand((32, N, 12), device="cuda:0", generator=g) * 2 - 1
    outs = []
    for word in ("0", "0x7fc00000"):
        os.environ["SOLORL_POISON_LDS"] = word
        try:
            env = SoloVecEnv(c, N, device="cuda:0", seed=9)
        finally:
            del os.environ["SOLORL_POISON_LDS"]
        env.reset()
        acc = torch.zeros(N, device="cuda:0"); nan = 0
        for t in range(450):
            o, r, d, info = env.step_inplace(acts[t % 32])
            acc += r
        outs.append((o.clone(), acc.clone(), int(info["nan_reset"].sum())))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and outs[0][2] == outs[1][2] == 0


def test_rollout_statistics_match_oracle(gpu_device):
    """No resynchronisation: 512 envs x 300 steps of the same random actions on the engine and on the oracle.
    Individual trajectories separate (contact-rich rigid-body dynamics is chaotic, and the engine computes in
    fp32), but the aggregate behaviour an RL algorithm sees must not: episode terminations, base height and the
    distribution of step rewards agree within sampling noise (quantiles: the reward has heavy tails -- velocity
    squared -- so its plain sum is dominated by a handful of events)."""
    c = cfg_for(ROBOT_SOLO12, TASK_WALK)
    N, T = 512, 300
    env, orc = make(c, N, seed=12)
    env.reset(); orc.reset()
    rng = np.random.default_rng(21)
    acts = rng.uniform(-1, 1, size=(32, N, 12)).astype(np.float32)
    g = dict(done=0, fell=0, z=0.0, rew=[]); o = dict(done=0, fell=0, z=0.0, rew=[])
    for t in range(T):
        a = acts[t % 32]
        obs, rew, done, info = env.step(torch.from_numpy(a).cuda())
        oobs, orew, odone, oinfo = orc.step(a.astype(np.float64))
        done = done.cpu().numpy(); rew = rew.cpu().numpy()[:, 0]
        g["done"] += int(done.sum()); o["done"] += int(odone.sum())
        g["rew"].append(rew[done == 0]); o["rew"].append(orew[odone == 0])        # (terminal steps carry the -10 / bonus)
        g["fell"] += int((rew == -10).sum()); o["fell"] += int((orew == -10).sum())
        g["z"] += float(obs[:, 0].mean()); o["z"] += float(oobs[:, 0].mean())
    assert g["done"] > 2000 and abs(g["done"] - o["done"]) < 0.05 * o["done"], (g["done"], o["done"])
    assert abs(g["fell"] - o["fell"]) < 0.05 * o["fell"], (g["fell"], o["fell"])
    # (mean base height: a few robots thrown into the air dominate it -- three seeds of tools/dev/agg_stats.py put the oracle at
    # 62.9 .. 68.4, the fp32 engine at 60.9 .. 63.5 and the fp64 engine at 63.2 .. 64.3; the termination counts agree to 0.5 %)
    assert abs(g["z"] - o["z"]) < 0.12 * o["z"], (g["z"], o["z"])
    rg, ro = np.concatenate(g["rew"]), np.concatenate(o["rew"])
    for qt in (0.1, 0.25, 0.5, 0.75, 0.9):
        a_, b_ = np.quantile(rg, qt), np.quantile(ro, qt)
        assert abs(a_ - b_) < 0.1 * abs(b_) + 0.02, (qt, a_, b_)
