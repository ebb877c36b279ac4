"""The HIP engine's physics source (solorl_amd/csrc/{spatial,dynamics}.hpp) compiled by g++ for ONE
lane (tests/host/) and checked against the fp64 oracle.  The two are different formulations
(articulated-body recursion + base-space PGS  vs  dense mass matrix + Cholesky + generalized PGS),
so agreement to rounding is the "ABA vs CRBA+Cholesky cross-check" of SURVEY.md section 7.
This is a TEST build of the kernel source; nothing in solorl_amd/ can call it."""
import numpy as np
import pytest

from solorl_amd.config import default_config, ROBOT_SOLO8, ROBOT_SOLO12, TASK_WALK
from oracle.oracle_py import Oracle
from tests.host import harness_py
from tests.util import clone, state_vec, load_state
from tests.golden.make_golden import stand_cfg, stand_action


@pytest.mark.parametrize("robot,n,urdf_inertia", [(ROBOT_SOLO8, 8, 0), (ROBOT_SOLO12, 12, 0), (ROBOT_SOLO8, 8, 1), (ROBOT_SOLO12, 12, 1)])
def test_free_flight_matches_dense_dynamics(robot, n, urdf_inertia):
    """urdf_inertia = 1: SURVEY K2's other branch -- the URDF tensors with their products of inertia (ixy on the shoulders,
    iyz on the upper and lower legs, signs mirrored left/right) instead of Bullet's default box rule."""
    c = default_config(robot, TASK_WALK); c.use_urdf_inertia = urdf_inertia
    rng = np.random.default_rng(0)
    for trial in range(5):
        o = Oracle(c, 1)
        s = o.get_state(0); s.pos[2] = 5.0
        for j in range(n):
            s.q[j] = rng.uniform(-2, 2); s.qd[j] = rng.uniform(-5, 5); s.tau[j] = rng.uniform(-3, 3)
        s.ang_vel[:] = list(rng.uniform(-2, 2, 3)); s.lin_vel[:] = list(rng.uniform(-1, 1, 3))
        q = rng.normal(size=4); q /= np.linalg.norm(q); s.quat[:] = list(q)
        o.set_state(0, s); h = clone(s)
        o.substep(0); harness_py.substep(h, c, False)
        assert np.abs(state_vec(o.get_state(0), n) - state_vec(h, n)).max() < 1e-11


@pytest.mark.parametrize("robot,n,treadmill,urdf_inertia", [(ROBOT_SOLO8, 8, 0, 0), (ROBOT_SOLO12, 12, 0, 0), (ROBOT_SOLO8, 8, 1, 0),
                                                            (ROBOT_SOLO12, 12, 1, 0), (ROBOT_SOLO12, 12, 0, 1)])
def test_contact_substeps_resynced(robot, n, treadmill, urdf_inertia):
    """Drop, land and thrash under random torques; every sub-step starts from the oracle's state.
    The contact sets must be identical.  Errors are judged statistically: Bullet-style PGS with
    mu = 1 box friction is a non-convergent fixed-point iteration in some multi-contact states
    (DESIGN.md "Solver sensitivity"; frictionless it converges to 1e-16), where 50 iterations
    amplify rounding by many orders of magnitude in BOTH implementations."""
    c = default_config(robot, TASK_WALK)
    c.use_treadmill = treadmill        # strip under the left feet: per-contact friction 0.5, strip bits in the mask
    c.use_urdf_inertia = urdf_inertia
    rng = np.random.default_rng(1)
    o = Oracle(c, 1)
    o.set_caps(8, 4)       # algebra check: the oracle solves the rows the engine's slots hold (what the caps cost: test_oracle_caps.py)
    if treadmill:
        s0 = o.get_state(0); s0.treadmill_y = 0.49; o.set_state(0, s0)
    e64, e32, strip = [], [], 0
    saw_contacts = 0
    for k in range(400):
        so = o.get_state(0)
        if k % 4 == 0:
            tau = rng.uniform(-1.0, 1.0, size=n)
        for j in range(n):
            so.tau[j] = tau[j] if k % 4 == 0 else 0.0
        o.set_state(0, so)
        h64, h32 = clone(so), clone(so)
        o.substep(0)
        harness_py.substep(h64, c, False); harness_py.substep(h32, c, True)
        a = o.get_state(0)
        assert a.contact_mask == h64.contact_mask
        saw_contacts += bin(a.contact_mask & 0xFFFFFF).count("1") > 0
        strip += (a.contact_mask >> 24) != 0
        e64.append(np.abs(state_vec(a, n) - state_vec(h64, n)).max())
        e32.append(np.abs(state_vec(a, n) - state_vec(h32, n)).max())
    e64, e32 = np.array(e64), np.array(e32)
    assert saw_contacts > 100 and (strip > 20 if treadmill else strip == 0)
    # (max: a solve whose residual sits at the K7 threshold may stop one sweep earlier in one implementation than in the
    # other; the continuing one then moves on by about the threshold, 3e-4 m/s, per sweep)
    assert np.median(e64) < 1e-11 and np.percentile(e64, 90) < 1e-8 and e64.max() < 2e-2
    assert np.median(e32) < 2e-4 and np.percentile(e32, 90) < 2e-2


def test_joint_limit_row():
    c = default_config(ROBOT_SOLO12, TASK_WALK)
    for jj, qv, qdv in [(2, 9.9, 30), (0, 9.9, 30), (1, -9.95, -40), (2, 10.1, 5)]:
        o = Oracle(c, 1)
        s = o.get_state(0); s.pos[2] = 5.0; s.q[jj] = qv; s.qd[jj] = qdv
        o.set_state(0, s); h = clone(s)
        o.substep(0); harness_py.substep(h, c, False)
        assert np.abs(state_vec(o.get_state(0), 12) - state_vec(h, 12)).max() < 1e-10


def test_joint_limit_selection_more_candidates_than_rows():
    """Four joints inside their limit windows, two rows: the engine's phase_detect picks the same two (smallest margins) as the oracle."""
    c = default_config(ROBOT_SOLO12, TASK_WALK)
    rng = np.random.default_rng(1)
    for trial in range(20):
        o = Oracle(c, 1)
        s = o.get_state(0); s.pos[2] = 5.0
        for j in rng.choice(12, size=4, replace=False):
            s.q[j] = float(rng.choice([-1, 1]) * (10.0 + rng.uniform(-0.45, 0.45)))
            s.qd[j] = float(rng.uniform(-20, 20))
        o.set_state(0, s); h = clone(s)
        for _ in range(3):
            o.substep(0); harness_py.substep(h, c, False)
        assert np.abs(state_vec(o.get_state(0), 12) - state_vec(h, 12)).max() < 1e-9


def test_standing_trajectory_fp32_within_1e3_rad():
    """The 1000-step trajectory-parity run (tests/golden: settled crouch start, PD hold, gentle
    sinusoidal references) through the kernel math in fp32 on the CPU: joint angles stay within the
    north-star tolerance 1e-3 rad of the fp64 oracle fixture."""
    c = stand_cfg()
    s = load_state("stand_pd_start.json")
    gold = np.load(__import__("os").path.join(__import__("tests.util").util.GOLDEN, "stand_pd_traj.npz"))
    h = clone(s)
    worst = 0.0
    for t in range(1000):
        a = stand_action(t)
        for ss in range(c.frame_skip):
            for j in range(12):   # PD torque computed once per control step from the pre-step state (solo.py:240-252)
                if ss == 0:
                    qref = np.clip(a[j], -1, 1) * 10
                    h.tau[j] = float(np.clip(c.kp * (qref - h.q[j]) - c.kd * h.qd[j], -3, 3))
            tau = [h.tau[j] for j in range(12)]
            harness_py.substep(h, c, True)
            for j in range(12):   # K8: cleared after the first sub-step unless hold_torque
                h.tau[j] = tau[j] if c.hold_torque else 0.0
        if (t + 1) % 50 == 0:
            k = (t + 1) // 50 - 1
            worst = max(worst, np.abs(np.array(h.q) - gold["q"][k]).max())
    assert worst < 1e-3, worst


def test_walk_torque_parity_input_divergence_horizon():
    """SURVEY.md 8(d) parity input verbatim (Solo12 walk, torque control, K = 8, a = 0.5 sin(2 pi t/60 + j pi/6), default
    torque lifetime, termination off).  1.5 N.m on a 2.5 kg robot folds it to the ground within 20 control steps and it
    thrashes there: a chaotic input.  The fp64 oracle ITSELF, perturbed by 1e-12 rad, leaves the 1e-3 rad band after ~100 steps
    (fixture `oracle_self_horizon`: 106; 75-100 over the round's model variants -- the scatter of a chaotic run) with round 4's
    defaults -- Bullet's implicit friction cone, contact ERP 0.08, the feet's hull profile -- against 54 with the friction PYRAMID on
    the same model (its corners are discontinuities of the solve) and 20 with round 3's model (pyramid AND a foot primitive whose
    contact point jumped 8 mm sideways at a tilt of 6 degrees): nine decades in ~100 steps = one e-fold every ~5 steps.  Divergence
    horizon = first step with max |dq| > 1e-3 rad vs the fixture: the kernel math in fp64 holds as long as the oracle's own horizon
    (measured 108); in fp32 the same growth rate starts from ~5e-5 rad at step 10 instead of 1e-13 and crosses 1e-3 at step 35 (17 in
    round 3) -- no fp32 engine can hold this input longer, whatever it computes."""
    import os
    from tests.golden.make_golden import walk_cfg, walk_action, divergence_horizon
    from tests.util import GOLDEN
    g = np.load(os.path.join(GOLDEN, "walk_torque_traj.npz"))
    self_h = int(g["oracle_self_horizon"])
    assert self_h == divergence_horizon(g["pert_dq"]) and 55 <= self_h <= 130         # measured 106 (cone, contact ERP 0.08, margin 1 mm, ring-profile feet); 20 in round 3
    c = walk_cfg()
    o = Oracle(c, 1, seed=1); o.reset()
    h = {True: clone(o.get_state(0)), False: clone(o.get_state(0))}
    dq = {True: [], False: []}
    for t in range(125):
        a = walk_action(t)
        o.step(a[None])
        assert np.array_equal(np.array(o.get_state(0).q), g["q"][t])          # the oracle reproduces its fixture exactly
        for use_float in (True, False):
            af = a.astype(np.float32).astype(np.float64) if use_float else a   # actions cross the boundary as float32
            for ss in range(c.frame_skip):
                for j in range(12):
                    h[use_float].tau[j] = float(np.clip(af[j], -1, 1) * c.max_torque) if ss == 0 else 0.0    # K8
                harness_py.substep(h[use_float], c, use_float)
            dq[use_float].append(np.abs(np.array(h[use_float].q) - g["q"][t]).max())
    h64, h32 = divergence_horizon(dq[False]), divergence_horizon(dq[True])
    assert h64 >= self_h - 15, (h64, self_h)                                   # measured 108
    assert h32 >= 22, h32                                                      # measured 35
    assert max(dq[False][:10]) < 1e-11 and max(dq[True][:10]) < 5e-4          # before the fall: rounding only
    # the same input with round 3's friction pyramid: the oracle's own horizon is five times shorter
    cp = walk_cfg(); cp.friction_model = 0; cp.contact_erp = 0.2; cp.collision_margin = 0.0
    a_, b_ = Oracle(cp, 1, seed=1), Oracle(cp, 1, seed=1)
    a_.reset(); b_.reset()
    s = b_.get_state(0); s.q[0] += 1e-12; b_.set_state(0, s)
    pert = []
    for t in range(self_h):
        a_.step(walk_action(t)[None]); b_.step(walk_action(t)[None])
        pert.append(np.abs(np.array(a_.get_state(0).q) - np.array(b_.get_state(0).q)).max())
    assert 20 <= divergence_horizon(pert) <= self_h - 20, divergence_horizon(pert)      # measured 54 (cone: 106)


@pytest.mark.parametrize("robot,n", [(ROBOT_SOLO8, 8), (ROBOT_SOLO12, 12)])
def test_residual_threshold_early_exit_matches_oracle(robot, n):
    """SURVEY Appendix B K7 / PyBullet's solverResidualThreshold (the default since round 3, include/solorl.h): the PGS loop of a sub-step
    stops after the first sweep whose largest velocity-level change is within sqrt(1e-7) = 3.2e-4.  Same thrashing
    robot as test_contact_substeps_resynced: oracle and kernel math stop after the same sweep (errors stay at rounding),
    most solves stop long before 50 sweeps, and the truncated result is within a few thresholds of the full solve."""
    c = default_config(robot, TASK_WALK)
    assert c.solver_residual_threshold == 1e-7 and c.warmstart == 0.0     # defaults: PyBullet's exit, no multibody warm start
    cfull = default_config(robot, TASK_WALK); cfull.solver_residual_threshold = 0.0          # fixed 50 sweeps
    rng = np.random.default_rng(1)
    o, ofull = Oracle(c, 1), Oracle(cfull, 1)
    o.set_caps(8, 4); ofull.set_caps(8, 4)      # (algebra check, see test_contact_substeps_resynced)
    e64, its, trunc = [], [], []
    for k in range(400):
        so = o.get_state(0)
        if k % 4 == 0:
            tau = rng.uniform(-1.0, 1.0, size=n)
        for j in range(n):
            so.tau[j] = tau[j] if k % 4 == 0 else 0.0
        o.set_state(0, so); ofull.set_state(0, clone(so))
        h64 = clone(so)
        o.substep(0); ofull.substep(0)
        harness_py.substep(h64, c, False)
        a = o.get_state(0)
        assert a.contact_mask == h64.contact_mask and ofull.last_iterations(0) in (0, 50)
        if a.contact_mask:
            its.append(o.last_iterations(0))
            trunc.append(np.abs(state_vec(a, n) - state_vec(ofull.get_state(0), n)).max())
        e64.append(np.abs(state_vec(a, n) - state_vec(h64, n)).max())
    e64, its, trunc = np.array(e64), np.array(its), np.array(trunc)
    assert len(its) > 100 and np.median(its) <= 20 and its.max() <= 50 and (its < 50).mean() > 0.7     # measured medians 16 (Solo8) / 12 (Solo12)
    assert np.median(e64) < 1e-11 and np.percentile(e64, 95) < 1e-8           # same exit sweep in both implementations
    assert np.median(trunc) < 2e-3                                            # velocities within a few thresholds of the full solve
