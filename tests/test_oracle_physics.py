"""Known-answer tests that pin the fp64 oracle's physics: the reference
holds no golden vectors for this path ("parity unpinned"), so the restatement is pinned by
closed-form physics instead."""
import numpy as np
import pytest

from solorl_amd.config import (default_config, ROBOT_SOLO8, ROBOT_SOLO12, TASK_WALK, TASK_STAND)
from oracle.oracle_py import Oracle, philox, euler_from_quat

ROBOTS = [(ROBOT_SOLO8, 8, 2.17785), (ROBOT_SOLO12, 12, 2.5)]


def airborne(robot, n, seed=0, damping=0.0, z=50.0):
    c = default_config(robot, TASK_WALK)
    c.damping = damping
    o = Oracle(c, 1)
    s = o.get_state(0)
    rng = np.random.default_rng(seed)
    s.pos[2] = z
    for j in range(n):
        s.q[j] = rng.uniform(-1, 1); s.qd[j] = rng.uniform(-3, 3)
    s.ang_vel[:] = [0.3, -0.2, 0.5]; s.lin_vel[:] = [0.1, 0.2, 0.0]
    q = rng.normal(size=4); q /= np.linalg.norm(q); s.quat[:] = list(q)
    o.set_state(0, s)
    return o, c


def test_philox_known_answer():
    # Random123 kat_vectors: philox4x32-10, counter 0 / key 0 and the all-ones vector
    assert philox(0, 0, 0, 0, 0, 0) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert philox(0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff) == \
        [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]


def test_euler_zyx():
    r, p, y = 0.3, -0.4, 1.1
    cr, sr, cp, sp, cy, sy = np.cos(r / 2), np.sin(r / 2), np.cos(p / 2), np.sin(p / 2), np.cos(y / 2), np.sin(y / 2)
    q = [sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy, cr * cp * cy + sr * sp * sy]
    assert np.allclose(euler_from_quat(q), [r, p, y], atol=1e-12)
    assert np.allclose(euler_from_quat([0, 0, 0, 1]), [0, 0, 0])
    # gimbal branch (|sarg| >= 0.99999): pitch = +-pi/2, roll = 0
    g = euler_from_quat([0, np.sin(np.pi / 4), 0, np.cos(np.pi / 4)])
    assert g[0] == 0 and abs(g[1] - np.pi / 2) < 1e-12


@pytest.mark.parametrize("robot,n,mass", ROBOTS)
def test_mass_matrix(robot, n, mass):
    o, _ = airborne(robot, n)
    M, h = o.mass_matrix(0)
    assert np.abs(M - M.T).max() < 1e-15
    assert np.linalg.eigvalsh(M).min() > 0
    assert np.allclose(M[3:6, 3:6], mass * np.eye(3), atol=1e-5)       # total mass (Appendix A)
    # forward dynamics == dense solve
    s = o.get_state(0)
    for j in range(n):
        s.tau[j] = 0.1 * (j + 1)
    o.set_state(0, s)
    rhs = -h.copy(); rhs[6:] += np.array(s.tau)[:n]
    assert np.allclose(o.forward_dynamics(0), np.linalg.solve(M, rhs), rtol=1e-10, atol=1e-10)


@pytest.mark.parametrize("robot,n,mass", ROBOTS)
def test_free_fall_exact(robot, n, mass):
    """No rotation, no joint motion, no damping: semi-implicit Euler z_k = z0 - g dt^2 k(k+1)/2."""
    c = default_config(robot, TASK_WALK); c.damping = 0.0
    o = Oracle(c, 1)
    s = o.get_state(0); s.pos[2] = 10.0; o.set_state(0, s)
    dt, g = c.sim_dt, c.gravity
    for k in range(1, 101):
        o.substep(0)
    s = o.get_state(0)
    assert abs(s.pos[2] - (10.0 - g * dt * dt * 100 * 101 / 2)) < 1e-9
    assert abs(s.lin_vel[2] + g * dt * 100) < 1e-9
    assert np.abs(np.array(s.q)[:n]).max() < 1e-9            # straight legs stay straight in free fall


@pytest.mark.parametrize("robot,n,mass", ROBOTS)
def test_momentum_balance_second_order(robot, n, mass):
    """Free flight with joint torques: linear momentum changes by m*g*dt only; the explicit-Euler
    error must shrink ~dt^2."""
    errs = []
    for dt in (1 / 240, 1 / 2400):
        c = default_config(robot, TASK_WALK); c.damping = 0.0; c.sim_dt = dt; c.max_velocity = 1e9
        o = Oracle(c, 1)
        s = o.get_state(0); s.pos[2] = 50.0
        rng = np.random.default_rng(0)
        for j in range(n):
            s.q[j] = rng.uniform(-1, 1); s.qd[j] = rng.uniform(-3, 3); s.tau[j] = rng.uniform(-1, 1)
        s.ang_vel[:] = [0.3, -0.2, 0.5]; s.lin_vel[:] = [0.1, 0.2, 0.0]
        o.set_state(0, s)
        e0 = o.energy_momentum(0); o.substep(0); e1 = o.energy_momentum(0)
        M, _ = o.mass_matrix(0)
        errs.append(np.abs(e1["p"] - e0["p"] - np.array([0, 0, -M[3, 3] * c.gravity * dt])).max())
    assert errs[0] < 1e-2 and errs[1] < errs[0] / 30


@pytest.mark.parametrize("robot,n,mass", ROBOTS)
def test_energy_drift_zero_damping(robot, n, mass):
    o, c = airborne(robot, n)
    e0 = o.energy_momentum(0)
    for _ in range(240):
        o.substep(0)
    e1 = o.energy_momentum(0)
    # dominant term: semi-implicit free fall loses m g^2 dt^2 k / 2 exactly; the rest is O(dt) joint motion
    M, _ = o.mass_matrix(0)
    expect = -0.5 * M[3, 3] * c.gravity ** 2 * c.sim_dt ** 2 * 240
    drift = (e1["T"] + e1["V"]) - (e0["T"] + e0["V"])
    assert abs(drift - expect) < 0.05 * e0["T"] + 0.02


@pytest.mark.parametrize("robot,n,mass", ROBOTS)
def test_damping_dissipates(robot, n, mass):
    o, c = airborne(robot, n, damping=0.04)
    o2, _ = airborne(robot, n, damping=0.0)
    for _ in range(120):
        o.substep(0); o2.substep(0)
    s, s2 = o.get_state(0), o2.get_state(0)
    assert np.linalg.norm(s.ang_vel) < np.linalg.norm(s2.ang_vel)
    assert abs(s.lin_vel[2]) < abs(s2.lin_vel[2])


@pytest.mark.parametrize("robot,n,mass", ROBOTS)
def test_static_stand_supports_weight(robot, n, mass):
    """After the reset settle the four feet carry m*g: sum(lambda_n)/dt = m g."""
    c = default_config(robot, TASK_STAND); c.settle_min = c.settle_max = 8
    o = Oracle(c, 1); o.reset()
    s = o.get_state(0)
    M, _ = o.mass_matrix(0)
    force = sum(s.lambda_prev) / c.sim_dt
    assert abs(force - M[3, 3] * c.gravity) < 0.02 * M[3, 3] * c.gravity
    assert bin(s.contact_mask).count("1") == 4 and all((s.contact_mask >> (13 + 2 * f)) & 1 for f in range(4))
    # geometry check of Appendix A: straight legs, feet 14 mm above ground at z=0.35 -> lands at ~0.336
    assert abs(s.pos[2] - 0.336) < 2e-3


def test_contact_is_unilateral_and_friction_bounded():
    c = default_config(ROBOT_SOLO12, TASK_STAND); c.settle_min = c.settle_max = 8
    o = Oracle(c, 1); o.reset()
    s = o.get_state(0); s.lin_vel[0] = 0.5; o.set_state(0, s)       # shove sideways
    for _ in range(20):
        o.substep(0)
        lam = o.last_lambda(0)
        assert (lam >= 0).all()
    s = o.get_state(0)
    assert abs(s.lin_vel[0]) < 0.5                                    # friction decelerates the slide


def test_joint_limit_stops_joint():
    """K5, btMultiBodyJointLimitConstraint: no row while the joint is inside its range (`penetration > 0` is skipped), so a joint
    approaching the limit at 30 rad/s passes it by 30/240 - 0.1 = 0.025 rad; from then on the row asks for erp * violation / dt
    back towards the range: the joint re-enters it at that speed (1.2 rad/s here), the row disappears and nothing holds it at the limit."""
    c = default_config(ROBOT_SOLO12, TASK_WALK)
    o = Oracle(c, 1)
    s = o.get_state(0); s.pos[2] = 5.0; s.q[2] = 9.9; s.qd[2] = 30.0; o.set_state(0, s)
    o.substep(0)
    s = o.get_state(0)
    assert o.last_counts(0)[2] == 0 and abs(s.q[2] - 10.025) < 2e-3 and s.qd[2] > 29.0          # free flight (damping only)
    viol = s.q[2] - 10.0
    o.substep(0)
    s = o.get_state(0)
    assert o.last_counts(0)[2:] == (1, 1)
    assert abs(s.qd[2] + c.erp * viol / c.sim_dt) < 1e-6                                          # -erp * violation / dt
    back = c.erp * viol / c.sim_dt
    for _ in range(50):
        o.substep(0)
    s = o.get_state(0)
    assert o.last_counts(0)[2] == 0 and 10.0 - back * 51 / 240 < s.q[2] < 10.0 and -back - 1e-6 < s.qd[2] < -0.8 * back     # drifting inwards


def test_velocity_clamp():
    c = default_config(ROBOT_SOLO12, TASK_WALK)
    o = Oracle(c, 1)
    s = o.get_state(0); s.pos[2] = 5.0
    for j in range(12):
        s.tau[j] = 3.0
    o.set_state(0, s)
    for _ in range(4):
        s = o.get_state(0)
        for j in range(12):
            s.tau[j] = 3.0
        o.set_state(0, s); o.substep(0)
    assert np.abs(np.array(o.get_state(0).qd)).max() <= 100.0 + 1e-9


def test_most_violated_joint_limits_are_the_ones_solved():
    """K5 solves at most two joint-limit rows: when more joints sit at their limits, the two with the smallest margin (the most
    violated) get them -- not the first two in joint order, which let a third joint run radians past its limit unopposed and
    be thrown back at erp * violation / dt later (robots launched metres into the air under a random policy, round 2)."""
    c = default_config(ROBOT_SOLO12, TASK_WALK)
    o = Oracle(c, 1)
    s = o.get_state(0); s.pos[2] = 5.0
    s.q[0], s.q[1], s.q[7], s.q[10] = 9.8, 9.7, 10.4, -10.2            # margins 0.2, 0.3, -0.4, -0.2: joints 7 and 10 are violated
    for j in (0, 1, 7, 10):
        s.qd[j] = 0.0
    o.set_state(0, s)
    o.substep(0)
    s1 = o.get_state(0)
    # the violated joints are pushed back at erp * violation / dt = 0.2 * 0.4 * 240 and 0.2 * 0.2 * 240 rad/s ...
    assert abs(s1.qd[7] + 0.2 * 0.4 * 240) < 0.5 and abs(s1.qd[10] - 0.2 * 0.2 * 240) < 0.5
    # ... and the two that are merely near their limits (and not moving towards them) are left alone
    assert abs(s1.qd[0]) < 2.0 and abs(s1.qd[1]) < 2.0
    # nothing builds up over time: a random-torque run from this state keeps every joint within a fraction of a radian of its range
    rng = np.random.default_rng(0)
    for _ in range(200):
        s2 = o.get_state(0)
        for j in range(12):
            s2.tau[j] = float(rng.uniform(-3, 3))
        s2.pos[2] = 5.0; s2.lin_vel[2] = 0.0
        o.set_state(0, s2); o.substep(0)
        assert max(abs(v) for v in o.get_state(0).q[:12]) < 10.6
