"""What the HIP engine's slot counts cost (Next #1a).  The oracle has no caps of its own -- like Bullet it gives every
joint at its limit a row and every contact point three -- while the engine's sweep holds at most 8 contact points and 4 joint-limit
rows per robot (limit rows 3 and 4 in place of contact points, solorl_amd/csrc/dynamics.hpp MAX_LIMITS).  `oracle_set_caps(8, 4)`
imposes exactly that on a second oracle stepped from the same states; this test counts how often the caps bind under random
policies and by how much they change the next joint angles.  (Round 2's caps, 8 and 2 with a 0.5 rad speculative limit window,
bound in 1.2-2.5 % of env-steps and moved joint angles by up to 0.75 rad: measured with this script before the change.)"""
import numpy as np
import pytest

from oracle.oracle_py import Oracle
from solorl_amd.config import default_config, ROBOT_SOLO8, ROBOT_SOLO12, TASK_WALK
from tests.util import clone


@pytest.mark.parametrize("robot,n,gauss", [(ROBOT_SOLO12, 12, False), (ROBOT_SOLO12, 12, True), (ROBOT_SOLO8, 8, True)])
def test_engine_slot_caps_rarely_bind(robot, n, gauss):
    N, T = 256, 200
    c = default_config(robot, TASK_WALK); c.num_history_stack = 1
    o = Oracle(c, N, seed=3, threads=8); o.reset()
    oc = Oracle(c, N, seed=3, threads=8); oc.reset(); oc.set_caps(8, 4)
    rng = np.random.default_rng(3)
    npts, nlim, dq, bound = [], [], [], 0
    for t in range(T):
        a = rng.normal(0, 1, (N, n)) if gauss else rng.uniform(-1, 1, (N, n))
        for i in range(N):
            oc.set_state(i, clone(o.get_state(i)))
        o.step(a); oc.step(a)
        for i in range(N):
            found, _, cand, _ = o.last_counts(i)
            _, solved_c, _, sel_c = oc.last_counts(i)
            npts.append(found); nlim.append(cand)
            so, sc = o.get_state(i), oc.get_state(i)
            if so.timestep == sc.timestep and so.timestep > 0:
                dq.append(np.abs(np.array(so.q)[:n] - np.array(sc.q)[:n]).max())
    npts, nlim, dq = np.array(npts), np.array(nlim), np.array(dq)
    f_c, f_l, f_d = (npts > 8).mean(), (nlim > 4).mean(), (dq > 1e-9).mean()
    print("robot %d %s policy, %d env-steps: contact points mean %.2f max %d, > 8 in %.4f %%; joints beyond a limit max %d, > 2 in %.3f %%, "
          "> 4 in %.4f %%; next joint angles differ capped vs uncapped in %.4f %% of env-steps (max %.2e rad)" % (
              robot, "gaussian" if gauss else "uniform", len(npts), npts.mean(), npts.max(), 100 * f_c, nlim.max(), 100 * (nlim > 2).mean(),
              100 * f_l, 100 * f_d, dq.max()))
    assert f_c < 5e-4 and f_l < 1e-4 and f_d < 1e-3
    assert (nlim > 0).mean() > 0.01 and (npts >= 4).mean() > 0.02          # the sample does visit limits and multi-contact states
