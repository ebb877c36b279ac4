"""K6 with a trained policy, on the CPU oracle (test infrastructure; nothing here touches the engine): success rate of a fixture policy of
tests/golden/policies/ (mean action / with its exploration noise; 256 envs x 450 steps) under
  * solver / friction variants on the policy's own contact model (is the gait brittle?),
  * Bullet-style hull manifolds for ONE link class, primitives elsewhere (which links carry the gap?),
  * two cheap stand-ins for the foot hull on the primitives (oracle option foot_points).
usage: k6_policy_ablation.py [walk12|walk8|stand8|pointgoal12] [checkpoint.pt]      (DESIGN.md section 3 K6 iii-iv, profiles/r04_notes.md)"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from solorl_amd.config import load_yaml, config_from_dict          # noqa: E402
from solorl_amd.ppo import Policy                                   # noqa: E402
from solorl_amd.vec_env import Box                                  # noqa: E402

NAME = sys.argv[1] if len(sys.argv) > 1 else "walk12"
CFG = {"walk12": ("basic12.yaml", "walk"), "walk8": ("basic.yaml", "walk"), "stand8": ("basic.yaml", "stand"), "pointgoal12": ("basic12.yaml", "pointgoal")}[NAME]
d = load_yaml(os.path.join(ROOT, "configs", CFG[0])); d["task"] = CFG[1]
c = config_from_dict(d)
pol = Policy((c.obs_dim,), Box(-np.ones(c.n_joints), np.ones(c.n_joints)), None, {"hidden_size": 64})
CKPT = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "tests", "golden", "policies", NAME + ".pt")      # optional: another checkpoint of the same shape
_sd = torch.load(CKPT, map_location="cpu", weights_only=False)
pol.load_state_dict(_sd["state_dict"] if "state_dict" in _sd else _sd); pol.eval()
std = pol.pi_dist.logstd.detach().exp().numpy().reshape(1, -1)
N, T = 256, 450
noise = np.random.default_rng(77).standard_normal((T, N, c.n_joints)).astype(np.float32)
NL = 17 if c.n_joints == 12 else 13
per = (NL - 1) // 4
cls = lambda k: sum(1 << (1 + per * leg + k) for leg in range(4))       # link class k of every leg (shoulder / upper / lower / foot)


def run(label, mask=None, persist=1, over=None, opts=None):
    res = []
    for stochastic in (False, True):
        if mask is None: os.environ.pop("ORACLE_MANIFOLD_LINKS", None)
        else: os.environ["ORACLE_MANIFOLD_LINKS"] = str(mask)
        from oracle.oracle_py import Oracle
        cc = c.copy()
        for k, v in (over or {}).items(): setattr(cc, k, v)
        o = Oracle(cc, N, seed=5, threads=min(8, os.cpu_count() or 1))
        if mask is not None: o.set_contact_model(1)
        o.set_option("manifold_persist", persist)
        for k, v in (opts or {}).items(): o.set_option(k, v)
        obs = o.reset(); succ = []
        for t in range(T):
            with torch.no_grad():
                a = pol.act(torch.from_numpy(obs.astype(np.float32)), deterministic=True)[1].numpy()
            if stochastic: a = a + std * noise[t]
            obs, r, dn, info = o.step(a.astype(np.float64))
            succ += list(info["success"][dn != 0])
        h = o.iteration_histogram(); tot = max(h.sum(), 1)
        res.append((np.mean(succ), (h * np.arange(h.size)).sum() / tot, 100.0 * h[min(cc.solver_iterations, h.size - 1):].sum() / tot))
    print("%-58s success %.3f / %.3f   sweeps %.1f, never converged %.1f %%" % (label, res[0][0], res[1][0], res[1][1], res[1][2]), flush=True)


ALL = (1 << NL) - 1
FEET = cls(per - 1)
print("== %s (%s --task %s%s): success with the mean action / with noise" % (NAME, CFG[0], CFG[1], ", " + os.path.relpath(CKPT, ROOT) if len(sys.argv) > 2 else ""))
run("primitives (the model it was trained on)")
run("  400 sweeps, no residual exit", over={"solver_iterations": 400, "solver_residual_threshold": 0.0})
run("  200 sweeps", over={"solver_iterations": 200})
run("  friction pyramid", over={"friction_model": 0})
run("  contact erp 0.2", over={"contact_erp": 0.2})
run("  both tread edges per foot (foot_points 2)", opts={"foot_points": 2})
run("  support vertex of a 53-gon tread (foot_points 3)", opts={"foot_points": 3})
run("  support vertex of the foot HULL, primitive path (foot_points 4)", opts={"foot_points": 4})
run("hull manifolds: all links", mask=ALL)
run("  feet only", mask=FEET)
run("  feet only, no persistent cache", mask=FEET, persist=0)
run("  lower legs only", mask=cls(per - 2))
run("  upper legs only", mask=cls(per - 3))
run("  base only", mask=1)
run("  all but the feet", mask=ALL & ~FEET)
run("  all links, 400 sweeps, no residual exit", mask=ALL, over={"solver_iterations": 400, "solver_residual_threshold": 0.0})
