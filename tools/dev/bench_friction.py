"""ms/step at 4096 / 8192 / 65536 envs (Solo12 walk, random policy, steady state) for the [K] ledger's config fields:
round 3's values (friction pyramid, contact ERP 0.2, no collision margin), each round-4 change alone, and round 4's defaults (cone, 0.08,
1 mm).  The foot geometry (hull profile rings, round 4) is part of the model table and the same in every row.  Not a pytest file."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from solorl_amd.config import *
from solorl_amd.vec_env import SoloVecEnv
VARIANTS = (("round-3 values: pyramid, erp 0.20, margin 0", 0, 0.2, 0.0), ("cone alone", 1, 0.2, 0.0), ("contact erp 0.08 alone", 0, 0.08, 0.0),
            ("collision margin 1 mm alone", 0, 0.2, 0.001), ("round-4 defaults: cone, erp 0.08, margin 1 mm", 1, 0.08, 0.001))
for N in (4096, 8192, 65536):
    for name, cone, cerp, cm in VARIANTS:
        c = default_config(ROBOT_SOLO12, TASK_WALK); c.num_history_stack = 1
        c.friction_model, c.contact_erp, c.collision_margin = cone, cerp, cm
        env = SoloVecEnv(c, N, device="cuda:0", seed=1); env.reset()
        a = torch.rand(64, N, env.act_dim, device="cuda:0") * 2 - 1
        for t in range(450): env.step_inplace(a[t % 64])
        K = 200
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for t in range(K): env.step_inplace(a[t % 64])
        ts = []
        for r in range(5):
            torch.cuda.synchronize(); t0 = time.time(); g.replay(); torch.cuda.synchronize(); ts.append(time.time() - t0)
        dt = sorted(ts)[2]
        st = env.episode_stats()
        print("N %6d %-48s %.4f ms/step  %.2f M env-steps/s | mean episode length %.1f" % (N, name + ":", dt / K * 1e3, N * K / dt / 1e6, st["episode_length"]), flush=True)
        env.close()
