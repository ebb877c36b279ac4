"""The C-ABI library: loads, exports every symbol include/solorl.h declares, agrees with the Python
mirror of the config, and fails loudly (no CPU fallback) without a GPU.  No compute calls here."""
import ctypes as C
import os
import re

import pytest

from solorl_amd import _native, build
from solorl_amd.config import (SoloConfig, EnvState, default_config, config_from_dict, load_yaml, ROBOT_SOLO8,
                               ROBOT_SOLO12, TASK_WALK, TASK_POINTGOAL, TASK_STAND, CONTROL_PD)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    build.build()
    return _native.lib()


def test_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "solorl.h")).read()
    declared = set(re.findall(r"\b(solorl_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(_native.SYMBOLS)
    for s in declared:
        assert hasattr(lib, s)
    assert b"gfx950" in lib.solorl_version()


def test_default_config_matches_python_mirror(lib):
    for robot in (ROBOT_SOLO8, ROBOT_SOLO12):
        for task in (TASK_STAND, TASK_WALK, TASK_POINTGOAL):
            c = SoloConfig()
            assert lib.solorl_default_config(C.byref(c), robot, task) == 0
            assert bytes(c) == bytes(default_config(robot, task))
    assert C.sizeof(SoloConfig) == 16 * 4 + 19 * 8        # 15 int32 (+ 4 bytes of padding before the doubles) + 19 doubles
    assert C.sizeof(EnvState) == 8 * (3 + 4 + 3 + 3 + 12 * 3 + 24 + 4 * 42 + 2 + 4 + 5 + 1) + 4 * 4       # hist[SOLORL_STATE_MAX_HISTORY = 4][42]


def test_error_convention(lib):
    import torch
    h = C.c_void_p()
    c = default_config(ROBOT_SOLO12, TASK_WALK)
    bad = c.copy(); bad.frame_skip = 0
    assert lib.solorl_create(C.byref(bad), 4, 0, 1, 0, C.byref(h)) == -1
    assert b"frame_skip" in lib.solorl_last_error()
    bad = c.copy(); bad.solver_residual_threshold = -1.0
    assert lib.solorl_create(C.byref(bad), 4, 0, 1, 0, C.byref(h)) == -1
    assert lib.solorl_dims(None, None, None, None) == -1
    if not torch.cuda.is_available():
        # product path must fail loudly without a GPU: no CPU fallback, no oracle routing
        assert lib.solorl_create(C.byref(c), 4, 0, 1, 0, C.byref(h)) == -2
        assert b"no CPU fallback" in lib.solorl_last_error()
        from solorl_amd.vec_env import SoloVecEnv
        with pytest.raises(_native.SoloRLError):
            SoloVecEnv(c, 4)


def test_policy_kernel_entry_points_reject_bad_arguments(lib):
    """The PPO-side entry points follow the same convention: negative code + message, never a crash -- null tables, a hidden size
    or an observation width the kernels are not built for, and (without a GPU) no silent fallback."""
    import torch
    P = _native.PolicyParams()
    assert lib.solorl_policy_act(None, None, None, 4, None, None, None, 0, None) == -1
    assert b"null policy parameters" in lib.solorl_last_error()
    P.obs_dim, P.act_dim, P.hidden = 76, 12, 32
    assert lib.solorl_policy_act(C.byref(P), None, None, 4, None, None, None, 0, None) == -1
    assert b"hidden size 64" in lib.solorl_last_error()
    P.hidden = 64
    assert lib.solorl_policy_act(C.byref(P), None, None, 4, None, None, None, 0, None) == -1          # null parameter pointers
    assert b"null policy parameter pointer" in lib.solorl_last_error()
    assert lib.solorl_ppo_grad_stage1(C.byref(P), None, None, 0, None) == -1
    assert lib.solorl_ppo_grad_stage2(C.byref(P), None, 64, None, 0, None) == -1
    assert lib.solorl_ppo_clip_adam(C.byref(P), None, None, 0, None) == -1
    assert lib.solorl_ppo_grad_count(76, 12) == 2 * 64 * 77 + 2 * 64 * 65 + 65 + 12 * 65
    # stage 2's working memory (ABI 5): one block of U x (K + 1) floats per layer and chunk of rows, fewer than 1024 chunks in all (one
    # wavefront per SIMD), chunks of whole 32-row tiles; a pure host function
    n = lib.solorl_ppo_scratch_count(76, 12, 32768)
    per_layer = [64 * 77, 64 * 65, 65, 64 * 77, 64 * 65, 12 * 65]
    assert n % 1 == 0 and sum(per_layer) * 64 < n < sum(per_layer) * 1024 / 6 * 1.5
    assert lib.solorl_ppo_scratch_count(76, 12, 0) == 0 and lib.solorl_ppo_scratch_count(76, 12, 64) == 2 * sum(per_layer)
    assert lib.solorl_ppo_scratch_count(60, 8, 32768) < n
    if not torch.cuda.is_available():
        for k, _t in P._fields_[4:]:
            setattr(P, k, 4096)                       # non-null (never dereferenced on the host)
        assert lib.solorl_policy_act(C.byref(P), None, None, 4, None, None, None, 0, None) == -2
        assert b"no CPU fallback" in lib.solorl_last_error()


def test_product_never_imports_oracle():
    for dp, _, fs in os.walk(os.path.join(ROOT, "solorl_amd")):
        for f in fs:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                txt = open(os.path.join(dp, f)).read()
                for pat in (r"^\s*(from|import)\s+oracle", r"#include\s*[\"<][^\">]*oracle", r"liboracle", r"oracle_py",
                            r"(dlopen|CDLL)\([^)]*oracle"):
                    assert not re.search(pat, txt, re.M), (f, pat)


def test_reference_yaml_configs_load():
    for name, robot, task, obs in [("basic.yaml", ROBOT_SOLO8, TASK_WALK, 60), ("basic12.yaml", ROBOT_SOLO12, TASK_POINTGOAL, 84)]:
        d = load_yaml(os.path.join(ROOT, "configs", name))
        c = config_from_dict(d)
        assert (c.robot, c.task, c.frame_skip, c.episode_length, c.num_history_stack) == (robot, task, 4, 400, 1)
        assert c.obs_dim == obs
    c = config_from_dict(load_yaml(os.path.join(ROOT, "configs", "basic_pd.yaml")))          # loads as shipped
    assert c.control == CONTROL_PD and (c.kp, c.kd) == (5.0, 0.2) and c.episode_length == 400 and c.robot == ROBOT_SOLO8
    c = config_from_dict(load_yaml(os.path.join(ROOT, "configs", "basic12.yaml")), task="walk")
    assert c.task == TASK_WALK and c.obs_dim == 76
    with pytest.raises(NotImplementedError):
        config_from_dict(dict(episode_length=10, control="vpd"))
    with pytest.raises(ValueError):
        config_from_dict(dict(episode_length=10, task="fly"))


def test_step_kernels_have_no_static_lds(lib):
    """The phase functions address the dynamic LDS from a constant base (dynamics.hpp, SOLO_LDS_BASE = 0): only true while no step
    kernel has static LDS (the compiler once promoted a private array of the fp64 team kernel into 3 KB of it).  solorl_step checks
    the same against the loaded code object (hipFuncGetAttributes)."""
    from solorl_amd import devcode
    sizes = {k: v for k, v in devcode.kernel_static_lds(build.LIB).items() if "step_kernel" in k}
    assert len(sizes) == 8, sorted(sizes)       # {lane, team} x {fp32, fp64} x {Solo8, Solo12}
    assert all(v == 0 for v in sizes.values()), sizes


def test_team_mode_phases_do_not_spill_or_use_flat(lib):
    """Static check of the built code object: the fp32 Solo12/Solo8 team-mode phases keep their state in VGPRs
    and LDS (no scratch traffic, no FLAT access to the context); the sweeps of up to 6 contacts -- 99.6 % of the
    launches of the headline workload -- do not spill either."""
    from solorl_amd import devcode
    st = devcode.function_stats(build.LIB)
    hot = [n for n in st if re.search(r"phase_(front|leg_rt|base_lead|finish|integrate)_?(team)?I?f", n) and "IfLi" in n and "CtxPriv" not in n
           and "RowLdsIfLi64" not in n]
    assert len(hot) >= 10, sorted(st)[:5]          # 5 phases x 2 robots
    for n in hot:
        assert st[n]["scratch"] <= 2, (n, st[n])   # (leaf phases: at most one register saved around the body)
        assert st[n]["flat"] <= 13, (n, st[n])     # phase_leg_rt reads PhysParams fields (9 + the treadmill friction per leg primitive) through its reference argument
        assert st[n]["global"] == 0, (n, st[n])
    # the 17 specialised sweeps: whatever their register pressure (the heaviest save callee-saved registers to scratch
    # around the body now that the kernel is held to 256 registers for two wavefronts per SIMD), the sweep LOOP itself
    # is register-only -- no scratch, no LDS read, no global or FLAT access (its only LDS writes are the cold block that
    # publishes a team's result when its residual falls below the K7 threshold) -- at 26 VALU instructions per slot
    sweeps = devcode.loop_stats(build.LIB, "pgs_team_variantIfNS")
    # 17 slot sets x {pipelined, K7 residual exit} + 16 plain ones (SOLORL_PGS_PIPE=0) with the friction pyramid, and the 16 slot sets
    # that have contacts x the same three with the friction cone (3 more VALU instructions per friction slot)
    assert len(sweeps) == 50 + 48
    for n, s in sweeps.items():
        m = re.search(r"Li(\d)ELi(\d)ELi(\d)ELb(\d)ELb(\d)ELb(\d)EEE", n)
        lim, nn, nf, early, pipe, cone = (int(x) for x in m.groups())
        nslots = lim + nn + nf + (3 * nf * cone) / 27.0
        if nn + nf == 0:
            continue                       # (the limit-only sweep is 1 slot)
        assert s["loop"][0] is not None, n
        assert s["scratch_in_loop"] == 0 and s["lds_reads_in_loop"] == 0 and s["vmem_in_loop"] == 0, (n, s)
        if early:
            assert s["valu_in_loop"] <= 27 * nslots + 16, (n, s)
        else:
            assert s["lds_in_loop"] == 0 and s["valu_in_loop"] <= 27 * nslots, (n, s)
        if nf <= 4:
            assert s["scratch"] <= (16 if early else 6), (n, s)       # callee-saved VGPR saves around the body only
