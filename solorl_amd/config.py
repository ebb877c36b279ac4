"""Task configuration: the reference's ``configs/*.yaml`` keys -> the C-ABI ``solorl_config``.

Mirrors how the reference consumes its config dict: ``SoloBaseEnv.__init__`` (baseEnv.py:8-16),
``is_episode_finished`` (baseEnv.py:164) and the YAML loader of training/train_ppo.py:47-50.  The
ctypes structures below are field-for-field ``include/solorl.h``.
"""
import ctypes as C
import os

import yaml

ROBOT_SOLO8, ROBOT_SOLO12 = 0, 1
TASK_STAND, TASK_WALK, TASK_POINTGOAL = 0, 1, 2
CONTROL_TORQUE, CONTROL_PD = 0, 1
PRECISION_F32, PRECISION_F64 = 0, 1
FRICTION_PYRAMID, FRICTION_CONE = 0, 1
ABI_VERSION = 5            # SOLORL_ABI_VERSION (include/solorl.h)

TASKS = {"stand": TASK_STAND, "walk": TASK_WALK, "pointgoal": TASK_POINTGOAL}
CONTROLS = {"torque": CONTROL_TORQUE, "pd": CONTROL_PD, "fpd": CONTROL_PD, "fixed_pd": CONTROL_PD}

MAX_DOF, MAX_PRIMS, MAX_OBS, MAX_HISTORY = 12, 24, 42, 4


class SoloConfig(C.Structure):
    """ctypes mirror of ``solorl_config`` (include/solorl.h)."""
    _fields_ = [
        ("robot", C.c_int32), ("task", C.c_int32), ("control", C.c_int32), ("frame_skip", C.c_int32),
        ("episode_length", C.c_int32), ("num_history_stack", C.c_int32), ("hold_torque", C.c_int32),
        ("use_urdf_inertia", C.c_int32), ("solver_iterations", C.c_int32), ("settle_min", C.c_int32),
        ("settle_max", C.c_int32), ("disable_termination", C.c_int32), ("precision", C.c_int32),
        ("use_treadmill", C.c_int32), ("friction_model", C.c_int32),
        ("kp", C.c_double), ("kd", C.c_double), ("max_torque", C.c_double), ("sim_dt", C.c_double),
        ("reward_dt", C.c_double), ("gravity", C.c_double), ("erp", C.c_double),
        ("linear_slop", C.c_double), ("warmstart", C.c_double), ("damping", C.c_double),
        ("max_velocity", C.c_double), ("joint_limit", C.c_double), ("goal_radius", C.c_double),
        ("treadmill_offset", C.c_double), ("treadmill_half_width", C.c_double), ("treadmill_friction", C.c_double),
        ("solver_residual_threshold", C.c_double), ("contact_erp", C.c_double), ("collision_margin", C.c_double),
    ]

    @property
    def n_joints(self):
        return 12 if self.robot == ROBOT_SOLO12 else 8

    @property
    def state_dim(self):
        return 14 + 2 * self.n_joints + (4 if self.task == TASK_POINTGOAL else 0)

    @property
    def obs_dim(self):
        return self.state_dim * (1 + self.num_history_stack)

    def copy(self):
        c = SoloConfig()
        C.memmove(C.byref(c), C.byref(self), C.sizeof(SoloConfig))
        return c


class EnvState(C.Structure):
    """ctypes mirror of ``solorl_env_state``."""
    _fields_ = [
        ("pos", C.c_double * 3), ("quat", C.c_double * 4), ("lin_vel", C.c_double * 3),
        ("ang_vel", C.c_double * 3), ("q", C.c_double * MAX_DOF), ("qd", C.c_double * MAX_DOF),
        ("tau", C.c_double * MAX_DOF), ("lambda_prev", C.c_double * MAX_PRIMS),
        ("hist", (C.c_double * MAX_OBS) * MAX_HISTORY), ("goal", C.c_double * 2), ("potential", C.c_double),
        ("progress", C.c_double), ("goals_reached", C.c_double), ("env_goals_reached", C.c_double),
        ("dr", C.c_double * 5), ("treadmill_y", C.c_double), ("timestep", C.c_int32), ("need_reset", C.c_int32),
        ("contact_mask", C.c_int32), ("rng_counter", C.c_int32),
    ]


class InfoSoA(C.Structure):
    """ctypes mirror of ``solorl_info_soa`` (device pointers)."""
    _fields_ = [(k, C.c_void_p) for k in (
        "timeout", "success", "nan_reset", "episode_length", "episode_reward", "goals_reached",
        "dr_stand", "dr_joint_pose", "dr_torque", "dr_balance", "dr_progress", "ep_stats", "applied_torque")]


EPSTAT_FIELDS = 10      # SOLORL_EPSTAT_FIELDS
EPSTAT_NAMES = ("episodes", "episode_reward", "episode_length", "success", "dr/stand_rew", "dr/joint_pose_rew",
                "dr/torque_rew", "dr/roll_pitch_balance_rew", "dr/progress_rew", "nan_resets")


def default_config(robot=ROBOT_SOLO12, task=TASK_WALK):
    """Reference defaults (baseEnv.py:8-16, solo.py:22,52-53,109-110,141) + PyBullet defaults
    (SURVEY.md Appendix B).  Kept identical to ``solorl_default_config`` in the C library."""
    c = SoloConfig()
    c.robot, c.task, c.control = robot, task, CONTROL_TORQUE
    c.frame_skip, c.episode_length, c.num_history_stack = 4, 400, 0
    c.hold_torque, c.use_urdf_inertia, c.solver_iterations = 0, 0, 50
    c.settle_min, c.settle_max, c.disable_termination, c.precision = 5, 11, 0, PRECISION_F32
    c.kp, c.kd, c.max_torque = 5.0, 0.2, 3.0
    c.sim_dt, c.reward_dt, c.gravity = 1.0 / 240.0, 1.0 / 60.0, 9.81
    c.erp, c.linear_slop, c.warmstart, c.damping = 0.2, 1e-5, 0.0, 0.04
    c.max_velocity, c.joint_limit, c.goal_radius = 100.0, 10.0, 2.0
    c.use_treadmill, c.treadmill_offset, c.treadmill_half_width, c.treadmill_friction = 0, 0.49, 0.5, 0.5
    c.solver_residual_threshold = 1e-7      # K7: PyBullet's solverResidualThreshold, see include/solorl.h
    # [K] ledger: Bullet's implicit friction cone and PyBullet's contact ERP (m_erp2 = 0.08); rounds 1-3: pyramid, 0.2
    c.friction_model, c.contact_erp = FRICTION_CONE, 0.08
    c.collision_margin = 0.001              # Bullet's margin around URDF hulls [K6]; rounds 1-3: the bare primitives (0)
    return c


def load_yaml(path):
    """YAML -> plain dict, as training/train_ppo.py:47-50."""
    with open(path) as f:
        return yaml.load(f, Loader=yaml.FullLoader)


def config_from_dict(d, **overrides):
    """Builds a ``SoloConfig`` from a reference-style config dict.

    Robot selection: ``solo12: True`` (configs/basic12.yaml:10) or the ``model_urdf`` basename
    (``solo12.urdf`` -> Solo12, ``solo.urdf`` -> Solo8); the absolute path of the reference's YAML
    (configs/basic.yaml:4 points into the author's home) is not opened -- the model tables are
    compiled in.  ``use_treadmill: True`` (configs/basic.yaml:10) enables the friction strip of
    simulation.py:45-77 (``solorl_config.treadmill_*``); as in the reference it only exists on flat
    ground (simulation.py:9).  ``flat_ground: False`` (heightfield terrains, simulation.py:79-154) is
    out of scope and rejected loudly.
    """
    d = dict(d)
    d.update(overrides)
    urdf = os.path.basename(str(d.get("model_urdf", "solo.urdf")))
    robot = ROBOT_SOLO12 if (d.get("solo12") or "12" in urdf) else ROBOT_SOLO8
    task = d.get("task", "stand")
    if task not in TASKS:
        raise ValueError("unknown task %r (stand / walk / pointgoal)" % (task,))
    control = d.get("control", "torque")
    if control in ("vpd", "variable_pd"):
        raise NotImplementedError("control 'vpd' is unreachable in the reference "
                                  "(solo.py:226 assert vs baseEnv.py:21-22) and is not provided")
    if control not in CONTROLS:
        raise NotImplementedError("control %r (reference solo.py:253-254)" % (control,))
    if not d.get("flat_ground", True):
        raise NotImplementedError("heightfield terrain (simulation.py:79-154) is out of scope (the reference itself fails to load it: "
                                  "simulation.py:28 calls Heightfield() without the required stepwidth)")
    c = default_config(robot, TASKS[task])
    c.control = CONTROLS[control]
    c.frame_skip = int(d.get("frame_skip", 4))
    c.episode_length = int(d["episode_length"])
    c.num_history_stack = int(d.get("num_history_stack", 0))
    c.use_treadmill = 1 if d.get("use_treadmill", False) else 0        # baseEnv.py:13, simulation.py:9
    if not 0 <= c.num_history_stack <= MAX_HISTORY:
        raise ValueError("num_history_stack must be 0..%d" % MAX_HISTORY)
    gains = d.get("gains", None)
    if c.control == CONTROL_PD:
        if gains is None:
            raise ValueError("control 'pd' needs gains: [Kp, Kd] (solo.py:240)")
        c.kp, c.kd = float(gains[0]), float(gains[1])
    for k in ("hold_torque", "use_urdf_inertia", "solver_iterations", "disable_termination", "settle_min",
              "settle_max", "precision", "warmstart", "erp", "damping", "reward_dt", "goal_radius", "treadmill_offset",
              "treadmill_half_width", "treadmill_friction", "solver_residual_threshold", "contact_erp", "friction_model", "collision_margin"):
        if k in d:
            v = d[k]
            if k == "friction_model" and isinstance(v, str):
                v = {"pyramid": FRICTION_PYRAMID, "cone": FRICTION_CONE}[v]
            setattr(c, k, type(getattr(c, k))(v))
    return c
