"""solorl_amd -- MI355X-native vectorised Solo8/Solo12 rollout engine (HIP) behind the
reference's gym-style vec-env surface.  The HIP library is loaded lazily by ``vec_env``."""
from .config import (SoloConfig, default_config, config_from_dict, load_yaml,  # noqa: F401
                     ROBOT_SOLO8, ROBOT_SOLO12, TASK_STAND, TASK_WALK, TASK_POINTGOAL,
                     CONTROL_TORQUE, CONTROL_PD, PRECISION_F32, PRECISION_F64)

__version__ = "0.1.0"
