"""PPO side of the rollout engine (stays in PyTorch-ROCm, per the north star): actor-critic MLP, rollout
storage with GAE, clipped-surrogate update with a flat-bucket RCCL gradient all-reduce, and the
training loop.  Same public surface as the reference's agents/ppo/ package."""
from .policy import Policy  # noqa: F401
from .storage import RolloutStorage, OPBuffer  # noqa: F401
from .ppo import PPO  # noqa: F401
