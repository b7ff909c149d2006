"""RSL-RL side of the hot path: VecEnv wrapper, rollout storage + GAE, PPO, on-policy runner."""

from .actor_critic import ActorCritic  # noqa: F401
from .ppo import PPO  # noqa: F401
from .runner import OnPolicyRunner  # noqa: F401
from .storage import RolloutStorage  # noqa: F401
from .vecenv_wrapper import RslRlVecEnvWrapper  # noqa: F401
