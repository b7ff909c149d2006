"""RSL-RL side of the hot path: VecEnv wrapper, rollout storage + GAE, PPO, on-policy runner."""

from .vecenv_wrapper import RslRlVecEnvWrapper  # noqa: F401
