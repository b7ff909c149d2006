"""``RslRlVecEnvWrapper`` with the reference's surface (isaaclab_rl/rsl_rl/vecenv_wrapper.py:14-209).

Differences that matter for speed, none for results: the action clamp (``:173-174``) is fused into
``imx_action_process``; ``dones`` is produced without touching the host.  The env writes its observation groups into
persistent buffers (fixed pointers: hipGraph capture); the reference's ObservationManager returns NEW tensors every step and
RSL-RL keeps references to them across ``env.step`` (``transition.observations = obs``), so this wrapper hands out copies.
The fused rollout of ``OnPolicyRunner`` talks to the env directly and never pays for them.
"""

from __future__ import annotations

import torch

from ..env import ManagerBasedRLEnv


def _own(obs):
    """Copies of the env's observation buffers (it rewrites them in place at the next step); groups may be dicts of terms."""
    return {k: _own(v) if isinstance(v, dict) else v.clone() for k, v in obs.items()}


class RslRlVecEnvWrapper:
    def __init__(self, env: ManagerBasedRLEnv, clip_actions: float | None = None):
        if not isinstance(env.unwrapped, ManagerBasedRLEnv):
            raise ValueError(
                f"The environment must be inherited from ManagerBasedRLEnv or DirectRLEnv. Environment type: {type(env)}")
        self.env = env
        self.clip_actions = clip_actions
        self.num_envs = self.unwrapped.num_envs
        self.device = self.unwrapped.device
        self.max_episode_length = self.unwrapped.max_episode_length
        self.num_actions = self.unwrapped.action_manager.total_action_dim
        self.num_obs = self.unwrapped.observation_manager.group_obs_dim["policy"][0]
        if "critic" in self.unwrapped.observation_manager.group_obs_dim:
            self.num_privileged_obs = self.unwrapped.observation_manager.group_obs_dim["critic"][0]
        else:
            self.num_privileged_obs = 0
        self.unwrapped.clip_actions = clip_actions
        # reset at the start since the RSL-RL runner does not call reset
        self.env.reset()

    def __str__(self):
        return f"<{type(self).__name__}{self.env}>"

    __repr__ = __str__

    @property
    def cfg(self):
        return self.unwrapped.cfg

    @property
    def render_mode(self):
        return self.env.render_mode

    @classmethod
    def class_name(cls) -> str:
        return cls.__name__

    @property
    def unwrapped(self) -> ManagerBasedRLEnv:
        return self.env.unwrapped

    def get_observations(self) -> tuple[torch.Tensor, dict]:
        obs_dict = _own(self.unwrapped.observation_manager.compute())
        return obs_dict["policy"], {"observations": obs_dict}

    @property
    def episode_length_buf(self) -> torch.Tensor:
        return self.unwrapped.episode_length_buf

    @episode_length_buf.setter
    def episode_length_buf(self, value: torch.Tensor):
        self.unwrapped.episode_length_buf = value

    def seed(self, seed: int = -1) -> int:
        return self.unwrapped.seed(seed)

    def reset(self) -> tuple[torch.Tensor, dict]:
        obs_dict, _ = self.env.reset()
        obs_dict = _own(obs_dict)
        return obs_dict["policy"], {"observations": obs_dict}

    def step(self, actions: torch.Tensor):
        # the clamp of vecenv_wrapper.py:173-174 happens inside imx_action_process (env.clip_actions)
        obs_dict, rew, terminated, truncated, extras = self.env.step(actions)
        dones = (terminated | truncated).to(dtype=torch.long)
        obs_dict = _own(obs_dict)
        obs = obs_dict["policy"]
        extras["observations"] = obs_dict
        if not self.unwrapped.is_finite_horizon:
            extras["time_outs"] = truncated
        return obs, rew, dones, extras

    def close(self):
        return self.env.close()
