"""``ActorCritic`` (upstream ``rsl_rl/modules/actor_critic.py`` @ v2.3.1; cfg: reference
isaaclab_rl/rsl_rl/rl_cfg.py:62-82).  Keeps ``.actor/.critic/.std/.is_recurrent`` so the reference's exporter
(isaaclab_rl/rsl_rl/exporter.py:11-45) and play script keep working.  GEMMs stay in torch (hipBLASLt)."""

from __future__ import annotations

import torch
import torch.nn as nn
from torch.distributions import Normal

_ACT = {"elu": nn.ELU, "selu": nn.SELU, "relu": nn.ReLU, "lrelu": nn.LeakyReLU, "tanh": nn.Tanh, "sigmoid": nn.Sigmoid,
        "crelu": nn.ReLU, "identity": nn.Identity}


def _mlp(inp, hidden, out, act):
    layers, d = [], inp
    for h in hidden:
        layers += [nn.Linear(d, h), _ACT[act]()]
        d = h
    layers.append(nn.Linear(d, out))
    return nn.Sequential(*layers)


class ActorCritic(nn.Module):
    is_recurrent = False

    def __init__(self, num_actor_obs, num_critic_obs, num_actions, actor_hidden_dims=(256, 256, 256),
                 critic_hidden_dims=(256, 256, 256), activation="elu", init_noise_std=1.0, noise_std_type="scalar", **kwargs):
        if kwargs:  # upstream prints and ignores them; a recurrent / cascade cfg silently trained as a plain MLP is worse than an error
            raise NotImplementedError(
                "ActorCritic got arguments it does not implement: " + ", ".join(sorted(kwargs)) + " -- recurrent (rnn_*), cascade "
                "(lidar_input_dim, mlp*_...) and other policy variants are outside the hot-path scope (SURVEY.md section 8)")
        super().__init__()
        self.actor = _mlp(num_actor_obs, list(actor_hidden_dims), num_actions, activation)
        self.critic = _mlp(num_critic_obs, list(critic_hidden_dims), 1, activation)
        self.noise_std_type = noise_std_type
        if noise_std_type == "scalar":
            self.std = nn.Parameter(init_noise_std * torch.ones(num_actions))
        elif noise_std_type == "log":
            self.log_std = nn.Parameter(torch.log(init_noise_std * torch.ones(num_actions)))
        else:
            raise ValueError(f"Unknown standard deviation type: {noise_std_type}. Should be 'scalar' or 'log'")
        self.distribution = None
        Normal.set_default_validate_args(False)

    def reset(self, dones=None):
        pass

    @property
    def action_mean(self):
        return self.distribution.mean

    @property
    def action_std(self):
        return self.distribution.stddev

    @property
    def entropy(self):
        return self.distribution.entropy().sum(dim=-1)

    def _std(self, mean):
        std = self.std if self.noise_std_type == "scalar" else torch.exp(self.log_std)
        return std.expand_as(mean)

    def update_distribution(self, observations):
        mean = self.actor(observations)
        self.distribution = Normal(mean, self._std(mean))

    def act(self, observations, **kwargs):
        self.update_distribution(observations)
        # == distribution.sample(); torch.normal(mean, std_tensor) validates std on the host (a stream sync that
        # cannot be captured in a hipGraph), randn_like does not
        return self.distribution.mean + self.distribution.stddev * torch.randn_like(self.distribution.mean)

    def get_actions_log_prob(self, actions):
        return self.distribution.log_prob(actions).sum(dim=-1)

    def act_inference(self, observations):
        return self.actor(observations)

    def evaluate(self, critic_observations, **kwargs):
        return self.critic(critic_observations)
