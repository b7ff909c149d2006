"""TEST INFRASTRUCTURE -- CPU restatement of the rsl-rl-lib v2.3.1 arithmetic on the hot path.  PARITY UNPINNED.

``rsl-rl-lib==2.3.1`` is a third-party dependency of the reference (pin: source/isaaclab_rl/setup.py:47,
scripts/reinforcement_learning/rsl_rl/train.py:56) that is absent from /root/reference (its two submodule directories
are empty) and not installable here (no network).  The reference holds no golden vectors for it
(source/isaaclab_rl/test/test_rsl_rl_wrapper.py:61-145 checks shapes / NaNs / the ``time_outs`` key only).  What is
restated below is the published upstream algorithm at tag v2.3.1:

  * ``rsl_rl/storage/rollout_storage.py::RolloutStorage.compute_returns``  (GAE(lambda) backward scan, advantage
    normalisation over all T*N with unbiased std + 1e-8)
  * ``rsl_rl/algorithms/ppo.py::PPO.process_env_step`` (time-out bootstrap) and ``PPO.update`` (clipped surrogate,
    clipped value loss, entropy bonus, adaptive-LR KL estimate)
  * ``rsl_rl/modules/actor_critic.py::ActorCritic`` (MLPs + scalar std, Normal distribution)

Until a real rsl_rl can be consulted, every GAE / PPO number checked against this file is "parity unpinned".
"""

from __future__ import annotations

import torch


def compute_returns(rewards, values, dones, last_values, gamma: float, lam: float, normalize_advantage: bool = True):
    """rewards/values/dones: (T,N,1); last_values: (N,1).  Returns (returns, advantages), both (T,N,1)."""
    T = rewards.shape[0]
    returns = torch.zeros_like(rewards)
    advantage = 0
    for step in reversed(range(T)):
        next_values = last_values if step == T - 1 else values[step + 1]
        next_is_not_terminal = 1.0 - dones[step].float()
        delta = rewards[step] + next_is_not_terminal * gamma * next_values - values[step]
        advantage = delta + next_is_not_terminal * gamma * lam * advantage
        returns[step] = advantage + values[step]
    advantages = returns - values
    if normalize_advantage:
        advantages = (advantages - advantages.mean()) / (advantages.std() + 1e-8)
    return returns, advantages


def bootstrap_time_outs(rewards, values, time_outs, gamma: float):
    """PPO.process_env_step: rewards += gamma * squeeze(values * time_outs.unsqueeze(1), 1)."""
    return rewards + gamma * torch.squeeze(values * time_outs.unsqueeze(1).to(values.dtype), 1)


def ppo_losses(mu, sigma, actions, old_logp, old_mu, old_sigma, advantages, returns, values, old_values,
               clip_param: float, use_clipped_value_loss: bool = True):
    """Elementwise part of PPO.update for one minibatch.  Shapes: mu/sigma/actions/old_mu/old_sigma (M,A); the
    rest (M,1).  Returns (surrogate_loss, value_loss, entropy_mean, kl_mean)."""
    dist = torch.distributions.Normal(mu, sigma)
    logp = dist.log_prob(actions).sum(dim=-1)
    entropy = dist.entropy().sum(dim=-1)
    kl = torch.sum(
        torch.log(sigma / old_sigma + 1.0e-5) + (torch.square(old_sigma) + torch.square(old_mu - mu)) / (2.0 * torch.square(sigma)) - 0.5,
        axis=-1,
    )
    ratio = torch.exp(logp - torch.squeeze(old_logp))
    adv = torch.squeeze(advantages)
    surrogate = -adv * ratio
    surrogate_clipped = -adv * torch.clamp(ratio, 1.0 - clip_param, 1.0 + clip_param)
    surrogate_loss = torch.max(surrogate, surrogate_clipped).mean()
    if use_clipped_value_loss:
        value_clipped = old_values + (values - old_values).clamp(-clip_param, clip_param)
        value_losses = (values - returns).pow(2)
        value_losses_clipped = (value_clipped - returns).pow(2)
        value_loss = torch.max(value_losses, value_losses_clipped).mean()
    else:
        value_loss = (returns - values).pow(2).mean()
    return surrogate_loss, value_loss, entropy.mean(), kl.mean()


def adaptive_lr(lr: float, kl_mean: float, desired_kl: float) -> float:
    """PPO.update, schedule == 'adaptive'."""
    if kl_mean > desired_kl * 2.0:
        return max(1e-5, lr / 1.5)
    if kl_mean < desired_kl / 2.0 and kl_mean > 0.0:
        return min(1e-2, lr * 1.5)
    return lr


class EmpiricalNormalizationOracle:
    """rsl_rl/modules/normalizer.py::EmpiricalNormalization (v2.3.1): running mean/var with the biased batch variance."""

    def __init__(self, dim: int, eps: float = 1e-2):
        self.eps = eps
        self.mean = torch.zeros(1, dim)
        self.var = torch.ones(1, dim)
        self.std = torch.ones(1, dim)
        self.count = 0

    def forward(self, x, training: bool = True):
        if training:
            count_x = x.shape[0]
            self.count += count_x
            rate = count_x / self.count
            var_x = torch.var(x, dim=0, unbiased=False, keepdim=True)
            mean_x = torch.mean(x, dim=0, keepdim=True)
            delta_mean = mean_x - self.mean
            self.mean = self.mean + rate * delta_mean
            self.var = self.var + rate * (var_x - self.var + delta_mean * (mean_x - self.mean))
            self.std = torch.sqrt(self.var)
        return (x - self.mean) / (self.std + self.eps)
