"""PPO (upstream ``rsl_rl/algorithms/ppo.py`` @ v2.3.1 -- third-party, absent from the reference tree; PARITY UNPINNED,
restated in oracle/rsl_rl_oracle.py; cfg: reference isaaclab_rl/rsl_rl/rl_cfg.py:107-163).

MI355X-first changes (results unchanged):
  * the elementwise loss (log-prob, entropy, KL, clipped surrogate, clipped value loss) is one HIP forward and one
    HIP backward kernel (``imx_ppo_loss_fwd/bwd``) behind a ``torch.autograd.Function``;
  * all parameters are views of ONE flat fp32 bucket, all gradients views of one flat gradient bucket: the
    data-parallel all-reduce is a single RCCL call on that bucket (+1 slot carrying the KL estimate), no
    ``torch.cat`` / scatter per minibatch;
  * grad-norm clipping + Adam are one kernel (``imx_adam_step``) reading the learning rate and the gradient norm
    from device memory, so the adaptive-KL schedule never synchronises with the host.
"""

from __future__ import annotations

import torch
import torch.distributed as dist
import torch.nn as nn

from .. import _lib
from .._lib import check, lib
from .storage import RolloutStorage


class _FusedPPOLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mu, sigma, value, actions, old_logp, old_mu, old_sigma, adv, ret, old_value, clip, clipped, vcoef, ecoef):
        L = lib()
        M, A = mu.shape
        mu_c, sg_c, v_c = mu.contiguous(), sigma.contiguous(), value.contiguous()
        args = [t.contiguous() for t in (actions, old_logp, old_mu, old_sigma, adv, ret, old_value)]
        actions, old_logp, old_mu, old_sigma, adv, ret, old_value = args
        out4 = torch.empty(4, device=mu.device)
        scratch = torch.empty(int(L.imx_ppo_scratch_bytes(M)), dtype=torch.uint8, device=mu.device)
        stream = _lib.current_stream(mu.device)
        check(L.imx_ppo_loss_fwd(M, A, mu_c.data_ptr(), sg_c.data_ptr(), actions.data_ptr(), old_logp.data_ptr(),
                                 old_mu.data_ptr(), old_sigma.data_ptr(), adv.data_ptr(), ret.data_ptr(), v_c.data_ptr(),
                                 old_value.data_ptr(), float(clip), int(clipped), out4.data_ptr(), scratch.data_ptr(), stream))
        ctx.save_for_backward(mu_c, sg_c, v_c, actions, old_logp, adv, ret, old_value)
        ctx.cfg = (float(clip), int(clipped), float(vcoef), float(ecoef))
        loss = out4[0] + vcoef * out4[1] - ecoef * out4[2]
        ctx.mark_non_differentiable(out4)
        return loss, out4

    @staticmethod
    def backward(ctx, g_loss, _g_stats):
        mu, sigma, value, actions, old_logp, adv, ret, old_value = ctx.saved_tensors
        clip, clipped, vcoef, ecoef = ctx.cfg
        M, A = mu.shape
        dmu, dsigma, dvalue = torch.empty_like(mu), torch.empty_like(sigma), torch.empty_like(value)
        check(lib().imx_ppo_loss_bwd(M, A, mu.data_ptr(), sigma.data_ptr(), actions.data_ptr(), old_logp.data_ptr(),
                                     adv.data_ptr(), ret.data_ptr(), value.data_ptr(), old_value.data_ptr(), clip, clipped,
                                     vcoef, ecoef, 1.0, dmu.data_ptr(), dsigma.data_ptr(), dvalue.data_ptr(),
                                     _lib.current_stream(mu.device)))
        if not _FusedPPOLoss.assume_unit_grad:
            dmu, dsigma, dvalue = dmu * g_loss, dsigma * g_loss, dvalue * g_loss
        return (dmu, dsigma, dvalue) + (None,) * 11

    assume_unit_grad = False


def fused_ppo_loss(mu, sigma, actions, old_logp, old_mu, old_sigma, adv, ret, value, old_value, clip_param,
                   use_clipped_value_loss, value_loss_coef, entropy_coef):
    """Returns ``(loss, stats)`` with ``stats = [surrogate, value_loss, entropy_mean, kl_mean]`` (device tensor)."""
    return _FusedPPOLoss.apply(mu, sigma, value, actions, old_logp, old_mu, old_sigma, adv, ret, old_value, clip_param,
                               use_clipped_value_loss, value_loss_coef, entropy_coef)


class FlatParams:
    """Re-homes every parameter (and gradient) of ``module`` into one contiguous fp32 bucket."""

    def __init__(self, module: nn.Module, extra_slots: int = 1):
        params = [p for p in module.parameters() if p.requires_grad]
        self.numel = sum(p.numel() for p in params)
        dev = params[0].device
        self.flat = torch.zeros(self.numel, device=dev)
        self.grad = torch.zeros(self.numel + extra_slots, device=dev)  # trailing slots: KL (piggy-backs the all-reduce)
        off = 0
        for p in params:
            n = p.numel()
            self.flat[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + n].view_as(p.data)
            p.grad = self.grad[off:off + n].view_as(p.data)
            off += n
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self.step = 0

    def zero_grad(self):
        self.grad.zero_()


def allreduce_mean_(bucket_grad: torch.Tensor, world_size: int) -> None:
    """Mean of the flat gradient bucket (+ trailing KL slot) over ranks with ONE collective (RCCL on GPUs)."""
    dist.all_reduce(bucket_grad, op=dist.ReduceOp.SUM)
    bucket_grad.div_(world_size)


def adaptive_lr_(lr: torch.Tensor, kl: torch.Tensor, desired_kl: float) -> None:
    """rsl_rl PPO.update 'adaptive' schedule, evaluated on the device (no host sync, identical on every rank
    because ``kl`` is the all-reduced value): lr /= 1.5 if kl > 2*desired; lr *= 1.5 if 0 < kl < desired/2."""
    up = torch.clamp(lr * 1.5, max=1e-2)
    down = torch.clamp(lr / 1.5, min=1e-5)
    lr.copy_(torch.where(kl > desired_kl * 2.0, down, torch.where((kl < desired_kl / 2.0) & (kl > 0.0), up, lr)))


class PPO:
    def __init__(self, policy, num_learning_epochs=1, num_mini_batches=1, clip_param=0.2, gamma=0.998, lam=0.95,
                 value_loss_coef=1.0, entropy_coef=0.0, learning_rate=1e-3, max_grad_norm=1.0,
                 use_clipped_value_loss=True, schedule="fixed", desired_kl=0.01, device="cpu",
                 normalize_advantage_per_mini_batch=False, rnd_cfg=None, symmetry_cfg=None, multi_gpu_cfg=None, **kwargs):
        if rnd_cfg is not None or symmetry_cfg is not None:
            raise NotImplementedError("RND / symmetry augmentation are outside the hot-path scope (SURVEY.md section 8)")
        self.device = torch.device(device)
        self.is_multi_gpu = multi_gpu_cfg is not None
        if self.is_multi_gpu:
            self.gpu_global_rank = multi_gpu_cfg["global_rank"]
            self.gpu_world_size = multi_gpu_cfg["world_size"]
        else:
            self.gpu_global_rank, self.gpu_world_size = 0, 1
        self.policy = policy.to(self.device)
        self.bucket = FlatParams(self.policy)
        self.storage: RolloutStorage | None = None
        self.transition = RolloutStorage.Transition()
        self.clip_param, self.num_learning_epochs, self.num_mini_batches = clip_param, num_learning_epochs, num_mini_batches
        self.value_loss_coef, self.entropy_coef, self.gamma, self.lam = value_loss_coef, entropy_coef, gamma, lam
        self.max_grad_norm, self.use_clipped_value_loss = max_grad_norm, use_clipped_value_loss
        self.desired_kl, self.schedule = desired_kl, schedule
        self.normalize_advantage_per_mini_batch = normalize_advantage_per_mini_batch
        self._lr = torch.full((1,), float(learning_rate), device=self.device)  # device-side: adaptive schedule w/o sync
        self.betas, self.eps = (0.9, 0.999), 1e-8
        self._stats = torch.zeros(5, device=self.device)  # running sums: value, surrogate, entropy, kl, count

    @property
    def learning_rate(self) -> float:
        return float(self._lr.item())

    # ---- storage / rollout -----------------------------------------------------------------------------------
    def init_storage(self, training_type, num_envs, num_transitions_per_env, actor_obs_shape, critic_obs_shape, actions_shape):
        self.storage = RolloutStorage(num_envs, num_transitions_per_env, actor_obs_shape, critic_obs_shape, actions_shape,
                                      self.device)

    def act(self, obs, critic_obs):
        tr = self.transition
        tr.actions = self.policy.act(obs).detach()
        tr.values = self.policy.evaluate(critic_obs).detach()
        tr.actions_log_prob = self.policy.get_actions_log_prob(tr.actions).detach()
        tr.action_mean = self.policy.action_mean.detach()
        tr.action_sigma = self.policy.action_std.detach()
        tr.observations = obs
        tr.privileged_observations = critic_obs
        return tr.actions

    def process_env_step(self, rewards, dones, infos):
        tr = self.transition
        tr.rewards = rewards.clone()
        tr.dones = dones
        if "time_outs" in infos:  # bootstrap on time-outs
            tr.rewards += self.gamma * torch.squeeze(tr.values * infos["time_outs"].unsqueeze(1).to(self.device), 1)
        self.storage.add_transitions(tr)
        tr.clear()
        self.policy.reset(dones)

    def compute_returns(self, last_critic_obs):
        last_values = self.policy.evaluate(last_critic_obs).detach()
        self.storage.compute_returns(last_values, self.gamma, self.lam,
                                     normalize_advantage=not self.normalize_advantage_per_mini_batch)

    # ---- multi-GPU (one process per GPU, RCCL over xGMI) ---------------------------------------------------
    def broadcast_parameters(self):
        dist.broadcast(self.bucket.flat, src=0)

    def reduce_parameters(self):
        """Mean of the flat gradient bucket (+ KL slot) over ranks: ONE all-reduce."""
        allreduce_mean_(self.bucket.grad, self.gpu_world_size)

    # ---- update ----------------------------------------------------------------------------------------------------
    def update(self):
        b = self.bucket
        L = lib()
        stream = _lib.current_stream(self.device)
        self._stats.zero_()
        _FusedPPOLoss.assume_unit_grad = True
        gen = self.storage.mini_batch_generator(self.num_mini_batches, self.num_learning_epochs)
        for (obs, critic_obs, actions, target_values, advantages, returns, old_logp, old_mu, old_sigma) in gen:
            if self.normalize_advantage_per_mini_batch:
                with torch.no_grad():
                    advantages = (advantages - advantages.mean()) / (advantages.std() + 1e-8)
            mu = self.policy.actor(obs)
            sigma = self.policy._std(mu)
            value = self.policy.critic(critic_obs)
            loss, stats = fused_ppo_loss(mu, sigma, actions, old_logp, old_mu, old_sigma, advantages, returns, value,
                                         target_values, self.clip_param, self.use_clipped_value_loss,
                                         self.value_loss_coef, self.entropy_coef)
            b.zero_grad()
            loss.backward()
            b.grad[-1] = stats[3]  # KL estimate rides in the gradient bucket
            if self.is_multi_gpu:
                self.reduce_parameters()
            if self.desired_kl is not None and self.schedule == "adaptive":
                adaptive_lr_(self._lr, b.grad[-1], self.desired_kl)
            g = b.grad[:b.numel]
            norm = torch.linalg.vector_norm(g).reshape(1) if self.max_grad_norm is not None else None
            b.step += 1
            check(L.imx_adam_step(b.numel, b.flat.data_ptr(), g.data_ptr(), b.exp_avg.data_ptr(), b.exp_avg_sq.data_ptr(),
                                  self._lr.data_ptr(), _lib.ptr(norm), float(self.max_grad_norm or 0.0), self.betas[0],
                                  self.betas[1], self.eps, b.step, stream))
            self._stats[:4] += torch.stack([stats[1], stats[0], stats[2], stats[3]])
            self._stats[4] += 1
        _FusedPPOLoss.assume_unit_grad = False
        self.storage.clear()
        return self._stats  # device tensor; .tolist() only when the caller wants to log

    def loss_dict(self) -> dict:
        s = self._stats.tolist()
        n = max(s[4], 1.0)
        return {"value_function": s[0] / n, "surrogate": s[1] / n, "entropy": s[2] / n, "kl": s[3] / n}
