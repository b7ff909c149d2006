"""Rollout storage + GAE for the RSL-RL PPO rollout (upstream ``rsl_rl/storage/rollout_storage.py`` @ v2.3.1, a
third-party dependency absent from the reference tree -- PARITY UNPINNED, see oracle/rsl_rl_oracle.py).

Buffers are preallocated ``(T, N, .)`` tensors that stay resident in HBM for the whole run (fixed pointers: the
rollout can be captured in a hipGraph); the GAE backward scan and the advantage normalisation are ``imx_gae``.
"""

from __future__ import annotations

import torch

from .. import _lib
from .._lib import check, lib


def gae_returns(rewards, values, dones, last_values, gamma: float, lam: float, normalize: bool = True,
                returns=None, advantages=None, scratch=None):
    """``RolloutStorage.compute_returns``: (T,N,1) rewards/values/dones(uint8) + (N,1) last values -> returns, advantages."""
    T, N = rewards.shape[0], rewards.shape[1]
    dev = rewards.device
    if returns is None:
        returns = torch.empty_like(rewards)
    if advantages is None:
        advantages = torch.empty_like(rewards)
    L = lib()
    if scratch is None:
        scratch = torch.zeros(int(L.imx_gae_scratch_bytes(T, N)), dtype=torch.uint8, device=dev)
    if dones.dtype not in (torch.uint8, torch.bool):
        raise TypeError("dones must be uint8/bool")
    check(L.imx_gae(T, N, _lib.ptr(rewards), _lib.ptr(values), _lib.ptr(dones), _lib.ptr(last_values), float(gamma),
                    float(lam), 1 if normalize else 0, _lib.ptr(returns), _lib.ptr(advantages), _lib.ptr(scratch),
                    _lib.current_stream(dev)))
    return returns, advantages


class RolloutStorage:
    class Transition:
        def __init__(self):
            self.observations = None
            self.privileged_observations = None
            self.actions = None
            self.rewards = None
            self.dones = None
            self.values = None
            self.actions_log_prob = None
            self.action_mean = None
            self.action_sigma = None

        def clear(self):
            self.__init__()

    def __init__(self, num_envs, num_transitions_per_env, obs_shape, privileged_obs_shape, actions_shape, device="cpu"):
        self.device = device
        self.num_envs = num_envs
        self.num_transitions_per_env = T = num_transitions_per_env
        N = num_envs
        z = lambda *s, **k: torch.zeros(*s, device=device, **k)  # noqa: E731
        self.observations = z(T, N, *obs_shape)
        self.privileged_observations = z(T, N, *privileged_obs_shape) if privileged_obs_shape and privileged_obs_shape[0] else None
        self.rewards = z(T, N, 1)
        self.actions = z(T, N, *actions_shape)
        self.dones = z(T, N, 1, dtype=torch.uint8)
        self.actions_log_prob = z(T, N, 1)
        self.values = z(T, N, 1)
        self.returns = z(T, N, 1)
        self.advantages = z(T, N, 1)
        self.mu = z(T, N, *actions_shape)
        self.sigma = z(T, N, *actions_shape)
        self._gae_scratch = None
        self.step = 0

    def add_transitions(self, transition: "RolloutStorage.Transition"):
        if self.step >= self.num_transitions_per_env:
            raise OverflowError("Rollout buffer overflow! You should call clear() before adding new transitions.")
        t = self.step
        self.observations[t].copy_(transition.observations)
        if self.privileged_observations is not None:
            self.privileged_observations[t].copy_(transition.privileged_observations)
        self.actions[t].copy_(transition.actions)
        self.rewards[t].copy_(transition.rewards.view(-1, 1))
        self.dones[t].copy_(transition.dones.view(-1, 1))
        self.values[t].copy_(transition.values)
        self.actions_log_prob[t].copy_(transition.actions_log_prob.view(-1, 1))
        self.mu[t].copy_(transition.action_mean)
        self.sigma[t].copy_(transition.action_sigma)
        self.step += 1

    def clear(self):
        self.step = 0

    def compute_returns(self, last_values, gamma, lam, normalize_advantage: bool = True):
        if self._gae_scratch is None:
            n = int(lib().imx_gae_scratch_bytes(self.num_transitions_per_env, self.num_envs))
            self._gae_scratch = torch.zeros(n, dtype=torch.uint8, device=self.device)  # (barrier counters start at zero)
        gae_returns(self.rewards, self.values, self.dones, last_values.contiguous(), gamma, lam, normalize_advantage,
                    self.returns, self.advantages, self._gae_scratch)

    def _minibatch_setup(self, num_mini_batches):
        """Sources, reusable destination buffers and argument arrays of the one-launch minibatch gather; the permutation lives in a
        persistent buffer (fixed address: a captured gather reads whatever permutation was drawn into it last)."""
        import ctypes

        batch_size = self.num_envs * self.num_transitions_per_env
        M = batch_size // num_mini_batches
        srcs = [self.observations.flatten(0, 1)]
        if self.privileged_observations is not None:
            srcs.append(self.privileged_observations.flatten(0, 1))
        srcs += [self.actions.flatten(0, 1), self.values.flatten(0, 1), self.advantages.flatten(0, 1),
                 self.returns.flatten(0, 1), self.actions_log_prob.flatten(0, 1), self.mu.flatten(0, 1),
                 self.sigma.flatten(0, 1)]
        key = (M, len(srcs), num_mini_batches)
        if getattr(self, "_mb_key", None) != key:
            # wide rows (observations) start on 16-byte boundaries: the dW kernel of the first layer reads them with 16-byte loads
            # even when the width is ragged (235 -> pitch 236: 81.8 -> 71.2 us, tools/dw_pitch.py); the views keep the true width
            # ONLY the observation arrays are padded: their consumers (GEMMs, imx_mlp_dw, imx_mlp_fwd_elu) take a row pitch; the per-sample
            # arrays of width A (actions, old mu / sigma) go to kernels that address them densely -- padding them too fed the loss kernels
            # wrong columns for every A >= 16 that is not a multiple of four (17, 18, 37: found by tools/fuzz_kernels.py in round 3)
            n_obs = 2 if self.privileged_observations is not None else 1
            pitch = [(s.shape[1] + 3) // 4 * 4 if (k < n_obs and s.shape[1] >= 16) else s.shape[1] for k, s in enumerate(srcs)]
            # ONE set of batch buffers PER MINIBATCH of the permutation: upstream draws the permutation once per update and walks the same
            # num_mini_batches index slices in every epoch, so each slice is gathered once (first epoch) and its buffers are reused in the
            # later epochs -- 4 gather launches per update instead of 20 (the whole storage once more in HBM: 108 MB at 4096 x 24)
            self._mb_dst_sets = [[torch.empty(M, p, device=self.device)[:, :s.shape[1]] for s, p in zip(srcs, pitch)] for _ in range(num_mini_batches)]
            self._mb_dst = self._mb_dst_sets[0]
            self._mb_perm = torch.empty(num_mini_batches * M, dtype=torch.int64, device=self.device)
            n = len(srcs)
            src_p = (ctypes.c_void_p * n)(*[s.data_ptr() for s in srcs])
            widths = (ctypes.c_int32 * n)(*[s.shape[1] for s in srcs])
            self._mb_args_sets = [(src_p, (ctypes.c_void_p * n)(*[d.data_ptr() for d in dst]), widths,
                                   (ctypes.c_int32 * n)(*[d.stride(0) for d in dst]), n, M) for dst in self._mb_dst_sets]
            self._mb_args = self._mb_args_sets[0]
            self._mb_key = key
        return self._mb_args

    def draw_permutation(self, num_mini_batches):
        """A fresh random permutation of the T*N transitions into the persistent index buffer (upstream: torch.randperm per update)."""
        self._minibatch_setup(num_mini_batches)
        torch.randperm(self._mb_perm.numel(), out=self._mb_perm)
        return self._mb_perm

    def gather_minibatch(self, i, num_mini_batches, stream=None, reuse: bool = False):
        """Minibatch ``i`` of the permutation in the index buffer -> ITS batch buffers (ONE ``imx_gather_rows`` launch; ``reuse``: the
        buffers still hold this slice from an earlier epoch of the same update -- nothing is launched); returns
        (obs, critic_obs, actions, values, advantages, returns, old_log_prob, old_mu, old_sigma)."""
        self._minibatch_setup(num_mini_batches)
        src_p, dst_p, widths, pitches, n, M = self._mb_args_sets[i]
        if not reuse:
            idx = self._mb_perm[i * M:(i + 1) * M]
            st = _lib.current_stream(torch.device(self.device)) if stream is None else stream
            check(lib().imx_gather_rows_pitched(M, idx.data_ptr(), n, src_p, dst_p, widths, pitches, st))
        dst = self._mb_dst_sets[i]
        return tuple(dst) if self.privileged_observations is not None else (dst[0], dst[0]) + tuple(dst[1:])

    def mini_batch_generator(self, num_mini_batches, num_epochs=8, copy_stream=None):
        """Yields (obs, critic_obs, actions, values, advantages, returns, old_log_prob, old_mu, old_sigma) minibatches
        of a random permutation of the T*N transitions; the nine gathers are ONE ``imx_gather_rows`` launch into
        reusable buffers (valid until the next minibatch is requested).  With ``copy_stream`` the gather is issued on
        that stream (after everything enqueued on the current stream so far) and ``(batch, ready_event)`` is yielded: the
        caller requests the next minibatch right after the backward pass and waits for ``ready_event`` only before it
        uses the data, so the HBM-bound gather runs beside the optimiser step instead of in front of the next forward."""
        self.draw_permutation(num_mini_batches)
        main = torch.cuda.current_stream(torch.device(self.device)) if copy_stream is not None else None
        for epoch in range(num_epochs):
            for i in range(num_mini_batches):
                if copy_stream is not None:
                    copy_stream.wait_stream(main)
                    batch = self.gather_minibatch(i, num_mini_batches, copy_stream.cuda_stream, reuse=epoch > 0)
                    ready = torch.cuda.Event()
                    ready.record(copy_stream)
                    yield batch, ready
                else:
                    yield self.gather_minibatch(i, num_mini_batches, reuse=epoch > 0)
