"""PPO (upstream ``rsl_rl/algorithms/ppo.py`` @ v2.3.1 -- third-party, absent from the reference tree; PARITY UNPINNED,
restated in oracle/rsl_rl_oracle.py; cfg: reference isaaclab_rl/rsl_rl/rl_cfg.py:107-163).

MI355X-first structure of ``update()`` (same arithmetic as upstream):
  * all parameters are views of ONE flat fp32 bucket, all gradients views of one flat gradient bucket (+1 slot that
    carries the KL estimate): the data-parallel exchange is a single RCCL ``all_reduce`` on that bucket, no
    ``torch.cat`` / scatter per minibatch;
  * the backward pass is written out (no autograd graph): per MLP layer one GEMM writes ``dW`` *directly into the flat
    gradient bucket* (no ``zero_grad``, no ``AccumulateGrad`` adds), one reduction ``db``, one GEMM ``dX`` and one
    ``elu_backward``; the GEMMs stay in torch (hipBLASLt / rocBLAS, recorded TunableOp selections);
  * the elementwise loss (log-prob, entropy, KL, clipped surrogate, clipped value loss) is one HIP forward + one HIP
    backward kernel (``imx_ppo_loss_fwd/bwd``) reading the shared ``std`` vector with stride 0;
  * the adaptive-KL learning-rate rule, Adam step bookkeeping, ``clip_grad_norm_`` and ``Adam.step`` are two kernels
    (``imx_adam_update``) working from device scalars: ``update()`` never synchronises with the host.
``fused_ppo_loss`` (autograd form) is kept for users who build their own update on torch autograd.
"""

from __future__ import annotations

import ctypes
import os

import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib
from .._lib import check, lib
from .storage import RolloutStorage


# ---------------------------------------------------------------------------------------------------- fused loss (autograd form)
class _FusedPPOLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mu, sigma, value, actions, old_logp, old_mu, old_sigma, adv, ret, old_value, clip, clipped, vcoef, ecoef):
        L = lib()
        M, A = mu.shape
        mu_c, sg_c, v_c = mu.contiguous(), sigma.contiguous(), value.contiguous()
        args = [t.contiguous() for t in (actions, old_logp, old_mu, old_sigma, adv, ret, old_value)]
        actions, old_logp, old_mu, old_sigma, adv, ret, old_value = args
        out8 = torch.empty(8, device=mu.device)
        scratch = torch.empty(int(L.imx_ppo_scratch_bytes(M)), dtype=torch.uint8, device=mu.device)
        check(L.imx_ppo_loss_fwd(M, A, mu_c.data_ptr(), sg_c.data_ptr(), A, actions.data_ptr(), old_logp.data_ptr(),
                                 old_mu.data_ptr(), old_sigma.data_ptr(), adv.data_ptr(), ret.data_ptr(), v_c.data_ptr(),
                                 old_value.data_ptr(), float(clip), int(clipped), float(vcoef), float(ecoef),
                                 out8.data_ptr(), None, scratch.data_ptr(), _lib.current_stream(mu.device)))
        ctx.save_for_backward(mu_c, sg_c, v_c, actions, old_logp, adv, ret, old_value)
        ctx.cfg = (float(clip), int(clipped), float(vcoef), float(ecoef))
        stats = out8[:4]
        ctx.mark_non_differentiable(stats)
        return out8[4], stats

    @staticmethod
    def backward(ctx, g_loss, _g_stats):
        mu, sigma, value, actions, old_logp, adv, ret, old_value = ctx.saved_tensors
        clip, clipped, vcoef, ecoef = ctx.cfg
        M, A = mu.shape
        dmu, dsigma, dvalue = torch.empty_like(mu), torch.empty_like(sigma), torch.empty_like(value)
        check(lib().imx_ppo_loss_bwd(M, A, mu.data_ptr(), sigma.data_ptr(), A, actions.data_ptr(), old_logp.data_ptr(),
                                     adv.data_ptr(), ret.data_ptr(), value.data_ptr(), old_value.data_ptr(), clip, clipped,
                                     vcoef, ecoef, 1.0, dmu.data_ptr(), dsigma.data_ptr(), dvalue.data_ptr(),
                                     _lib.current_stream(mu.device)))
        return (dmu * g_loss, dsigma * g_loss, dvalue * g_loss) + (None,) * 11


def fused_ppo_loss(mu, sigma, actions, old_logp, old_mu, old_sigma, adv, ret, value, old_value, clip_param,
                   use_clipped_value_loss, value_loss_coef, entropy_coef):
    """Returns ``(loss, stats)`` with ``stats = [surrogate, value_loss, entropy_mean, kl_mean]`` (device tensor)."""
    return _FusedPPOLoss.apply(mu, sigma, value, actions, old_logp, old_mu, old_sigma, adv, ret, old_value, clip_param,
                               use_clipped_value_loss, value_loss_coef, entropy_coef)


# ---------------------------------------------------------------------------------------------------- flat buckets
class FlatParams:
    """Re-homes every parameter (and gradient) of ``module`` into one contiguous fp32 bucket."""

    def __init__(self, module: nn.Module, extra_slots: int = 1, first=()):
        """``first``: parameters to lay out first, in this order (e.g. the two first-layer weights next to each other so that
        they can be used as one stacked matrix); the rest follows in ``module.parameters()`` order."""
        params = [p for p in module.parameters() if p.requires_grad]
        head = [p for p in first if any(p is q for q in params)]
        params = head + [p for p in params if not any(p is q for q in head)]
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

    def zero_grad(self):
        self.grad.zero_()


def allreduce_mean_(bucket_grad: torch.Tensor, world_size: int) -> None:
    """Mean of the flat gradient bucket (+ trailing KL slot) over ranks with ONE collective (RCCL on GPUs)."""
    dist.all_reduce(bucket_grad, op=dist.ReduceOp.SUM)
    bucket_grad.div_(world_size)


def adaptive_lr_(lr: torch.Tensor, kl: torch.Tensor, desired_kl: float) -> None:
    """rsl_rl PPO.update 'adaptive' schedule as tensor ops (reference semantics of ``k_adam_prepare``; identical on
    every rank because ``kl`` is the all-reduced value): lr /= 1.5 if kl > 2*desired; lr *= 1.5 if 0 < kl < desired/2."""
    up = torch.clamp(lr * 1.5, max=1e-2)
    down = torch.clamp(lr / 1.5, min=1e-5)
    lr.copy_(torch.where(kl > desired_kl * 2.0, down, torch.where((kl < desired_kl / 2.0) & (kl > 0.0), up, lr)))


# ---------------------------------------------------------------------------------------------------- explicit MLP fwd/bwd
def _mlp_layers(seq: nn.Sequential):
    """[(Linear, activation module or None)] of an actor/critic ``nn.Sequential`` built by ``ActorCritic``."""
    mods = list(seq)
    out = []
    for i, m in enumerate(mods):
        if isinstance(m, nn.Linear):
            act = mods[i + 1] if i + 1 < len(mods) and not isinstance(mods[i + 1], nn.Linear) else None
            out.append((m, act))
    return out


class FusedInference:
    """``imx_mlp_infer`` bound to one or two Linear+ELU stacks (actor, critic) that share their input: argument arrays are
    built once (the parameters live at fixed addresses in the flat bucket)."""

    MAX_WIDTH, MAX_LAYERS = 512, 4

    def __init__(self, *nets):
        import ctypes

        self.ok = 1 <= len(nets) <= 2
        dims, ws, bs, alphas, nl = [], [], [], [], []
        for layers in nets:
            acts = [a for _, a in layers[:-1]]
            if (len(layers) > self.MAX_LAYERS or layers[-1][1] is not None or any(not isinstance(a, nn.ELU) for a in acts)
                    or any(max(lin.in_features, lin.out_features) > self.MAX_WIDTH for lin, _ in layers)
                    or layers[0][0].in_features != nets[0][0][0].in_features):
                self.ok = False
                return
            nl.append(len(layers))
            dims += [layers[0][0].in_features] + [lin.out_features for lin, _ in layers]
            ws += [lin.weight for lin, _ in layers]
            bs += [lin.bias for lin, _ in layers]
            alphas.append(float(acts[0].alpha) if acts else 1.0)
        self.in_features = dims[0]
        self.out_features = [layers[-1][0].out_features for layers in nets]
        # Layers whose in-features are not a multiple of 32 (235 observations) get a zero-padded copy of their weights with a
        # row pitch that is (the kernel issues unconditional 16-byte loads over whole 32-wide reduction groups);
        # ``refresh()`` re-copies them after the parameters changed (once per rollout).
        self._padded = []
        pitch, wp = [], []
        for w in ws:
            k = w.shape[1]
            if k % 32 == 0 and w.data_ptr() % 16 == 0:
                pitch.append(k)
                wp.append(w)
            else:
                kp = (k + 31) & ~31
                buf = torch.zeros(w.shape[0], kp, device=w.device, dtype=w.dtype)
                self._padded.append((buf, w))
                pitch.append(kp)
                wp.append(buf)
        # ... and every layer a PACKED copy in the lane order of the 32-sample kernel (imx_mlp_pack_weights): one contiguous KiB per load
        # instruction instead of 16 bytes from each of 32 rows; refreshed with the padded copies
        L = lib()
        self._packed = [torch.zeros(int(L.imx_mlp_packed_floats(w.shape[0], w.shape[1])), device=w.device, dtype=w.dtype) for w in ws]
        self._wpk = (ctypes.c_void_p * len(ws))(*[b.data_ptr() for b in self._packed]) if os.getenv("IMX_INFER_PACKED", "1") != "0" else None
        nw = len(ws)
        self._pack_args = (nw, (ctypes.c_int * nw)(*[w.shape[0] for w in ws]), (ctypes.c_int * nw)(*[w.shape[1] for w in ws]),
                           (ctypes.c_void_p * nw)(*[w.data_ptr() for w in ws]), (ctypes.c_int64 * nw)(*[w.stride(0) for w in ws]),
                           (ctypes.c_void_p * nw)(*[b.data_ptr() for b in self._packed]))
        self._keep = (ws, bs, wp)
        self._nl = (ctypes.c_int * len(nl))(*nl)
        self._dims = (ctypes.c_int * len(dims))(*dims)
        self._w = (ctypes.c_void_p * len(wp))(*[w.data_ptr() for w in wp])
        self._pitch = (ctypes.c_int * len(pitch))(*pitch)
        self._b = (ctypes.c_void_p * len(bs))(*[b.data_ptr() for b in bs])
        self._alpha = (ctypes.c_float * len(alphas))(*alphas)
        self._n = len(nets)
        self._ctypes = ctypes
        self.refresh()

    @torch.no_grad()
    def refresh(self):
        """Re-copy the padded weight buffers from the live parameters (capturable: plain device copies)."""
        for buf, w in self._padded:
            buf[:, :w.shape[1]].copy_(w)
        if self._wpk is not None:  # all layers in one launch
            check(lib().imx_mlp_pack_weights_batch(*self._pack_args, _lib.current_stream(self._packed[0].device)))

    def __call__(self, x: torch.Tensor, *outs: torch.Tensor, act=None):
        """``act``: an ``ImxPolicyAct`` -- the actor's workgroups then also sample the action, write the transition into the storage
        slot it names and run the action through the env's action terms (``imx_mlp_infer_act``); ``outs[0]`` may be None then."""
        if x.stride(1) != 1 or x.shape[1] != self.in_features or any(o is not None and not o.is_contiguous() for o in outs):
            raise _lib.ImxError("imx_mlp_infer needs a row-major input and contiguous outputs")
        out_p = (self._ctypes.c_void_p * self._n)(*[None if o is None else o.data_ptr() for o in outs])
        check(lib().imx_mlp_infer_act(x.shape[0], x.data_ptr(), x.stride(0), self._n, self._nl, self._dims, self._w, self._pitch, self._wpk,
                                      self._b, self._alpha, out_p, self._ctypes.byref(act) if act is not None else None,
                                      _lib.current_stream(x.device)))


HEAD_MAX_OUT = 64  # imx_mlp_head_*: output layers up to 64 wide (action means, value)


def _is_head(lin: nn.Linear, x: torch.Tensor) -> bool:
    return (lin.out_features <= HEAD_MAX_OUT and lin.in_features % 32 == 0 and lin.in_features <= 256
            and x.stride(1) == 1 and x.stride(0) % 4 == 0)


FUSED_FIRST_LAYER_MAX_K = 256  # imx_mlp_fwd_elu keeps a column's weights in registers

# Measured alternatives that stay switchable (NOTES.md has the numbers); read ONCE, at import:
FUSED_L0 = os.getenv("IMX_FUSED_L0", "1") != "0"        # first Linear + ELU as one imx_mlp_fwd_elu launch (off: library GEMM + ELU pass)
FUSED_HEAD = os.getenv("IMX_FUSED_HEAD", "0")           # "1" / "a" / "c": output layer forward + loss + backward in one launch (slower in situ)
LOSS_ON_SIDE = os.getenv("IMX_LOSS_ON_SIDE", "1") == "1"  # loss-value kernels behind the critic's backward pass (off: a third stream)


def fused_first_layer_ok(x: torch.Tensor, w: torch.Tensor) -> bool:
    return (x.is_cuda and x.dtype == torch.float32 and x.stride(1) == 1 and w.is_contiguous() and w.shape[1] == x.shape[1]
            and x.shape[1] <= FUSED_FIRST_LAYER_MAX_K and FUSED_L0)


def mlp_scratch(layers, M: int, device) -> torch.Tensor:
    """Scratch for ``mlp_backward`` (one per network: actor and critic run on two streams at once)."""
    n = max(int(lib().imx_mlp_scratch_bytes(M, lin.out_features, lin.in_features)) for lin, _ in layers)
    return torch.empty(n, dtype=torch.uint8, device=device)


class DeferredReductions:
    """Per-layer scratch + an ``imx_reduce_batch``: the split-partial sums of every layer of one network are flushed in ONE
    launch at the end of ``mlp_backward`` (dW / db are only read by the optimiser step)."""

    def __init__(self, layers, M: int, device):
        import ctypes

        L = lib()
        self.scratch = [torch.empty(int(L.imx_mlp_scratch_bytes(M, lin.out_features, lin.in_features)), dtype=torch.uint8, device=device)
                        for lin, _ in layers]
        h = ctypes.c_void_p()
        check(L.imx_reduce_batch_create(ctypes.byref(h)))
        self.handle = h

    def __del__(self):
        try:
            lib().imx_reduce_batch_destroy(self.handle)
        except Exception:
            pass


def _head_bwd_args(scratch, which="a"):
    """What ``mlp_forward`` needs to run the output layer's backward inside its forward launch (``imx_mlp_head_fwd_bwd``): that layer's
    scratch and the reduce batch its partial sums are queued on; None = the split pair.  OFF by default: alone the one launch is faster
    (38 vs 45 us policy head, 21 vs 31 us value head, tools/head_bench.py) but inside the update, next to the other network's kernels, it
    measured slower (16.3 vs 16.15 ms; NOTES.md).  IMX_FUSED_HEAD=1 both heads, =a / =c the actor's / the critic's only."""
    mode = FUSED_HEAD
    if mode == "0" or (mode == "c" and which != "c") or (mode == "a" and which != "a"):
        return None
    if isinstance(scratch, DeferredReductions):
        return {"scratch": scratch.scratch[-1], "batch": scratch.handle}
    return {"scratch": scratch, "batch": None}


def mlp_forward(layers, x, out=None, head_loss=None, first=None):
    """Returns (output, saved layer inputs).  Wide layers: library GEMM + bias epilogue, ELU in place on its output;
    the narrow output layer: ``imx_mlp_head_fwd`` (written into ``out`` when given).  ``head_loss`` = {"loss": ImxHeadLoss}:
    when the LAST layer goes through the head kernel, the loss gradient is computed in the same launch
    (``imx_mlp_head_fwd_loss``) and ``head_loss["applied"]`` is set.  ``first``: the (already activated) output of layer 0 when
    it was computed elsewhere (actor and critic first layers as one stacked GEMM): the walk starts at layer 1."""
    saved = [x]
    h = x
    pending_elu = None  # ELU of the last hidden layer is applied by the head kernel on its way in (in place)
    for li, (lin, act) in enumerate(layers):
        if li == 0 and first is not None:
            h = first
            saved.append(h)
            continue
        if act is None and _is_head(lin, h):
            if out is not None and li == len(layers) - 1:
                z = out
            else:
                z = torch.empty(h.shape[0], lin.out_features, device=h.device, dtype=h.dtype)
            fb = head_loss.get("bwd") if head_loss is not None and li == len(layers) - 1 else None
            if (fb is not None and lin.in_features in (128, 256) and lin.out_features <= 16 and h.stride(0) % 4 == 0
                    and (pending_elu is not None or li == 0 or layers[li - 1][1] is None)):
                # forward, loss gradient AND this layer's backward in one pass over the last hidden layer (imx_mlp_head_fwd_bwd): its
                # activated values never travel to memory; the gradient handed to the layer below comes back as head_loss["dprev"]
                scr = fb["scratch"]
                dprev = torch.empty(h.shape[0], lin.in_features, device=h.device, dtype=h.dtype)
                check(lib().imx_mlp_head_fwd_bwd(h.shape[0], lin.in_features, lin.out_features, h.data_ptr(), h.stride(0), lin.weight.data_ptr(),
                                                 lin.bias.data_ptr(), z.data_ptr(), int(pending_elu is not None), float(pending_elu or 0.0),
                                                 ctypes.byref(head_loss["loss"]), dprev.data_ptr(), lin.weight.grad.data_ptr(),
                                                 lin.bias.grad.data_ptr(), scr.data_ptr(), scr.numel(), fb.get("batch"),
                                                 _lib.current_stream(h.device)))
                head_loss["applied"] = True
                head_loss["dprev"] = dprev
            elif head_loss is not None and li == len(layers) - 1:
                check(lib().imx_mlp_head_fwd_loss(h.shape[0], lin.in_features, lin.out_features, h.data_ptr(), h.stride(0),
                                                  lin.weight.data_ptr(), lin.bias.data_ptr(), z.data_ptr(), int(pending_elu is not None),
                                                  float(pending_elu or 0.0), ctypes.byref(head_loss["loss"]),
                                                  _lib.current_stream(h.device)))
                head_loss["applied"] = True
            else:
                check(lib().imx_mlp_head_fwd(h.shape[0], lin.in_features, lin.out_features, h.data_ptr(), h.stride(0),
                                             lin.weight.data_ptr(), lin.bias.data_ptr(), z.data_ptr(), int(pending_elu is not None),
                                             float(pending_elu or 0.0), _lib.current_stream(h.device)))
            pending_elu = None
        else:
            if pending_elu is not None:
                h = F.elu(h, alpha=pending_elu, inplace=True)
                pending_elu = None
            # (not when the NEXT layer is the output head: that kernel applies this layer's ELU itself on its way in)
            if li == 0 and isinstance(act, nn.ELU) and li + 1 < len(layers) - 1 and fused_first_layer_ok(h, lin.weight):
                z = torch.empty(h.shape[0], lin.out_features, device=h.device)  # Linear + ELU in one launch (imx_mlp_fwd_elu)
                check(lib().imx_mlp_fwd_elu(h.shape[0], lin.out_features, lin.in_features, h.data_ptr(), h.stride(0), lin.weight.data_ptr(),
                                            lin.bias.data_ptr(), float(act.alpha), 1, z.data_ptr(), z.stride(0), _lib.current_stream(h.device)))
                h = z
                saved.append(h)
                continue
            z = torch.addmm(lin.bias, h, lin.weight.t())  # GEMM + bias epilogue (hipBLASLt)
        if act is None:
            h = z
        elif isinstance(act, nn.ELU):
            nxt = layers[li + 1][0] if li + 1 < len(layers) else None
            if nxt is not None and layers[li + 1][1] is None and _is_head(nxt, z) and li + 1 == len(layers) - 1:
                pending_elu = float(act.alpha)  # z becomes ELU(z) inside the head kernel, same storage
                h = z
            else:
                h = F.elu(z, alpha=act.alpha, inplace=True)
            saved.append(h)
        else:
            h = act(z)
            saved.append((z, h))
    if pending_elu is not None:
        h = F.elu(h, alpha=pending_elu, inplace=True)
    if out is not None and h is not out:
        out.copy_(h)
        h = out
    return h, saved


def mlp_backward(layers, saved, dout, scratch=None, head_dprev=None):
    """Writes dW/db of every layer straight into ``param.grad`` (views of the flat bucket).  dW / db: ``imx_mlp_dw``
    (split over the samples on the f32 MFMA); output layer: ``imx_mlp_head_bwd`` (dW, db, dX and the ELU' of the layer
    below in one pass); the wide dX GEMMs stay in the library, and the ELU backward of a hidden layer is folded into that
    layer's ``imx_mlp_dw_elu`` (no separate elementwise pass)."""
    L = lib()
    M = dout.shape[0]
    if scratch is None:
        scratch = mlp_scratch(layers, M, dout.device)
    stream = _lib.current_stream(dout.device)
    deferred = scratch if isinstance(scratch, DeferredReductions) else None
    if deferred is not None:
        check(L.imx_reduce_batch_begin(deferred.handle))
    try:
        _mlp_backward_layers(L, layers, saved, dout, M, scratch, deferred, stream, head_dprev)
    finally:
        if deferred is not None:
            check(L.imx_reduce_batch_flush(deferred.handle, stream))


def _mlp_backward_layers(L, layers, saved, dout, M, scratch, deferred, stream, head_dprev=None):
    d = dout          # gradient w.r.t. the pre-activation output of layer i ...
    pending = None    # ... or (gradient w.r.t. its ELU output, that output, alpha): resolved inside imx_mlp_dw_elu
    top = len(layers) - 1
    if head_dprev is not None:  # the output layer's backward already ran inside imx_mlp_head_fwd_bwd (mlp_forward)
        d, top = head_dprev, top - 1
    for i in range(top, -1, -1):
        lin, _ = layers[i]
        x = saved[i] if not isinstance(saved[i], tuple) else saved[i][1]
        prev_act = layers[i - 1][1] if i > 0 else None
        N, K = lin.out_features, lin.in_features
        scr = deferred.scratch[i] if deferred is not None else scratch
        if pending is not None:
            dh, h, alpha = pending
            pending = None
            d = torch.empty_like(dh) if i > 0 else None  # the input layer has nothing below it: dZ is not materialised
            check(L.imx_mlp_dw_elu(M, N, K, dh.data_ptr(), dh.stride(0), h.data_ptr(), h.stride(0), float(alpha),
                                   _lib.ptr(d), N, x.data_ptr(), x.stride(0), lin.weight.grad.data_ptr(),
                                   lin.bias.grad.data_ptr(), scr.data_ptr(), scr.numel(), stream))
        else:
            if not d.is_contiguous():
                d = d.contiguous()
            if i > 0 and _is_head(lin, x) and (prev_act is None or isinstance(prev_act, nn.ELU)):
                dprev = torch.empty(M, K, device=d.device, dtype=d.dtype)
                check(L.imx_mlp_head_bwd(M, K, N, d.data_ptr(), x.data_ptr(), x.stride(0), lin.weight.data_ptr(),
                                         float(prev_act.alpha) if prev_act is not None else 0.0, int(prev_act is not None),
                                         dprev.data_ptr(), lin.weight.grad.data_ptr(), lin.bias.grad.data_ptr(),
                                         scr.data_ptr(), scr.numel(), stream))
                d = dprev
                continue
            check(L.imx_mlp_dw(M, N, K, d.data_ptr(), d.stride(0), x.data_ptr(), x.stride(0), lin.weight.grad.data_ptr(),
                               lin.bias.grad.data_ptr(), scr.data_ptr(), scr.numel(), stream))
        if i > 0:
            dx = torch.mm(d, lin.weight)
            if isinstance(prev_act, nn.ELU) and dx.stride(1) == 1 and saved[i].stride(1) == 1:
                pending = (dx, saved[i], prev_act.alpha)  # ELU'(z) from the saved output: y > 0 ? 1 : y + alpha
            elif isinstance(prev_act, nn.ELU):
                d = torch.ops.aten.elu_backward(dx, prev_act.alpha, 1.0, 1.0, True, saved[i])
            elif prev_act is None:
                d = dx
            else:
                z, _h = saved[i]
                with torch.enable_grad():
                    zz = z.detach().requires_grad_(True)
                    (d,) = torch.autograd.grad(prev_act(zz), zz, dx)


class PPO:
    def __init__(self, policy, num_learning_epochs=1, num_mini_batches=1, clip_param=0.2, gamma=0.998, lam=0.95,
                 value_loss_coef=1.0, entropy_coef=0.0, learning_rate=1e-3, max_grad_norm=1.0,
                 use_clipped_value_loss=True, schedule="fixed", desired_kl=0.01, device="cpu",
                 normalize_advantage_per_mini_batch=False, rnd_cfg=None, symmetry_cfg=None, multi_gpu_cfg=None, two_streams=True,
                 **kwargs):
        if kwargs:  # e.g. the PPOCA / distillation cfg fields of isaaclab_rl/rsl_rl/rl_cfg.py:166-176
            raise NotImplementedError("PPO got arguments it does not implement: " + ", ".join(sorted(kwargs)))
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
        # actor and critic first layers side by side in the bucket: when both read the same observations they are ONE stacked GEMM
        pair = self._first_layer_pair()
        self.bucket = FlatParams(self.policy, extra_slots=8,  # trailing slots = the 8 loss scalars (KL among them)
                                 first=() if pair is None else (pair[0].weight, pair[1].weight, pair[0].bias, pair[1].bias))
        self._joint0 = None
        if pair is not None:
            la, lc, _alpha = pair
            H, K = la.out_features, la.in_features
            if (lc.weight.data_ptr() == la.weight.data_ptr() + 4 * H * K and lc.bias.data_ptr() == la.bias.data_ptr() + 4 * H
                    and la.weight.is_contiguous() and lc.weight.is_contiguous()):
                w0 = self.bucket.flat[:2 * H * K].view(2 * H, K)
                b0 = self.bucket.flat[2 * H * K:2 * H * K + 2 * H]
                if w0.data_ptr() == la.weight.data_ptr() and b0.data_ptr() == la.bias.data_ptr():
                    self._joint0 = (w0, b0, H, float(pair[2]))
        self.storage: RolloutStorage | None = None
        self.transition = RolloutStorage.Transition()
        self.clip_param, self.num_learning_epochs, self.num_mini_batches = clip_param, num_learning_epochs, num_mini_batches
        self.value_loss_coef, self.entropy_coef, self.gamma, self.lam = value_loss_coef, entropy_coef, gamma, lam
        self.max_grad_norm, self.use_clipped_value_loss = max_grad_norm, use_clipped_value_loss
        self.desired_kl, self.schedule = desired_kl, schedule
        self.normalize_advantage_per_mini_batch = normalize_advantage_per_mini_batch
        self.betas, self.eps = (0.9, 0.999), 1e-8
        # device-side optimiser state: [lr, step, beta1^t, beta2^t, clip coef, lr/bias1, sqrt(bias2), -]
        self._adam = torch.tensor([float(learning_rate), 0.0, 1.0, 1.0, 1.0, 0.0, 1.0, 0.0], device=self.device)
        self._stats = torch.zeros(5, device=self.device)  # running sums: value, surrogate, entropy, kl, count
        # imx_ppo_loss_fwd writes {surrogate, value loss, entropy, KL, loss, -, -, -} straight behind the gradients: the KL
        # estimate rides in the gradient bucket's all-reduce without a copy
        self._out8 = self.bucket.grad[self.bucket.numel:self.bucket.numel + 8]
        self._kl = self._out8[3:4]
        self._norm_scratch = torch.zeros(int(lib().imx_adam_norm_scratch_bytes(self.bucket.numel)), dtype=torch.uint8, device=self.device) \
            if self.device.type == "cuda" else None
        self._actor_layers = _mlp_layers(self.policy.actor)
        self._critic_layers = _mlp_layers(self.policy.critic)
        self._ws: dict = {}
        self._side = None
        if self.device.type == "cuda" and os.getenv("IMX_DW_CU_BUDGET"):  # experiments (tools/): sample splits of imx_mlp_dw over fewer CUs
            lib().imx_mlp_set_dw_cu_budget(int(os.environ["IMX_DW_CU_BUDGET"]))
        self.two_streams = bool(two_streams)
        if self.device.type == "cuda":
            # HIP binds a stream to one of a few hardware queues at its FIRST use, round-robin: touch the update's streams
            # now, in a fixed order, so that main / side / aux never end up sharing a queue depending on what ran before
            # (measured: update 19.2 ms when the side stream was first used after the rollout graph, 18.3 ms otherwise)
            for st in (self._side_stream(), self._aux_stream()):
                if st is not None:
                    with torch.cuda.stream(st):
                        torch.zeros(1, device=self.device)
            torch.cuda.synchronize(self.device)

    @property
    def learning_rate(self) -> float:
        return float(self._adam[0].item())

    def optimizer_state_dict(self) -> dict:
        """The layout of ``torch.optim.Adam(policy.parameters()).state_dict()`` -- what upstream's runner saves and loads
        (``state`` by parameter index in ``policy.parameters()`` order: step, exp_avg, exp_avg_sq; one ``param_groups`` entry) --
        plus ``imx_adam_state``: the 8 device-side scalars (lr, step, beta powers ...) for an exact resume here."""
        base = self.bucket.flat.data_ptr()
        state, params = {}, []
        step = self._adam[1:2].clone().reshape(())
        for i, p in enumerate(q for q in self.policy.parameters() if q.requires_grad):
            off = (p.data_ptr() - base) // 4
            params.append(i)
            if float(step) > 0:
                state[i] = {"step": step.clone(), "exp_avg": self.bucket.exp_avg[off:off + p.numel()].view_as(p).clone(),
                            "exp_avg_sq": self.bucket.exp_avg_sq[off:off + p.numel()].view_as(p).clone()}
        group = {"lr": self.learning_rate, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": 0, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "decoupled_weight_decay": False, "params": params}
        return {"state": state, "param_groups": [group], "imx_adam_state": self._adam.clone()}

    def load_optimizer_state_dict(self, o: dict) -> None:
        """Accepts a torch Adam state dict (upstream checkpoints, and the ones written here) or the name-keyed layout of round 1."""
        base = self.bucket.flat.data_ptr()
        if "state" in o and "param_groups" in o:
            plist = [q for q in self.policy.parameters() if q.requires_grad]
            if len(o["param_groups"]) != 1 or len(o["param_groups"][0]["params"]) != len(plist):
                raise ValueError("optimizer state dict does not match the policy: "
                                 f"{len(o['param_groups'][0]['params'])} parameters in the checkpoint, {len(plist)} here")
            step = 0.0
            self.bucket.exp_avg.zero_()
            self.bucket.exp_avg_sq.zero_()
            for i, p in enumerate(plist):
                st = o["state"].get(i)
                if st is None:
                    continue
                if tuple(st["exp_avg"].shape) != tuple(p.shape):
                    raise ValueError(f"optimizer state of parameter {i}: shape {tuple(st['exp_avg'].shape)} vs {tuple(p.shape)}")
                off = (p.data_ptr() - base) // 4
                self.bucket.exp_avg[off:off + p.numel()].copy_(st["exp_avg"].reshape(-1))
                self.bucket.exp_avg_sq[off:off + p.numel()].copy_(st["exp_avg_sq"].reshape(-1))
                step = float(st["step"])
            if "imx_adam_state" in o:
                self._adam.copy_(o["imx_adam_state"])
            else:  # an upstream checkpoint: rebuild the device scalars from lr / step / betas
                g = o["param_groups"][0]
                b1, b2 = g.get("betas", self.betas)
                self._adam.copy_(torch.tensor([float(g["lr"]), step, b1 ** step, b2 ** step, 1.0, 0.0, 1.0, 0.0]))
            return
        if isinstance(o["exp_avg"], dict):
            for name, p in self.policy.named_parameters():
                if p.requires_grad and name in o["exp_avg"]:
                    off = (p.data_ptr() - base) // 4
                    self.bucket.exp_avg[off:off + p.numel()].copy_(o["exp_avg"][name].reshape(-1))
                    self.bucket.exp_avg_sq[off:off + p.numel()].copy_(o["exp_avg_sq"][name].reshape(-1))
        else:  # flat tensors in bucket order (checkpoints written before the moments were keyed by name)
            self.bucket.exp_avg.copy_(o["exp_avg"])
            self.bucket.exp_avg_sq.copy_(o["exp_avg_sq"])
        self._adam.copy_(o["adam_state"])

    def _first_layer_pair(self):
        """(actor Linear 0, critic Linear 0, ELU alpha) when the two first layers have the same shape and an ELU behind them."""
        try:
            (la, aa), (lc, ac) = _mlp_layers(self.policy.actor)[0], _mlp_layers(self.policy.critic)[0]
        except (IndexError, AttributeError, TypeError):
            return None
        if len(_mlp_layers(self.policy.actor)) < 2 or len(_mlp_layers(self.policy.critic)) < 2:
            return None
        same = (la.in_features == lc.in_features and la.out_features == lc.out_features and la.bias is not None and lc.bias is not None
                and isinstance(aa, nn.ELU) and isinstance(ac, nn.ELU) and aa.alpha == ac.alpha)
        return (la, lc, aa.alpha) if same else None

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

    _critic_infer = None

    def _evaluate(self, critic_obs):
        """``policy.evaluate`` for the bootstrap values of ``compute_returns``: the critic's whole stack in one launch (``imx_mlp_infer``,
        one network) where it applies, the module otherwise.  Same arithmetic as the rollout's value estimates."""
        if self.device.type == "cuda" and critic_obs.is_contiguous() and critic_obs.dtype == torch.float32:
            if self._critic_infer is None:
                self._critic_infer = FusedInference(self._critic_layers)
                self._last_values = torch.empty(critic_obs.shape[0], 1, device=self.device)
            if self._critic_infer.ok and self._last_values.shape[0] == critic_obs.shape[0]:
                self._critic_infer.refresh()
                self._critic_infer(critic_obs, self._last_values)
                return self._last_values
        return self.policy.evaluate(critic_obs).detach()

    def compute_returns(self, last_critic_obs):
        last_values = self._evaluate(last_critic_obs)
        self.storage.compute_returns(last_values, self.gamma, self.lam,
                                     normalize_advantage=not self.normalize_advantage_per_mini_batch)

    # ---- multi-GPU (one process per GPU, RCCL over xGMI) ---------------------------------------------------
    def broadcast_parameters(self):
        dist.broadcast(self.bucket.flat, src=0)

    time_allreduce = False  # bench.py: HIP events around every bucket all-reduce (device time on the update's stream, incl. the wait for peers)
    _allreduce_events: list = []

    def reduce_parameters(self):
        """Mean of the flat gradient bucket (+ KL slot) over ranks: ONE all-reduce."""
        if self.time_allreduce and self.device.type == "cuda":
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            allreduce_mean_(self.bucket.grad, self.gpu_world_size)
            e1.record()
            self._allreduce_events.append((e0, e1))
            return
        allreduce_mean_(self.bucket.grad, self.gpu_world_size)

    def allreduce_us(self) -> dict | None:
        """Mean / max device time of the timed bucket all-reduces (None when none was timed); clears the record."""
        ev, self._allreduce_events = self._allreduce_events, []
        if not ev:
            return None
        ev[-1][1].synchronize()
        t = [a.elapsed_time(b) * 1e3 for a, b in ev]
        return {"mean": sum(t) / len(t), "max": max(t), "count": len(t)}

    # ---- update ----------------------------------------------------------------------------------------------------
    def _side_stream(self):
        if not self.two_streams or self.device.type != "cuda":
            return None
        if self._side is None:
            self._side = torch.cuda.Stream(self.device)
        return self._side

    def _aux_stream(self):
        if getattr(self, "_aux", None) is None:
            self._aux = torch.cuda.Stream(self.device)
        return self._aux

    def _workspace(self, M: int, A: int):
        ws = self._ws.get((M, A))
        if ws is None:
            dev = self.device
            ws = dict(dmu=torch.empty(M, A, device=dev), dsigma=torch.empty(M, A, device=dev), dvalue=torch.empty(M, 1, device=dev),
                      scratch=torch.empty(int(lib().imx_ppo_scratch_bytes(M)), dtype=torch.uint8, device=dev),
                      colsum=torch.empty(int(lib().imx_colsum_scratch_bytes()), dtype=torch.uint8, device=dev),
                      mlp_a=DeferredReductions(self._actor_layers, M, dev), mlp_c=DeferredReductions(self._critic_layers, M, dev))
            self._ws[(M, A)] = ws
        return ws

    def minibatch_step(self, obs, critic_obs, actions, target_values, advantages, returns, old_logp, old_mu, old_sigma):
        """Forward, loss, explicit backward into the flat gradient bucket (no optimiser step)."""
        L = lib()
        pol = self.policy
        stream = _lib.current_stream(self.device)
        M, A = actions.shape
        for name, t in (("actions", actions), ("target_values", target_values), ("advantages", advantages), ("returns", returns),
                        ("old_actions_log_prob", old_logp), ("old_mu", old_mu), ("old_sigma", old_sigma)):
            if not t.is_contiguous():  # the loss kernels address these densely (observations alone may carry a row pitch)
                raise _lib.ImxError(f"PPO.minibatch_step: {name} must be contiguous (shape {tuple(t.shape)}, strides {t.stride()})")
        ws = self._workspace(M, A)
        if pol.noise_std_type == "scalar":
            sigma, sstride = pol.std, 0
        else:
            sigma, sstride = None, A  # expanded below, once mu exists
        clipf, vclip = float(self.clip_param), int(self.use_clipped_value_loss)
        vcoef, ecoef = float(self.value_loss_coef), float(self.entropy_coef)

        heads = {"a": None, "c": None}  # gradient below the output layer when imx_mlp_head_fwd_bwd already ran its backward

        def actor_pass(st):
            nonlocal sigma
            hl = None
            if sigma is not None:  # shared std: the head kernel computes the policy gradient from mu in the same launch
                hl = {"loss": _lib.ImxHeadLoss(mode=1, sigma_stride=0, use_clipped_value_loss=vclip, clip_param=clipf, value_loss_coef=vcoef,
                                               entropy_coef=ecoef, grad_scale=1.0, sigma_d=sigma.data_ptr(), actions_d=actions.data_ptr(),
                                               old_logp_d=old_logp.data_ptr(), advantages_d=advantages.data_ptr(),
                                               dmu_d=ws["dmu"].data_ptr(), dsigma_d=ws["dsigma"].data_ptr())}
            if hl is not None:
                hl["bwd"] = _head_bwd_args(ws["mlp_a"], "a")
            mu, saved_a = mlp_forward(self._actor_layers, obs, head_loss=hl, first=first_a)
            heads["a"] = hl.get("dprev") if hl is not None else None
            if sigma is None:
                sigma = torch.exp(pol.log_std).expand_as(mu).contiguous()
            if hl is None or not hl.get("applied"):
                check(L.imx_ppo_loss_bwd(M, A, mu.data_ptr(), sigma.data_ptr(), sstride, actions.data_ptr(), old_logp.data_ptr(),
                                         advantages.data_ptr(), None, None, None, clipf, vclip, vcoef, ecoef, 1.0,
                                         ws["dmu"].data_ptr(), ws["dsigma"].data_ptr(), None, st))
            return mu, saved_a

        def critic_pass(st):
            hl = {"loss": _lib.ImxHeadLoss(mode=2, sigma_stride=0, use_clipped_value_loss=vclip, clip_param=clipf, value_loss_coef=vcoef,
                                           entropy_coef=ecoef, grad_scale=1.0, returns_d=returns.data_ptr(),
                                           old_values_d=target_values.data_ptr(), dvalue_d=ws["dvalue"].data_ptr())}
            hl["bwd"] = _head_bwd_args(ws["mlp_c"], "c")
            value, saved_c = mlp_forward(self._critic_layers, critic_obs, head_loss=hl, first=first_c)
            heads["c"] = hl.get("dprev")
            if not hl.get("applied"):
                check(L.imx_ppo_loss_bwd(M, A, None, None, sstride, None, None, None, returns.data_ptr(), value.data_ptr(),
                                         target_values.data_ptr(), clipf, vclip, vcoef, ecoef, 1.0, None, None, ws["dvalue"].data_ptr(), st))
            return value, saved_c

        def loss_values(mu, value, st):  # logging / adaptive-LR inputs and the sigma gradient: off the critical path
            check(L.imx_ppo_loss_fwd(M, A, mu.data_ptr(), sigma.data_ptr(), sstride, actions.data_ptr(), old_logp.data_ptr(),
                                     old_mu.data_ptr(), old_sigma.data_ptr(), advantages.data_ptr(), returns.data_ptr(),
                                     value.data_ptr(), target_values.data_ptr(), clipf, vclip, vcoef, ecoef,
                                     self._out8.data_ptr(), self._stats.data_ptr(), ws["scratch"].data_ptr(), st))
            # std.grad = dsigma.sum(0) (log-std: (dsigma * sigma).sum(0)); torch.sum on a (24576, 37) tensor is a 320 us launch
            if pol.noise_std_type == "scalar" and A <= 64:
                check(L.imx_colsum(M, A, ws["dsigma"].data_ptr(), None, pol.std.grad.data_ptr(), ws["colsum"].data_ptr(), st))
            elif A <= 64:
                check(L.imx_colsum(M, A, ws["dsigma"].data_ptr(), sigma.data_ptr(), pol.log_std.grad.data_ptr(), ws["colsum"].data_ptr(), st))
            elif pol.noise_std_type == "scalar":
                torch.sum(ws["dsigma"], dim=0, out=pol.std.grad)
            else:
                torch.sum(ws["dsigma"] * sigma, dim=0, out=pol.log_std.grad)

        # The policy gradient needs only mu, the value gradient only the critic's output: actor and critic run
        # forward -> loss gradient -> backward on two streams and meet once, at the end of the minibatch.  The loss
        # VALUES (logging, KL for the adaptive LR) need both heads and run behind the critic's backward pass.
        # Same observations for both networks (no privileged group) and same first-layer shape: ONE stacked GEMM [W_a; W_c] and one
        # ELU over (M, 2H) -- the observations are read once, one launch each instead of two; the halves are strided views
        first_a = first_c = z0 = None
        joint_elu = None
        if self._joint0 is not None and critic_obs.data_ptr() == obs.data_ptr() and critic_obs.shape == obs.shape:
            w0, b0, H0, alpha0 = self._joint0
            if fused_first_layer_ok(obs, w0):
                # bias + ELU in the epilogue of the layer's own MFMA kernel (imx_mlp_fwd_elu): the 100 MB output is written once instead of
                # written, read and written again by a separate activation pass (library GEMM 106 us + ELU 35 us per minibatch before)
                z0 = torch.empty(obs.shape[0], 2 * H0, device=obs.device)
                check(L.imx_mlp_fwd_elu(obs.shape[0], 2 * H0, obs.shape[1], obs.data_ptr(), obs.stride(0), w0.data_ptr(), b0.data_ptr(),
                                        alpha0, 1, z0.data_ptr(), z0.stride(0), stream))
            else:
                z0 = torch.addmm(b0, obs, w0.t())
                joint_elu = alpha0  # the ELU of each half is applied on ITS network's stream, after the fork
            first_a, first_c = z0[:, :H0], z0[:, H0:]
        side = self._side_stream()
        main = torch.cuda.current_stream(self.device)
        if side is not None:
            # Issue order matters as much as the streams: the host needs ~8 us per launch, so the two networks are fed
            # alternately (forward, forward, backward, backward) -- enqueueing one network's ~30 launches first would
            # leave the other stream empty for the first 200 us of every minibatch.
            # the loss VALUES ride at the end of the critic's stream (its backward chain is the shorter one) rather than on a third
            # stream: one fork and one join fewer per minibatch, each a cross-queue release / acquire (16.28 -> 16.15 ms per update)
            loss_on_side = LOSS_ON_SIDE
            aux = self._aux_stream()
            side.wait_stream(main)
            if joint_elu is not None:
                F.elu(first_a, alpha=joint_elu, inplace=True)
            mu, saved_a = actor_pass(stream)
            mu_ready = torch.cuda.Event()
            mu_ready.record(main)
            with torch.cuda.stream(side):
                if joint_elu is not None:
                    F.elu(first_c, alpha=joint_elu, inplace=True)
                value, saved_c = critic_pass(side.cuda_stream)
                value_ready = torch.cuda.Event()
                value_ready.record(side)
            if not loss_on_side:
                aux.wait_stream(main)
                aux.wait_event(value_ready)
            mlp_backward(self._actor_layers, saved_a, ws["dmu"], ws["mlp_a"], head_dprev=heads["a"])
            with torch.cuda.stream(side):
                mlp_backward(self._critic_layers, saved_c, ws["dvalue"], ws["mlp_c"], head_dprev=heads["c"])
                if loss_on_side:
                    side.wait_event(mu_ready)
                    loss_values(mu, value, side.cuda_stream)
            if not loss_on_side:
                with torch.cuda.stream(aux):
                    loss_values(mu, value, aux.cuda_stream)
            main.wait_stream(side)
            if not loss_on_side:
                main.wait_stream(aux)
            mu.record_stream(aux)
            mu.record_stream(side)
            value.record_stream(aux)
            value.record_stream(main)
            if z0 is not None:
                z0.record_stream(side)
        else:
            if joint_elu is not None:
                F.elu(z0, alpha=joint_elu, inplace=True)
            mu, saved_a = actor_pass(stream)
            value, saved_c = critic_pass(stream)
            loss_values(mu, value, stream)
            mlp_backward(self._actor_layers, saved_a, ws["dmu"], ws["mlp_a"], head_dprev=heads["a"])
            mlp_backward(self._critic_layers, saved_c, ws["dvalue"], ws["mlp_c"], head_dprev=heads["c"])
        return self._out8

    update_graph = False  # set by the runner (use_graph=True): the update may run as one hipGraph replay, see update()
    _update_calls = 0
    _update_g = None
    _update_t = None  # [eager ms, graph ms] while the choice is open; "eager" / "graph" once it is made

    @torch.no_grad()
    def update(self):
        """5 epochs x 4 minibatches of forward / loss / backward / (all-reduce) / Adam.  With ``update_graph`` the ~1200 launches of
        an update on three streams -- no host decision anywhere: permutation, losses, KL schedule, grad norm and Adam scalars all
        live on the device -- are captured into a hipGraph; the captured ``torch.randperm`` draws a fresh permutation on every replay
        (graph-registered generator) and a replay is bit-identical to the eager update (tests/test_kernels_gpu.py).  Whether the
        replay is FASTER depends on the network: it removes the host's launch cost (Isaac-Velocity-Flat-Anymal-C-v0: 8.6 -> 6.2 ms,
        Rough-Anymal-C 17.7 -> 17.3 ms) but the runtime's own placement of the three branches can lose to the hand-ordered eager
        issue (Rough-G1: 21.7 -> 24.4 ms).  So it is measured, once: call 1 eager (allocations, GEMM tuning), call 2 eager between
        two events, call 3 capture + replay, call 4 replay between two events, and from call 5 on the faster of the two."""
        if not (self.update_graph and self.device.type == "cuda" and not self.is_multi_gpu) or self._update_t == "eager":
            return self._update_eager()
        self._update_calls += 1
        c = self._update_calls
        if c == 1:
            return self._update_eager()
        if c == 2:
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            self._update_ev = ev
            ev[0].record()
            out = self._update_eager()
            ev[1].record()
            return out
        if c == 5 and self._update_t is None and os.getenv("IMX_UPDATE_GRAPH") == "force":
            self._update_t = "graph"  # A/B runs (tools/): keep the replay whatever the measurement says
        if c == 5 and self._update_t is None:
            ev = self._update_ev
            ev[3].synchronize()
            t_eager, t_graph = ev[0].elapsed_time(ev[1]), ev[2].elapsed_time(ev[3])
            self._update_times_ms = (t_eager, t_graph)
            self._update_t = "graph" if t_graph < t_eager else "eager"
            if self._update_t == "eager":
                self._update_g = None
                return self._update_eager()
        # inference mode: the generator's graph-safe seed / offset tensors may have been created under it (the runner captures the
        # rollout graph in inference mode) and capture_begin / replay update them in place
        with torch.inference_mode():
            if self._update_g is None:
                torch.cuda.synchronize(self.device)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._update_eager()
                self._update_g = g
            if c == 4:
                self._update_ev[2].record()
            self._update_g.replay()
            if c == 4:
                self._update_ev[3].record()
        self.storage.clear()
        return self._stats

    def _update_eager(self):
        b = self.bucket
        L = lib()
        stream = _lib.current_stream(self.device)
        self._stats.zero_()
        adaptive = self.desired_kl is not None and self.schedule == "adaptive"
        side = self._side_stream()
        copy_stream = self._aux_stream() if side is not None else None
        main = torch.cuda.current_stream(self.device) if side is not None else None
        gen = iter(self.storage.mini_batch_generator(self.num_mini_batches, self.num_learning_epochs, copy_stream=copy_stream))
        item = next(gen, None)
        while item is not None:
            if copy_stream is not None:
                batch, ready = item
                main.wait_event(ready)
            else:
                batch = item
            if self.normalize_advantage_per_mini_batch:
                batch = list(batch)
                adv = batch[4]
                batch[4] = (adv - adv.mean()) / (adv.std() + 1e-8)
            self.minibatch_step(*batch)
            # the next minibatch's gather (HBM-bound) is issued now, on the third stream: it starts when the backward pass
            # has finished with the buffers and runs beside the all-reduce / norm / Adam kernels below
            item = next(gen, None)
            if self.is_multi_gpu:
                self.reduce_parameters()
            # clip_grad_norm_ + adaptive-KL learning rate + Adam: two launches (norm reduction with the schedule in its
            # last block, then the parameter update), all scalars on the device
            check(L.imx_adam_update_norm(b.numel, b.flat.data_ptr(), b.grad.data_ptr(), b.exp_avg.data_ptr(), b.exp_avg_sq.data_ptr(),
                                         self._adam.data_ptr(), self._kl.data_ptr() if adaptive else None,
                                         float(self.desired_kl or 0.0), float(self.max_grad_norm or 0.0), self.betas[0],
                                         self.betas[1], self.eps, self._norm_scratch.data_ptr(), self._norm_scratch.numel(), stream))
        if copy_stream is not None:
            main.wait_stream(copy_stream)
        self.storage.clear()
        return self._stats  # device tensor; .tolist() only when the caller wants to log

    def loss_dict(self) -> dict:
        s = self._stats.tolist()
        n = max(s[4], 1.0)
        return {"value_function": s[0] / n, "surrogate": s[1] / n, "entropy": s[2] / n, "kl": s[3] / n}
