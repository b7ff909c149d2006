"""GPU: standalone kernels through the C ABI -- root frame, ray-cast, GAE, PPO loss."""

import ctypes
import os

import numpy as np
import pytest
import torch

from _util import FLOAT_TOL, GOLDEN, assert_close

pytestmark = pytest.mark.gpu


def test_root_frame_matches_reference_math(libimx):
    from isaaclab_amd import _lib

    z = np.load(os.path.join(GOLDEN, "math.npz"))
    q, v = torch.from_numpy(z["q"]).cuda(), torch.from_numpy(z["v"]).cuda()
    out = torch.empty_like(v)
    g = torch.empty_like(v)
    _lib.check(libimx.imx_root_frame(q.shape[0], q.data_ptr(), v.data_ptr(), None, 0.0, 0.0, -1.0, out.data_ptr(), None,
                                     g.data_ptr(), torch.cuda.current_stream().cuda_stream))
    assert_close(out, torch.from_numpy(z["quat_rotate_inverse"]), 1e-6, "quat_rotate_inverse")


def test_raycast_matches_brute_force_oracles():
    from isaaclab_amd.env import TerrainMesh
    from isaaclab_amd.terrain import make_rough_terrain
    from oracle.raycast import raycast_f64, raycast_woop_f32

    v, t, ext = make_rough_terrain(2, 3, tile=4.0, border=2.0, seed=21)
    rng = np.random.default_rng(0)
    R = 20000
    starts = np.stack([rng.uniform(-6.2, 6.2, R), rng.uniform(-8.2, 8.2, R), rng.uniform(19, 21, R)], 1).astype(np.float32)
    # lattice-aligned rays too: exactly on vertices / edges of the height field
    starts[:2000, 0] = np.round(starts[:2000, 0] * 10) / 10
    starts[1000:3000, 1] = np.round(starts[1000:3000, 1] * 10) / 10
    dirs = np.tile(np.array([0, 0, -1], np.float32), (R, 1))
    for cell in (0.1, 0.0, 0.37):
        mesh = TerrainMesh(v, t, cell)
        hits, dist, _, face = mesh.raycast(torch.from_numpy(starts).cuda(), torch.from_numpy(dirs).cuda(), 1e6, True, True)
        h32, t32, f32 = raycast_woop_f32(v, t, starts, dirs)
        h64, t64, f64 = raycast_f64(v, t, starts, dirs)
        hits, dist, face = hits.cpu().numpy(), dist.cpu().numpy(), face.cpu().numpy()
        miss = ~np.isfinite(t32)
        assert np.array_equal(~np.isfinite(dist), miss), f"cell={cell}: miss masks differ"
        assert miss.sum() < R // 10
        # same fp32 arithmetic as the Woop oracle: bit-exact distances; the fp64 geometric truth within 1e-5
        assert np.array_equal(dist[~miss], t32[~miss]), f"cell={cell}"
        assert np.abs(hits[~miss] - h64[~miss]).max() <= 1e-5
        assert mesh.num_triangles == len(t)
    # upward rays (no top-sorted early exit on this side): from below the terrain into box bottoms / height-field quads
    up_s = starts[:4000].copy()
    up_s[:, 2] = -5.0
    up_d = np.tile(np.array([0, 0, 1], np.float32), (4000, 1))
    _, dist, _, _ = mesh.raycast(torch.from_numpy(up_s).cuda(), torch.from_numpy(up_d).cuda(), 1e6, True, True)
    _, t32, _ = raycast_woop_f32(v, t, up_s, up_d)
    dist = dist.cpu().numpy()
    assert np.array_equal(np.isfinite(dist), np.isfinite(t32)) and np.isfinite(t32).sum() > 3000
    assert np.array_equal(dist[np.isfinite(t32)], t32[np.isfinite(t32)])
    # general (slanted) rays through the DDA path
    R2 = 4000
    s2 = np.stack([rng.uniform(-5, 5, R2), rng.uniform(-7, 7, R2), rng.uniform(0.5, 3, R2)], 1).astype(np.float32)
    d2 = rng.normal(size=(R2, 3)).astype(np.float32)
    d2[:, 2] = -np.abs(d2[:, 2]) * 0.3
    d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
    mesh = TerrainMesh(v, t, 0.1)
    hits, dist, _, face = mesh.raycast(torch.from_numpy(s2).cuda(), torch.from_numpy(d2).cuda(), 50.0, True, True)
    dist, face = dist.cpu().numpy(), face.cpu().numpy()
    # (1) against the brute force in the SAME fp32 Woop arithmetic: the grid walk must find exactly the triangle the exhaustive search
    #     finds -- every disagreement is enumerated, none is tolerated
    h32, t32, f32 = raycast_woop_f32(v, t, s2, d2, 50.0)
    disagree = np.flatnonzero((np.isfinite(dist) != np.isfinite(t32)) | (np.isfinite(t32) & (dist != t32)))
    assert disagree.size == 0, [(int(i), float(dist[i]), float(t32[i]), int(face[i]), int(f32[i])) for i in disagree[:10]]
    assert np.isfinite(t32).mean() > 0.3  # shallow rays leave the small mesh; 1400+ hits
    # (2) against the fp64 geometric truth: fp32 edge functions lose the hit point along the surface by ~eps * |coordinates|, which a ray
    #     meeting the surface at grazing angle turns into eps * |coordinates| / cos(incidence) along the ray.  The 1e-5 contract of the
    #     height scanner (vertical rays on a terrain) is met by every ray that is not grazing; for the others the bound scales with
    #     1 / cos, and hit / miss may flip only for rays that pass an edge or the max_dist sphere within that distance.
    h64, t64, f64 = raycast_f64(v, t, s2, d2, 50.0)
    both = np.isfinite(dist) & np.isfinite(t64)
    tri = v[t[f64[both]]]
    n = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]).astype(np.float64)
    cosi = np.abs((n * d2[both]).sum(1)) / np.maximum(np.linalg.norm(n, axis=1), 1e-30)
    err = np.abs(dist[both] - t64[both])
    steep = cosi > 0.2
    rel = err / np.maximum(t64[both], 1.0)
    print("DDA vs fp64: steep rays %d, max rel err steep %.2e, max err*cos %.2e, hit/miss flips %d" % (
        int(steep.sum()), float(rel[steep].max()), float((err * cosi).max()), int((np.isfinite(dist) != np.isfinite(t64)).sum())))
    assert steep.mean() > 0.5 and rel[steep].max() <= 1e-5
    assert (err * np.maximum(cosi, 1e-3)).max() <= 1e-5  # measured 1.4e-6: the along-ray error is <= 1e-5 / cos(incidence)
    flips = np.flatnonzero(np.isfinite(dist) != np.isfinite(t64))
    assert flips.size == 0, [(int(i), float(dist[i]), float(t64[i])) for i in flips]  # no hit / miss disagreement with the fp64 truth
    # empty input
    e = torch.empty(0, 3, device="cuda")
    assert mesh.raycast(e, e)[0].shape == (0, 3)


@pytest.mark.parametrize("T,N", [(24, 4096), (16, 64), (5, 1), (24, 100_001), (40, 300), (32, 16_384), (24, 16_385)])
def test_gae_matches_rsl_rl_restatement(libimx, T, N):
    """(T <= 32 and N <= 64 x #CUs: the single-launch kernel with the grid barrier; otherwise scan + normalisation kernels.)"""
    from isaaclab_amd.rsl_rl.storage import gae_returns
    from oracle.rsl_rl_oracle import compute_returns

    g = torch.Generator().manual_seed(T * 1000 + N)
    rew = torch.randn(T, N, 1, generator=g)
    val = torch.randn(T, N, 1, generator=g)
    dones = (torch.rand(T, N, 1, generator=g) < 0.05).to(torch.uint8)
    last = torch.randn(N, 1, generator=g)
    for norm in (True, False):
        if N * T == 5 and norm:
            pass
        ret0, adv0 = compute_returns(rew, val, dones, last, 0.99, 0.95, norm)
        ret1, adv1 = gae_returns(rew.cuda(), val.cuda(), dones.cuda(), last.cuda(), 0.99, 0.95, norm)
        assert_close(ret1, ret0, FLOAT_TOL, "returns")
        assert_close(adv1, adv0, FLOAT_TOL, "advantages")
    # property: with gamma*lam = 0 the advantage is the one-step TD error
    ret, adv = gae_returns(rew.cuda(), val.cuda(), dones.cuda(), last.cuda(), 0.9, 0.0, False)
    nv = torch.cat([val[1:], last.unsqueeze(0)], 0)
    td = rew + (1 - dones.float()) * 0.9 * nv - val
    assert_close(adv, td, FLOAT_TOL, "TD(0)")
    # the same scratch block over many calls (the storage keeps one): the grid-barrier counters re-arm, results repeat bit for bit
    scr = torch.zeros(int(libimx.imx_gae_scratch_bytes(T, N)), dtype=torch.uint8, device="cuda")
    args = (rew.cuda(), val.cuda(), dones.cuda(), last.cuda(), 0.99, 0.95, True)
    ret_a, adv_a = (x.clone() for x in gae_returns(*args, scratch=scr))
    for _ in range(5):
        ret_b, adv_b = gae_returns(*args, scratch=scr)
        assert torch.equal(ret_a, ret_b) and torch.equal(adv_a, adv_b)


@pytest.mark.parametrize("M,A", [(24576, 12), (1000, 37), (7, 1)])
def test_ppo_loss_matches_autograd(libimx, M, A):
    from isaaclab_amd.rsl_rl.ppo import fused_ppo_loss
    from oracle.rsl_rl_oracle import ppo_losses

    g = torch.Generator().manual_seed(M + A)
    mu = torch.randn(M, A, generator=g)
    sigma = torch.rand(M, A, generator=g) * 0.8 + 0.3
    act = mu + sigma * torch.randn(M, A, generator=g)
    old_mu = mu + 0.1 * torch.randn(M, A, generator=g)
    old_sigma = sigma * (1 + 0.1 * torch.randn(M, A, generator=g)).clamp(0.5, 1.5)
    old_logp = torch.distributions.Normal(old_mu, old_sigma).log_prob(act).sum(-1, keepdim=True)
    adv, ret, val, old_val = (torch.randn(M, 1, generator=g) for _ in range(4))
    for clipped in (True, False):
        mu_c, sg_c, v_c = (x.clone().requires_grad_(True) for x in (mu, sigma, val))
        s, v, e, kl = ppo_losses(mu_c, sg_c, act, old_logp, old_mu, old_sigma, adv, ret, v_c, old_val, 0.2, clipped)
        loss = s + 1.0 * v - 0.005 * e
        loss.backward()
        mu_g, sg_g, v_g = (x.clone().cuda().requires_grad_(True) for x in (mu, sigma, val))
        loss_g, stats = fused_ppo_loss(mu_g, sg_g, act.cuda(), old_logp.cuda(), old_mu.cuda(), old_sigma.cuda(), adv.cuda(),
                                       ret.cuda(), v_g, old_val.cuda(), 0.2, clipped, 1.0, 0.005)
        loss_g.backward()
        assert_close(stats[:4], torch.stack([s, v, e, kl]).detach(), FLOAT_TOL, "loss terms")
        assert_close(loss_g.detach(), loss.detach(), FLOAT_TOL, "loss")
        assert_close(mu_g.grad * M, mu_c.grad * M, 1e-4, "dmu")
        assert_close(sg_g.grad * M, sg_c.grad * M, 1e-4, "dsigma")
        assert_close(v_g.grad * M, v_c.grad * M, 1e-4, "dvalue")


def _tiny_ppo(device="cuda:0", D=37, A=5, hidden=(64, 32)):
    from isaaclab_amd.rsl_rl.actor_critic import ActorCritic
    from isaaclab_amd.rsl_rl.ppo import PPO

    torch.manual_seed(3)
    pol = ActorCritic(D, D, A, actor_hidden_dims=list(hidden), critic_hidden_dims=list(hidden), init_noise_std=0.7)
    return PPO(pol, num_learning_epochs=1, num_mini_batches=1, schedule="adaptive", desired_kl=0.01, learning_rate=1e-3,
               entropy_coef=0.005, max_grad_norm=1.0, device=device)


def test_explicit_backward_matches_autograd(libimx):
    """PPO.minibatch_step (hand-written backward into the flat bucket) vs torch autograd through the same modules."""
    from oracle.rsl_rl_oracle import ppo_losses

    M, D, A = 1000, 37, 5
    alg = _tiny_ppo()
    g = torch.Generator().manual_seed(0)
    obs = torch.randn(M, D, generator=g).cuda()
    act = torch.randn(M, A, generator=g).cuda()
    old_mu = (0.3 * torch.randn(M, A, generator=g)).cuda()
    old_sigma = (torch.rand(M, A, generator=g) * 0.5 + 0.5).cuda()
    old_logp = torch.distributions.Normal(old_mu, old_sigma).log_prob(act).sum(-1, keepdim=True)
    adv, ret, old_val = (torch.randn(M, 1, generator=g).cuda() for _ in range(3))
    with torch.no_grad():
        out8 = alg.minibatch_step(obs, obs, act, old_val, adv, ret, old_logp, old_mu, old_sigma).clone()
    got = alg.bucket.grad.clone()
    # autograd reference on the same parameters
    pol = alg.policy
    alg.bucket.zero_grad()
    mu = pol.actor(obs)
    sigma = pol.std.expand_as(mu)
    val = pol.critic(obs)
    s, v, e, kl = ppo_losses(mu, sigma, act, old_logp, old_mu, old_sigma, adv, ret, val, old_val, 0.2, True)
    loss = s + 1.0 * v - 0.005 * e
    loss.backward()
    n = alg.bucket.numel
    assert_close(out8[:5], torch.stack([s, v, e, kl, loss]).detach(), 1e-5, "loss terms")
    ref = alg.bucket.grad[:n]
    err = (got[:n] - ref).abs().max() / ref.abs().max()
    assert float(err) < 1e-5, f"flat gradient bucket: max err / max |g| = {float(err):.2e}"  # fp32 GEMM summation order
    off = 0  # per parameter tensor: error relative to that tensor's largest gradient (summation-order noise scales with the
    for name, p in pol.named_parameters():  # sum of |terms|, not with each element's own magnitude)
        k = p.numel()
        e_p = float((got[off:off + k] - ref[off:off + k]).abs().max())
        assert e_p <= 1e-5 * max(float(ref[off:off + k].abs().max()), 1e-6), f"{name}: max err {e_p:.2e}"
        off += k
    assert abs(float(got[n + 3]) - float(kl)) < 1e-6  # the loss scalars (KL = slot 3) ride behind the gradients


@pytest.mark.parametrize("M,N,K", [(24576, 512, 235), (1000, 256, 512), (4099, 128, 256), (333, 130, 37), (5, 64, 48)])
def test_mlp_dw_matches_fp64(libimx, M, N, K):
    """imx_mlp_dw (split over samples on the f32 MFMA): dW = dY^T X, db = colsum(dY) against an fp64 product."""
    from isaaclab_amd import _lib

    g = torch.Generator().manual_seed(M + N)
    dY = torch.randn(M, N, generator=g).cuda()
    X = torch.randn(M, K, generator=g).cuda()
    dW = torch.full((N, K), float("nan"), device="cuda")
    db = torch.full((N,), float("nan"), device="cuda")
    nbytes = int(libimx.imx_mlp_scratch_bytes(M, N, K))
    scratch = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(libimx.imx_mlp_dw(M, N, K, dY.data_ptr(), N, X.data_ptr(), K, dW.data_ptr(), db.data_ptr(), scratch.data_ptr(), nbytes, st))
    ref = dY.double().t() @ X.double()
    scale = float((dY.double().abs().t() @ X.double().abs()).max())  # sum |a b|: the fp32 error scale of a dot product
    assert float((dW.double() - ref).abs().max()) <= 2e-6 * scale
    assert float((db.double() - dY.double().sum(0)).abs().max()) <= 2e-6 * float(dY.abs().sum(0).max())
    # same call again: bit-identical (fixed summation order), also with a strided (non-16-byte-pitch) input view
    dW2 = torch.empty_like(dW)
    _lib.check(libimx.imx_mlp_dw(M, N, K, dY.data_ptr(), N, X.data_ptr(), K, dW2.data_ptr(), None, scratch.data_ptr(), nbytes, st))
    assert torch.equal(dW, dW2)
    Xp = torch.zeros(M, K + 3, device="cuda")
    Xp[:, :K] = X
    _lib.check(libimx.imx_mlp_dw(M, N, K, dY.data_ptr(), N, Xp.data_ptr(), K + 3, dW2.data_ptr(), None, scratch.data_ptr(), nbytes, st))
    assert torch.equal(dW, dW2)
    assert libimx.imx_mlp_dw(M, N, K, dY.data_ptr(), N, X.data_ptr(), K, dW.data_ptr(), None, scratch.data_ptr(), 16, st) != 0


@pytest.mark.parametrize("M,N,K", [(24576, 512, 235), (4099, 128, 256), (333, 130, 37)])
def test_mlp_dw_elu_fuses_the_activation_backward(libimx, M, N, K):
    """imx_mlp_dw_elu == aten elu_backward followed by imx_mlp_dw, bit for bit (same arithmetic, same summation order)."""
    from isaaclab_amd import _lib

    g = torch.Generator().manual_seed(M + K)
    dH = torch.randn(M, N, generator=g).cuda()
    H = torch.nn.functional.elu(torch.randn(M, N, generator=g)).cuda()
    X = torch.randn(M, K, generator=g).cuda()
    nbytes = int(libimx.imx_mlp_scratch_bytes(M, N, K))
    scratch = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    dZ_ref = torch.ops.aten.elu_backward(dH, 1.0, 1.0, 1.0, True, H)
    dW0, db0 = torch.empty(N, K, device="cuda"), torch.empty(N, device="cuda")
    _lib.check(libimx.imx_mlp_dw(M, N, K, dZ_ref.data_ptr(), N, X.data_ptr(), K, dW0.data_ptr(), db0.data_ptr(), scratch.data_ptr(), nbytes, st))
    dW1, db1, dZ = torch.empty(N, K, device="cuda"), torch.empty(N, device="cuda"), torch.full((M, N), float("nan"), device="cuda")
    _lib.check(libimx.imx_mlp_dw_elu(M, N, K, dH.data_ptr(), N, H.data_ptr(), N, 1.0, dZ.data_ptr(), N, X.data_ptr(), K, dW1.data_ptr(),
                                     db1.data_ptr(), scratch.data_ptr(), nbytes, st))
    assert torch.equal(dZ, dZ_ref) and torch.equal(dW1, dW0) and torch.equal(db1, db0)
    # no layer below: dZ is not materialised
    _lib.check(libimx.imx_mlp_dw_elu(M, N, K, dH.data_ptr(), N, H.data_ptr(), N, 1.0, None, 0, X.data_ptr(), K, dW1.data_ptr(),
                                     db1.data_ptr(), scratch.data_ptr(), nbytes, st))
    assert torch.equal(dW1, dW0)
    assert libimx.imx_mlp_dw_elu(M, N, K, dH.data_ptr(), N, H.data_ptr(), N, 1.0, dH.data_ptr(), N, X.data_ptr(), K, dW1.data_ptr(),
                                 db1.data_ptr(), scratch.data_ptr(), nbytes, st) != 0  # in-place is refused


@pytest.mark.parametrize("M,D,hidden,A", [(4096, 235, (512, 256, 128), 12), (2048, 235, (512, 256, 128), 12),  # 32- and 16-sample tiles
                                          (100, 48, (128, 128, 128), 12), (33, 310, (512, 256, 128), 37),
                                          (5, 4, (32,), 1), (70, 37, (64, 32), 5)])
def test_mlp_infer_matches_torch(libimx, M, D, hidden, A):
    """imx_mlp_infer (both networks, all layers, one launch) against the nn.Sequential stacks in fp64."""
    from isaaclab_amd.rsl_rl.actor_critic import ActorCritic
    from isaaclab_amd.rsl_rl.ppo import FusedInference, _mlp_layers

    torch.manual_seed(M + D)
    pol = ActorCritic(D, D, A, actor_hidden_dims=list(hidden), critic_hidden_dims=list(hidden), init_noise_std=1.0).cuda()
    for p in pol.parameters():  # biases are zero-initialised: make them count
        if p.dim() == 1:
            p.data.normal_(0.0, 0.3)
    x = torch.randn(M, D, device="cuda")
    inf = FusedInference(_mlp_layers(pol.actor), _mlp_layers(pol.critic))
    assert inf.ok
    mu, val = torch.full((M, A), float("nan"), device="cuda"), torch.full((M, 1), float("nan"), device="cuda")
    inf(x, mu, val)
    with torch.no_grad():  # parameters change -> refresh() updates the padded copies
        for p in pol.parameters():
            p.mul_(1.5)
    inf.refresh()
    inf(x, mu, val)
    # the packed weight images (imx_mlp_pack_weights: the 32-sample kernel's loads made contiguous) change where the bytes come from,
    # nothing else: bit-identical to the row layout
    assert inf._wpk is not None
    packed, inf._wpk = inf._wpk, None
    mu_r, val_r = torch.full_like(mu, float("nan")), torch.full_like(val, float("nan"))
    inf(x, mu_r, val_r)
    inf._wpk = packed
    assert torch.equal(mu, mu_r) and torch.equal(val, val_r)
    ref_mu = pol.actor.double()(x.double())
    ref_v = pol.critic.double()(x.double())
    assert float((mu.double() - ref_mu).abs().max()) <= 1e-5 * max(1.0, float(ref_mu.abs().max()))
    assert float((val.double() - ref_v).abs().max()) <= 1e-5 * max(1.0, float(ref_v.abs().max()))
    wide = ActorCritic(D, D, A, actor_hidden_dims=[1024], critic_hidden_dims=[1024]).cuda()
    assert not FusedInference(_mlp_layers(wide.actor), _mlp_layers(wide.critic)).ok  # wider than 512: the library path is used


def test_deferred_reductions_match_immediate(libimx):
    """imx_reduce_batch_*: the partial sums of several layers flushed in one launch == the per-call reductions, bit for bit."""
    from isaaclab_amd import _lib

    M = 3000
    shapes = [(256, 512), (128, 256), (12, 128)]
    g = torch.Generator().manual_seed(4)
    st = torch.cuda.current_stream().cuda_stream
    data, ref = [], []
    for N, K in shapes:
        dY, X = torch.randn(M, N, generator=g).cuda(), torch.randn(M, K, generator=g).cuda()
        nb = int(libimx.imx_mlp_scratch_bytes(M, N, K))
        scr = torch.empty(nb, dtype=torch.uint8, device="cuda")
        dW, db = torch.empty(N, K, device="cuda"), torch.empty(N, device="cuda")
        _lib.check(libimx.imx_mlp_dw(M, N, K, dY.data_ptr(), N, X.data_ptr(), K, dW.data_ptr(), db.data_ptr(), scr.data_ptr(), nb, st))
        ref.append((dW.clone(), db.clone()))
        data.append((N, K, dY, X, scr, nb))
    h = ctypes.c_void_p()
    _lib.check(libimx.imx_reduce_batch_create(ctypes.byref(h)))
    _lib.check(libimx.imx_reduce_batch_begin(h))
    assert libimx.imx_reduce_batch_begin(h) != 0  # already open on this thread
    outs = []
    for N, K, dY, X, scr, nb in data:
        dW, db = torch.full((N, K), float("nan"), device="cuda"), torch.full((N,), float("nan"), device="cuda")
        _lib.check(libimx.imx_mlp_dw(M, N, K, dY.data_ptr(), N, X.data_ptr(), K, dW.data_ptr(), db.data_ptr(), scr.data_ptr(), nb, st))
        outs.append((dW, db))
    torch.cuda.synchronize()
    assert all(bool(torch.isnan(dW).all()) for dW, _ in outs)  # nothing reduced yet
    _lib.check(libimx.imx_reduce_batch_flush(h, st))
    for (dW, db), (rW, rb) in zip(outs, ref):
        assert torch.equal(dW, rW) and torch.equal(db, rb)
    assert libimx.imx_reduce_batch_flush(h, st) != 0  # not open any more
    libimx.imx_reduce_batch_destroy(h)


@pytest.mark.parametrize("M,K,A", [(24576, 128, 12), (24576, 128, 1), (1000, 256, 16), (37, 32, 5), (4099, 128, 37), (300, 64, 64)])
def test_mlp_head_matches_autograd(libimx, M, K, A):
    """imx_mlp_head_fwd / imx_mlp_head_bwd against torch (fp64) for the layer  y = ELU(z) W^T + b."""
    from isaaclab_amd import _lib

    g = torch.Generator().manual_seed(K + A)
    z = torch.randn(M, K, generator=g).cuda()
    W = (0.2 * torch.randn(A, K, generator=g)).cuda()
    b = torch.randn(A, generator=g).cuda()
    dY = torch.randn(M, A, generator=g).cuda()
    alpha = 1.0
    zd = z.double().requires_grad_(True)
    Wd, bd = W.double().requires_grad_(True), b.double().requires_grad_(True)
    hd = torch.nn.functional.elu(zd, alpha)
    yd = hd @ Wd.t() + bd
    yd.backward(dY.double())
    h = hd.detach().float()
    st = torch.cuda.current_stream().cuda_stream
    y = torch.empty(M, A, device="cuda")
    _lib.check(libimx.imx_mlp_head_fwd(M, K, A, h.data_ptr(), K, W.data_ptr(), b.data_ptr(), y.data_ptr(), 0, 0.0, st))
    assert float((y.double() - yd.detach()).abs().max()) <= 1e-5
    # ELU of the layer below applied on the way in, in place: z -> h (bit-identical to aten elu), same y
    zz = z.clone()
    y2 = torch.empty_like(y)
    _lib.check(libimx.imx_mlp_head_fwd(M, K, A, zz.data_ptr(), K, W.data_ptr(), b.data_ptr(), y2.data_ptr(), 1, alpha, st))
    assert float((zz - torch.nn.functional.elu(z, alpha)).abs().max()) <= 1e-6
    assert float((y2.double() - yd.detach()).abs().max()) <= 1e-5
    nbytes = int(libimx.imx_mlp_scratch_bytes(M, A, K))
    scratch = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    dprev, dW, db = torch.empty(M, K, device="cuda"), torch.empty(A, K, device="cuda"), torch.empty(A, device="cuda")
    _lib.check(libimx.imx_mlp_head_bwd(M, K, A, dY.data_ptr(), h.data_ptr(), K, W.data_ptr(), alpha, 1, dprev.data_ptr(), dW.data_ptr(),
                                       db.data_ptr(), scratch.data_ptr(), nbytes, st))
    assert float((dprev.double() - zd.grad).abs().max()) <= 1e-5
    assert float((dW.double() - Wd.grad).abs().max()) <= 2e-6 * float((dY.double().abs().t() @ hd.detach().abs()).max())
    assert float((db.double() - bd.grad).abs().max()) <= 2e-6 * float(dY.abs().sum(0).max())
    # no activation below: dprev = dY W
    _lib.check(libimx.imx_mlp_head_bwd(M, K, A, dY.data_ptr(), h.data_ptr(), K, W.data_ptr(), 0.0, 0, dprev.data_ptr(), dW.data_ptr(),
                                       db.data_ptr(), scratch.data_ptr(), nbytes, st))
    assert float((dprev.double() - dY.double() @ W.double()).abs().max()) <= 1e-5
    assert libimx.imx_mlp_head_fwd(M, K, 65, h.data_ptr(), K, W.data_ptr(), b.data_ptr(), y.data_ptr(), 0, 0.0, st) != 0


@pytest.mark.parametrize("M,K,A,elu_in", [(24576, 128, 12, 1), (3001, 128, 1, 1), (77, 256, 16, 1), (1000, 128, 5, 0), (40, 256, 1, 0), (5000, 128, 13, 1)])
def test_head_forward_loss_backward_in_one_launch_matches_the_split_pair(libimx, M, K, A, elu_in):
    """imx_mlp_head_fwd_bwd (ELU on the way in, outputs, loss gradient, dprev, dW, db in one pass) against imx_mlp_head_fwd_loss +
    imx_mlp_head_bwd: outputs / loss gradients to 1e-6 relative (the dot products are summed in another order), the backward results to
    the tolerance of test_mlp_head_matches_autograd; z is left untouched; the reduction lands on the batch it was deferred to."""
    from isaaclab_amd import _lib

    L = _lib.lib()
    g = torch.Generator().manual_seed(M + K + A)
    z = torch.randn(M, K, generator=g).cuda()
    W, b = (0.1 * torch.randn(A, K, generator=g)).cuda(), (0.1 * torch.randn(A, generator=g)).cuda()
    sigma = (0.5 + torch.rand(A, generator=g)).cuda()
    act = torch.randn(M, A, generator=g).cuda()
    old_logp = (-1.0 - torch.rand(M, generator=g) * A).cuda()
    adv, ret, old_v = torch.randn(M, generator=g).cuda(), torch.randn(M, generator=g).cuda(), torch.randn(M, generator=g).cuda()
    st = _lib.current_stream(z.device)

    def loss(dmu, dsg, dv):
        if A > 1:
            return _lib.ImxHeadLoss(mode=1, sigma_stride=0, use_clipped_value_loss=1, clip_param=0.2, value_loss_coef=1.0, entropy_coef=0.005,
                                    grad_scale=1.0, sigma_d=sigma.data_ptr(), actions_d=act.data_ptr(), old_logp_d=old_logp.data_ptr(),
                                    advantages_d=adv.data_ptr(), dmu_d=dmu.data_ptr(), dsigma_d=dsg.data_ptr())
        return _lib.ImxHeadLoss(mode=2, sigma_stride=0, use_clipped_value_loss=1, clip_param=0.2, value_loss_coef=1.0, entropy_coef=0.005,
                                grad_scale=1.0, returns_d=ret.data_ptr(), old_values_d=old_v.data_ptr(), dvalue_d=dv.data_ptr())

    nbytes = int(L.imx_mlp_scratch_bytes(M, A, K))
    res = []
    for fused in (False, True):
        zz, y = z.clone(), torch.empty(M, A, device="cuda")
        dmu, dsg, dv = torch.zeros(M, A, device="cuda"), torch.zeros(M, A, device="cuda"), torch.zeros(M, 1, device="cuda")
        dprev, dW, db = torch.empty(M, K, device="cuda"), torch.full((A, K), float("nan"), device="cuda"), torch.full((A,), float("nan"), device="cuda")
        scr = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        hl = loss(dmu, dsg, dv)
        if fused:
            h = ctypes.c_void_p()
            _lib.check(L.imx_reduce_batch_create(ctypes.byref(h)))
            _lib.check(L.imx_mlp_head_fwd_bwd(M, K, A, zz.data_ptr(), K, W.data_ptr(), b.data_ptr(), y.data_ptr(), elu_in, 1.0, ctypes.byref(hl),
                                              dprev.data_ptr(), dW.data_ptr(), db.data_ptr(), scr.data_ptr(), nbytes, h, st))
            torch.cuda.synchronize()
            assert torch.equal(zz, z) and bool(torch.isnan(dW).all())  # input untouched; reduction still queued
            _lib.check(L.imx_reduce_batch_begin(h))
            _lib.check(L.imx_reduce_batch_flush(h, st))
            L.imx_reduce_batch_destroy(h)
        else:
            _lib.check(L.imx_mlp_head_fwd_loss(M, K, A, zz.data_ptr(), K, W.data_ptr(), b.data_ptr(), y.data_ptr(), elu_in, 1.0, ctypes.byref(hl), st))
            _lib.check(L.imx_mlp_head_bwd(M, K, A, (dmu if A > 1 else dv).data_ptr(), zz.data_ptr(), K, W.data_ptr(), 1.0, elu_in, dprev.data_ptr(),
                                          dW.data_ptr(), db.data_ptr(), scr.data_ptr(), nbytes, st))
        torch.cuda.synchronize()
        res.append((y, dmu, dsg, dv, dprev, dW, db))
    (y0, dmu0, dsg0, dv0, dp0, dW0, db0), (y1, dmu1, dsg1, dv1, dp1, dW1, db1) = res
    rel = lambda a_, b_: float((a_ - b_).abs().max()) / max(float(b_.abs().max()), 1e-30)
    assert rel(y1, y0) <= 2e-6
    if A > 1:
        assert rel(dmu1, dmu0) <= 2e-5 and rel(dsg1, dsg0) <= 2e-5
    else:
        assert rel(dv1, dv0) <= 2e-5
    assert rel(dp1, dp0) <= 2e-5 and rel(dW1, dW0) <= 2e-5 and rel(db1, db0) <= 2e-5
    assert L.imx_mlp_head_fwd_bwd(M, 96, A, z.data_ptr(), K, W.data_ptr(), b.data_ptr(), y.data_ptr(), elu_in, 1.0, ctypes.byref(hl),
                                  dprev.data_ptr(), dW.data_ptr(), db.data_ptr(), scr.data_ptr(), nbytes, None, st) != 0  # K must be 128 or 256


@pytest.mark.parametrize("fused_norm", [False, True])
def test_adam_update_matches_torch_adam_and_adaptive_lr(libimx, fused_norm):
    """imx_adam_update (norm supplied) and imx_adam_update_norm (norm reduced in-kernel, last-block schedule) against
    torch.nn.utils.clip_grad_norm_ + torch.optim.Adam + the upstream adaptive-KL rule."""
    from isaaclab_amd import _lib

    n = 10_007 if not fused_norm else 571_801
    g0 = torch.Generator().manual_seed(1)
    p = torch.randn(n, generator=g0).cuda()
    ref_p = torch.nn.Parameter(p.clone())
    opt = torch.optim.Adam([ref_p], lr=1e-3)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    state = torch.tensor([1e-3, 0, 1, 1, 1, 0, 1, 0], dtype=torch.float32, device="cuda")
    nb = int(libimx.imx_adam_norm_scratch_bytes(n))
    scratch = torch.zeros(nb, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    lr_ref = 1e-3
    for step, kl in enumerate([0.004, 0.03, 0.012, 0.0, 0.05, 0.001]):
        grad = (torch.randn(n, generator=g0) * (3.0 if step % 2 else 0.01) / (n / 10_007) ** 0.5).cuda()
        ref_p.grad = grad.clone()
        # upstream order: LR decision from KL, then clip, then step
        if kl > 0.02: lr_ref = max(1e-5, lr_ref / 1.5)
        elif 0.0 < kl < 0.005: lr_ref = min(1e-2, lr_ref * 1.5)
        for gr in opt.param_groups: gr["lr"] = lr_ref
        torch.nn.utils.clip_grad_norm_([ref_p], 1.0)
        opt.step()
        norm = torch.linalg.vector_norm(grad).reshape(1)
        kl_t = torch.tensor([kl], device="cuda")
        if fused_norm:
            _lib.check(libimx.imx_adam_update_norm(n, p.data_ptr(), grad.data_ptr(), m.data_ptr(), v.data_ptr(), state.data_ptr(),
                                                   kl_t.data_ptr(), 0.01, 1.0, 0.9, 0.999, 1e-8, scratch.data_ptr(), nb, st))
            assert abs(float(state[7]) - float(norm)) <= 2e-6 * float(norm)  # the norm it clipped with
        else:
            _lib.check(libimx.imx_adam_update(n, p.data_ptr(), grad.data_ptr(), m.data_ptr(), v.data_ptr(), state.data_ptr(),
                                              kl_t.data_ptr(), 0.01, norm.data_ptr(), 1.0, 0.9, 0.999, 1e-8, st))
        assert abs(float(state[0]) - lr_ref) < 1e-9 and int(state[1]) == step + 1
        assert_close(p, ref_p.detach(), 1e-5, f"adam step {step}")


def test_minibatch_generator_is_a_permutation_gather(libimx):
    from isaaclab_amd.rsl_rl.storage import RolloutStorage

    T, N, D, A = 6, 50, 7, 3
    st = RolloutStorage(N, T, [D], [0], [A], device="cuda:0")
    g = torch.Generator().manual_seed(0)
    for name in ("observations", "actions", "values", "advantages", "returns", "actions_log_prob", "mu", "sigma"):
        getattr(st, name).copy_(torch.randn(getattr(st, name).shape, generator=g))
    # tag every transition with its flat index so that the permutation can be read back
    st.values.copy_(torch.arange(T * N, dtype=torch.float32).view(T, N, 1))
    seen = []
    for (obs, cobs, act, val, adv, ret, logp, mu, sg) in st.mini_batch_generator(4, 2):
        assert obs.shape == (T * N // 4, D) and cobs is obs
        idx = val[:, 0].long()
        assert torch.equal(obs, st.observations.flatten(0, 1)[idx]) and torch.equal(act, st.actions.flatten(0, 1)[idx])
        assert torch.equal(adv, st.advantages.flatten(0, 1)[idx]) and torch.equal(sg, st.sigma.flatten(0, 1)[idx])
        assert torch.equal(logp, st.actions_log_prob.flatten(0, 1)[idx]) and torch.equal(mu, st.mu.flatten(0, 1)[idx])
        seen.append(idx.clone())
    first_epoch = torch.cat(seen[:4]).sort().values
    assert torch.equal(first_epoch.cpu(), torch.arange(T * N))  # every transition exactly once per epoch


def test_policy_act_samples_and_logprob(libimx):
    from isaaclab_amd import _lib

    N, A, D = 4096, 12, 20
    g = torch.Generator().manual_seed(2)
    mu = torch.randn(N, A, generator=g).cuda()
    std = (torch.rand(A, generator=g) * 0.8 + 0.2).cuda()
    value = torch.randn(N, 1, generator=g).cuda()
    obs = torch.randn(N, D, generator=g).cuda()
    step = torch.tensor([7], dtype=torch.int32, device="cuda")
    out = {k: torch.empty(N, w, device="cuda") for k, w in dict(act=A, logp=1, mu=A, sigma=A, val=1, obs=D, env=A).items()}
    s = torch.cuda.current_stream().cuda_stream

    def run(seed):
        _lib.check(libimx.imx_policy_act(N, A, D, mu.data_ptr(), std.data_ptr(), value.data_ptr(), obs.data_ptr(), seed,
                                         step.data_ptr(), out["act"].data_ptr(), out["logp"].data_ptr(), out["mu"].data_ptr(),
                                         out["sigma"].data_ptr(), out["val"].data_ptr(), out["obs"].data_ptr(),
                                         out["env"].data_ptr(), s))

    run(123)
    a1 = out["act"].clone()
    z = (a1 - mu) / std
    assert abs(float(z.mean())) < 0.02 and abs(float(z.std()) - 1.0) < 0.02 and float(z.abs().max()) < 6.5
    assert abs(float((z[:, 0] * z[:, 1]).mean())) < 0.05  # neighbouring draws uncorrelated
    ref_logp = torch.distributions.Normal(mu, std.expand_as(mu)).log_prob(a1).sum(-1, keepdim=True)
    assert_close(out["logp"], ref_logp, 1e-5, "log_prob")
    assert torch.equal(out["mu"], mu) and torch.equal(out["sigma"], std.expand_as(mu)) and torch.equal(out["val"], value)
    assert torch.equal(out["obs"], obs) and torch.equal(out["env"], a1)
    run(123)
    assert torch.equal(out["act"], a1)  # counter-based: same (seed, step) -> same draws
    step += 1
    run(123)
    assert not torch.equal(out["act"], a1)


def test_rollout_post_bootstrap_and_episode_stats(libimx):
    from isaaclab_amd import _lib
    from oracle.rsl_rl_oracle import bootstrap_time_outs

    N = 1000
    g = torch.Generator().manual_seed(4)
    rew = torch.randn(N, generator=g).cuda()
    val = torch.randn(N, 1, generator=g).cuda()
    term = (torch.rand(N, generator=g) < 0.1).cuda()
    trunc = (torch.rand(N, generator=g) < 0.1).cuda()
    rew_out = torch.empty(N, 1, device="cuda"); dones = torch.empty(N, 1, dtype=torch.uint8, device="cuda")
    dl = torch.empty(N, dtype=torch.long, device="cuda")
    cur_r, cur_l = torch.rand(N, generator=g).cuda(), torch.randint(0, 50, (N,), generator=g).float().cuda()
    cr0, cl0 = cur_r.clone(), cur_l.clone()
    stats = torch.zeros(3, device="cuda")
    log_in = torch.randn(300, generator=g).cuda()  # more entries than a workgroup has threads
    log_acc = torch.ones(300, device="cuda")
    _lib.check(libimx.imx_rollout_post(N, rew.data_ptr(), term.data_ptr(), trunc.data_ptr(), val.data_ptr(), 0.99, 1,
                                       rew_out.data_ptr(), dones.data_ptr(), dl.data_ptr(), cur_r.data_ptr(), cur_l.data_ptr(),
                                       stats.data_ptr(), log_in.data_ptr(), log_acc.data_ptr(), 300, torch.cuda.current_stream().cuda_stream))
    assert torch.equal(log_acc, 1.0 + log_in)  # the runner's per-iteration sum of extras["log"]
    assert_close(rew_out[:, 0], bootstrap_time_outs(rew, val, trunc, 0.99), 1e-6, "time-out bootstrap")
    d = term | trunc
    assert torch.equal(dones[:, 0].bool(), d) and torch.equal(dl, d.long())
    assert_close(stats, torch.stack([((cr0 + rew) * d).sum(), ((cl0 + 1) * d).sum(), d.sum().float()]), 1e-5, "episode stats")
    assert_close(cur_r, (cr0 + rew) * (~d), 1e-6, "running reward") and assert_close(cur_l, (cl0 + 1) * (~d), 0, "running length") is None


@pytest.mark.parametrize("Dc,hidden,fused_head,tol,A", [(0, [64, 32], "0", 2e-5, 5), (53, [64, 32], "0", 2e-5, 5), (0, [256, 128], "1", 1e-4, 5),
                                                        (41, [128, 128], "1", 1e-4, 5), (41, [128, 128], "0", 1e-4, 5),
                                                        (0, [64, 32], "0", 2e-5, 16), (0, [64, 32], "0", 2e-5, 17), (0, [64, 32], "0", 2e-5, 37)])
def test_whole_update_matches_torch_reference(libimx, monkeypatch, Dc, hidden, fused_head, tol, A):
    """PPO.update end to end (minibatch gather, both MLPs forward/backward on the HIP kernels, loss, grad-norm clip, adaptive
    KL learning rate, Adam, 2 epochs x 3 minibatches) against the same algorithm written with torch autograd, torch.optim.Adam
    and the rsl_rl restatement (oracle/rsl_rl_oracle.py) on identical data and the identical minibatch permutation.
    Cases: shared observations (the stacked first layer in one imx_mlp_fwd_elu launch) and a privileged critic group (Dc > 0: each
    network's first layer through imx_mlp_fwd_elu on its own stream); the split output-layer pair and imx_mlp_head_fwd_bwd.
    Tolerance on the parameters after six Adam steps: 2e-5 for the narrow networks; 1e-4 (a tenth of ONE learning-rate step) for the
    128-wide hidden layers, where Adam's g / (sqrt(v) + eps) turns the rounding noise of a near-zero gradient element into a visible
    step -- the split pair and the fused head differ from torch by the same 1e-5 .. 5e-5 there."""
    import copy

    from isaaclab_amd.rsl_rl.actor_critic import ActorCritic
    from isaaclab_amd.rsl_rl.ppo import PPO
    from oracle.rsl_rl_oracle import adaptive_lr, ppo_losses

    import isaaclab_amd.rsl_rl.ppo as ppo_mod

    monkeypatch.setattr(ppo_mod, "FUSED_HEAD", fused_head)
    T, N, D = 6, 50, 37
    torch.manual_seed(11)
    pol = ActorCritic(D, Dc or D, A, actor_hidden_dims=list(hidden), critic_hidden_dims=list(hidden), init_noise_std=0.8)
    ref_pol = copy.deepcopy(pol).cuda()
    kw = dict(num_learning_epochs=2, num_mini_batches=3, schedule="adaptive", desired_kl=0.01, learning_rate=1e-3, entropy_coef=0.005,
              max_grad_norm=1.0, clip_param=0.2, value_loss_coef=1.0, use_clipped_value_loss=True)
    alg = PPO(pol, device="cuda:0", **kw)
    alg.init_storage("rl", N, T, (D,), (Dc,), (A,))  # Dc = 0: no privileged observations, the critic sees the policy observations
    st = alg.storage
    g = torch.Generator().manual_seed(5)
    st.observations.copy_(torch.randn(T, N, D, generator=g))
    if Dc:
        st.privileged_observations.copy_(torch.randn(T, N, Dc, generator=g))
    cobs_all = st.privileged_observations if Dc else st.observations
    with torch.no_grad():
        mu = ref_pol.actor(st.observations.flatten(0, 1)).view(T, N, A)
        val = ref_pol.critic(cobs_all.flatten(0, 1)).view(T, N, 1)
    sigma = ref_pol.std.detach().expand(T, N, A).contiguous()
    act = mu + sigma * torch.randn(T, N, A, generator=g).cuda()
    st.mu.copy_(mu); st.sigma.copy_(sigma); st.actions.copy_(act); st.values.copy_(val)
    st.actions_log_prob.copy_(torch.distributions.Normal(mu, sigma).log_prob(act).sum(-1, keepdim=True))
    st.returns.copy_(val + 0.3 * torch.randn(T, N, 1, generator=g).cuda())
    st.advantages.copy_(torch.randn(T, N, 1, generator=g))
    st.step = T
    flat = lambda x: x.flatten(0, 1)  # noqa: E731
    data = [flat(x).clone() for x in (st.observations, st.actions, st.values, st.advantages, st.returns, st.actions_log_prob, st.mu, st.sigma, cobs_all)]

    # ---- product path
    torch.manual_seed(99)  # seeds the randperm of the minibatch generator
    alg.update()
    torch.cuda.synchronize()

    # ---- reference: same permutation, torch autograd + clip_grad_norm_ + Adam + the upstream adaptive-LR rule
    torch.manual_seed(99)
    M = T * N // kw["num_mini_batches"]
    perm = torch.randperm(kw["num_mini_batches"] * M, device="cuda:0")
    opt = torch.optim.Adam(ref_pol.parameters(), lr=kw["learning_rate"])
    lr = kw["learning_rate"]
    for _ in range(kw["num_learning_epochs"]):
        for i in range(kw["num_mini_batches"]):
            idx = perm[i * M:(i + 1) * M]
            obs, a_, v_old, adv, ret, logp_old, mu_old, sg_old, cobs = (x[idx] for x in data)
            mu_b = ref_pol.actor(obs)
            s, v, e, kl = ppo_losses(mu_b, ref_pol.std.expand_as(mu_b), a_, logp_old, mu_old, sg_old, adv, ret, ref_pol.critic(cobs), v_old,
                                     kw["clip_param"], True)
            lr = adaptive_lr(lr, float(kl), kw["desired_kl"])
            for gr in opt.param_groups:
                gr["lr"] = lr
            loss = s + kw["value_loss_coef"] * v - kw["entropy_coef"] * e
            opt.zero_grad()
            loss.backward()
            torch.nn.utils.clip_grad_norm_(ref_pol.parameters(), kw["max_grad_norm"])
            opt.step()
    assert abs(alg.learning_rate - lr) <= 1e-9 * max(1.0, lr), (alg.learning_rate, lr)
    for (name, p), q in zip(pol.named_parameters(), ref_pol.parameters()):
        err = float((p - q).abs().max())
        assert err <= tol * max(1.0, float(q.abs().max())), f"{name}: max err {err:.2e} after {kw['num_learning_epochs'] * kw['num_mini_batches']} optimiser steps"
    stats = alg.loss_dict()
    assert all(np.isfinite(x) for x in stats.values())


@pytest.mark.gpu
def test_update_graph_replay_equals_eager_update(libimx):
    """PPO.update captured as one hipGraph (update_graph, used by the runner with use_graph=True) against the eager update: same
    storage, same seed, six updates each -- the captured torch.randperm consumes the generator exactly like the eager one, so
    parameters, Adam state, learning rate and logged losses must agree to the last bit."""
    import copy

    from isaaclab_amd.rsl_rl.actor_critic import ActorCritic
    from isaaclab_amd.rsl_rl.ppo import PPO

    T, N, D, A = 8, 96, 48, 12
    torch.manual_seed(3)
    pol0 = ActorCritic(D, D, A, actor_hidden_dims=[128, 64], critic_hidden_dims=[128, 64], init_noise_std=1.0)
    kw = dict(num_learning_epochs=2, num_mini_batches=4, schedule="adaptive", desired_kl=0.01, learning_rate=1e-3, entropy_coef=0.005,
              max_grad_norm=1.0, clip_param=0.2, value_loss_coef=1.0, use_clipped_value_loss=True)
    g = torch.Generator().manual_seed(8)
    obs = torch.randn(T, N, D, generator=g)
    noise = torch.randn(T, N, A, generator=g)
    ret_noise, adv = 0.3 * torch.randn(T, N, 1, generator=g), torch.randn(T, N, 1, generator=g)
    results = []
    for graph in (False, True):
        alg = PPO(copy.deepcopy(pol0), device="cuda:0", **kw)
        alg.update_graph = graph
        alg.init_storage("rl", N, T, (D,), (0,), (A,))
        torch.manual_seed(1234)
        stats = []
        for it in range(6):  # graph mode: eager, eager (timed), capture + replay, replay (timed), choice, chosen
            if graph and it == 4:
                alg._update_t = "graph"  # pin the choice: this test is about the equality, not about which is faster here
            st = alg.storage
            st.observations.copy_(obs + 0.1 * it)
            with torch.no_grad():
                mu = alg.policy.actor(st.observations.flatten(0, 1)).view(T, N, A)
                val = alg.policy.critic(st.observations.flatten(0, 1)).view(T, N, 1)
            sigma = alg.policy.std.detach().expand(T, N, A).contiguous()
            act = mu + sigma * noise.cuda()
            st.mu.copy_(mu); st.sigma.copy_(sigma); st.actions.copy_(act); st.values.copy_(val)
            st.actions_log_prob.copy_(torch.distributions.Normal(mu, sigma).log_prob(act).sum(-1, keepdim=True))
            st.returns.copy_(val + ret_noise.cuda())
            st.advantages.copy_(adv)
            st.step = T
            alg.update()
            stats.append(alg.loss_dict())
        torch.cuda.synchronize()
        assert (alg._update_g is not None) == graph
        results.append((alg.bucket.flat.clone(), alg.bucket.exp_avg.clone(), alg.bucket.exp_avg_sq.clone(), alg.learning_rate, stats))
    (p0, m0, v0, lr0, s0), (p1, m1, v1, lr1, s1) = results
    assert lr0 == lr1 and s0 == s1
    assert torch.equal(p0, p1) and torch.equal(m0, m1) and torch.equal(v0, v1)


@pytest.mark.gpu
@pytest.mark.parametrize("M,A", [(24576, 12), (24576, 37), (1000, 64), (7, 1)])
def test_colsum_matches_torch(libimx, M, A):
    """imx_colsum (std / log-std gradient from the per-sample dsigma) against a float64 column sum; deterministic across calls."""
    from isaaclab_amd import _lib

    L = _lib.lib()
    g = torch.Generator().manual_seed(M + A)
    x = torch.randn(M, A, generator=g).cuda()
    y = (torch.rand(M, A, generator=g) + 0.5).cuda()
    scratch = torch.empty(int(L.imx_colsum_scratch_bytes()), dtype=torch.uint8, device="cuda")
    for other in (None, y):
        out = torch.full((A,), float("nan"), device="cuda")
        _lib.check(L.imx_colsum(M, A, x.data_ptr(), None if other is None else other.data_ptr(), out.data_ptr(), scratch.data_ptr(),
                                _lib.current_stream(x.device)))
        ref = (x.double() * (1.0 if other is None else other.double())).sum(0)
        scale = float((x.double() * (1.0 if other is None else other.double())).abs().sum(0).max())
        assert float((out.double() - ref).abs().max()) <= 1e-6 * scale
        out2 = torch.empty_like(out)
        _lib.check(L.imx_colsum(M, A, x.data_ptr(), None if other is None else other.data_ptr(), out2.data_ptr(), scratch.data_ptr(),
                                _lib.current_stream(x.device)))
        assert torch.equal(out, out2)
    with pytest.raises(_lib.ImxError):
        _lib.check(L.imx_colsum(M, 65, x.data_ptr(), None, out.data_ptr(), scratch.data_ptr(), _lib.current_stream(x.device)))


@pytest.mark.gpu
@pytest.mark.parametrize("A", [12, 37, 1])
def test_head_forward_with_fused_loss_gradient_equals_two_launches(libimx, A):
    """imx_mlp_head_fwd_loss = imx_mlp_head_fwd + imx_ppo_loss_bwd in one launch: outputs and loss gradients bit-identical to
    the two separate launches (policy head for A > 1, value head for A = 1)."""
    import ctypes

    from isaaclab_amd import _lib

    L = _lib.lib()
    M, K = 3000, 128
    g = torch.Generator().manual_seed(40 + A)
    h = torch.randn(M, K, generator=g).cuda()
    W, b = (0.1 * torch.randn(A, K, generator=g)).cuda(), (0.1 * torch.randn(A, generator=g)).cuda()
    sigma = (0.5 + torch.rand(A, generator=g)).cuda()
    act = torch.randn(M, A, generator=g).cuda()
    old_logp = (-1.0 - torch.rand(M, generator=g) * A).cuda()
    adv, ret, old_v = torch.randn(M, generator=g).cuda(), torch.randn(M, generator=g).cuda(), torch.randn(M, generator=g).cuda()
    st = _lib.current_stream(h.device)
    clip, vcoef, ecoef = 0.2, 1.0, 0.005
    # two launches
    h1, y1 = h.clone(), torch.empty(M, A, device="cuda")
    _lib.check(L.imx_mlp_head_fwd(M, K, A, h1.data_ptr(), K, W.data_ptr(), b.data_ptr(), y1.data_ptr(), 1, 1.0, st))
    dmu1, dsg1, dv1 = torch.empty(M, A, device="cuda"), torch.empty(M, A, device="cuda"), torch.empty(M, 1, device="cuda")
    if A > 1:
        _lib.check(L.imx_ppo_loss_bwd(M, A, y1.data_ptr(), sigma.data_ptr(), 0, act.data_ptr(), old_logp.data_ptr(), adv.data_ptr(), None, None,
                                      None, clip, 1, vcoef, ecoef, 1.0, dmu1.data_ptr(), dsg1.data_ptr(), None, st))
    else:
        _lib.check(L.imx_ppo_loss_bwd(M, A, None, None, 0, None, None, None, ret.data_ptr(), y1.data_ptr(), old_v.data_ptr(), clip, 1, vcoef,
                                      ecoef, 1.0, None, None, dv1.data_ptr(), st))
    # one launch
    h2, y2 = h.clone(), torch.empty(M, A, device="cuda")
    dmu2, dsg2, dv2 = torch.empty(M, A, device="cuda"), torch.empty(M, A, device="cuda"), torch.empty(M, 1, device="cuda")
    if A > 1:
        hl = _lib.ImxHeadLoss(mode=1, sigma_stride=0, use_clipped_value_loss=1, clip_param=clip, value_loss_coef=vcoef, entropy_coef=ecoef,
                              grad_scale=1.0, sigma_d=sigma.data_ptr(), actions_d=act.data_ptr(), old_logp_d=old_logp.data_ptr(),
                              advantages_d=adv.data_ptr(), dmu_d=dmu2.data_ptr(), dsigma_d=dsg2.data_ptr())
    else:
        hl = _lib.ImxHeadLoss(mode=2, sigma_stride=0, use_clipped_value_loss=1, clip_param=clip, value_loss_coef=vcoef, entropy_coef=ecoef,
                              grad_scale=1.0, returns_d=ret.data_ptr(), old_values_d=old_v.data_ptr(), dvalue_d=dv2.data_ptr())
    _lib.check(L.imx_mlp_head_fwd_loss(M, K, A, h2.data_ptr(), K, W.data_ptr(), b.data_ptr(), y2.data_ptr(), 1, 1.0, ctypes.byref(hl), st))
    torch.cuda.synchronize()
    assert torch.equal(y1, y2) and torch.equal(h1, h2)
    if A > 1:
        assert torch.equal(dmu1, dmu2) and torch.equal(dsg1, dsg2) and bool(torch.isfinite(dmu2).all())
    else:
        assert torch.equal(dv1, dv2)
    bad = _lib.ImxHeadLoss(mode=2, returns_d=ret.data_ptr(), dvalue_d=dv2.data_ptr(), use_clipped_value_loss=0)
    if A > 1:
        with pytest.raises(_lib.ImxError):  # the value head has one output
            _lib.check(L.imx_mlp_head_fwd_loss(M, K, A, h2.data_ptr(), K, W.data_ptr(), b.data_ptr(), y2.data_ptr(), 0, 1.0, ctypes.byref(bad), st))


@pytest.mark.gpu
def test_update_through_the_rccl_path_single_rank(libimx):
    """The multi-GPU branch of PPO.update (bucket all-reduce over RCCL + division by the world size, eager update) on a single-rank
    nccl group: one rank's mean is its own gradient, so parameters must equal the non-distributed update bit for bit."""
    import copy
    import socket

    import torch.distributed as dist

    from isaaclab_amd.rsl_rl.actor_critic import ActorCritic
    from isaaclab_amd.rsl_rl.ppo import PPO

    T, N, D, A = 8, 64, 48, 12
    torch.manual_seed(5)
    pol0 = ActorCritic(D, D, A, actor_hidden_dims=[128, 64], critic_hidden_dims=[128, 64], init_noise_std=1.0)
    kw = dict(num_learning_epochs=2, num_mini_batches=4, schedule="adaptive", desired_kl=0.01, learning_rate=1e-3, entropy_coef=0.005,
              max_grad_norm=1.0, clip_param=0.2, value_loss_coef=1.0, use_clipped_value_loss=True)
    g = torch.Generator().manual_seed(9)
    obs, noise = torch.randn(T, N, D, generator=g), torch.randn(T, N, A, generator=g)
    ret_noise, adv = 0.3 * torch.randn(T, N, 1, generator=g), torch.randn(T, N, 1, generator=g)

    def run(multi, iters=2):
        alg = PPO(copy.deepcopy(pol0), device="cuda:0", multi_gpu_cfg={"global_rank": 0, "local_rank": 0, "world_size": 1} if multi else None, **kw)
        alg.update_graph = False
        alg.init_storage("rl", N, T, (D,), (0,), (A,))
        torch.manual_seed(77)
        for it in range(iters):
            st = alg.storage
            st.observations.copy_(obs + 0.1 * it)
            with torch.no_grad():
                mu = alg.policy.actor(st.observations.flatten(0, 1)).view(T, N, A)
                val = alg.policy.critic(st.observations.flatten(0, 1)).view(T, N, 1)
            sigma = alg.policy.std.detach().expand(T, N, A).contiguous()
            act = mu + sigma * noise.cuda()
            st.mu.copy_(mu); st.sigma.copy_(sigma); st.actions.copy_(act); st.values.copy_(val)
            st.actions_log_prob.copy_(torch.distributions.Normal(mu, sigma).log_prob(act).sum(-1, keepdim=True))
            st.returns.copy_(val + ret_noise.cuda())
            st.advantages.copy_(adv)
            st.step = T
            alg.update()
        torch.cuda.synchronize()
        return alg.bucket.flat.clone(), alg.learning_rate

    ref, lr_ref = run(False)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        got, lr = run(True)
    finally:
        dist.destroy_process_group()
    assert lr == lr_ref and torch.equal(got, ref)


@pytest.mark.parametrize("M,N,K,pitch,w_off", [(24576, 1024, 235, 236, 0), (1000, 512, 48, 48, 0), (77, 96, 4, 4, 0), (4099, 640, 256, 256, 0),
                                               (64, 1024, 235, 235, 0), (20, 128, 235, 236, 0), (3000, 200, 235, 236, 0), (2049, 256, 200, 200, 0),
                                               (515, 384, 130, 132, 0), (1500, 256, 235, 236, 1), (700, 130, 48, 48, 3),
                                               # ragged last column block whose float count is not a multiple of four (found by tools/fuzz_kernels.py)
                                               (31, 31, 223, 224, 0), (2000, 31, 143, 148, 0), (300, 161, 235, 236, 0)])
def test_fused_first_layer_forward_matches_torch(libimx, M, N, K, pitch, w_off):
    """imx_mlp_fwd_elu (Linear + ELU of the first layer, weights in registers, bias + ELU on the accumulators) against torch's
    addmm + elu in fp64-checked fp32: ragged tiles, a single tile, column counts that are not multiples of 128 (down to whole waves
    without a column), a pitch that pads the rows (the pad columns hold NaNs on purpose: they must never reach the result), unaligned
    rows (4-byte loads), K well below the step count of its kernel (many tile columns to zero), a weight matrix off 16-byte alignment."""
    from isaaclab_amd import _lib

    g = torch.Generator().manual_seed(M + N + K)
    xb = torch.full((M, pitch), float("nan"))
    xb[:, :K] = torch.randn(M, K, generator=g)
    x = xb.cuda()[:, :K]
    wflat = torch.full((N * K + 8,), float("nan"))
    wflat[w_off:w_off + N * K] = torch.randn(N * K, generator=g) / K ** 0.5
    w, b = wflat.cuda()[w_off:w_off + N * K].view(N, K), torch.randn(N, generator=g).cuda()
    for elu in (1, 0):
        y = torch.full((M, N + 3), 7.0, device="cuda")
        _lib.check(libimx.imx_mlp_fwd_elu(M, N, K, x.data_ptr(), x.stride(0), w.data_ptr(), b.data_ptr(), 0.7, elu, y.data_ptr(), y.stride(0),
                                          torch.cuda.current_stream().cuda_stream))
        ref = torch.addmm(b.double(), x.double(), w.double().t())
        if elu:
            ref = torch.nn.functional.elu(ref, alpha=0.7)
        assert_close(y[:, :N], ref.float(), 2e-5, f"fused first layer (elu={elu})")
        assert bool((y[:, N:] == 7.0).all())  # nothing written beyond the N columns
    with pytest.raises(_lib.ImxError):
        _lib.check(libimx.imx_mlp_fwd_elu(8, 8, 300, x.data_ptr(), 300, w.data_ptr(), b.data_ptr(), 1.0, 1, y.data_ptr(), 8, 0))
