// RSL-RL rollout kernels: GAE backward scan + advantage normalisation, fused PPO elementwise loss (fwd/bwd).
//
// rsl-rl-lib==2.3.1 is a third-party dependency that is NOT in /root/reference (empty submodule; pinned in
// source/isaaclab_rl/setup.py:47 and scripts/reinforcement_learning/rsl_rl/train.py:56).  The arithmetic below
// restates the published upstream v2.3.1 algorithm (rsl_rl/storage/rollout_storage.py::compute_returns,
// rsl_rl/algorithms/ppo.py::update); PARITY UNPINNED by the reference -- pinned only against oracle/rsl_rl_oracle.py.
//
// GAE layout: (T,N) row-major (the (T,N,1) storage tensors), one env per lane, serial over T -- loads/stores of a
// wave are 64 consecutive floats at every t.
#include "imx_internal.h"

// Welford/Chan accumulator
struct Mom {
    float n, mean, m2;
};
IMX_DEV Mom mom_merge(Mom a, Mom b) {
    Mom r;
    r.n = a.n + b.n;
    if (r.n == 0.0f) { r.mean = 0.0f; r.m2 = 0.0f; return r; }
    const float d = b.mean - a.mean;
    r.mean = a.mean + d * (b.n / r.n);
    r.m2 = a.m2 + b.m2 + d * d * (a.n * b.n / r.n);
    return r;
}
IMX_DEV Mom mom_wave(Mom m) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        Mom b;
        b.n = __shfl_xor(m.n, o, 64);
        b.mean = __shfl_xor(m.mean, o, 64);
        b.m2 = __shfl_xor(m.m2, o, 64);
        // keep the merge order symmetric so both partners compute identical bits
        const bool lo = ((threadIdx.x & o) == 0);
        m = lo ? mom_merge(m, b) : mom_merge(b, m);
    }
    return m;
}

constexpr int GAE_BLOCK = 64;  // one wave per workgroup: 4096 envs = 64 workgroups on 64 CUs instead of 16 on 16
constexpr int GAE_CHUNK = 8;   // steps whose loads are in flight together

extern "C" size_t imx_gae_scratch_bytes(int64_t T, int64_t N) {
    (void)T;
    const size_t nblk = (size_t)((N + GAE_BLOCK - 1) / GAE_BLOCK);
    return (nblk * 3 + 8) * sizeof(float);
}

// The scan over t is a true recurrence, its LOADS are not: reward, value and done flag of every step are known before the kernel
// starts.  The walk goes in chunks of GAE_CHUNK steps, the 3 x GAE_CHUNK loads of chunk c+1 issued before the arithmetic of chunk c
// (two register sets, indices clamped so that every load is unconditional) -- the kernel is one or two memory round trips deep
// instead of T of them (round 2: three dependent loads inside every trip of the recurrence, 13 us for T = 24).
struct GaeChunk {
    float r[GAE_CHUNK], v[GAE_CHUNK];
    uint8_t d[GAE_CHUNK];
};
IMX_DEV void gae_load(GaeChunk& c, int64_t t_hi, int64_t N, int64_t e, const float* __restrict__ rew, const float* __restrict__ val,
                      const uint8_t* __restrict__ dones) {
#pragma unroll
    for (int u = 0; u < GAE_CHUNK; ++u) {
        const int64_t t = t_hi - u > 0 ? t_hi - u : 0;
        c.r[u] = rew[t * N + e];
        c.v[u] = val[t * N + e];
        c.d[u] = dones[t * N + e];
    }
}

// scratch: [0..3*nblk) per-workgroup moments of the raw advantages
__global__ void __launch_bounds__(GAE_BLOCK)
k_gae(int64_t T, int64_t N, const float* __restrict__ rew, const float* __restrict__ val,
      const uint8_t* __restrict__ dones, const float* __restrict__ last_val, float gamma, float lam,
      float* __restrict__ ret, float* __restrict__ adv, float* __restrict__ scratch) {
    const int64_t e0 = (int64_t)blockIdx.x * GAE_BLOCK + threadIdx.x;
    const bool live = e0 < N;
    const int64_t e = live ? e0 : N - 1;  // dead lanes walk a valid env, never store
    Mom m{0.0f, 0.0f, 0.0f};
    float advantage = 0.0f;
    float next_v = last_val[e];
    GaeChunk ca, cb;
    auto compute = [&](const GaeChunk& c, int64_t t_hi) {
#pragma unroll
        for (int u = 0; u < GAE_CHUNK; ++u) {
            const int64_t t = t_hi - u;
            if (t < 0) break;  // uniform
            const float v = c.v[u];
            const float nnt = 1.0f - (c.d[u] ? 1.0f : 0.0f);
            const float delta = c.r[u] + nnt * gamma * next_v - v;
            advantage = delta + nnt * gamma * lam * advantage;
            const float r = advantage + v;
            const float a = r - v;  // self.advantages = self.returns - self.values
            if (live) {
                ret[t * N + e] = r;
                adv[t * N + e] = a;
            }
            // Welford update
            m.n += 1.0f;
            const float d = a - m.mean;
            m.mean += d / m.n;
            m.m2 += d * (a - m.mean);
            next_v = v;
        }
    };
    int64_t t_hi = T - 1;
    gae_load(ca, t_hi, N, e, rew, val, dones);
    while (t_hi >= 0) {
        gae_load(cb, t_hi - GAE_CHUNK, N, e, rew, val, dones);  // (clamped: a redundant reload of step 0 at the end)
        compute(ca, t_hi);
        t_hi -= GAE_CHUNK;
        if (t_hi < 0) break;
        gae_load(ca, t_hi - GAE_CHUNK, N, e, rew, val, dones);
        compute(cb, t_hi);
        t_hi -= GAE_CHUNK;
    }
    if (!live) m = Mom{0.0f, 0.0f, 0.0f};
    m = mom_wave(m);
    if (threadIdx.x == 0) {
        scratch[3 * blockIdx.x + 0] = m.n;
        scratch[3 * blockIdx.x + 1] = m.mean;
        scratch[3 * blockIdx.x + 2] = m.m2;
    }
}

// every wave merges the per-workgroup moments in the same fixed order -- lane i takes partials i, i + 64, ... in sequence, then the
// symmetric shuffle tree of mom_wave -- and normalises its slice, four elements per lane and trip
__global__ void __launch_bounds__(256)
k_adv_normalize(int64_t total, int nblk, const float* __restrict__ scratch, float* __restrict__ adv) {
    const int lane = threadIdx.x & 63;
    Mom b{0.0f, 0.0f, 0.0f};
    for (int i = lane; i < nblk; i += 64) {
        Mom c{scratch[3 * i], scratch[3 * i + 1], scratch[3 * i + 2]};
        b = mom_merge(b, c);
    }
    b = mom_wave(b);
    const float var = b.n > 1.0f ? b.m2 / (b.n - 1.0f) : 0.0f;  // torch.std: unbiased
    const float mean = b.mean, inv = 1.0f / (sqrtf(var) + 1.0e-8f);
    const int64_t nth = (int64_t)gridDim.x * blockDim.x, tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if ((reinterpret_cast<uintptr_t>(adv) & 15) == 0) {
        float4* a4 = reinterpret_cast<float4*>(adv);
        const int64_t n4 = total >> 2;
        for (int64_t i = tid; i < n4; i += nth) {
            float4 x = a4[i];
            x.x = (x.x - mean) * inv; x.y = (x.y - mean) * inv; x.z = (x.z - mean) * inv; x.w = (x.w - mean) * inv;
            a4[i] = x;
        }
        for (int64_t i = (n4 << 2) + tid; i < total; i += nth) adv[i] = (adv[i] - mean) * inv;
    } else {
        for (int64_t i = tid; i < total; i += nth) adv[i] = (adv[i] - mean) * inv;
    }
}

// The usual rollout lengths (T <= 32): EVERY load of the scan in flight at once -- one memory round trip for the whole kernel.
// (A single-launch form -- the lane keeps its T advantages in registers, the workgroups meet at an in-kernel grid barrier, every wave
// merges the moments and stores normalised advantages once -- was built and measured: 16.1 us against 12.8 us for the two launches.
// An agent-scope release / acquire pair on this eight-XCD part is an L2 write-back + invalidate, ~5 us, the same finding as for the
// step tail in round 2.  Two launches stay.)
constexpr int GAE_TMAX = 32;
__global__ void __launch_bounds__(GAE_BLOCK)
k_gae_short(int T, int64_t N, const float* __restrict__ rew, const float* __restrict__ val, const uint8_t* __restrict__ dones,
            const float* __restrict__ last_val, float gamma, float lam, float* __restrict__ ret, float* __restrict__ adv,
            float* __restrict__ scratch) {
    const int64_t e0 = (int64_t)blockIdx.x * GAE_BLOCK + threadIdx.x;
    const bool live = e0 < N;
    const int64_t e = live ? e0 : N - 1;
    float r[GAE_TMAX], v[GAE_TMAX];
    uint8_t d[GAE_TMAX];
    float next_v = last_val[e];
#pragma unroll
    for (int u = 0; u < GAE_TMAX; ++u) {  // (rows past T - 1 re-read the last one: every load unconditional)
        const int64_t t = u < T ? u : T - 1;
        r[u] = rew[t * N + e];
        v[u] = val[t * N + e];
        d[u] = dones[t * N + e];
    }
    Mom m{0.0f, 0.0f, 0.0f};
    float advantage = 0.0f;
#pragma unroll
    for (int u = GAE_TMAX - 1; u >= 0; --u) {
        if (u < T) {  // uniform
            const float nnt = 1.0f - (d[u] ? 1.0f : 0.0f);
            const float delta = r[u] + nnt * gamma * next_v - v[u];
            advantage = delta + nnt * gamma * lam * advantage;
            const float rt = advantage + v[u];
            const float at = rt - v[u];  // self.advantages = self.returns - self.values
            if (live) {
                ret[(int64_t)u * N + e] = rt;
                adv[(int64_t)u * N + e] = at;
            }
            m.n += 1.0f;
            const float dd = at - m.mean;
            m.mean += dd / m.n;
            m.m2 += dd * (at - m.mean);
            next_v = v[u];
        }
    }
    if (!live) m = Mom{0.0f, 0.0f, 0.0f};
    m = mom_wave(m);
    if (threadIdx.x == 0) {
        scratch[3 * blockIdx.x + 0] = m.n;
        scratch[3 * blockIdx.x + 1] = m.mean;
        scratch[3 * blockIdx.x + 2] = m.m2;
    }
}

extern "C" int imx_gae(int64_t T, int64_t N, const float* rew, const float* val, const uint8_t* dones,
                       const float* last_val, float gamma, float lam, int normalize, float* ret, float* adv,
                       void* scratch, imx_stream_t stream) {
    IMX_REQUIRE(T > 0 && N > 0, "imx_gae: empty rollout (T=%lld N=%lld)", (long long)T, (long long)N);
    IMX_REQUIRE(rew && val && dones && last_val && ret && adv && scratch, "imx_gae: null argument");
    const unsigned nblk = (unsigned)((N + GAE_BLOCK - 1) / GAE_BLOCK);
    if (T <= GAE_TMAX)
        hipLaunchKernelGGL(k_gae_short, dim3(nblk), dim3(GAE_BLOCK), 0, (hipStream_t)stream, (int)T, N, rew, val, dones, last_val, gamma, lam,
                           ret, adv, (float*)scratch);
    else
        hipLaunchKernelGGL(k_gae, dim3(nblk), dim3(GAE_BLOCK), 0, (hipStream_t)stream, T, N, rew, val, dones, last_val, gamma, lam,
                           ret, adv, (float*)scratch);
    IMX_HIP(hipGetLastError());
    if (normalize) {
        const int64_t total = T * N;
        const unsigned grid = (unsigned)std::min<int64_t>((total / 4 + 255) / 256 + 1, 1024);
        hipLaunchKernelGGL(k_adv_normalize, dim3(grid), dim3(256), 0, (hipStream_t)stream, total, (int)nblk,
                           (const float*)scratch, adv);
        IMX_HIP(hipGetLastError());
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------- PPO loss
// Per sample i (minibatch of M), action dim A, sigma broadcast over the batch when sigma_stride == 0:
//   logp   = sum_a( -(a-mu)^2/(2 s^2) - log s - 0.5 log(2 pi) )          Normal.log_prob(...).sum(-1)
//   ent    = sum_a( 0.5 + 0.5 log(2 pi) + log s )                         Normal.entropy().sum(-1)
//   kl     = sum_a( log(s/so + 1e-5) + (so^2 + (muo-mu)^2)/(2 s^2) - 0.5 )
//   ratio  = exp(logp - old_logp); surrogate = max(-adv*ratio, -adv*clamp(ratio, 1-c, 1+c))
//   value  = clipped: max((v-R)^2, (vo + clamp(v-vo,-c,c) - R)^2) ; else (R-v)^2
// out4 = means over the minibatch.

extern "C" size_t imx_ppo_scratch_bytes(int64_t M) { return ((size_t)((M + 255) / 256) * 4 + 8) * sizeof(float); }

__global__ void __launch_bounds__(256)
k_ppo_fwd(int64_t M, int A, const float* __restrict__ mu, const float* __restrict__ sigma, int sigma_stride,
          const float* __restrict__ act, const float* __restrict__ old_logp, const float* __restrict__ old_mu,
          const float* __restrict__ old_sigma, const float* __restrict__ adv, const float* __restrict__ ret,
          const float* __restrict__ val, const float* __restrict__ old_val, float clip, int clipped_value,
          float* __restrict__ part) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float sur = 0.0f, vl = 0.0f, ent = 0.0f, kl = 0.0f;
    if (i < M) {
        float logp = 0.0f;
        for (int a = 0; a < A; ++a) {
            const float m = mu[i * A + a], s = sigma[i * sigma_stride + a];
            const float d = act[i * A + a] - m;
            const float ls = logf(s);
            logp += -(d * d) / (2.0f * s * s) - ls - IMX_HALF_LOG_2PI;
            ent += 0.5f + IMX_HALF_LOG_2PI + ls;
            if (old_mu) {
                const float mo = old_mu[i * A + a], so = old_sigma[i * A + a];
                const float dm = mo - m;
                kl += logf(s / so + 1.0e-5f) + (so * so + dm * dm) / (2.0f * s * s) - 0.5f;
            }
        }
        const float ratio = expf(logp - old_logp[i]);
        const float ad = adv[i];
        const float s1 = -ad * ratio, s2 = -ad * fminf(fmaxf(ratio, 1.0f - clip), 1.0f + clip);
        sur = fmaxf(s1, s2);
        const float v = val[i], R = ret[i];
        if (clipped_value) {
            const float vo = old_val[i];
            const float vc = vo + fminf(fmaxf(v - vo, -clip), clip);
            const float l1 = (v - R) * (v - R), l2 = (vc - R) * (vc - R);
            vl = fmaxf(l1, l2);
        } else {
            vl = (R - v) * (R - v);
        }
    }
    __shared__ float sm[4][4];
    float r4[4] = {sur, vl, ent, kl};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float x = r4[q];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
        if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6][q] = x;
    }
    __syncthreads();
    if (threadIdx.x < 4) part[4 * blockIdx.x + threadIdx.x] = (sm[0][threadIdx.x] + sm[1][threadIdx.x]) + (sm[2][threadIdx.x] + sm[3][threadIdx.x]);
}

// out8 = {surrogate, value_loss, entropy_mean, kl_mean, loss, 0, 0, 0}; accum (optional, 5 floats) += {value_loss,
// surrogate, entropy, kl, 1} (the running means rsl_rl logs per update)
__global__ void k_ppo_finish(int nblk, float inv_m, float vcoef, float ecoef, const float* __restrict__ part,
                             float* __restrict__ out8, float* __restrict__ accum) {
    // 64 lanes: lane = (chunk of blocks, quantity); chunks combined in a fixed order (deterministic)
    __shared__ float red[16][4];
    __shared__ float r[4];
    const int q = threadIdx.x & 3, chunk = threadIdx.x >> 2;
    float s = 0.0f;
    for (int b = chunk; b < nblk; b += 16) s += part[4 * b + q];
    red[chunk][q] = s;
    __syncthreads();
    if (threadIdx.x < 4) {
        float tot = 0.0f;
        for (int c = 0; c < 16; ++c) tot += red[c][threadIdx.x];
        r[threadIdx.x] = tot * inv_m;
        out8[threadIdx.x] = r[threadIdx.x];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        out8[4] = r[0] + vcoef * r[1] - ecoef * r[2];
        if (accum) {
            accum[0] += r[1]; accum[1] += r[0]; accum[2] += r[2]; accum[3] += r[3]; accum[4] += 1.0f;
        }
    }
}

// d/d(mu, sigma, value) of  mean(sur) + vcoef*mean(vl) - ecoef*mean(ent), times grad_scale (the upstream grad)
__global__ void __launch_bounds__(256)
k_ppo_bwd(int64_t M, int A, const float* __restrict__ mu, const float* __restrict__ sigma, int sigma_stride,
          const float* __restrict__ act, const float* __restrict__ old_logp, const float* __restrict__ adv,
          const float* __restrict__ ret, const float* __restrict__ val, const float* __restrict__ old_val, float clip,
          int clipped_value, float vcoef, float ecoef, float gscale, float* __restrict__ dmu,
          float* __restrict__ dsigma, float* __restrict__ dval) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    const float inv_m = gscale / (float)M;
    if (dmu) {  // policy part (the value part below is independent: the two may be launched separately)
        float logp = 0.0f;
        for (int a = 0; a < A; ++a) {
            const float m = mu[i * A + a], s = sigma[i * sigma_stride + a];
            const float d = act[i * A + a] - m;
            logp += -(d * d) / (2.0f * s * s) - logf(s) - IMX_HALF_LOG_2PI;
        }
        const float ratio = expf(logp - old_logp[i]);
        const float ad = adv[i];
        const float s1 = -ad * ratio, s2 = -ad * fminf(fmaxf(ratio, 1.0f - clip), 1.0f + clip);
        // torch.max picks the first argument's gradient on ties? (max backward splits evenly on ties; ties only when
        // ratio is inside the clip range where both branches have the same derivative -ad*ratio)
        float dsur_dlogp;
        if (s1 >= s2) dsur_dlogp = -ad * ratio;
        else dsur_dlogp = (ratio > 1.0f - clip && ratio < 1.0f + clip) ? -ad * ratio : 0.0f;
        const float g = dsur_dlogp * inv_m;
        for (int a = 0; a < A; ++a) {
            const float m = mu[i * A + a], s = sigma[i * sigma_stride + a];
            const float d = act[i * A + a] - m;
            dmu[i * A + a] = g * (d / (s * s));
            // dlogp/ds = d^2/s^3 - 1/s ; dent/ds = 1/s
            dsigma[i * A + a] = g * ((d * d) / (s * s * s) - 1.0f / s) - ecoef * inv_m / s;
        }
    }
    if (!dval) return;
    const float v = val[i], R = ret[i];
    float dv;
    if (clipped_value) {
        const float vo = old_val[i];
        const float dlt = v - vo;
        const float vc = vo + fminf(fmaxf(dlt, -clip), clip);
        const float l1 = (v - R) * (v - R), l2 = (vc - R) * (vc - R);
        if (l1 >= l2) dv = 2.0f * (v - R);
        else dv = (dlt > -clip && dlt < clip) ? 2.0f * (vc - R) : 0.0f;
    } else {
        dv = -2.0f * (R - v);
    }
    dval[i] = vcoef * dv * inv_m;
}

extern "C" int imx_ppo_loss_fwd(int64_t M, int64_t A, const float* mu, const float* sigma, int64_t sigma_stride,
                                const float* act, const float* old_logp, const float* old_mu, const float* old_sigma,
                                const float* adv, const float* ret, const float* val, const float* old_val, float clip,
                                int clipped_value, float vcoef, float ecoef, float* out8, float* accum, void* scratch,
                                imx_stream_t stream) {
    IMX_REQUIRE(M > 0 && A > 0 && A <= 4096, "imx_ppo_loss_fwd: bad sizes M=%lld A=%lld", (long long)M, (long long)A);
    IMX_REQUIRE(sigma_stride == 0 || sigma_stride == A, "imx_ppo_loss_fwd: sigma_stride must be 0 (shared std) or A");
    IMX_REQUIRE(mu && sigma && act && old_logp && adv && ret && val && out8 && scratch, "imx_ppo_loss_fwd: null argument");
    IMX_REQUIRE(!clipped_value || old_val, "imx_ppo_loss_fwd: clipped value loss needs old values");
    IMX_REQUIRE((old_mu == nullptr) == (old_sigma == nullptr), "imx_ppo_loss_fwd: old_mu/old_sigma must come together");
    const unsigned nblk = (unsigned)((M + 255) / 256);
    hipLaunchKernelGGL(k_ppo_fwd, dim3(nblk), dim3(256), 0, (hipStream_t)stream, M, (int)A, mu, sigma, (int)sigma_stride, act,
                       old_logp, old_mu, old_sigma, adv, ret, val, old_val, clip, clipped_value, (float*)scratch);
    hipLaunchKernelGGL(k_ppo_finish, dim3(1), dim3(64), 0, (hipStream_t)stream, (int)nblk, 1.0f / (float)M, vcoef, ecoef,
                       (const float*)scratch, out8, accum);
    IMX_HIP(hipGetLastError());
    return 0;
}

extern "C" int imx_ppo_loss_bwd(int64_t M, int64_t A, const float* mu, const float* sigma, int64_t sigma_stride,
                                const float* act, const float* old_logp, const float* adv, const float* ret, const float* val,
                                const float* old_val, float clip, int clipped_value, float vcoef, float ecoef, float gscale,
                                float* dmu, float* dsigma, float* dval, imx_stream_t stream) {
    IMX_REQUIRE(M > 0 && A > 0, "imx_ppo_loss_bwd: bad sizes");
    IMX_REQUIRE(sigma_stride == 0 || sigma_stride == A, "imx_ppo_loss_bwd: sigma_stride must be 0 (shared std) or A");
    IMX_REQUIRE(dmu || dval, "imx_ppo_loss_bwd: nothing to compute (dmu and dval are both null)");
    IMX_REQUIRE(!dmu || (mu && sigma && act && old_logp && adv && dsigma), "imx_ppo_loss_bwd: the policy part needs mu, sigma, "
                "actions, old log-prob, advantages and dsigma");
    IMX_REQUIRE(!dval || (ret && val), "imx_ppo_loss_bwd: the value part needs returns and values");
    IMX_REQUIRE(!dval || !clipped_value || old_val, "imx_ppo_loss_bwd: clipped value loss needs old values");
    hipLaunchKernelGGL(k_ppo_bwd, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream, M, (int)A, mu, sigma,
                       (int)sigma_stride, act, old_logp, adv, ret, val, old_val, clip, clipped_value, vcoef, ecoef, gscale, dmu,
                       dsigma, dval);
    IMX_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------- Adam
// torch.nn.utils.clip_grad_norm_ + torch.optim.Adam.step on ONE flat parameter bucket, with the learning rate and
// the gradient norm read from device memory (no host round trip for the adaptive-KL schedule).
//   clip_coef = min(1, max_norm / (norm + 1e-6));  g *= clip_coef
//   m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g*g ; p -= lr/bias1 * m / (sqrt(v)/sqrt(bias2) + eps)
__global__ void __launch_bounds__(256)
k_adam(int64_t n, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
       const float* __restrict__ lr_d, const float* __restrict__ norm_d, float max_norm, float b1, float b2, float eps,
       float bias1, float sqrt_bias2) {
    const float lr = lr_d[0];
    float coef = 1.0f;
    if (norm_d) coef = fminf(max_norm / (norm_d[0] + 1.0e-6f), 1.0f);
    const float step_size = lr / bias1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float gi = g[i] * coef;
        const float mi = b1 * m[i] + (1.0f - b1) * gi;
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] = p[i] - step_size * (mi / (sqrtf(vi) / sqrt_bias2 + eps));
    }
}

extern "C" int imx_adam_step(int64_t n, float* p, const float* g, float* m, float* v, const float* lr_d,
                             const float* grad_norm_d, float max_norm, float beta1, float beta2, float eps,
                             int64_t step, imx_stream_t stream) {
    IMX_REQUIRE(n > 0 && p && g && m && v && lr_d && step > 0, "imx_adam_step: bad arguments");
    const float bias1 = 1.0f - powf(beta1, (float)step);
    const float sb2 = sqrtf(1.0f - powf(beta2, (float)step));
    const unsigned grid = (unsigned)std::min<int64_t>((n + 255) / 256, 2048);
    hipLaunchKernelGGL(k_adam, dim3(grid), dim3(256), 0, (hipStream_t)stream, n, p, g, m, v, lr_d, grad_norm_d, max_norm,
                       beta1, beta2, eps, bias1, sb2);
    IMX_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------- Adam, device-side schedule
// state (8 floats, device): [0] lr  [1] step count  [2] beta1^t  [3] beta2^t  [4] clip coef  [5] lr/bias1  [6] sqrt(bias2)
// k_adam_prepare (one thread) = rsl_rl's adaptive-KL learning-rate rule + Adam step bookkeeping + clip_grad_norm_'s
// coefficient, all from device scalars: nothing of the schedule lives on the host, so the whole update is capturable.
__global__ void k_adam_prepare(float* __restrict__ st, const float* __restrict__ kl_d, float desired_kl,
                               const float* __restrict__ norm_d, float max_norm, float b1, float b2) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float lr = st[0];
    if (kl_d) {  // PPO.update 'adaptive': lr /= 1.5 if kl > 2*desired; lr *= 1.5 if 0 < kl < desired/2
        const float kl = kl_d[0];
        if (kl > desired_kl * 2.0f) lr = fmaxf(1.0e-5f, lr / 1.5f);
        else if (kl < desired_kl / 2.0f && kl > 0.0f) lr = fminf(1.0e-2f, lr * 1.5f);
    }
    const float p1 = st[2] * b1, p2 = st[3] * b2;
    st[0] = lr;
    st[1] += 1.0f;
    st[2] = p1;
    st[3] = p2;
    st[4] = norm_d ? fminf(max_norm / (norm_d[0] + 1.0e-6f), 1.0f) : 1.0f;
    st[5] = lr / (1.0f - p1);
    st[6] = sqrtf(1.0f - p2);
}

// clip_grad_norm_'s total norm + the schedule above in ONE launch: every block sums the squares of its slice in a fixed
// order, the last block to finish (agent-scope release / acquire around a ticket) adds the partials in block order and
// runs k_adam_prepare's arithmetic.  Replaces torch.linalg.vector_norm (a fill + a reduce kernel) + k_adam_prepare.
// scratch: [0, G) partial sums, [G] ticket (must be zero on entry; the last block re-arms it).
constexpr int ADAM_NORM_BLOCK = 256, ADAM_NORM_PER_THREAD = 16;
__global__ void __launch_bounds__(ADAM_NORM_BLOCK)
k_adam_norm_prepare(int64_t n, const float* __restrict__ g, float* __restrict__ st, const float* __restrict__ kl_d, float desired_kl,
                    float max_norm, float b1, float b2, float* __restrict__ scratch) {
    __shared__ float red[ADAM_NORM_BLOCK];
    __shared__ bool last;
    const int G = gridDim.x;
    const int64_t base = (int64_t)blockIdx.x * ADAM_NORM_BLOCK * ADAM_NORM_PER_THREAD;
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < ADAM_NORM_PER_THREAD; ++k) {
        const int64_t i = base + (int64_t)k * ADAM_NORM_BLOCK + threadIdx.x;
        if (i < n) s += g[i] * g[i];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = ADAM_NORM_BLOCK / 2; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    unsigned int* ticket = reinterpret_cast<unsigned int*>(scratch + G);
    if (threadIdx.x == 0) {
        scratch[blockIdx.x] = red[0];
        __atomic_thread_fence(__ATOMIC_RELEASE);  // agent scope: the partial is visible before the ticket
        const unsigned int t = atomicAdd(ticket, 1u);
        last = (t == (unsigned int)G - 1u);
        if (last) __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    __syncthreads();
    if (!last) return;
    // last block: all lanes fetch partials in parallel (a single lane walking them pays one uncached round trip each),
    // then the same fixed-order tree as above
    float part = 0.0f;
    for (int b = threadIdx.x; b < G; b += ADAM_NORM_BLOCK) part += __builtin_nontemporal_load(scratch + b);
    red[threadIdx.x] = part;
    __syncthreads();
    for (int w = ADAM_NORM_BLOCK / 2; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    const float tot = red[0];
    *ticket = 0u;
    const float norm = sqrtf(tot);
    float lr = st[0];
    if (kl_d) {
        const float kl = kl_d[0];
        if (kl > desired_kl * 2.0f) lr = fmaxf(1.0e-5f, lr / 1.5f);
        else if (kl < desired_kl / 2.0f && kl > 0.0f) lr = fminf(1.0e-2f, lr * 1.5f);
    }
    const float p1 = st[2] * b1, p2 = st[3] * b2;
    st[0] = lr;
    st[1] += 1.0f;
    st[2] = p1;
    st[3] = p2;
    st[4] = max_norm > 0.0f ? fminf(max_norm / (norm + 1.0e-6f), 1.0f) : 1.0f;
    st[5] = lr / (1.0f - p1);
    st[6] = sqrtf(1.0f - p2);
    st[7] = norm;
}

__global__ void __launch_bounds__(256)
k_adam_apply(int64_t n, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
             const float* __restrict__ st, float b1, float b2, float eps) {
    const float coef = st[4], step_size = st[5], sqrt_bias2 = st[6];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float gi = g[i] * coef;
        const float mi = b1 * m[i] + (1.0f - b1) * gi;
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] = p[i] - step_size * (mi / (sqrtf(vi) / sqrt_bias2 + eps));
    }
}

extern "C" int imx_adam_update(int64_t n, float* p, const float* g, float* m, float* v, float* state8, const float* kl_d,
                               float desired_kl, const float* grad_norm_d, float max_norm, float beta1, float beta2,
                               float eps, imx_stream_t stream) {
    IMX_REQUIRE(n > 0 && p && g && m && v && state8, "imx_adam_update: bad arguments");
    hipLaunchKernelGGL(k_adam_prepare, dim3(1), dim3(64), 0, (hipStream_t)stream, state8, kl_d, desired_kl, grad_norm_d,
                       max_norm, beta1, beta2);
    const unsigned grid = (unsigned)std::min<int64_t>((n + 255) / 256, 2048);
    hipLaunchKernelGGL(k_adam_apply, dim3(grid), dim3(256), 0, (hipStream_t)stream, n, p, g, m, v, (const float*)state8, beta1,
                       beta2, eps);
    IMX_HIP(hipGetLastError());
    return 0;
}


static int adam_norm_blocks(int64_t n) {
    return (int)((n + (int64_t)ADAM_NORM_BLOCK * ADAM_NORM_PER_THREAD - 1) / ((int64_t)ADAM_NORM_BLOCK * ADAM_NORM_PER_THREAD));
}

extern "C" size_t imx_adam_norm_scratch_bytes(int64_t n) { return n > 0 ? ((size_t)adam_norm_blocks(n) + 1) * sizeof(float) : 0; }

extern "C" int imx_adam_update_norm(int64_t n, float* p, const float* g, float* m, float* v, float* state8, const float* kl_d,
                                    float desired_kl, float max_norm, float beta1, float beta2, float eps, void* scratch_d,
                                    size_t scratch_bytes, imx_stream_t stream) {
    IMX_REQUIRE(n > 0 && p && g && m && v && state8 && scratch_d, "imx_adam_update_norm: bad arguments");
    const int G = adam_norm_blocks(n);
    IMX_REQUIRE(scratch_bytes >= ((size_t)G + 1) * sizeof(float), "imx_adam_update_norm: scratch too small (see imx_adam_norm_scratch_bytes)");
    hipLaunchKernelGGL(k_adam_norm_prepare, dim3((unsigned)G), dim3(ADAM_NORM_BLOCK), 0, (hipStream_t)stream, n, g, state8, kl_d, desired_kl,
                       max_norm, beta1, beta2, (float*)scratch_d);
    const unsigned grid = (unsigned)std::min<int64_t>((n + 255) / 256, 2048);
    hipLaunchKernelGGL(k_adam_apply, dim3(grid), dim3(256), 0, (hipStream_t)stream, n, p, g, m, v, (const float*)state8, beta1,
                       beta2, eps);
    IMX_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------- minibatch gather
// RolloutStorage.mini_batch_generator: obs[idx], actions[idx], values[idx], ... -- one launch for all (<= 12) arrays.
// One wave per sample row (two rows in flight): idx[r] is read once per row, the array loop is uniform (no per-element
// division, no per-lane pointer table look-up), lanes run along the row so loads and stores are contiguous.
struct GatherArgs {
    const float* src[12];
    float* dst[12];
    int width[12];
    int dpitch[12];  // row pitch of dst in floats (>= width)
    int offset[13];  // prefix sums of width
    int n;
};
__global__ void __launch_bounds__(256) k_gather_rows(int64_t M, const int64_t* __restrict__ idx, GatherArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t r0 = 2 * wave; r0 < M; r0 += 2 * nw) {
        const int64_t r1 = r0 + 1;
        const bool two = r1 < M;
        const int64_t s0 = idx[r0], s1 = two ? idx[r1] : 0;
        for (int k = 0; k < a.n; ++k) {
            const int w = a.width[k], dp = a.dpitch[k];
            const float* __restrict__ src = a.src[k];
            float* __restrict__ dst = a.dst[k];
            for (int c = lane; c < w; c += 64) {
                const float v0 = src[s0 * w + c];
                const float v1 = two ? src[s1 * w + c] : 0.0f;
                dst[r0 * dp + c] = v0;
                if (two) dst[r1 * dp + c] = v1;
            }
        }
    }
}

extern "C" int imx_gather_rows_pitched(int64_t M, const int64_t* idx_d, int n, const void* const* src_d, void* const* dst_d,
                                       const int32_t* width_floats, const int32_t* dst_pitch_floats, imx_stream_t stream) {
    IMX_REQUIRE(M > 0 && idx_d && n > 0 && n <= 12 && src_d && dst_d && width_floats, "imx_gather_rows: bad arguments");
    GatherArgs a;
    a.n = n;
    a.offset[0] = 0;
    for (int k = 0; k < 12; ++k) {
        a.src[k] = k < n ? (const float*)src_d[k] : nullptr;
        a.dst[k] = k < n ? (float*)dst_d[k] : nullptr;
        a.width[k] = k < n ? width_floats[k] : 0;
        a.dpitch[k] = k < n ? (dst_pitch_floats ? dst_pitch_floats[k] : width_floats[k]) : 0;
        IMX_REQUIRE(k >= n || (a.src[k] && a.dst[k] && a.width[k] > 0), "imx_gather_rows: array %d is null or empty", k);
        IMX_REQUIRE(k >= n || a.dpitch[k] >= a.width[k], "imx_gather_rows: array %d: destination pitch %d < width %d", k, a.dpitch[k], a.width[k]);
        a.offset[k + 1] = a.offset[k] + a.width[k];
    }
    const unsigned grid = (unsigned)std::min<int64_t>((M + 7) / 8, 4096);  // 4 waves per block, 2 rows per wave and trip
    hipLaunchKernelGGL(k_gather_rows, dim3(grid), dim3(256), 0, (hipStream_t)stream, M, idx_d, a);
    IMX_HIP(hipGetLastError());
    return 0;
}

extern "C" int imx_gather_rows(int64_t M, const int64_t* idx_d, int n, const void* const* src_d, void* const* dst_d,
                               const int32_t* width_floats, imx_stream_t stream) {
    return imx_gather_rows_pitched(M, idx_d, n, src_d, dst_d, width_floats, nullptr, stream);
}

// ------------------------------------------------------------------------------------------------- rollout step fusions
// imx_policy_act = PPO.act after the two MLPs (upstream ppo.py::act + actor_critic.py::act/get_actions_log_prob):
//   a = mu + std * eps, eps ~ N(0,1)   (distribution.sample(); in-kernel counter-based Box-Muller)
//   log_prob = sum_a( -(a-mu)^2/(2 std^2) - log std - 0.5 log 2pi )
// and the write of the transition (obs, actions, log-prob, mu, sigma, value) straight into slot t of the storage.
__global__ void __launch_bounds__(256)
k_policy_act(int64_t N, int A, int D, const float* __restrict__ mu, const float* __restrict__ std_a,
             const float* __restrict__ value, const float* __restrict__ obs, uint64_t seed,
             const int32_t* __restrict__ step_d, float* __restrict__ act_out, float* __restrict__ logp_out,
             float* __restrict__ mu_out, float* __restrict__ sigma_out, float* __restrict__ val_out,
             float* __restrict__ obs_out, float* __restrict__ act_env) {
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nth = (int64_t)gridDim.x * blockDim.x;
    const uint32_t step = step_d ? (uint32_t)step_d[0] : 0u;
    for (int64_t e = tid; e < N; e += nth) {
        float logp = 0.0f;
        for (int a = 0; a < A; ++a) {
            const float m = mu[e * A + a], s = std_a[a];
            const float u1 = 1.0f - uniform01(seed, step, (uint64_t)(e * A + a) * 2);        // (0,1]
            const float u2 = uniform01(seed ^ 0x5851F42D4C957F2Dull, step, (uint64_t)(e * A + a) * 2 + 1);
            const float z = sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
            const float x = m + s * z;
            const float d = x - m;
            logp += -(d * d) / (2.0f * s * s) - logf(s) - IMX_HALF_LOG_2PI;
            act_out[e * A + a] = x;
            if (act_env) act_env[e * A + a] = x;
            mu_out[e * A + a] = m;
            sigma_out[e * A + a] = s;
        }
        logp_out[e] = logp;
        val_out[e] = value[e];
    }
    const int64_t nobs = N * D;
    for (int64_t i = tid; i < nobs; i += nth) obs_out[i] = obs[i];
}

extern "C" int imx_policy_act(int64_t N, int64_t A, int64_t D, const float* mu_d, const float* std_d, const float* value_d,
                              const float* obs_d, uint64_t seed, const int32_t* step_counter_d, float* actions_out_d,
                              float* logp_out_d, float* mu_out_d, float* sigma_out_d, float* values_out_d,
                              float* obs_out_d, float* actions_env_d, imx_stream_t stream) {
    IMX_REQUIRE(N > 0 && A > 0 && D > 0, "imx_policy_act: bad sizes");
    IMX_REQUIRE(mu_d && std_d && value_d && obs_d && actions_out_d && logp_out_d && mu_out_d && sigma_out_d &&
                    values_out_d && obs_out_d, "imx_policy_act: null argument");
    const unsigned grid = (unsigned)std::min<int64_t>((N * D + 255) / 256, 4096);
    hipLaunchKernelGGL(k_policy_act, dim3(grid), dim3(256), 0, (hipStream_t)stream, N, (int)A, (int)D, mu_d, std_d, value_d,
                       obs_d, seed, step_counter_d, actions_out_d, logp_out_d, mu_out_d, sigma_out_d, values_out_d, obs_out_d,
                       actions_env_d);
    IMX_HIP(hipGetLastError());
    return 0;
}

// imx_rollout_post = RslRlVecEnvWrapper.step's dones + PPO.process_env_step's time-out bootstrap + the runner's
// episode book-keeping, writing slot t of the storage:
//   dones = terminated | truncated;  rewards_t = rew + gamma * value_t * time_out;  episode sums / lengths / counts.
// ep_stats (3 floats) accumulates finished episodes' {sum reward, sum length, count} with float atomics (logging only).
__global__ void __launch_bounds__(256)
k_rollout_post(int64_t N, const float* __restrict__ rew, const uint8_t* __restrict__ terminated,
               const uint8_t* __restrict__ truncated, const float* __restrict__ value_t, float gamma, int bootstrap,
               float* __restrict__ rew_out, uint8_t* __restrict__ dones_out, int64_t* __restrict__ dones_long,
               float* __restrict__ cur_rew, float* __restrict__ cur_len, float* __restrict__ ep_stats,
               const float* __restrict__ log_in, float* __restrict__ log_accum, int nlog) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // the runner's ep_infos (upstream on_policy_runner.py: infos["log"] appended every step, averaged per iteration): the env's
    // Episode_* entries of this step are added to a running sum by ONE workgroup (fixed order: deterministic)
    if (log_accum && blockIdx.x == 0)
        for (int i = threadIdx.x; i < nlog; i += blockDim.x) log_accum[i] += log_in[i];
    float s_r = 0.0f, s_l = 0.0f, s_c = 0.0f;
    if (e < N) {
        const bool to = truncated[e] != 0, done = to || terminated[e] != 0;
        const float r = rew[e];
        rew_out[e] = bootstrap ? r + gamma * (value_t[e] * (to ? 1.0f : 0.0f)) : r;
        dones_out[e] = done ? 1 : 0;
        if (dones_long) dones_long[e] = done ? 1 : 0;
        if (cur_rew) {
            const float cr = cur_rew[e] + r, cl = cur_len[e] + 1.0f;
            if (done) { s_r = cr; s_l = cl; s_c = 1.0f; }
            cur_rew[e] = done ? 0.0f : cr;
            cur_len[e] = done ? 0.0f : cl;
        }
    }
    if (ep_stats) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            s_r += __shfl_xor(s_r, o, 64); s_l += __shfl_xor(s_l, o, 64); s_c += __shfl_xor(s_c, o, 64);
        }
        if ((threadIdx.x & 63) == 0 && s_c > 0.0f) {
            atomicAdd(&ep_stats[0], s_r); atomicAdd(&ep_stats[1], s_l); atomicAdd(&ep_stats[2], s_c);
        }
    }
}

extern "C" int imx_rollout_post(int64_t N, const float* reward_d, const uint8_t* terminated_d, const uint8_t* truncated_d,
                                const float* value_t_d, float gamma, int bootstrap_time_outs, float* rewards_out_d,
                                uint8_t* dones_out_d, int64_t* dones_long_d, float* cur_reward_sum_d, float* cur_ep_len_d,
                                float* ep_stats3_d, const float* log_in_d, float* log_accum_d, int num_log, imx_stream_t stream) {
    IMX_REQUIRE(N > 0 && reward_d && terminated_d && truncated_d && value_t_d && rewards_out_d && dones_out_d,
                "imx_rollout_post: bad arguments");
    IMX_REQUIRE(!log_accum_d || (log_in_d && num_log > 0), "imx_rollout_post: log accumulation needs log_in_d and num_log > 0");
    IMX_REQUIRE((cur_reward_sum_d == nullptr) == (cur_ep_len_d == nullptr), "imx_rollout_post: episode buffers come together");
    hipLaunchKernelGGL(k_rollout_post, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, N, reward_d,
                       terminated_d, truncated_d, value_t_d, gamma, bootstrap_time_outs, rewards_out_d, dones_out_d,
                       dones_long_d, cur_reward_sum_d, cur_ep_len_d, ep_stats3_d, log_in_d, log_accum_d, num_log);
    IMX_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------- column sums
// out[c] = sum_m x[m][c] (* y[m][c]): the gradient of the action-noise parameter from the per-sample dsigma (M x A, A <= 64).
// torch.sum(x, dim=0) on a 24576 x 37 tensor is one 320 us launch (its reduction kernel is laid out for long rows); here
// COLSUM_BLOCKS workgroups each reduce a slab of rows -- a wave reads whole rows, lane = column, so loads are contiguous -- and
// the last-arriving workgroup is NOT used: a second 64-thread launch adds the per-workgroup partials in a fixed order
// (deterministic, and the two launches together are ~10 us).
#define COLSUM_BLOCKS 256
__global__ void __launch_bounds__(256)
k_colsum_part(int64_t M, int A, const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ part) {
    __shared__ float s[4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t rows = (M + COLSUM_BLOCKS - 1) / COLSUM_BLOCKS;
    const int64_t r0 = blockIdx.x * rows, r1 = min(r0 + rows, M);
    float acc = 0.0f;
    if (lane < A)
        for (int64_t r = r0 + wv; r < r1; r += 4) acc += y ? x[r * A + lane] * y[r * A + lane] : x[r * A + lane];
    s[wv][lane] = acc;
    __syncthreads();
    if (wv == 0 && lane < A) part[(size_t)blockIdx.x * 64 + lane] = (s[0][lane] + s[1][lane]) + (s[2][lane] + s[3][lane]);
}
__global__ void __launch_bounds__(64) k_colsum_final(int A, const float* __restrict__ part, float* __restrict__ out) {
    const int c = threadIdx.x;
    if (c >= A) return;
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
    for (int b = 0; b < COLSUM_BLOCKS; b += 4) {
        a0 += part[(size_t)b * 64 + c]; a1 += part[(size_t)(b + 1) * 64 + c];
        a2 += part[(size_t)(b + 2) * 64 + c]; a3 += part[(size_t)(b + 3) * 64 + c];
    }
    out[c] = (a0 + a1) + (a2 + a3);
}

extern "C" size_t imx_colsum_scratch_bytes(void) { return (size_t)COLSUM_BLOCKS * 64 * sizeof(float); }

extern "C" int imx_colsum(int64_t M, int64_t A, const float* x_d, const float* y_d, float* out_d, void* scratch_d,
                          imx_stream_t stream) {
    IMX_REQUIRE(M > 0 && A > 0 && A <= 64 && x_d && out_d && scratch_d, "imx_colsum: bad arguments (A <= 64)");
    float* part = reinterpret_cast<float*>(scratch_d);
    hipLaunchKernelGGL(k_colsum_part, dim3(COLSUM_BLOCKS), dim3(256), 0, (hipStream_t)stream, M, (int)A, x_d, y_d, part);
    hipLaunchKernelGGL(k_colsum_final, dim3(1), dim3(64), 0, (hipStream_t)stream, (int)A, part, out_d);
    IMX_HIP(hipGetLastError());
    return 0;
}
