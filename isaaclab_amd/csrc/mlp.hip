// Actor / critic MLP backward + policy-head kernels of the PPO minibatch step (fp32 in, fp32 accumulate on the f32 MFMA).
//
// Replaces, inside rsl_rl PPO.update's `loss.backward()` (upstream rsl_rl/algorithms/ppo.py::update, autograd of the
// nn.Linear / nn.ELU stack built by rsl_rl/modules/actor_critic.py), the three pieces a library GEMM does badly here:
//   * imx_mlp_dw        dW = dY^T X, db = colsum(dY): a reduction over the M = 24576 samples of a minibatch into a
//                       small (out x in) matrix -- split over M across all CUs, partials summed in a fixed order;
//   * imx_mlp_head_fwd  the 128 -> 12 / 128 -> 1 output layer (a skinny GEMM: 4 lanes per sample, VALU);
//   * imx_mlp_head_bwd  the same layer backward: dW, db, dX and the ELU' of the layer below in one pass over the
//                       activations (the accumulator tile of the dX MFMA and the B operand of the dW MFMA share lanes).
// The wide forward / dX GEMMs stay in the library (hipBLASLt, 100-130 TFLOP/s at these shapes; a hand-written fused
// Linear+ELU forward on the same skeleton measured 52-78 TFLOP/s -- slower than library GEMM + separate ELU pass -- and
// was dropped).
// v_mfma_f32_32x32x2_f32 is an exact f32 fma chain (one rounding per product): numerics = an fp32 GEMM with this
// summation order.  PARITY UNPINNED like the rest of the rsl_rl restatement (checked against torch autograd in tests).
#include <algorithm>

#include "imx_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int DW_T = 128;          // output tile: 128 out-features x 128 in-features per workgroup (4 waves, 64x64 each)
constexpr int DW_BM = 32;          // samples per stage
constexpr int DW_LD = DW_T + 32;   // LDS row pitch (floats): the two half-waves (sample m, m+1) land on disjoint banks

struct DwArgs {
    int64_t M;
    int N, K;
    const float* dY;
    int64_t ldy;
    const float* X;
    int64_t ldx;
    float* part;     // [S][N][K]
    float* part_db;  // [S][N] or null
    int tn, tk, S;
    int64_t rps;     // rows (samples) per split, multiple of DW_BM
    // ACT variant: dY holds the gradient w.r.t. the layer's ACTIVATED output; the movers multiply it by ELU'(H) (H = the saved
    // output, aten elu_backward with is_result) on its way into LDS and the first column tile writes the product to Dout
    const float* H;
    int64_t ldh;
    float* Dout;
    int64_t ldd;
    float alpha;
};

// row of a 32x32 MFMA accumulator register: lanes 0-31 hold rows {0-3, 8-11, 16-19, 24-27}, lanes 32-63 the others
__device__ __forceinline__ int acc_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

// one [32 x 128] stage of a row-major matrix -> 16 registers per thread (zero beyond row_end / ncols)
template <bool VEC>
__device__ __forceinline__ void dw_load(float (&r)[16], const float* __restrict__ src, int64_t ld, int64_t row0, int64_t row_end,
                                        int col0, int ncols, int t) {
    if (VEC) {
        const int c = col0 + 4 * (t & 31);
        const int rr = t >> 5;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t row = row0 + rr + 8 * i;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < row_end && c < ncols) v = *reinterpret_cast<const float4*>(src + row * ld + c);
            r[4 * i] = v.x; r[4 * i + 1] = v.y; r[4 * i + 2] = v.z; r[4 * i + 3] = v.w;
        }
    } else {
        const int c = col0 + (t & 127);
        const int rr = t >> 7;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int64_t row = row0 + rr + 2 * i;
            r[i] = (row < row_end && c < ncols) ? src[row * ld + c] : 0.0f;
        }
    }
}

template <bool VEC>
__device__ __forceinline__ void dw_store(const float (&r)[16], float* __restrict__ s, int t) {
    if (VEC) {
        const int c = 4 * (t & 31), rr = t >> 5;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *reinterpret_cast<float4*>(s + (rr + 8 * i) * DW_LD + c) = make_float4(r[4 * i], r[4 * i + 1], r[4 * i + 2], r[4 * i + 3]);
    } else {
        const int c = t & 127, rr = t >> 7;
#pragma unroll
        for (int i = 0; i < 16; ++i) s[(rr + 2 * i) * DW_LD + c] = r[i];
    }
}

// Eight waves per workgroup, two per SIMD, specialised: waves 0-3 only multiply (64 MFMAs per stage from LDS, operands
// ping-ponged so the LDS latency hides behind the previous four MFMAs), waves 4-7 only move data (global -> registers one
// stage ahead, registers -> the other LDS buffer).  One barrier per stage; the movers' ~1000 cycles of address
// arithmetic, predicated loads and LDS writes run on the issue slots the 4096-cycle MFMA stream leaves free, instead of
// in front of it (a single-role kernel measured 58 % of the f32 MFMA peak in its main loop for exactly that reason).
template <bool VEC>
__device__ __forceinline__ void dw_store_global(const float (&r)[16], float* __restrict__ dst, int64_t ld, int64_t row0, int64_t row_end,
                                                int col0, int ncols, int t) {
    if (VEC) {
        const int c = col0 + 4 * (t & 31);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t row = row0 + (t >> 5) + 8 * i;
            if (row < row_end && c < ncols)
                *reinterpret_cast<float4*>(dst + row * ld + c) = make_float4(r[4 * i], r[4 * i + 1], r[4 * i + 2], r[4 * i + 3]);
        }
    } else {
        const int c = col0 + (t & 127);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int64_t row = row0 + (t >> 7) + 2 * i;
            if (row < row_end && c < ncols) dst[row * ld + c] = r[i];
        }
    }
}

template <bool YVEC, bool XVEC, bool ACT>
__global__ void __launch_bounds__(512, 1) k_mlp_dw(DwArgs a) {
    __shared__ float sY[2][DW_BM * DW_LD];
    __shared__ float sX[2][DW_BM * DW_LD];
    // XCD-aware order: consecutive workgroup ids go round-robin over the 8 XCDs, so give every XCD whole splits -- the
    // tn*tk tiles of one split read the same [rps x (N + K)] slab of dY and X and share it through that XCD's L2
    const int T = a.tn * a.tk;
    int tile, s;
    if ((a.S & 7) == 0) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        tile = j % T;
        s = (j / T) * 8 + xcd;
    } else {
        tile = blockIdx.x % T;
        s = blockIdx.x / T;
    }
    const int n0 = (tile / a.tk) * DW_T, k0 = (tile % a.tk) * DW_T;
    const int64_t m_begin = (int64_t)s * a.rps, m_end = std::min<int64_t>(m_begin + a.rps, a.M);
    const int nst = (int)((m_end - m_begin + DW_BM - 1) / DW_BM);
    const bool want_db = a.part_db != nullptr && (tile % a.tk) == 0;

    if (threadIdx.x >= 256) {
        // ------------------------------------------------------------------------------------------------ movers
        const int t = threadIdx.x - 256;
        float ry[16], rx[16], rh[ACT ? 16 : 1];
        float dbv[4] = {0.f, 0.f, 0.f, 0.f};  // column sums of the dY slab (YVEC: 4 columns x rows t>>5 + 8i; else 1 column)
        int64_t loaded_row0 = 0;
        const bool write_d = ACT && a.Dout != nullptr && (tile % a.tk) == 0;
        auto load = [&](int st) {
            const int64_t row0 = m_begin + (int64_t)st * DW_BM;  // rows >= m_end load as zeros
            loaded_row0 = row0;
            dw_load<YVEC>(ry, a.dY, a.ldy, row0, m_end, n0, a.N, t);
            if constexpr (ACT) dw_load<YVEC>(rh, a.H, a.ldh, row0, m_end, n0, a.N, t);
            dw_load<XVEC>(rx, a.X, a.ldx, row0, m_end, k0, a.K, t);
        };
        auto store = [&](int buf) {
            if constexpr (ACT) {  // gradient through ELU from its saved output: y > 0 ? 1 : y + alpha
#pragma unroll
                for (int q = 0; q < 16; ++q) ry[q] *= rh[q] > 0.0f ? 1.0f : rh[q] + a.alpha;
                if (write_d) dw_store_global<YVEC>(ry, a.Dout, a.ldd, loaded_row0, m_end, n0, a.N, t);
            }
            dw_store<YVEC>(ry, sY[buf], t);
            dw_store<XVEC>(rx, sX[buf], t);
            if (want_db) {
                if (YVEC) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) { dbv[0] += ry[4 * i]; dbv[1] += ry[4 * i + 1]; dbv[2] += ry[4 * i + 2]; dbv[3] += ry[4 * i + 3]; }
                } else {
#pragma unroll
                    for (int i = 0; i < 16; ++i) dbv[0] += ry[i];
                }
            }
        };
        if (nst > 0) {
            load(0);
            store(0);
            if (nst > 1) load(1);
        }
        __syncthreads();  // stage 0 is in LDS
        for (int st = 0; st < nst; ++st) {
            if (st + 1 < nst) {
                store((st + 1) & 1);  // that buffer was last read in stage st-1: every wave is past that barrier
                if (st + 2 < nst) load(st + 2);
            }
            __syncthreads();
        }
        if (want_db) {  // fixed-order sum of the row groups through LDS (the tile buffers are free now)
            float* red = sY[0];
            if (YVEC) {
                const int c = 4 * (t & 31), rr = t >> 5;
                *reinterpret_cast<float4*>(red + rr * DW_T + c) = make_float4(dbv[0], dbv[1], dbv[2], dbv[3]);
            } else {
                red[(t >> 7) * DW_T + (t & 127)] = dbv[0];
            }
        }
        __syncthreads();
        if (want_db && t < DW_T && n0 + t < a.N) {
            const float* red = sY[0];
            float sum = 0.0f;
            const int groups = YVEC ? 8 : 2;
            for (int g = 0; g < groups; ++g) sum += red[g * DW_T + t];
            a.part_db[(size_t)s * a.N + n0 + t] = sum;
        }
        return;
    }
    // ------------------------------------------------------------------------------------------------ multipliers
    const int t = threadIdx.x;
    const int lane = t & 63, w = t >> 6, r = lane & 31, half = lane >> 5;
    const int wn = w & 1, wk = w >> 1;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.0f;
    __syncthreads();
    for (int st = 0; st < nst; ++st) {
        const float* py = sY[st & 1] + half * DW_LD + wn * 64 + r;
        const float* px = sX[st & 1] + half * DW_LD + wk * 64 + r;
        // Operand ping-pong (P, Q): the LDS reads of step kk+1 are issued BEFORE the four MFMAs of step kk and waited
        // for after them; the scheduling barriers keep the compiler from sinking the reads behind the MFMAs again.
        float p0 = py[0], p1 = py[32], p2 = px[0], p3 = px[32], q0, q1, q2, q3;
#pragma unroll
        for (int kk = 0; kk < DW_BM / 2; kk += 2) {
            q0 = py[(kk + 1) * 2 * DW_LD]; q1 = py[(kk + 1) * 2 * DW_LD + 32];
            q2 = px[(kk + 1) * 2 * DW_LD]; q3 = px[(kk + 1) * 2 * DW_LD + 32];
            __builtin_amdgcn_sched_barrier(0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(p0, p2, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(p0, p3, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(p1, p2, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(p1, p3, acc[1][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (kk + 2 < DW_BM / 2) {
                p0 = py[(kk + 2) * 2 * DW_LD]; p1 = py[(kk + 2) * 2 * DW_LD + 32];
                p2 = px[(kk + 2) * 2 * DW_LD]; p3 = px[(kk + 2) * 2 * DW_LD + 32];
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(q0, q2, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(q0, q3, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(q1, q2, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(q1, q3, acc[1][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }
    // partial tile of this split (rows = out-features on the accumulator registers, columns = in-features on the lanes)
    float* part = a.part + (size_t)s * a.N * a.K;
    const bool interior = n0 + DW_T <= a.N && k0 + DW_T <= a.K;  // unconditional stores: no wait between them
#pragma unroll
    for (int bi = 0; bi < 2; ++bi)
#pragma unroll
        for (int bj = 0; bj < 2; ++bj) {
            const int k = k0 + wk * 64 + bj * 32 + r;
            if (interior) {
#pragma unroll
                for (int q = 0; q < 16; ++q) part[(size_t)(n0 + wn * 64 + bi * 32 + acc_row(q, half)) * a.K + k] = acc[bi][bj][q];
            } else {
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int n = n0 + wn * 64 + bi * 32 + acc_row(q, half);
                    if (n < a.N && k < a.K) part[(size_t)n * a.K + k] = acc[bi][bj][q];
                }
            }
        }
    __syncthreads();  // pairs with the movers' barrier before their bias-gradient sum
}

// out[i] = sum_s part[s][i]: 64 consecutive outputs x 16 interleaved slices of the S partials per workgroup, slices
// combined in a fixed order (deterministic); the bias gradients ride behind the weights in the same index space
constexpr int RED_Y = 16;
__global__ void __launch_bounds__(64 * RED_Y) k_mlp_reduce(int64_t total, int S, const float* __restrict__ part, float* __restrict__ out,
                                                          int N, const float* __restrict__ part_db, float* __restrict__ db) {
    __shared__ float red[RED_Y][64];
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int64_t i = (int64_t)blockIdx.x * 64 + tx;
    const float* src = nullptr;
    int64_t stride = 0;
    float* dst = nullptr;
    if (i < total) {
        src = part + i; stride = total; dst = out + i;
    } else if (db && i - total < N) {
        src = part_db + (i - total); stride = N; dst = db + (i - total);
    }
    float acc = 0.0f;
    if (src)
        for (int s = ty; s < S; s += RED_Y) acc += src[(size_t)s * stride];
    red[ty][tx] = acc;
    __syncthreads();
    if (ty == 0 && dst) {
        float sum = 0.0f;
#pragma unroll
        for (int y = 0; y < RED_Y; ++y) sum += red[y][tx];
        *dst = sum;
    }
}

// Several partial reductions in ONE launch (deferred mode, see imx_reduce_batch_*): the weight / bias gradients are only
// needed by the optimiser step, so the per-layer k_mlp_reduce launches -- 6-7 us each, sitting between the big kernels of
// the backward chain -- are collected on the host and flushed together.
constexpr int RED_MAX_SEG = 8;
struct ReduceSeg {
    int64_t total;
    const float* part;
    float* out;
    const float* part_db;
    float* db;
    int S, N;
    unsigned first_block;
    int vec;  // 1: a thread owns FOUR consecutive outputs (16-byte loads; total, N multiples of 4, 16-byte aligned partials and outputs)
};
struct ReduceBatchArgs {
    ReduceSeg seg[RED_MAX_SEG];
    int n;
};
__global__ void __launch_bounds__(64 * RED_Y) k_mlp_reduce_batch(ReduceBatchArgs a) {
    __shared__ float red[RED_Y][64];
    int k = 0;
#pragma unroll
    for (int q = 1; q < RED_MAX_SEG; ++q) k += (q < a.n && blockIdx.x >= a.seg[q].first_block) ? 1 : 0;
    const ReduceSeg& g = a.seg[k];
    const int tx = threadIdx.x, ty = threadIdx.y;
    if (g.vec) {  // four outputs per thread: the same per-output summation order (slices ty, ty + 16, ... then y = 0..15), a quarter of the loads
        __shared__ float4 red4[RED_Y][64];
        const int64_t i = ((int64_t)(blockIdx.x - g.first_block) * 64 + tx) * 4;
        const float* src = nullptr;
        int64_t stride = 0;
        float* dst = nullptr;
        if (i < g.total) {
            src = g.part + i; stride = g.total; dst = g.out + i;
        } else if (g.db && i - g.total < g.N) {
            src = g.part_db + (i - g.total); stride = g.N; dst = g.db + (i - g.total);
        }
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (src)
            for (int s = ty; s < g.S; s += RED_Y) {
                const float4 v = *reinterpret_cast<const float4*>(src + (size_t)s * stride);
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        red4[ty][tx] = acc;
        __syncthreads();
        if (ty == 0 && dst) {
            float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int y = 0; y < RED_Y; ++y) { const float4 v = red4[y][tx]; sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w; }
            *reinterpret_cast<float4*>(dst) = sum;
        }
        return;
    }
    const int64_t i = (int64_t)(blockIdx.x - g.first_block) * 64 + tx;
    const float* src = nullptr;
    int64_t stride = 0;
    float* dst = nullptr;
    if (i < g.total) {
        src = g.part + i; stride = g.total; dst = g.out + i;
    } else if (g.db && i - g.total < g.N) {
        src = g.part_db + (i - g.total); stride = g.N; dst = g.db + (i - g.total);
    }
    float acc = 0.0f;
    if (src)
        for (int s = ty; s < g.S; s += RED_Y) acc += src[(size_t)s * stride];
    red[ty][tx] = acc;
    __syncthreads();
    if (ty == 0 && dst) {
        float sum = 0.0f;
#pragma unroll
        for (int y = 0; y < RED_Y; ++y) sum += red[y][tx];
        *dst = sum;
    }
}

struct DwPlan {
    int tn, tk, S;
    int64_t rps;
};

int g_num_cu = 0;
int g_dw_cu_budget = 0;  // imx_mlp_set_dw_cu_budget: workgroups (= CUs) one imx_mlp_dw launch may spread its sample splits over; 0 = all

DwPlan dw_plan(int64_t M, int N, int K) {
    if (g_num_cu == 0) {
        int dev = 0, cu = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cu <= 0)
            cu = 256;
        g_num_cu = cu;
    }
    DwPlan p;
    p.tn = (N + DW_T - 1) / DW_T;
    p.tk = (K + DW_T - 1) / DW_T;
    const int T = p.tn * p.tk;
    const int cus = g_dw_cu_budget > 0 ? std::min(g_dw_cu_budget, g_num_cu) : g_num_cu;
    int64_t S = std::max<int64_t>(1, (int64_t)cus / T);  // one 8-wave workgroup (80 KB of LDS) per CU
    S = std::min<int64_t>(S, std::max<int64_t>(1, M / (2 * DW_BM)));  // at least two stages per split
    if (S >= 8) S &= ~(int64_t)7;
    p.rps = ((M + S - 1) / S + DW_BM - 1) / DW_BM * DW_BM;
    p.S = (int)((M + p.rps - 1) / p.rps);
    return p;
}

// ------------------------------------------------------------------------------------------------------------- head
constexpr int HEAD_A = 64;  // widest output layer handled (actions / value): 37 action means of a humanoid fit

// ELU_IN: h holds the PRE-activation output of the layer below; ELU is applied on the way in and written back in place
// (every element is read by exactly one lane), which saves that layer's separate activation pass.
#define IMX_HALF_LOG_2PI_F 0.91893853320467274178f
template <bool ELU_IN, int AMAX, int LOSS>
__global__ void __launch_bounds__(256) k_head_fwd(int64_t M, int K, int A, float* __restrict__ h, int64_t ldh,
                                                  const float* __restrict__ W, const float* __restrict__ b, float* __restrict__ y,
                                                  float alpha, imx_head_loss_t L) {
    extern __shared__ float sW[];  // [A][K]
    for (int i = threadIdx.x; i < A * K; i += blockDim.x) sW[i] = W[i];
    __syncthreads();
    const int q = threadIdx.x & 3;
    const int64_t row = (int64_t)blockIdx.x * 64 + (threadIdx.x >> 2);
    const bool live = row < M;
    float acc[AMAX];
#pragma unroll
    for (int o = 0; o < AMAX; ++o) acc[o] = 0.0f;
    float* hr = h + (live ? row : 0) * ldh;
    for (int k = 4 * q; k < K; k += 16) {  // the 4 lanes of a sample read one 64-byte run per trip
        float4 hv = *reinterpret_cast<const float4*>(hr + k);
        if (ELU_IN) {
            // branch-free: exp evaluated unconditionally (on min(x, 0)), result chosen by a select
            const float ex = (expf(fminf(hv.x, 0.0f)) - 1.0f) * alpha, ey = (expf(fminf(hv.y, 0.0f)) - 1.0f) * alpha;
            const float ez = (expf(fminf(hv.z, 0.0f)) - 1.0f) * alpha, ew = (expf(fminf(hv.w, 0.0f)) - 1.0f) * alpha;
            hv.x = hv.x > 0.0f ? hv.x : ex; hv.y = hv.y > 0.0f ? hv.y : ey; hv.z = hv.z > 0.0f ? hv.z : ez; hv.w = hv.w > 0.0f ? hv.w : ew;
            if (live) *reinterpret_cast<float4*>(hr + k) = hv;
        }
#pragma unroll
        for (int o = 0; o < AMAX; ++o)
            if (o < A) {
                const float4 wv = *reinterpret_cast<const float4*>(sW + o * K + k);
                acc[o] = fmaf(hv.w, wv.w, fmaf(hv.z, wv.z, fmaf(hv.y, wv.y, fmaf(hv.x, wv.x, acc[o]))));
            }
    }
    // reduce over the 4 lanes of a sample first, then ONE guarded region with all the stores back to back (a store inside its
    // own divergent branch would wait for the previous one: 9 us instead of 4 for 12 outputs)
#pragma unroll
    for (int o = 0; o < AMAX; ++o) {
        float v = acc[o];
        v += __shfl_xor(v, 1);
        v += __shfl_xor(v, 2);
        acc[o] = v + b[o < A ? o : 0];
    }
    if (q == 0 && live) {
#pragma unroll
        for (int o = 0; o < AMAX; ++o)
            if (o < A) y[row * A + o] = acc[o];  // uniform condition
        // the loss gradient straight from the outputs in registers: line by line k_ppo_bwd (rollout.hip), which stays the reference
        if (LOSS == 1) {
            const float inv_m = L.grad_scale / (float)M;
            float logp = 0.0f;
#pragma unroll
            for (int o = 0; o < AMAX; ++o)
                if (o < A) {
                    const float m = acc[o], s = L.sigma_d[row * L.sigma_stride + o];
                    const float d = L.actions_d[row * A + o] - m;
                    logp += -(d * d) / (2.0f * s * s) - logf(s) - IMX_HALF_LOG_2PI_F;
                }
            const float ratio = expf(logp - L.old_logp_d[row]);
            const float ad = L.advantages_d[row], clip = L.clip_param;
            const float s1 = -ad * ratio, s2 = -ad * fminf(fmaxf(ratio, 1.0f - clip), 1.0f + clip);
            float dsur_dlogp;
            if (s1 >= s2) dsur_dlogp = -ad * ratio;
            else dsur_dlogp = (ratio > 1.0f - clip && ratio < 1.0f + clip) ? -ad * ratio : 0.0f;
            const float g = dsur_dlogp * inv_m;
#pragma unroll
            for (int o = 0; o < AMAX; ++o)
                if (o < A) {
                    const float m = acc[o], s = L.sigma_d[row * L.sigma_stride + o];
                    const float d = L.actions_d[row * A + o] - m;
                    L.dmu_d[row * A + o] = g * (d / (s * s));
                    L.dsigma_d[row * A + o] = g * ((d * d) / (s * s * s) - 1.0f / s) - L.entropy_coef * inv_m / s;
                }
        } else if (LOSS == 2) {
            const float inv_m = L.grad_scale / (float)M;
            const float v = acc[0], R = L.returns_d[row], clip = L.clip_param;
            float dv;
            if (L.use_clipped_value_loss) {
                const float vo = L.old_values_d[row];
                const float dlt = v - vo;
                const float vc = vo + fminf(fmaxf(dlt, -clip), clip);
                const float l1 = (v - R) * (v - R), l2 = (vc - R) * (vc - R);
                if (l1 >= l2) dv = 2.0f * (v - R);
                else dv = (dlt > -clip && dlt < clip) ? 2.0f * (vc - R) : 0.0f;
            } else {
                dv = -2.0f * (R - v);
            }
            L.dvalue_d[row] = L.value_loss_coef * dv * inv_m;
        }
    }
}

// One wave per 32 in-features.  Per block of 32 samples: dX = dY W on the matrix core (A <= 32 reduction steps of 2), the
// accumulator tile (sample rows on the registers, in-feature on the lane) times ELU'(h) -> d of the layer below, and the
// same h registers as the B operand of dW += dY^T h.
// AB = 32-output blocks of the dW accumulator (1: A <= 32, 2: A <= 64).
template <int AB>
__global__ void __launch_bounds__(512) k_head_bwd(int64_t M, int K, int A, const float* __restrict__ dY, const float* __restrict__ h,
                                                    int64_t ldh, const float* __restrict__ W, float alpha, int has_act,
                                                    float* __restrict__ dprev, float* __restrict__ part, float* __restrict__ part_db) {
    const int lane = threadIdx.x & 63, kb = threadIdx.x >> 6, r = lane & 31, half = lane >> 5;
    const int kc = kb * 32 + r;  // this lane's in-feature
    const int steps = (A + 1) >> 1;
    float wreg[AB * 16];
#pragma unroll
    for (int s = 0; s < AB * 16; ++s) {
        const int o = 2 * s + half;
        wreg[s] = (s < steps && o < A) ? W[o * K + kc] : 0.0f;
    }
    f32x16 wacc[AB];
    float dbacc[AB];
#pragma unroll
    for (int b = 0; b < AB; ++b) {
        dbacc[b] = 0.0f;
#pragma unroll
        for (int q = 0; q < 16; ++q) wacc[b][q] = 0.0f;
    }
    // the 32 x A slab of dY of a row block is contiguous in memory: the workgroup copies it to LDS once (coalesced, double-buffered,
    // one barrier per row block) and the 16*AB + ceil(A/2) operand reads below come from there instead of 51 strided global loads
    // per wave at A = 37.  Row pitch odd: the column reads of the dX phase (lane = row) spread over the banks.
    __shared__ float sdy[2][32 * (HEAD_A + 1)];
    const int P = A | 1;
    const int64_t nblk = (M + 31) / 32;
    int buf = 0;
    for (int64_t rb = blockIdx.x; rb < nblk; rb += gridDim.x, buf ^= 1) {
        const int64_t m0 = rb * 32;
        {
            const int64_t base = m0 * A, lim = M * (int64_t)A;
            for (int e = threadIdx.x; e < 32 * A; e += blockDim.x) {
                const int row = e / A, col = e - row * A;
                sdy[buf][row * P + col] = base + e < lim ? dY[base + e] : 0.0f;
            }
        }
        __syncthreads();
        const float* __restrict__ ty = sdy[buf];
        float hv[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int64_t row = m0 + acc_row(q, half);
            hv[q] = row < M ? h[row * ldh + kc] : 0.0f;
        }
        f32x16 dx;
#pragma unroll
        for (int q = 0; q < 16; ++q) dx[q] = 0.0f;
#pragma unroll
        for (int s = 0; s < AB * 16; ++s)
            if (s < steps) {
                const int o = 2 * s + half;
                const float av = o < A ? ty[r * P + o] : 0.0f;  // rows past M are zero in the slab
                dx = __builtin_amdgcn_mfma_f32_32x32x2f32(av, wreg[s], dx, 0, 0, 0);
            }
        if (m0 + 32 <= M) {  // interior block: unconditional stores
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const float g = (has_act && hv[q] <= 0.0f) ? hv[q] + alpha : 1.0f;  // ELU' from the saved output
                dprev[(m0 + acc_row(q, half)) * (int64_t)K + kc] = dx[q] * g;
            }
        } else {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int64_t row = m0 + acc_row(q, half);
                const float g = (has_act && hv[q] <= 0.0f) ? hv[q] + alpha : 1.0f;
                if (row < M) dprev[row * (int64_t)K + kc] = dx[q] * g;
            }
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
#pragma unroll
            for (int b = 0; b < AB; ++b) {
                const int o = 32 * b + r;
                const float av = o < A ? ty[acc_row(q, half) * P + o] : 0.0f;
                wacc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, hv[q], wacc[b], 0, 0, 0);
                dbacc[b] += av;
            }
        }
    }
    float* p = part + (size_t)blockIdx.x * A * K;
#pragma unroll
    for (int b = 0; b < AB; ++b) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int o = 32 * b + acc_row(q, half);
            if (o < A) p[o * K + kc] = wacc[b][q];
        }
        if (kb == 0) {
            const float tot = dbacc[b] + __shfl_xor(dbacc[b], 32);
            if (half == 0 && 32 * b + r < A) part_db[(size_t)blockIdx.x * A + 32 * b + r] = tot;
        }
    }
}

// Output layer forward, loss gradient and backward in ONE pass over the activations of the last hidden layer (k_head_fwd + k_head_bwd
// read and wrote them four times between them: ELU in place, then the saved output again).  Per block of 32 samples the workgroup
//   P0  stores ELU(z) of the tile (prefetched into registers one block ahead) to LDS,
//   P1  forms the outputs (K/16 lanes per sample, float4 dot products against the weights in LDS, shuffle reduction), writes them and --
//       one lane per sample -- the loss gradient dY straight from them (line by line k_head_fwd's LOSS branches), into LDS and to global,
//   P2  is k_head_bwd on that block with both operands out of LDS: dX = dY W on the matrix core, times ELU'(h), and dW += dY^T h.
// Nothing but the outputs, the loss gradient and dprev is written; two barriers per block.
template <int K, int AMAX, int LOSS, bool ELU_IN>
__global__ void __launch_bounds__(2 * K) k_head_fused(int64_t M, int A, const float* __restrict__ z, int64_t ldz, const float* __restrict__ W,
                                                      const float* __restrict__ bias, float alpha, float* __restrict__ y, imx_head_loss_t L,
                                                      float* __restrict__ dprev, float* __restrict__ part, float* __restrict__ part_db) {
    constexpr int AB = (AMAX + 31) / 32;  // 32-output blocks of the dW accumulator
    extern __shared__ float smem[];
    constexpr int P = K + 32;
    const int PA = A | 1;                        // row pitches: h tile (b128 reads of 2 rows x 8 parts cover the 64 banks), dY slab (odd)
    float* sW = smem;                            // [A][K]
    float* sh = sW + ((A * K + 3) & ~3);         // [2][32][P]
    float* sdy = sh + 2 * 32 * P;                // [32][PA]
    float* sbs = sdy + 32 * PA;                  // [2][AMAX]: bias, shared sigma
    constexpr int T = 2 * K;                     // threads = K / 32 waves
    const int lane = threadIdx.x & 63, kb = threadIdx.x >> 6, r = lane & 31, half = lane >> 5;
    const int kc = kb * 32 + r;                  // backward: this lane's in-feature
    constexpr int tpr = T >> 5;                  // forward: lanes per sample (K / 16: a power of two)
    const int frow = threadIdx.x / tpr, fpart = threadIdx.x - frow * tpr;
    for (int i = threadIdx.x; i < A * K; i += T) sW[i] = W[i];
    if (threadIdx.x < AMAX) {
        sbs[threadIdx.x] = threadIdx.x < A ? bias[threadIdx.x] : 0.0f;
        if (LOSS == 1) sbs[AMAX + threadIdx.x] = (threadIdx.x < A && L.sigma_stride == 0) ? L.sigma_d[threadIdx.x] : 1.0f;
    }
    const int steps = (A + 1) >> 1;
    float wreg[AB * 16];
#pragma unroll
    for (int s = 0; s < AB * 16; ++s) {
        const int o = 2 * s + half;
        wreg[s] = (s < steps && o < A) ? W[o * K + kc] : 0.0f;
    }
    f32x16 wacc[AB];
    float dbacc[AB];
#pragma unroll
    for (int b = 0; b < AB; ++b) {
        dbacc[b] = 0.0f;
#pragma unroll
        for (int q = 0; q < 16; ++q) wacc[b][q] = 0.0f;
    }
    constexpr int kq = K >> 2;  // float4 per row
    const int64_t nblk = (M + 31) / 32;
    float4 zt[4];
    // The loss inputs, prefetched with the tile (a load inside P1 would put a memory round trip between the two barriers of every block).
    // Policy loss: the K/16 lanes of a sample share its outputs -- lane `fpart` takes outputs fpart, fpart + tpr, ... -- so every lane
    // holds the actions (and per-sample sigmas) of ITS outputs only; value loss: one lane per sample.
    constexpr int JCAP = LOSS == 1 ? (AMAX + tpr - 1) / tpr : 1;  // outputs per lane
    float la[JCAP], ls[JCAP], l0 = 0.0f, l1 = 0.0f;        // (as loaded for the NEXT block; P0 takes copies)
    const bool own_sigma = LOSS == 1 && L.sigma_stride != 0;
    auto load_tile = [&](int64_t rb) {
        {
            const int64_t grow = rb * 32 + frow < M ? rb * 32 + frow : M - 1;
            if (LOSS == 1) {
#pragma unroll
                for (int j = 0; j < JCAP; ++j) {
                    const int o = fpart + tpr * j;
                    la[j] = L.actions_d[grow * A + (o < A ? o : 0)];
                    ls[j] = own_sigma ? L.sigma_d[grow * A + (o < A ? o : 0)] : 0.0f;
                }
                l0 = L.old_logp_d[grow];
                l1 = L.advantages_d[grow];
            } else {
                l0 = L.returns_d[grow];
                l1 = L.use_clipped_value_loss ? L.old_values_d[grow] : 0.0f;
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = threadIdx.x + T * i;
            const int row = idx / kq, c4 = idx - row * kq;
            const int64_t grow = rb * 32 + row;
            zt[i] = *reinterpret_cast<const float4*>(z + (grow < M ? grow : M - 1) * ldz + 4 * c4);  // (rows past M: re-read, never used)
        }
    };
    if ((int64_t)blockIdx.x < nblk) load_tile(blockIdx.x);
    int buf = 0;
    for (int64_t rb = blockIdx.x; rb < nblk; rb += gridDim.x, buf ^= 1) {
        const int64_t m0 = rb * 32;
        float* th = sh + buf * 32 * P;
        // ---- P0
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = threadIdx.x + T * i;
            const int row = idx / kq, c4 = idx - row * kq;
            float4 hv = zt[i];
            if (ELU_IN) {
                const float ex = (expf(fminf(hv.x, 0.0f)) - 1.0f) * alpha, ey = (expf(fminf(hv.y, 0.0f)) - 1.0f) * alpha;
                const float ez = (expf(fminf(hv.z, 0.0f)) - 1.0f) * alpha, ew = (expf(fminf(hv.w, 0.0f)) - 1.0f) * alpha;
                hv.x = hv.x > 0.0f ? hv.x : ex; hv.y = hv.y > 0.0f ? hv.y : ey; hv.z = hv.z > 0.0f ? hv.z : ez; hv.w = hv.w > 0.0f ? hv.w : ew;
            }
            *reinterpret_cast<float4*>(th + row * P + 4 * c4) = hv;
        }
        float ca[JCAP], cs[JCAP];
#pragma unroll
        for (int j = 0; j < JCAP; ++j) { ca[j] = la[j]; cs[j] = ls[j]; }
        const float c0 = l0, c1 = l1;
        if (rb + gridDim.x < nblk) load_tile(rb + gridDim.x);  // in flight during P1 and P2
        __syncthreads();
        // ---- P1
        {
            float acc[AMAX];
#pragma unroll
            for (int o = 0; o < AMAX; ++o) acc[o] = 0.0f;
#pragma unroll
            for (int k0 = 0; k0 < K; k0 += 4 * tpr) {
                const int k = k0 + 4 * fpart;
                const float4 hv = *reinterpret_cast<const float4*>(th + frow * P + k);
#pragma unroll
                for (int o = 0; o < AMAX; ++o)
                    if (o < A) {
                        const float4 wv = *reinterpret_cast<const float4*>(sW + o * K + k);
                        acc[o] = fmaf(hv.w, wv.w, fmaf(hv.z, wv.z, fmaf(hv.y, wv.y, fmaf(hv.x, wv.x, acc[o]))));
                    }
            }
#pragma unroll
            for (int o = 0; o < AMAX; ++o)
                if (o < A) {
                    float v = acc[o];
#pragma unroll
                    for (int off = 1; off < tpr; off <<= 1) v += __shfl_xor(v, off);
                    acc[o] = v + sbs[o];
                }
            const int64_t row = m0 + frow;
            const bool live = row < M;
            if (fpart == 0 && live) {
#pragma unroll
                for (int o = 0; o < AMAX; ++o)
                    if (o < A) y[row * A + o] = acc[o];
            }
            if (LOSS == 1) {
                // every lane of the sample holds all A outputs (the xor butterfly leaves the sums everywhere) and works on its own
                const float inv_m = L.grad_scale / (float)M;
                float mj[JCAP], sj[JCAP];
                float logp = 0.0f;
#pragma unroll
                for (int j = 0; j < JCAP; ++j) {
                    mj[j] = 0.0f; sj[j] = 1.0f;
                    if (j * tpr < A) {  // (uniform)
                        const int o = fpart + tpr * j;
                        float m = 0.0f, sg = 1.0f;
#pragma unroll
                        for (int oo = 0; oo < AMAX; ++oo) {
                            m = oo == o ? acc[oo] : m;
                            sg = oo == o ? sbs[AMAX + oo] : sg;
                        }
                        if (own_sigma) sg = cs[j];
                        mj[j] = m; sj[j] = sg;
                        const float d = ca[j] - m;
                        logp += o < A ? -(d * d) / (2.0f * sg * sg) - logf(sg) - IMX_HALF_LOG_2PI_F : 0.0f;
                    }
                }
#pragma unroll
                for (int off = 1; off < tpr; off <<= 1) logp += __shfl_xor(logp, off);
                const float ratio = expf(logp - c0);
                const float ad = c1, clip = L.clip_param;
                const float s1 = -ad * ratio, s2 = -ad * fminf(fmaxf(ratio, 1.0f - clip), 1.0f + clip);
                float dsur_dlogp;
                if (s1 >= s2) dsur_dlogp = -ad * ratio;
                else dsur_dlogp = (ratio > 1.0f - clip && ratio < 1.0f + clip) ? -ad * ratio : 0.0f;
                const float g = dsur_dlogp * inv_m;
#pragma unroll
                for (int j = 0; j < JCAP; ++j)
                    if (j * tpr < A) {
                        const int o = fpart + tpr * j;
                        const float m = mj[j], sg = sj[j];
                        const float d = ca[j] - m;
                        const float dm = g * (d / (sg * sg));
                        if (o < A) {
                            if (live) {
                                L.dmu_d[row * A + o] = dm;
                                L.dsigma_d[row * A + o] = g * ((d * d) / (sg * sg * sg) - 1.0f / sg) - L.entropy_coef * inv_m / sg;
                            }
                            sdy[frow * PA + o] = live ? dm : 0.0f;  // rows past M contribute nothing
                        }
                    }
            } else if (fpart == 0) {
                float g = 0.0f;
                if (live) {
                    const float inv_m = L.grad_scale / (float)M;
                    const float v = acc[0], R = c0, clip = L.clip_param;
                    float dv;
                    if (L.use_clipped_value_loss) {
                        const float vo = c1;
                        const float dlt = v - vo;
                        const float vc = vo + fminf(fmaxf(dlt, -clip), clip);
                        const float l1v = (v - R) * (v - R), l2v = (vc - R) * (vc - R);
                        if (l1v >= l2v) dv = 2.0f * (v - R);
                        else dv = (dlt > -clip && dlt < clip) ? 2.0f * (vc - R) : 0.0f;
                    } else {
                        dv = -2.0f * (R - v);
                    }
                    g = L.value_loss_coef * dv * inv_m;
                    L.dvalue_d[row] = g;
                }
                sdy[frow * PA] = g;
            }
        }
        __syncthreads();
        // ---- P2
        float hv[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) hv[q] = m0 + acc_row(q, half) < M ? th[acc_row(q, half) * P + kc] : 0.0f;
        f32x16 dx;
#pragma unroll
        for (int q = 0; q < 16; ++q) dx[q] = 0.0f;
#pragma unroll
        for (int s = 0; s < AB * 16; ++s)
            if (s < steps) {
                const int o = 2 * s + half;
                const float av = o < A ? sdy[r * PA + o] : 0.0f;
                dx = __builtin_amdgcn_mfma_f32_32x32x2f32(av, wreg[s], dx, 0, 0, 0);
            }
        if (m0 + 32 <= M) {  // interior block: unconditional stores
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const float g = (ELU_IN && hv[q] <= 0.0f) ? hv[q] + alpha : 1.0f;  // ELU' from the activated value
                dprev[(m0 + acc_row(q, half)) * (int64_t)K + kc] = dx[q] * g;
            }
        } else {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int64_t row = m0 + acc_row(q, half);
                const float g = (ELU_IN && hv[q] <= 0.0f) ? hv[q] + alpha : 1.0f;
                if (row < M) dprev[row * (int64_t)K + kc] = dx[q] * g;
            }
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
#pragma unroll
            for (int b = 0; b < AB; ++b) {
                const int o = 32 * b + r;
                const float av = o < A ? sdy[acc_row(q, half) * PA + o] : 0.0f;
                wacc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, hv[q], wacc[b], 0, 0, 0);
                dbacc[b] += av;
            }
        }
    }
    float* p = part + (size_t)blockIdx.x * A * K;
#pragma unroll
    for (int b = 0; b < AB; ++b) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int o = 32 * b + acc_row(q, half);
            if (o < A) p[o * K + kc] = wacc[b][q];
        }
        if (kb == 0) {
            const float tot = dbacc[b] + __shfl_xor(dbacc[b], 32);
            if (half == 0 && 32 * b + r < A) part_db[(size_t)blockIdx.x * A + 32 * b + r] = tot;
        }
    }
}

int head_grid(int64_t M) { return (int)std::min<int64_t>((M + 31) / 32, 256); }

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int imx_mlp_set_dw_cu_budget(int cus) {
    const int old = g_dw_cu_budget;
    g_dw_cu_budget = cus > 0 ? cus : 0;
    return old;
}

extern "C" size_t imx_mlp_scratch_bytes(int64_t M, int out_features, int in_features) {
    if (M <= 0 || out_features <= 0 || in_features <= 0) return 0;
    size_t dw = 0;
    {
        const DwPlan p = dw_plan(M, out_features, in_features);
        dw = ((size_t)p.S * out_features * in_features + (size_t)p.S * out_features) * sizeof(float);
    }
    const size_t head = out_features <= HEAD_A
                            ? ((size_t)head_grid(M) * out_features * in_features + (size_t)head_grid(M) * out_features) * sizeof(float)
                            : 0;
    return std::max(dw, head) + 256;
}

typedef struct imx_reduce_batch imx_reduce_batch_t;
struct imx_reduce_batch {
    ReduceBatchArgs args;
    unsigned blocks;
};
static thread_local imx_reduce_batch* t_batch = nullptr;  // bound by imx_reduce_batch_begin on the issuing host thread

namespace {

// launch the reduction now, or queue it on the batch bound to this thread
int reduce_or_defer(const char* who, int64_t total, int S, const float* part, float* out, int N, const float* part_db, float* db,
                    hipStream_t st) {
    const int64_t threads = total + (db ? N : 0);
    const unsigned blocks = (unsigned)((threads + 63) / 64);
    if (t_batch) {
        IMX_REQUIRE(t_batch->args.n < RED_MAX_SEG, "%s: more than %d deferred reductions in one batch", who, RED_MAX_SEG);
        ReduceSeg& g = t_batch->args.seg[t_batch->args.n++];
        g.total = total; g.part = part; g.out = out; g.part_db = part_db; g.db = db; g.S = S; g.N = N;
        g.vec = (total % 4 == 0) && (!db || N % 4 == 0) && aligned16(part) && aligned16(out) && (!db || (aligned16(part_db) && aligned16(db)));
        g.first_block = t_batch->blocks;
        t_batch->blocks += g.vec ? (unsigned)((threads / 4 + 63) / 64) : blocks;
        return 0;
    }
    hipLaunchKernelGGL(k_mlp_reduce, dim3(blocks), dim3(64, RED_Y), 0, st, total, S, part, out, N, part_db, db);
    IMX_HIP(hipGetLastError());
    return 0;
}

}  // namespace

static int mlp_dw_launch(const char* who, int64_t M, int N, int K, const float* dY_d, int64_t ldy, const float* H_d, int64_t ldh,
                         float alpha, float* Dout_d, int64_t ldd, const float* X_d, int64_t ldx, float* dW_d, float* db_d,
                         void* scratch_d, size_t scratch_bytes, imx_stream_t stream) {
    IMX_REQUIRE(M > 0 && N > 0 && K > 0 && dY_d && X_d && dW_d && scratch_d, "%s: bad arguments", who);
    IMX_REQUIRE(ldy >= N && ldx >= K, "%s: row pitch smaller than the row (ldy=%lld N=%d, ldx=%lld K=%d)", who, (long long)ldy, N,
                (long long)ldx, K);
    const bool act = H_d != nullptr;
    IMX_REQUIRE(!act || (ldh >= N && (!Dout_d || ldd >= N)), "%s: activation / output pitch smaller than the row", who);
    IMX_REQUIRE(!act || Dout_d != dY_d, "%s: d_out must not alias the incoming gradient (other tiles still read it)", who);
    const DwPlan p = dw_plan(M, N, K);
    const size_t need = ((size_t)p.S * N * K + (size_t)p.S * N) * sizeof(float);
    IMX_REQUIRE(scratch_bytes >= need, "%s: scratch too small (%zu < %zu bytes; see imx_mlp_scratch_bytes)", who, scratch_bytes, need);
    DwArgs a;
    a.M = M; a.N = N; a.K = K;
    a.dY = dY_d; a.ldy = ldy; a.X = X_d; a.ldx = ldx;
    a.part = (float*)scratch_d;
    a.part_db = db_d ? a.part + (size_t)p.S * N * K : nullptr;
    a.tn = p.tn; a.tk = p.tk; a.S = p.S; a.rps = p.rps;
    a.H = H_d; a.ldh = ldh; a.Dout = Dout_d; a.ldd = ldd; a.alpha = alpha;
    bool yvec = (N % 4 == 0) && (ldy % 4 == 0) && aligned16(dY_d);
    if (act) yvec = yvec && (ldh % 4 == 0) && aligned16(H_d) && (!Dout_d || ((ldd % 4 == 0) && aligned16(Dout_d)));
    // 16-byte loads of X need aligned rows; K itself may be ragged when the row pitch covers the last vector (the columns past K land in
    // dW columns that are never written)
    const bool xvec = (ldx % 4 == 0) && aligned16(X_d) && (K % 4 == 0 || ldx >= ((K + 3) & ~3));
    const dim3 grid((unsigned)(p.tn * p.tk * p.S)), block(512);
    hipStream_t st = (hipStream_t)stream;
#define IMX_DW_LAUNCH(Y, X, A) hipLaunchKernelGGL((k_mlp_dw<Y, X, A>), grid, block, 0, st, a)
    if (act) {
        if (yvec && xvec) IMX_DW_LAUNCH(true, true, true);
        else if (yvec) IMX_DW_LAUNCH(true, false, true);
        else if (xvec) IMX_DW_LAUNCH(false, true, true);
        else IMX_DW_LAUNCH(false, false, true);
    } else {
        if (yvec && xvec) IMX_DW_LAUNCH(true, true, false);
        else if (yvec) IMX_DW_LAUNCH(true, false, false);
        else if (xvec) IMX_DW_LAUNCH(false, true, false);
        else IMX_DW_LAUNCH(false, false, false);
    }
#undef IMX_DW_LAUNCH
    IMX_HIP(hipGetLastError());
    return reduce_or_defer(who, (int64_t)N * K, p.S, a.part, dW_d, N, a.part_db, db_d, st);
}

extern "C" int imx_reduce_batch_create(imx_reduce_batch_t** out) {
    IMX_REQUIRE(out, "imx_reduce_batch_create: null argument");
    *out = new imx_reduce_batch();
    (*out)->args.n = 0;
    (*out)->blocks = 0;
    return 0;
}

extern "C" void imx_reduce_batch_destroy(imx_reduce_batch_t* b) {
    if (t_batch == b) t_batch = nullptr;
    delete b;
}

extern "C" int imx_reduce_batch_begin(imx_reduce_batch_t* b) {
    IMX_REQUIRE(b, "imx_reduce_batch_begin: null batch");
    IMX_REQUIRE(t_batch == nullptr, "imx_reduce_batch_begin: another batch is still open on this thread (flush it first)");
    t_batch = b;  // (reductions queued on it directly -- imx_mlp_head_fwd_bwd -- stay: the batch empties at flush)
    return 0;
}

extern "C" int imx_reduce_batch_flush(imx_reduce_batch_t* b, imx_stream_t stream) {
    IMX_REQUIRE(b && t_batch == b, "imx_reduce_batch_flush: this batch is not the one open on this thread");
    t_batch = nullptr;
    if (b->args.n == 0) return 0;
    hipLaunchKernelGGL(k_mlp_reduce_batch, dim3(b->blocks), dim3(64, RED_Y), 0, (hipStream_t)stream, b->args);
    b->args.n = 0;
    b->blocks = 0;
    IMX_HIP(hipGetLastError());
    return 0;
}

extern "C" int imx_mlp_dw(int64_t M, int N, int K, const float* dY_d, int64_t ldy, const float* X_d, int64_t ldx, float* dW_d,
                          float* db_d, void* scratch_d, size_t scratch_bytes, imx_stream_t stream) {
    return mlp_dw_launch("imx_mlp_dw", M, N, K, dY_d, ldy, nullptr, 0, 0.0f, nullptr, 0, X_d, ldx, dW_d, db_d, scratch_d, scratch_bytes,
                         stream);
}

extern "C" int imx_mlp_dw_elu(int64_t M, int N, int K, const float* dH_d, int64_t ldg, const float* H_d, int64_t ldh, float elu_alpha,
                              float* dZ_out_d, int64_t ldd, const float* X_d, int64_t ldx, float* dW_d, float* db_d, void* scratch_d,
                              size_t scratch_bytes, imx_stream_t stream) {
    IMX_REQUIRE(H_d, "imx_mlp_dw_elu: the saved activation output is required");
    return mlp_dw_launch("imx_mlp_dw_elu", M, N, K, dH_d, ldg, H_d, ldh, elu_alpha, dZ_out_d, ldd, X_d, ldx, dW_d, db_d, scratch_d,
                         scratch_bytes, stream);
}

static int head_fwd_launch(int64_t M, int K, int A, float* h_d, int64_t ldh, const float* W_d, const float* b_d, float* y_d,
                           int elu_in_place, float elu_alpha, const imx_head_loss_t* loss, imx_stream_t stream) {
    IMX_REQUIRE(M > 0 && h_d && W_d && b_d && y_d, "imx_mlp_head_fwd: bad arguments");
    IMX_REQUIRE(A >= 1 && A <= HEAD_A, "imx_mlp_head_fwd: %d outputs (1..%d supported; wider layers are library GEMMs)", A, HEAD_A);
    IMX_REQUIRE(K >= 16 && K % 16 == 0 && K <= 2048 && ldh >= K && ldh % 4 == 0 && aligned16(h_d),
                "imx_mlp_head_fwd: in-features %d (pitch %lld) must be a multiple of 16, 16-byte aligned rows", K, (long long)ldh);
    IMX_REQUIRE((size_t)A * K * sizeof(float) <= 64 * 1024, "imx_mlp_head_fwd: %d x %d weights do not fit the 64 KB of LDS used", A, K);
    imx_head_loss_t L{};
    int mode = 0;
    if (loss) {
        L = *loss;
        mode = L.mode;
        IMX_REQUIRE(mode == 1 || mode == 2, "imx_mlp_head_fwd_loss: mode %d (1 policy, 2 value)", mode);
        if (mode == 1) {
            IMX_REQUIRE(L.sigma_d && L.actions_d && L.old_logp_d && L.advantages_d && L.dmu_d && L.dsigma_d, "imx_mlp_head_fwd_loss: null policy argument");
            IMX_REQUIRE(L.sigma_stride == 0 || L.sigma_stride == A, "imx_mlp_head_fwd_loss: sigma_stride must be 0 (shared std) or A");
        } else {
            IMX_REQUIRE(A == 1 && L.returns_d && L.dvalue_d && (!L.use_clipped_value_loss || L.old_values_d), "imx_mlp_head_fwd_loss: value head needs A = 1, returns, old values");
        }
    }
    const dim3 grid((unsigned)((M + 63) / 64)), block(256);
    const size_t lds = (size_t)A * K * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
#define IMX_HEAD_FWD(E, AM, LS) hipLaunchKernelGGL((k_head_fwd<E, AM, LS>), grid, block, lds, st, M, K, A, h_d, ldh, W_d, b_d, y_d, elu_alpha, L)
#define IMX_HEAD_FWD_E(AM, LS) do { if (elu_in_place) IMX_HEAD_FWD(true, AM, LS); else IMX_HEAD_FWD(false, AM, LS); } while (0)
    if (A <= 16) {
        if (mode == 0) IMX_HEAD_FWD_E(16, 0); else if (mode == 1) IMX_HEAD_FWD_E(16, 1); else IMX_HEAD_FWD_E(16, 2);
    } else {
        if (mode == 0) IMX_HEAD_FWD_E(HEAD_A, 0); else if (mode == 1) IMX_HEAD_FWD_E(HEAD_A, 1); else IMX_HEAD_FWD_E(HEAD_A, 2);
    }
#undef IMX_HEAD_FWD_E
#undef IMX_HEAD_FWD
    IMX_HIP(hipGetLastError());
    return 0;
}

extern "C" int imx_mlp_head_fwd(int64_t M, int K, int A, float* h_d, int64_t ldh, const float* W_d, const float* b_d, float* y_d,
                                int elu_in_place, float elu_alpha, imx_stream_t stream) {
    return head_fwd_launch(M, K, A, h_d, ldh, W_d, b_d, y_d, elu_in_place, elu_alpha, nullptr, stream);
}

extern "C" int imx_mlp_head_fwd_loss(int64_t M, int K, int A, float* h_d, int64_t ldh, const float* W_d, const float* b_d, float* y_d,
                                     int elu_in_place, float elu_alpha, const imx_head_loss_t* loss, imx_stream_t stream) {
    IMX_REQUIRE(loss, "imx_mlp_head_fwd_loss: null loss description");
    return head_fwd_launch(M, K, A, h_d, ldh, W_d, b_d, y_d, elu_in_place, elu_alpha, loss, stream);
}

extern "C" int imx_mlp_head_bwd(int64_t M, int K, int A, const float* dY_d, const float* h_d, int64_t ldh, const float* W_d, float elu_alpha,
                                int has_activation, float* dprev_d, float* dW_d, float* db_d, void* scratch_d, size_t scratch_bytes,
                                imx_stream_t stream) {
    IMX_REQUIRE(M > 0 && dY_d && h_d && W_d && dprev_d && dW_d && db_d && scratch_d, "imx_mlp_head_bwd: bad arguments");
    IMX_REQUIRE(A >= 1 && A <= HEAD_A, "imx_mlp_head_bwd: %d outputs (1..%d supported)", A, HEAD_A);
    IMX_REQUIRE(K >= 32 && K % 32 == 0 && K <= 256 && ldh >= K, "imx_mlp_head_bwd: in-features %d must be a multiple of 32, at most 256", K);
    const int G = head_grid(M);
    const size_t need = ((size_t)G * A * K + (size_t)G * A) * sizeof(float);
    IMX_REQUIRE(scratch_bytes >= need, "imx_mlp_head_bwd: scratch too small (%zu < %zu bytes)", scratch_bytes, need);
    float* part = (float*)scratch_d;
    float* part_db = part + (size_t)G * A * K;
    hipStream_t st = (hipStream_t)stream;
    if (A <= 32)
        hipLaunchKernelGGL((k_head_bwd<1>), dim3((unsigned)G), dim3((unsigned)(K / 32 * 64)), 0, st, M, K, A, dY_d, h_d, ldh, W_d, elu_alpha,
                           has_activation, dprev_d, part, part_db);
    else
        hipLaunchKernelGGL((k_head_bwd<2>), dim3((unsigned)G), dim3((unsigned)(K / 32 * 64)), 0, st, M, K, A, dY_d, h_d, ldh, W_d, elu_alpha,
                           has_activation, dprev_d, part, part_db);
    IMX_HIP(hipGetLastError());
    return reduce_or_defer("imx_mlp_head_bwd", (int64_t)A * K, G, part, dW_d, A, part_db, db_d, st);
}

extern "C" int imx_mlp_head_fwd_bwd(int64_t M, int K, int A, const float* z_d, int64_t ldz, const float* W_d, const float* b_d, float* y_d,
                                    int elu_in, float elu_alpha, const imx_head_loss_t* loss, float* dprev_d, float* dW_d, float* db_d,
                                    void* scratch_d, size_t scratch_bytes, imx_reduce_batch_t* defer_to, imx_stream_t stream) {
    IMX_REQUIRE(M > 0 && z_d && W_d && b_d && y_d && loss && dprev_d && dW_d && db_d && scratch_d, "imx_mlp_head_fwd_bwd: bad arguments");
    IMX_REQUIRE(A >= 1 && A <= HEAD_A, "imx_mlp_head_fwd_bwd: %d outputs (1..%d supported)", A, HEAD_A);
    IMX_REQUIRE((K == 128 || K == 256) && ldz >= K && ldz % 4 == 0 && aligned16(z_d),
                "imx_mlp_head_fwd_bwd: in-features %d (pitch %lld) must be 128 or 256 with 16-byte aligned rows", K, (long long)ldz);
    IMX_REQUIRE(A <= 16, "imx_mlp_head_fwd_bwd: %d outputs (at most 16: wider heads run as imx_mlp_head_fwd_loss + imx_mlp_head_bwd)", A);
    const imx_head_loss_t L = *loss;
    IMX_REQUIRE(L.mode == 1 || L.mode == 2, "imx_mlp_head_fwd_bwd: loss mode %d (1 = policy, 2 = value)", L.mode);
    if (L.mode == 1) {
        IMX_REQUIRE(L.sigma_d && L.actions_d && L.old_logp_d && L.advantages_d && L.dmu_d && L.dsigma_d, "imx_mlp_head_fwd_bwd: null policy argument");
        IMX_REQUIRE(L.sigma_stride == 0 || L.sigma_stride == A, "imx_mlp_head_fwd_bwd: sigma_stride must be 0 (shared std) or A");
    } else {
        IMX_REQUIRE(A == 1 && L.returns_d && L.dvalue_d && (!L.use_clipped_value_loss || L.old_values_d),
                    "imx_mlp_head_fwd_bwd: value head needs A = 1, returns, old values");
    }
    const int G = head_grid(M);
    const size_t need = ((size_t)G * A * K + (size_t)G * A) * sizeof(float);
    IMX_REQUIRE(scratch_bytes >= need, "imx_mlp_head_fwd_bwd: scratch too small (%zu < %zu bytes)", scratch_bytes, need);
    float* part = (float*)scratch_d;
    float* part_db = part + (size_t)G * A * K;
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = ((size_t)((A * K + 3) & ~3) + 2ull * 32 * (K + 32) + 32ull * (A | 1) + 2 * HEAD_A) * sizeof(float);
    const dim3 grid((unsigned)G), block((unsigned)(2 * K));
#define IMX_HEAD_FUSED(K_, AM, LS, E)                                                                                                      \
    do {                                                                                                                                  \
        static bool attr = false;                                                                                                         \
        if (!attr) {                                                                                                                      \
            IMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_head_fused<K_, AM, LS, E>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                        160 * 1024));                                                                                    \
            attr = true;                                                                                                                  \
        }                                                                                                                                 \
        hipLaunchKernelGGL((k_head_fused<K_, AM, LS, E>), grid, block, lds, st, M, A, z_d, ldz, W_d, b_d, elu_alpha, y_d, L, dprev_d, part, \
                           part_db);                                                                                                      \
    } while (0)
#define IMX_HEAD_FUSED_K(AM, LS, E) do { if (K == 128) IMX_HEAD_FUSED(128, AM, LS, E); else IMX_HEAD_FUSED(256, AM, LS, E); } while (0)
    if (L.mode == 2) {
        if (elu_in) IMX_HEAD_FUSED_K(1, 2, true); else IMX_HEAD_FUSED_K(1, 2, false);
    } else {
        if (elu_in) IMX_HEAD_FUSED_K(16, 1, true); else IMX_HEAD_FUSED_K(16, 1, false);
    }
#undef IMX_HEAD_FUSED_K
#undef IMX_HEAD_FUSED
    IMX_HIP(hipGetLastError());
    if (defer_to) {  // queue the reduction of the split partials on that batch (flushed by the backward pass that follows)
        imx_reduce_batch* saved = t_batch;
        t_batch = defer_to;
        const int rc = reduce_or_defer("imx_mlp_head_fwd_bwd", (int64_t)A * K, G, part, dW_d, A, part_db, db_d, st);
        t_batch = saved;
        return rc;
    }
    return reduce_or_defer("imx_mlp_head_fwd_bwd", (int64_t)A * K, G, part, dW_d, A, part_db, db_d, st);
}

// ------------------------------------------------------------------------------------------- fused MLP inference (rollout)
// PPO.act / evaluate during the rollout (upstream actor_critic.py::act_inference / evaluate): the whole
// Linear+ELU stack of BOTH networks in one launch.  A workgroup owns 32 samples and carries them through every layer with
// the activations in LDS (two ping-pong buffers); weights go global -> registers (every weight is used once per
// workgroup, so LDS staging would buy nothing): per 32 reduction indices a lane loads one full 64-byte half of a weight
// row (the two half-waves cover one 128-byte line per row) and the matching 4 x 16 bytes of activations from LDS, then
// issues 16 MFMAs per 32-column block.  blockIdx selects the network: actor and critic fill all CUs together at
// N = 4096 (2 x 128 workgroups), where per-layer library GEMMs on M = 4096 rows leave most of the chip idle.
constexpr int INF_ROWS = 32, INF_MAXD = 512, INF_PITCH = INF_MAXD + 4, INF_MAXL = 4;

struct InferNet {
    const float* W[INF_MAXL];
    const float* Wp[INF_MAXL];  // the same weights in the lane order of the 32-sample kernel (imx_mlp_pack_weights), or null
    const float* b[INF_MAXL];
    int ldw[INF_MAXL];   // row pitch of W (floats): multiple of 32, zero padded
    int dim[INF_MAXL + 1];
    int nlayers;
    float alpha;     // ELU alpha of the hidden layers
    float* out;      // (M, dim[nlayers])
};
struct InferArgs {
    int64_t M;
    const float* X;
    int64_t ldx;
    int tiles;       // workgroups per network
    InferNet net[2];
    int nnets;
};

// imx_mlp_infer_act: PPO.act's sampling / log-prob / storage writes (imx_policy_act) and ActionManager.process_action
// (imx_action_process) in the epilogue of the actor head.  The actor workgroup keeps its 32 (16) action means in LDS instead of
// storing them, then lane = (sample, action column): a = mu + std * z with the SAME counter-based draws as k_policy_act (keyed by
// seed, step counter, env * A + column), transition -> slot t of the storage, the action through the env's action terms; one lane
// per sample adds the log-prob terms in column order (bit-identical to k_policy_act's sequential sum).
struct ActArgs {
    int enabled, A, has_plan;
    float pre_clip;
    const float* std_a;
    uint64_t seed;
    const int32_t* step_d;
    float *act_out, *logp_out, *mu_out, *sigma_out, *obs_out;
    PlanView P;
    imx_state_t S;
    imx_buffers_t Bf;
};

template <int ROWS>
__device__ __forceinline__ void act_epilogue(const ActArgs& c, const float* __restrict__ mu_s, float* __restrict__ term_s, int64_t m0, int64_t M) {
    const int A = c.A;
    const uint32_t step = c.step_d ? (uint32_t)c.step_d[0] : 0u;
    for (int i = threadIdx.x; i < ROWS * A; i += blockDim.x) {
        const int row = i / A, a = i - row * A;
        const int64_t e = m0 + row;
        float term = 0.0f;
        if (e < M) {
            const float m = mu_s[row * INF_PITCH + a], s = c.std_a[a];
            const float u1 = 1.0f - uniform01(c.seed, step, (uint64_t)(e * A + a) * 2);        // (0,1]
            const float u2 = uniform01(c.seed ^ 0x5851F42D4C957F2Dull, step, (uint64_t)(e * A + a) * 2 + 1);
            const float z = sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
            const float x = m + s * z;
            const float d = x - m;
            term = -(d * d) / (2.0f * s * s) - logf(s) - IMX_HALF_LOG_2PI;
            c.act_out[e * A + a] = x;
            c.mu_out[e * A + a] = m;
            c.sigma_out[e * A + a] = s;
            if (c.has_plan) action_process_element(c.P, c.S, c.Bf, e, a, x, c.pre_clip);
        }
        term_s[row * INF_PITCH + a] = term;
    }
    __syncthreads();
    for (int row = threadIdx.x; row < ROWS; row += blockDim.x) {
        const int64_t e = m0 + row;
        if (e >= M) continue;
        float logp = 0.0f;
        for (int a = 0; a < A; ++a) logp += term_s[row * INF_PITCH + a];
        c.logp_out[e] = logp;
    }
}

// One layer for the 32 samples of the workgroup.  NBW = 32-column output blocks per wave (4, 2 or 1); a reduction GROUP is
// GS = 4 / NBW sub-groups of 32 indices, so that every group is 64 MFMAs per wave (4096 cycles, more than an L2 round trip)
// whatever the layer width.  Two operand sets (P, Q) ping-pong: the loads of group g+1 are issued before the MFMAs of group
// g.  Every load in the loop is UNCONDITIONAL (indices clamped; the host guarantees zero-padded weight rows with a pitch
// that is a multiple of 32, activation columns beyond K are zero in LDS, column blocks beyond N re-read row N-1 and are
// dropped in the epilogue): a load inside a divergent branch makes the compiler wait for all outstanding loads at the
// join, which measured 2x slower here.
// PACKED: W points at the packed copy (imx_mlp_pack_weights): chunk ((cb * nsub + sc) * 4 + i) holds, lane by lane, exactly the float4
// the row layout's load (column block cb, sub-group sc, piece i) gives each lane -- one contiguous KiB per wave instruction instead of
// 32 rows x 2 x 16 bytes.  Lanes reading 32 different weight rows is what held this kernel at ~4 TB/s of L2 traffic (256 workgroups x
// 1.1 MB of weights per launch); contiguous pieces stream from L2 at several times that.
template <int NBW, bool PACKED>
__device__ __forceinline__ void infer_layer(const float* __restrict__ sIn, int K, const float* __restrict__ W, int ldw,
                                            const float* __restrict__ bias, int N, bool elu, float alpha, float* __restrict__ sOut,
                                            float* __restrict__ gOut, int64_t m0, int64_t M) {
    constexpr int GS = 4 / NBW;
    if ((int)(threadIdx.x >> 6) * 32 >= ((N + 31) & ~31)) return;  // this wave owns no column block of a narrow layer (the action head): wave-uniform
    // (Eight waves per workgroup -- two per SIMD splitting the column blocks, so that one multiplies while the other waits for weight rows --
    //  were measured: 106.8 us against 68.2 us for both networks at 4096 samples, the same 49 us for one network on half the chip.  What
    //  slows the launch down when all 256 CUs run it is the shared weight stream out of L2, not exposed latency inside a SIMD.)
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r = lane & 31, half = lane >> 5;
    f32x16 acc[NBW];
#pragma unroll
    for (int j = 0; j < NBW; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[j][q] = 0.0f;
    const int nsub = (K + 31) >> 5;            // 32-index sub-groups
    const int ng = (nsub + GS - 1) / GS;       // groups
    const float* wrow[NBW];
#pragma unroll
    for (int j = 0; j < NBW; ++j) {
        if (PACKED) {
            const int ncb = (N + 31) >> 5, cb = w + 4 * j;
            wrow[j] = W + ((size_t)(cb < ncb ? cb : ncb - 1) * nsub * 4) * 256 + 4 * lane;  // (blocks past N re-read the last one; dropped below)
        } else {
            const int n = (w + 4 * j) * 32 + r;
            wrow[j] = W + (size_t)(n < N ? n : N - 1) * ldw + 16 * half;
        }
    }
    constexpr int WSTEP_SC = PACKED ? 4 * 256 : 32, WSTEP_I = PACKED ? 256 : 4;  // floats between sub-groups / between the four pieces
    const float* arow = sIn + r * INF_PITCH + 16 * half;
    float4 Pa[GS][4], Qa[GS][4], Pb[NBW][GS][4], Qb[NBW][GS][4];
    // sub-groups past the end (a group may be partial) are clamped to the last one for the loads and multiplied by zero
    auto load = [&](float4 (&a)[GS][4], float4 (&b)[NBW][GS][4], int g) {
#pragma unroll
        for (int u = 0; u < GS; ++u) {
            const int sg = g * GS + u;
            const int sc = sg < nsub ? sg : nsub - 1;
#pragma unroll
            for (int j = 0; j < NBW; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) b[j][u][i] = *reinterpret_cast<const float4*>(wrow[j] + WSTEP_SC * sc + WSTEP_I * i);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float4 v = *reinterpret_cast<const float4*>(arow + 32 * sc + 4 * i);
                if (sg >= nsub) v = make_float4(0.f, 0.f, 0.f, 0.f);
                a[u][i] = v;
            }
        }
    };
    auto mult = [&](const float4 (&a)[GS][4], const float4 (&b)[NBW][GS][4]) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < GS; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int j = 0; j < NBW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][i].x, b[j][u][i].x, acc[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < NBW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][i].y, b[j][u][i].y, acc[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < NBW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][i].z, b[j][u][i].z, acc[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < NBW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][i].w, b[j][u][i].w, acc[j], 0, 0, 0);
            }
        __builtin_amdgcn_sched_barrier(0);
    };
    load(Pa, Pb, 0);
    int g = 0;
    for (; g + 1 < ng; g += 2) {
        load(Qa, Qb, g + 1);
        mult(Pa, Pb);
        load(Pa, Pb, g + 2 < ng ? g + 2 : ng - 1);  // clamped: at most one redundant reload at the end
        mult(Qa, Qb);
    }
    if (g < ng) mult(Pa, Pb);  // odd number of groups: P holds the last one
    // epilogue: accumulator rows = samples (registers), columns = out-features (lanes).  Branch-free arithmetic (ELU as a
    // select over an unconditionally evaluated exp) and unconditional LDS stores: a store inside a divergent branch makes
    // the compiler wait for the previous one.  Columns beyond N of a hidden layer land in the zero-padding region, which
    // is re-zeroed by the caller before the next layer reads it -- only the last layer (global stores) is guarded.
#pragma unroll
    for (int j = 0; j < NBW; ++j) {
        const int n = (w + 4 * j) * 32 + r;
        const float bv = bias[n < N ? n : N - 1];
        float v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            float x = acc[j][q] + bv;
            const float e = (expf(fminf(x, 0.0f)) - 1.0f) * alpha;
            v[q] = (elu && x <= 0.0f) ? e : x;
        }
        if (sOut) {
            if (n < INF_MAXD) {
#pragma unroll
                for (int q = 0; q < 16; ++q) sOut[acc_row(q, half) * INF_PITCH + n] = n < N ? v[q] : 0.0f;
            }
        } else {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int row = acc_row(q, half);
                if (n < N && m0 + row < M) gOut[(m0 + row) * (int64_t)N + n] = v[q];
            }
        }
    }
}

template <bool PACKED>
__global__ void __launch_bounds__(256, 1) k_mlp_infer(InferArgs a, ActArgs act) {
    extern __shared__ float smem[];  // two activation buffers of INF_ROWS x INF_PITCH floats
    float* buf0 = smem;
    float* buf1 = smem + INF_ROWS * INF_PITCH;
    const int which = blockIdx.x / a.tiles;
    const InferNet& net = a.net[which];
    const int64_t m0 = (int64_t)(blockIdx.x - which * a.tiles) * INF_ROWS;
    const bool act_here = act.enabled && which == 0;  // the actor's workgroups finish PPO.act + ActionManager.process_action themselves
    // input rows -> LDS (the 32 rows are one contiguous run when ldx == dim[0]); columns up to the next multiple of 32 are zeroed
    const int K0 = net.dim[0], K0p = (K0 + 31) & ~31;
    {
        // wave w takes rows w, w+4, ...; lanes run along the row.  ALL loads of the tile are issued before the first LDS store -- one HBM
        // round trip for the whole input (a load -> store loop pays one per iteration: 16 us for 235 columns; two half tiles paid two) --
        // and only the 64-column pieces the input has (a clamped load of a piece past K0 is still a memory request).
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        const int ncc = (K0 + 63) >> 6;  // (uniform)
        float v[8][INF_MAXD / 64];
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
            const int row = w + 4 * rr;
            const float* src = a.X + (m0 + row < a.M ? m0 + row : 0) * a.ldx;
#pragma unroll
            for (int cc = 0; cc < INF_MAXD / 64; ++cc) {
                if (cc < ncc) {
                    const int c = lane + 64 * cc;
                    const float x = src[c < K0 ? c : 0];  // clamped, unconditional within the piece
                    v[rr][cc] = (c < K0 && m0 + row < a.M) ? x : 0.0f;
                } else {
                    v[rr][cc] = 0.0f;
                }
            }
        }
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
            const int row = w + 4 * rr;
#pragma unroll
            for (int cc = 0; cc < INF_MAXD / 64; ++cc) {
                const int c = lane + 64 * cc;
                if (c < K0p) buf0[row * INF_PITCH + c] = v[rr][cc];
                if (act_here && c < K0 && m0 + row < a.M) act.obs_out[(m0 + row) * (int64_t)K0 + c] = v[rr][cc];  // storage.observations[t]
            }
        }
    }
    __syncthreads();
    float* in = buf0;
    float* out = buf1;
    for (int l = 0; l < net.nlayers; ++l) {
        const int K = net.dim[l], N = net.dim[l + 1];
        const bool last = l == net.nlayers - 1;
        const int nbw = ((N + 31) / 32 + 3) / 4;  // 32-column blocks per wave
        float* so = (last && !act_here) ? nullptr : out;  // (the actor head of imx_mlp_infer_act keeps its means in LDS)
        if (!last) {  // zero the padding columns the next layer's 32-wide reduction groups will read
            const int Np = (N + 31) & ~31;
            for (int i = threadIdx.x; i < INF_ROWS * (Np - N); i += blockDim.x) {
                const int row = i / (Np - N), col = N + i - row * (Np - N);
                out[row * INF_PITCH + col] = 0.0f;
            }
        }
        const float* Wl = PACKED ? net.Wp[l] : net.W[l];
        if (nbw <= 1) infer_layer<1, PACKED>(in, K, Wl, net.ldw[l], net.b[l], N, !last, net.alpha, so, net.out, m0, a.M);
        else if (nbw == 2) infer_layer<2, PACKED>(in, K, Wl, net.ldw[l], net.b[l], N, !last, net.alpha, so, net.out, m0, a.M);
        else infer_layer<4, PACKED>(in, K, Wl, net.ldw[l], net.b[l], N, !last, net.alpha, so, net.out, m0, a.M);
        __syncthreads();
        float* t = in; in = out; out = t;
    }
    if (act_here) act_epilogue<INF_ROWS>(act, in, out, m0, a.M);
}

// ---- 16-sample variant (v_mfma_f32_16x16x4_f32) for small batches: twice the workgroups (and half the LDS each) when
// 32-sample tiles would leave half of the CUs idle (<= 2048 envs for two networks on 256 CUs).  Same structure as above:
// NBW = 16-column blocks per wave (8, 4, 2, 1), a sub-group is 32 reduction indices = 8 MFMA steps per block (lane quarter
// kq supplies k = 32s + 4kq + {0..3} and 32s + 16 + 4kq + {0..3}: the four quarters cover one 128-byte weight line).
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int INF16_ROWS = 16;

template <int NBW>
__device__ __forceinline__ void infer_layer16(const float* __restrict__ sIn, int K, const float* __restrict__ W, int ldw,
                                              const float* __restrict__ bias, int N, bool elu, float alpha, float* __restrict__ sOut,
                                              float* __restrict__ gOut, int64_t m0, int64_t M) {
    constexpr int GS = NBW >= 8 ? 1 : (NBW == 4 ? 2 : 4);  // sub-groups per group: >= 64 MFMAs (2048 cycles) per group
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, j = lane & 15, kq = lane >> 4;
    f32x4 acc[NBW];
#pragma unroll
    for (int b = 0; b < NBW; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[b][q] = 0.0f;
    const int nsub = (K + 31) >> 5;
    const int ng = (nsub + GS - 1) / GS;
    const float* wrow[NBW];
#pragma unroll
    for (int b = 0; b < NBW; ++b) {
        const int n = (w + 4 * b) * 16 + j;
        wrow[b] = W + (size_t)(n < N ? n : N - 1) * ldw + 4 * kq;
    }
    const float* arow = sIn + j * INF_PITCH + 4 * kq;
    float4 Pa[GS][2], Qa[GS][2], Pb[NBW][GS][2], Qb[NBW][GS][2];
    auto load = [&](float4 (&a)[GS][2], float4 (&bq)[NBW][GS][2], int g) {
#pragma unroll
        for (int u = 0; u < GS; ++u) {
            const int sg = g * GS + u;
            const int sc = sg < nsub ? sg : nsub - 1;
#pragma unroll
            for (int b = 0; b < NBW; ++b) {
                bq[b][u][0] = *reinterpret_cast<const float4*>(wrow[b] + 32 * sc);
                bq[b][u][1] = *reinterpret_cast<const float4*>(wrow[b] + 32 * sc + 16);
            }
            float4 v0 = *reinterpret_cast<const float4*>(arow + 32 * sc);
            float4 v1 = *reinterpret_cast<const float4*>(arow + 32 * sc + 16);
            if (sg >= nsub) { v0 = make_float4(0.f, 0.f, 0.f, 0.f); v1 = v0; }
            a[u][0] = v0;
            a[u][1] = v1;
        }
    };
    auto mult = [&](const float4 (&a)[GS][2], const float4 (&bq)[NBW][GS][2]) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < GS; ++u)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int b = 0; b < NBW; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][i].x, bq[b][u][i].x, acc[b], 0, 0, 0);
#pragma unroll
                for (int b = 0; b < NBW; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][i].y, bq[b][u][i].y, acc[b], 0, 0, 0);
#pragma unroll
                for (int b = 0; b < NBW; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][i].z, bq[b][u][i].z, acc[b], 0, 0, 0);
#pragma unroll
                for (int b = 0; b < NBW; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][i].w, bq[b][u][i].w, acc[b], 0, 0, 0);
            }
        __builtin_amdgcn_sched_barrier(0);
    };
    load(Pa, Pb, 0);
    int g = 0;
    for (; g + 1 < ng; g += 2) {
        load(Qa, Qb, g + 1);
        mult(Pa, Pb);
        load(Pa, Pb, g + 2 < ng ? g + 2 : ng - 1);
        mult(Qa, Qb);
    }
    if (g < ng) mult(Pa, Pb);
    // epilogue: 16x16 accumulator: column = lane & 15, row = (lane >> 4) * 4 + register (branch-free, as in infer_layer)
#pragma unroll
    for (int b = 0; b < NBW; ++b) {
        const int n = (w + 4 * b) * 16 + j;
        const float bv = bias[n < N ? n : N - 1];
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float x = acc[b][q] + bv;
            const float e = (expf(fminf(x, 0.0f)) - 1.0f) * alpha;
            v[q] = (elu && x <= 0.0f) ? e : x;
        }
        if (sOut) {
            if (n < INF_MAXD) {
#pragma unroll
                for (int q = 0; q < 4; ++q) sOut[(kq * 4 + q) * INF_PITCH + n] = n < N ? v[q] : 0.0f;
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = kq * 4 + q;
                if (n < N && m0 + row < M) gOut[(m0 + row) * (int64_t)N + n] = v[q];
            }
        }
    }
}

__global__ void __launch_bounds__(256, 2) k_mlp_infer16(InferArgs a, ActArgs act) {
    extern __shared__ float smem[];  // two activation buffers of INF16_ROWS x INF_PITCH floats
    float* buf0 = smem;
    float* buf1 = smem + INF16_ROWS * INF_PITCH;
    const int which = blockIdx.x / a.tiles;
    const InferNet& net = a.net[which];
    const int64_t m0 = (int64_t)(blockIdx.x - which * a.tiles) * INF16_ROWS;
    const bool act_here = act.enabled && which == 0;
    const int K0 = net.dim[0], K0p = (K0 + 31) & ~31;
    {
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        float v[4][INF_MAXD / 64];  // wave w: rows w, w+4, w+8, w+12; all loads before the first LDS store
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int row = w + 4 * rr;
            const float* src = a.X + (m0 + row < a.M ? m0 + row : 0) * a.ldx;
#pragma unroll
            for (int cc = 0; cc < INF_MAXD / 64; ++cc) {
                const int c = lane + 64 * cc;
                const float x = src[c < K0 ? c : 0];
                v[rr][cc] = (c < K0 && m0 + row < a.M) ? x : 0.0f;
            }
        }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int row = w + 4 * rr;
#pragma unroll
            for (int cc = 0; cc < INF_MAXD / 64; ++cc) {
                const int c = lane + 64 * cc;
                if (c < K0p) buf0[row * INF_PITCH + c] = v[rr][cc];
                if (act_here && c < K0 && m0 + row < a.M) act.obs_out[(m0 + row) * (int64_t)K0 + c] = v[rr][cc];
            }
        }
    }
    __syncthreads();
    float* in = buf0;
    float* out = buf1;
    for (int l = 0; l < net.nlayers; ++l) {
        const int K = net.dim[l], N = net.dim[l + 1];
        const bool last = l == net.nlayers - 1;
        const int nbw = ((N + 15) / 16 + 3) / 4;  // 16-column blocks per wave
        float* so = (last && !act_here) ? nullptr : out;
        if (!last) {
            const int Np = (N + 31) & ~31;
            for (int i = threadIdx.x; i < INF16_ROWS * (Np - N); i += blockDim.x) {
                const int row = i / (Np - N), col = N + i - row * (Np - N);
                out[row * INF_PITCH + col] = 0.0f;
            }
        }
        if (nbw <= 1) infer_layer16<1>(in, K, net.W[l], net.ldw[l], net.b[l], N, !last, net.alpha, so, net.out, m0, a.M);
        else if (nbw == 2) infer_layer16<2>(in, K, net.W[l], net.ldw[l], net.b[l], N, !last, net.alpha, so, net.out, m0, a.M);
        else if (nbw <= 4) infer_layer16<4>(in, K, net.W[l], net.ldw[l], net.b[l], N, !last, net.alpha, so, net.out, m0, a.M);
        else infer_layer16<8>(in, K, net.W[l], net.ldw[l], net.b[l], N, !last, net.alpha, so, net.out, m0, a.M);
        __syncthreads();
        float* t = in; in = out; out = t;
    }
    if (act_here) act_epilogue<INF16_ROWS>(act, in, out, m0, a.M);
}

extern "C" int imx_mlp_infer(int64_t M, const float* X_d, int64_t ldx, int nnets, const int* nlayers, const int* dims,
                             const float* const* weights_d, const int* weight_pitch, const float* const* biases_d, const float* elu_alpha,
                             float* const* out_d, imx_stream_t stream) {
    return imx_mlp_infer_act(M, X_d, ldx, nnets, nlayers, dims, weights_d, weight_pitch, nullptr, biases_d, elu_alpha, out_d, nullptr, stream);
}

// one thread per float4 of the packed images of up to 8 layers (one launch per refresh)
constexpr int PACK_MAX = 8;
struct PackArgs {
    const float* W[PACK_MAX];
    float* out[PACK_MAX];
    int64_t ldw[PACK_MAX];
    int64_t first4[PACK_MAX + 1];  // float4 offset of layer k in the launch's index space
    int N[PACK_MAX], K[PACK_MAX], nsub[PACK_MAX];
    int n;
};
__global__ void k_pack_weights(PackArgs a) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= a.first4[a.n]) return;
    int k = 0;
#pragma unroll
    for (int q = 1; q < PACK_MAX; ++q) k += (q < a.n && g >= a.first4[q]) ? 1 : 0;
    const int64_t t = g - a.first4[k];
    const int N = a.N[k], K = a.K[k], nsub = a.nsub[k];
    const int lane = (int)(t & 63);
    const int64_t chunk = t >> 6;
    const int i = (int)(chunk & 3), sc = (int)((chunk >> 2) % nsub), cb = (int)((chunk >> 2) / nsub);
    const int r = lane & 31, half = lane >> 5;
    const int n = cb * 32 + r, k0 = 32 * sc + 16 * half + 4 * i;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n < N) {
        const float* src = a.W[k] + (size_t)n * a.ldw[k] + k0;
        if (k0 + 0 < K) v.x = src[0];
        if (k0 + 1 < K) v.y = src[1];
        if (k0 + 2 < K) v.z = src[2];
        if (k0 + 3 < K) v.w = src[3];
    }
    reinterpret_cast<float4*>(a.out[k])[t] = v;
}

extern "C" size_t imx_mlp_packed_floats(int out_features, int in_features) {
    return (size_t)((out_features + 31) / 32 * 32) * (size_t)((in_features + 31) / 32 * 32);
}

extern "C" int imx_mlp_pack_weights_batch(int nlayers, const int* out_features, const int* in_features, const float* const* W_d,
                                          const int64_t* ldw, float* const* packed_d, imx_stream_t stream) {
    IMX_REQUIRE(nlayers >= 1 && nlayers <= PACK_MAX && out_features && in_features && W_d && ldw && packed_d,
                "imx_mlp_pack_weights_batch: bad arguments (1..%d layers per call)", PACK_MAX);
    PackArgs a{};
    a.n = nlayers;
    a.first4[0] = 0;
    for (int k = 0; k < nlayers; ++k) {
        IMX_REQUIRE(out_features[k] > 0 && in_features[k] > 0 && W_d[k] && packed_d[k] && ldw[k] >= in_features[k] && aligned16(packed_d[k]),
                    "imx_mlp_pack_weights_batch: layer %d: bad shape, pitch or alignment", k);
        a.W[k] = W_d[k]; a.out[k] = packed_d[k]; a.ldw[k] = ldw[k];
        a.N[k] = out_features[k]; a.K[k] = in_features[k]; a.nsub[k] = (in_features[k] + 31) / 32;
        a.first4[k + 1] = a.first4[k] + (int64_t)imx_mlp_packed_floats(out_features[k], in_features[k]) / 4;
    }
    hipLaunchKernelGGL(k_pack_weights, dim3((unsigned)((a.first4[nlayers] + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    IMX_HIP(hipGetLastError());
    return 0;
}

extern "C" int imx_mlp_pack_weights(int out_features, int in_features, const float* W_d, int64_t ldw, float* packed_d, imx_stream_t stream) {
    return imx_mlp_pack_weights_batch(1, &out_features, &in_features, &W_d, &ldw, &packed_d, stream);
}

extern "C" int imx_mlp_infer_act(int64_t M, const float* X_d, int64_t ldx, int nnets, const int* nlayers, const int* dims,
                                 const float* const* weights_d, const int* weight_pitch, const float* const* packed_weights_d,
                                 const float* const* biases_d, const float* elu_alpha, float* const* out_d, const imx_policy_act_t* pa,
                                 imx_stream_t stream) {
    IMX_REQUIRE(M > 0 && X_d && nnets >= 1 && nnets <= 2 && nlayers && dims && weights_d && biases_d && elu_alpha && out_d,
                "imx_mlp_infer: bad arguments");
    ActArgs act{};
    if (pa) {
        IMX_REQUIRE(pa->std_d && pa->actions_out_d && pa->logp_out_d && pa->mu_out_d && pa->sigma_out_d && pa->obs_out_d,
                    "imx_mlp_infer_act: null argument in imx_policy_act_t");
        act.enabled = 1;
        act.std_a = pa->std_d; act.seed = pa->seed; act.step_d = pa->step_counter_d;
        act.act_out = pa->actions_out_d; act.logp_out = pa->logp_out_d; act.mu_out = pa->mu_out_d; act.sigma_out = pa->sigma_out_d;
        act.obs_out = pa->obs_out_d;
        act.pre_clip = pa->pre_clip;
        act.A = dims[nlayers[0]];  // width of network 0's last layer = number of actions
        if (pa->plan) {
            IMX_REQUIRE(pa->state && pa->buf && pa->plan->dev, "imx_mlp_infer_act: plan without state / buffers / device tables");
            IMX_REQUIRE(pa->plan->A == act.A, "imx_mlp_infer_act: the actor has %d outputs, the plan %d action columns", act.A, pa->plan->A);
            if (imx_check_action_inputs(pa->plan, pa->state, pa->buf, "imx_mlp_infer_act")) return 1;
            act.has_plan = 1;
            act.P = imx_plan_view(pa->plan);
            act.S = *pa->state;
            act.Bf = *pa->buf;
        }
    }
    InferArgs a;
    a.M = M; a.X = X_d; a.ldx = ldx; a.nnets = nnets;
    a.tiles = (int)((M + INF_ROWS - 1) / INF_ROWS);
    int wi = 0, di = 0;
    for (int k = 0; k < nnets; ++k) {
        InferNet& n = a.net[k];
        n.nlayers = nlayers[k];
        IMX_REQUIRE(n.nlayers >= 1 && n.nlayers <= INF_MAXL, "imx_mlp_infer: network %d has %d layers (1..%d supported)", k, n.nlayers, INF_MAXL);
        for (int l = 0; l <= n.nlayers; ++l) {
            n.dim[l] = dims[di++];
            IMX_REQUIRE(n.dim[l] >= 1 && n.dim[l] <= INF_MAXD, "imx_mlp_infer: layer width %d outside 1..%d", n.dim[l], INF_MAXD);
        }
        IMX_REQUIRE(n.dim[0] == a.net[0].dim[0] && ldx >= n.dim[0], "imx_mlp_infer: the networks must share the input (width %d, pitch %lld)",
                    a.net[0].dim[0], (long long)ldx);
        for (int l = 0; l < n.nlayers; ++l) {
            n.W[l] = weights_d[wi];
            n.Wp[l] = packed_weights_d ? packed_weights_d[wi] : nullptr;
            IMX_REQUIRE(!packed_weights_d || (n.Wp[l] && aligned16(n.Wp[l])), "imx_mlp_infer: null / unaligned packed weights (network %d, layer %d)", k, l);
            n.b[l] = biases_d[wi];
            n.ldw[l] = weight_pitch ? weight_pitch[wi] : n.dim[l];
            ++wi;
            IMX_REQUIRE(n.W[l] && n.b[l], "imx_mlp_infer: null weight / bias (network %d, layer %d)", k, l);
            IMX_REQUIRE(n.ldw[l] >= n.dim[l] && n.ldw[l] % 32 == 0 && aligned16(n.W[l]),
                        "imx_mlp_infer: weights of network %d layer %d need a 16-byte aligned, zero-padded row pitch that is a multiple "
                        "of 32 floats (pitch %d, in-features %d)", k, l, n.ldw[l], n.dim[l]);
        }
        n.alpha = elu_alpha[k];
        n.out = out_d[k];
        IMX_REQUIRE(n.out || (pa && k == 0), "imx_mlp_infer: null output (network %d)", k);
    }
    if (g_num_cu == 0) (void)dw_plan(64, 32, 32);  // fills g_num_cu
    // 16-sample tiles double the weight traffic out of L2 (every workgroup reads every weight; measured ceiling ~6 TB/s for
    // these 16-byte-per-lane row gathers: at 4096 envs both variants sit on it, and a deeper, three-set prefetch ring changed
    // nothing), so they only pay when 32-sample tiles would leave at least half of the CUs without a workgroup
    const bool small = (int64_t)a.tiles * nnets * 2 <= g_num_cu;
    static bool attr_set = false;
    if (!attr_set) {
        IMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_mlp_infer<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)(2ull * INF_ROWS * INF_PITCH * sizeof(float))));
        IMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_mlp_infer<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)(2ull * INF_ROWS * INF_PITCH * sizeof(float))));
        IMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_mlp_infer16), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)(2ull * INF16_ROWS * INF_PITCH * sizeof(float))));
        attr_set = true;
    }
    if (small) {
        a.tiles = (int)((M + INF16_ROWS - 1) / INF16_ROWS);
        hipLaunchKernelGGL(k_mlp_infer16, dim3((unsigned)(a.tiles * nnets)), dim3(256), 2ull * INF16_ROWS * INF_PITCH * sizeof(float),
                           (hipStream_t)stream, a, act);
    } else if (packed_weights_d) {
        hipLaunchKernelGGL(k_mlp_infer<true>, dim3((unsigned)(a.tiles * nnets)), dim3(256), 2ull * INF_ROWS * INF_PITCH * sizeof(float),
                           (hipStream_t)stream, a, act);
    } else {
        hipLaunchKernelGGL(k_mlp_infer<false>, dim3((unsigned)(a.tiles * nnets)), dim3(256), 2ull * INF_ROWS * INF_PITCH * sizeof(float),
                           (hipStream_t)stream, a, act);
    }
    IMX_HIP(hipGetLastError());
    return 0;
}
