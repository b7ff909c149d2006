// First-layer forward of the PPO update with the activation fused: Y = ELU(X W^T + b), in-features K <= 256 (235 observations).
//
// In the update (rsl_rl PPO.update -> ActorCritic.actor / .critic forward, upstream rsl_rl/modules/actor_critic.py) this layer was a
// library GEMM (24576 x 235) . (235 x 1024) -- actor and critic first layers stacked -- followed by an ELU pass that reads and writes the
// 100 MB result once more: 111 us + 37 us per minibatch.  One launch here, 141 us (tools/fwd_bench.py; 22 vs 30 us at K = 48):
//   * a workgroup owns 128 output columns; each of its four waves keeps the W rows of ITS 32 columns in REGISTERS for the whole launch
//     (B operand of v_mfma_f32_32x32x2_f32: lane = column, NS k-steps = NS VGPRs), fetched once through LDS;
//   * it walks its share of the M samples in tiles of 32 rows, double-buffered in LDS and filled by LDS DMA (K > 128) or through
//     registers (K <= 128, where the registers are there);
//   * bias + ELU are applied to the accumulators and the activated tile is stored once -- for K > 128 in between the MFMAs of the next tile.
// What bounds it (NOTES.md, round 3): the MFMA chain itself is 86 us (157 TFLOP/s fp32 peak -> 78 us); each vector-memory instruction a
// wave issues between MFMAs (8 DMAs + 16 row stores per 120 MFMAs) costs it ~100 issue cycles that a second wave per SIMD does not hide.
// Exact fp32 products accumulated in fp32; the summation order differs from the library's (k and NS + k paired per step), tested against
// torch within the tolerance of the whole-update test.  PARITY UNPINNED like the rest of the rsl_rl restatement.
#include "imx_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int FW_KMAX = 256;          // in-features supported (k held in registers: FW_KMAX / 2 VGPRs per lane)
constexpr int FW_ROWS = 32;           // samples per tile
constexpr int FW_KP = FW_KMAX + 4;    // LDS row pitch in floats: rows stay 16-byte aligned and 8 consecutive rows cover the 32 banks with b128 reads
constexpr int FW_COLS = 128;          // output columns per workgroup (4 waves x 32)

__device__ __forceinline__ int acc_row32(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

// Epilogue of one 32 x 32 accumulator tile: rows = samples on the accumulator registers, columns = out-features on the lanes.
// `y` points at this lane's column in the tile's first row of its half (row0 + 4 half); `rows` = rows of the tile that exist.
// Kept lean on purpose: the two workgroups of a CU fall into step (both in the MFMA phase, then both here), so every VALU
// instruction of the epilogue is time the MFMA pipe idles -- one 64-bit pointer walked by ldy, the ELU branch and the row guard hoisted.
template <bool ELU, bool FULL>
__device__ __forceinline__ void store_rows(const f32x16& acc, float bias, float alpha, float* y, int64_t ldy, int rows, int half) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        float* yg = y + (int64_t)(8 * g) * ldy;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float x = acc[4 * g + j] + bias;
            if (ELU) {
                const float e = (__expf(fminf(x, 0.0f)) - 1.0f) * alpha;
                x = x <= 0.0f ? e : x;
            }
            if (FULL || 8 * g + j + 4 * half < rows) yg[0] = x;
            yg += ldy;
        }
    }
}

__device__ __forceinline__ void store_tile(const f32x16& acc, float bias, float alpha, int elu, float* y, int64_t ldy, int rows, int half) {
    if (rows >= FW_ROWS) {
        if (elu) store_rows<true, true>(acc, bias, alpha, y, ldy, rows, half);
        else store_rows<false, true>(acc, bias, alpha, y, ldy, rows, half);
    } else {
        if (elu) store_rows<true, false>(acc, bias, alpha, y, ldy, rows, half);
        else store_rows<false, false>(acc, bias, alpha, y, ldy, rows, half);
    }
}

struct FwArgs {
    int64_t M;
    int N, K;
    const float* X;
    int64_t ldx;
    const float* W;   // (N, K) row-major, dense
    const float* b;   // (N)
    float* Y;
    int64_t ldy;
    float alpha;
    int ncb;          // column blocks = ceil(N / 128)
    int rs;           // row splits
    int64_t rps;      // rows per split (multiple of FW_ROWS)
    int elu;
};

// NS = MFMA steps per tile (each consumes two k): the k range [0, 2 NS) covers K; k >= K meets zero weights AND zero tile columns.
// MFMA step s pairs k = s (lanes 0..31) with k = NS + s (lanes 32..63): both operands are then CONTIGUOUS in s per lane, so the A operand
// comes out of LDS sixteen bytes (four steps) at a time and nothing in the step loop depends on K.
template <int NS, bool XVEC>
__global__ void __launch_bounds__(256, 2) k_mlp_fwd_elu(FwArgs a) {
    static_assert(NS % 4 == 0 && 2 * NS <= FW_KMAX, "step count");
    __shared__ __attribute__((aligned(16))) float sA[2][FW_ROWS * FW_KP];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, r = lane & 31, half = lane >> 5;
    const int cb = blockIdx.x % a.ncb, split = blockIdx.x / a.ncb;  // the column blocks of one row split run together: X tiles shared through L2
    const int col = cb * FW_COLS + w * 32 + r;
    const bool col_ok = col < a.N;
    const int K = a.K;
    // ---- this wave's weights: lane (column r, k-half `half`) holds W[col][half * NS + s], s = 0 .. NS
    float breg[NS];
    {
        const float* wr = a.W + (size_t)(col_ok ? col : a.N - 1) * K;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int k = half * NS + s;
            const float v = wr[k < K ? k : K - 1];  // (unconditional load, clamped: no branch per register)
            breg[s] = (k < K && col_ok) ? v : 0.0f;
        }
    }
    const float bias = col_ok ? a.b[col] : 0.0f;
    const int64_t m_begin = (int64_t)split * a.rps, m_end = min(m_begin + a.rps, a.M);
    const int ntile = (int)((m_end - m_begin + FW_ROWS - 1) / FW_ROWS);
    // ---- staging: 256 threads move one 32 x 2NS tile as float4; XVEC: 16-byte global loads (rows start on 16-byte boundaries), else 4-byte
    constexpr int C4 = 2 * NS / 4;                      // float4 per tile row
    constexpr int NV = (FW_ROWS * C4 + 255) / 256;      // float4 per thread per tile
    float4 pre[NV];
    auto g_load = [&](int tile) {
        const int64_t row0 = m_begin + (int64_t)tile * FW_ROWS;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = t + 256 * i;
            const int rr = idx / C4, c4 = idx - rr * C4;
            const int64_t row = min(row0 + rr, m_end - 1);      // (rows past the end are loaded again and never stored)
            const int k0 = 4 * c4;
            const float* p = a.X + row * a.ldx + (k0 < K ? k0 : 0);
            float4 v;
            if (XVEC) {
                v = *reinterpret_cast<const float4*>(p);
            } else {
                v.x = p[0];
                v.y = p[k0 + 1 < K ? 1 : 0];
                v.z = p[k0 + 2 < K ? 2 : 0];
                v.w = p[k0 + 3 < K ? 3 : 0];
            }
            // columns >= K (the pad of the row pitch, and the tail of the 2 NS range) must be ZERO in the tile: a NaN there would survive the zero weight
            v.x = k0 + 0 < K ? v.x : 0.f;
            v.y = k0 + 1 < K ? v.y : 0.f;
            v.z = k0 + 2 < K ? v.z : 0.f;
            v.w = k0 + 3 < K ? v.w : 0.f;
            pre[i] = v;
        }
    };
    auto s_store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = t + 256 * i;
            const int rr = idx / C4, c4 = idx - rr * C4;
            if (FW_ROWS * C4 % 256 == 0 || rr < FW_ROWS) *reinterpret_cast<float4*>(sA[buf] + rr * FW_KP + 4 * c4) = pre[i];
        }
    };
    if (ntile > 0) {
        g_load(0);
        s_store(0);
    }
    __syncthreads();
    for (int tile = 0; tile < ntile; ++tile) {
        const int buf = tile & 1;
        if (tile + 1 < ntile) g_load(tile + 1);  // in flight during the MFMAs below
        f32x16 acc;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = 0.0f;
        const float4* pa = reinterpret_cast<const float4*>(sA[buf] + r * FW_KP + half * NS);  // A operand: lane (row r, k-half) reads A[r][half NS + s]
#pragma unroll
        for (int s4 = 0; s4 < NS / 4; ++s4) {
            const float4 av = pa[s4];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, breg[4 * s4 + 0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, breg[4 * s4 + 1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, breg[4 * s4 + 2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, breg[4 * s4 + 3], acc, 0, 0, 0);
        }
        const int64_t row0 = m_begin + (int64_t)tile * FW_ROWS;
        if (col_ok) store_tile(acc, bias, a.alpha, a.elu, a.Y + (row0 + 4 * half) * a.ldy + col, a.ldy, (int)min<int64_t>(m_end - row0, FW_ROWS), half);
        if (tile + 1 < ntile) s_store(buf ^ 1);  // that buffer was last read in tile - 1: every wave is past the barrier below
        __syncthreads();
    }
}

// The same tile walk with the tile written into LDS by the load itself (global_load_lds_dwordx4: no staging registers, no LDS write
// instructions) -- what lets NS = 120 weights + the accumulator fit under 256 VGPRs at two workgroups per CU.  One DMA per tile row
// (lane = float4 of the row: the destination of a wave instruction is contiguous by lane, so a row pitch of >= 1 KiB keeps rows apart);
// wave w moves rows 8 w .. 8 w + 7 and, once they have landed, zeroes their columns >= K (row-pitch pad / tail of the 2 NS range: a NaN
// there would survive the zero weight).  Needs 16-byte aligned rows (the XVEC condition).
template <int NS>
__global__ void __launch_bounds__(256, 2) k_mlp_fwd_elu_dma(FwArgs a) {
    static_assert(NS % 4 == 0 && 2 * NS <= FW_KMAX, "step count");
    __shared__ __attribute__((aligned(16))) float sA[2][FW_ROWS * FW_KP];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, r = lane & 31, half = lane >> 5;
    // consecutive workgroup ids go to different XCDs (8, each with its own L2): renumber so that the column blocks of one row split --
    // which read the same X rows -- are neighbours on ONE XCD
    const int nb = gridDim.x, per = nb >> 3;
    const int vb = (nb & 7) == 0 ? (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int cb = vb % a.ncb, split = vb / a.ncb;
    const int col = cb * FW_COLS + w * 32 + r;
    const bool col_ok = col < a.N;
    const int K = a.K;
    float breg[NS];
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)&sA[0][0];
    {
        // The 32 rows of W this wave owns are ONE contiguous block of 32 K floats.  Lane = column wants a row each, i.e. a 4 K-byte stride
        // between lanes: read that way every load instruction touches 64 cache lines (measured: ~50 us of a 160 us launch).  Instead the
        // block comes through LDS (the tile buffers are still free): read flat and coalesced, then every lane picks its row.  The two
        // tile buffers hold two such blocks, so the waves take turns in pairs.
        float* wl = &sA[0][0] + (w & 1) * (FW_ROWS * FW_KP);                         // one tile buffer: 8320 floats >= 32 * 256
        const int64_t blk0 = (int64_t)(cb * FW_COLS + w * 32) * K;                   // flat offset of the block in W
        const int64_t avail = min<int64_t>((int64_t)32 * K, (int64_t)a.N * K - blk0); // floats of it that exist (last column block; may be <= 0)
        const bool wvec = ((reinterpret_cast<uintptr_t>(a.W) & 15) == 0);            // (blk0 is a multiple of 32 floats)
        const float* wrow = wl + r * K + half * NS;
        for (int turn = 0; turn < 2; ++turn) {
            if ((w >> 1) == turn) {
                const int64_t avail4 = avail & ~(int64_t)3;  // floats of the block covered by whole, aligned float4
                if (wvec && avail4 >= 4) {
                    // flat and contiguous on both sides: exactly the shape of the LDS DMA (1 KiB per instruction, nothing staged in registers,
                    // all of them in flight together -- a load/store loop pays one memory latency per iteration: ~20 us of the launch)
                    const unsigned wl_lds = lds_base + (unsigned)((w & 1) * FW_ROWS * FW_KP * 4);
                    for (int i = 0; i < 32 * K; i += 256) {
                        const float* src = a.W + blk0 + min<int64_t>(i + 4 * lane, avail4 - 4);  // (past the block: re-read its last whole vector; those columns are masked)
                        unsigned keep;
                        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                                     : "=&s"(keep) : "v"(src), "s"(__builtin_amdgcn_readfirstlane(wl_lds + (unsigned)i * 4u)) : "memory");
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    // a ragged last column block whose float count is not a multiple of four: its last 1-3 floats belong to a VALID column
                    // and sit in a vector the clamp above replaced -- fetch them one by one (LDS operations of a wave execute in order)
                    if (lane < avail - avail4) wl[avail4 + lane] = a.W[blk0 + avail4 + lane];
                } else {
                    for (int i = 4 * lane; i < 32 * K; i += 256) {
                        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                        const float* src = a.W + blk0 + i;
                        if (i + 4 <= avail) {
                            if (wvec) v = *reinterpret_cast<const float4*>(src);
                            else { v.x = src[0]; v.y = src[1]; v.z = src[2]; v.w = src[3]; }
                        } else {
                            if (i + 0 < avail) v.x = src[0];
                            if (i + 1 < avail) v.y = src[1];
                            if (i + 2 < avail) v.z = src[2];
                        }
                        *reinterpret_cast<float4*>(wl + i) = v;
                    }
                }
                // (LDS operations of one wave execute in order: the reads below see the writes above without a barrier)
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const int k = half * NS + s;
                    const float v = wrow[k < K ? s : 0];  // (unconditional, clamped: the reads pipeline)
                    breg[s] = (k < K && col_ok) ? v : 0.0f;
                }
            }
            __syncthreads();  // the other pair / the first tile lands on top of these staging areas
        }
    }
    const float bias = col_ok ? a.b[col] : 0.0f;
    const int64_t m_begin = (int64_t)split * a.rps, m_end = min(m_begin + a.rps, a.M);
    const int ntile = (int)((m_end - m_begin + FW_ROWS - 1) / FW_ROWS);
    constexpr int C4 = 2 * NS / 4;  // float4 per tile row (<= 64 lanes)
    const int k0 = 4 * lane;
    const int64_t lane_off = (lane < C4 && k0 < K) ? k0 : 0;  // lanes past the row's data re-read its first vector (in bounds; zeroed below / never read)
    const int nz = 2 * NS - K;                                 // columns to zero per row
    auto dma_tile = [&](int tile, int buf) {
        const int64_t row0 = m_begin + (int64_t)tile * FW_ROWS + 8 * w;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int64_t row = min(row0 + i, m_end - 1);  // (rows past the end: loaded again, never stored)
            const float* src = a.X + row * a.ldx + lane_off;
            // destination: wave-uniform LDS byte address in M0, the hardware adds lane * 16 bytes.  Inline asm rather than
            // __builtin_amdgcn_global_load_lds: after the builtin the compiler waits vmcnt(0) before the next LDS read (it cannot tell the
            // two tile buffers apart), which would put the DMA latency in front of every tile's MFMAs; the waits are explicit below.
            const unsigned dst = lds_base + (unsigned)((buf * FW_ROWS + 8 * w + i) * FW_KP * 4);
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(src), "s"(__builtin_amdgcn_readfirstlane(dst)) : "memory");
        }
    };
    const int zoff = lane < 8 * nz ? (8 * w + lane / nz) * FW_KP + K + lane % nz : -1;  // this lane's element of the first 64 (nz <= 8: all of them)
    auto zero_tail = [&](int buf) {
        if (zoff >= 0) sA[buf][zoff] = 0.0f;
        for (int i = lane + 64; i < 8 * nz; i += 64) {
            const int rr = i / nz, c = i - rr * nz;
            sA[buf][(8 * w + rr) * FW_KP + K + c] = 0.0f;
        }
    };
    if (ntile > 0) {
        dma_tile(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        zero_tail(0);
    }
    __syncthreads();
    auto mfma_tile = [&](int buf, f32x16& acc) {
        const float4* pa = reinterpret_cast<const float4*>(sA[buf] + r * FW_KP + half * NS);
#pragma unroll
        for (int s4 = 0; s4 < NS / 4; ++s4) {
            const float4 av = pa[s4];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, breg[4 * s4 + 0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, breg[4 * s4 + 1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, breg[4 * s4 + 2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, breg[4 * s4 + 3], acc, 0, 0, 0);
        }
    };
    auto end_of_tile = [&](int tile, bool stores_behind) {
        // The DMAs of the next tile have had the whole MFMA phase.  The sixteen row stores issued AFTER them have not -- a store is
        // acknowledged microseconds later under load -- and need not be waited for: the counter retires in issue order, so "at most the
        // sixteen youngest outstanding" says the DMAs have landed.
        if (stores_behind) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tile + 1 < ntile) zero_tail((tile & 1) ^ 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // (raw: a __syncthreads() here would also wait for stores issued after the wait above)
    };
    const bool pipelined = a.elu && (cb + 1) * FW_COLS <= a.N && ntile >= 2;
    if (pipelined) {
        // The two workgroups of a CU fall into step -- both in their MFMA phase, then both in their epilogue -- so an epilogue of its own
        // is time the MFMA pipe idles.  Here the epilogue of tile t - 1 (bias, ELU, 16 row stores) is issued BETWEEN the MFMAs of tile t,
        // whose dependent accumulator chain leaves ~60 issue cycles free after each one.  Every tile but the last is full and, on this
        // path, every column exists: the interleaved stores need no predicate (a predicate would split the block the scheduler mixes).
        f32x16 prev;
#pragma unroll
        for (int q = 0; q < 16; ++q) prev[q] = 0.0f;
        dma_tile(1, 1);
        mfma_tile(0, prev);
        end_of_tile(0, false);
        for (int tile = 1; tile < ntile; ++tile) {
            const int buf = tile & 1;
            if (tile + 1 < ntile) dma_tile(tile + 1, buf ^ 1);  // that buffer was last read in tile - 1: every wave is past its barrier
            f32x16 acc;
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[q] = 0.0f;
            float* y = a.Y + (m_begin + (int64_t)(tile - 1) * FW_ROWS + 4 * half) * a.ldy + col;
            const float4* pa = reinterpret_cast<const float4*>(sA[buf] + r * FW_KP + half * NS);
            constexpr int EVERY = (NS / 4) / 16 >= 1 ? (NS / 4) / 16 : 1;  // one output row per EVERY blocks of four MFMAs
#pragma unroll
            for (int s4 = 0; s4 < NS / 4; ++s4) {
                const float4 av = pa[s4];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, breg[4 * s4 + 0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, breg[4 * s4 + 1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, breg[4 * s4 + 2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, breg[4 * s4 + 3], acc, 0, 0, 0);
                if (s4 % EVERY == 0 && s4 / EVERY < 16) {
                    const int q = s4 / EVERY;
                    float x = prev[q] + bias;
                    const float e = (__expf(fminf(x, 0.0f)) - 1.0f) * a.alpha;
                    x = x <= 0.0f ? e : x;
                    *y = x;
                    y += ((q & 3) == 3 ? 5 : 1) * a.ldy;  // accumulator register q -> row (q & 3) + 8 (q >> 2) of this half
                    __builtin_amdgcn_sched_barrier(0);    // (left alone the scheduler sinks all sixteen stores below the last MFMA, right in front of the wait)
                }
            }
            static_assert(NS / 4 >= 16, "sixteen output rows need sixteen blocks");
            end_of_tile(tile, true);
            prev = acc;
        }
        const int64_t row0 = m_begin + (int64_t)(ntile - 1) * FW_ROWS;
        store_tile(prev, bias, a.alpha, a.elu, a.Y + (row0 + 4 * half) * a.ldy + col, a.ldy, (int)min<int64_t>(m_end - row0, FW_ROWS), half);
        return;
    }
    for (int tile = 0; tile < ntile; ++tile) {
        const int buf = tile & 1;
        if (tile + 1 < ntile) dma_tile(tile + 1, buf ^ 1);  // that buffer was last read in tile - 1: every wave is past the barrier below
        f32x16 acc;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = 0.0f;
        mfma_tile(buf, acc);
        const int64_t row0 = m_begin + (int64_t)tile * FW_ROWS;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // wait for the DMAs BEFORE the stores below go out
        if (tile + 1 < ntile) zero_tail(buf ^ 1);
        if (col_ok) store_tile(acc, bias, a.alpha, a.elu, a.Y + (row0 + 4 * half) * a.ldy + col, a.ldy, (int)min<int64_t>(m_end - row0, FW_ROWS), half);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
}

template <bool XVEC>
void launch_fwd(const FwArgs& a, dim3 grid, hipStream_t st) {
    const dim3 block(256);
    const int ns = (a.K + 1) / 2;
    if (ns <= 32) hipLaunchKernelGGL((k_mlp_fwd_elu<32, XVEC>), grid, block, 0, st, a);
    else if (ns <= 64) hipLaunchKernelGGL((k_mlp_fwd_elu<64, XVEC>), grid, block, 0, st, a);
    else if (XVEC && ns <= 96) hipLaunchKernelGGL((k_mlp_fwd_elu_dma<96>), grid, block, 0, st, a);
    else if (XVEC && ns <= 120) hipLaunchKernelGGL((k_mlp_fwd_elu_dma<120>), grid, block, 0, st, a);
    else if (XVEC) hipLaunchKernelGGL((k_mlp_fwd_elu_dma<128>), grid, block, 0, st, a);
    else if (ns <= 96) hipLaunchKernelGGL((k_mlp_fwd_elu<96, false>), grid, block, 0, st, a);  // unaligned rows: staged through registers (spills above 96 steps)
    else if (ns <= 120) hipLaunchKernelGGL((k_mlp_fwd_elu<120, false>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((k_mlp_fwd_elu<128, false>), grid, block, 0, st, a);
}

}  // namespace

extern "C" int imx_mlp_fwd_elu(int64_t M, int N, int K, const float* X_d, int64_t ldx, const float* W_d, const float* b_d, float elu_alpha,
                               int apply_elu, float* Y_d, int64_t ldy, imx_stream_t stream) {
    IMX_REQUIRE(M > 0 && N > 0 && K > 0 && X_d && W_d && b_d && Y_d, "imx_mlp_fwd_elu: bad arguments");
    IMX_REQUIRE(K <= FW_KMAX, "imx_mlp_fwd_elu: %d in-features (at most %d: the weights of a column live in registers)", K, FW_KMAX);
    IMX_REQUIRE(ldx >= K && ldy >= N, "imx_mlp_fwd_elu: row pitch smaller than the row");
    static int num_cu = 0;
    if (num_cu == 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&num_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || num_cu <= 0)
            num_cu = 256;
    }
    FwArgs a;
    a.M = M; a.N = N; a.K = K; a.X = X_d; a.ldx = ldx; a.W = W_d; a.b = b_d; a.Y = Y_d; a.ldy = ldy; a.alpha = elu_alpha; a.elu = apply_elu;
    a.ncb = (N + FW_COLS - 1) / FW_COLS;
    // two workgroups per CU (LDS 66 KB each, <= 256 VGPRs): row splits so that the grid is ~2 x #CU workgroups, each >= 4 tiles
    int64_t rs = std::max<int64_t>(1, (2 * (int64_t)num_cu) / a.ncb);
    rs = std::min<int64_t>(rs, std::max<int64_t>(1, M / (4 * FW_ROWS)));
    a.rps = ((M + rs - 1) / rs + FW_ROWS - 1) / FW_ROWS * FW_ROWS;
    a.rs = (int)((M + a.rps - 1) / a.rps);
    const bool xvec = (ldx % 4 == 0) && ((reinterpret_cast<uintptr_t>(X_d) & 15) == 0) && (K % 4 == 0 || ldx >= ((K + 3) & ~3));
    const dim3 grid((unsigned)(a.ncb * a.rs));
    if (xvec) launch_fwd<true>(a, grid, (hipStream_t)stream);
    else launch_fwd<false>(a, grid, (hipStream_t)stream);
    IMX_HIP(hipGetLastError());
    return 0;
}
