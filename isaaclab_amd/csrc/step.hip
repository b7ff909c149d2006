// Post-physics env-step kernels (gfx950): action affine, terminations + rewards + reset bookkeeping, observations.
//
// Layout: one environment per lane ("env-per-lane") for the reduction-shaped work (terminations, rewards: every term
// is a small per-env reduction over joints/bodies), one OUTPUT ELEMENT per lane for the observation matrix so that
// the (N,D) row-major store is perfectly coalesced.  Term tables are wave-uniform, so the interpreter's switch is a
// scalar branch; per-term constants come through the scalar cache.  All arithmetic is fp32 in the reference's
// association order (compile with -ffp-contract=off).
#include "imx_internal.h"
#include "imx_raycast.h"
// Scalar-register budget of the lean observation kernel.  On gfx950 a SIMD has 800 SGPRs, allocated in blocks of 16: the 106 the compiler
// takes by default (100 + VCC / FLAT_SCRATCH / XNACK) round to 112 = SEVEN waves per SIMD although the 57 VGPRs allow eight.  Capped,
// the compiler parks ~30 rarely used scalars in VGPR lanes; measured at 65 536 envs: 102 -> 142 us, 96 -> 132, 88 -> 132, 80 -> 126,
// 72 -> 129, 64 -> 135 (at 4096 envs all within 0.2 us).
#ifndef IMX_LEAN_SGPRS
#define IMX_LEAN_SGPRS 80
#endif

// ------------------------------------------------------------------------------------------------- helpers
IMX_DEV float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
IMX_DEV int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Sequential-order sum of f(ids[i]), i = 0..n-1 (torch.sum's left-to-right order for these short rows).  Full trips of 8 issue
// their loads before the first add; the remainder runs one by one -- no padded duplicates: for the force-history terms one
// element costs ~40 instructions and the slowest term is the kernel's critical path.
template <class F>
IMX_DEV float sum_ids(const int32_t* __restrict__ ids, int n, F f) {
    float acc = 0.0f;
    int i = 0;
    for (; i + 8 <= n; i += 8) {
        float x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = f(ids[i + u]);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += x[u];
    }
    for (; i + 4 <= n; i += 4) {
        float x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) x[u] = f(ids[i + u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += x[u];
    }
    for (; i < n; ++i) acc += f(ids[i]);
    return acc;
}
template <class F>
IMX_DEV float sum_range(int n, F f) {
    float acc = 0.0f;
    int i = 0;
    for (; i + 8 <= n; i += 8) {
        float x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = f(i + u);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += x[u];
    }
    for (; i + 4 <= n; i += 4) {
        float x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) x[u] = f(i + u);
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += x[u];
    }
    for (; i < n; ++i) acc += f(i);
    return acc;
}
template <class F>
IMX_DEV bool any_ids(const int32_t* __restrict__ ids, int n, F f) {
    bool acc = false;
    int i = 0;
    for (; i + 4 <= n; i += 4) {
        bool x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) x[u] = f(ids[i + u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = acc || x[u];
    }
    for (; i < n; ++i) acc = f(ids[i]) || acc;
    return acc;
}

// max over history of the force norm on body b (rewards.py:266, terminations.py:157): max_h sqrt(s_h) with s_h = (x^2 + y^2) + z^2.
// sqrtf is correctly rounded and monotone, so max_h sqrt(s_h) == sqrt(max_h s_h) bit for bit: ONE square root per body.
IMX_DEV float max_hist_force(const float* __restrict__ F, int64_t e, int H, int B, int b) {
    const float* f = F + ((size_t)e * H * B + b) * 3;
    float m = 0.0f;  // squared norms are >= 0
    int h = 0;
    for (; h + 3 <= H; h += 3) {  // the usual history_length = 3 in one trip
        float v[9];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const float* g = f + (size_t)(h + u) * B * 3;
            v[3 * u] = g[0]; v[3 * u + 1] = g[1]; v[3 * u + 2] = g[2];
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) m = fmaxf(m, (v[3 * u] * v[3 * u] + v[3 * u + 1] * v[3 * u + 1]) + v[3 * u + 2] * v[3 * u + 2]);
    }
    for (; h < H; ++h) {
        const float* g = f + (size_t)h * B * 3;
        m = fmaxf(m, (g[0] * g[0] + g[1] * g[1]) + g[2] * g[2]);
    }
    return sqrtf(m);
}

// scratch layout for k_term_rew (4-byte words, nw = number of env groups; sized for the smallest group: ceil(N/16) groups)
//   [0, nw*KA)               float  per-group partial sums of episode_sums over reset envs
//   [.., + nw*NT)            int    per-group counts of term_dones over reset envs
//   [.., + nw)               int    per-group reset counts
//   [.., + nw*64)            int    per-group compacted local reset ids
struct StepScratch {
    float* log_part;
    int* term_part;
    int* wave_cnt;
    int* ids_local;
};
// envs per workgroup of k_term_rew: every lane of a wave reads its own row of the (N, dof|bodies) state tensors, so a load instruction
// touches one cache line per live lane and the CU's L1 serves one line per clock -- at 4096 envs 64-env groups keep 64 of the 256
// CUs busy walking 64 lines per load (phase 1 measured 9.4 us), 16-env groups spread the same lines over all 256 CUs.  Full waves
// win once there are enough groups to fill the chip anyway.
static inline int step_group_size(int64_t N) { return N <= 8192 ? 16 : (N <= 16384 ? 32 : 64); }
static inline size_t step_scratch_words(int64_t N, int KA, int NT) {
    const size_t nw = (size_t)((N + 15) / 16);
    return nw * KA + nw * NT + nw + nw * 64;
}
static inline StepScratch carve(void* base, int64_t N, int KA, int NT) {
    const size_t nw = (size_t)((N + 15) / 16);
    StepScratch s;
    s.log_part = (float*)base;
    s.term_part = (int*)(s.log_part + nw * KA);
    s.wave_cnt = s.term_part + nw * NT;
    s.ids_local = s.wave_cnt + nw;
    return s;
}

// per-env frame table written by k_frame, read by k_obs: IMX_ES_WORDS floats per env, behind the step scratch
#define IMX_ES_WORDS 24  // 0-2 lin vel b, 3-5 ang vel b, 6-8 projected gravity, 9-11 root pos, 12-15 root quat, 16-17 scanner yaw quat (w, z),
                         // 18-19 free, 20 scanner flags (1 = cast this step, 2 = keep the hit heights), 21-22 sensor x / y, 23 data.pos_w z
static inline size_t frame_offset_bytes(const imx_plan_t* plan, int64_t N) {
    const size_t b = 4 * step_scratch_words(N, plan->nrew_all > 0 ? plan->nrew_all : 1, plan->nterm > 0 ? plan->nterm : 1);
    return (b + 255) & ~(size_t)255;
}

extern "C" size_t imx_plan_scratch_bytes(const imx_plan_t* plan, int64_t num_envs) {
    if (!plan || num_envs <= 0) return 0;
    return frame_offset_bytes(plan, num_envs) + (size_t)num_envs * IMX_ES_WORDS * sizeof(float) + 256;
}

// ------------------------------------------------------------------------------------------------- action affine
// ActionManager.process_action (action_manager.py:318-337): prev <- cur; cur <- a; per term
// processed = raw*scale + offset [clamp]  (joint_actions.py:130-139)
__global__ void k_action(PlanView P, int64_t N, const float* __restrict__ actions, float pre_clip, imx_state_t S,
                         imx_buffers_t Bf) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * P.A) return;
    const int64_t e = i / P.A;
    const int c = (int)(i - e * P.A);
    action_process_element(P, S, Bf, e, c, actions[i], pre_clip);
}

// The height scanner as a SensorBase (sensor_base.py:182-205,287-297; ray_caster.py:107-114,236-237): per-env timestamps decide whether
// this env's rays are cast this step (update_period), a per-env drift re-drawn at reset moves the sensor frame.  One row per env
// {timestamp, last update, drift xyz, data.pos_w z, outdated, step stamp}, advanced by ONE lane per env where the env's frame row is
// produced (the step kernel, which knows the reset flag, or k_frame) -- the observation kernel only reads the outcome from the frame
// row with its other scalar loads: no barrier, no extra round trip there.  The stamp tells a repeated call within the same step
// (ObservationManager.compute() by user code), which repeats the decision instead of advancing the clock again.
// Returns {flags (1 cast, 2 keep hit heights), sensor x, sensor y, data.pos_w z}.
IMX_DEV float4 scanner_step(const PlanView& P, const imx_buffers_t& Bf, int64_t e, uint32_t step, bool was_reset, float rx, float ry, float rz) {
    if (!P.scan_stateful) return make_float4(1.0f, rx, ry, rz);
    const uint64_t seed = (uint64_t)(uint32_t)Bf.counters[4] | ((uint64_t)(uint32_t)Bf.counters[5] << 32);  // the caller's drift seed
    float* row = Bf.scan_state + (size_t)e * 8;
    float ts = row[0], last = row[1], drx = row[2], dry = row[3], drz = row[4], pz_data = row[5];
    bool outdated = row[6] != 0.0f;
    const bool repeat = __float_as_uint(row[7]) == step + 1u;  // stamp = step + 1 (0 = never)
    if (!repeat) {
        for (int k = 0; k < P.scan_substeps; ++k) ts = ts + P.scan_dt;  // SensorBase.update(dt): once per physics step, fp32 like the tensor
        outdated = outdated || (ts - last + 1.0e-6f >= P.scan_period);
    } else {
        outdated = last == ts;  // the first call of this step refreshed the sensor: cast again (same pose, same hits)
    }
    // SensorBase.reset + RayCaster.reset: timers to zero, outdated, new drift.  Also on a repeated call of the same step: env.reset()
    // / env.reset(env_ids) after a step go through k_frame with the stamp k_term_rew left (manager_based_env.py:264-315 -> scene.reset);
    // for an env the step itself reset this repeats the same assignment (same seed, step, env -> same drift).
    if (was_reset) {
        ts = 0.0f; last = 0.0f; outdated = true;
        if (Bf.scan_drift_feed) {
            drx = Bf.scan_drift_feed[e * 3]; dry = Bf.scan_drift_feed[e * 3 + 1]; drz = Bf.scan_drift_feed[e * 3 + 2];
        } else {
            const float w = P.drift_hi - P.drift_lo;  // Tensor.uniform_(lo, hi)
            drx = uniform01(seed ^ 0xD21F7ull, step, (uint64_t)e * 3) * w + P.drift_lo;
            dry = uniform01(seed ^ 0xD21F7ull, step, (uint64_t)e * 3 + 1) * w + P.drift_lo;
            drz = uniform01(seed ^ 0xD21F7ull, step, (uint64_t)e * 3 + 2) * w + P.drift_lo;
        }
    }
    float flags = 0.0f;
    if (outdated) {  // _update_buffers_impl: pos_w = root pos + drift; _update_outdated_buffers: last update <- timestamp
        pz_data = rz + drz;
        last = ts;
        float ts2 = ts;  // will this env's sensor be outdated at the next step?  If not, its hit heights must survive this one
        for (int k = 0; k < P.scan_substeps; ++k) ts2 = ts2 + P.scan_dt;
        flags = !(ts2 - last + 1.0e-6f >= P.scan_period) ? 3.0f : 1.0f;
    }
    float4* o = reinterpret_cast<float4*>(row);
    o[0] = make_float4(ts, last, drx, dry);
    o[1] = make_float4(drz, pz_data, 0.0f, __uint_as_float(step + 1u));
    return make_float4(flags, rx + (outdated ? drx : 0.0f), ry + (outdated ? dry : 0.0f), pz_data);
}

#ifdef IMX_TRACE  // tools/trace_kobs.py only: per-wave start / end stamps (100 MHz wall clock) and placement; never in libimx.so
__device__ uint64_t* g_trace = nullptr;
extern "C" int imx_debug_trace(uint64_t* buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_trace), &buf, sizeof(buf)); }
#endif
#ifdef IMX_TRACE
#define IMX_STAMP(k) do { if (g_trace && lane == 0) g_trace[((size_t)blockIdx.x * 16 + wv) * 8 + (k)] = wall_clock64(); } while (0)
#else
#define IMX_STAMP(k) do { } while (0)
#endif
#define IMX_TR_MAX_WAVES 16
// The end of the step: ordered concatenation of the per-group reset-id lists (reset_env_ids = reset_buf.nonzero(),
// manager_based_rl_env.py:215), the reset count, and the Episode_* log reductions of RewardManager.reset / TerminationManager.reset
// (reward_manager.py:115-121, termination_manager.py:142-144) in a fixed order (deterministic, no float atomics).  ONE workgroup
// runs it after every group's partials are visible: the last-arriving workgroup of k_term_rew, or -- inside env.step() -- an extra
// workgroup of the observation kernel that follows (a kernel boundary instead of a fence + ticket: 4.9 us off the step kernel).
// The work splits in `nparts` workgroups so that it can ride along a kernel of small workgroups: part 0 orders the reset ids, parts
// 1.. share the log entries (each wave re-derives the reset count from the group counts: a handful of loads); nparts == 1 does both.
IMX_DEV void step_tail(const PlanView& P, int64_t N, const imx_buffers_t& Bf, const StepScratch& sc, int G, int part, int nparts) {
    __shared__ int s_scan[IMX_TR_MAX_WAVES];
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int NW = blockDim.x >> 6;
    const int nw = (int)((N + G - 1) / G);
    const int TB = blockDim.x;
    const int t = threadIdx.x;
    int total = 0;
    if (part == 0) {
        // exclusive scan of group counts: thread t owns groups [t*chunk, (t+1)*chunk); wave shuffles + per-wave totals
        const int chunk = (nw + TB - 1) / TB;
        int local = 0;
        for (int w = t * chunk; w < min((t + 1) * chunk, nw); ++w) local += __builtin_nontemporal_load(&sc.wave_cnt[w]);
        int incl = local;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_up(incl, o, 64);
            if (lane >= o) incl += up;
        }
        if (lane == 63) s_scan[wv] = incl;
        __syncthreads();
        int wave_base = 0;
        for (int w = 0; w < NW; ++w) {
            if (w < wv) wave_base += s_scan[w];
            total += s_scan[w];
        }
        if (t == 0) {
            Bf.counters[0] = total;  // number of reset envs
            // the runner's ep_infos sum (imx_buffers.log_accum): this step's extras["log"] -- refreshed below when something was reset,
            // the entries of the last refresh otherwise -- added to the running sum by the lane that owns the entry
            if (Bf.log_accum) {
                const int kc = P.nrew_all + P.nterm;
                Bf.log_accum[kc] += total > 0 ? (float)total : Bf.log_out[kc];
            }
        }
        int off = wave_base + incl - local;
        for (int w = t * chunk; w < min((t + 1) * chunk, nw); ++w) {
            const int c = __builtin_nontemporal_load(&sc.wave_cnt[w]);
            for (int j = 0; j < c; ++j)
                Bf.reset_env_ids[off + j] = (int64_t)w * G + __builtin_nontemporal_load(&sc.ids_local[w * 64 + j]);
            off += c;
        }
        if (total > 0 && t == 0) Bf.log_out[P.nrew_all + P.nterm] = (float)total;  // (after the read of the old value above: same lane)
        if (nparts > 1) return;
    } else {
        for (int w = lane; w < nw; w += 64) total += __builtin_nontemporal_load(&sc.wave_cnt[w]);
        total = wave_sum_i(total);
    }
    // reference only refreshes extras["log"] when something was reset (:216)
    // one wave per log entry: lanes stride over the groups, fixed-shape shuffle tree -> deterministic
    const int nlog = P.nrew_all + P.nterm;
    const int first = nparts > 1 ? (part - 1) * NW + wv : wv, stride = nparts > 1 ? (nparts - 1) * NW : NW;
    if (total > 0) {
        for (int k = first; k < nlog; k += stride) {
            float v;
            if (k < P.nrew_all) {
                float s = 0.0f;
                for (int w = lane; w < nw; w += 64) s += __builtin_nontemporal_load(&sc.log_part[(size_t)w * P.nrew_all + k]);
                s = wave_sum(s);
                v = s / (float)total / P.max_ep_len_s;
            } else {
                const int kt = k - P.nrew_all;
                int s = 0;
                for (int w = lane; w < nw; w += 64) s += __builtin_nontemporal_load(&sc.term_part[(size_t)w * P.nterm + kt]);
                s = wave_sum_i(s);
                v = (float)s;
            }
            if (lane == 0) {
                Bf.log_out[k] = v;
                if (Bf.log_accum) Bf.log_accum[k] += v;
            }
        }
    } else if (Bf.log_accum) {
        for (int k = first; k < nlog; k += stride)
            if (lane == 0) Bf.log_accum[k] += Bf.log_out[k];
    }
    // the orchestration's entries (imx_reset_orchestrate ran between the step kernel and this tail): Metrics/<command>/error_vel_xy|yaw =
    // mean over the reset envs of the metric before its reset (command_manager.py:123-149), Curriculum/terrain_levels = mean level over ALL
    // envs (curriculums.py:55, curriculum_manager.py:95-118); slots behind the reset count
    if (Bf.ev_part) {
        const int ng = (int)((N + 63) / 64);
        for (int j = first; j < 3; j += stride) {
            if (!((Bf.ev_flags >> (j < 2 ? 0 : 1)) & 1)) continue;
            const int slot = nlog + 1 + j;
            if (total > 0) {
                float s = 0.0f;
                for (int w = lane; w < ng; w += 64) s += __builtin_nontemporal_load(&Bf.ev_part[(size_t)w * 4 + j]);
                s = wave_sum(s);
                const float v = j < 2 ? s / (float)total : s / (float)N;
                if (lane == 0) {
                    Bf.log_out[slot] = v;
                    if (Bf.log_accum) Bf.log_accum[slot] += v;
                }
            } else if (Bf.log_accum && lane == 0) {
                Bf.log_accum[slot] += Bf.log_out[slot];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------- terminations + rewards
// Block = G consecutive envs x NW waves (NW <= 16).  Lane l of every wave owns env G*blockIdx + l ("env per lane").  The work items --
// nterm termination terms, then nrew reward terms -- are dealt round-robin to the waves, so that the state loads of ALL terms of an
// env are in flight together and the kernel is two memory round trips deep, not one chain per term:
//   phase 0  root state (and the episodic sum of the wave's first reward item) -- no table needed, issued first;
//   phase 1  each wave evaluates its items: record and id list through the scalar cache (wave-uniform, L2-resident), state loads,
//            a termination's bit into an LDS mask, a reward's RAW function value f and the env's running episodic sum into LDS;
//   phase 2  with every termination known (is_alive / is_terminated / the reset flag need them), each wave finishes its reward
//            items: value = f * weight * dt, episodic sum, step_reward, reset-log partial;
//   phase 3  wave 0 adds the values IN TERM ORDER (bit-identical to the reference's sequential `+=`), writes the env outputs and
//            the ordered reset ids of the group; the other waves store term_dones and the termination-log counts.
// What the per-wave timeline (tools/trace_step.py) showed on the way here: a memory round trip costs ~2 us at this size, the previous
// layout (8 waves, two terms per wave back to back, 64-env groups) walked ~6 of them in a row (24 us); with one item per wave the
// critical path became the SLOWEST item's instruction count -- the force-history terms evaluated 8 padded slots x 4 history slots
// with a square root each (9.7 us for undesired_contacts against 2 us for a joint sum) -- hence exact trip counts and one square
// root per body (max_hist_force), and the end of the step moved out of the kernel (step_tail: 4.9 us of fence + ticket).
__global__ void __launch_bounds__(64 * IMX_TR_MAX_WAVES)
k_term_rew(PlanView P, int64_t N, imx_state_t S, imx_buffers_t Bf, StepScratch sc, float* __restrict__ frame, int G, int defer_tail,
           imx_rollout_slot_t ro) {
    extern __shared__ int32_t smem[];
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int NW = blockDim.x >> 6;
    const int64_t grp = blockIdx.x;
    const int64_t e = grp * G + lane;
    const bool live = lane < G && e < N;
    const int64_t ec = live ? e : min(grp * G, N - 1);  // dead lanes compute on a valid env of the group (no extra cache lines), never store
    const int J = P.J, Bn = P.B, H = P.H, A = P.A;
    const int nterm = P.nterm, nrew = P.nrew;
    const int32_t* __restrict__ W = P.w;  // term tables: wave-uniform addresses -> scalar loads (L2-resident, a few KB)
    // The step counter (keys the in-kernel random streams, stamps the sensor rows) is advanced here without a read-modify-write race:
    // this kernel reads the shadow counters[3] the last observation launch left and publishes counters[2] = shadow + 1; the observation
    // kernel reads counters[2] and writes the shadow.
    const uint32_t step = (uint32_t)Bf.counters[3] + 1u;
    if (blockIdx.x == 0 && threadIdx.x == 0) Bf.counters[2] = (int32_t)step;
    IMX_STAMP(0);
    // LDS: [termination values: nterm x 64 u32 (bit 0 value, bit 1 time-out term)][f: nrew x 64][es: nrew x 64][val: nrew x 64];
    // every slot is written by exactly one wave before the barrier that publishes it: no zero fill, no atomics
    uint32_t* s_tv = reinterpret_cast<uint32_t*>(smem);
    float* s_f = reinterpret_cast<float*>(s_tv + (nterm > 0 ? nterm : 1) * 64);
    float* s_es = s_f + nrew * 64;
    float* s_val = s_es + nrew * 64;

    // -- phase 0: everything that needs no table is issued at once: root state, and the episodic sum of this wave's first reward item
    const int first_rew = wv >= nterm ? wv - nterm : wv - nterm + ((nterm - wv + NW - 1) / NW) * NW;  // first item >= nterm of this wave
    const float es_first = (live && first_rew < nrew) ? Bf.episode_sums[(size_t)first_rew * N + e] : 0.0f;
    // slot t of the rollout storage (imx_terminations_rewards_rollout): what wave 0 needs at the very end is requested now
    const bool ro_on = ro.rewards_out != nullptr && wv == 0 && live;
    const float ro_value = ro_on ? ro.value_t[e] : 0.0f;
    const float ro_cur_rew = (ro_on && ro.cur_reward_sum) ? ro.cur_reward_sum[e] : 0.0f;
    const float ro_cur_len = (ro_on && ro.cur_ep_len) ? ro.cur_ep_len[e] : 0.0f;
    const float4 q4 = reinterpret_cast<const float4*>(S.root_quat_w)[ec];
    const float qw = q4.x, qx = q4.y, qy = q4.z, qz = q4.w;
    const float lwx = S.root_lin_vel_w[ec * 3], lwy = S.root_lin_vel_w[ec * 3 + 1], lwz = S.root_lin_vel_w[ec * 3 + 2];
    const float awx = S.root_ang_vel_w[ec * 3], awy = S.root_ang_vel_w[ec * 3 + 1], awz = S.root_ang_vel_w[ec * 3 + 2];
    const int64_t ep = Bf.episode_length_buf[ec] + 1;  // manager_based_rl_env.py:200
    const float cmdx = P.CMD > 0 ? S.command[ec * P.CMD + 0] : 0.0f;
    const float cmdy = P.CMD > 1 ? S.command[ec * P.CMD + 1] : 0.0f;
    const float cmdz = P.CMD > 2 ? S.command[ec * P.CMD + 2] : 0.0f;
    float lbx, lby, lbz, abx, aby, abz, pgx, pgy, pgz;
    quat_rotate_inverse(qw, qx, qy, qz, lwx, lwy, lwz, lbx, lby, lbz);
    quat_rotate_inverse(qw, qx, qy, qz, awx, awy, awz, abx, aby, abz);
    quat_rotate_inverse(qw, qx, qy, qz, P.gx, P.gy, P.gz, pgx, pgy, pgz);

    // -- the env's frame table for the observation kernel of the same step (what k_frame writes: same functions, same inputs, so
    //    bit-identical); the last wave has the fewest items
    if (frame && wv == NW - 1 && live) {
        float4 o[5];
        o[0] = make_float4(lbx, lby, lbz, abx);
        o[1] = make_float4(aby, abz, pgx, pgy);
        o[2] = make_float4(pgz, S.root_pos_w[e * 3], S.root_pos_w[e * 3 + 1], S.root_pos_w[e * 3 + 2]);
        o[3] = q4;
        o[4] = make_float4(1.0f, 0.0f, 0.0f, 0.0f);
        if (P.R > 0 && P.ray_yaw_only) yaw_quat_wz(qw, qx, qy, qz, o[4].x, o[4].y);
        float4* dst = reinterpret_cast<float4*>(frame) + e * 6;
#pragma unroll
        for (int k = 0; k < 5; ++k) dst[k] = o[k];
        if (P.R > 0 && !P.scan_stateful) dst[5] = make_float4(1.0f, o[2].y, o[2].z, o[2].w);  // a sensor without clock: cast, from the root
    }
    IMX_STAMP(1);
    IMX_STAMP(2);

    // -- phase 1: one item per trip.  TerminationManager.compute (termination_manager.py:151-174) / the reward functions
    const bool moving = sqrtf(cmdx * cmdx + cmdy * cmdy) > 0.1f;  // torch.norm(cmd[:, :2]) > 0.1
    for (int item = wv; item < nterm + nrew; item += NW) {
        if (item < nterm) {
            const int k = item;
            const int32_t* r = W + P.term_off + k * IMX_REC_WORDS;
            const int op = r[IMX_R_OP];
            const int n = r[IMX_R_NIDS];
            const int32_t* ids = W + (op == IMX_T_COMMAND_RESAMPLE ? 0 : r[IMX_R_IDS_OFF]);
            const float p0 = f_of(r[IMX_R_P0]), p1 = f_of(r[IMX_R_P1]);
            bool v = false;
            switch (op) {
                case IMX_T_TIME_OUT: v = ep >= (int64_t)P.max_ep_len; break;
                case IMX_T_ILLEGAL_CONTACT:
                    v = any_ids(ids, n, [&](int b) { return max_hist_force(S.net_forces_w_history, ec, H, Bn, b) > p0; });
                    break;
                case IMX_T_JOINT_POS_MANUAL_LIMIT:
                    v = any_ids(ids, n, [&](int j) { const float q = S.joint_pos[ec * J + j]; return (q > p1) || (q < p0); });
                    break;
                case IMX_T_BAD_ORIENTATION: v = fabsf(acosf(-pgz)) > p0; break;
                case IMX_T_ROOT_HEIGHT_BELOW_MIN: v = S.root_pos_w[ec * 3 + 2] < p0; break;
                case IMX_T_JOINT_VEL_LIMIT:
                    v = any_ids(ids, n, [&](int j) { return fabsf(S.joint_vel[ec * J + j]) > S.soft_joint_vel_limits[ec * J + j]; });
                    break;
                case IMX_T_JOINT_VEL_MANUAL_LIMIT:
                    v = any_ids(ids, n, [&](int j) { return fabsf(S.joint_vel[ec * J + j]) > p0; });
                    break;
                case IMX_T_JOINT_EFFORT_LIMIT:  // torch.isclose(computed, applied): |a-b| <= atol + rtol*|b|
                    v = any_ids(ids, n, [&](int j) {
                        const float a = S.computed_torque[ec * J + j], b = S.applied_torque[ec * J + j];
                        return fabsf(a - b) <= 1.0e-8f + 1.0e-5f * fabsf(b);
                    });
                    break;
                case IMX_T_TERRAIN_OUT_OF_BOUNDS:
                    v = (fabsf(S.root_pos_w[ec * 3]) > p0) || (fabsf(S.root_pos_w[ec * 3 + 1]) > p1);
                    break;
                case IMX_T_EXTERNAL: v = S.ext_term[ec * (int64_t)P.w[IMX_H_NEXT_TERM] + r[IMX_R_AUX0]] != 0; break;
                case IMX_T_COMMAND_RESAMPLE:  // (time_left <= step_dt) & (command_counter == num_resamples)
                    v = (S.command_time_left[ec] <= p0) && (S.command_counter[ec] == (int64_t)n);
                    break;
                default: break;
            }
            s_tv[k * 64 + lane] = (v ? 1u : 0u) | (r[IMX_R_WEIGHT] ? 2u : 0u);  // bit 1: a time-out term (termination_manager.py:166-169)
            continue;
        }
        const int k = item - nterm;
        const int32_t* r = W + P.rew_off + k * IMX_REC_WORDS;
        const float es0 = k == first_rew ? es_first : (live ? Bf.episode_sums[(size_t)k * N + e] : 0.0f);
        float f = 0.0f;
        if (f_of(r[IMX_R_WEIGHT]) != 0.0f) {  // a zero-weight term is not evaluated (reward_manager.py:145)
            const int op = r[IMX_R_OP];
            const int n = r[IMX_R_NIDS];
            const int32_t* ids = W + r[IMX_R_IDS_OFF];
            const float p0 = f_of(r[IMX_R_P0]);
            switch (op) {
                // IS_ALIVE / IS_TERMINATED / IS_TERMINATED_TERM need the termination results: phase 2
                case IMX_W_LIN_VEL_Z_L2: f = lbz * lbz; break;
                case IMX_W_ANG_VEL_XY_L2: f = abx * abx + aby * aby; break;
                case IMX_W_FLAT_ORIENTATION_L2: f = pgx * pgx + pgy * pgy; break;
                case IMX_W_BASE_HEIGHT_L2: { const float d = S.root_pos_w[ec * 3 + 2] - p0; f = d * d; } break;
                case IMX_W_JOINT_TORQUES_L2:
                    f = sum_ids(ids, n, [&](int j) { const float x = S.applied_torque[ec * J + j]; return x * x; });
                    break;
                case IMX_W_JOINT_VEL_L1: f = sum_ids(ids, n, [&](int j) { return fabsf(S.joint_vel[ec * J + j]); }); break;
                case IMX_W_JOINT_VEL_L2:
                    f = sum_ids(ids, n, [&](int j) { const float x = S.joint_vel[ec * J + j]; return x * x; });
                    break;
                case IMX_W_JOINT_ACC_L2:
                    f = sum_ids(ids, n, [&](int j) { const float x = S.joint_acc[ec * J + j]; return x * x; });
                    break;
                case IMX_W_JOINT_DEVIATION_L1:
                    f = sum_ids(ids, n, [&](int j) { return fabsf(S.joint_pos[ec * J + j] - S.default_joint_pos[ec * J + j]); });
                    break;
                case IMX_W_JOINT_POS_LIMITS:
                    f = sum_ids(ids, n, [&](int j) {
                        const float q = S.joint_pos[ec * J + j];
                        const float2 lim = reinterpret_cast<const float2*>(S.soft_joint_pos_limits)[ec * J + j];
                        float o = -fminf(q - lim.x, 0.0f);
                        o += fmaxf(q - lim.y, 0.0f);
                        return o;
                    });
                    break;
                case IMX_W_JOINT_VEL_LIMITS:
                    f = sum_ids(ids, n, [&](int j) {
                        const float o = fabsf(S.joint_vel[ec * J + j]) - S.soft_joint_vel_limits[ec * J + j] * p0;
                        return fminf(fmaxf(o, 0.0f), 1.0f);
                    });
                    break;
                case IMX_W_APPLIED_TORQUE_LIMITS:
                    f = sum_ids(ids, n, [&](int j) { return fabsf(S.applied_torque[ec * J + j] - S.computed_torque[ec * J + j]); });
                    break;
                case IMX_W_ACTION_RATE_L2:
                    f = sum_range(A, [&](int i) { const float d = Bf.action[ec * A + i] - Bf.prev_action[ec * A + i]; return d * d; });
                    break;
                case IMX_W_ACTION_L2:
                    f = sum_range(A, [&](int i) { const float a = Bf.action[ec * A + i]; return a * a; });
                    break;
                case IMX_W_UNDESIRED_CONTACTS:
                    f = sum_ids(ids, n, [&](int b) { return (max_hist_force(S.net_forces_w_history, ec, H, Bn, b) > p0) ? 1.0f : 0.0f; });
                    break;
                case IMX_W_CONTACT_FORCES:
                    f = sum_ids(ids, n, [&](int b) { return fmaxf(max_hist_force(S.net_forces_w_history, ec, H, Bn, b) - p0, 0.0f); });
                    break;
                case IMX_W_TRACK_LIN_VEL_XY_EXP: {
                    const float ex = cmdx - lbx, ey = cmdy - lby;
                    f = expf(-(ex * ex + ey * ey) / p0);  // p0 = std**2
                } break;
                case IMX_W_TRACK_ANG_VEL_Z_EXP: { const float ez = cmdz - abz; f = expf(-(ez * ez) / p0); } break;
                case IMX_W_FEET_AIR_TIME: {
                    // first_contact = (cct > 0) * (cct < dt + abs_tol); p1 = float32(step_dt + 1e-8)
                    const float p1 = f_of(r[IMX_R_P1]);
                    f = sum_ids(ids, n, [&](int b) {
                        const float cct = S.current_contact_time[ec * Bn + b];
                        const float fc = (cct > 0.0f && cct < p1) ? 1.0f : 0.0f;
                        return (S.last_air_time[ec * Bn + b] - p0) * fc;
                    });
                    f *= moving ? 1.0f : 0.0f;
                } break;
                case IMX_W_FEET_AIR_TIME_POSITIVE_BIPED: {
                    int n_contact = 0;
                    float mn = __builtin_huge_valf();
                    for (int i = 0; i < n; ++i) n_contact += (S.current_contact_time[ec * Bn + ids[i]] > 0.0f) ? 1 : 0;
                    for (int i = 0; i < n; ++i) {
                        const float ct = S.current_contact_time[ec * Bn + ids[i]], at = S.current_air_time[ec * Bn + ids[i]];
                        const float in_mode = (ct > 0.0f) ? ct : at;
                        mn = fminf(mn, (n_contact == 1) ? in_mode : 0.0f);
                    }
                    f = fminf(mn, p0);
                    f *= moving ? 1.0f : 0.0f;
                } break;
                case IMX_W_FEET_SLIDE: {
                    const int32_t* ids2 = W + r[IMX_R_IDS2_OFF];
                    for (int i = 0; i < n; ++i) {
                        const float c = (max_hist_force(S.net_forces_w_history, ec, H, Bn, ids[i]) > 1.0f) ? 1.0f : 0.0f;
                        const float* v = S.body_lin_vel_w + ((size_t)ec * P.NB + ids2[i]) * 3;
                        f += sqrtf(v[0] * v[0] + v[1] * v[1]) * c;
                    }
                } break;
                case IMX_W_TRACK_LIN_VEL_XY_YAW_FRAME_EXP: {
                    float yw, yz, vx, vy, vz;
                    yaw_quat_wz(qw, qx, qy, qz, yw, yz);
                    quat_rotate_inverse(yw, 0.0f, 0.0f, yz, lwx, lwy, lwz, vx, vy, vz);
                    const float ex = cmdx - vx, ey = cmdy - vy;
                    f = expf(-(ex * ex + ey * ey) / p0);
                } break;
                case IMX_W_TRACK_ANG_VEL_Z_WORLD_EXP: { const float ez = cmdz - awz; f = expf(-(ez * ez) / p0); } break;
                case IMX_W_JOINT_POS_TARGET_L2:
                    f = sum_ids(ids, n, [&](int j) { const float d = wrap_to_pi(S.joint_pos[ec * J + j]) - p0; return d * d; });
                    break;
                case IMX_W_EXTERNAL: f = S.ext_reward[ec * (int64_t)P.w[IMX_H_NEXT_REW] + r[IMX_R_AUX0]]; break;
                case IMX_W_BODY_LIN_ACC_L2:  // sum over bodies of ||body_lin_acc_w|| (rewards.py:125-128)
                    f = sum_ids(ids, n, [&](int b) {
                        const float* a = S.body_lin_acc_w + ((size_t)ec * P.NB + b) * 3;
                        return norm3(a[0], a[1], a[2]);
                    });
                    break;
                default: break;
            }
        }
        s_f[k * 64 + lane] = f;
        s_es[k * 64 + lane] = es0;
    }
    IMX_STAMP(3);
    __syncthreads();
    IMX_STAMP(4);

    // -- phase 2: every termination is known
    uint32_t term_bits = 0u, trunc_mask = 0u;
    for (int k = 0; k < nterm; ++k) {
        const uint32_t x = s_tv[k * 64 + lane];
        term_bits |= (x & 1u) << k;
        trunc_mask |= ((x >> 1) & 1u) << k;
    }
    const bool truncated = (term_bits & trunc_mask) != 0u, terminated = (term_bits & ~trunc_mask) != 0u;
    const bool reset = live && (terminated || truncated);
    if (frame && P.scan_stateful && wv == NW - 1 && live) {  // the sensor's clock, now that the env's reset flag is known (scene.reset(env_ids))
        reinterpret_cast<float4*>(frame)[e * 6 + 5] =
            scanner_step(P, Bf, e, step, reset, S.root_pos_w[e * 3], S.root_pos_w[e * 3 + 1], S.root_pos_w[e * 3 + 2]);
    }
    // RewardManager.compute (reward_manager.py:128-157): value = f * w * dt; sums += value; step_reward = value/dt
    const float dt = P.step_dt;
    for (int k = wv; k < nrew; k += NW) {
        const int32_t* r = W + P.rew_off + k * IMX_REC_WORDS;
        const float weight = f_of(r[IMX_R_WEIGHT]);
        const float es0 = s_es[k * 64 + lane];
        float es = es0, value = 0.0f;
        if (weight != 0.0f) {
            float f = s_f[k * 64 + lane];
            const int op = r[IMX_R_OP];
            if (op == IMX_W_IS_ALIVE) f = terminated ? 0.0f : 1.0f;
            if (op == IMX_W_IS_TERMINATED) f = terminated ? 1.0f : 0.0f;
            if (op == IMX_W_IS_TERMINATED_TERM) {
                const int n = r[IMX_R_NIDS];
                const int32_t* ids = W + r[IMX_R_IDS_OFF];
                float sum = 0.0f;
                for (int i = 0; i < n; ++i) sum += ((term_bits >> ids[i]) & 1u) ? 1.0f : 0.0f;
                f = sum * (truncated ? 0.0f : 1.0f);
            }
            value = f * weight * dt;
            es = es0 + value;
            if (live) Bf.step_reward[(size_t)e * nrew + k] = value / dt;
        }
        // (a skipped term leaves step_reward as it was and still takes part in the reset / log pass, reward_manager.py:100-126,145)
        s_val[k * 64 + lane] = value;
        if (live && (weight != 0.0f || reset)) Bf.episode_sums[(size_t)k * N + e] = reset ? 0.0f : es;
        // RewardManager.reset log (reward_manager.py:115-121): mean over reset envs of the episodic sum
        const float part = wave_sum(reset ? es : 0.0f);
        if (lane == 0) sc.log_part[grp * nrew + k] = part;
    }
    // termination bookkeeping by the waves from the far end (wave 0 is busy below)
    for (int k = NW - 1 - wv; k < nterm; k += NW) {
        const bool v = (term_bits >> k) & 1u;
        if (live) Bf.term_dones[(size_t)k * N + e] = v ? 1 : 0;
        // TerminationManager.reset log (termination_manager.py:142-144): count_nonzero(term_dones[reset ids])
        const int c = wave_sum_i((reset && v) ? 1 : 0);
        if (lane == 0) sc.term_part[grp * nterm + k] = c;
    }
    IMX_STAMP(5);
    __syncthreads();
    IMX_STAMP(6);

    // -- phase 3
    if (wv == 0) {
        float reward = 0.0f;
        for (int k = 0; k < nrew; ++k) reward += s_val[k * 64 + lane];  // term order (a skipped term holds +0: x + 0 == x bit for bit)
        // -- outputs + manager-side _reset_idx (manager_based_rl_env.py:347-392)
        if (live) {
            Bf.reward_buf[e] = reward;
            Bf.terminated[e] = terminated ? 1 : 0;
            Bf.truncated[e] = truncated ? 1 : 0;
            Bf.reset_buf[e] = reset ? 1 : 0;
            Bf.episode_length_buf[e] = reset ? 0 : ep;
            if (reset)
                for (int i = 0; i < A; ++i) {  // ActionManager.reset (action_manager.py:306-316)
                    Bf.action[e * A + i] = 0.0f;
                    Bf.prev_action[e * A + i] = 0.0f;
                }
        }
        // slot t of the RolloutStorage (what imx_rollout_post does in a launch of its own): RslRlVecEnvWrapper.step's dones
        // (vecenv_wrapper.py:178), PPO.process_env_step's time-out bootstrap, the runner's episode book-keeping -- same expressions,
        // bit-identical
        if (ro.rewards_out) {
            float s_r = 0.0f, s_l = 0.0f, s_c = 0.0f;
            if (live) {
                ro.rewards_out[e] = ro.bootstrap_time_outs ? reward + ro.gamma * (ro_value * (truncated ? 1.0f : 0.0f)) : reward;
                ro.dones_out[e] = reset ? 1 : 0;
                if (ro.cur_reward_sum) {
                    const float cr = ro_cur_rew + reward, cl = ro_cur_len + 1.0f;
                    if (reset) { s_r = cr; s_l = cl; s_c = 1.0f; }
                    ro.cur_reward_sum[e] = reset ? 0.0f : cr;
                    ro.cur_ep_len[e] = reset ? 0.0f : cl;
                }
            }
            if (ro.ep_stats3) {
                s_r = wave_sum(s_r); s_l = wave_sum(s_l); s_c = wave_sum(s_c);
                if (lane == 0 && s_c > 0.0f) {
                    atomicAdd(&ro.ep_stats3[0], s_r); atomicAdd(&ro.ep_stats3[1], s_l); atomicAdd(&ro.ep_stats3[2], s_c);
                }
            }
        }
        // ordered compaction inside the group: reset_env_ids = reset_buf.nonzero() (manager_based_rl_env.py:215)
        const unsigned long long ballot = __ballot(reset);
        const int before = __popcll(ballot & ((1ull << lane) - 1ull));
        if (reset) sc.ids_local[grp * 64 + before] = lane;
        if (lane == 0) sc.wave_cnt[grp] = __popcll(ballot);
    }

    IMX_STAMP(7);
    if (defer_tail) return;  // imx_observations of the same step finishes (step_tail in an extra workgroup of k_obs)
    // producer side (cdna_hip_programming.md G16, R1): every storing wave drains its stores, the block meets at the
    // barrier, ONE lane releases at agent scope and takes the ticket
    __shared__ int s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int ticket = atomicAdd(&Bf.counters[1], 1);
        s_last = (ticket == (int)gridDim.x - 1);
        if (s_last) {  // consumer side: one agent-scope acquire, completed before the barrier releases the readers
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            Bf.counters[1] = 0;  // re-arm the ticket
        }
    }
    __syncthreads();
    if (!s_last) return;
    step_tail(P, N, Bf, sc, G, 0, 1);
}

// ------------------------------------------------------------------------------------------------- observations
// One WAVE = one environment (4 envs per 256-thread block).  Every lane first derives the env's root-frame vectors
// and scanner yaw from the same 13 root-state floats (wave-uniform loads: one transaction, no LDS, no barrier);
// then lane l produces the output columns xcol[l], xcol[l+64], ... (ray columns first) and stores obs[e*D + c]:
// within a term the 64 lanes of a wave write 64 consecutive floats.
//
// xcol is the per-column expansion of the observation records built once by imx_plan_create (16 words per column:
// everything a lane needs arrives with four 16-byte loads instead of a chain order -> column -> record -> id table).
// Height-scanner rays (yaw-only frame, vertical direction) take a fast path that keeps up to four rays of a lane
// in flight: cell-table loads of all four, then the 48-byte triangle records of all four, then the Woop tests.
#define IMX_OBS_ENVS_PER_BLOCK 4
#define IMX_XCOL_WORDS 16
enum { XC_COL = 0, XC_OP, XC_J, XC_FLAGS, XC_P0, XC_NLO, XC_NHI, XC_CLO, XC_CHI, XC_SCALE, XC_AUX, XC_RX, XC_RY, XC_RZ, XC_HIST, XC_HSTRIDE };

struct XCol {
    int4 a, b, c, d;
};
IMX_DEV XCol load_xcol(const int32_t* __restrict__ W, int off, int i) {
    const int4* p = reinterpret_cast<const int4*>(W + off) + (size_t)i * 4;
    XCol x;
    x.a = p[0]; x.b = p[1]; x.c = p[2]; x.d = p[3];
    return x;
}

// The noise term of one element: uniform_noise u * (n_max - n_min) + n_min (noise_model.py:62-66; constant_noise is the case n_min == n_max)
// or gaussian_noise mean + std * z (:87-92).  The sample is the fed one (the reference's recorded rand_like / randn_like draw) or comes
// from the counter-based generator (Box-Muller on two of its uniforms for z).
// GAUSS = false: plans without a gaussian term (the LEAN kernels: logf / cosf / sqrtf in their instruction stream cost 2 % at 4096 envs
// and 7 % at 65 536, taken or not; a plan with gaussian noise runs the general kernels).
template <bool GAUSS>
IMX_DEV float noise_sample(int flags, float lo, float hi, const float* __restrict__ noise_u, uint64_t seed, uint32_t step, int64_t e, int D, int c) {
    if (GAUSS && (flags & IMX_F_NOISE_GAUSS)) {
        float z;
        if (noise_u) {
            z = noise_u[e * D + c];
        } else {
            const float u1 = uniform01(seed, step, (uint64_t)e * D + c), u2 = uniform01(seed ^ 0x6A09E667F3BCC909ull, step, (uint64_t)e * D + c);
            z = sqrtf(-2.0f * logf(1.0f - u1)) * cosf(6.28318530717958647692f * u2);  // 1 - u1 in (0, 1]: no log(0)
        }
        return lo + hi * z;
    }
    const float u = noise_u ? noise_u[e * D + c] : uniform01(seed, step, (uint64_t)e * D + c);
    return u * (hi - lo) + lo;
}

// D = width of the whole column space (all groups side by side), gbase = first column of this entry's group in it: the parity-mode
// uniforms are one (N, D) array, group after group
template <bool GAUSS>
IMX_DEV float obs_post(const XCol& x, float v, int corrupt, const float* __restrict__ noise_u, uint64_t seed, uint32_t step,
                       int64_t e, int D, int gbase) {
    const int flags = x.a.w;
    if (corrupt && (flags & (IMX_F_NOISE_ADD | IMX_F_NOISE_SCALE | IMX_F_NOISE_ABS))) {
        // a term with a history window draws for its first (oldest-slot) columns, like rand_like on the (N, d) term value
        const int c = gbase + x.a.x - (x.d.z - 1) * x.d.w;
        const float lo = f_of(x.b.y), hi = f_of(x.b.z);
        const float nz = noise_sample<GAUSS>(flags, lo, hi, noise_u, seed, step, e, D, c);
        v = (flags & IMX_F_NOISE_ADD) ? v + nz : ((flags & IMX_F_NOISE_SCALE) ? v * nz : nz);
    }
    if (flags & IMX_F_CLIP) v = fminf(fmaxf(v, f_of(x.b.w)), f_of(x.c.x));
    if (flags & IMX_F_SCALE) v = v * f_of(x.c.y);
    return v;
}

// ObservationTermCfg.modifiers (observation_manager.py:310-312; utils/modifiers/modifier.py): the term's modifier program,
// run on the raw value of ONE element; filter / integrator state of that element lives in the env's mod_state row at
// st[(soff + k) * d] (element-minor, so the lanes of a term touch consecutive floats).  `zero`: the env was reset this step
// (ObservationManager.reset zeroes the state before the next compute) -- old state is ignored, new state is written.
IMX_DEV float apply_modifiers(const int32_t* __restrict__ W, int xmod, float v, float* __restrict__ row, bool zero) {
    const int4 m = *reinterpret_cast<const int4*>(W + xmod);
    const int po = m.x, pn = m.y, d = m.w;
    float* st = row + m.z;
    for (int q = 0; q < pn; q += 4) {
        const int op = W[po + q];
        const float a = f_of(W[po + q + 1]), b = f_of(W[po + q + 2]);
        const int soff = W[po + q + 3];
        switch (op) {
            case IMX_M_SCALE: v = v * a; break;
            case IMX_M_BIAS: v = v + a; break;
            case IMX_M_CLIP: v = fminf(fmaxf(v, a), b); break;
            case IMX_M_INTEGRATOR: {  // integral += (data + y_prev) / 2 * dt; y_prev = data (modifier.py:247-259)
                float* s = st + soff * d;
                const float yp = zero ? 0.0f : s[d];
                const float integ = (zero ? 0.0f : s[0]) + (v + yp) / 2.0f * a;
                s[0] = integ; s[d] = v;
                v = integ;
            } break;
            case IMX_M_DIGITAL_FILTER: {  // y = x_n . B - y_n . A with both windows rolled by one (modifier.py:160-176)
                const int na = W[po + q + 1], nb = W[po + q + 2];
                float* xs = st + soff * d;
                float* ys = xs + nb * d;
                const int32_t* A = W + po + q + 4;
                const int32_t* B = A + na;
                float accx = 0.0f, accy = 0.0f, carry = v;
                for (int k = 0; k < nb; ++k) {
                    const float old = zero ? 0.0f : xs[k * d];
                    xs[k * d] = carry;
                    accx += carry * f_of(B[k]);
                    carry = old;
                }
                for (int k = 0; k < na; ++k) accy += (zero ? 0.0f : ys[k * d]) * f_of(A[k]);
                const float y = accx - accy;
                carry = y;
                for (int k = 0; k < na; ++k) {
                    const float old = zero ? 0.0f : ys[k * d];
                    ys[k * d] = carry;
                    carry = old;
                }
                v = y;
                q += na + nb;
            } break;
            default: break;
        }
    }
    return v;
}

// k_frame: one lane per env -- root-frame vectors (ArticulationData.root_lin_vel_b / root_ang_vel_b /
// projected_gravity_b), sensor position and the yaw-only sensor quaternion (yaw_quat, utils/math.py:521-542), once per
// env per step instead of once per wave of k_obs (PMC: the transcendental prologue was ~40 % of k_obs's VALU work).
__global__ void __launch_bounds__(64)
k_frame(PlanView P, int64_t N, imx_state_t S, imx_buffers_t Bf, float* __restrict__ frame, int fill_all) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N) return;
    const float4 q4 = reinterpret_cast<const float4*>(S.root_quat_w)[e];
    float4 o[5];
    quat_rotate_inverse(q4.x, q4.y, q4.z, q4.w, S.root_lin_vel_w[e * 3], S.root_lin_vel_w[e * 3 + 1],
                        S.root_lin_vel_w[e * 3 + 2], o[0].x, o[0].y, o[0].z);
    quat_rotate_inverse(q4.x, q4.y, q4.z, q4.w, S.root_ang_vel_w[e * 3], S.root_ang_vel_w[e * 3 + 1],
                        S.root_ang_vel_w[e * 3 + 2], o[0].w, o[1].x, o[1].y);
    quat_rotate_inverse(q4.x, q4.y, q4.z, q4.w, P.gx, P.gy, P.gz, o[1].z, o[1].w, o[2].x);
    o[2].y = S.root_pos_w[e * 3]; o[2].z = S.root_pos_w[e * 3 + 1]; o[2].w = S.root_pos_w[e * 3 + 2];
    o[3] = q4;
    o[4] = make_float4(1.0f, 0.0f, 0.0f, 0.0f);
    if (P.R > 0 && P.ray_yaw_only) yaw_quat_wz(q4.x, q4.y, q4.z, q4.w, o[4].x, o[4].y);
    float4* dst = reinterpret_cast<float4*>(frame) + e * 6;
#pragma unroll
    for (int k = 0; k < 5; ++k) dst[k] = o[k];
    if (P.R > 0)
        dst[5] = scanner_step(P, Bf, e, (uint32_t)Bf.counters[2], fill_all || (Bf.reset_buf && Bf.reset_buf[e]), o[2].y, o[2].z, o[2].w);
}

// k_obs: one BLOCK = one environment, one lane = one output column (ray columns first in xcol).  No prologue, no
// barrier: the env's 20-float frame is read with wave-uniform loads (scalar cache), then every lane is an independent,
// register-light stream: xcol (four 16-byte loads) -> value -> noise/clip/scale -> obs[e*D + c]; consecutive lanes
// write consecutive floats of one obs row.  The ray path is cast_ray_vertical (imx_raycast.h): cell descriptor and
// the four shared lattice corners are loaded together -- one dependent memory level per ray on height-field terrain.
// value of one non-ray observation column (every op but HEIGHT_SCAN); es = the env's frame (k_frame)
IMX_DEV float obs_plain_value(const PlanView& P, const imx_state_t& S, const imx_buffers_t& Bf, const float* __restrict__ es,
                              int64_t e, const XCol& x) {
    const int32_t* __restrict__ W = P.w;
    const int op = x.a.y, j = x.a.z, flags = x.a.w, aux = x.c.z, J = P.J;
    switch (op) {
        case IMX_O_BASE_POS_Z: return es[11];
        case IMX_O_BASE_LIN_VEL: return es[0 + j];
        case IMX_O_BASE_ANG_VEL: return es[3 + j];
        case IMX_O_PROJECTED_GRAVITY: return es[6 + j];
        case IMX_O_ROOT_POS_W: return es[9 + j] - S.env_origins[e * 3 + j];
        case IMX_O_ROOT_QUAT_W: return ((flags & IMX_F_QUAT_UNIQUE) && es[12] < 0.0f) ? -es[12 + j] : es[12 + j];
        case IMX_O_ROOT_LIN_VEL_W: return S.root_lin_vel_w[e * 3 + j];
        case IMX_O_ROOT_ANG_VEL_W: return S.root_ang_vel_w[e * 3 + j];
        case IMX_O_JOINT_POS: return S.joint_pos[e * J + aux];
        case IMX_O_JOINT_POS_REL: return S.joint_pos[e * J + aux] - S.default_joint_pos[e * J + aux];
        case IMX_O_JOINT_POS_LIMIT_NORMALIZED: {  // scale_transform (utils/math.py:22-40)
            const float2 lim = reinterpret_cast<const float2*>(S.soft_joint_pos_limits)[e * J + aux];
            const float offset = (lim.x + lim.y) * 0.5f;
            return 2.0f * (S.joint_pos[e * J + aux] - offset) / (lim.y - lim.x);
        }
        case IMX_O_JOINT_VEL: return S.joint_vel[e * J + aux];
        case IMX_O_JOINT_VEL_REL: return S.joint_vel[e * J + aux] - S.default_joint_vel[e * J + aux];
        case IMX_O_LAST_ACTION: return Bf.action[e * P.A + j];
        case IMX_O_GENERATED_COMMANDS: return S.command[e * P.CMD + j];
        case IMX_O_EXTERNAL: return S.ext_obs[e * (int64_t)W[IMX_H_NEXT_OBS] + aux + j];
        default: return 0.0f;
    }
}

// modifiers -> noise -> clip -> scale -> history window -> obs[e][c] for computed column i (observation_manager.py:305-335)
// LEAN: the plan has one observation group, no modifiers and no history windows (every task config of BASELINE.json) -- the code for
// those features is compiled out, which is worth registers (= resident waves) to every launch that does not need them.
template <bool LEAN>
IMX_DEV void obs_finish(const PlanView& P, const imx_buffers_t& Bf, const XCol& x, int i, float v, int64_t e, int corrupt,
                        bool fill_all, const float* __restrict__ noise_u, uint64_t seed, uint32_t step) {
    const int c = x.a.x;
    if (LEAN) {
        Bf.obs[e * P.gD[0] + c] = obs_post<false>(x, v, corrupt & P.gcorrupt, noise_u, seed, step, e, P.D, 0);
        return;
    }
    const int g = (x.a.w >> 8) & 3;  // observation group of this column (ObservationManager.compute loops over the groups)
    const int gD = g == 0 ? P.gD[0] : (g == 1 ? P.gD[1] : (g == 2 ? P.gD[2] : P.gD[3]));
    const int gb = g == 0 ? 0 : (g == 1 ? P.gbase[1] : (g == 2 ? P.gbase[2] : P.gbase[3]));
    float* grow = g == 0 ? Bf.obs : (g == 1 ? Bf.obs_extra1 : (g == 2 ? Bf.obs_extra2 : Bf.obs_extra3));
    if (x.a.w & IMX_F_MODIFIERS)
        v = apply_modifiers(P.w, P.xmod_off + 4 * i, v, Bf.mod_state + e * P.MS, fill_all || Bf.reset_buf[e]);
    const float vp = obs_post<true>(x, v, corrupt & (P.gcorrupt >> g), noise_u, seed, step, e, P.D, gb);
    float* o = grow + e * gD + c;  // newest slot
    const int hist = x.d.z;
    if (hist > 1) {
        // CircularBuffer.append (utils/buffers/circular_buffer.py:107-135) on the window kept in the obs row itself: this
        // lane owns element j of every slot.  Envs reset this step (or all, at env.reset) have zero pushes: every slot
        // takes the first value; otherwise the window slides by one.
        const int hs = x.d.w;
        if (fill_all || Bf.reset_buf[e]) {
            for (int h = 1; h < hist; ++h) o[-h * hs] = vp;
        } else {
            for (int h = hist - 1; h >= 1; --h) o[-h * hs] = o[-(h - 1) * hs];
        }
    }
    *o = vp;
}


// One ray of the scanner: RayCaster._update_buffers_impl (ray_caster.py:242-260) -> hit height (+inf on a miss, ops.py:70)
template <bool GENERAL_RAYS>
IMX_DEV float scan_ray(const PlanView& P, const MeshView& M, const float* __restrict__ es, const float* __restrict__ ray_local, int j,
                       float px, float py, float pz, float yw, float yz, float* __restrict__ ray_hits_out, int64_t e) {
    const float3 l3 = reinterpret_cast<const float3*>(ray_local)[j];  // one 12-byte load
    const float lx = l3.x, ly = l3.y, lz = l3.z;
    float sx, sy, sz, dx = P.rdx, dy = P.rdy, dz = P.rdz;
    if (!GENERAL_RAYS || P.ray_yaw_only) {  // (the vertical-ray variant is only launched for a yaw-aligned sensor)
        quat_apply_yaw_only(yw, yz, lx, ly, lz, sx, sy, sz);
    } else {  // ray_caster.py:249-252: full orientation for starts and directions
        quat_apply(es[12], es[13], es[14], es[15], lx, ly, lz, sx, sy, sz);
        quat_apply(es[12], es[13], es[14], es[15], P.rdx, P.rdy, P.rdz, dx, dy, dz);
    }
    sx += px; sy += py; sz += pz;
    float t;
    int32_t face;
    const bool hit = GENERAL_RAYS ? cast_ray(M, sx, sy, sz, dx, dy, dz, P.ray_max_dist, t, face)
                                  : cast_ray_vertical(M, sx, sy, sz, dz, P.rinv_dz, P.ray_max_dist, t, face);
    const float hz = hit ? sz + t * dz : __builtin_huge_valf();  // kernels.py:69
    if (ray_hits_out) {
        float* o = ray_hits_out + ((size_t)e * P.R + j) * 3;
        o[0] = hit ? sx + t * dx : __builtin_huge_valf();
        o[1] = hit ? sy + t * dy : __builtin_huge_valf();
        o[2] = hz;
    }
    return hz;
}

// The observation kernel of a LEAN plan with a height scanner (one observation group, no modifier programs, no history windows: every
// rough-terrain task config of BASELINE.json).  Workgroup = ONE WAVE: wave `role` of env e casts rays 64*role .. 64*role+63 and finishes
// THEIR height_scan columns on the spot -- the term's offset / noise / clip / scale are the same for all R columns, so they travel in
// scalar registers (one record, scalar loads) instead of a 16-word column record per lane; the env's last wave fills its other columns
// (base velocity, joints, actions ...).  No LDS, no barrier: the sensor's per-step decision (cast? where? keep the hits?) sits in the
// frame row, put there by the kernel that produced the row.  Single-wave workgroups because what bounds this kernel is how fast the
// chip re-fills wave slots after the first residency round (tools/trace_kobs.py): a one-wave workgroup fits any free slot, a four-wave
// one needs four on one CU.
template <bool GENERAL_RAYS>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_num_sgpr(IMX_LEAN_SGPRS)))
k_obs_lean(PlanView P, int64_t N, imx_state_t S, imx_buffers_t Bf, MeshView M, const float* __restrict__ frame,
           const float* __restrict__ noise_u, uint64_t seed, int corrupt, float* __restrict__ ray_hits_out, StepScratch sc, int tail_G,
           int tail_parts, int scan_rec, int waves_per_env, uint32_t div_magic, int div_shift, const int32_t* __restrict__ W) {
    // (W = P.w once more, as a kernel argument of its own: `noalias` there is what lets the compiler read the term record through the
    // scalar cache instead of four vector loads + v_readfirstlane per wave)
    if ((int)blockIdx.x < tail_parts) {  // extra workgroups, first in the grid: the step tail k_term_rew deferred (kernel boundary = its partials are complete)
        step_tail(P, N, Bf, sc, tail_G, (int)blockIdx.x, tail_parts);
        return;
    }
    const bool role_major = waves_per_env < 0;
    if (role_major) waves_per_env = -waves_per_env;
    const uint32_t step = (uint32_t)Bf.counters[2];
    const bool fill_all = (corrupt & 2) != 0;
    const bool keep_all_hits = (corrupt & 8) != 0;
    corrupt &= 1;
    const int D = P.gD[0];
    // One wave per work item (env, role).  (Persistent waves -- as many as the chip holds, each walking the items with the grid's
    // stride -- were measured: 265 us against 204 at 65 536 envs, 21.5 against 19.7 at 4096; more resident waves only made each slower.)
    const unsigned b = blockIdx.x - (unsigned)tail_parts;
    int64_t e;
    int role;
    // b / divisor by multiply-high with the host's magic number (two scalar instructions instead of the ~25 of a 32-bit division)
    const uint32_t q = (uint32_t)(((uint64_t)b * div_magic) >> 32) >> div_shift;
    if (!role_major) {
        e = q;
        role = (int)(b - q * (unsigned)waves_per_env);
    } else {  // role-major order (the host's choice for small grids)
        role = (int)q;
        e = b - q * (unsigned)N;
    }
    const float* __restrict__ es = frame + e * IMX_ES_WORDS;  // wave-uniform address
    if (role < waves_per_env - 1) {
        // the ray's local start depends on the lane only: requested before the first answer of the frame row is looked at (one round
        // trip of the wave's life less: frame row and ray table travel together)
        const int j0 = role * 64 + (int)threadIdx.x;
        const float3 l3_early = reinterpret_cast<const float3*>(W + P.ray_off)[j0 < P.R ? j0 : P.R - 1];
        // height_scan (observations.py:165-173): sensor.data.pos_w z - hit z - offset, then noise -> clip -> scale (observation_manager.py:313-318)
        const int sflags = (int)es[20];
        const bool cast = (sflags & 1) != 0, cache_z = P.scan_stateful && ((sflags & 2) || keep_all_hits);
        const float px = es[21], py = es[22], pz = es[23];
        const int32_t* r = W + P.obs_off + scan_rec * IMX_REC_WORDS;
        const int out = r[IMX_R_OUT], flags = r[IMX_R_FLAGS];
        const float off = f_of(r[IMX_R_P0]), nlo = f_of(r[IMX_R_NOISE_LO]), nhi = f_of(r[IMX_R_NOISE_HI]);
        const float clo = f_of(r[IMX_R_CLIP_LO]), chi = f_of(r[IMX_R_CLIP_HI]), scale = f_of(r[IMX_R_SCALE]);
        const bool noisy = (corrupt & P.gcorrupt) && (flags & (IMX_F_NOISE_ADD | IMX_F_NOISE_SCALE | IMX_F_NOISE_ABS));
        const float* __restrict__ ray_local = reinterpret_cast<const float*>(W + P.ray_off);
        const float yw = es[16], yz = es[17];
        const int j = role * 64 + (int)threadIdx.x;  // one ray per lane
        const bool has = j < P.R;  // (no early exit: the wave casts together, see cast_ray_vertical_wave)
        // per-env bases are wave-uniform (scalar); the lane adds a 32-bit column offset
        float* __restrict__ obs_row = Bf.obs + e * D;
        float* __restrict__ hitz_row = P.scan_stateful ? Bf.scan_hit_z + (size_t)e * P.R : nullptr;
        float hz = 0.0f;
        if (cast && !GENERAL_RAYS) {
            // RayCaster._update_buffers_impl (ray_caster.py:242-260), yaw-aligned sensor, vertical rays: start = yaw(q) * local + sensor pos
            const float3 l3 = l3_early;
            float sx, sy, sz, t = 0.0f;
            quat_apply_yaw_only(yw, yz, l3.x, l3.y, l3.z, sx, sy, sz);
            sx += px; sy += py; sz += pz;
            const bool hit = cast_ray_vertical_wave(M, has, sx, sy, sz, P.rdz, P.rinv_dz, P.ray_max_dist, t);
            hz = hit ? sz + t * P.rdz : __builtin_huge_valf();  // kernels.py:69
            if (ray_hits_out && has) {
                float* o = ray_hits_out + ((size_t)e * P.R + j) * 3;
                o[0] = hit ? sx + t * P.rdx : __builtin_huge_valf();
                o[1] = hit ? sy + t * P.rdy : __builtin_huge_valf();
                o[2] = hz;
            }
        } else if (cast && has) {
            hz = scan_ray<GENERAL_RAYS>(P, M, es, ray_local, j, px, py, pz, yw, yz, ray_hits_out, e);
        }
        if (!has) return;
        if (cast) {
            if (cache_z) hitz_row[(unsigned)j] = hz;
        } else {
            hz = hitz_row[(unsigned)j];  // data.ray_hits_w of the last update
        }
        float v = pz - hz - off;
        const int c = out + j;
        if (noisy) {
            const float nz = noise_sample<false>(flags, nlo, nhi, noise_u, seed, step, e, P.D, c);
            v = (flags & IMX_F_NOISE_ADD) ? v + nz : ((flags & IMX_F_NOISE_SCALE) ? v * nz : nz);
        }
        if (flags & IMX_F_CLIP) v = fminf(fmaxf(v, clo), chi);
        if (flags & IMX_F_SCALE) v = v * scale;
        obs_row[(unsigned)c] = v;
    } else {
        if (e == 0 && threadIdx.x == 0) Bf.counters[3] = (int32_t)step;  // the shadow the next step kernel counts on from
        for (int i = P.n_ray_cols + (int)threadIdx.x; i < P.DC; i += 64) {  // xcol lists the ray columns first
            const XCol x = load_xcol(W, P.xcol_off, i);
            obs_finish<true>(P, Bf, x, i, obs_plain_value(P, S, Bf, es, e, x), e, corrupt, fill_all, noise_u, seed, step);
        }
    }
}

template <bool GENERAL_RAYS, bool LEAN>
__global__ void __launch_bounds__(256)
k_obs(PlanView P, int64_t N, imx_state_t S, imx_buffers_t Bf, MeshView M, const float* __restrict__ frame,
      const float* __restrict__ noise_u, uint64_t seed, int corrupt, float* __restrict__ ray_hits_out, StepScratch sc, int tail_G,
      int tail_parts) {
    // One workgroup per env.  (A resident grid walking the envs -- sized to what the chip holds at once -- was measured and dropped:
    // the loop costs 13 more VGPRs, i.e. one wave per SIMD less, and three uneven rounds: 35 us against 29.)
    if ((int)blockIdx.x < tail_parts) {  // extra workgroups, first in the grid: the step tail k_term_rew deferred (kernel boundary = its partials are complete)
        step_tail(P, N, Bf, sc, tail_G, (int)blockIdx.x, tail_parts);
        return;
    }
    const int64_t e = blockIdx.x - (unsigned)tail_parts;
    const int32_t* __restrict__ W = P.w;
    const uint32_t step = (uint32_t)Bf.counters[2];
    const bool fill_all = (corrupt & 2) != 0;
    const bool keep_all_hits = (corrupt & 8) != 0;
    corrupt &= 1;
    extern __shared__ float s_hz[];  // R hit heights of this env
    const float* __restrict__ es = frame + e * IMX_ES_WORDS;  // wave-uniform address
    const float yw = es[16], yz = es[17];
    // the sensor's decision for this step (scanner_step, made where the frame row was produced): cast?  keep the hits?  from where?
    const int sflags = (int)es[20];
    const bool cast = (sflags & 1) != 0, cache_z = P.scan_stateful && ((sflags & 2) || keep_all_hits);
    const float px = es[21], py = es[22], pz = es[23];
#ifdef IMX_TRACE
    const uint64_t trace_tp = 0;
    const uint64_t trace_t0 = wall_clock64();
    uint64_t trace_t1 = 0, trace_t2 = 0;
#endif
    // -- phase A: the env's rays.  RayCaster._update_buffers_impl (ray_caster.py:242-260): ray j of the scanner -> hit height into LDS.
    //    Kept apart from the column loop below on purpose: what limits this kernel is how many waves a SIMD can hold (wave life x
    //    residency, tools/trace_kobs.py), i.e. its register count -- a ray lane carries the ray and one cell's data, a column lane its
    //    16-word column record, never both.
    if (P.R > 0) {
        const float* __restrict__ ray_local = reinterpret_cast<const float*>(W + P.ray_off);
        for (int j = threadIdx.x; j < P.R; j += blockDim.x) {
            float hz;
            if (cast) {
                hz = scan_ray<GENERAL_RAYS>(P, M, es, ray_local, j, px, py, pz, yw, yz, ray_hits_out, e);
                if (cache_z) Bf.scan_hit_z[(size_t)e * P.R + j] = hz;
            } else {
                hz = Bf.scan_hit_z[(size_t)e * P.R + j];  // data.ray_hits_w of the last update
            }
            s_hz[j] = hz;
        }
#ifdef IMX_TRACE
        trace_t1 = wall_clock64();
#endif
        __syncthreads();
#ifdef IMX_TRACE
        trace_t2 = wall_clock64();
#endif
    }
    if (e == 0 && threadIdx.x == 0) Bf.counters[3] = (int32_t)step;  // the shadow the next step kernel counts on from
    // -- phase B: the observation columns (ObservationManager.compute_group, observation_manager.py:260-335)
    for (int i = threadIdx.x; i < P.DC; i += blockDim.x) {
        const XCol x = load_xcol(W, P.xcol_off, i);
        const int op = x.a.y, j = x.a.z;
        float v;
        if (op == IMX_O_HEIGHT_SCAN) {  // height_scan (observations.py:165-173): sensor.data.pos_w z - hit z - offset
            const float hz = s_hz[j];
            v = pz - hz - f_of(x.b.x);
            // further height_scan terms on the same sensor (another group, another offset / noise / clip): same hit, own post-processing
            for (int nx = LEAN ? 0 : x.c.z; nx != 0;) {
                const XCol tw = load_xcol(W, P.xcol_off, nx - 1);
                obs_finish<false>(P, Bf, tw, nx - 1, pz - hz - f_of(tw.b.x), e, corrupt, fill_all, noise_u, seed, step);
                nx = tw.c.z;
            }
        } else {
            v = obs_plain_value(P, S, Bf, es, e, x);
        }
        obs_finish<LEAN>(P, Bf, x, i, v, e, corrupt, fill_all, noise_u, seed, step);
    }
#ifdef IMX_TRACE
    if (g_trace && (threadIdx.x & 63) == 0) {
        uint64_t* t = g_trace + ((size_t)e * 4 + (threadIdx.x >> 6)) * 8;
        t[0] = trace_t0; t[1] = wall_clock64();
        t[2] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_ID (wave/simd/cu/sh/se ids)
        t[3] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));   // XCC_ID
        t[4] = trace_t1; t[5] = trace_t2; t[6] = trace_tp;
    }
#endif
}

// ------------------------------------------------------------------------------------------------- root frame
__global__ void k_root_frame(int64_t N, const float* __restrict__ q, const float* __restrict__ lv,
                             const float* __restrict__ av, float gx, float gy, float gz, float* __restrict__ olv,
                             float* __restrict__ oav, float* __restrict__ opg) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N) return;
    const float w = q[e * 4], x = q[e * 4 + 1], y = q[e * 4 + 2], z = q[e * 4 + 3];
    float a, b, c;
    if (olv) { quat_rotate_inverse(w, x, y, z, lv[e * 3], lv[e * 3 + 1], lv[e * 3 + 2], a, b, c); olv[e * 3] = a; olv[e * 3 + 1] = b; olv[e * 3 + 2] = c; }
    if (oav) { quat_rotate_inverse(w, x, y, z, av[e * 3], av[e * 3 + 1], av[e * 3 + 2], a, b, c); oav[e * 3] = a; oav[e * 3 + 1] = b; oav[e * 3 + 2] = c; }
    if (opg) { quat_rotate_inverse(w, x, y, z, gx, gy, gz, a, b, c); opg[e * 3] = a; opg[e * 3 + 1] = b; opg[e * 3 + 2] = c; }
}

// ------------------------------------------------------------------------------------------------- raycast_mesh
__global__ void k_raycast(MeshView M, const float* __restrict__ starts, const float* __restrict__ dirs, int64_t n,
                          float max_dist, float* __restrict__ hits, float* __restrict__ dist, int32_t* __restrict__ faces) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float ox = starts[i * 3], oy = starts[i * 3 + 1], oz = starts[i * 3 + 2];
    const float dx = dirs[i * 3], dy = dirs[i * 3 + 1], dz = dirs[i * 3 + 2];
    float t;
    int32_t f;
    const float inf = __builtin_huge_valf();
    if (cast_ray(M, ox, oy, oz, dx, dy, dz, max_dist, t, f)) {
        hits[i * 3] = ox + t * dx; hits[i * 3 + 1] = oy + t * dy; hits[i * 3 + 2] = oz + t * dz;
        if (dist) dist[i] = t;
        if (faces) faces[i] = f;
    } else {
        hits[i * 3] = inf; hits[i * 3 + 1] = inf; hits[i * 3 + 2] = inf;
        if (dist) dist[i] = inf;
        if (faces) faces[i] = -1;
    }
}

// ------------------------------------------------------------------------------------------------- C ABI
// Magic number for floor(b / d), 0 <= b < 2^31, d >= 2: q = mulhi(b, magic) >> shift.  With l = ceil(log2 d) and magic =
// floor(2^(31+l) / d) + 1 (< 2^32 because d > 2^(l-1)), b * magic / 2^(31+l) = b/d + eps with eps < 2^31 * d / (d * 2^(31+l)) <= 1/d:
// never enough to carry frac(b/d) <= (d-1)/d over the next integer.
static void imx_magic_u31(uint32_t d, uint32_t& magic, int& shift) {
    int l = 1;
    while ((1ull << l) < d) ++l;
    magic = (uint32_t)(((1ull << (31 + l)) / d) + 1);
    shift = l - 1;
}

static int check_common(const imx_plan_t* plan, int64_t N, const imx_state_t* st, const imx_buffers_t* bf) {
    IMX_REQUIRE(plan && st && bf, "null plan/state/buffers");
    IMX_REQUIRE(plan->dev, "plan has no device copy (no GPU visible when it was created)");
    IMX_REQUIRE(N > 0 && N < (1ll << 31), "num_envs out of range: %lld", (long long)N);
    return 0;
}

int imx_check_action_inputs(const imx_plan_t* plan, const imx_state_t* st, const imx_buffers_t* bf, const char* who) {
    IMX_REQUIRE(bf->action && bf->prev_action && bf->processed_action, "%s: null action buffer", who);
    IMX_REQUIRE(plan->A > 0, "%s: plan has no action columns", who);
    for (int k = 0; k < plan->nact; ++k) {
        const int flags = plan->host[plan->act_off + k * IMX_REC_WORDS + IMX_R_FLAGS];
        if (flags & IMX_F_ACT_DEFAULT_POS_OFFSET) IMX_REQUIRE(st->default_joint_pos, "%s: default_joint_pos missing", who);
        if (flags & IMX_F_ACT_DEFAULT_VEL_OFFSET) IMX_REQUIRE(st->default_joint_vel, "%s: default_joint_vel missing", who);
        if (flags & (IMX_F_ACT_TO_LIMITS | IMX_F_ACT_EMA)) IMX_REQUIRE(st->soft_joint_pos_limits, "%s: soft_joint_pos_limits missing", who);
        if (flags & IMX_F_ACT_EMA) IMX_REQUIRE(st->joint_pos, "%s: joint_pos missing", who);
    }
    return 0;
}

extern "C" int imx_action_process(const imx_plan_t* plan, int64_t N, const float* actions_d, float pre_clip,
                                  const imx_state_t* st, const imx_buffers_t* bf, imx_stream_t stream) {
    if (check_common(plan, N, st, bf)) return 1;
    IMX_REQUIRE(actions_d, "imx_action_process: null actions");
    if (imx_check_action_inputs(plan, st, bf, "imx_action_process")) return 1;
    const int64_t n = N * plan->A;
    const int bs = 256;
    hipLaunchKernelGGL(k_action, dim3((unsigned)((n + bs - 1) / bs)), dim3(bs), 0, (hipStream_t)stream,
                       imx_plan_view(plan), N, actions_d, pre_clip, *st, *bf);
    IMX_HIP(hipGetLastError());
    return 0;
}

extern "C" int imx_terminations_rewards(const imx_plan_t* plan, int64_t N, const imx_state_t* st,
                                        const imx_buffers_t* bf, int flags, imx_stream_t stream) {
    return imx_terminations_rewards_rollout(plan, N, st, bf, flags, nullptr, stream);
}

extern "C" int imx_terminations_rewards_rollout(const imx_plan_t* plan, int64_t N, const imx_state_t* st, const imx_buffers_t* bf,
                                                int flags, const imx_rollout_slot_t* slot, imx_stream_t stream) {
    if (check_common(plan, N, st, bf)) return 1;
    imx_rollout_slot_t ro{};
    if (slot) {
        ro = *slot;
        IMX_REQUIRE(ro.value_t && ro.rewards_out && ro.dones_out, "imx_terminations_rewards_rollout: value_t, rewards_out and dones_out are required");
        IMX_REQUIRE((ro.cur_reward_sum == nullptr) == (ro.cur_ep_len == nullptr), "imx_terminations_rewards_rollout: episode buffers come together");
        IMX_REQUIRE(!ro.ep_stats3 || ro.cur_reward_sum, "imx_terminations_rewards_rollout: ep_stats3 needs the episode buffers");
    }
    IMX_REQUIRE(st->root_quat_w && st->root_lin_vel_w && st->root_ang_vel_w, "root state missing");
    IMX_REQUIRE(bf->episode_length_buf && bf->reward_buf && bf->episode_sums && bf->step_reward && bf->term_dones &&
                    bf->terminated && bf->truncated && bf->reset_buf && bf->reset_env_ids && bf->counters &&
                    bf->log_out && bf->scratch && bf->action && bf->prev_action,
                "imx_terminations_rewards: null buffer");
    // inputs each op dereferences must be present
    const auto& w = plan->host;
    auto need = [&](const void* p, const char* name) -> int {
        IMX_REQUIRE(p, "state tensor '%s' is required by a term but missing", name);
        return 0;
    };
    for (int k = 0; k < plan->nterm; ++k) {
        switch (w[plan->term_off + k * IMX_REC_WORDS + IMX_R_OP]) {
            case IMX_T_ILLEGAL_CONTACT: if (need(st->net_forces_w_history, "net_forces_w_history")) return 1; break;
            case IMX_T_JOINT_POS_MANUAL_LIMIT: if (need(st->joint_pos, "joint_pos")) return 1; break;
            case IMX_T_ROOT_HEIGHT_BELOW_MIN: case IMX_T_TERRAIN_OUT_OF_BOUNDS: if (need(st->root_pos_w, "root_pos_w")) return 1; break;
            case IMX_T_JOINT_VEL_LIMIT: if (need(st->joint_vel, "joint_vel") || need(st->soft_joint_vel_limits, "soft_joint_vel_limits")) return 1; break;
            case IMX_T_JOINT_VEL_MANUAL_LIMIT: if (need(st->joint_vel, "joint_vel")) return 1; break;
            case IMX_T_JOINT_EFFORT_LIMIT: if (need(st->computed_torque, "computed_torque") || need(st->applied_torque, "applied_torque")) return 1; break;
            case IMX_T_EXTERNAL: if (need(st->ext_term, "ext_term")) return 1; break;
            case IMX_T_COMMAND_RESAMPLE: if (need(st->command_time_left, "command_time_left") || need(st->command_counter, "command_counter")) return 1; break;
            default: break;
        }
    }
    for (int k = 0; k < plan->nrew; ++k) {
        // (zero-weight terms too: imx_plan_update / set_term_cfg can wake one in place while a captured graph keeps replaying this
        //  launch with the state pointers it was captured with -- a NULL tensor would then be a device fault, not an error return)
        switch (w[plan->rew_off + k * IMX_REC_WORDS + IMX_R_OP]) {
            case IMX_W_BASE_HEIGHT_L2: if (need(st->root_pos_w, "root_pos_w")) return 1; break;
            case IMX_W_JOINT_TORQUES_L2: if (need(st->applied_torque, "applied_torque")) return 1; break;
            case IMX_W_JOINT_VEL_L1: case IMX_W_JOINT_VEL_L2: if (need(st->joint_vel, "joint_vel")) return 1; break;
            case IMX_W_JOINT_ACC_L2: if (need(st->joint_acc, "joint_acc")) return 1; break;
            case IMX_W_JOINT_DEVIATION_L1: if (need(st->joint_pos, "joint_pos") || need(st->default_joint_pos, "default_joint_pos")) return 1; break;
            case IMX_W_JOINT_POS_LIMITS: if (need(st->joint_pos, "joint_pos") || need(st->soft_joint_pos_limits, "soft_joint_pos_limits")) return 1; break;
            case IMX_W_JOINT_VEL_LIMITS: if (need(st->joint_vel, "joint_vel") || need(st->soft_joint_vel_limits, "soft_joint_vel_limits")) return 1; break;
            case IMX_W_APPLIED_TORQUE_LIMITS: if (need(st->applied_torque, "applied_torque") || need(st->computed_torque, "computed_torque")) return 1; break;
            case IMX_W_UNDESIRED_CONTACTS: case IMX_W_CONTACT_FORCES: if (need(st->net_forces_w_history, "net_forces_w_history")) return 1; break;
            case IMX_W_TRACK_LIN_VEL_XY_EXP: case IMX_W_TRACK_ANG_VEL_Z_EXP: case IMX_W_TRACK_LIN_VEL_XY_YAW_FRAME_EXP:
            case IMX_W_TRACK_ANG_VEL_Z_WORLD_EXP: if (need(st->command, "command")) return 1; break;
            case IMX_W_FEET_AIR_TIME: if (need(st->command, "command") || need(st->current_contact_time, "current_contact_time") || need(st->last_air_time, "last_air_time")) return 1; break;
            case IMX_W_FEET_AIR_TIME_POSITIVE_BIPED: if (need(st->command, "command") || need(st->current_contact_time, "current_contact_time") || need(st->current_air_time, "current_air_time")) return 1; break;
            case IMX_W_FEET_SLIDE: if (need(st->net_forces_w_history, "net_forces_w_history") || need(st->body_lin_vel_w, "body_lin_vel_w")) return 1; break;
            case IMX_W_JOINT_POS_TARGET_L2: if (need(st->joint_pos, "joint_pos")) return 1; break;
            case IMX_W_EXTERNAL: if (need(st->ext_reward, "ext_reward")) return 1; break;
            case IMX_W_BODY_LIN_ACC_L2: if (need(st->body_lin_acc_w, "body_lin_acc_w")) return 1; break;
            default: break;
        }
    }
    if (plan->CMD > 0 && !st->command) IMX_FAIL("command tensor missing");
    IMX_REQUIRE(!plan->scan_stateful || !st->root_pos_w || bf->scan_state,
                "imx_terminations_rewards: the height scanner has an update period / drift range: scan_state (N,8) is required");
    const int G = step_group_size(N);
    const unsigned grid = (unsigned)((N + G - 1) / G);
    StepScratch sc = carve(bf->scratch, N, plan->nrew_all > 0 ? plan->nrew_all : 1, plan->nterm > 0 ? plan->nterm : 1);
    // one wave per work item (termination or reward term) up to 16 waves; more items go round-robin
    const int items = plan->nterm + plan->nrew;
    const int NW = items < 2 ? 2 : (items > IMX_TR_MAX_WAVES ? IMX_TR_MAX_WAVES : items);
    const size_t lds = ((size_t)(plan->nterm > 0 ? plan->nterm : 1) * 64 + 3 * (size_t)(plan->nrew > 0 ? plan->nrew : 1) * 64) * 4;
    // with the root position at hand the kernel also leaves the frame table imx_observations needs (flag 4 there skips k_frame)
    float* frame = st->root_pos_w ? reinterpret_cast<float*>(reinterpret_cast<char*>(bf->scratch) + frame_offset_bytes(plan, N)) : nullptr;
    hipLaunchKernelGGL(k_term_rew, dim3(grid), dim3(64 * NW), lds, (hipStream_t)stream, imx_plan_view(plan), N, *st, *bf, sc, frame, G,
                       flags & 1, ro);
    IMX_HIP(hipGetLastError());
    return 0;
}

// Which observation kernel a plan gets (imx_observations below; imx_observations_kernel_name reports it to benchmarks).
struct ObsKernelChoice {
    bool lean;         // one group, no modifier programs, no history windows, no gaussian noise (DC == D also rules out twin scan columns)
    bool vertical;     // height-scanner frame yaw-only + vertical direction (the reference cfg): register-lean single-cell ray path
    bool single_wave;  // k_obs_lean: one single-wave workgroup per 64 rays + one for the env's other columns
    int scan_rec;
};
static ObsKernelChoice choose_obs_kernel(const imx_plan_t* plan) {
    const auto& w = plan->host;
    const PlanView pv = imx_plan_view(plan);
    ObsKernelChoice c;
    c.lean = plan->ngroups == 1 && plan->MS == 0 && plan->DC == plan->D && plan->DX == plan->DC;
    for (int k = 0; k < plan->nobs && c.lean; ++k) c.lean = !(w[plan->obs_off + k * IMX_REC_WORDS + IMX_R_FLAGS] & (IMX_F_MODIFIERS | IMX_F_NOISE_GAUSS));
    c.vertical = pv.R == 0 || (pv.ray_yaw_only && pv.rdx == 0.0f && pv.rdy == 0.0f && pv.rdz != 0.0f);
    c.scan_rec = -1;
    for (int k = 0; k < plan->nobs; ++k)
        if (w[plan->obs_off + k * IMX_REC_WORDS + IMX_R_OP] == IMX_O_HEIGHT_SCAN) c.scan_rec = k;
    c.single_wave = c.lean && c.scan_rec >= 0 && pv.R > 0 && pv.R <= 64 * 15;
    return c;
}

extern "C" const char* imx_observations_kernel_name(const imx_plan_t* plan) {
    if (!plan) return "";
    const ObsKernelChoice c = choose_obs_kernel(plan);
    if (c.single_wave) return c.vertical ? "k_obs_lean<false>" : "k_obs_lean<true>";
    if (c.vertical) return c.lean ? "k_obs<false,true>" : "k_obs<false,false>";
    return c.lean ? "k_obs<true,true>" : "k_obs<true,false>";
}

extern "C" int imx_observations(const imx_plan_t* plan, int64_t N, const imx_state_t* st, const imx_buffers_t* bf,
                                const imx_mesh_t* mesh, const float* noise_u_d, uint64_t seed, int enable_corruption,
                                float* ray_hits_out_d, imx_stream_t stream) {
    if (check_common(plan, N, st, bf)) return 1;
    IMX_REQUIRE(bf->obs && bf->counters, "imx_observations: null obs/counters buffer");
    IMX_REQUIRE(st->root_quat_w && st->root_lin_vel_w && st->root_ang_vel_w && st->root_pos_w, "root state missing");
    IMX_REQUIRE(!plan->needs_mesh || mesh, "plan has a height_scan term but no mesh was given");
    IMX_REQUIRE(plan->DC == plan->D || bf->reset_buf, "imx_observations: observation history needs the reset mask (reset_buf)");
    {
        const float* extra[3] = {bf->obs_extra1, bf->obs_extra2, bf->obs_extra3};
        for (int g = 1; g < plan->ngroups; ++g)
            IMX_REQUIRE(extra[g - 1], "imx_observations: the plan has %d observation groups but obs_extra%d is NULL", plan->ngroups, g);
    }
    IMX_REQUIRE(!plan->scan_stateful || (bf->scan_state && bf->scan_hit_z && bf->reset_buf),
                "imx_observations: the height scanner has an update period / drift range: scan_state (N,8), scan_hit_z (N,R) and reset_buf are required");
    IMX_REQUIRE(plan->MS == 0 || (bf->mod_state && bf->reset_buf),
                "imx_observations: the plan has stateful observation modifiers: mod_state (N x %d floats) and reset_buf are required", plan->MS);
    const auto& w = plan->host;
    for (int k = 0; k < plan->nobs; ++k) {
        const int op = w[plan->obs_off + k * IMX_REC_WORDS + IMX_R_OP];
        const void* p = (const void*)1;
        const char* name = "";
        switch (op) {
            case IMX_O_ROOT_POS_W: p = st->env_origins; name = "env_origins"; break;
            case IMX_O_JOINT_POS: p = st->joint_pos; name = "joint_pos"; break;
            case IMX_O_JOINT_POS_REL: p = (st->joint_pos && st->default_joint_pos) ? (const void*)1 : nullptr; name = "joint_pos/default_joint_pos"; break;
            case IMX_O_JOINT_POS_LIMIT_NORMALIZED: p = (st->joint_pos && st->soft_joint_pos_limits) ? (const void*)1 : nullptr; name = "joint_pos/soft_joint_pos_limits"; break;
            case IMX_O_JOINT_VEL: p = st->joint_vel; name = "joint_vel"; break;
            case IMX_O_JOINT_VEL_REL: p = (st->joint_vel && st->default_joint_vel) ? (const void*)1 : nullptr; name = "joint_vel/default_joint_vel"; break;
            case IMX_O_LAST_ACTION: p = bf->action; name = "action"; break;
            case IMX_O_GENERATED_COMMANDS: p = st->command; name = "command"; break;
            case IMX_O_EXTERNAL: p = st->ext_obs; name = "ext_obs"; break;
            default: break;
        }
        IMX_REQUIRE(p, "state tensor '%s' is required by an observation term but missing", name);
    }
    MeshView mv{};
    if (mesh) mv = mesh->v;
    // one block per env; lanes take column i, i + blockDim, ...  Ray columns come first and are the long ones: when they
    // fit in three waves the remaining (cheap) columns ride as a second trip of the first lanes instead of a fourth wave
    int bs = plan->DC <= 64 ? 64 : (plan->DC <= 128 ? 128 : (plan->DC <= 192 ? 192 : 256));
    if (plan->DC > 192 && plan->n_ray_cols > 0 && plan->n_ray_cols <= 192 && plan->DC <= 2 * 192) bs = 192;
    const PlanView pv = imx_plan_view(plan);
    IMX_REQUIRE(bf->scratch, "imx_observations: scratch buffer missing");
    float* frame = reinterpret_cast<float*>(reinterpret_cast<char*>(bf->scratch) + frame_offset_bytes(plan, N));
    if (!(enable_corruption & 4))  // bit 2: imx_terminations_rewards ran on this very state and left the frame table behind
        hipLaunchKernelGGL(k_frame, dim3((unsigned)((N + 63) / 64)), dim3(64), 0, (hipStream_t)stream, pv, N, *st, *bf, frame, (enable_corruption >> 1) & 1);
    const ObsKernelChoice ch = choose_obs_kernel(plan);
    const bool vertical = ch.vertical;
    // bit 4: finish the step tail imx_terminations_rewards (flags bit 0) left to this call -- one extra workgroup
    const bool tail = (enable_corruption & 16) != 0;
    StepScratch sc{};
    int tail_G = 0, tail_parts = 0;
    if (tail) {
        IMX_REQUIRE(bf->reset_env_ids && bf->log_out, "imx_observations: finishing the step tail needs reset_env_ids and log_out");
        sc = carve(bf->scratch, N, plan->nrew_all > 0 ? plan->nrew_all : 1, plan->nterm > 0 ? plan->nterm : 1);
        tail_G = step_group_size(N);
    }
    const size_t lds = (size_t)(pv.R > 0 ? pv.R : 1) * 4;
    const int nlog = plan->nrew_all + plan->nterm;
    const bool lean = ch.lean;
#define IMX_LAUNCH_OBS(G, L)                                                                                                     \
    hipLaunchKernelGGL((k_obs<G, L>), dim3(grid), dim3(bs), lds, (hipStream_t)stream, pv, N, *st, *bf, mv, frame, noise_u_d, seed, \
                       enable_corruption, ray_hits_out_d, sc, tail_G, tail_parts)
    const int scan_rec = ch.scan_rec;
    if (ch.single_wave) {
        // one single-wave workgroup per 64 rays + one for the env's other columns
        const int wpe = (pv.R + 63) / 64 + 1;
        IMX_REQUIRE((uint64_t)N * wpe + 1 < (1ull << 31), "imx_observations: %lld envs x %d waves exceed the grid", (long long)N, wpe);
        if (tail) tail_parts = 1 + (nlog < 16 ? nlog : 16);  // one wave orders the reset ids, one per log entry (<= 16)
        // Up to ~2 residency rounds the waves go out role by role (all envs' first 64 rays, ..., the short column waves last: a
        // shorter drain, 17.2 us against 18.6 at 4096 envs); beyond that env by env (an env's rays share mesh lines: 173 us against
        // 191 at 65536 envs).
        const bool role_major = N <= 8192 && N >= 2;
        const uint32_t divisor = role_major ? (uint32_t)N : (uint32_t)wpe;
        uint32_t div_magic = 0;
        int div_shift = 0;
        imx_magic_u31(divisor, div_magic, div_shift);
        const unsigned lgrid = (unsigned)(N * wpe) + (unsigned)tail_parts;
        if (vertical)
            hipLaunchKernelGGL(k_obs_lean<false>, dim3(lgrid), dim3(64), 0, (hipStream_t)stream, pv, N, *st, *bf, mv, frame, noise_u_d, seed,
                               enable_corruption, ray_hits_out_d, sc, tail_G, tail_parts, scan_rec, role_major ? -wpe : wpe, div_magic, div_shift, pv.w);
        else
            hipLaunchKernelGGL(k_obs_lean<true>, dim3(lgrid), dim3(64), 0, (hipStream_t)stream, pv, N, *st, *bf, mv, frame, noise_u_d, seed,
                               enable_corruption, ray_hits_out_d, sc, tail_G, tail_parts, scan_rec, role_major ? -wpe : wpe, div_magic, div_shift, pv.w);
    } else {
        if (tail) tail_parts = 1 + ((nlog < 16 ? nlog : 16) * 64 + bs - 1) / bs;
        const unsigned grid = (unsigned)N + (unsigned)tail_parts;
        if (vertical) { if (lean) IMX_LAUNCH_OBS(false, true); else IMX_LAUNCH_OBS(false, false); }
        else { if (lean) IMX_LAUNCH_OBS(true, true); else IMX_LAUNCH_OBS(true, false); }
    }
#undef IMX_LAUNCH_OBS
    IMX_HIP(hipGetLastError());
    return 0;
}

extern "C" int imx_root_frame(int64_t N, const float* q, const float* lv, const float* av, float gx, float gy, float gz,
                              float* olv, float* oav, float* opg, imx_stream_t stream) {
    IMX_REQUIRE(N > 0 && q, "imx_root_frame: bad arguments");
    IMX_REQUIRE((!olv || lv) && (!oav || av), "imx_root_frame: output requested without its input");
    hipLaunchKernelGGL(k_root_frame, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, N, q, lv, av,
                       gx, gy, gz, olv, oav, opg);
    IMX_HIP(hipGetLastError());
    return 0;
}

extern "C" int imx_raycast(const imx_mesh_t* mesh, const float* starts, const float* dirs, int64_t n, float max_dist,
                           float* hits, float* dist, int32_t* faces, imx_stream_t stream) {
    if (n == 0) return 0;
    IMX_REQUIRE(mesh && starts && dirs && hits, "imx_raycast: null argument");
    IMX_REQUIRE(n > 0, "imx_raycast: negative ray count");
    hipLaunchKernelGGL(k_raycast, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, mesh->v, starts,
                       dirs, n, max_dist, hits, dist, faces);
    IMX_HIP(hipGetLastError());
    return 0;
}
