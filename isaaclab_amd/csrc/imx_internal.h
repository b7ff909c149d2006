// Internal declarations shared by the libimx translation units (gfx950 only; no portability layer).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/imx.h"

#define IMX_WAVE 64

// ---- error plumbing ---------------------------------------------------------------------------------------------
void imx_set_error(const char* fmt, ...);
#define IMX_FAIL(...)               \
    do {                            \
        imx_set_error(__VA_ARGS__); \
        return 1;                   \
    } while (0)
#define IMX_HIP(expr)                                                                      \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess) IMX_FAIL("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)
#define IMX_REQUIRE(cond, ...) \
    do {                       \
        if (!(cond)) IMX_FAIL(__VA_ARGS__); \
    } while (0)

// ---- plan ---------------------------------------------------------------------------------------------------------
struct imx_plan {
    std::vector<int32_t> host;  // validated blob + appended column tables
    int32_t* dev = nullptr;     // device copy (null when no GPU is visible)
    int J = 0, B = 0, H = 0, A = 0, D = 0, R = 0, NB = 0, CMD = 3;
    int nterm = 0, nrew = 0, nobs = 0, nact = 0, nrew_all = 0;
    int term_off = 0, rew_off = 0, obs_off = 0, act_off = 0, ray_off = 0;
    int col_off = 0;    // D words: (obs record index << 16) | local index j
    int order_off = 0;  // D words: column permutation, ray columns first
    int n_ray_cols = 0;
    int skip_off = 0, nskip = 0;  // reward slots with no record (zero weight)
    int xcol_off = 0;             // DC x 16 words: per-column expansion of the obs records, ray columns first
    int DC = 0;                   // computed observation columns (= D unless a term keeps a history window)
    int xmod_off = 0;             // DC x 4 words: [program offset, program words, state offset of this column, term width]
    int MS = 0;                   // floats of modifier state per env
    int DX = 0;                   // xcol entries: DC scheduled columns + twin height-scan columns
    int ngroups = 1;              // observation groups; group g = columns [gbase[g], gbase[g] + gD[g]) of the D-wide column space
    int gD[IMX_MAX_OBS_GROUPS] = {0, 0, 0, 0}, gbase[IMX_MAX_OBS_GROUPS] = {0, 0, 0, 0};
    bool gcorrupt[IMX_MAX_OBS_GROUPS] = {false, false, false, false};
    bool scan_stateful = false;
    bool needs_mesh = false;
};

// view passed by value to kernels
struct PlanView {
    const int32_t* w;  // device blob
    int J, B, H, A, D, R, NB, CMD;
    int nterm, nrew, nobs, nact, nrew_all;
    int term_off, rew_off, obs_off, act_off, ray_off, col_off, order_off, n_ray_cols, skip_off, nskip, xcol_off, DC, xmod_off, MS;
    int max_ep_len;
    float step_dt, max_ep_len_s;
    float gx, gy, gz;
    float rdx, rdy, rdz, ray_max_dist;
    float rinv_dz;  // 1 / rdz, divided on the host (IEEE, the same bits as the in-kernel division it replaces: ~12 VALU per ray)
    int ray_yaw_only;
    int ngroups, gD[IMX_MAX_OBS_GROUPS], gbase[IMX_MAX_OBS_GROUPS], gcorrupt;
    int scan_stateful, scan_substeps;
    float scan_period, scan_dt, drift_lo, drift_hi;
};
PlanView imx_plan_view(const imx_plan* p);

// ---- mesh ---------------------------------------------------------------------------------------------------------
// Cells of the xy grid are stored in 8x8 tiles (tile-major), so the ~17x11-cell footprint of one height scanner maps to a handful
// of contiguous runs.  32 bytes per cell: a = four floats, b = {kind, ...} (mesh.hip pass 2 says what qualifies):
//   kind (low 8 bits of b.x; bits 8..16 = continuity mask of LATTICE / QH cells, mesh.hip pass 2b) 0 EMPTY
//        1 LATTICE  the two triangles (a,b,c),(a,d,b) of one height-field quad (convert_height_field_to_mesh topology) with corners
//                   on the grid's coordinate lines gx[ix], gx[ix+1], gy[iy], gy[iy+1]: a = heights of a (ix,iy), d (ix+1,iy),
//                   c (ix,iy+1), b (ix+1,iy+1); b.y = face id of (a,b,c), the other triangle is b.y + 1
//        2 QH       the highest surface over the cell's interior is horizontal in each of the <= 4 rectangles the lines x = cx,
//                   y = cy cut it into: a = {h00 (x<cx,y<cy), h10, h01, h11}; b = {kind, cx, cy, a face id} (+inf = no line)
//        3 GENERAL  a = {id2, id0, id1, zrest after the first pair}; b = {kind, first reference, count (even), id3}: the first two
//                   pairs of its reference list inline
//   QH and GENERAL cells also have cell_list[c] = {first reference, count}: a reference is {triangle id, zrest}, sorted by descending
//   triangle top (max z), zrest = highest top among the references after it (a downward ray stops once its hit is above zrest);
//   refs holds them as pairs {id0, zrest0, id1, zrest1}.  Triangle records (48 B, ONE per triangle, shared by all the cells that
//   reference it -- a box face covering hundreds of cells stays hot in L2): ax ay az bx | by bz cx cy | cz face(int) ztop 0
#define IMX_CELL_EMPTY 0
#define IMX_CELL_LATTICE 1
#define IMX_CELL_QH 2
#define IMX_CELL_GENERAL 3
struct MeshView {
    const int4* cells;         // (ntx*nty*64 x 2) {a, b} per cell
    const int2* cell_list;     // (ntx*nty*64) {first reference, count} of the QH / GENERAL cells
    const float* gx;           // (nx+1) lattice coordinate lines
    const float* gy;           // (ny+1)
    const int4* refs;          // (references / 2) pairs {id0, zrest0, id1, zrest1}
    const float4* tri_rec;     // (F, 3 x float4)
    int ntx, nty;
    int nx, ny;
    float x0, y0, cell, inv_cell;
    int64_t F;
};
struct imx_mesh {
    MeshView v{};
    int32_t* d_cell = nullptr;
    int32_t* d_cell_list = nullptr;
    float* d_gx = nullptr;
    float* d_gy = nullptr;
    float* d_tri_rec = nullptr;
    int32_t* d_refs = nullptr;
    int64_t num_refs = 0;  // general cell references
    int32_t max_refs = 0;
    int64_t n_lattice = 0, n_general = 0, n_flat = 0;  // n_general counts QH (n_flat) and GENERAL cells
};
// linear index of cell (ix, iy) in the 8x8-tiled layout
static __host__ __device__ __forceinline__ int imx_cell_index(int ix, int iy, int ntx) {
    return (((iy >> 3) * ntx + (ix >> 3)) << 6) | ((iy & 7) << 3) | (ix & 7);
}
#define IMX_GRID_TAU 1.0e-3f  // boundary snap tolerance in cell units (see raycast.hip)

// ---- device helpers ------------------------------------------------------------------------------------------------
#define IMX_DEV static __device__ __forceinline__

IMX_DEV float f_of(int32_t w) { return __int_as_float(w); }

// quat_rotate_inverse (isaaclab/utils/math.py:605-625): a - b + c with
//   a = v*(2 w^2 - 1), b = cross(q_vec, v)*w*2, c = q_vec*dot(q_vec, v)*2   (same association as the reference)
IMX_DEV void quat_rotate_inverse(float w, float x, float y, float z, float vx, float vy, float vz, float& ox, float& oy,
                                 float& oz) {
    const float s = 2.0f * (w * w) - 1.0f;
    const float ax = vx * s, ay = vy * s, az = vz * s;
    const float cx = y * vz - z * vy, cy = z * vx - x * vz, cz = x * vy - y * vx;
    const float bx = cx * w * 2.0f, by = cy * w * 2.0f, bz = cz * w * 2.0f;
    const float d = (x * vx + y * vy) + z * vz;  // bmm: sequential dot
    const float ccx = x * d * 2.0f, ccy = y * d * 2.0f, ccz = z * d * 2.0f;
    ox = ax - bx + ccx;
    oy = ay - by + ccy;
    oz = az - bz + ccz;
}

// yaw_quat (isaaclab/utils/math.py:521-542) -> (qw, qz) of the yaw-only quaternion (x = y = 0)
IMX_DEV void yaw_quat_wz(float w, float x, float y, float z, float& yw, float& yz) {
    const float yaw = atan2f(2.0f * (w * z + x * y), 1.0f - 2.0f * (y * y + z * z));
    const float s = sinf(yaw * 0.5f), c = cosf(yaw * 0.5f);
    const float n = fmaxf(sqrtf(c * c + s * s), 1.0e-9f);  // normalize(): x / norm.clamp(min=eps)
    yw = c / n;
    yz = s / n;
}

// quat_apply (isaaclab/utils/math.py:545-564) for a yaw-only quaternion (w,0,0,z):
//   t = 2*cross(xyz, v);  out = v + w*t + cross(xyz, t)
IMX_DEV void quat_apply_yaw_only(float w, float z, float vx, float vy, float vz, float& ox, float& oy, float& oz) {
    const float tx = (0.0f * vz - z * vy) * 2.0f;
    const float ty = (z * vx - 0.0f * vz) * 2.0f;
    const float tz = (0.0f * vy - 0.0f * vx) * 2.0f;
    ox = vx + w * tx + (0.0f * tz - z * ty);
    oy = vy + w * ty + (z * tx - 0.0f * tz);
    oz = vz + w * tz + (0.0f * ty - 0.0f * tx);
}

// full quat_apply
IMX_DEV void quat_apply(float w, float x, float y, float z, float vx, float vy, float vz, float& ox, float& oy,
                        float& oz) {
    const float tx = (y * vz - z * vy) * 2.0f, ty = (z * vx - x * vz) * 2.0f, tz = (x * vy - y * vx) * 2.0f;
    ox = vx + w * tx + (y * tz - z * ty);
    oy = vy + w * ty + (z * tx - x * tz);
    oz = vz + w * tz + (x * ty - y * tx);
}

// wrap_to_pi (isaaclab/utils/math.py:95-117), torch.remainder semantics
IMX_DEV float wrap_to_pi(float a) {
    const float PI = 3.14159265358979323846f, TWO_PI = 6.28318530717958647692f;
    float m = fmodf(a + PI, TWO_PI);
    if (m != 0.0f && m < 0.0f) m += TWO_PI;
    return (m == 0.0f && a > 0.0f) ? PI : m - PI;
}

#define IMX_HALF_LOG_2PI 0.91893853320467274178f

// One element of ActionManager.process_action (action_manager.py:318-337): prev <- cur; cur <- clamp(a); the owning term's
// processed = raw*scale + offset [clamp | to-limits | EMA]  (joint_actions.py:130-139, joint_actions_to_limits.py).  Shared by k_action and
// by the actor head of k_mlp_infer (imx_mlp_infer_act), which calls it for the action it has just sampled.
IMX_DEV void action_process_element(const PlanView& P, const imx_state_t& S, const imx_buffers_t& Bf, int64_t e, int c, float a, float pre_clip) {
    const int64_t i = e * P.A + c;
    if (pre_clip < __builtin_huge_valf()) a = fminf(fmaxf(a, -pre_clip), pre_clip);  // torch.clamp
    Bf.prev_action[i] = Bf.action[i];
    Bf.action[i] = a;
    // find the term that owns column c (few terms; uniform loop)
    for (int k = 0; k < P.nact; ++k) {
        const int32_t* r = P.w + P.act_off + k * IMX_REC_WORDS;
        const int o = r[IMX_R_OUT], d = r[IMX_R_DIM];
        if (c < o || c >= o + d) continue;
        const int j = c - o;
        const int flags = r[IMX_R_FLAGS];
        const float scale = r[IMX_R_AUX0] ? f_of(P.w[r[IMX_R_AUX0] + j]) : f_of(r[IMX_R_P0]);
        float offset = r[IMX_R_AUX1] ? f_of(P.w[r[IMX_R_AUX1] + j]) : f_of(r[IMX_R_P1]);
        const int jid = P.w[r[IMX_R_IDS_OFF] + j];
        if (flags & IMX_F_ACT_DEFAULT_POS_OFFSET) offset = S.default_joint_pos[e * P.J + jid];
        if (flags & IMX_F_ACT_DEFAULT_VEL_OFFSET) offset = S.default_joint_vel[e * P.J + jid];
        float v = (flags & IMX_F_ACT_EMA) ? a * scale : a * scale + offset;  // (an EMA term has no offset: the slot carries alpha)
        if (flags & IMX_F_ACT_CLIP) {
            const float lo = f_of(P.w[r[IMX_R_IDS2_OFF] + 2 * j]), hi = f_of(P.w[r[IMX_R_IDS2_OFF] + 2 * j + 1]);
            v = fminf(fmaxf(v, lo), hi);
        }
        if (flags & IMX_F_ACT_TO_LIMITS) {  // clamp(-1, 1), then unscale_transform (utils/math.py:43-61): x * (upper - lower) * 0.5 + (lower + upper) * 0.5
            const float2 lim = reinterpret_cast<const float2*>(S.soft_joint_pos_limits)[e * P.J + jid];
            v = fminf(fmaxf(v, -1.0f), 1.0f);
            v = v * (lim.y - lim.x) * 0.5f + (lim.x + lim.y) * 0.5f;
        }
        if (flags & IMX_F_ACT_EMA) {  // joint_actions_to_limits.py:219-230
            const float2 lim = reinterpret_cast<const float2*>(S.soft_joint_pos_limits)[e * P.J + jid];
            const float prev = (Bf.reset_buf && Bf.reset_buf[e]) ? S.joint_pos[e * P.J + jid] : Bf.processed_action[i];
            v = offset * v + (1.0f - offset) * prev;
            v = fminf(fmaxf(v, lim.x), lim.y);
        }
        Bf.processed_action[i] = v;
    }
}
// host-side validation of what action_process_element dereferences (step.hip)
int imx_check_action_inputs(const imx_plan_t* plan, const imx_state_t* st, const imx_buffers_t* bf, const char* who);

IMX_DEV float norm3(float x, float y, float z) { return sqrtf((x * x + y * y) + z * z); }

// counter-based uniform [0,1): two rounds of a 32-bit multiply-xorshift hash (Wellons' "lowbias32") over
// (seed, step, element index); 24-bit mantissa like torch.rand.  ~12 VALU ops (a 64-bit splitmix cost ~40).
IMX_DEV float uniform01(uint64_t seed, uint32_t step, uint64_t idx) {
    uint32_t x = (uint32_t)idx ^ ((uint32_t)(idx >> 32) * 0x9E3779B9u) ^ (uint32_t)seed ^ ((uint32_t)(seed >> 32) * 0x85EBCA6Bu);
    x += step * 0x9E3779B9u + 0x7F4A7C15u;
    x ^= x >> 16; x *= 0x7FEB352Du;
    x ^= x >> 15; x *= 0x846CA68Bu;
    x ^= x >> 16;
    x += step; x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15;
    return (float)(x >> 8) * (1.0f / 16777216.0f);
}
