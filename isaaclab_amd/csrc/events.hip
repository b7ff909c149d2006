// Reset / interval events and the terrain curriculum as masked kernels.
//
// Reference (isaaclab/isaaclab/envs/mdp/events.py): reset_root_state_uniform :823-868, reset_joints_by_scale :987-1015,
// reset_joints_by_offset :1020-1049, push_by_setting_velocity :795-820; terrain_levels_vel
// (isaaclab_tasks/.../locomotion/velocity/mdp/curriculums.py:26-55) + TerrainImporter.update_env_origins
// (isaaclab/isaaclab/terrains/terrain_importer.py:307-326).  The reference runs them on a compacted env_ids list with a
// host sync per term (len(env_ids), nonzero); here they take the reset mask of imx_terminations_rewards and rewrite only
// the flagged rows of the "to simulator" state buffers (what write_root_pose_to_sim / write_root_velocity_to_sim /
// write_joint_state_to_sim receive).  Random draws: sample_uniform (utils/math.py:1313-1331) = u * (hi - lo) + lo with u
// from the counter-based generator, or from a caller-supplied table in parity runs (the reference's torch.rand stream
// cannot be reproduced in a kernel).
#include <algorithm>

#include "imx_internal.h"

namespace {

struct ResetCfg {
    float pose_lo[6], pose_hi[6], vel_lo[6], vel_hi[6];
    float jpos_lo, jpos_hi, jvel_lo, jvel_hi;
    int joint_mode;  // 0 scale, 1 offset, < 0: joints untouched
};

__device__ __forceinline__ float draw(const float* __restrict__ U, int64_t stride, int64_t e, int col, uint64_t seed, uint32_t step) {
    return U ? U[e * stride + col] : uniform01(seed + 0x9E3779B97F4A7C15ull * (uint64_t)(col + 1), step, (uint64_t)e);
}

// index space: [0, N) -> root state of env e; [N, N + N*J) -> joint (e, j)
__global__ void __launch_bounds__(256)
k_reset_events(int64_t N, int J, ResetCfg c, const uint8_t* __restrict__ mask, const float* __restrict__ drs,
               const float* __restrict__ origins, const float* __restrict__ djp, const float* __restrict__ djv,
               const float* __restrict__ plim, const float* __restrict__ vlim, const float* __restrict__ U, uint64_t seed,
               const int32_t* __restrict__ step_d, float* __restrict__ pose, float* __restrict__ vel, float* __restrict__ jpos,
               float* __restrict__ jvel) {
    const uint32_t step = step_d ? (uint32_t)step_d[0] : 0u;
    const int64_t ustride = 12 + 2 * (int64_t)J;
    const int64_t total = N + (c.joint_mode >= 0 ? N * J : 0);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        if (i < N) {
            const int64_t e = i;
            if (mask && !mask[e]) continue;
            const float* d = drs + e * 13;
            float rs[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) rs[k] = draw(U, ustride, e, k, seed, step) * (c.pose_hi[k] - c.pose_lo[k]) + c.pose_lo[k];
            // positions = default + env origin + sample (events.py:852)
            pose[e * 7 + 0] = d[0] + origins[e * 3 + 0] + rs[0];
            pose[e * 7 + 1] = d[1] + origins[e * 3 + 1] + rs[1];
            pose[e * 7 + 2] = d[2] + origins[e * 3 + 2] + rs[2];
            // quat_from_euler_xyz(roll, pitch, yaw) (math.py:266-276), then quat_mul(default, delta) (math.py:486-497)
            const float cy = cosf(rs[5] * 0.5f), sy = sinf(rs[5] * 0.5f), cr = cosf(rs[3] * 0.5f), sr = sinf(rs[3] * 0.5f);
            const float cp = cosf(rs[4] * 0.5f), sp = sinf(rs[4] * 0.5f);
            const float w2 = cy * cr * cp + sy * sr * sp, x2 = cy * sr * cp - sy * cr * sp, y2 = cy * cr * sp + sy * sr * cp,
                        z2 = sy * cr * cp - cy * sr * sp;
            const float w1 = d[3], x1 = d[4], y1 = d[5], z1 = d[6];
            const float ww = (z1 + x1) * (x2 + y2), yy = (w1 - y1) * (w2 + z2), zz = (w1 + y1) * (w2 - z2);
            const float xx = ww + yy + zz;
            const float qq = 0.5f * (xx + (z1 - x1) * (x2 - y2));
            pose[e * 7 + 3] = qq - ww + (z1 - y1) * (y2 - z2);
            pose[e * 7 + 4] = qq - xx + (x1 + w1) * (x2 + w2);
            pose[e * 7 + 5] = qq - yy + (w1 - x1) * (y2 + z2);
            pose[e * 7 + 6] = qq - zz + (z1 + y1) * (w2 - x2);
#pragma unroll
            for (int k = 0; k < 6; ++k)
                vel[e * 6 + k] = d[7 + k] + (draw(U, ustride, e, 6 + k, seed, step) * (c.vel_hi[k] - c.vel_lo[k]) + c.vel_lo[k]);
        } else {
            const int64_t q = i - N;
            const int64_t e = q / J;
            const int j = (int)(q - e * J);
            if (mask && !mask[e]) continue;
            const float sp = draw(U, ustride, e, 12 + j, seed, step) * (c.jpos_hi - c.jpos_lo) + c.jpos_lo;
            const float sv = draw(U, ustride, e, 12 + J + j, seed, step) * (c.jvel_hi - c.jvel_lo) + c.jvel_lo;
            float p = c.joint_mode == 1 ? djp[q] + sp : djp[q] * sp;
            float v = c.joint_mode == 1 ? djv[q] + sv : djv[q] * sv;
            p = fminf(fmaxf(p, plim[2 * q]), plim[2 * q + 1]);  // clamp_(lo, hi): min(max(x, lo), hi)
            v = fminf(fmaxf(v, -vlim[q]), vlim[q]);
            jpos[q] = p;
            jvel[q] = v;
        }
    }
}

struct PushCfg {
    float lo[6], hi[6];
};

__global__ void __launch_bounds__(256)
k_push_velocity(int64_t N, PushCfg c, const uint8_t* __restrict__ mask, const float* __restrict__ U, uint64_t seed,
                const int32_t* __restrict__ step_d, float* __restrict__ vel) {
    const uint32_t step = step_d ? (uint32_t)step_d[0] : 0u;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N * 6; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = i / 6;
        const int k = (int)(i - e * 6);
        if (mask && !mask[e]) continue;
        vel[i] += draw(U, 6, e, k, seed + 77u, step) * (c.hi[k] - c.lo[k]) + c.lo[k];
    }
}

// apply_external_force_torque (events.py:764-791): U(range) forces and torques on the selected bodies of the masked envs
__global__ void __launch_bounds__(256)
k_external_force_torque(int64_t N, int NB, int nb, const int32_t* __restrict__ body_ids, float f_lo, float f_hi, float t_lo, float t_hi,
                        const uint8_t* __restrict__ mask, const float* __restrict__ U, uint64_t seed, const int32_t* __restrict__ step_d,
                        float* __restrict__ forces, float* __restrict__ torques) {
    const uint32_t step = step_d ? (uint32_t)step_d[0] : 0u;
    const int64_t per_env = (int64_t)nb * 3, total = N * per_env;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = i / per_env;
        const int q = (int)(i - e * per_env), b = q / 3, c = q - 3 * b;
        if (mask && !mask[e]) continue;
        const int body = body_ids ? body_ids[b] : b;
        // uniforms: (2, N, nb, 3) -- forces first, torques second, like the two sample_uniform calls of the reference
        const float uf = U ? U[i] : uniform01(seed + 4242u, step, (uint64_t)i);
        const float ut = U ? U[total + i] : uniform01(seed + 4243u, step, (uint64_t)i);
        const int64_t o = (e * NB + body) * 3 + c;
        forces[o] = uf * (f_hi - f_lo) + f_lo;
        torques[o] = ut * (t_hi - t_lo) + t_lo;
    }
}

__global__ void __launch_bounds__(256)
k_terrain_levels(int64_t N, int R, int C, const uint8_t* __restrict__ mask, const float* __restrict__ root_pos,
                 const float* __restrict__ command, const float* __restrict__ terrain_origins, const int64_t* __restrict__ types,
                 float half_size, float max_len_s, const int64_t* __restrict__ rand_levels, uint64_t seed,
                 const int32_t* __restrict__ step_d, int64_t* __restrict__ levels, float* __restrict__ origins) {
    const uint32_t step = step_d ? (uint32_t)step_d[0] : 0u;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < N; e += (int64_t)gridDim.x * blockDim.x) {
        if (mask && !mask[e]) continue;
        const float dx = root_pos[e * 3] - origins[e * 3], dy = root_pos[e * 3 + 1] - origins[e * 3 + 1];
        const float dist = sqrtf(dx * dx + dy * dy);  // torch.norm(dim=1)
        const float cx = command[e * 3], cy = command[e * 3 + 1];
        const bool up = dist > half_size;
        const bool down = (dist < sqrtf(cx * cx + cy * cy) * max_len_s * 0.5f) && !up;
        int64_t lv = levels[e] + (up ? 1 : 0) - (down ? 1 : 0);
        if (lv >= R) {  // solved the last level: a random one (randint_like(levels, max_terrain_level))
            lv = rand_levels ? rand_levels[e] : (int64_t)(uniform01(seed + 991u, step, (uint64_t)e) * (float)R);
            lv = lv >= R ? R - 1 : lv;
        } else if (lv < 0) {
            lv = 0;
        }
        levels[e] = lv;
        const float* o = terrain_origins + ((size_t)lv * C + (size_t)types[e]) * 3;
        origins[e * 3] = o[0]; origins[e * 3 + 1] = o[1]; origins[e * 3 + 2] = o[2];
    }
}

// mean of terrain_levels.float() in a fixed order: one workgroup, strided partial sums, tree in LDS
__global__ void __launch_bounds__(1024) k_mean_levels(int64_t N, const int64_t* __restrict__ levels, float* __restrict__ out) {
    __shared__ float red[1024];
    float s = 0.0f;
    for (int64_t e = threadIdx.x; e < N; e += 1024) s += (float)levels[e];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0] / (float)N;
}

}  // namespace

extern "C" int imx_reset_events(int64_t N, int64_t J, const uint8_t* reset_mask_d, const float* ranges28, int joint_mode,
                                const float* default_root_state_d, const float* env_origins_d, const float* default_joint_pos_d,
                                const float* default_joint_vel_d, const float* soft_joint_pos_limits_d,
                                const float* soft_joint_vel_limits_d, const float* uniforms_d, uint64_t seed,
                                const int32_t* step_counter_d, float* root_pose_d, float* root_vel_d, float* joint_pos_d,
                                float* joint_vel_d, imx_stream_t stream) {
    IMX_REQUIRE(N > 0 && ranges28 && default_root_state_d && env_origins_d && root_pose_d && root_vel_d,
                "imx_reset_events: bad arguments");
    IMX_REQUIRE(joint_mode < 0 || (J > 0 && default_joint_pos_d && default_joint_vel_d && soft_joint_pos_limits_d &&
                                   soft_joint_vel_limits_d && joint_pos_d && joint_vel_d),
                "imx_reset_events: joint reset (mode %d) needs the joint defaults, limits and outputs", joint_mode);
    IMX_REQUIRE(joint_mode <= 1, "imx_reset_events: joint_mode %d (0 = by scale, 1 = by offset, < 0 = skip)", joint_mode);
    ResetCfg c;
    for (int k = 0; k < 6; ++k) {
        c.pose_lo[k] = ranges28[2 * k]; c.pose_hi[k] = ranges28[2 * k + 1];
        c.vel_lo[k] = ranges28[12 + 2 * k]; c.vel_hi[k] = ranges28[12 + 2 * k + 1];
    }
    c.jpos_lo = ranges28[24]; c.jpos_hi = ranges28[25]; c.jvel_lo = ranges28[26]; c.jvel_hi = ranges28[27];
    c.joint_mode = joint_mode;
    const int64_t total = N + (joint_mode >= 0 ? N * J : 0);
    const unsigned grid = (unsigned)std::min<int64_t>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(k_reset_events, dim3(grid), dim3(256), 0, (hipStream_t)stream, N, (int)J, c, reset_mask_d, default_root_state_d,
                       env_origins_d, default_joint_pos_d, default_joint_vel_d, soft_joint_pos_limits_d, soft_joint_vel_limits_d,
                       uniforms_d, seed, step_counter_d, root_pose_d, root_vel_d, joint_pos_d, joint_vel_d);
    IMX_HIP(hipGetLastError());
    return 0;
}

extern "C" int imx_push_velocity(int64_t N, const uint8_t* mask_d, const float* ranges12, const float* uniforms_d, uint64_t seed,
                                 const int32_t* step_counter_d, float* root_vel_w_d, imx_stream_t stream) {
    IMX_REQUIRE(N > 0 && ranges12 && root_vel_w_d, "imx_push_velocity: bad arguments");
    PushCfg c;
    for (int k = 0; k < 6; ++k) { c.lo[k] = ranges12[2 * k]; c.hi[k] = ranges12[2 * k + 1]; }
    const unsigned grid = (unsigned)std::min<int64_t>((N * 6 + 255) / 256, 4096);
    hipLaunchKernelGGL(k_push_velocity, dim3(grid), dim3(256), 0, (hipStream_t)stream, N, c, mask_d, uniforms_d, seed, step_counter_d,
                       root_vel_w_d);
    IMX_HIP(hipGetLastError());
    return 0;
}

extern "C" int imx_external_force_torque(int64_t N, int64_t num_bodies, const uint8_t* mask_d, const int32_t* body_ids_d, int64_t num_ids,
                                         const float* ranges4, const float* uniforms_d, uint64_t seed, const int32_t* step_counter_d,
                                         float* forces_d, float* torques_d, imx_stream_t stream) {
    IMX_REQUIRE(N > 0 && num_bodies > 0 && ranges4 && forces_d && torques_d, "imx_external_force_torque: bad arguments");
    const int64_t nb = body_ids_d ? num_ids : num_bodies;
    IMX_REQUIRE(nb > 0 && nb <= num_bodies, "imx_external_force_torque: %lld selected bodies of %lld", (long long)nb, (long long)num_bodies);
    const unsigned grid = (unsigned)std::min<int64_t>((N * nb * 3 + 255) / 256, 4096);
    hipLaunchKernelGGL(k_external_force_torque, dim3(grid), dim3(256), 0, (hipStream_t)stream, N, (int)num_bodies, (int)nb, body_ids_d,
                       ranges4[0], ranges4[1], ranges4[2], ranges4[3], mask_d, uniforms_d, seed, step_counter_d, forces_d, torques_d);
    IMX_HIP(hipGetLastError());
    return 0;
}

extern "C" int imx_terrain_levels(int64_t N, int64_t num_levels, int64_t num_types, const uint8_t* mask_d, const float* root_pos_w_d,
                                  const float* command_d, const float* terrain_origins_d, const int64_t* terrain_types_d,
                                  float terrain_size_x, float max_episode_length_s, const int64_t* rand_levels_d, uint64_t seed,
                                  const int32_t* step_counter_d, int64_t* terrain_levels_d, float* env_origins_d, float* mean_level_d,
                                  imx_stream_t stream) {
    IMX_REQUIRE(N > 0 && num_levels > 0 && num_types > 0 && root_pos_w_d && command_d && terrain_origins_d && terrain_types_d &&
                    terrain_levels_d && env_origins_d, "imx_terrain_levels: bad arguments");
    const unsigned grid = (unsigned)std::min<int64_t>((N + 255) / 256, 4096);
    hipLaunchKernelGGL(k_terrain_levels, dim3(grid), dim3(256), 0, (hipStream_t)stream, N, (int)num_levels, (int)num_types, mask_d,
                       root_pos_w_d, command_d, terrain_origins_d, terrain_types_d, 0.5f * terrain_size_x, max_episode_length_s,
                       rand_levels_d, seed, step_counter_d, terrain_levels_d, env_origins_d);
    IMX_HIP(hipGetLastError());
    if (mean_level_d) {
        hipLaunchKernelGGL(k_mean_levels, dim3(1), dim3(1024), 0, (hipStream_t)stream, N, terrain_levels_d, mean_level_d);
        IMX_HIP(hipGetLastError());
    }
    return 0;
}
