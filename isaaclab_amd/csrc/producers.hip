// Input producers of the hot path that the reference runs in eager torch every step:
//   * ContactSensor.update -> _update_buffers_impl: force history shift + air/contact-time bookkeeping
//     (reference isaaclab/sensors/contact_sensor/contact_sensor.py:320-379, sensors/sensor_base.py:196-205,287-297)
//   * CommandManager.compute / UniformVelocityCommand (isaaclab/managers/command_manager.py:122-187,
//     isaaclab/envs/mdp/commands/velocity_command.py:111-160): metrics, resampling timer, uniform resampling,
//     heading P-controller (wrap_to_pi), standing envs.
#include "imx_internal.h"
#include "imx_producers.h"

// ------------------------------------------------------------------------------------------------- contact sensor
// lane = (env, body).  SensorBase.update: timestamp += dt; outdated |= timestamp - last_update + 1e-6 >= update_period;
// outdated envs run _update_buffers_impl with elapsed = timestamp - last_update, then last_update = timestamp.
__global__ void __launch_bounds__(256)
k_contact_update(int64_t N, int B, int H, const float* __restrict__ new_forces, float dt, float update_period,
                 float force_threshold, int track_air_time, float* __restrict__ timestamp,
                 float* __restrict__ timestamp_last, uint8_t* __restrict__ is_outdated, float* __restrict__ net_forces,
                 float* __restrict__ history, float* __restrict__ last_air, float* __restrict__ cur_air,
                 float* __restrict__ last_contact, float* __restrict__ cur_contact) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * B) return;
    const int64_t e = i / B;
    const int b = (int)(i - e * B);
    const float ts = timestamp[e] + dt;  // every lane of an env computes the same value; body 0 stores it
    const float tl = timestamp_last[e];
    const bool outdated = (is_outdated[e] != 0) || (ts - tl + 1.0e-6f >= update_period);
    if (outdated) {
        const float fx = new_forces[i * 3], fy = new_forces[i * 3 + 1], fz = new_forces[i * 3 + 2];
        net_forces[i * 3] = fx; net_forces[i * 3 + 1] = fy; net_forces[i * 3 + 2] = fz;
        if (history && H > 0) {  // history[:, 1:] = history[:, :-1]; history[:, 0] = net_forces
            for (int h = H - 1; h >= 1; --h) {
                float* dst = history + (((size_t)e * H + h) * B + b) * 3;
                const float* src = history + (((size_t)e * H + h - 1) * B + b) * 3;
                dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2];
            }
            float* d0 = history + ((size_t)e * H * B + b) * 3;
            d0[0] = fx; d0[1] = fy; d0[2] = fz;
        }
        if (track_air_time) {
            const float elapsed = ts - tl;
            const bool is_contact = norm3(fx, fy, fz) > force_threshold;
            const float ca = cur_air[i], cc = cur_contact[i];
            const bool first_contact = (ca > 0.0f) && is_contact;
            const bool first_detached = (cc > 0.0f) && !is_contact;
            if (first_contact) last_air[i] = ca + elapsed;
            cur_air[i] = !is_contact ? ca + elapsed : 0.0f;
            if (first_detached) last_contact[i] = cc + elapsed;
            cur_contact[i] = is_contact ? cc + elapsed : 0.0f;
        }
    }
}

// second pass (the per-env scalars may only change once every body lane has read them)
__global__ void k_contact_stamp(int64_t N, float dt, float update_period, float* __restrict__ timestamp,
                                float* __restrict__ timestamp_last, uint8_t* __restrict__ is_outdated) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N) return;
    const float ts = timestamp[e] + dt;
    const bool outdated = (is_outdated[e] != 0) || (ts - timestamp_last[e] + 1.0e-6f >= update_period);
    timestamp[e] = ts;
    if (outdated) {
        timestamp_last[e] = ts;
        is_outdated[e] = 0;
    }
}

extern "C" int imx_contact_sensor_update(int64_t N, int64_t B, int64_t H, const float* new_net_forces_d, float dt,
                                         float update_period, float force_threshold, int track_air_time,
                                         float* timestamp_d, float* timestamp_last_update_d, uint8_t* is_outdated_d,
                                         float* net_forces_w_d, float* net_forces_w_history_d, float* last_air_time_d,
                                         float* current_air_time_d, float* last_contact_time_d,
                                         float* current_contact_time_d, imx_stream_t stream) {
    IMX_REQUIRE(N > 0 && B > 0 && H >= 0, "imx_contact_sensor_update: bad sizes");
    IMX_REQUIRE(new_net_forces_d && timestamp_d && timestamp_last_update_d && is_outdated_d && net_forces_w_d,
                "imx_contact_sensor_update: null argument");
    IMX_REQUIRE(H == 0 || net_forces_w_history_d, "imx_contact_sensor_update: history buffer missing");
    IMX_REQUIRE(!track_air_time || (last_air_time_d && current_air_time_d && last_contact_time_d && current_contact_time_d),
                "imx_contact_sensor_update: air-time buffers missing");
    const int64_t n = N * B;
    hipLaunchKernelGGL(k_contact_update, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, N, (int)B, (int)H,
                       new_net_forces_d, dt, update_period, force_threshold, track_air_time, timestamp_d,
                       timestamp_last_update_d, is_outdated_d, net_forces_w_d, net_forces_w_history_d, last_air_time_d,
                       current_air_time_d, last_contact_time_d, current_contact_time_d);
    hipLaunchKernelGGL(k_contact_stamp, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, N, dt,
                       update_period, timestamp_d, timestamp_last_update_d, is_outdated_d);
    IMX_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------- velocity command
// lane = env (the per-env logic is velocity_command_env, imx_producers.h: shared with the orchestration kernel)
__global__ void __launch_bounds__(256)
k_velocity_command(int64_t N, VelCmdCfg c, float dt, int do_compute, const float* __restrict__ quat, const float* __restrict__ lin_w,
                   const float* __restrict__ ang_w, const uint8_t* __restrict__ reset_mask,
                   const float* __restrict__ uniforms, uint64_t seed, const int32_t* __restrict__ step_d, VelCmdState s) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N) return;
    const uint32_t step = step_d ? (uint32_t)step_d[0] : 0u;
    float m0, m1;
    velocity_command_env(N, e, c, dt, do_compute, quat, lin_w, ang_w, reset_mask && reset_mask[e], uniforms, seed, step, s, m0, m1);
}

extern "C" int imx_velocity_command(int64_t N, const float* cfg15, int heading_command, float dt, int do_compute,
                                    const float* root_quat_w_d,
                                    const float* root_lin_vel_w_d, const float* root_ang_vel_w_d,
                                    const uint8_t* reset_mask_d, const float* uniforms_d, uint64_t seed,
                                    const int32_t* step_counter_d, float* vel_command_b_d, float* heading_target_d,
                                    uint8_t* is_heading_env_d, uint8_t* is_standing_env_d, float* time_left_d,
                                    int64_t* command_counter_d, float* metric_error_vel_xy_d,
                                    float* metric_error_vel_yaw_d, imx_stream_t stream) {
    IMX_REQUIRE(N > 0 && cfg15, "imx_velocity_command: bad arguments");
    IMX_REQUIRE(root_quat_w_d && root_lin_vel_w_d && root_ang_vel_w_d && vel_command_b_d && heading_target_d &&
                    is_heading_env_d && is_standing_env_d && time_left_d && command_counter_d && metric_error_vel_xy_d &&
                    metric_error_vel_yaw_d, "imx_velocity_command: null argument");
    const VelCmdCfg c = vel_cmd_cfg_from15(cfg15, heading_command);
    IMX_REQUIRE(c.max_command_step > 0.0f, "imx_velocity_command: max_command_step must be positive");
    VelCmdState st{vel_command_b_d, heading_target_d, is_heading_env_d, is_standing_env_d, time_left_d, command_counter_d,
                   metric_error_vel_xy_d, metric_error_vel_yaw_d};
    hipLaunchKernelGGL(k_velocity_command, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, N, c, dt,
                       do_compute, root_quat_w_d, root_lin_vel_w_d, root_ang_vel_w_d, reset_mask_d, uniforms_d, seed, step_counter_d, st);
    IMX_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------- articulation data
// ArticulationData.root_state_w (reference isaaclab/assets/articulation/articulation_data.py:365-380): PhysX root
// transforms are (pos, quat XYZW) -> convert_quat(to="wxyz") (utils/math.py:177-222) and root velocities (lin, ang);
// ArticulationData.joint_acc (:546-556): finite difference (joint_vel - previous_joint_vel) / elapsed, previous <- current.
// One launch: lanes [0, N) split the root state, all lanes stride over the N*J joint entries.
__global__ void __launch_bounds__(256)
k_articulation_update(int64_t N, int J, const float* __restrict__ root_tf, const float* __restrict__ root_vel,
                      const float* __restrict__ dof_vel, float elapsed, float* __restrict__ pos, float* __restrict__ quat,
                      float* __restrict__ lin, float* __restrict__ ang, float* __restrict__ prev_vel,
                      float* __restrict__ joint_acc) {
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nth = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = tid; e < N; e += nth) {
        const float* t = root_tf + e * 7;
        pos[e * 3] = t[0]; pos[e * 3 + 1] = t[1]; pos[e * 3 + 2] = t[2];
        quat[e * 4] = t[6]; quat[e * 4 + 1] = t[3]; quat[e * 4 + 2] = t[4]; quat[e * 4 + 3] = t[5];  // xyzw -> wxyz
        const float* v = root_vel + e * 6;
        lin[e * 3] = v[0]; lin[e * 3 + 1] = v[1]; lin[e * 3 + 2] = v[2];
        ang[e * 3] = v[3]; ang[e * 3 + 1] = v[4]; ang[e * 3 + 2] = v[5];
    }
    if (joint_acc) {
        const int64_t n = N * J;
        for (int64_t i = tid; i < n; i += nth) {
            const float v = dof_vel[i];
            joint_acc[i] = (v - prev_vel[i]) / elapsed;
            prev_vel[i] = v;
        }
    }
}

extern "C" int imx_articulation_update(int64_t N, int64_t J, const float* root_transforms_xyzw_d,
                                       const float* root_velocities_d, const float* dof_velocities_d, float time_elapsed,
                                       float* root_pos_w_d, float* root_quat_w_d, float* root_lin_vel_w_d,
                                       float* root_ang_vel_w_d, float* previous_joint_vel_d, float* joint_acc_d,
                                       imx_stream_t stream) {
    IMX_REQUIRE(N > 0 && J >= 0, "imx_articulation_update: bad sizes");
    IMX_REQUIRE(root_transforms_xyzw_d && root_velocities_d && root_pos_w_d && root_quat_w_d && root_lin_vel_w_d &&
                    root_ang_vel_w_d, "imx_articulation_update: null argument");
    IMX_REQUIRE(!joint_acc_d || (dof_velocities_d && previous_joint_vel_d && J > 0 && time_elapsed > 0.0f),
                "imx_articulation_update: joint_acc needs dof velocities, the previous velocities and a positive elapsed time");
    const int64_t n = std::max<int64_t>(N, joint_acc_d ? N * J : 0);
    const unsigned grid = (unsigned)std::min<int64_t>((n + 255) / 256, 2048);
    hipLaunchKernelGGL(k_articulation_update, dim3(grid), dim3(256), 0, (hipStream_t)stream, N, (int)J, root_transforms_xyzw_d,
                       root_velocities_d, dof_velocities_d, time_elapsed, root_pos_w_d, root_quat_w_d, root_lin_vel_w_d,
                       root_ang_vel_w_d, previous_joint_vel_d, joint_acc_d);
    IMX_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------- empirical normalization
// rsl_rl EmpiricalNormalization (upstream rsl_rl/modules/normalizer.py @ v2.3.1; enabled by
// RslRlOnPolicyRunnerCfg.empirical_normalization, reference isaaclab_rl/rsl_rl/rl_cfg.py): running mean / variance
// of the observations (Chan's parallel update with the batch mean / biased variance), then (x - mean) / (std + eps).
// PARITY UNPINNED (rsl_rl absent).  One block per 64 columns: column-wise batch moments, fixed-order merge.
__global__ void __launch_bounds__(256)
k_norm_update(int64_t N, int D, const float* __restrict__ x, float* __restrict__ mean, float* __restrict__ var,
              float* __restrict__ stdv, float* __restrict__ count_d, int update_count) {
    __shared__ float s_sum[4][64], s_sq[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rgrp = threadIdx.x >> 6;
    float bm = 0.0f, bv = 0.0f;
    // two-pass batch moments (mean, then biased variance), rows strided over the 4 waves
    float s = 0.0f;
    if (c < D)
        for (int64_t r = rgrp; r < N; r += 4) s += x[r * D + c];
    s_sum[rgrp][threadIdx.x & 63] = s;
    __syncthreads();
    if (c < D) bm = ((s_sum[0][threadIdx.x & 63] + s_sum[1][threadIdx.x & 63]) + (s_sum[2][threadIdx.x & 63] + s_sum[3][threadIdx.x & 63])) / (float)N;
    float q = 0.0f;
    if (c < D)
        for (int64_t r = rgrp; r < N; r += 4) { const float d = x[r * D + c] - bm; q += d * d; }
    s_sq[rgrp][threadIdx.x & 63] = q;
    __syncthreads();
    if (rgrp == 0 && c < D) {
        bv = ((s_sq[0][threadIdx.x] + s_sq[1][threadIdx.x]) + (s_sq[2][threadIdx.x] + s_sq[3][threadIdx.x])) / (float)N;
        const float cnt = count_d[0];
        const float rate = (float)N / (cnt + (float)N);
        const float delta = bm - mean[c];
        const float m_new = mean[c] + rate * delta;
        const float v_new = var[c] + rate * (bv - var[c] + delta * (bm - m_new));
        mean[c] = m_new;
        var[c] = v_new;
        stdv[c] = sqrtf(v_new);
    }
    (void)update_count;
}
__global__ void k_norm_count(int64_t N, float* __restrict__ count_d) {
    if (threadIdx.x == 0 && blockIdx.x == 0) count_d[0] += (float)N;
}
__global__ void __launch_bounds__(256)
k_norm_apply(int64_t n, int D, const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ stdv,
             float eps, float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % D);
        out[i] = (x[i] - mean[c]) / (stdv[c] + eps);
    }
}

extern "C" int imx_empirical_normalization(int64_t N, int64_t D, const float* x_d, int update, float eps, float* mean_d,
                                           float* var_d, float* std_d, float* count_d, float* out_d, imx_stream_t stream) {
    IMX_REQUIRE(N > 0 && D > 0 && x_d && mean_d && var_d && std_d && count_d && out_d, "imx_empirical_normalization: bad arguments");
    if (update) {
        hipLaunchKernelGGL(k_norm_update, dim3((unsigned)((D + 63) / 64)), dim3(256), 0, (hipStream_t)stream, N, (int)D, x_d,
                           mean_d, var_d, std_d, count_d, 1);
        hipLaunchKernelGGL(k_norm_count, dim3(1), dim3(64), 0, (hipStream_t)stream, N, count_d);
    }
    const int64_t n = N * D;
    hipLaunchKernelGGL(k_norm_apply, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 4096)), dim3(256), 0, (hipStream_t)stream,
                       n, (int)D, x_d, mean_d, std_d, eps, out_d);
    IMX_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------- explicit actuator models
// IdealPDActuator.compute (isaaclab/isaaclab/actuators/actuator_pd.py:184-199; ImplicitActuator.compute :115-145 evaluates the
// same law for reporting): computed = kp*(q_des - q) + kd*(qd_des - qd) + ff; applied = clip(computed, +-effort_limit)
// (actuator_base.py:309-318).  DCMotor (:264-286): the clip window depends on the joint velocity,
// max = clip(sat*(1 - qd/v_lim), 0, limit), min = clip(sat*(-1 - qd/v_lim), -limit, 0).  SURVEY 8f row 4.
__global__ void __launch_bounds__(256)
k_actuator_pd(int64_t n, int dc_motor, float saturation, const float* __restrict__ q_des, const float* __restrict__ qd_des,
              const float* __restrict__ ff, const float* __restrict__ q, const float* __restrict__ qd,
              const float* __restrict__ kp, const float* __restrict__ kd, const float* __restrict__ elim,
              const float* __restrict__ vlim, float* __restrict__ computed, float* __restrict__ applied) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = qd[i];
        const float err_p = q_des[i] - q[i];
        const float err_v = (qd_des ? qd_des[i] : 0.0f) - v;
        const float c = kp[i] * err_p + kd[i] * err_v + (ff ? ff[i] : 0.0f);
        const float lim = elim[i];
        float lo = -lim, hi = lim;
        if (dc_motor) {
            const float r = v / vlim[i];
            hi = fminf(fmaxf(saturation * (1.0f - r), 0.0f), lim);   // torch.clip(x, min=a, max=b) = min(max(x, a), b)
            lo = fminf(fmaxf(saturation * (-1.0f - r), -lim), 0.0f);
        }
        computed[i] = c;
        applied[i] = fminf(fmaxf(c, lo), hi);
    }
}

extern "C" int imx_actuator_pd(int64_t N, int64_t J, int dc_motor, float saturation_effort, const float* joint_pos_target_d,
                               const float* joint_vel_target_d, const float* effort_ff_d, const float* joint_pos_d,
                               const float* joint_vel_d, const float* stiffness_d, const float* damping_d, const float* effort_limit_d,
                               const float* velocity_limit_d, float* computed_effort_d, float* applied_effort_d, imx_stream_t stream) {
    IMX_REQUIRE(N > 0 && J > 0 && joint_pos_target_d && joint_pos_d && joint_vel_d && stiffness_d && damping_d && effort_limit_d &&
                    computed_effort_d && applied_effort_d, "imx_actuator_pd: bad arguments");
    IMX_REQUIRE(!dc_motor || velocity_limit_d, "imx_actuator_pd: the DC motor model needs the velocity limits");
    const int64_t n = N * J;
    hipLaunchKernelGGL(k_actuator_pd, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 4096)), dim3(256), 0, (hipStream_t)stream, n,
                       dc_motor, saturation_effort, joint_pos_target_d, joint_vel_target_d, effort_ff_d, joint_pos_d, joint_vel_d,
                       stiffness_d, damping_d, effort_limit_d, velocity_limit_d, computed_effort_d, applied_effort_d);
    IMX_HIP(hipGetLastError());
    return 0;
}

// DelayedPDActuator / RemotizedPDActuator (actuators/actuator_pd.py:289-412).  The three DelayBuffers of the reference share
// pointer, push counts and lags: one ring (L+1, N, 3, J) (slot-major: a step writes one contiguous slab).  Host-known scalars
// replace the reference's host syncs: `step` = appends so far (pointer = step mod (L+1)), pushes of env e = step - reset_step[e]
// (reset_step[e] = the step index of the first append after its last reset).  A fresh env (zero pushes) fills every slot with
// its first sample (circular_buffer.py:141-146); the sample read back is `min(lag, pushes - 1)` appends old (:160-165).
// lookup (K,3) != NULL: angle-dependent torque limit by linear interpolation (linear_interpolation.py:54-86) after the PD law.
__global__ void __launch_bounds__(256)
k_actuator_delayed_pd(int64_t N, int J, int L1, int64_t step, const int32_t* __restrict__ lags, const int64_t* __restrict__ reset_step,
                      float* __restrict__ ring, const float* __restrict__ q_des, const float* __restrict__ qd_des,
                      const float* __restrict__ ff, const float* __restrict__ q, const float* __restrict__ qd,
                      const float* __restrict__ kp, const float* __restrict__ kd, const float* __restrict__ elim,
                      const float* __restrict__ lookup, int K, float* __restrict__ computed, float* __restrict__ applied) {
    const int64_t n = N * J;
    const int ptr = (int)(step % L1);
    const int64_t slot = N * 3 * (int64_t)J;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = i / J;
        const int j = (int)(i - e * J);
        const int64_t pushes = step - reset_step[e];  // before this append
        const float in[3] = {q_des[i], qd_des ? qd_des[i] : 0.0f, ff ? ff[i] : 0.0f};
        float* base = ring + (e * 3) * (int64_t)J + j;
        float d[3];
        if (pushes <= 0) {
            for (int sl = 0; sl < L1; ++sl)
                for (int c = 0; c < 3; ++c) base[sl * slot + c * J] = in[c];
            for (int c = 0; c < 3; ++c) d[c] = in[c];
        } else {
            const int64_t valid = min((int64_t)lags[e], pushes);  // min(lag, pushes_after - 1)
            int idx = (int)((ptr - valid) % L1);
            if (idx < 0) idx += L1;
            for (int c = 0; c < 3; ++c) {
                base[ptr * slot + c * J] = in[c];
                d[c] = valid == 0 ? in[c] : base[idx * slot + c * J];
            }
        }
        const float qi = q[i];
        const float c = kp[i] * (d[0] - qi) + kd[i] * (d[1] - qd[i]) + d[2];
        float lim = elim ? elim[i] : __builtin_huge_valf();
        float a = fminf(fmaxf(c, -lim), lim);
        if (lookup) {
            int ns = 0;
            for (int k = 0; k < K; ++k) ns += lookup[3 * k] < qi;
            const int lb = max(ns - 1, 0), ub = min(ns, K - 1);
            const float xl = lookup[3 * lb], xu = lookup[3 * ub], yl = lookup[3 * lb + 2], yu = lookup[3 * ub + 2];
            const float w = ub == lb ? 0.0f : (qi - xl) / (xu - xl);
            lim = yl + w * (yu - yl);
            a = fminf(fmaxf(a, -lim), lim);
        }
        computed[i] = c;
        applied[i] = a;
    }
}

extern "C" int imx_actuator_delayed_pd(int64_t N, int64_t J, int max_delay, int64_t step, const int32_t* time_lags_d,
                                       const int64_t* reset_step_d, float* ring_d, const float* joint_pos_target_d,
                                       const float* joint_vel_target_d, const float* effort_ff_d, const float* joint_pos_d,
                                       const float* joint_vel_d, const float* stiffness_d, const float* damping_d,
                                       const float* effort_limit_d, const float* lookup_d, int num_lookup,
                                       float* computed_effort_d, float* applied_effort_d, imx_stream_t stream) {
    IMX_REQUIRE(N > 0 && J > 0 && max_delay >= 0 && max_delay < 1024 && step >= 0, "imx_actuator_delayed_pd: bad sizes");
    IMX_REQUIRE(time_lags_d && reset_step_d && ring_d && joint_pos_target_d && joint_pos_d && joint_vel_d && stiffness_d && damping_d &&
                    computed_effort_d && applied_effort_d, "imx_actuator_delayed_pd: null argument");
    IMX_REQUIRE(!lookup_d || num_lookup > 0, "imx_actuator_delayed_pd: empty lookup table");  // linear_interpolation.py:46-47
    const int64_t n = N * J;
    hipLaunchKernelGGL(k_actuator_delayed_pd, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 4096)), dim3(256), 0,
                       (hipStream_t)stream, N, (int)J, max_delay + 1, step, time_lags_d, reset_step_d, ring_d, joint_pos_target_d,
                       joint_vel_target_d, effort_ff_d, joint_pos_d, joint_vel_d, stiffness_d, damping_d, effort_limit_d, lookup_d,
                       num_lookup, computed_effort_d, applied_effort_d);
    IMX_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// ActuatorNetLSTM / ActuatorNetMLP .compute (actuators/actuator_net.py:29-104, 107-195): the learned actuator models of ANYmal
// (ANYDRIVE_3_LSTM_ACTUATOR_CFG is the actuator of the Anymal-C task robot).  The reference evaluates a TorchScript network on
// (N*J, ...) inputs; here one lane = one (env, joint) pair carries its sample through the whole network -- LSTM stack + MLP head, or
// the history MLP -- with the weights staged once per workgroup in LDS (every lane reads the same weight: broadcast) and the
// lane's activation vectors in LDS columns (lane-minor: conflict-free, dynamically indexable, any layer width up to 64).
//
// Packed network (floats), written by isaaclab_amd/producers.py:
//   LSTM layer l (in_0 = 2, in_l = H): W_ih (4H x in_l) row-major, W_hh (4H x H), b_ih (4H), b_hh (4H); gate rows i, f, g, o (torch)
//   dense layer k: W (out_k x in_k) row-major, b (out_k); `act` between the dense layers, none after the last
struct ImxNetDesc {
    int num_lstm, hidden, num_dense, act;  // act: 0 identity 1 softsign 2 tanh 3 relu 4 elu
    int dense_out[4];
    int num_weights;
};
#define IMX_NET_MAXW 64
IMX_DEV float net_act(float x, int act) {
    switch (act) {
        case 1: return x / (1.0f + fabsf(x));
        case 2: return tanhf(x);
        case 3: return fmaxf(x, 0.0f);
        case 4: return x > 0.0f ? x : expm1f(x);
        default: return x;
    }
}
IMX_DEV float net_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// dense layers on the lane's vector in column `va` (in_dim entries) -> result in va again (vb is scratch); returns the last width
IMX_DEV int net_dense(const ImxNetDesc& D, const float* __restrict__ w, int in_dim, float* va, float* vb, int lane) {
    for (int k = 0; k < D.num_dense; ++k) {
        const int out_dim = D.dense_out[k];
        const float* W = w;
        const float* b = w + (size_t)out_dim * in_dim;
        for (int j = 0; j < out_dim; ++j) {
            float acc = 0.0f;
            for (int q = 0; q < in_dim; ++q) acc += W[j * in_dim + q] * va[q * 64 + lane];
            acc += b[j];
            vb[j * 64 + lane] = k + 1 < D.num_dense ? net_act(acc, D.act) : acc;
        }
        for (int j = 0; j < out_dim; ++j) va[j * 64 + lane] = vb[j * 64 + lane];
        w = b + out_dim;
        in_dim = out_dim;
    }
    return in_dim;
}

IMX_DEV float dc_motor_clip(float c, float v, float saturation, float elim, float vlim) {  // DCMotor._clip_effort (actuator_pd.py:276-286)
    const float r = v / vlim;
    const float hi = fminf(fmaxf(saturation * (1.0f - r), 0.0f), elim);
    const float lo = fminf(fmaxf(saturation * (-1.0f - r), -elim), 0.0f);
    return fminf(fmaxf(c, lo), hi);
}

__global__ void __launch_bounds__(64)
k_actuator_net_lstm(int64_t n, ImxNetDesc D, const float* __restrict__ weights, const float* __restrict__ q_des, const float* __restrict__ q,
                    const float* __restrict__ qd, float* __restrict__ hid, float* __restrict__ cell, float saturation,
                    const float* __restrict__ elim, const float* __restrict__ vlim, float* __restrict__ computed, float* __restrict__ applied) {
    extern __shared__ float s_net[];
    float* s_w = s_net;                          // the packed network
    float* va = s_net + ((D.num_weights + 3) & ~3);  // lane vectors: IMX_NET_MAXW x 64 each
    float* vb = va + IMX_NET_MAXW * 64;
    float* vh = vb + IMX_NET_MAXW * 64;          // previous hidden state of the layer
    const int lane = threadIdx.x;
    for (int k = lane; k < D.num_weights; k += 64) s_w[k] = weights[k];
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * 64 + lane;
    if (i >= n) return;
    const int H = D.hidden;
    const float vel = qd[i];
    va[lane] = q_des[i] - q[i];  // sea_input[:, 0, 0] = pos error, [:, 0, 1] = joint velocity (:77-78)
    va[64 + lane] = vel;
    const float* w = s_w;
    int in_dim = 2;
    for (int l = 0; l < D.num_lstm; ++l) {
        const float* W_ih = w;
        const float* W_hh = W_ih + (size_t)4 * H * in_dim;
        const float* b_ih = W_hh + (size_t)4 * H * H;
        const float* b_hh = b_ih + 4 * H;
        float* hs = hid + ((size_t)l * n + i) * H;   // (num_layers, N*J, H), the reference's layout
        float* cs = cell + ((size_t)l * n + i) * H;
        for (int k = 0; k < H; ++k) vh[k * 64 + lane] = hs[k];
        for (int k = 0; k < H; ++k) {
            float g4[4];
#pragma unroll
            for (int gI = 0; gI < 4; ++gI) {  // gates i, f, g, o of unit k: linear(input) + linear(hidden) (torch's LSTM cell)
                const int r = gI * H + k;
                float a = 0.0f, b = 0.0f;
                for (int qn = 0; qn < in_dim; ++qn) a += W_ih[r * in_dim + qn] * va[qn * 64 + lane];
                for (int qn = 0; qn < H; ++qn) b += W_hh[r * H + qn] * vh[qn * 64 + lane];
                g4[gI] = (a + b_ih[r]) + (b + b_hh[r]);
            }
            const float c_new = net_sigmoid(g4[1]) * cs[k] + net_sigmoid(g4[0]) * tanhf(g4[2]);
            const float h_new = net_sigmoid(g4[3]) * tanhf(c_new);
            cs[k] = c_new;
            vb[k * 64 + lane] = h_new;
        }
        for (int k = 0; k < H; ++k) {
            const float h_new = vb[k * 64 + lane];
            hs[k] = h_new;
            va[k * 64 + lane] = h_new;  // input of the next layer / of the head
        }
        w = b_hh + 4 * H;
        in_dim = H;
    }
    net_dense(D, w, in_dim, va, vb, lane);
    const float c = va[lane];
    computed[i] = c;
    applied[i] = dc_motor_clip(c, vel, saturation, elim[i], vlim[i]);
}

// The ANYdrive shapes (hidden size 8, one or two LSTM layers, a head of 8 -> 1 or 8 -> D0 -> 1) with everything in REGISTERS: the lane's
// input / hidden / cell vectors are compile-time sized arrays, every loop is unrolled, and the weights -- the same address for all 64
// lanes -- arrive through the scalar cache as SGPR operands of the multiply-adds.  No LDS at all: the generic kernel above keeps
// 3 x 64 x 64 floats of lane vectors + the network in LDS (54 KB per single-wave workgroup = two waves per CU), which made
// ActuatorNetLSTM.compute 56 us per physics substep at 4096 x 12 samples -- 4 x 56 us per env step against 30 us for the whole
// post-physics path (profiles/r03_*).  Same accumulation order as the generic kernel (and as torch's LSTM cell: gates i, f, g, o).
template <int H, int D0>
__global__ void __launch_bounds__(256)
k_actuator_net_lstm_reg(int64_t n, int num_lstm, int act, const float* __restrict__ weights, const float* __restrict__ q_des,
                        const float* __restrict__ q, const float* __restrict__ qd, float* __restrict__ hid, float* __restrict__ cell,
                        float saturation, const float* __restrict__ elim, const float* __restrict__ vlim, float* __restrict__ computed,
                        float* __restrict__ applied) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float vel = qd[i];
    float x[H];  // input of the current layer (layer 0: {position error, velocity})
    x[0] = q_des[i] - q[i];  // sea_input[:, 0, 0] = pos error, [:, 0, 1] = joint velocity (:77-78)
    x[1] = vel;
    const float* __restrict__ w = weights;
    for (int l = 0; l < num_lstm; ++l) {  // (uniform trip count)
        const int in_dim = l == 0 ? 2 : H;
        const float* __restrict__ W_ih = w;
        const float* __restrict__ W_hh = W_ih + 4 * H * in_dim;
        const float* __restrict__ b_ih = W_hh + 4 * H * H;
        const float* __restrict__ b_hh = b_ih + 4 * H;
        float* hs = hid + ((size_t)l * n + i) * H;   // (num_layers, N*J, H), the reference's layout
        float* cs = cell + ((size_t)l * n + i) * H;
        float h[H], c[H], hn[H];
#pragma unroll
        for (int k = 0; k < H; k += 4) {
            const float4 hv = *reinterpret_cast<const float4*>(hs + k), cv = *reinterpret_cast<const float4*>(cs + k);
            h[k] = hv.x; h[k + 1] = hv.y; h[k + 2] = hv.z; h[k + 3] = hv.w;
            c[k] = cv.x; c[k + 1] = cv.y; c[k + 2] = cv.z; c[k + 3] = cv.w;
        }
#pragma unroll
        for (int k = 0; k < H; ++k) {
            float g4[4];
#pragma unroll
            for (int gI = 0; gI < 4; ++gI) {
                const int r = gI * H + k;
                float a = 0.0f, b = 0.0f;
                if (l == 0) {
#pragma unroll
                    for (int qn = 0; qn < 2; ++qn) a += W_ih[r * 2 + qn] * x[qn];
                } else {
#pragma unroll
                    for (int qn = 0; qn < H; ++qn) a += W_ih[r * H + qn] * x[qn];
                }
#pragma unroll
                for (int qn = 0; qn < H; ++qn) b += W_hh[r * H + qn] * h[qn];
                g4[gI] = (a + b_ih[r]) + (b + b_hh[r]);
            }
            const float c_new = net_sigmoid(g4[1]) * c[k] + net_sigmoid(g4[0]) * tanhf(g4[2]);
            hn[k] = net_sigmoid(g4[3]) * tanhf(c_new);
            c[k] = c_new;
        }
#pragma unroll
        for (int k = 0; k < H; k += 4) {
            *reinterpret_cast<float4*>(hs + k) = make_float4(hn[k], hn[k + 1], hn[k + 2], hn[k + 3]);
            *reinterpret_cast<float4*>(cs + k) = make_float4(c[k], c[k + 1], c[k + 2], c[k + 3]);
        }
#pragma unroll
        for (int k = 0; k < H; ++k) x[k] = hn[k];
        w = b_hh + 4 * H;
    }
    // the head: H -> 1 (D0 == 0) or H -> D0 -> 1 with `act` in between
    float out;
    if (D0 == 0) {
        float acc = 0.0f;
#pragma unroll
        for (int qn = 0; qn < H; ++qn) acc += w[qn] * x[qn];
        out = acc + w[H];
    } else {
        constexpr int DD = D0 > 0 ? D0 : 1;
        float y[DD];
        const float* __restrict__ b0 = w + DD * H;
#pragma unroll
        for (int j = 0; j < DD; ++j) {
            float acc = 0.0f;
#pragma unroll
            for (int qn = 0; qn < H; ++qn) acc += w[j * H + qn] * x[qn];
            y[j] = net_act(acc + b0[j], act);
        }
        const float* __restrict__ W1 = b0 + DD;
        float acc = 0.0f;
#pragma unroll
        for (int qn = 0; qn < DD; ++qn) acc += W1[qn] * y[qn];
        out = acc + W1[DD];
    }
    computed[i] = out;
    applied[i] = dc_motor_clip(out, vel, saturation, elim[i], vlim[i]);
}

// The same shapes with EIGHT lanes per sample, lane k = hidden unit k: a lane computes the four gates of ITS unit (4 x (in + 8)
// multiply-adds + 5 transcendentals per layer) and the group shares the new hidden vector through lane shuffles (ds_bpermute: no LDS
// allocation, no barrier).  One lane carrying a whole sample (k_actuator_net_lstm_reg) is ~5000 dependent instructions on 768 waves --
// less than one wave per SIMD, nothing to hide a stall behind: 22 us; eight lanes per sample are ~600 instructions on 6144 waves.
// State loads / stores become perfectly coalesced (lane = consecutive float of the (num_layers, N*J, 8) tensors).  The weights sit in
// LDS (4.4 KB, lanes of a group read 8 different rows).  Same per-gate accumulation order as the other two kernels; the head's final
// sum over the group is a fixed-shape shuffle tree instead of a left-to-right loop (<= 1e-7 relative; tests: 1e-5 against torch).
template <int D0>
__global__ void __launch_bounds__(256)
k_actuator_net_lstm_lanes(int64_t n, int num_lstm, int act, int num_weights, const float* __restrict__ weights, const float* __restrict__ q_des,
                          const float* __restrict__ q, const float* __restrict__ qd, float* __restrict__ hid, float* __restrict__ cell,
                          float saturation, const float* __restrict__ elim, const float* __restrict__ vlim, float* __restrict__ computed,
                          float* __restrict__ applied) {
    constexpr int H = 8;
    extern __shared__ float s_w[];
    for (int k = threadIdx.x; k < num_weights; k += blockDim.x) s_w[k] = weights[k];
    __syncthreads();
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t i0 = gid >> 3;            // sample = (env, joint)
    const int k = (int)(gid & 7);           // hidden unit of this lane
    const bool live = i0 < n;
    const int64_t i = live ? i0 : n - 1;    // (whole groups are live or dead: n x 8 lanes, 8 | 64)
    const int lane = threadIdx.x & 63, base = lane & ~7;
    const float vel = qd[i];
    float x[H];
    x[0] = q_des[i] - q[i];
    x[1] = vel;
    const float* __restrict__ w = s_w;
    for (int l = 0; l < num_lstm; ++l) {  // (uniform)
        const int in_dim = l == 0 ? 2 : H;
        const float* __restrict__ W_ih = w;
        const float* __restrict__ W_hh = W_ih + 4 * H * in_dim;
        const float* __restrict__ b_ih = W_hh + 4 * H * H;
        const float* __restrict__ b_hh = b_ih + 4 * H;
        const size_t at = ((size_t)l * n + i) * H + k;
        const float h_own = hid[at], c_own = cell[at];
        float h[H];
#pragma unroll
        for (int qn = 0; qn < H; ++qn) h[qn] = __shfl(h_own, base + qn, 64);
        float g4[4];
#pragma unroll
        for (int gI = 0; gI < 4; ++gI) {
            const int r = gI * H + k;
            float a = 0.0f, b = 0.0f;
            if (l == 0) {
#pragma unroll
                for (int qn = 0; qn < 2; ++qn) a += W_ih[r * 2 + qn] * x[qn];
            } else {
#pragma unroll
                for (int qn = 0; qn < H; ++qn) a += W_ih[r * H + qn] * x[qn];
            }
#pragma unroll
            for (int qn = 0; qn < H; ++qn) b += W_hh[r * H + qn] * h[qn];
            g4[gI] = (a + b_ih[r]) + (b + b_hh[r]);
        }
        const float c_new = net_sigmoid(g4[1]) * c_own + net_sigmoid(g4[0]) * tanhf(g4[2]);
        const float h_new = net_sigmoid(g4[3]) * tanhf(c_new);
        if (live) { cell[at] = c_new; hid[at] = h_new; }
#pragma unroll
        for (int qn = 0; qn < H; ++qn) x[qn] = __shfl(h_new, base + qn, 64);
        w = b_hh + 4 * H;
    }
    // the head: H -> 1 (D0 == 0) or H -> D0 -> 1; lane k takes the k-th share of the sum
    float part;
    if (D0 == 0) {
        part = w[k] * x[k];
    } else {
        constexpr int DD = D0 > 0 ? D0 : 8;
        const float* __restrict__ b0 = w + DD * H;
        const float* __restrict__ W1 = b0 + DD;
        part = 0.0f;
#pragma unroll
        for (int jj = 0; jj < DD / H; ++jj) {  // outputs k, k + 8, ...
            const int j = k + H * jj;
            float acc = 0.0f;
#pragma unroll
            for (int qn = 0; qn < H; ++qn) acc += w[j * H + qn] * x[qn];
            part += W1[j] * net_act(acc + b0[j], act);
        }
    }
    part += __shfl_xor(part, 1, 64);
    part += __shfl_xor(part, 2, 64);
    part += __shfl_xor(part, 4, 64);
    const float bias_out = D0 == 0 ? w[H] : (w + (D0 > 0 ? D0 : 8) * H + (D0 > 0 ? D0 : 8))[D0 > 0 ? D0 : 8];
    const float out = part + bias_out;
    if (live && k == 0) {
        computed[i] = out;
        applied[i] = dc_motor_clip(out, vel, saturation, elim[i], vlim[i]);
    }
}

// The same shapes on the matrix core: a wave owns 32 samples and forms the 32 gate pre-activations of a layer as
//   G^T (32 gates x 32 samples) = W (32 x k) . [x ; h]^T (k x 32 samples),   v_mfma_f32_32x32x2_f32, one step per pair of inputs.
// Two choices make everything after the MFMAs lane-local (no shuffle, no LDS traffic per sample):
//   * gate rows are ORDERED so that accumulator register q of lane-half hf is gate type q >> 2 of hidden unit 4 hf + (q & 3): lane
//     (sample, hf) ends up with i, f, g, o of units 4 hf .. 4 hf + 3 of ITS sample -- the cell update is 4 independent scalars per lane,
//     and the state it needs / writes is one float4 of the (layers, n, 8) tensors;
//   * the reduction index is PAIRED as (unit j, unit 4 + j) per step, so the lane half that owns units 4 hf .. supplies exactly its own
//     new hidden values as the B operand of the next layer.
// Weights: every lane keeps the A-operand values of its gate row in registers (13 per layer pair of the ANYdrive net) plus 16 bias sums.
// 8 lanes per sample (k_actuator_net_lstm_lanes) spent its time in 128 LDS-fed multiply-adds and 32 shuffles per lane and layer: 13.7 us;
// here a layer is 5 or 8 MFMAs per 32 samples.  Summation order differs from torch's (pairs of inputs per step): <= 1e-6 relative.
typedef float prod_f32x16 __attribute__((ext_vector_type(16)));

// sigmoid / tanh through v_exp_f32 and v_rcp_f32 (absolute error ~1e-7: inside the 1e-5 of the actuator tests; expf / tanhf are
// 20-40 instructions each and a lane of the MFMA kernel evaluates 40 of them)
IMX_DEV float fast_sigmoid(float x) { return __frcp_rn(1.0f + __expf(-x)); }
IMX_DEV float fast_tanh(float x) { return 2.0f * fast_sigmoid(2.0f * x) - 1.0f; }

template <int D0>
__global__ void __launch_bounds__(256)
k_actuator_net_lstm_mfma(int64_t n, int num_lstm, int act, int num_weights, const float* __restrict__ weights, const float* __restrict__ q_des,
                         const float* __restrict__ q, const float* __restrict__ qd, float* __restrict__ hid, float* __restrict__ cell,
                         float saturation, const float* __restrict__ elim, const float* __restrict__ vlim, float* __restrict__ computed,
                         float* __restrict__ applied) {
    constexpr int H = 8, MAXL = 4;
    extern __shared__ float s_w[];
    const int lane = threadIdx.x & 63, r = lane & 31, hf = lane >> 5;
    const int64_t i0 = ((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 32 + r;  // this lane's sample
    const bool live = i0 < n;
    const int64_t i = live ? i0 : n - 1;
    // inputs and the state of EVERY layer are requested first, the weights copied to LDS behind them: one memory round trip for all of it
    const float vel = qd[i];
    const float perr = q_des[i] - q[i];
    float4 hall[MAXL], call[MAXL];
#pragma unroll
    for (int l = 0; l < MAXL; ++l) {
        const size_t at = ((size_t)(l < num_lstm ? l : 0) * n + i) * H + 4 * hf;
        hall[l] = *reinterpret_cast<const float4*>(hid + at);
        call[l] = *reinterpret_cast<const float4*>(cell + at);
    }
    for (int k = threadIdx.x; k < num_weights; k += blockDim.x) s_w[k] = weights[k];
    __syncthreads();
    // A-operand row of this lane: accumulator row rho = r  <->  gate type rho >> 3 of unit (rho & 3) + 4 ((rho >> 2) & 1)  ->  weight row
    const int wr = (r >> 3) * H + (r & 3) + 4 * ((r >> 2) & 1);
    float xin[4];
    xin[0] = hf ? vel : perr;
    xin[1] = xin[2] = xin[3] = 0.0f;
    const float* __restrict__ w = s_w;
    float hn[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int l = 0; l < MAXL; ++l) {
        if (l >= num_lstm) break;  // (uniform)
        const int in_dim = l == 0 ? 2 : H;
        const float* __restrict__ W_ih = w;
        const float* __restrict__ W_hh = W_ih + 4 * H * in_dim;
        const float* __restrict__ b_ih = W_hh + 4 * H * H;
        const float* __restrict__ b_hh = b_ih + 4 * H;
        const size_t at = ((size_t)l * n + i) * H + 4 * hf;
        const float4 hv = hall[l], cv = call[l];
        prod_f32x16 acc;
#pragma unroll
        for (int qq = 0; qq < 16; ++qq) acc[qq] = 0.0f;
        if (l == 0) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(W_ih[wr * 2 + hf], xin[0], acc, 0, 0, 0);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(W_ih[wr * H + j + 4 * hf], xin[j], acc, 0, 0, 0);
        }
        const float hvv[4] = {hv.x, hv.y, hv.z, hv.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(W_hh[wr * H + j + 4 * hf], hvv[j], acc, 0, 0, 0);
        const float cvv[4] = {cv.x, cv.y, cv.z, cv.w};
        float cn[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int unit = 4 * hf + u;
            const float gi = (acc[u] + b_ih[unit]) + b_hh[unit];
            const float gf = (acc[4 + u] + b_ih[H + unit]) + b_hh[H + unit];
            const float gg = (acc[8 + u] + b_ih[2 * H + unit]) + b_hh[2 * H + unit];
            const float go = (acc[12 + u] + b_ih[3 * H + unit]) + b_hh[3 * H + unit];
            cn[u] = fast_sigmoid(gf) * cvv[u] + fast_sigmoid(gi) * fast_tanh(gg);
            hn[u] = fast_sigmoid(go) * fast_tanh(cn[u]);
        }
        if (live) {
            *reinterpret_cast<float4*>(cell + at) = make_float4(cn[0], cn[1], cn[2], cn[3]);
            *reinterpret_cast<float4*>(hid + at) = make_float4(hn[0], hn[1], hn[2], hn[3]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) xin[u] = hn[u];
        w = b_hh + 4 * H;
    }
    // the head: H -> 1 (D0 == 0) or H -> D0 -> 1; each lane half sums over what it holds, one shuffle joins the halves
    float part = 0.0f, bias_out;
    if (D0 == 0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) part += w[4 * hf + u] * hn[u];
        bias_out = w[H];
    } else {
        constexpr int DD = D0 > 0 ? D0 : 32;
        const float* __restrict__ b0 = w + DD * H;
        const float* __restrict__ W1 = b0 + DD;
        prod_f32x16 acc;
#pragma unroll
        for (int qq = 0; qq < 16; ++qq) acc[qq] = 0.0f;
        const int rowd = r < DD ? r : 0;  // accumulator row rho = dense output rho (rows >= D0 multiply by zero)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(r < DD ? w[rowd * H + j + 4 * hf] : 0.0f, hn[j], acc, 0, 0, 0);
#pragma unroll
        for (int qq = 0; qq < 16; ++qq) {
            const int rho = (qq & 3) + 8 * (qq >> 2) + 4 * hf;
            if (rho < DD) part += W1[rho] * net_act(acc[qq] + b0[rho], act);
        }
        bias_out = W1[DD];
    }
    part += __shfl_xor(part, 32, 64);
    const float out = part + bias_out;
    if (live && hf == 0) {
        computed[i] = out;
        applied[i] = dc_motor_clip(out, vel, saturation, elim[i], vlim[i]);
    }
}

// ActuatorNetMLP: the (N, history, J) queues of position error and velocity are rolled by one and topped up (:164-170); the inputs of
// sample (env, joint) are the entries `input_idx` of both queues, scaled, position block first or second (:172-188).
__global__ void __launch_bounds__(64)
k_actuator_net_mlp(int64_t N, int J, ImxNetDesc D, const float* __restrict__ weights, int hist, const int32_t* __restrict__ input_idx,
                   int num_idx, float pos_scale, float vel_scale, float torque_scale, int vel_first, const float* __restrict__ q_des,
                   const float* __restrict__ q, const float* __restrict__ qd, float* __restrict__ pos_hist, float* __restrict__ vel_hist,
                   float saturation, const float* __restrict__ elim, const float* __restrict__ vlim, float* __restrict__ computed,
                   float* __restrict__ applied) {
    extern __shared__ float s_net[];
    float* s_w = s_net;
    float* va = s_net + ((D.num_weights + 3) & ~3);
    float* vb = va + IMX_NET_MAXW * 64;
    const int lane = threadIdx.x;
    for (int k = lane; k < D.num_weights; k += 64) s_w[k] = weights[k];
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * 64 + lane;
    if (i >= N * J) return;
    const int64_t e = i / J;
    const int j = (int)(i - e * J);
    const float vel = qd[i];
    float* ph = pos_hist + (size_t)e * hist * J + j;  // (N, history, J): entry h of this joint at ph[h * J]
    float* vh = vel_hist + (size_t)e * hist * J + j;
    for (int h = hist - 1; h >= 1; --h) { ph[(size_t)h * J] = ph[(size_t)(h - 1) * J]; vh[(size_t)h * J] = vh[(size_t)(h - 1) * J]; }
    ph[0] = q_des[i] - q[i];
    vh[0] = vel;
    for (int k = 0; k < num_idx; ++k) {
        const int h = input_idx[k];
        const float p = ph[(size_t)h * J] * pos_scale, v = vh[(size_t)h * J] * vel_scale;
        va[((vel_first ? num_idx : 0) + k) * 64 + lane] = p;
        va[((vel_first ? 0 : num_idx) + k) * 64 + lane] = v;
    }
    net_dense(D, s_w, 2 * num_idx, va, vb, lane);
    const float c = va[lane] * torque_scale;
    computed[i] = c;
    applied[i] = dc_motor_clip(c, vel, saturation, elim[i], vlim[i]);
}

static int check_net(const char* who, int num_lstm, int hidden, int num_dense, const int32_t* dense_out, int act, int first_in,
                     int64_t num_weights, ImxNetDesc& D) {
    IMX_REQUIRE(num_lstm >= 0 && num_lstm <= 8 && num_dense >= 1 && num_dense <= 4 && act >= 0 && act <= 4, "%s: unsupported network (LSTM layers %d, dense layers %d, activation %d)", who, num_lstm, num_dense, act);
    IMX_REQUIRE(num_lstm == 0 || (hidden >= 1 && hidden <= IMX_NET_MAXW), "%s: hidden size %d (1..%d)", who, hidden, IMX_NET_MAXW);
    int64_t need = 0;
    int in_dim = first_in;
    for (int l = 0; l < num_lstm; ++l) { need += (int64_t)4 * hidden * (in_dim + hidden) + 8 * hidden; in_dim = hidden; }
    IMX_REQUIRE(in_dim >= 1 && in_dim <= IMX_NET_MAXW, "%s: input width %d (1..%d)", who, in_dim, IMX_NET_MAXW);
    for (int k = 0; k < num_dense; ++k) {
        IMX_REQUIRE(dense_out && dense_out[k] >= 1 && dense_out[k] <= IMX_NET_MAXW, "%s: dense layer %d width out of range (1..%d)", who, k, IMX_NET_MAXW);
        need += (int64_t)dense_out[k] * in_dim + dense_out[k];
        in_dim = dense_out[k];
        D.dense_out[k] = dense_out[k];
    }
    IMX_REQUIRE(in_dim == 1, "%s: the last dense layer must have one output (the torque), not %d", who, in_dim);
    IMX_REQUIRE(need == num_weights, "%s: the packed network holds %lld floats, the layer sizes need %lld", who, (long long)num_weights, (long long)need);
    IMX_REQUIRE((size_t)(((num_weights + 3) & ~3) + 3 * IMX_NET_MAXW * 64) * sizeof(float) <= 160 * 1024, "%s: network too large for one workgroup's LDS (%lld weights)", who, (long long)num_weights);
    D.num_lstm = num_lstm; D.hidden = hidden; D.num_dense = num_dense; D.act = act; D.num_weights = (int)num_weights;
    return 0;
}

extern "C" int imx_actuator_net_lstm(int64_t N, int64_t J, int num_lstm, int hidden, int num_dense, const int32_t* dense_out_h, int act,
                                     const float* weights_d, int64_t num_weights, const float* joint_pos_target_d, const float* joint_pos_d,
                                     const float* joint_vel_d, float* hidden_state_d, float* cell_state_d, float saturation_effort,
                                     const float* effort_limit_d, const float* velocity_limit_d, float* computed_effort_d,
                                     float* applied_effort_d, imx_stream_t stream) {
    IMX_REQUIRE(N > 0 && J > 0 && weights_d && joint_pos_target_d && joint_pos_d && joint_vel_d && hidden_state_d && cell_state_d &&
                    effort_limit_d && velocity_limit_d && computed_effort_d && applied_effort_d, "imx_actuator_net_lstm: bad arguments");
    IMX_REQUIRE(num_lstm >= 1, "imx_actuator_net_lstm: no LSTM layer");
    ImxNetDesc D{};
    if (check_net("imx_actuator_net_lstm", num_lstm, hidden, num_dense, dense_out_h, act, 2, num_weights, D)) return 1;
    const int64_t n = N * J;
    // the ANYdrive shapes run from registers (k_actuator_net_lstm_reg); anything else through the generic LDS kernel
    if (hidden == 8 && num_lstm <= 4 && (reinterpret_cast<uintptr_t>(hidden_state_d) & 15) == 0 && (reinterpret_cast<uintptr_t>(cell_state_d) & 15) == 0 &&
        ((num_dense == 1) || (num_dense == 2 && (dense_out_h[0] == 16 || dense_out_h[0] == 32)))) {
        static const char* const which = getenv("IMX_LSTM_KERNEL");  // read once.  default: matrix core; "l": eight lanes per sample; "r": one lane per sample
        if (which == nullptr || (which[0] != 'r' && which[0] != 'l')) {
            const unsigned gm = (unsigned)((n + 127) / 128);  // 4 waves x 32 samples per workgroup
            const size_t ldsm = (size_t)num_weights * sizeof(float);
#define IMX_LSTM_MFMA(D0)                                                                                                                 \
    hipLaunchKernelGGL((k_actuator_net_lstm_mfma<D0>), dim3(gm), dim3(256), ldsm, (hipStream_t)stream, n, num_lstm, act, (int)num_weights, \
                       weights_d, joint_pos_target_d, joint_pos_d, joint_vel_d, hidden_state_d, cell_state_d, saturation_effort,         \
                       effort_limit_d, velocity_limit_d, computed_effort_d, applied_effort_d)
            if (num_dense == 1) IMX_LSTM_MFMA(0);
            else if (dense_out_h[0] == 16) IMX_LSTM_MFMA(16);
            else IMX_LSTM_MFMA(32);
#undef IMX_LSTM_MFMA
            IMX_HIP(hipGetLastError());
            return 0;
        }
        if (which[0] != 'r') {  // eight lanes per sample
            const unsigned g8 = (unsigned)((n * 8 + 255) / 256);
            const size_t lds8 = (size_t)num_weights * sizeof(float);
#define IMX_LSTM_LANES(D0)                                                                                                                \
    hipLaunchKernelGGL((k_actuator_net_lstm_lanes<D0>), dim3(g8), dim3(256), lds8, (hipStream_t)stream, n, num_lstm, act, (int)num_weights, \
                       weights_d, joint_pos_target_d, joint_pos_d, joint_vel_d, hidden_state_d, cell_state_d, saturation_effort,         \
                       effort_limit_d, velocity_limit_d, computed_effort_d, applied_effort_d)
            if (num_dense == 1) IMX_LSTM_LANES(0);
            else if (dense_out_h[0] == 16) IMX_LSTM_LANES(16);
            else IMX_LSTM_LANES(32);
#undef IMX_LSTM_LANES
            IMX_HIP(hipGetLastError());
            return 0;
        }
        const unsigned grid = (unsigned)((n + 255) / 256);
#define IMX_LSTM_REG(D0)                                                                                                                  \
    hipLaunchKernelGGL((k_actuator_net_lstm_reg<8, D0>), dim3(grid), dim3(256), 0, (hipStream_t)stream, n, num_lstm, act, weights_d,      \
                       joint_pos_target_d, joint_pos_d, joint_vel_d, hidden_state_d, cell_state_d, saturation_effort, effort_limit_d,   \
                       velocity_limit_d, computed_effort_d, applied_effort_d)
        if (num_dense == 1) IMX_LSTM_REG(0);
        else if (dense_out_h[0] == 16) IMX_LSTM_REG(16);
        else IMX_LSTM_REG(32);
#undef IMX_LSTM_REG
        IMX_HIP(hipGetLastError());
        return 0;
    }
    const size_t lds = (size_t)(((num_weights + 3) & ~3) + 3 * IMX_NET_MAXW * 64) * sizeof(float);
    IMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_actuator_net_lstm), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_actuator_net_lstm, dim3((unsigned)((n + 63) / 64)), dim3(64), lds, (hipStream_t)stream, n, D, weights_d,
                       joint_pos_target_d, joint_pos_d, joint_vel_d, hidden_state_d, cell_state_d, saturation_effort, effort_limit_d,
                       velocity_limit_d, computed_effort_d, applied_effort_d);
    IMX_HIP(hipGetLastError());
    return 0;
}

extern "C" int imx_actuator_net_mlp(int64_t N, int64_t J, int num_dense, const int32_t* dense_out_h, int act, const float* weights_d,
                                    int64_t num_weights, int history_length, const int32_t* input_idx_d, int num_idx, float pos_scale,
                                    float vel_scale, float torque_scale, int vel_first, const float* joint_pos_target_d,
                                    const float* joint_pos_d, const float* joint_vel_d, float* pos_error_history_d, float* vel_history_d,
                                    float saturation_effort, const float* effort_limit_d, const float* velocity_limit_d,
                                    float* computed_effort_d, float* applied_effort_d, imx_stream_t stream) {
    IMX_REQUIRE(N > 0 && J > 0 && weights_d && input_idx_d && joint_pos_target_d && joint_pos_d && joint_vel_d && pos_error_history_d &&
                    vel_history_d && effort_limit_d && velocity_limit_d && computed_effort_d && applied_effort_d, "imx_actuator_net_mlp: bad arguments");
    IMX_REQUIRE(history_length >= 1 && num_idx >= 1 && 2 * num_idx <= IMX_NET_MAXW, "imx_actuator_net_mlp: history %d / %d input indices", history_length, num_idx);
    ImxNetDesc D{};
    if (check_net("imx_actuator_net_mlp", 0, 0, num_dense, dense_out_h, act, 2 * num_idx, num_weights, D)) return 1;
    const int64_t n = N * J;
    const size_t lds = (size_t)(((num_weights + 3) & ~3) + 2 * IMX_NET_MAXW * 64) * sizeof(float);
    IMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_actuator_net_mlp), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_actuator_net_mlp, dim3((unsigned)((n + 63) / 64)), dim3(64), lds, (hipStream_t)stream, N, (int)J, D, weights_d,
                       history_length, input_idx_d, num_idx, pos_scale, vel_scale, torque_scale, vel_first, joint_pos_target_d, joint_pos_d,
                       joint_vel_d, pos_error_history_d, vel_history_d, saturation_effort, effort_limit_d, velocity_limit_d,
                       computed_effort_d, applied_effort_d);
    IMX_HIP(hipGetLastError());
    return 0;
}
