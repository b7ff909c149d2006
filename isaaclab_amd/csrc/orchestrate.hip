// The reset / interval orchestration around the env step as ONE masked kernel (include/imx.h imx_orch_t).
//
// Reference, per step (isaaclab/isaaclab/envs/manager_based_rl_env.py:215-236):
//     reset_env_ids = reset_buf.nonzero()                      # host sync
//     if len(reset_env_ids) > 0: _reset_idx(reset_env_ids)     # :347-392
//         curriculum_manager.compute(env_ids)                  # terrain_levels_vel -> TerrainImporter.update_env_origins
//         scene.reset(env_ids)                                 # ContactSensor.reset, actuator reset, (height scanner: k_term_rew)
//         event_manager.apply("reset", env_ids, step count)    # managers/event_manager.py:233-260 (min_step_count_between_reset)
//         ... manager resets -> extras["log"]                  # command_manager.reset: Metrics/*; curriculum_manager.reset: Curriculum/*
//     command_manager.compute(dt)                              # managers/command_manager.py:151-187
//     event_manager.apply("interval", dt)                      # event_manager.py:205-232: nonzero() + len() per term
// Here: one lane per env walks that list for its env; what the reference decides from id lists it decides from the env's reset flag and
// its timers.  Random draws come from the counter-based generator keyed by (seed, term, column, global step, env) or, in parity runs,
// from caller-supplied tables (the reference's torch.rand stream cannot be reproduced in a kernel).  The two means the log needs
// (metrics over the reset envs, terrain level over all envs) leave as per-workgroup partial sums for the step tail (step.hip).
#include "imx_internal.h"
#include "imx_producers.h"

namespace {

constexpr int ORCH_BLOCK = 64;

__device__ __forceinline__ float draw(const float* __restrict__ U, int64_t stride, int64_t e, int col, uint64_t seed, int term, uint32_t step) {
    return U ? U[e * stride + col] : uniform01(seed + 0x9E3779B97F4A7C15ull * (uint64_t)(term * 256 + col + 1), step, (uint64_t)e);
}

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ void __launch_bounds__(ORCH_BLOCK) k_reset_orchestrate(imx_orch_t o) {
    const int lane = threadIdx.x;
    const int64_t N = o.num_envs;
    const int64_t e0 = (int64_t)blockIdx.x * ORCH_BLOCK + lane;
    const bool live = e0 < N;
    const int64_t e = live ? e0 : N - 1;  // dead lanes read a valid env, never store
    const int J = (int)o.num_joints, NB = (int)o.num_bodies;
    const uint32_t step = o.step_counter_d ? (uint32_t)o.step_counter_d[0] : 0u;
    const bool reset = live && (o.reset_mask_d ? o.reset_mask_d[e] != 0 : true);

    // ---- CurriculumManager.compute: terrain_levels_vel (curriculums.py:26-55) + update_env_origins (terrain_importer.py:307-326)
    float ox = o.env_origins_d[e * 3], oy = o.env_origins_d[e * 3 + 1], oz = o.env_origins_d[e * 3 + 2];
    float level_f = 0.0f;
    if (o.terrain_levels_d) {
        int64_t lv = o.terrain_levels_d[e];
        if (reset) {
            const float dx = o.root_pos_w_d[e * 3] - ox, dy = o.root_pos_w_d[e * 3 + 1] - oy;
            const float dist = sqrtf(dx * dx + dy * dy);  // torch.norm(dim=1)
            const float cx = o.vel_command_b_d[e * 3], cy = o.vel_command_b_d[e * 3 + 1];  // the command BEFORE its reset (:379-380 comes later)
            const bool up = dist > 0.5f * o.terrain_size_x;
            const bool down = (dist < sqrtf(cx * cx + cy * cy) * o.max_episode_length_s * 0.5f) && !up;
            lv = lv + (up ? 1 : 0) - (down ? 1 : 0);
            if (lv >= o.terrain_rows) {  // solved the last level: a random one (randint_like(levels, max_terrain_level))
                lv = o.rand_levels_d ? o.rand_levels_d[e] : (int64_t)(uniform01(o.seed + 991u, step, (uint64_t)e) * (float)o.terrain_rows);
                lv = lv >= o.terrain_rows ? o.terrain_rows - 1 : lv;
            } else if (lv < 0) {
                lv = 0;
            }
            o.terrain_levels_d[e] = lv;
            const float* t = o.terrain_origins_d + ((size_t)lv * o.terrain_cols + (size_t)o.terrain_types_d[e]) * 3;
            ox = t[0]; oy = t[1]; oz = t[2];
            o.env_origins_d[e * 3] = ox; o.env_origins_d[e * 3 + 1] = oy; o.env_origins_d[e * 3 + 2] = oz;
        }
        level_f = live ? (float)lv : 0.0f;
    }

    // ---- scene.reset(env_ids): the env-owned sensors / actuators of a reset env start over.  The WAVE zeroes the rows of each of its
    //      reset envs together (lanes stride over the row: coalesced stores, ~10 instructions per env) -- one lane walking its own env's
    //      272 + 384 floats was 650 store instructions in a divergent branch that 7 of 10 waves enter (25 us for the kernel).
    const uint64_t reset_lanes = __ballot(reset);
    if (o.cs_timestamp_d || o.lstm_hidden_d) {
        for (uint64_t m = reset_lanes; m != 0ull; m &= m - 1ull) {
            const int64_t er = (int64_t)blockIdx.x * ORCH_BLOCK + (__ffsll((long long)m) - 1);  // wave-uniform
            if (o.cs_timestamp_d) {  // ContactSensor.reset (contact_sensor.py:143-165) + SensorBase.reset (sensor_base.py:182-194)
                const int B = o.cs_num_bodies, H = o.cs_history_length;
                if (lane == 0) { o.cs_timestamp_d[er] = 0.0f; o.cs_timestamp_last_update_d[er] = 0.0f; o.cs_is_outdated_d[er] = 1; }
                for (int i = lane; i < B * 3; i += ORCH_BLOCK) o.cs_net_forces_w_d[(size_t)er * B * 3 + i] = 0.0f;
                if (o.cs_net_forces_w_history_d)
                    for (int i = lane; i < H * B * 3; i += ORCH_BLOCK) o.cs_net_forces_w_history_d[(size_t)er * H * B * 3 + i] = 0.0f;
                if (o.cs_last_air_time_d)
                    for (int b = lane; b < B; b += ORCH_BLOCK) {
                        o.cs_last_air_time_d[(size_t)er * B + b] = 0.0f; o.cs_current_air_time_d[(size_t)er * B + b] = 0.0f;
                        o.cs_last_contact_time_d[(size_t)er * B + b] = 0.0f; o.cs_current_contact_time_d[(size_t)er * B + b] = 0.0f;
                    }
            }
            if (o.lstm_hidden_d) {  // ActuatorNetLSTM.reset (actuators/actuator_net.py:66-70): hidden / cell state of the env's joints
                const int Hd = o.lstm_hidden_dim;
                for (int l = 0; l < o.lstm_layers; ++l)
                    for (int i = lane; i < J * Hd; i += ORCH_BLOCK) {
                        const size_t at = ((size_t)l * N * J + (size_t)er * J) * Hd + i;
                        o.lstm_hidden_d[at] = 0.0f;
                        o.lstm_cell_d[at] = 0.0f;
                    }
            }
        }
    }

    // ---- EventManager.apply("reset", env_ids, global_env_step_count) (event_manager.py:233-260), terms in cfg order
    for (int t = 0; t < o.num_terms; ++t) {
        const imx_event_term_t& T = o.terms[t];
        if (T.mode != 0) continue;
        bool valid = reset;
        if (reset) {
            if (T.min_step_count_between_reset == 0) {
                T.last_triggered_step_d[e] = (int32_t)step;
                T.triggered_once_d[e] = 1;
            } else {
                const int32_t last = T.last_triggered_step_d[e];
                const bool once = T.triggered_once_d[e] != 0;
                valid = ((int32_t)step - last >= T.min_step_count_between_reset) || (last == 0 && !once);
                if (valid) {
                    T.triggered_once_d[e] = 1;
                    T.last_triggered_step_d[e] = (int32_t)step;
                }
            }
        }
        const float* __restrict__ U = T.uniforms_d;
        if (T.op == IMX_E_RESET_JOINTS_BY_SCALE || T.op == IMX_E_RESET_JOINTS_BY_OFFSET) {  // events.py:987-1049
            // lane = joint of one valid env at a time (the same draws, keyed by (env, column)): 2 J samples + clamps per env spread over
            // the wave instead of a J-trip loop on the env's own lane
            const bool by_offset = T.op == IMX_E_RESET_JOINTS_BY_OFFSET;
            for (uint64_t m = __ballot(valid); m != 0ull; m &= m - 1ull) {
                const int64_t er = (int64_t)blockIdx.x * ORCH_BLOCK + (__ffsll((long long)m) - 1);  // wave-uniform
                for (int j = lane; j < J; j += ORCH_BLOCK) {
                    const size_t q = (size_t)er * J + j;
                    const float sp = draw(U, 2 * (int64_t)J, er, j, o.seed, t, step) * (T.ranges[1] - T.ranges[0]) + T.ranges[0];
                    const float sv = draw(U, 2 * (int64_t)J, er, J + j, o.seed, t, step) * (T.ranges[3] - T.ranges[2]) + T.ranges[2];
                    float p = by_offset ? o.default_joint_pos_d[q] + sp : o.default_joint_pos_d[q] * sp;
                    float v = by_offset ? o.default_joint_vel_d[q] + sv : o.default_joint_vel_d[q] * sv;
                    p = fminf(fmaxf(p, o.soft_joint_pos_limits_d[2 * q]), o.soft_joint_pos_limits_d[2 * q + 1]);  // clamp_(lo, hi)
                    v = fminf(fmaxf(v, -o.soft_joint_vel_limits_d[q]), o.soft_joint_vel_limits_d[q]);
                    o.joint_pos_out_d[q] = p;
                    o.joint_vel_out_d[q] = v;
                }
            }
            continue;
        }
        if (!valid) continue;
        switch (T.op) {
            case IMX_E_RESET_ROOT_STATE_UNIFORM: {  // events.py:823-868
                const float* d = o.default_root_state_d + e * 13;
                float rs[6];
#pragma unroll
                for (int k = 0; k < 6; ++k) rs[k] = draw(U, 12, e, k, o.seed, t, step) * (T.ranges[2 * k + 1] - T.ranges[2 * k]) + T.ranges[2 * k];
                float* pose = o.root_pose_out_d + e * 7;
                pose[0] = d[0] + ox + rs[0];  // positions = default + env origin + sample (:852)
                pose[1] = d[1] + oy + rs[1];
                pose[2] = d[2] + oz + rs[2];
                // quat_from_euler_xyz(roll, pitch, yaw) (math.py:266-276), then quat_mul(default, delta) (math.py:486-497)
                const float cy = cosf(rs[5] * 0.5f), sy = sinf(rs[5] * 0.5f), cr = cosf(rs[3] * 0.5f), sr = sinf(rs[3] * 0.5f);
                const float cp = cosf(rs[4] * 0.5f), sp = sinf(rs[4] * 0.5f);
                const float w2 = cy * cr * cp + sy * sr * sp, x2 = cy * sr * cp - sy * cr * sp, y2 = cy * cr * sp + sy * sr * cp,
                            z2 = sy * cr * cp - cy * sr * sp;
                const float w1 = d[3], x1 = d[4], y1 = d[5], z1 = d[6];
                const float ww = (z1 + x1) * (x2 + y2), yy = (w1 - y1) * (w2 + z2), zz = (w1 + y1) * (w2 - z2);
                const float xx = ww + yy + zz;
                const float qq = 0.5f * (xx + (z1 - x1) * (x2 - y2));
                pose[3] = qq - ww + (z1 - y1) * (y2 - z2);
                pose[4] = qq - xx + (x1 + w1) * (x2 + w2);
                pose[5] = qq - yy + (w1 - x1) * (y2 + z2);
                pose[6] = qq - zz + (z1 + y1) * (w2 - x2);
#pragma unroll
                for (int k = 0; k < 6; ++k)
                    o.root_vel_out_d[e * 6 + k] =
                        d[7 + k] + (draw(U, 12, e, 6 + k, o.seed, t, step) * (T.ranges[12 + 2 * k + 1] - T.ranges[12 + 2 * k]) + T.ranges[12 + 2 * k]);
            } break;
            case IMX_E_RESET_JOINTS_BY_SCALE:
            case IMX_E_RESET_JOINTS_BY_OFFSET: break;  // (worked off by the whole wave, below)
            case IMX_E_APPLY_EXTERNAL_FORCE_TORQUE: {  // events.py:764-791: two sample_uniform calls (forces, then torques) of (k, nb, 3)
                const int nb = T.body_ids_d ? T.num_body_ids : NB;
                for (int b = 0; b < nb; ++b) {
                    const int body = T.body_ids_d ? T.body_ids_d[b] : b;
                    for (int c = 0; c < 3; ++c) {
                        const float uf = draw(U, 6 * (int64_t)nb, e, b * 3 + c, o.seed, t, step);
                        const float ut = draw(U, 6 * (int64_t)nb, e, nb * 3 + b * 3 + c, o.seed, t, step);
                        const size_t at = ((size_t)e * NB + body) * 3 + c;
                        o.ext_force_out_d[at] = uf * (T.ranges[1] - T.ranges[0]) + T.ranges[0];
                        o.ext_torque_out_d[at] = ut * (T.ranges[3] - T.ranges[2]) + T.ranges[2];
                    }
                }
            } break;
            case IMX_E_PUSH_BY_SETTING_VELOCITY: {  // a push configured as a reset event
                const float v6[6] = {o.root_lin_vel_w_d[e * 3], o.root_lin_vel_w_d[e * 3 + 1], o.root_lin_vel_w_d[e * 3 + 2],
                                     o.root_ang_vel_w_d[e * 3], o.root_ang_vel_w_d[e * 3 + 1], o.root_ang_vel_w_d[e * 3 + 2]};
#pragma unroll
                for (int k = 0; k < 6; ++k)
                    o.root_vel_out_d[e * 6 + k] = v6[k] + (draw(U, 6, e, k, o.seed, t, step) * (T.ranges[2 * k + 1] - T.ranges[2 * k]) + T.ranges[2 * k]);
            } break;
            default: break;
        }
    }

    // ---- CommandTerm.reset for the reset envs (logs + zeroes the metrics, resamples), then CommandManager.compute(dt)
    float mxy0 = 0.0f, myaw0 = 0.0f;
    if (o.has_command && live) {
        VelCmdCfg c;
        c.resample_lo = o.command_cfg[0]; c.resample_hi = o.command_cfg[1];
        c.lin_x_lo = o.command_cfg[2]; c.lin_x_hi = o.command_cfg[3]; c.lin_y_lo = o.command_cfg[4]; c.lin_y_hi = o.command_cfg[5];
        c.ang_z_lo = o.command_cfg[6]; c.ang_z_hi = o.command_cfg[7]; c.heading_lo = o.command_cfg[8]; c.heading_hi = o.command_cfg[9];
        c.rel_standing = o.command_cfg[10]; c.rel_heading = o.command_cfg[11]; c.stiffness = o.command_cfg[12];
        c.max_command_step = o.command_cfg[13];
        c.heading_command = o.heading_command;
        const VelCmdState st{o.vel_command_b_d, o.heading_target_d, o.is_heading_env_d, o.is_standing_env_d, o.command_time_left_d,
                             o.command_counter_d, o.metric_error_vel_xy_d, o.metric_error_vel_yaw_d};
        velocity_command_env(N, e, c, o.dt, o.do_step, o.root_quat_w_d, o.root_lin_vel_w_d, o.root_ang_vel_w_d, reset,
                             o.command_uniforms_d, o.seed ^ 0xC0FFEEull, step, st, mxy0, myaw0);
    }

    // ---- EventManager.apply("interval", dt) (event_manager.py:205-232)
    if (o.do_step) {
        for (int t = 0; t < o.num_terms; ++t) {
            const imx_event_term_t& T = o.terms[t];
            if (T.mode != 1) continue;
            bool fire;
            if (T.is_global_time) {  // one timer: slot [step & 1] -> slot [(step + 1) & 1]; a single draw; the term runs on EVERY env
                float tl = T.time_left_d[step & 1u] - o.dt;
                fire = tl < 1.0e-6f;
                if (fire) {
                    const float u = T.interval_uniforms_d ? T.interval_uniforms_d[0]
                                                          : uniform01(o.seed + 0x1717ull * (uint64_t)(t + 1), step, 0xFFFFFFFFull);
                    tl = u * (T.interval_hi - T.interval_lo) + T.interval_lo;
                }
                if (blockIdx.x == 0 && lane == 0) T.time_left_d[(step + 1u) & 1u] = tl;
            } else {
                float tl = T.time_left_d[e] - o.dt;
                fire = tl < 1.0e-6f;
                if (fire) {
                    const float u = T.interval_uniforms_d ? T.interval_uniforms_d[e]
                                                          : uniform01(o.seed + 0x1717ull * (uint64_t)(t + 1), step, (uint64_t)e);
                    tl = u * (T.interval_hi - T.interval_lo) + T.interval_lo;
                }
                if (live) T.time_left_d[e] = tl;
            }
            if (!fire || !live) continue;
            const float* __restrict__ U = T.uniforms_d;
            if (T.op == IMX_E_PUSH_BY_SETTING_VELOCITY) {  // events.py:795-820: root_vel_w + sample -> write_root_velocity_to_sim
                const float v6[6] = {o.root_lin_vel_w_d[e * 3], o.root_lin_vel_w_d[e * 3 + 1], o.root_lin_vel_w_d[e * 3 + 2],
                                     o.root_ang_vel_w_d[e * 3], o.root_ang_vel_w_d[e * 3 + 1], o.root_ang_vel_w_d[e * 3 + 2]};
#pragma unroll
                for (int k = 0; k < 6; ++k)
                    o.root_vel_out_d[e * 6 + k] = v6[k] + (draw(U, 6, e, k, o.seed, t, step) * (T.ranges[2 * k + 1] - T.ranges[2 * k]) + T.ranges[2 * k]);
            } else if (T.op == IMX_E_APPLY_EXTERNAL_FORCE_TORQUE) {
                const int nb = T.body_ids_d ? T.num_body_ids : NB;
                for (int b = 0; b < nb; ++b) {
                    const int body = T.body_ids_d ? T.body_ids_d[b] : b;
                    for (int c = 0; c < 3; ++c) {
                        const size_t at = ((size_t)e * NB + body) * 3 + c;
                        o.ext_force_out_d[at] = draw(U, 6 * (int64_t)nb, e, b * 3 + c, o.seed, t, step) * (T.ranges[1] - T.ranges[0]) + T.ranges[0];
                        o.ext_torque_out_d[at] = draw(U, 6 * (int64_t)nb, e, nb * 3 + b * 3 + c, o.seed, t, step) * (T.ranges[3] - T.ranges[2]) + T.ranges[2];
                    }
                }
            }
        }
    }

    // ---- partial sums for the log (fixed-shape shuffle tree: deterministic)
    if (o.ev_part_d) {
        const float a = wsum(reset ? mxy0 : 0.0f), b = wsum(reset ? myaw0 : 0.0f), c = wsum(level_f), d = wsum(reset ? 1.0f : 0.0f);
        if (lane == 0) {
            float4* p = reinterpret_cast<float4*>(o.ev_part_d) + blockIdx.x;
            *p = make_float4(a, b, c, d);
        }
    }
}

}  // namespace

extern "C" size_t imx_orch_part_floats(int64_t num_envs) { return num_envs > 0 ? (size_t)((num_envs + ORCH_BLOCK - 1) / ORCH_BLOCK) * 4 : 0; }

extern "C" int imx_reset_orchestrate(const imx_orch_t* o, imx_stream_t stream) {
    IMX_REQUIRE(o, "imx_reset_orchestrate: null descriptor");
    IMX_REQUIRE(o->num_envs > 0 && o->num_envs < (1ll << 31), "imx_reset_orchestrate: num_envs out of range");
    IMX_REQUIRE(o->num_terms >= 0 && o->num_terms <= IMX_ORCH_MAX_TERMS, "imx_reset_orchestrate: %d event terms (at most %d)", o->num_terms,
                IMX_ORCH_MAX_TERMS);
    IMX_REQUIRE(o->env_origins_d, "imx_reset_orchestrate: env_origins missing");
    bool need_root = false, need_vel = false;
    for (int t = 0; t < o->num_terms; ++t) {
        const imx_event_term_t& T = o->terms[t];
        IMX_REQUIRE(T.mode == 0 || T.mode == 1, "imx_reset_orchestrate: term %d has mode %d (0 reset, 1 interval)", t, T.mode);
        if (T.mode == 0) IMX_REQUIRE(T.last_triggered_step_d && T.triggered_once_d, "imx_reset_orchestrate: reset term %d lacks its trigger state", t);
        if (T.mode == 1) {
            IMX_REQUIRE(T.time_left_d, "imx_reset_orchestrate: interval term %d lacks its timer", t);
            IMX_REQUIRE(T.op == IMX_E_PUSH_BY_SETTING_VELOCITY || T.op == IMX_E_APPLY_EXTERNAL_FORCE_TORQUE,
                        "imx_reset_orchestrate: interval term %d: op %d is not an interval event here", t, T.op);
        }
        switch (T.op) {
            case IMX_E_RESET_ROOT_STATE_UNIFORM:
                IMX_REQUIRE(o->default_root_state_d && o->root_pose_out_d && o->root_vel_out_d, "imx_reset_orchestrate: reset_root_state_uniform needs "
                            "default_root_state, root_pose_out and root_vel_out");
                break;
            case IMX_E_RESET_JOINTS_BY_SCALE: case IMX_E_RESET_JOINTS_BY_OFFSET:
                IMX_REQUIRE(o->num_joints > 0 && o->default_joint_pos_d && o->default_joint_vel_d && o->soft_joint_pos_limits_d &&
                            o->soft_joint_vel_limits_d && o->joint_pos_out_d && o->joint_vel_out_d,
                            "imx_reset_orchestrate: a joint reset needs the joint defaults, limits and outputs");
                break;
            case IMX_E_PUSH_BY_SETTING_VELOCITY:
                need_vel = true;
                IMX_REQUIRE(o->root_vel_out_d, "imx_reset_orchestrate: push_by_setting_velocity needs root_vel_out");
                break;
            case IMX_E_APPLY_EXTERNAL_FORCE_TORQUE:
                IMX_REQUIRE(o->num_bodies > 0 && o->ext_force_out_d && o->ext_torque_out_d, "imx_reset_orchestrate: apply_external_force_torque needs "
                            "the body count and the force / torque outputs");
                IMX_REQUIRE(!T.body_ids_d || (T.num_body_ids > 0 && T.num_body_ids <= o->num_bodies), "imx_reset_orchestrate: bad body id count");
                break;
            default: IMX_FAIL("imx_reset_orchestrate: term %d has unknown op %d", t, T.op);
        }
    }
    if (o->terrain_levels_d) {
        need_root = true;
        IMX_REQUIRE(o->terrain_origins_d && o->terrain_types_d && o->terrain_rows > 0 && o->terrain_cols > 0 && o->vel_command_b_d,
                    "imx_reset_orchestrate: the terrain curriculum needs terrain origins, types, the grid size and the velocity command");
    }
    if (o->has_command) {
        need_vel = true;
        IMX_REQUIRE(o->root_quat_w_d && o->vel_command_b_d && o->heading_target_d && o->is_heading_env_d && o->is_standing_env_d &&
                    o->command_time_left_d && o->command_counter_d && o->metric_error_vel_xy_d && o->metric_error_vel_yaw_d,
                    "imx_reset_orchestrate: the command term lacks a state tensor");
        IMX_REQUIRE(o->command_cfg[13] > 0.0f, "imx_reset_orchestrate: max_command_step must be positive");
    }
    IMX_REQUIRE(!need_root || o->root_pos_w_d, "imx_reset_orchestrate: root_pos_w missing");
    IMX_REQUIRE(!need_vel || (o->root_lin_vel_w_d && o->root_ang_vel_w_d), "imx_reset_orchestrate: root velocities missing");
    if (o->cs_timestamp_d)
        IMX_REQUIRE(o->cs_timestamp_last_update_d && o->cs_is_outdated_d && o->cs_net_forces_w_d && o->cs_num_bodies > 0 &&
                    (o->cs_last_air_time_d == nullptr) == (o->cs_current_air_time_d == nullptr) &&
                    (o->cs_last_air_time_d == nullptr) == (o->cs_last_contact_time_d == nullptr) &&
                    (o->cs_last_air_time_d == nullptr) == (o->cs_current_contact_time_d == nullptr),
                    "imx_reset_orchestrate: incomplete contact-sensor state");
    if (o->lstm_hidden_d) IMX_REQUIRE(o->lstm_cell_d && o->lstm_layers > 0 && o->lstm_hidden_dim > 0 && o->num_joints > 0,
                                      "imx_reset_orchestrate: incomplete actuator-net state");
    const unsigned grid = (unsigned)((o->num_envs + ORCH_BLOCK - 1) / ORCH_BLOCK);
    hipLaunchKernelGGL(k_reset_orchestrate, dim3(grid), dim3(ORCH_BLOCK), 0, (hipStream_t)stream, *o);
    IMX_HIP(hipGetLastError());
    return 0;
}
