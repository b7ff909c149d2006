// Device code shared by the stand-alone producer kernels (producers.hip) and the reset / interval orchestration kernel (orchestrate.hip).
#pragma once
#include "imx_internal.h"

// ------------------------------------------------------------------------------------------------- velocity command
// CommandTerm.reset / compute + UniformVelocityCommand (isaaclab/managers/command_manager.py:119-187,
// isaaclab/envs/mdp/commands/velocity_command.py:111-160) for ONE env.
// uniforms: (2,N,7) samples in [0,1) for {time_left, lin_x, lin_y, ang_z, heading, is_heading, is_standing} of the (up to) two
// resamplings of a call (reset, timer) -- parity mode, the reference draws them with Tensor.uniform_ on the CPU generator -- or NULL ->
// counter-based in-kernel.
struct VelCmdCfg {
    float resample_lo, resample_hi;
    float lin_x_lo, lin_x_hi, lin_y_lo, lin_y_hi, ang_z_lo, ang_z_hi, heading_lo, heading_hi;
    float rel_standing, rel_heading, stiffness;
    int heading_command;
    float max_command_step;  // resampling_time_range[1] / step_dt
};
struct VelCmdState {
    float* cmd;              // (N,3) vel_command_b
    float* heading_target;   // (N)
    uint8_t* is_heading;     // (N)
    uint8_t* is_standing;    // (N)
    float* time_left;        // (N)
    int64_t* counter;        // (N) command_counter
    float* metric_xy;        // (N) metrics["error_vel_xy"]
    float* metric_yaw;       // (N)
};

IMX_DEV float u_at(const float* __restrict__ U, int64_t e, int k, uint64_t seed, uint32_t step) {
    return U ? U[e * 7 + k] : uniform01(seed + 0x1234567ull * (uint64_t)(k + 1), step, (uint64_t)e);
}

// `reset`: CommandTerm.reset for this env first (metrics logged and zeroed, counter zeroed, resample); `do_compute`: then
// CommandTerm.compute(dt).  mxy0 / myaw0 receive the metrics as they stood BEFORE the reset (what CommandTerm.reset logs).
IMX_DEV void velocity_command_env(int64_t N, int64_t e, const VelCmdCfg& c, float dt, int do_compute, const float* __restrict__ quat,
                                  const float* __restrict__ lin_w, const float* __restrict__ ang_w, bool reset,
                                  const float* __restrict__ uniforms, uint64_t seed, uint32_t step, const VelCmdState& s, float& mxy0,
                                  float& myaw0) {
    const float qw = quat[e * 4], qx = quat[e * 4 + 1], qy = quat[e * 4 + 2], qz = quat[e * 4 + 3];
    float cx = s.cmd[e * 3], cy = s.cmd[e * 3 + 1], cz = s.cmd[e * 3 + 2];
    float tl = s.time_left[e];
    int64_t cnt = s.counter[e];
    float mxy = s.metric_xy[e], myaw = s.metric_yaw[e];
    mxy0 = mxy; myaw0 = myaw;
    float htgt = s.heading_target[e];
    bool head = s.is_heading[e] != 0, stand = s.is_standing[e] != 0;
    bool resample = false;
    int draw = 0;  // which of the two possible resamplings of this call (reset, timer) -> distinct in-kernel streams
    if (reset) {  // CommandTerm.reset (command_manager.py:119-147): metrics, counter, resample
        mxy = 0.0f; myaw = 0.0f; cnt = 0;
        resample = true;
    }
    for (int pass = 0; pass < (do_compute ? 2 : 1); ++pass) {
        if (pass == 1) {
            // CommandTerm.compute (:149-166): metrics on the current command, timer, resample when it ran out
            float lbx, lby, lbz, abx, aby, abz;
            quat_rotate_inverse(qw, qx, qy, qz, lin_w[e * 3], lin_w[e * 3 + 1], lin_w[e * 3 + 2], lbx, lby, lbz);
            quat_rotate_inverse(qw, qx, qy, qz, ang_w[e * 3], ang_w[e * 3 + 1], ang_w[e * 3 + 2], abx, aby, abz);
            const float ex = cx - lbx, ey = cy - lby;
            mxy += sqrtf(ex * ex + ey * ey) / c.max_command_step;  // velocity_command.py:117-123
            myaw += fabsf(cz - abz) / c.max_command_step;
            tl -= dt;
            resample = tl <= 0.0f;
        }
        if (resample) {  // CommandTerm._resample (:172-187) + _resample_command (velocity_command.py:125-140)
            const float* U = uniforms ? uniforms + (size_t)draw * N * 7 : nullptr;
            const uint64_t sd = seed + 0x9E3779B97F4A7C15ull * (uint64_t)draw;
            tl = u_at(U, e, 0, sd, step) * (c.resample_hi - c.resample_lo) + c.resample_lo;
            cnt += 1;
            cx = u_at(U, e, 1, sd, step) * (c.lin_x_hi - c.lin_x_lo) + c.lin_x_lo;
            cy = u_at(U, e, 2, sd, step) * (c.lin_y_hi - c.lin_y_lo) + c.lin_y_lo;
            cz = u_at(U, e, 3, sd, step) * (c.ang_z_hi - c.ang_z_lo) + c.ang_z_lo;
            if (c.heading_command) {
                htgt = u_at(U, e, 4, sd, step) * (c.heading_hi - c.heading_lo) + c.heading_lo;
                head = u_at(U, e, 5, sd, step) <= c.rel_heading;
            }
            stand = u_at(U, e, 6, sd, step) <= c.rel_standing;
            ++draw;
        }
        resample = false;
    }
    // _update_command (velocity_command.py:142-160)
    if (do_compute && c.heading_command && head) {
        float fx, fy, fz;
        quat_apply(qw, qx, qy, qz, 1.0f, 0.0f, 0.0f, fx, fy, fz);  // heading_w (articulation_data.py:518-526)
        const float heading = atan2f(fy, fx);
        const float err = wrap_to_pi(htgt - heading);
        cz = fminf(fmaxf(c.stiffness * err, c.ang_z_lo), c.ang_z_hi);
    }
    if (do_compute && stand) { cx = 0.0f; cy = 0.0f; cz = 0.0f; }
    s.cmd[e * 3] = cx; s.cmd[e * 3 + 1] = cy; s.cmd[e * 3 + 2] = cz;
    s.heading_target[e] = htgt;
    s.is_heading[e] = head ? 1 : 0;
    s.is_standing[e] = stand ? 1 : 0;
    s.time_left[e] = tl;
    s.counter[e] = cnt;
    s.metric_xy[e] = mxy;
    s.metric_yaw[e] = myaw;
}

static inline VelCmdCfg vel_cmd_cfg_from15(const float* cfg15, int heading_command) {
    VelCmdCfg c;
    c.resample_lo = cfg15[0]; c.resample_hi = cfg15[1];
    c.lin_x_lo = cfg15[2]; c.lin_x_hi = cfg15[3]; c.lin_y_lo = cfg15[4]; c.lin_y_hi = cfg15[5];
    c.ang_z_lo = cfg15[6]; c.ang_z_hi = cfg15[7]; c.heading_lo = cfg15[8]; c.heading_hi = cfg15[9];
    c.rel_standing = cfg15[10]; c.rel_heading = cfg15[11]; c.stiffness = cfg15[12];
    c.max_command_step = cfg15[13];
    c.heading_command = heading_command;
    return c;
}
