/* imx.h -- C ABI of libimx: the MI355X (gfx950) implementation of IsaacLab's post-physics env-step hot path.
 *
 * The reference (godk1122/IsaacLab v2.1.0) has NO native ABI on this path: the boundary is two Python protocols
 * (ManagerBasedRLEnv.step/reset and RslRlVecEnvWrapper).  This header is what the Python
 * shim (isaaclab_amd/_lib.py, ctypes) binds; each entry point cites the reference code it replaces.
 * Paths are relative to /root/reference/source/isaaclab/isaaclab unless noted.
 *
 * Conventions
 *   - every pointer named *_d / inside imx_state_t / imx_buffers_t is a DEVICE pointer owned by the caller
 *     (torch tensors); the library never allocates per call, never synchronises the stream, never frees them;
 *   - all launches go to the hipStream_t passed in (the caller passes torch's current stream);
 *   - return value: 0 = ok, non-zero = error; imx_last_error() gives the message (thread-local);
 *   - floats are fp32, bools are 1-byte (torch.bool layout), indices int64 (torch.long), row-major contiguous.
 */
#ifndef IMX_H_
#define IMX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct imx_plan imx_plan_t; /* compiled term tables (one per env cfg) */
typedef struct imx_mesh imx_mesh_t; /* terrain mesh + 2-D uniform grid acceleration structure */
typedef void* imx_stream_t;         /* hipStream_t */

/* ---- plan blob: int32 words written by isaaclab_amd/plan.py -------------------------------------------------- */
#define IMX_MAGIC 0x31584D49 /* "IMX1" */
#define IMX_PLAN_VERSION 3
#define IMX_HEADER_WORDS 48
#define IMX_MAX_OBS_GROUPS 4
#define IMX_REC_WORDS 20

enum imx_header_word {
    IMX_H_MAGIC = 0, IMX_H_VERSION, IMX_H_J, IMX_H_B, IMX_H_H, IMX_H_A, IMX_H_D, IMX_H_R, IMX_H_NTERM, IMX_H_NREW,
    IMX_H_NOBS, IMX_H_NACT, IMX_H_MAX_EP_LEN, IMX_H_STEP_DT /*f32*/, IMX_H_TERM_OFF, IMX_H_REW_OFF, IMX_H_OBS_OFF,
    IMX_H_ACT_OFF, IMX_H_TOTAL_WORDS, IMX_H_NB /* articulation bodies */, IMX_H_GRAV_X /*f32*/, IMX_H_GRAV_Y,
    IMX_H_GRAV_Z, IMX_H_NREW_ALL /* reward terms incl. zero-weight */, IMX_H_RAY_OFF /* R*3 f32 local ray starts */,
    IMX_H_RAYDIR_X /*f32*/, IMX_H_RAYDIR_Y, IMX_H_RAYDIR_Z, IMX_H_RAY_MAXDIST /*f32*/, IMX_H_MAX_EP_LEN_S /*f32*/,
    IMX_H_NEXT_REW, IMX_H_NEXT_TERM, IMX_H_NEXT_OBS, IMX_H_RAY_YAW_ONLY, IMX_H_CMD_DIM,
    IMX_H_MOD_STATE /* floats of observation-modifier state per env (imx_buffers.mod_state row width) */,
    IMX_H_NGROUPS /* observation groups (ObservationManager group loop, observation_manager.py:238-258), 1..IMX_MAX_OBS_GROUPS */,
    IMX_H_GROUP_OFF /* NGROUPS x 4 words: D_g, enable_corruption, first obs record, number of obs records (IMX_H_D = sum of D_g) */,
    /* height scanner as a SensorBase (sensors/sensor_base.py:182-205,287-297; ray_caster.py:107-114,236-237) */
    IMX_H_SCAN_PERIOD /* f32 cfg.update_period */, IMX_H_SCAN_DT /* f32 physics dt: SensorBase.update(dt) per physics step */,
    IMX_H_SCAN_SUBSTEPS /* decimation: update() calls per env step */, IMX_H_SCAN_DRIFT_LO /* f32 cfg.drift_range */,
    IMX_H_SCAN_DRIFT_HI, IMX_H_SCAN_STATEFUL /* 1: per-env timestamps / drift are kept (imx_buffers.scan_state) */
};

/* record layout (IMX_REC_WORDS int32/f32 words) */
enum imx_rec_word {
    IMX_R_OP = 0, IMX_R_IDS_OFF, IMX_R_NIDS, IMX_R_IDS2_OFF, IMX_R_NIDS2,
    IMX_R_WEIGHT /*f32 reward weight (0 = skipped, reward_manager.py:145); termination: time_out flag; observation: group index*/,
    IMX_R_P0 /*f32*/, IMX_R_P1, IMX_R_P2, IMX_R_P3, IMX_R_OUT /* obs: column offset; reward/term: term index */,
    IMX_R_DIM, IMX_R_FLAGS, IMX_R_NOISE_LO /*f32*/, IMX_R_NOISE_HI, IMX_R_CLIP_LO, IMX_R_CLIP_HI, IMX_R_SCALE,
    IMX_R_AUX0, IMX_R_AUX1
};
/* obs post-processing flags (observation_manager.py:310-318) */
#define IMX_F_NOISE_ADD 1
#define IMX_F_NOISE_SCALE 2
#define IMX_F_NOISE_ABS 4
#define IMX_F_CLIP 8
#define IMX_F_SCALE 16
#define IMX_F_QUAT_UNIQUE 32
/* the obs record carries a modifier program (ObservationTermCfg.modifiers, observation_manager.py:310-312), run on the raw
 * term value before noise/clip/scale: IMX_R_IDS2_OFF = word offset of the program, IMX_R_NIDS2 = its length in words,
 * IMX_R_P1 (int) = first float of the term's state in a mod_state row.  Program = ops back to back, 4 words each
 * [op, a, b, soff] (DIGITAL_FILTER: [op, na, nb, soff] followed by na + nb f32 coefficients A then B).  State of element
 * j of a term of width d: mod_state[e][P1 + (soff + k) * d + j]; zeroed for envs reset this step before it is used. */
#define IMX_F_MODIFIERS 64
/* a height_scan record that reuses the ray hits of an earlier height_scan record (IMX_R_AUX0 = that record's index), e.g. the
 * "critic" group scanning the same sensor as "policy": one cast per ray and step, as the reference's lazily updated sensor. */
#define IMX_F_SCAN_TWIN 128
/* the noise of the term is gaussian_noise (utils/noise/noise_model.py:71-94): NOISE_LO = mean, NOISE_HI = std, combined by the ADD / SCALE / ABS bit;
 * samples: the fed array holds the reference's randn draws, else an in-kernel Box-Muller pair of the counter-based uniforms.
 * (constant_noise, :17-39, needs no flag: a uniform noise with NOISE_LO = NOISE_HI = bias is bit-identical) */
#define IMX_F_NOISE_GAUSS 1024
enum imx_mod_op {            /* utils/modifiers/modifier.py */
    IMX_M_SCALE = 1,         /* :22-32   a = multiplier */
    IMX_M_BIAS,              /* :49-60   a = value */
    IMX_M_CLIP,              /* :35-46   a = lo (-inf if None), b = hi (+inf if None) */
    IMX_M_INTEGRATOR,        /* :179-259 a = dt; state: integral, y_prev */
    IMX_M_DIGITAL_FILTER     /* :63-176  state: x_n[nb], y_n[na] */
};
/* action record flags */
#define IMX_F_ACT_DEFAULT_POS_OFFSET 1 /* JointPositionAction use_default_offset (joint_actions.py:152-154) */
#define IMX_F_ACT_DEFAULT_VEL_OFFSET 2 /* JointVelocityAction (joint_actions.py:206-208) */
#define IMX_F_ACT_CLIP 4
#define IMX_F_ACT_EMA 16 /* EMAJointPositionToLimitsAction (joint_actions_to_limits.py:142-230): alpha * processed + (1 - alpha) * previous applied action,
                           clamped to the soft limits; alpha = P1 or the per-joint table AUX1; the previous applied action is processed_action
                           itself (an env reset in the last step starts from its joint positions, :208-217: reset_buf is read) */
#define IMX_F_ACT_TO_LIMITS 8 /* JointPositionToLimitsAction rescale_to_limits: clamp(-1, 1), unscale_transform onto soft_joint_pos_limits (joint_actions_to_limits.py:116-126) */

/* termination ops -- envs/mdp/terminations.py */
enum imx_term_op {
    IMX_T_TIME_OUT = 1,            /* :30-32 */
    IMX_T_ILLEGAL_CONTACT,         /* :150-158  ids=bodies p0=threshold */
    IMX_T_JOINT_POS_MANUAL_LIMIT,  /* :90-105   ids=joints p0=lo p1=hi */
    IMX_T_BAD_ORIENTATION,         /* :50-59    p0=limit_angle */
    IMX_T_ROOT_HEIGHT_BELOW_MIN,   /* :62-72    p0=minimum_height */
    IMX_T_JOINT_VEL_LIMIT,         /* :108-114  ids=joints */
    IMX_T_JOINT_VEL_MANUAL_LIMIT,  /* :117-124  ids=joints p0=max_velocity */
    IMX_T_JOINT_EFFORT_LIMIT,      /* :127-142  ids=joints (torch.isclose(computed, applied)) */
    IMX_T_TERRAIN_OUT_OF_BOUNDS,   /* isaaclab_tasks .../velocity/mdp/terminations.py:24-52  p0=x_lim p1=y_lim */
    IMX_T_EXTERNAL,                /* value computed by a Python term; aux0 = column in ext_term */
    IMX_T_COMMAND_RESAMPLE         /* :35-42    p0 = f32(step_dt), nids = num_resamples; state.command_time_left / command_counter */
};

/* reward ops -- envs/mdp/rewards.py unless noted */
enum imx_rew_op {
    IMX_W_IS_ALIVE = 1,               /* :31-33 */
    IMX_W_IS_TERMINATED,              /* :36-38 */
    IMX_W_IS_TERMINATED_TERM,         /* :41-68   ids = termination term indices */
    IMX_W_LIN_VEL_Z_L2,               /* :76-80 */
    IMX_W_ANG_VEL_XY_L2,              /* :83-87 */
    IMX_W_FLAT_ORIENTATION_L2,        /* :90-97 */
    IMX_W_BASE_HEIGHT_L2,             /* :100-122 (no sensor) p0=target */
    IMX_W_JOINT_TORQUES_L2,           /* :136-143 */
    IMX_W_JOINT_VEL_L1,               /* :146-150 */
    IMX_W_JOINT_VEL_L2,               /* :153-160 */
    IMX_W_JOINT_ACC_L2,               /* :163-170 */
    IMX_W_JOINT_DEVIATION_L1,         /* :173-179 */
    IMX_W_JOINT_POS_LIMITS,           /* :182-196 */
    IMX_W_JOINT_VEL_LIMITS,           /* :199-218 p0=soft_ratio */
    IMX_W_APPLIED_TORQUE_LIMITS,      /* :226-242 */
    IMX_W_ACTION_RATE_L2,             /* :245-247 */
    IMX_W_ACTION_L2,                  /* :250-252 */
    IMX_W_UNDESIRED_CONTACTS,         /* :260-268 ids=bodies p0=threshold */
    IMX_W_CONTACT_FORCES,             /* :271-279 ids=bodies p0=threshold */
    IMX_W_TRACK_LIN_VEL_XY_EXP,       /* :287-298 p0=std */
    IMX_W_TRACK_ANG_VEL_Z_EXP,        /* :301-309 p0=std */
    IMX_W_FEET_AIR_TIME,              /* isaaclab_tasks .../velocity/mdp/rewards.py:25-44 ids=bodies p0=threshold */
    IMX_W_FEET_AIR_TIME_POSITIVE_BIPED, /* ...:47-66 */
    IMX_W_FEET_SLIDE,                 /* ...:69-83  ids=sensor bodies ids2=asset bodies */
    IMX_W_TRACK_LIN_VEL_XY_YAW_FRAME_EXP, /* ...:86-96 */
    IMX_W_TRACK_ANG_VEL_Z_WORLD_EXP,  /* ...:99-106 */
    IMX_W_JOINT_POS_TARGET_L2,        /* isaaclab_tasks .../classic/cartpole/mdp/rewards.py:19-26 p0=target */
    IMX_W_EXTERNAL,                   /* aux0 = column in ext_reward */
    IMX_W_BODY_LIN_ACC_L2             /* :125-128 ids = asset bodies; state.body_lin_acc_w */
};

/* observation ops -- envs/mdp/observations.py */
enum imx_obs_op {
    IMX_O_BASE_POS_Z = 1,   /* :33-37 */
    IMX_O_BASE_LIN_VEL,     /* :40-44 */
    IMX_O_BASE_ANG_VEL,     /* :47-51 */
    IMX_O_PROJECTED_GRAVITY,/* :54-58 */
    IMX_O_ROOT_POS_W,       /* :61-65 (minus env origins) */
    IMX_O_ROOT_QUAT_W,      /* :68-83 */
    IMX_O_ROOT_LIN_VEL_W,   /* :85-89 */
    IMX_O_ROOT_ANG_VEL_W,   /* :92-96 */
    IMX_O_JOINT_POS,        /* :104-111 */
    IMX_O_JOINT_POS_REL,    /* :114-121 */
    IMX_O_JOINT_POS_LIMIT_NORMALIZED, /* :124-137 */
    IMX_O_JOINT_VEL,        /* :140-147 */
    IMX_O_JOINT_VEL_REL,    /* :150-157 */
    IMX_O_HEIGHT_SCAN,      /* :165-173 p0=offset */
    IMX_O_LAST_ACTION,      /* :512-521 */
    IMX_O_GENERATED_COMMANDS, /* :529-531 */
    IMX_O_EXTERNAL          /* aux0 = column offset in ext_obs */
};

/* action ops -- envs/mdp/actions/joint_actions.py:130-139 (raw*scale+offset[,clamp]) */
enum imx_act_op { IMX_A_JOINT_AFFINE = 1 };

/* ---- per-step inputs: the tensors ArticulationData / ContactSensorData / CommandManager expose ---------------- */
typedef struct imx_state {
    const float* root_pos_w;            /* (N,3)   assets/articulation/articulation_data.py root_pos_w */
    const float* root_quat_w;           /* (N,4) w,x,y,z */
    const float* root_lin_vel_w;        /* (N,3) */
    const float* root_ang_vel_w;        /* (N,3) */
    const float* joint_pos;             /* (N,J) */
    const float* joint_vel;             /* (N,J) */
    const float* joint_acc;             /* (N,J) */
    const float* applied_torque;        /* (N,J) */
    const float* computed_torque;       /* (N,J) */
    const float* default_joint_pos;     /* (N,J) */
    const float* default_joint_vel;     /* (N,J) */
    const float* soft_joint_pos_limits; /* (N,J,2) */
    const float* soft_joint_vel_limits; /* (N,J) */
    const float* body_lin_vel_w;        /* (N,NB,3) */
    const float* command;               /* (N,CMD) command_manager.get_command() */
    const float* net_forces_w_history;  /* (N,H,B,3) sensors/contact_sensor/contact_sensor_data.py */
    const float* last_air_time;         /* (N,B) */
    const float* current_air_time;      /* (N,B) */
    const float* current_contact_time;  /* (N,B) */
    const float* env_origins;           /* (N,3) scene.env_origins */
    const float* ext_reward;            /* (N,next_rew) values of Python-evaluated reward terms, or NULL */
    const uint8_t* ext_term;            /* (N,next_term) */
    const float* ext_obs;               /* (N,next_obs) */
    const float* body_lin_acc_w;        /* (N,NB,3) ArticulationData.body_lin_acc_w */
    const float* command_time_left;     /* (N)   CommandTerm.time_left (managers/command_manager.py:60-61) */
    const int64_t* command_counter;     /* (N)   CommandTerm.command_counter */
} imx_state_t;

/* ---- manager state + outputs (caller-owned, persistent across steps) ------------------------------------------ */
typedef struct imx_buffers {
    int64_t* episode_length_buf; /* (N)   envs/manager_based_rl_env.py:200 */
    float* action;               /* (N,A) managers/action_manager.py:318-331 */
    float* prev_action;          /* (N,A) */
    float* processed_action;     /* (N,A) */
    float* reward_buf;           /* (N)   managers/reward_manager.py:128-157 */
    float* episode_sums;         /* (NREW_ALL,N) */
    float* step_reward;          /* (N,NREW_ALL) */
    uint8_t* term_dones;         /* (NTERM,N) managers/termination_manager.py:151-174 */
    uint8_t* terminated;         /* (N) */
    uint8_t* truncated;          /* (N) */
    uint8_t* reset_buf;          /* (N) */
    int64_t* reset_env_ids;      /* (N) ascending ids of reset envs; valid prefix = counters[0] */
    int32_t* counters;           /* (8) [0] reset count [1] ticket [2] step counter (RNG streams, sensor stamps): imx_terminations_rewards
                                    publishes [3] + 1 there [3] its shadow, written by imx_observations (a step = one of each)
                                    [4],[5] low / high word of the sensor-drift seed (caller-written) [6..] reserved */
    float* log_out;              /* (NREW_ALL + NTERM + 1 + 3) Episode_Reward/<term>, Episode_Termination/<term>, count, then the three
                                    orchestration entries (see ev_part) */
    float* obs;                  /* (N,D_0) first observation group; managers/observation_manager.py:238-335 */
    void* scratch;               /* imx_plan_scratch_bytes(plan, N) bytes */
    float* mod_state;            /* (N, IMX_H_MOD_STATE) DigitalFilter / Integrator state; NULL when the plan has none */
    float* obs_extra1;           /* (N,D_1) second observation group (e.g. "critic"); NULL when the plan has one group */
    float* obs_extra2;           /* (N,D_2) */
    float* obs_extra3;           /* (N,D_3) */
    float* scan_state;           /* (N,8) [timestamp, timestamp_last_update, drift xyz, data.pos_w z, outdated, step stamp] of the
                                    height scanner; required when IMX_H_SCAN_STATEFUL; the caller starts it with zeros and
                                    outdated = 1 (sensors start outdated, sensor_base.py _initialize_impl).  The sensor clock is advanced
                                    where the env's frame row is produced: by imx_terminations_rewards (which knows the step's reset
                                    flags) or, without it, by imx_observations.  The stamp (counters[2] + 1 as uint bits) makes a
                                    second call within one step repeat the first one's decision instead of advancing the clock again */
    float* scan_hit_z;           /* (N,R) data.ray_hits_w[..., 2] kept for envs whose sensor is not outdated at the next step */
    const float* scan_drift_feed;/* optional (N,3): drift values taken at a sensor reset instead of the in-kernel draw (parity runs) */
    float* log_accum;            /* optional (NREW_ALL + NTERM + 1): running sum over the steps of log_out -- the runner's per-iteration
                                    mean of extras["log"] (upstream on_policy_runner.py: ep_infos.append(infos["log"]) every step; the
                                    dict is only REFRESHED on steps that reset something, manager_based_rl_env.py:216, so a step without
                                    resets adds the entries of the last refresh again).  Updated where log_out is (the step tail) */
    const float* ev_part;        /* optional: imx_orch_t.ev_part_d of the imx_reset_orchestrate launch between imx_terminations_rewards and
                                    imx_observations -- the step tail then also writes log_out[NREW_ALL + NTERM + 1 ..]: the two
                                    Metrics/<command>/error_vel_xy|yaw entries (command_manager.py:123-149: means over the reset envs) and
                                    Curriculum/terrain_levels (curriculum_manager.py:95-118: mean level over all envs); needs the deferred
                                    tail (flags bit 0 / enable_corruption bit 4) */
    int64_t ev_flags;            /* bit 0: the metrics entries exist, bit 1: the curriculum entry exists */
} imx_buffers_t;

/* ---- slot t of an rsl_rl RolloutStorage, filled by the step kernel itself (imx_terminations_rewards_rollout) ------------------ */
typedef struct imx_rollout_slot {
    const float* value_t;        /* (N)   critic values of this step (storage.values[t]) -- the time-out bootstrap reads them */
    float* rewards_out;          /* (N)   storage.rewards[t] = reward + gamma * value_t * time_out   (PPO.process_env_step) */
    uint8_t* dones_out;          /* (N)   storage.dones[t] = terminated | time_out                   (vecenv_wrapper.py:178) */
    float* cur_reward_sum;       /* (N)   optional, with cur_ep_len: the runner's running episode return / length ... */
    float* cur_ep_len;           /* (N) */
    float* ep_stats3;            /* (3)   optional: finished episodes {sum return, sum length, count} (float atomics: logging only) */
    float gamma;
    int32_t bootstrap_time_outs; /* 0 for a finite-horizon task (vecenv_wrapper.py:184-185: no "time_outs" key) */
} imx_rollout_slot_t;

/* ---- the reset / interval orchestration around the step (imx_reset_orchestrate) ------------------------------------------------
 * ManagerBasedRLEnv._reset_idx (envs/manager_based_rl_env.py:347-392) for the envs the step reset -- CurriculumManager.compute
 * (terrain_levels_vel + TerrainImporter.update_env_origins), scene.reset (contact sensor, actuator state), EventManager.apply("reset")
 * with min_step_count_between_reset (managers/event_manager.py:233-260), CommandTerm.reset (metrics for the log, resample) -- followed by
 * what ManagerBasedRLEnv.step does next (:232-236): CommandManager.compute(dt) and EventManager.apply("interval", dt) with per-env or
 * global timers re-sampled in the kernel (event_manager.py:205-232).  One lane per env, ONE launch, no host read: the reference walks
 * nonzero() id lists with a host sync per interval term and per manager reset. */
#define IMX_ORCH_MAX_TERMS 8
enum imx_event_op {
    IMX_E_RESET_ROOT_STATE_UNIFORM = 1, /* envs/mdp/events.py:823-868;  ranges: pose lo/hi x 6 [0..11], velocity lo/hi x 6 [12..23] */
    IMX_E_RESET_JOINTS_BY_SCALE,        /* :987-1015;  ranges: position lo, hi, velocity lo, hi */
    IMX_E_RESET_JOINTS_BY_OFFSET,       /* :1020-1049 */
    IMX_E_PUSH_BY_SETTING_VELOCITY,     /* :795-820;   ranges: lo/hi x 6 */
    IMX_E_APPLY_EXTERNAL_FORCE_TORQUE   /* :764-791;   ranges: force lo, hi, torque lo, hi; body ids */
};
typedef struct imx_event_term {
    int32_t op;                            /* enum imx_event_op */
    int32_t mode;                          /* 0 = "reset", 1 = "interval" (EventTermCfg.mode) */
    int32_t is_global_time;                /* interval terms: one timer for all envs (event_manager.py:213-221) */
    int32_t min_step_count_between_reset;  /* reset terms (:233-260) */
    float interval_lo, interval_hi;        /* interval_range_s */
    float ranges[24];
    int32_t num_body_ids;
    int32_t reserved;
    const int32_t* body_ids_d;             /* IMX_E_APPLY_EXTERNAL_FORCE_TORQUE: resolved body ids (NULL = all bodies) */
    int32_t* last_triggered_step_d;        /* (N) reset terms: _reset_term_last_triggered_step_id */
    uint8_t* triggered_once_d;             /* (N) reset terms: _reset_term_last_triggered_once */
    float* time_left_d;                    /* interval terms: (N) per-env timers, or (2) for a global timer -- slot [step & 1] is read,
                                              slot [(step + 1) & 1] written (no read-modify-write race across workgroups) */
    const float* uniforms_d;               /* optional (N, width) samples in [0,1) replacing the in-kernel draws (parity runs); width =
                                              12 (root state), 2J (joints), 6 (push), 6 x bodies (wrench: forces first) */
    const float* interval_uniforms_d;      /* optional (N) samples for the timer re-sampling (a global timer takes [0]) */
} imx_event_term_t;

typedef struct imx_orch {
    int64_t num_envs, num_joints, num_bodies;
    const uint8_t* reset_mask_d;           /* (N) imx_buffers.reset_buf of this step; NULL = every env (ManagerBasedEnv.reset) */
    const int32_t* step_counter_d;         /* the global env step count on the device (_sim_step_counter // decimation): buf->counters + 2 */
    uint64_t seed;
    float dt;                              /* step_dt */
    int32_t do_step;                       /* 1: inside step() -- CommandManager.compute + interval events follow the resets; 0: reset() */
    int32_t num_terms;
    int32_t reserved0;
    imx_event_term_t terms[IMX_ORCH_MAX_TERMS];   /* cfg order */
    /* the asset: defaults, limits, the current (post-physics) root state */
    const float* default_root_state_d;     /* (N,13) */
    const float* default_joint_pos_d;      /* (N,J) */
    const float* default_joint_vel_d;
    const float* soft_joint_pos_limits_d;  /* (N,J,2) */
    const float* soft_joint_vel_limits_d;  /* (N,J) */
    const float* root_pos_w_d;             /* (N,3) */
    const float* root_quat_w_d;            /* (N,4) */
    const float* root_lin_vel_w_d;         /* (N,3) */
    const float* root_ang_vel_w_d;         /* (N,3) */
    float* env_origins_d;                  /* (N,3) scene.env_origins: read by the events, updated by the curriculum */
    /* what write_root_pose_to_sim / write_root_velocity_to_sim / write_joint_state_to_sim / set_external_force_and_torque receive */
    float* root_pose_out_d;                /* (N,7) */
    float* root_vel_out_d;                 /* (N,6) */
    float* joint_pos_out_d;                /* (N,J) */
    float* joint_vel_out_d;                /* (N,J) */
    float* ext_force_out_d;                /* (N,NB,3) */
    float* ext_torque_out_d;               /* (N,NB,3) */
    /* CurriculumManager: terrain_levels_vel (isaaclab_tasks/.../velocity/mdp/curriculums.py:26-55); NULL terrain_levels_d = none */
    const float* terrain_origins_d;        /* (R,C,3) */
    const int64_t* terrain_types_d;        /* (N) */
    int64_t* terrain_levels_d;             /* (N) */
    const int64_t* rand_levels_d;          /* optional (N): the randint_like draw (parity runs) */
    int32_t terrain_rows, terrain_cols;
    float terrain_size_x, max_episode_length_s;
    /* CommandManager: UniformVelocityCommand; has_command = 0: none */
    int32_t has_command, heading_command;
    float command_cfg[16];                 /* as imx_velocity_command's cfg15 */
    float* vel_command_b_d;                /* (N,3) */
    float* heading_target_d;
    uint8_t* is_heading_env_d;
    uint8_t* is_standing_env_d;
    float* command_time_left_d;
    int64_t* command_counter_d;
    float* metric_error_vel_xy_d;
    float* metric_error_vel_yaw_d;
    const float* command_uniforms_d;       /* optional (2,N,7) */
    /* scene.reset(env_ids): ContactSensor.reset (contact_sensor.py:143-165 + sensor_base.py:182-194), ActuatorNetLSTM.reset */
    float* cs_timestamp_d;                 /* NULL: the scene has no env-owned contact sensor */
    float* cs_timestamp_last_update_d;
    uint8_t* cs_is_outdated_d;
    float* cs_net_forces_w_d;              /* (N,B,3) */
    float* cs_net_forces_w_history_d;      /* (N,H,B,3) or NULL */
    float* cs_last_air_time_d;             /* (N,B) or NULL (track_air_time off) */
    float* cs_current_air_time_d;
    float* cs_last_contact_time_d;
    float* cs_current_contact_time_d;
    int32_t cs_num_bodies, cs_history_length;
    float* lstm_hidden_d;                  /* (L, N*J, H) sea_hidden_state or NULL */
    float* lstm_cell_d;
    int32_t lstm_layers, lstm_hidden_dim;
    /* per-workgroup partial sums for the step tail's log entries (imx_buffers.ev_part): {error_vel_xy, error_vel_yaw over the reset
     * envs; terrain levels over all envs; reset count} x ceil(N / 64) */
    float* ev_part_d;
} imx_orch_t;
/* number of floats of ev_part for num_envs */
size_t imx_orch_part_floats(int64_t num_envs);
int imx_reset_orchestrate(const imx_orch_t* orch, imx_stream_t stream);

/* ---- library ---------------------------------------------------------------------------------------------------- */
const char* imx_version(void);
/* sizeof of an ABI struct as this library was compiled (which: 0 imx_state_t, 1 imx_buffers_t, 2 imx_head_loss_t, 3 imx_rollout_slot_t, 4 imx_policy_act_t, 5 imx_orch_t, 6 imx_event_term_t), 0 for an unknown index:
 * a binding checks its own layout against it at load time. */
size_t imx_struct_size(int which);
const char* imx_last_error(void);
int imx_device_count(void);

/* term compiler output -> device tables.  Replaces ManagerBase._prepare_terms (managers/manager_base.py:160). */
int imx_plan_create(const int32_t* blob, size_t nwords, imx_plan_t** out);
void imx_plan_destroy(imx_plan_t* plan);
/* RewardManager.set_term_cfg / TerminationManager.set_term_cfg (managers/reward_manager.py:163-176,
 * managers/termination_manager.py:207-220; used by envs/mdp/curriculums.py:20-37 modify_reward_weight): replace the tables of
 * `plan` IN PLACE by a recompiled blob of the same shape (same term counts, widths and table sizes -- a changed weight, threshold,
 * id list of equal length ...).  The copy is enqueued on `stream`; graphs that captured launches on `plan` stay valid. */
int imx_plan_update(imx_plan_t* plan, const int32_t* blob, size_t nwords, imx_stream_t stream);
size_t imx_plan_scratch_bytes(const imx_plan_t* plan, int64_t num_envs);
int imx_plan_obs_dim(const imx_plan_t* plan);

/* ActionManager.process_action (managers/action_manager.py:318-337) + JointAction.process_actions
 * (envs/mdp/actions/joint_actions.py:130-139); pre_clip = RslRlVecEnvWrapper clip_actions
 * (isaaclab_rl/rsl_rl/vecenv_wrapper.py:173-174) or +inf. */
int imx_action_process(const imx_plan_t* plan, int64_t num_envs, const float* actions_d, float pre_clip,
                       const imx_state_t* state, const imx_buffers_t* buf, imx_stream_t stream);

/* Post-physics, pre-reset half of ManagerBasedRLEnv.step (envs/manager_based_rl_env.py:200-230):
 * episode_length_buf += 1; TerminationManager.compute; RewardManager.compute(dt); reset_env_ids = nonzero(reset_buf);
 * the manager-side part of _reset_idx (:347-392): Episode_* log reductions, episode sums / actions / episode length
 * zeroed for reset envs.
 * flags bit 0: leave the end of the step -- reset_env_ids, the reset count (counters[0]) and the Episode_* entries of log_out,
 * which need every env group's partial results -- to the imx_observations call that follows on the same stream (its
 * enable_corruption bit 4): a kernel boundary then orders them instead of a fence + ticket inside this launch.  Masks, rewards,
 * episodic sums and the reset of manager state are complete either way.
 * With state->root_pos_w present the launch also leaves the per-env frame rows (body-frame velocities, projected gravity, the
 * height scanner's update decision: SensorBase.update / reset, sensors/sensor_base.py:182-205) imx_observations reads when called
 * with enable_corruption bit 2. */
int imx_terminations_rewards(const imx_plan_t* plan, int64_t num_envs, const imx_state_t* state,
                             const imx_buffers_t* buf, int flags, imx_stream_t stream);
/* The same launch also doing what imx_rollout_post does (RslRlVecEnvWrapper.step dones, PPO.process_env_step time-out bootstrap, the
 * runner's episode statistics) for slot t of the storage: the lanes that produce reward and masks write them there too -- one launch
 * and one pass over the masks less per rollout step.  Bit-identical to imx_terminations_rewards followed by imx_rollout_post.
 * slot == NULL: plain imx_terminations_rewards.  (The log accumulation of imx_rollout_post is buf->log_accum.) */
int imx_terminations_rewards_rollout(const imx_plan_t* plan, int64_t num_envs, const imx_state_t* state, const imx_buffers_t* buf,
                                     int flags, const imx_rollout_slot_t* slot, imx_stream_t stream);

/* ObservationManager.compute (managers/observation_manager.py:238-335) fused with RayCaster._update_buffers_impl
 * (sensors/ray_caster/ray_caster.py:220-260) + raycast_mesh (utils/warp/ops.py:24-127).
 * noise_u_d: optional (N,D) uniform [0,1) samples replacing torch.rand_like (parity mode); NULL -> in-kernel
 * counter-based RNG keyed by (seed, step counter, env, column).  ray_hits_out_d: optional (N,R,3).
 * Terms with ObservationTermCfg.history_length > 0 keep their flattened (H, d) window (oldest first; CircularBuffer,
 * utils/buffers/circular_buffer.py:107-135) in the obs row itself: envs flagged in buf->reset_buf -- or every env when bit 1
 * of enable_corruption is set (env.reset()) -- take the new value in every slot, the others slide by one.  * enable_corruption bit 2 (value 4): the per-env frame table (root-frame vectors, scanner yaw) is current -- imx_terminations_rewards was
 * called with root_pos_w on the SAME state tensors since they last changed and wrote it -- so the k_frame launch is skipped.
 * enable_corruption bit 4 (value 16): finish the step tail a preceding imx_terminations_rewards (flags bit 0) deferred.
 * enable_corruption bit 3 (value 8): with a stateful height scanner, keep the hit heights of EVERY env in scan_hit_z, not only of
 * those whose fp32 timestamps say they will skip the next update (a caller that is about to overwrite the timestamps in scan_state). */
int imx_observations(const imx_plan_t* plan, int64_t num_envs, const imx_state_t* state, const imx_buffers_t* buf,
                     const imx_mesh_t* mesh, const float* noise_u_d, uint64_t seed, int enable_corruption,
                     float* ray_hits_out_d, imx_stream_t stream);
/* Name of the kernel imx_observations launches for this plan ("k_obs_lean<false>", "k_obs<false,true>", ...): benchmarks and
 * profiles attribute their timings to the kernel that actually ran.  Static string. */
const char* imx_observations_kernel_name(const imx_plan_t* plan);

/* ArticulationData.root_lin_vel_b / root_ang_vel_b / projected_gravity_b
 * (assets/articulation/articulation_data.py:512-515,603-619) = quat_rotate_inverse (utils/math.py:605-625). */
int imx_root_frame(int64_t num_envs, const float* quat_wxyz_d, const float* lin_vel_w_d, const float* ang_vel_w_d,
                   float gx, float gy, float gz, float* lin_vel_b_d, float* ang_vel_b_d, float* proj_gravity_d,
                   imx_stream_t stream);

/* convert_to_warp_mesh (utils/warp/ops.py:130-145): host vertices/triangles -> device mesh + grid. cell<=0: auto */
int imx_mesh_create(const float* vertices_h, int64_t num_vertices, const uint32_t* triangles_h, int64_t num_triangles,
                    float cell_size, imx_mesh_t** out);
void imx_mesh_destroy(imx_mesh_t* mesh);
/* info[0..7] = nx, ny, num_triangles, general triangle records, max refs per cell (low 32 bits) | FLAT cells (high 32 bits),
 * lattice cells, general cells (FLAT ones included), cell size (float bits) */
int imx_mesh_info(const imx_mesh_t* mesh, int64_t* info8);

/* raycast_mesh (utils/warp/ops.py:24-127): closest hit per ray, misses = +inf / face -1. */
int imx_raycast(const imx_mesh_t* mesh, const float* ray_starts_d, const float* ray_dirs_d, int64_t num_rays,
                float max_dist, float* ray_hits_d, float* ray_distance_d, int32_t* ray_face_id_d,
                imx_stream_t stream);

/* rsl_rl (3rd-party, v2.3.1) RolloutStorage.compute_returns: GAE backward scan over (T,N) + advantage
 * normalisation over all T*N.  scratch_d: >= imx_gae_scratch_bytes(T,N).  PARITY UNPINNED (rsl_rl absent). */
size_t imx_gae_scratch_bytes(int64_t T, int64_t N);
int imx_gae(int64_t T, int64_t N, const float* rewards_d, const float* values_d, const uint8_t* dones_d,
            const float* last_values_d, float gamma, float lam, int normalize, float* returns_d,
            float* advantages_d, void* scratch_d, imx_stream_t stream);

/* rsl_rl PPO.update elementwise part: surrogate / clipped value / entropy losses and the KL estimate for a
 * minibatch of M samples with A action dims.  sigma_stride = A for a per-sample (M,A) std, 0 for the shared (A) std
 * parameter.  fwd writes out8 = {surrogate_loss, value_loss, entropy_mean, kl_mean, loss, -, -, -} with
 * loss = surrogate + value_loss_coef*value_loss - entropy_coef*entropy, and (optional) accum5 += {value_loss,
 * surrogate, entropy, kl, 1}; bwd writes grad_scale * d(loss)/d(mu), /d(sigma) (per sample, (M,A)), /d(value); the policy
 * part (dmu_d, dsigma_d) and the value part (dvalue_d) are independent: pass NULL for one to launch them separately
 * (actor and critic then run forward -> loss gradient -> backward on their own streams without meeting).
 * PARITY UNPINNED (rsl_rl absent). */
size_t imx_ppo_scratch_bytes(int64_t M);
int imx_ppo_loss_fwd(int64_t M, int64_t A, const float* mu_d, const float* sigma_d, int64_t sigma_stride,
                     const float* actions_d, const float* old_logp_d, const float* old_mu_d, const float* old_sigma_d,
                     const float* advantages_d, const float* returns_d, const float* values_d, const float* old_values_d,
                     float clip_param, int use_clipped_value_loss, float value_loss_coef, float entropy_coef,
                     float* out8_d, float* accum5_d, void* scratch_d, imx_stream_t stream);
int imx_ppo_loss_bwd(int64_t M, int64_t A, const float* mu_d, const float* sigma_d, int64_t sigma_stride,
                     const float* actions_d, const float* old_logp_d, const float* advantages_d, const float* returns_d,
                     const float* values_d, const float* old_values_d, float clip_param, int use_clipped_value_loss,
                     float value_loss_coef, float entropy_coef, float grad_scale, float* dmu_d, float* dsigma_d,
                     float* dvalue_d, imx_stream_t stream);

/* out[c] = sum_m x[m][c] * (y ? y[m][c] : 1), x / y (M,A) row-major, A <= 64: the action-noise gradient from the per-sample
 * dsigma (``std.grad = dsigma.sum(0)``, or ``(dsigma * sigma).sum(0)`` for the log-std parametrisation of rsl_rl's ActorCritic).
 * Two launches, fixed summation order.  scratch_d: imx_colsum_scratch_bytes() bytes. */
size_t imx_colsum_scratch_bytes(void);
int imx_colsum(int64_t M, int64_t A, const float* x_d, const float* y_d, float* out_d, void* scratch_d, imx_stream_t stream);

/* torch.nn.utils.clip_grad_norm_ + torch.optim.Adam.step (as used by rsl_rl PPO.update) on one flat fp32 bucket.
 * lr_d / grad_norm_d are DEVICE scalars (adaptive-KL learning rate, ||g||_2); grad_norm_d NULL = no clipping.
 * step = 1-based Adam step count. */
int imx_adam_step(int64_t n, float* param_d, const float* grad_d, float* exp_avg_d, float* exp_avg_sq_d,
                  const float* lr_d, const float* grad_norm_d, float max_norm, float beta1, float beta2, float eps,
                  int64_t step, imx_stream_t stream);

/* The same with the whole schedule on the device (capturable in a hipGraph): state8_d = {lr, step, beta1^t, beta2^t,
 * clip coef, lr/bias1, sqrt(bias2), -}.  A one-thread kernel applies rsl_rl's adaptive-KL rule to lr when kl_d != NULL
 * (lr /= 1.5 if kl > 2*desired_kl; lr *= 1.5 if 0 < kl < desired_kl/2; bounds [1e-5, 1e-2]), advances the Adam step
 * and derives the clip coefficient from grad_norm_d; a second kernel applies Adam to the bucket. */
int imx_adam_update(int64_t n, float* param_d, const float* grad_d, float* exp_avg_d, float* exp_avg_sq_d, float* state8_d,
                    const float* kl_d, float desired_kl, const float* grad_norm_d, float max_norm, float beta1,
                    float beta2, float eps, imx_stream_t stream);

/* Same, with clip_grad_norm_'s total norm computed inside (one launch does the norm reduction, the adaptive-KL rule and
 * the step bookkeeping; state8[7] receives the norm).  max_norm <= 0 disables clipping.  scratch_d:
 * imx_adam_norm_scratch_bytes(n) bytes, ZERO-filled once by the caller (the kernel re-arms its ticket). */
size_t imx_adam_norm_scratch_bytes(int64_t n);
int imx_adam_update_norm(int64_t n, float* param_d, const float* grad_d, float* exp_avg_d, float* exp_avg_sq_d, float* state8_d,
                         const float* kl_d, float desired_kl, float max_norm, float beta1, float beta2, float eps,
                         void* scratch_d, size_t scratch_bytes, imx_stream_t stream);

/* rsl_rl RolloutStorage.mini_batch_generator: dst[k][r, :] = src[k][idx[r], :] for n <= 12 fp32 arrays of
 * width_floats[k] columns, one launch.  src_d / dst_d / width_floats are HOST arrays of device pointers / widths. */
int imx_gather_rows(int64_t M, const int64_t* idx_d, int n, const void* const* src_d, void* const* dst_d,
                    const int32_t* width_floats, imx_stream_t stream);
/* The same with a row pitch per destination array (floats, >= width; NULL = packed): a minibatch buffer whose rows start on 16-byte
 * boundaries lets imx_mlp_dw read it with 16-byte loads when the observation width is not a multiple of four (235, 310). */
int imx_gather_rows_pitched(int64_t M, const int64_t* idx_d, int n, const void* const* src_d, void* const* dst_d,
                            const int32_t* width_floats, const int32_t* dst_pitch_floats, imx_stream_t stream);

/* rsl_rl PPO.act after the actor / critic GEMMs: sample a = mu + std*N(0,1) (counter-based in-kernel generator keyed by
 * seed and *step_counter_d), log-prob, and the transition written into slot t of the RolloutStorage (obs, actions,
 * log-prob, mu, sigma, value); actions_env_d (optional) receives a second copy for env.step.  PARITY UNPINNED. */
int imx_policy_act(int64_t N, int64_t A, int64_t D, const float* mu_d, const float* std_d, const float* value_d,
                   const float* obs_d, uint64_t seed, const int32_t* step_counter_d, float* actions_out_d,
                   float* logp_out_d, float* mu_out_d, float* sigma_out_d, float* values_out_d, float* obs_out_d,
                   float* actions_env_d, imx_stream_t stream);

/* RslRlVecEnvWrapper.step dones (isaaclab_rl/rsl_rl/vecenv_wrapper.py:178) + rsl_rl PPO.process_env_step time-out
 * bootstrap (rewards += gamma * values * time_outs) + the runner's episode statistics, into slot t of the storage.
 * log_accum_d (optional, num_log floats) += log_in_d: the per-step extras["log"] entries (imx_buffers.log_out,
 * envs/manager_based_rl_env.py:365-389) summed on the device for the runner's per-iteration ep_infos mean. */
int imx_rollout_post(int64_t N, const float* reward_d, const uint8_t* terminated_d, const uint8_t* truncated_d,
                     const float* value_t_d, float gamma, int bootstrap_time_outs, float* rewards_out_d,
                     uint8_t* dones_out_d, int64_t* dones_long_d, float* cur_reward_sum_d, float* cur_ep_len_d,
                     float* ep_stats3_d, const float* log_in_d, float* log_accum_d, int num_log, imx_stream_t stream);

/* ---- input producers -------------------------------------------------------------------- */

/* SensorBase.update + ContactSensor._update_buffers_impl (sensors/sensor_base.py:196-205,287-297;
 * sensors/contact_sensor/contact_sensor.py:320-379): timestamp += dt; for outdated envs (flag set or
 * timestamp - last_update + 1e-6 >= update_period): net_forces_w <- new forces (N,B,3), history shifted by one
 * ((N,H,B,3), H may be 0), air/contact-time state machine with elapsed = timestamp - last_update; stamps updated. */
int imx_contact_sensor_update(int64_t N, int64_t B, int64_t H, const float* new_net_forces_d, float dt, float update_period,
                              float force_threshold, int track_air_time, float* timestamp_d,
                              float* timestamp_last_update_d, uint8_t* is_outdated_d, float* net_forces_w_d,
                              float* net_forces_w_history_d, float* last_air_time_d, float* current_air_time_d,
                              float* last_contact_time_d, float* current_contact_time_d, imx_stream_t stream);

/* CommandTerm.reset (for envs flagged in reset_mask_d, may be NULL) followed -- when do_compute != 0 -- by
 * CommandTerm.compute(dt) of
 * UniformVelocityCommand (managers/command_manager.py:119-187, envs/mdp/commands/velocity_command.py:111-160).
 * cfg15 (HOST floats) = {resampling_time lo,hi, lin_vel_x lo,hi, lin_vel_y lo,hi, ang_vel_z lo,hi, heading lo,hi,
 * rel_standing_envs, rel_heading_envs, heading_control_stiffness, resampling_time_range[1]/step_dt, -}.
 * uniforms_d: optional (2,N,7) samples in [0,1) {time_left, lin_x, lin_y, ang_z, heading, is_heading, is_standing} for
 * the first / second resampling of an env within this call (parity mode); NULL -> counter-based in-kernel generator. */
int imx_velocity_command(int64_t N, const float* cfg15, int heading_command, float dt, int do_compute,
                         const float* root_quat_w_d,
                         const float* root_lin_vel_w_d, const float* root_ang_vel_w_d, const uint8_t* reset_mask_d,
                         const float* uniforms_d, uint64_t seed, const int32_t* step_counter_d, float* vel_command_b_d,
                         float* heading_target_d, uint8_t* is_heading_env_d, uint8_t* is_standing_env_d,
                         float* time_left_d, int64_t* command_counter_d, float* metric_error_vel_xy_d,
                         float* metric_error_vel_yaw_d, imx_stream_t stream);

/* ArticulationData.root_state_w + joint_acc (assets/articulation/articulation_data.py:365-380,546-556): split PhysX's
 * root transforms (N,7: pos, quat XYZW) / velocities (N,6) into root_pos_w, root_quat_w (WXYZ, convert_quat
 * utils/math.py:177-222), root_lin_vel_w, root_ang_vel_w; joint_acc = (dof_vel - previous)/time_elapsed, previous <-
 * dof_vel (pass joint_acc_d = NULL to skip).  SURVEY 8f row 4. */
int imx_articulation_update(int64_t N, int64_t J, const float* root_transforms_xyzw_d, const float* root_velocities_d,
                            const float* dof_velocities_d, float time_elapsed, float* root_pos_w_d, float* root_quat_w_d,
                            float* root_lin_vel_w_d, float* root_ang_vel_w_d, float* previous_joint_vel_d,
                            float* joint_acc_d, imx_stream_t stream);

/* IdealPDActuator / ImplicitActuator (dc_motor = 0) and DCMotor (dc_motor = 1) .compute (actuators/actuator_pd.py:115-145,
 * 184-199, 264-286): computed = kp (q_des - q) + kd (qd_des - qd) + ff, applied = computed clipped to +-effort_limit or, for
 * the DC motor, to the velocity-dependent window built from saturation_effort and velocity_limit.  All arrays (N,J);
 * joint_vel_target_d / effort_ff_d may be NULL (zeros).  SURVEY 8f row 4. */
int imx_actuator_pd(int64_t N, int64_t J, int dc_motor, float saturation_effort, const float* joint_pos_target_d,
                    const float* joint_vel_target_d, const float* effort_ff_d, const float* joint_pos_d, const float* joint_vel_d,
                    const float* stiffness_d, const float* damping_d, const float* effort_limit_d, const float* velocity_limit_d,
                    float* computed_effort_d, float* applied_effort_d, imx_stream_t stream);

/* DelayedPDActuator / RemotizedPDActuator .compute (actuators/actuator_pd.py:289-412; DelayBuffer utils/buffers/delay_buffer.py,
 * CircularBuffer utils/buffers/circular_buffer.py).  ring_d: (max_delay+1, N, 3, J) f32, caller-owned, persistent; `step` = number of
 * calls so far (host counter, replaces the reference's pointer); reset_step_d[e] = the `step` of env e's first call after its last
 * reset (the caller sets it, with the new time_lags_d[e] in [min_delay, max_delay], when it resets the env); effort_limit_d NULL =
 * no box limit; lookup_d (num_lookup,3) = angle / transmission ratio / max torque samples, ascending angles, or NULL. */
int imx_actuator_delayed_pd(int64_t N, int64_t J, int max_delay, int64_t step, const int32_t* time_lags_d,
                            const int64_t* reset_step_d, float* ring_d, const float* joint_pos_target_d,
                            const float* joint_vel_target_d, const float* effort_ff_d, const float* joint_pos_d,
                            const float* joint_vel_d, const float* stiffness_d, const float* damping_d,
                            const float* effort_limit_d, const float* lookup_d, int num_lookup, float* computed_effort_d,
                            float* applied_effort_d, imx_stream_t stream);

/* ActuatorNetLSTM.compute (actuators/actuator_net.py:72-104; ANYDRIVE_3_LSTM_ACTUATOR_CFG, isaaclab_assets/robots/anymal.py:45-51,
 * is the actuator of ANYmal-B/C): network input (pos target - pos, vel) per (env, joint) sample, LSTM stack, dense head, the
 * DC-motor clip of the torque (DCMotor._clip_effort, actuators/actuator_pd.py:276-286).  The network travels as one packed float
 * array: per LSTM layer W_ih (4H x in), W_hh (4H x H), b_ih (4H), b_hh (4H) (torch's gate order i, f, g, o; in = 2 for the first
 * layer, H after), then per dense layer W (out x in), b (out); dense_out_h[k] = width of dense layer k (the last must be 1);
 * act = activation BETWEEN dense layers (0 identity, 1 softsign, 2 tanh, 3 relu, 4 elu).  hidden_state_d / cell_state_d:
 * (num_lstm, N*J, H), the reference's sea_hidden_state / sea_cell_state, updated in place (the caller zeroes rows at reset, :65-69).
 * All other arrays (N,J).  SURVEY 8f row 4. */
int imx_actuator_net_lstm(int64_t N, int64_t J, int num_lstm, int hidden, int num_dense, const int32_t* dense_out_h, int act,
                          const float* weights_d, int64_t num_weights, const float* joint_pos_target_d, const float* joint_pos_d,
                          const float* joint_vel_d, float* hidden_state_d, float* cell_state_d, float saturation_effort,
                          const float* effort_limit_d, const float* velocity_limit_d, float* computed_effort_d,
                          float* applied_effort_d, imx_stream_t stream);

/* ActuatorNetMLP.compute (actuators/actuator_net.py:160-195): the position-error and velocity queues (N, history_length, J) are
 * rolled by one and topped up in place; the sample of (env, joint) is the entries input_idx_d[0..num_idx) of both queues, scaled
 * by pos_scale / vel_scale, position block first (vel_first = 0, "pos_vel") or second ("vel_pos"); dense network as above (packed
 * W, b per layer; first width 2 * num_idx); torque * torque_scale, then the DC-motor clip. */
int imx_actuator_net_mlp(int64_t N, int64_t J, int num_dense, const int32_t* dense_out_h, int act, const float* weights_d,
                         int64_t num_weights, int history_length, const int32_t* input_idx_d, int num_idx, float pos_scale,
                         float vel_scale, float torque_scale, int vel_first, const float* joint_pos_target_d,
                         const float* joint_pos_d, const float* joint_vel_d, float* pos_error_history_d, float* vel_history_d,
                         float saturation_effort, const float* effort_limit_d, const float* velocity_limit_d,
                         float* computed_effort_d, float* applied_effort_d, imx_stream_t stream);

/* rsl_rl EmpiricalNormalization.forward (3rd party v2.3.1, absent): if update != 0 fold the batch (N,D) into the running
 * mean / variance (count, mean, var, std are device buffers; Chan's update with the biased batch variance), then
 * out = (x - mean) / (std + eps).  PARITY UNPINNED.  SURVEY 8f row 3. */
int imx_empirical_normalization(int64_t N, int64_t D, const float* x_d, int update, float eps, float* mean_d, float* var_d,
                                float* std_d, float* count_d, float* out_d, imx_stream_t stream);

/* ---- actor / critic MLP inside PPO.update (rsl_rl v2.3.1 ppo.py::update `loss.backward()` through the nn.Linear / nn.ELU
 * stacks of actor_critic.py; 3rd party, absent: PARITY UNPINNED, checked against torch autograd).  fp32 on the f32 MFMA.
 * The wide forward / dX GEMMs stay library calls; these are the shapes a library GEMM handles badly. */

/* Bytes of scratch imx_mlp_dw / imx_mlp_head_bwd need for a layer (out_features x in_features) at M samples. */
size_t imx_mlp_scratch_bytes(int64_t M, int out_features, int in_features);

/* nn.Linear backward, parameter part: dW[N][K] = dY^T X and (db_d != NULL) db[N] = column sums of dY, for
 * dY (M,N; row pitch ldy floats) and the layer input X (M,K; pitch ldx).  Split over the M samples across all CUs,
 * partials summed in a fixed order (deterministic). */
int imx_mlp_dw(int64_t M, int N, int K, const float* dY_d, int64_t ldy, const float* X_d, int64_t ldx, float* dW_d,
               float* db_d, void* scratch_d, size_t scratch_bytes, imx_stream_t stream);

/* Deferred reductions: imx_mlp_dw / imx_mlp_dw_elu / imx_mlp_head_bwd end with a small kernel that sums their split partials
 * into dW / db.  Between imx_reduce_batch_begin(b) and imx_reduce_batch_flush(b, stream), calls issued from the SAME host
 * thread queue that step on `b` instead (at most 8) and flush launches one kernel for all of them: dW / db are only needed
 * by the optimiser, so the backward chain loses a 6 us launch per layer.  Every deferred call needs its OWN scratch.
 * A batch empties at flush (not at begin): imx_mlp_head_fwd_bwd(..., defer_to = b, ...) may queue on it before it is opened. */
typedef struct imx_reduce_batch imx_reduce_batch_t;
int imx_reduce_batch_create(imx_reduce_batch_t** out);
void imx_reduce_batch_destroy(imx_reduce_batch_t* batch);
/* How many workgroups (one per CU) an imx_mlp_dw / imx_mlp_dw_elu launch may split its samples over; 0 = every CU (the default).  The PPO
 * update runs actor and critic backward passes on two streams at once: with half the chip each (128) the two launches stop competing for
 * the same CUs and the split partials (written, then re-read by the reduction) halve.  Returns the previous value.  Scratch sized
 * (imx_mlp_scratch_bytes) under a budget is valid for any budget <= that one. */
int imx_mlp_set_dw_cu_budget(int cus);
int imx_reduce_batch_begin(imx_reduce_batch_t* batch);
int imx_reduce_batch_flush(imx_reduce_batch_t* batch, imx_stream_t stream);

/* Same with the ELU backward of this layer fused in: dH_d is the gradient w.r.t. the layer's ACTIVATED output h = ELU(z),
 * H_d that saved output; dZ = dH * (h > 0 ? 1 : h + elu_alpha) (aten elu_backward with is_result) is formed on the way
 * into LDS -- no separate elementwise pass -- and written to dZ_out_d (M,N; may be NULL when no layer below needs it; must
 * not alias dH_d); dW = dZ^T X, db = colsum(dZ). */
int imx_mlp_dw_elu(int64_t M, int N, int K, const float* dH_d, int64_t ldg, const float* H_d, int64_t ldh, float elu_alpha,
                   float* dZ_out_d, int64_t ldd, const float* X_d, int64_t ldx, float* dW_d, float* db_d, void* scratch_d,
                   size_t scratch_bytes, imx_stream_t stream);

/* Rollout inference (rsl_rl actor_critic.py::act / evaluate): the whole Linear+ELU stack of up to two
 * networks sharing the input X (M, dims[0]; pitch ldx) in ONE launch, 32 samples per workgroup, activations in LDS.
 * nlayers[k] <= 4 Linear layers per network, ELU(elu_alpha[k]) after every layer but the last; dims = the layer widths of
 * network 0 (nlayers[0]+1 values, each <= 512) followed by those of network 1; weights_d / biases_d = HOST arrays of device
 * pointers, network 0's layers first; W (out,in) row-major with row pitch weight_pitch[i] floats (NULL = dense), which must
 * be a multiple of 32 with ZERO padding and 16-byte aligned rows -- a 235-wide first layer is passed as a padded copy
 * (pitch 256) -- so that every weight load is an unconditional 16-byte load; out_d[k] (M, last width). */
int imx_mlp_infer(int64_t M, const float* X_d, int64_t ldx, int nnets, const int* nlayers, const int* dims,
                  const float* const* weights_d, const int* weight_pitch, const float* const* biases_d, const float* elu_alpha,
                  float* const* out_d, imx_stream_t stream);

/* PPO.act (imx_policy_act) and ActionManager.process_action (imx_action_process) in the epilogue of the actor head of imx_mlp_infer:
 * the actor's workgroups keep their action means in LDS, sample a = mu + std * N(0,1) with the draws imx_policy_act makes (same seed,
 * step counter and element index), write the transition {obs, actions, log-prob, mu, sigma} into slot t of the RolloutStorage and -- when
 * `plan` is given -- run the sampled action through the env's action terms (buf->prev_action <- action <- clamp(a, pre_clip),
 * buf->processed_action: managers/action_manager.py:318-337, envs/mdp/actions/joint_actions.py:130-139).  All of it is pre-physics, so
 * it is valid with a simulator in the loop as well.  The critic's value goes where out_d[1] points (storage.values[t]).  Three launches
 * (k_mlp_infer, k_policy_act, k_action) become one; results are bit-identical to the three. */
typedef struct imx_policy_act {
    const float* std_d;            /* (A) action std (scalar noise_std_type) */
    uint64_t seed;
    const int32_t* step_counter_d; /* device step counter keying the draws (buf->counters + 2), or NULL */
    float* actions_out_d;          /* (N,A) storage.actions[t] */
    float* logp_out_d;             /* (N)   storage.actions_log_prob[t] */
    float* mu_out_d;               /* (N,A) storage.mu[t] */
    float* sigma_out_d;            /* (N,A) storage.sigma[t] */
    float* obs_out_d;              /* (N,D) storage.observations[t] (dense rows) */
    const imx_plan_t* plan;        /* optional: also process the action (with state, buf, pre_clip) */
    const imx_state_t* state;
    const imx_buffers_t* buf;
    float pre_clip;                /* RslRlVecEnvWrapper clip_actions or +inf */
} imx_policy_act_t;
/* out_d[0] may be NULL when act != NULL (the means go to mu_out_d); act == NULL: plain imx_mlp_infer.
 * packed_weights_d (optional, HOST array of device pointers like weights_d): the same weights re-ordered by imx_mlp_pack_weights.  In
 * the row layout a wave's load instruction reads 16 bytes from each of 32 weight rows; the packed image holds those 64 x 16 bytes
 * contiguously, in the order the 32-sample kernel consumes them, so every load is one full KiB.  Same values, same arithmetic, same
 * results; weights_d must still be given (the 16-sample kernel for small M reads the rows). */
int imx_mlp_infer_act(int64_t M, const float* X_d, int64_t ldx, int nnets, const int* nlayers, const int* dims,
                      const float* const* weights_d, const int* weight_pitch, const float* const* packed_weights_d,
                      const float* const* biases_d, const float* elu_alpha, float* const* out_d, const imx_policy_act_t* act,
                      imx_stream_t stream);
/* Floats of the packed image of an (out_features x in_features) nn.Linear weight (both rounded up to multiples of 32), and the packing
 * itself (W_d row-major with row pitch ldw >= in_features; packed_d 16-byte aligned; to be repeated whenever the weights change). */
size_t imx_mlp_packed_floats(int out_features, int in_features);
int imx_mlp_pack_weights(int out_features, int in_features, const float* W_d, int64_t ldw, float* packed_d, imx_stream_t stream);
/* The same for up to 8 layers in ONE launch (HOST arrays of nlayers entries): the refresh after an optimiser step. */
int imx_mlp_pack_weights_batch(int nlayers, const int* out_features, const int* in_features, const float* const* W_d, const int64_t* ldw,
                               float* const* packed_d, imx_stream_t stream);

/* First-layer forward of the update with the activation fused: Y (M,N; pitch ldy) = ELU(X W^T + b) (apply_elu = 0: no activation),
 * X (M,K; pitch ldx), W (N,K) dense row-major, K <= 256 (the observation width: 235, 48, 4 ...).  A library GEMM + a separate ELU pass
 * writes, reads and writes the output again.  Each wave keeps the weight rows of its 32 output columns in registers, the samples stream
 * through LDS in tiles of 32 rows (LDS DMA for K > 128 with 16-byte aligned rows), bias + ELU are applied to the accumulators.  fp32
 * products accumulated in fp32 (v_mfma_f32_32x32x2_f32; inputs k and NS + k paired per step, NS = the kernel's step count: another
 * summation order than a library GEMM's, same error class).  nn.Linear + nn.ELU of rsl_rl's ActorCritic MLPs (upstream
 * rsl_rl/modules/actor_critic.py). */
int imx_mlp_fwd_elu(int64_t M, int N, int K, const float* X_d, int64_t ldx, const float* W_d, const float* b_d, float elu_alpha,
                    int apply_elu, float* Y_d, int64_t ldy, imx_stream_t stream);

/* Output layer forward, A <= 64 outputs (action means / value): y[M][A] = h W^T + b, h (M,K; pitch ldh), W (A,K).
 * elu_in_place != 0: h_d holds the PRE-activation output of the layer below; h <- ELU(h) (aten elu, alpha = elu_alpha) is
 * applied on the way in and written back in place, so that layer needs no separate activation pass. */
int imx_mlp_head_fwd(int64_t M, int K, int A, float* h_d, int64_t ldh, const float* W_d, const float* b_d, float* y_d,
                     int elu_in_place, float elu_alpha, imx_stream_t stream);

/* imx_mlp_head_fwd followed by imx_ppo_loss_bwd in ONE launch: the lane that holds a sample's outputs computes the loss gradient
 * from them right away (policy head: mode 1 -> dmu_d, dsigma_d from mu = y; value head, A = 1: mode 2 -> dvalue_d from value = y).
 * Same arithmetic, same order as k_ppo_bwd (bit-identical gradients); saves a dependent launch on the update's critical chain. */
typedef struct imx_head_loss {
    int mode;                  /* 1 policy gradient, 2 value gradient */
    int sigma_stride;          /* 0 (shared std) or A */
    int use_clipped_value_loss;
    float clip_param, value_loss_coef, entropy_coef, grad_scale;
    const float* sigma_d;      /* mode 1 */
    const float* actions_d;
    const float* old_logp_d;
    const float* advantages_d;
    const float* returns_d;    /* mode 2 */
    const float* old_values_d;
    float* dmu_d;              /* mode 1 outputs (M,A) */
    float* dsigma_d;
    float* dvalue_d;           /* mode 2 output (M,1) */
} imx_head_loss_t;
int imx_mlp_head_fwd_loss(int64_t M, int K, int A, float* h_d, int64_t ldh, const float* W_d, const float* b_d, float* y_d,
                          int elu_in_place, float elu_alpha, const imx_head_loss_t* loss, imx_stream_t stream);

/* Output layer backward in one pass over h: dW[A][K] = dY^T h, db[A] = colsum(dY), and the gradient handed to the layer
 * below, dprev[M][K] = (dY W) * ELU'(h) with ELU' taken from the saved output h (h > 0 ? 1 : h + elu_alpha; aten
 * elu_backward with is_result) when has_activation != 0, else dprev = dY W. */
int imx_mlp_head_bwd(int64_t M, int K, int A, const float* dY_d, const float* h_d, int64_t ldh, const float* W_d, float elu_alpha,
                     int has_activation, float* dprev_d, float* dW_d, float* db_d, void* scratch_d, size_t scratch_bytes,
                     imx_stream_t stream);

/* imx_mlp_head_fwd_loss and imx_mlp_head_bwd in one pass over the last hidden layer's output z (M,K; pitch ldz; K = 128 or 256, A <= 16):
 * h = ELU(z) when elu_in != 0 (else h = z), y = h W^T + b, the loss gradient dY from y (loss->mode 1 or 2, written to loss->dmu_d /
 * dsigma_d / dvalue_d as imx_mlp_head_fwd_loss does), dprev = (dY W) * ELU'(h), dW = dY^T h, db = colsum(dY).  z is only read: the
 * activated values never travel to memory (the split pair wrote them in place and read them again).  defer_to: a reduce batch to
 * queue the reduction of the split partials on (NULL: the batch open on this thread, or an immediate launch). */
int imx_mlp_head_fwd_bwd(int64_t M, int K, int A, const float* z_d, int64_t ldz, const float* W_d, const float* b_d, float* y_d,
                         int elu_in, float elu_alpha, const imx_head_loss_t* loss, float* dprev_d, float* dW_d, float* db_d,
                         void* scratch_d, size_t scratch_bytes, imx_reduce_batch_t* defer_to, imx_stream_t stream);

/* ---- reset / interval events and terrain curriculum, masked: only rows with mask != 0 are rewritten
 * (mask NULL = every env).  The reference runs these on a compacted env_ids list (EventManager.apply,
 * managers/event_manager.py:150-273); draws are sample_uniform (utils/math.py:1313) = u*(hi-lo)+lo with u from the
 * counter-based generator (seed, *step_counter_d, env, column) or from uniforms_d in parity runs. */

/* reset_root_state_uniform (envs/mdp/events.py:823-868) and reset_joints_by_scale / _by_offset (:987-1049).
 * ranges28 (HOST) = pose {x,y,z,roll,pitch,yaw} (lo,hi) x6, velocity (lo,hi) x6, joint position lo,hi, joint velocity lo,hi.
 * joint_mode 0 = by scale, 1 = by offset, < 0 = joints untouched.  default_root_state (N,13: pos, quat wxyz, lin, ang),
 * soft_joint_pos_limits (N,J,2), soft_joint_vel_limits (N,J).  uniforms_d: optional (N, 12 + 2J) = [pose 6 | velocity 6 |
 * joint pos J | joint vel J].  Outputs = what write_root_pose_to_sim (N,7), write_root_velocity_to_sim (N,6) and
 * write_joint_state_to_sim (N,J)x2 receive. */
int imx_reset_events(int64_t N, int64_t J, const uint8_t* reset_mask_d, const float* ranges28, int joint_mode,
                     const float* default_root_state_d, const float* env_origins_d, const float* default_joint_pos_d,
                     const float* default_joint_vel_d, const float* soft_joint_pos_limits_d, const float* soft_joint_vel_limits_d,
                     const float* uniforms_d, uint64_t seed, const int32_t* step_counter_d, float* root_pose_d, float* root_vel_d,
                     float* joint_pos_d, float* joint_vel_d, imx_stream_t stream);

/* push_by_setting_velocity (events.py:795-820): root_vel_w (N,6) += sample; ranges12 (HOST) = (lo,hi) x6; uniforms_d (N,6). */
int imx_push_velocity(int64_t N, const uint8_t* mask_d, const float* ranges12, const float* uniforms_d, uint64_t seed,
                      const int32_t* step_counter_d, float* root_vel_w_d, imx_stream_t stream);

/* apply_external_force_torque (events.py:764-791): forces / torques ~ U(range) for bodies body_ids_d[0..num_ids) (device int32;
 * NULL = all num_bodies) of the masked envs, written into the (N, num_bodies, 3) buffers set_external_force_and_torque fills.
 * ranges4 (HOST) = force lo,hi, torque lo,hi; uniforms_d: optional (2, N, nb, 3), forces first. */
int imx_external_force_torque(int64_t N, int64_t num_bodies, const uint8_t* mask_d, const int32_t* body_ids_d, int64_t num_ids,
                              const float* ranges4, const float* uniforms_d, uint64_t seed, const int32_t* step_counter_d,
                              float* forces_d, float* torques_d, imx_stream_t stream);

/* terrain_levels_vel (isaaclab_tasks/.../velocity/mdp/curriculums.py:26-55) + TerrainImporter.update_env_origins
 * (terrains/terrain_importer.py:307-326): move up when the robot walked more than half a tile, down when less than half
 * the commanded distance; levels past the last are replaced by rand_levels_d[e] (randint_like; NULL -> in-kernel draw);
 * env_origins <- terrain_origins[level, type] (num_levels, num_types, 3); mean_level_d (optional) <- mean level. */
int imx_terrain_levels(int64_t N, int64_t num_levels, int64_t num_types, const uint8_t* mask_d, const float* root_pos_w_d,
                       const float* command_d, const float* terrain_origins_d, const int64_t* terrain_types_d, float terrain_size_x,
                       float max_episode_length_s, const int64_t* rand_levels_d, uint64_t seed, const int32_t* step_counter_d,
                       int64_t* terrain_levels_d, float* env_origins_d, float* mean_level_d, imx_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* IMX_H_ */
