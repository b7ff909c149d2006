"""ctypes binding of libimx.so (``include/imx.h``).  There is NO fallback: if the library cannot be loaded the
product path raises -- a silent eager/CPU path would void every parity and performance claim."""

from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libimx.so")

# field order must match include/imx.h
STATE_FIELDS = (
    "root_pos_w", "root_quat_w", "root_lin_vel_w", "root_ang_vel_w", "joint_pos", "joint_vel", "joint_acc",
    "applied_torque", "computed_torque", "default_joint_pos", "default_joint_vel", "soft_joint_pos_limits",
    "soft_joint_vel_limits", "body_lin_vel_w", "command", "net_forces_w_history", "last_air_time",
    "current_air_time", "current_contact_time", "env_origins", "ext_reward", "ext_term", "ext_obs",
    "body_lin_acc_w", "command_time_left", "command_counter",
)
BUFFER_FIELDS = (
    "episode_length_buf", "action", "prev_action", "processed_action", "reward_buf", "episode_sums", "step_reward",
    "term_dones", "terminated", "truncated", "reset_buf", "reset_env_ids", "counters", "log_out", "obs", "scratch", "mod_state",
    "obs_extra1", "obs_extra2", "obs_extra3", "scan_state", "scan_hit_z", "scan_drift_feed", "log_accum",
    "ev_part", "ev_flags",  # (ev_flags is an int64 carried in a pointer-sized field)
)


class ImxState(ctypes.Structure):
    _fields_ = [(n, c_void_p) for n in STATE_FIELDS]


class ImxBuffers(ctypes.Structure):
    _fields_ = [(n, c_void_p) for n in BUFFER_FIELDS]


class ImxHeadLoss(ctypes.Structure):  # imx_head_loss_t
    _fields_ = [("mode", ctypes.c_int), ("sigma_stride", ctypes.c_int), ("use_clipped_value_loss", ctypes.c_int),
                ("clip_param", ctypes.c_float), ("value_loss_coef", ctypes.c_float), ("entropy_coef", ctypes.c_float),
                ("grad_scale", ctypes.c_float)] + [(n, c_void_p) for n in (
                    "sigma_d", "actions_d", "old_logp_d", "advantages_d", "returns_d", "old_values_d", "dmu_d", "dsigma_d", "dvalue_d")]


class ImxRolloutSlot(ctypes.Structure):  # imx_rollout_slot_t
    _fields_ = [(n, c_void_p) for n in ("value_t", "rewards_out", "dones_out", "cur_reward_sum", "cur_ep_len", "ep_stats3")] + [
        ("gamma", ctypes.c_float), ("bootstrap_time_outs", ctypes.c_int32)]


class ImxPolicyAct(ctypes.Structure):  # imx_policy_act_t
    _fields_ = [("std_d", c_void_p), ("seed", ctypes.c_uint64), ("step_counter_d", c_void_p), ("actions_out_d", c_void_p),
                ("logp_out_d", c_void_p), ("mu_out_d", c_void_p), ("sigma_out_d", c_void_p), ("obs_out_d", c_void_p),
                ("plan", c_void_p), ("state", POINTER(ImxState)), ("buf", POINTER(ImxBuffers)), ("pre_clip", ctypes.c_float)]


class ImxEventTerm(ctypes.Structure):  # imx_event_term_t
    _fields_ = [("op", c_int32), ("mode", c_int32), ("is_global_time", c_int32), ("min_step_count_between_reset", c_int32),
                ("interval_lo", c_float), ("interval_hi", c_float), ("ranges", c_float * 24), ("num_body_ids", c_int32), ("reserved", c_int32)] + [
        (n, c_void_p) for n in ("body_ids_d", "last_triggered_step_d", "triggered_once_d", "time_left_d", "uniforms_d", "interval_uniforms_d")]


ORCH_MAX_TERMS = 8


class ImxOrch(ctypes.Structure):  # imx_orch_t
    _fields_ = ([("num_envs", c_int64), ("num_joints", c_int64), ("num_bodies", c_int64), ("reset_mask_d", c_void_p), ("step_counter_d", c_void_p),
                 ("seed", c_uint64), ("dt", c_float), ("do_step", c_int32), ("num_terms", c_int32), ("reserved0", c_int32),
                 ("terms", ImxEventTerm * ORCH_MAX_TERMS)]
                + [(n, c_void_p) for n in (
                    "default_root_state_d", "default_joint_pos_d", "default_joint_vel_d", "soft_joint_pos_limits_d", "soft_joint_vel_limits_d",
                    "root_pos_w_d", "root_quat_w_d", "root_lin_vel_w_d", "root_ang_vel_w_d", "env_origins_d", "root_pose_out_d", "root_vel_out_d",
                    "joint_pos_out_d", "joint_vel_out_d", "ext_force_out_d", "ext_torque_out_d", "terrain_origins_d", "terrain_types_d",
                    "terrain_levels_d", "rand_levels_d")]
                + [("terrain_rows", c_int32), ("terrain_cols", c_int32), ("terrain_size_x", c_float), ("max_episode_length_s", c_float),
                   ("has_command", c_int32), ("heading_command", c_int32), ("command_cfg", c_float * 16)]
                + [(n, c_void_p) for n in (
                    "vel_command_b_d", "heading_target_d", "is_heading_env_d", "is_standing_env_d", "command_time_left_d", "command_counter_d",
                    "metric_error_vel_xy_d", "metric_error_vel_yaw_d", "command_uniforms_d", "cs_timestamp_d", "cs_timestamp_last_update_d",
                    "cs_is_outdated_d", "cs_net_forces_w_d", "cs_net_forces_w_history_d", "cs_last_air_time_d", "cs_current_air_time_d",
                    "cs_last_contact_time_d", "cs_current_contact_time_d")]
                + [("cs_num_bodies", c_int32), ("cs_history_length", c_int32), ("lstm_hidden_d", c_void_p), ("lstm_cell_d", c_void_p),
                   ("lstm_layers", c_int32), ("lstm_hidden_dim", c_int32), ("ev_part_d", c_void_p)])


class ImxError(RuntimeError):
    pass


_lib = None

_SIGNATURES = {
    "imx_version": (c_char_p, []),
    "imx_last_error": (c_char_p, []),
    "imx_device_count": (c_int, []),
    "imx_plan_create": (c_int, [c_void_p, c_size_t, POINTER(c_void_p)]),
    "imx_plan_destroy": (None, [c_void_p]),
    "imx_plan_update": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "imx_struct_size": (c_size_t, [c_int]),
    "imx_plan_scratch_bytes": (c_size_t, [c_void_p, c_int64]),
    "imx_plan_obs_dim": (c_int, [c_void_p]),
    "imx_action_process": (c_int, [c_void_p, c_int64, c_void_p, c_float, POINTER(ImxState), POINTER(ImxBuffers), c_void_p]),
    "imx_terminations_rewards": (c_int, [c_void_p, c_int64, POINTER(ImxState), POINTER(ImxBuffers), c_int, c_void_p]),
    "imx_terminations_rewards_rollout": (c_int, [c_void_p, c_int64, POINTER(ImxState), POINTER(ImxBuffers), c_int, POINTER(ImxRolloutSlot),
                                                 c_void_p]),
    "imx_observations_kernel_name": (c_char_p, [c_void_p]),
    "imx_orch_part_floats": (c_size_t, [c_int64]),
    "imx_reset_orchestrate": (c_int, [POINTER(ImxOrch), c_void_p]),
    "imx_observations": (c_int, [c_void_p, c_int64, POINTER(ImxState), POINTER(ImxBuffers), c_void_p, c_void_p, c_uint64,
                                 c_int, c_void_p, c_void_p]),
    "imx_root_frame": (c_int, [c_int64, c_void_p, c_void_p, c_void_p, c_float, c_float, c_float, c_void_p, c_void_p,
                               c_void_p, c_void_p]),
    "imx_mesh_create": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_float, POINTER(c_void_p)]),
    "imx_mesh_destroy": (None, [c_void_p]),
    "imx_mesh_info": (c_int, [c_void_p, POINTER(c_int64)]),
    "imx_raycast": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_float, c_void_p, c_void_p, c_void_p, c_void_p]),
    "imx_gae_scratch_bytes": (c_size_t, [c_int64, c_int64]),
    "imx_gae": (c_int, [c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, c_int, c_void_p,
                        c_void_p, c_void_p, c_void_p]),
    "imx_ppo_scratch_bytes": (c_size_t, [c_int64]),
    "imx_ppo_loss_fwd": (c_int, [c_int64, c_int64, c_void_p, c_void_p, c_int64] + [c_void_p] * 8 + [c_float, c_int, c_float, c_float,
                                c_void_p, c_void_p, c_void_p, c_void_p]),
    "imx_ppo_loss_bwd": (c_int, [c_int64, c_int64, c_void_p, c_void_p, c_int64] + [c_void_p] * 6 + [c_float, c_int, c_float, c_float,
                                c_float, c_void_p, c_void_p, c_void_p, c_void_p]),
    "imx_adam_step": (c_int, [c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, c_float,
                              c_float, c_int64, c_void_p]),
    "imx_adam_update": (c_int, [c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_float,
                                c_float, c_float, c_float, c_void_p]),
    "imx_colsum_scratch_bytes": (c_size_t, []),
    "imx_colsum": (c_int, [c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "imx_adam_norm_scratch_bytes": (c_size_t, [c_int64]),
    "imx_adam_update_norm": (c_int, [c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, c_float,
                                     c_float, c_float, c_void_p, c_size_t, c_void_p]),
    "imx_gather_rows": (c_int, [c_int64, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "imx_gather_rows_pitched": (c_int, [c_int64, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "imx_policy_act": (c_int, [c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_uint64, c_void_p] + [c_void_p] * 7
                       + [c_void_p]),
    "imx_rollout_post": (c_int, [c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "imx_contact_sensor_update": (c_int, [c_int64, c_int64, c_int64, c_void_p, c_float, c_float, c_float, c_int] + [c_void_p] * 9
                                  + [c_void_p]),
    "imx_velocity_command": (c_int, [c_int64, c_void_p, c_int, c_float, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_uint64, c_void_p] + [c_void_p] * 8 + [c_void_p]),
    "imx_articulation_update": (c_int, [c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_float] + [c_void_p] * 6 + [c_void_p]),
    "imx_actuator_pd": (c_int, [c_int64, c_int64, c_int, c_float] + [c_void_p] * 11 + [c_void_p]),
    "imx_actuator_delayed_pd": (c_int, [c_int64, c_int64, c_int, c_int64] + [c_void_p] * 12 + [c_int] + [c_void_p] * 2 + [c_void_p]),
    "imx_actuator_net_lstm": (c_int, [c_int64, c_int64, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_int64] + [c_void_p] * 5 + [c_float]
                              + [c_void_p] * 4 + [c_void_p]),
    "imx_actuator_net_mlp": (c_int, [c_int64, c_int64, c_int, c_void_p, c_int, c_void_p, c_int64, c_int, c_void_p, c_int, c_float, c_float,
                                     c_float, c_int] + [c_void_p] * 5 + [c_float] + [c_void_p] * 4 + [c_void_p]),
    "imx_empirical_normalization": (c_int, [c_int64, c_int64, c_void_p, c_int, c_float] + [c_void_p] * 5 + [c_void_p]),
    "imx_reset_events": (c_int, [c_int64, c_int64, c_void_p, c_void_p, c_int] + [c_void_p] * 7 + [c_uint64, c_void_p] + [c_void_p] * 4
                         + [c_void_p]),
    "imx_push_velocity": (c_int, [c_int64, c_void_p, c_void_p, c_void_p, c_uint64, c_void_p, c_void_p, c_void_p]),
    "imx_external_force_torque": (c_int, [c_int64, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_uint64, c_void_p, c_void_p,
                                         c_void_p, c_void_p]),
    "imx_terrain_levels": (c_int, [c_int64, c_int64, c_int64] + [c_void_p] * 5 + [c_float, c_float, c_void_p, c_uint64, c_void_p]
                           + [c_void_p] * 3 + [c_void_p]),
    "imx_mlp_scratch_bytes": (c_size_t, [c_int64, c_int, c_int]),
    "imx_mlp_dw": (c_int, [c_int64, c_int, c_int, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "imx_mlp_set_dw_cu_budget": (c_int, [c_int]),
    "imx_reduce_batch_create": (c_int, [POINTER(c_void_p)]),
    "imx_reduce_batch_destroy": (None, [c_void_p]),
    "imx_reduce_batch_begin": (c_int, [c_void_p]),
    "imx_reduce_batch_flush": (c_int, [c_void_p, c_void_p]),
    "imx_mlp_dw_elu": (c_int, [c_int64, c_int, c_int, c_void_p, c_int64, c_void_p, c_int64, c_float, c_void_p, c_int64, c_void_p, c_int64,
                               c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "imx_mlp_infer": (c_int, [c_int64, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                              c_void_p]),
    "imx_mlp_infer_act": (c_int, [c_int64, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                  POINTER(ImxPolicyAct), c_void_p]),
    "imx_mlp_packed_floats": (c_size_t, [c_int, c_int]),
    "imx_mlp_pack_weights": (c_int, [c_int, c_int, c_void_p, c_int64, c_void_p, c_void_p]),
    "imx_mlp_pack_weights_batch": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "imx_mlp_fwd_elu": (c_int, [c_int64, c_int, c_int, c_void_p, c_int64, c_void_p, c_void_p, c_float, c_int, c_void_p, c_int64, c_void_p]),
    "imx_mlp_head_fwd_bwd": (c_int, [c_int64, c_int, c_int, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int, c_float, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),
    "imx_mlp_head_fwd": (c_int, [c_int64, c_int, c_int, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int, c_float, c_void_p]),
    "imx_mlp_head_fwd_loss": (c_int, [c_int64, c_int, c_int, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int, c_float, c_void_p, c_void_p]),
    "imx_mlp_head_bwd": (c_int, [c_int64, c_int, c_int, c_void_p, c_void_p, c_int64, c_void_p, c_float, c_int, c_void_p, c_void_p,
                                 c_void_p, c_void_p, c_size_t, c_void_p]),
}

EXPORTS = tuple(_SIGNATURES)


def lib():
    """Load libimx.so (building it when hipcc is available and the library is stale or missing)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        from . import build as _build

        _build.build()
    # torch first: libimx.so must bind to the HIP runtime torch ships and has (or will have) initialised -- loaded on its own it pulls
    # /opt/rocm's libamdhip64 in, and a process with two HIP runtimes sees no GPU from the second one ("no GPU visible")
    import torch  # noqa: F401

    try:
        L = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # fail loudly: never substitute another implementation
        raise ImxError(f"cannot load {LIB_PATH}: {e}. Run `python -m isaaclab_amd.build` (hipcc, gfx950).") from e
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    for which, cls in enumerate((ImxState, ImxBuffers, ImxHeadLoss, ImxRolloutSlot, ImxPolicyAct, ImxOrch, ImxEventTerm)):  # the binding's struct layouts against the library's
        if int(L.imx_struct_size(which)) != ctypes.sizeof(cls):
            raise ImxError(f"{LIB_PATH}: sizeof({cls.__name__}) is {int(L.imx_struct_size(which))} in the library, {ctypes.sizeof(cls)} in the "
                           "binding -- rebuild with `python -m isaaclab_amd.build`")
    _lib = L
    return L


def check(status: int):
    if status != 0:
        raise ImxError(lib().imx_last_error().decode())


def ptr(t) -> int | None:
    """Device/host pointer of a contiguous torch tensor (None stays NULL)."""
    if t is None:
        return None
    if not t.is_contiguous():
        raise ImxError("libimx needs contiguous tensors")
    return t.data_ptr()


def current_stream(device) -> int:
    import torch

    if device.type != "cuda":
        raise ImxError("libimx kernels only run on a GPU (device=%s)" % device)
    return torch.cuda.current_stream(device).cuda_stream
