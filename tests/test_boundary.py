"""CPU: the C-ABI library loads, exports every symbol include/imx.h declares, validates plans on the host, and
the product package never reaches into oracle/."""

import ctypes
import os
import re

import numpy as np
import pytest

from _util import KITCHEN, TASKS
from isaaclab_amd import _lib, plan as planmod
from isaaclab_amd.env import load_task_cfg
from isaaclab_amd.robots import ROBOTS

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = open(os.path.join(ROOT, "include", "imx.h")).read()


def declared_functions():
    names = re.findall(r"^\s*(?:const\s+)?[A-Za-z_][\w\s\*]*?\b(imx_[a-z0-9_]+)\s*\(", HEADER, flags=re.M)
    return sorted(set(n for n in names if not n.endswith("_t")))


def test_library_exports_every_declared_symbol(libimx):
    decl = declared_functions()
    assert len(decl) >= 20
    for name in decl:
        assert hasattr(libimx, name), f"{name} declared in include/imx.h but not exported by libimx.so"
    assert set(decl) == set(_lib.EXPORTS), set(decl) ^ set(_lib.EXPORTS)
    assert b"gfx950" in libimx.imx_version()


def test_header_constants_match_plan_compiler():
    def define(name):
        return int(re.search(rf"#define {name}\s+(\S+)", HEADER).group(1), 0)

    assert define("IMX_MAGIC") == planmod.MAGIC
    assert define("IMX_PLAN_VERSION") == planmod.PLAN_VERSION
    assert define("IMX_HEADER_WORDS") == planmod.HEADER_WORDS
    assert define("IMX_REC_WORDS") == planmod.REC_WORDS

    def enum(name):
        body = re.search(rf"enum {name}\s*\{{(.*?)\}};", HEADER, flags=re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        out, nxt = {}, 0
        for item in [x.strip() for x in body.split(",") if x.strip()]:
            if "=" in item:
                k, v = [s.strip() for s in item.split("=")]
                nxt = int(v, 0)
            else:
                k = item
            out[k] = nxt
            nxt += 1
        return out

    for prefix, table, en in (("IMX_T_", planmod.T_OPS, "imx_term_op"), ("IMX_W_", planmod.W_OPS, "imx_rew_op"),
                              ("IMX_O_", planmod.O_OPS, "imx_obs_op"), ("IMX_H_", planmod.H, "imx_header_word"),
                              ("IMX_R_", planmod.R, "imx_rec_word")):
        e = enum(en)
        for k, v in table.items():
            assert e[prefix + k] == v, (prefix + k, e[prefix + k], v)
    # struct field order
    for struct, fields in (("imx_state", _lib.STATE_FIELDS), ("imx_buffers", _lib.BUFFER_FIELDS)):
        body = re.search(rf"typedef struct {struct} \{{(.*?)\}}", HEADER, flags=re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        names = re.findall(r"[\*\s]([a-z_0-9]+)\s*;", body)  # pointer and scalar members
        assert tuple(names) == tuple(fields), struct


@pytest.mark.parametrize("task", TASKS + (KITCHEN,))
def test_plan_blob_validates_through_the_c_abi(libimx, task):
    fx = load_task_cfg(task)
    p = planmod.compile_plan(fx["env"], ROBOTS[fx["robot"]])
    blob = np.ascontiguousarray(p.blob, np.int32)
    h = ctypes.c_void_p()
    rc = libimx.imx_plan_create(blob.ctypes.data, blob.size, ctypes.byref(h))
    assert rc == 0, libimx.imx_last_error()
    assert libimx.imx_plan_obs_dim(h) == p.obs_dim
    assert libimx.imx_plan_scratch_bytes(h, 4096) > 0
    libimx.imx_plan_destroy(h)


def test_plan_update_in_place_accepts_same_shape_only(libimx):
    """imx_plan_update = RewardManager/TerminationManager.set_term_cfg: a recompiled blob of the same shape replaces the tables, a
    different shape (another observation width) is refused."""
    import copy

    fx = load_task_cfg("Isaac-Velocity-Flat-Anymal-C-v0")
    robot = ROBOTS[fx["robot"]]
    p = planmod.compile_plan(fx["env"], robot)
    blob = np.ascontiguousarray(p.blob, np.int32)
    h = ctypes.c_void_p()
    assert libimx.imx_plan_create(blob.ctypes.data, blob.size, ctypes.byref(h)) == 0
    cfg = copy.deepcopy(fx["env"])
    cfg["rewards"]["flat_orientation_l2"]["weight"] = 0.0  # non-zero -> zero: same shape in plan v3
    cfg["rewards"]["dof_pos_limits"]["weight"] = -1.0      # zero -> non-zero
    cfg["terminations"]["base_contact"]["params"]["threshold"] = 2.0
    b2 = np.ascontiguousarray(planmod.compile_plan(cfg, robot).blob, np.int32)
    assert b2.size == blob.size
    assert libimx.imx_plan_update(h, b2.ctypes.data, b2.size, None) == 0, libimx.imx_last_error()
    del cfg["observations"]["policy"]["actions"]
    b3 = np.ascontiguousarray(planmod.compile_plan(cfg, robot).blob, np.int32)
    assert libimx.imx_plan_update(h, b3.ctypes.data, b3.size, None) != 0 and b"different shape" in libimx.imx_last_error()
    libimx.imx_plan_destroy(h)


def test_plan_rejects_corrupt_blobs(libimx):
    fx = load_task_cfg("Isaac-Velocity-Flat-Anymal-C-v0")
    p = planmod.compile_plan(fx["env"], ROBOTS[fx["robot"]])
    h = ctypes.c_void_p()

    def create(b):
        b = np.ascontiguousarray(b, np.int32)
        return libimx.imx_plan_create(b.ctypes.data, b.size, ctypes.byref(h))

    bad = p.blob.copy(); bad[planmod.H["MAGIC"]] = 0
    assert create(bad) != 0 and b"magic" in libimx.imx_last_error()
    bad = p.blob.copy(); bad[planmod.H["TOTAL_WORDS"]] += 1
    assert create(bad) != 0
    assert create(p.blob[:10]) != 0
    # a joint index outside [0, J)
    bad = p.blob.copy()
    rec = bad[planmod.H["REW_OFF"]] + 4 * planmod.REC_WORDS  # dof_torques_l2
    bad[bad[rec + planmod.R["IDS_OFF"]]] = 99
    assert create(bad) != 0 and b"outside" in libimx.imx_last_error()
    # an observation column written twice
    bad = p.blob.copy()
    rec = bad[planmod.H["OBS_OFF"]] + 1 * planmod.REC_WORDS
    bad[rec + planmod.R["OUT"]] = 0
    assert create(bad) != 0


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "isaaclab_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                if f.endswith(".py"):  # nothing in the product may read the reference at run time
                    assert "/root/reference" not in src, f


def test_env_fails_loudly_without_a_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from isaaclab_amd.env import ManagerBasedRLEnv

    with pytest.raises(_lib.ImxError):
        ManagerBasedRLEnv("Isaac-Cartpole-v0", num_envs=8)


def test_runner_refuses_policy_and_algorithm_classes_it_does_not_build():
    """isaaclab_rl/rsl_rl/rl_cfg.py:22,86-99,166: ActorCriticRecurrent / ActorCriticCascade / PPOCA / Distillation cfgs must not train
    a plain feed-forward PPO silently."""
    from isaaclab_amd.env import load_task_cfg
    from isaaclab_amd.rsl_rl import OnPolicyRunner
    from isaaclab_amd.rsl_rl.actor_critic import ActorCritic
    from isaaclab_amd.rsl_rl.ppo import PPO

    agent = load_task_cfg("Isaac-Cartpole-v0")["agent"]
    for section, name in (("policy", "ActorCriticRecurrent"), ("policy", "ActorCriticCascade"), ("algorithm", "PPOCA"), ("algorithm", "Distillation")):
        cfg = {**agent, section: {**agent[section], "class_name": name}}
        with pytest.raises(NotImplementedError, match=name):
            OnPolicyRunner(env=None, train_cfg=cfg, device="cpu")  # refused before the env is touched
    with pytest.raises(NotImplementedError, match="rnn_type"):
        ActorCritic(4, 4, 1, rnn_type="lstm", rnn_hidden_dim=64, rnn_num_layers=1)
    pol = ActorCritic(4, 4, 1, actor_hidden_dims=[8], critic_hidden_dims=[8])
    with pytest.raises(NotImplementedError, match="teacher_coef"):
        PPO(pol, teacher_coef=1.0)


def test_seed_is_callable_on_the_class_like_the_reference_staticmethod():
    """envs/manager_based_env.py:425-443: ``ManagerBasedEnv.seed`` is a @staticmethod."""
    import torch

    from isaaclab_amd.env import ManagerBasedRLEnv

    assert ManagerBasedRLEnv.seed(42) == 42
    a = torch.rand(3)
    assert ManagerBasedRLEnv.seed(42) == 42
    assert torch.equal(a, torch.rand(3))


def test_only_observation_minibatch_buffers_carry_a_row_pitch():
    """The minibatch buffers of the update: observation rows start on 16-byte boundaries (their consumers -- GEMMs, imx_mlp_dw,
    imx_mlp_fwd_elu -- take a row pitch), every per-sample array (actions, old mu / sigma, values ...) is dense: the loss kernels index
    them as (M, A) without a pitch.  Padding those too (round 2 until the end of round 3) fed the update wrong columns for action counts
    >= 16 that are not a multiple of four (G1: 37)."""
    from isaaclab_amd.rsl_rl.storage import RolloutStorage

    for D, Dc, A in ((310, 0, 37), (235, 187, 17), (48, 0, 12), (37, 53, 18)):
        st = RolloutStorage(8, 4, (D,), (Dc,), (A,), device="cpu")
        st._minibatch_setup(2)
        n_obs = 2 if Dc else 1
        for dst in st._mb_dst_sets:
            for k, buf in enumerate(dst):
                if k < n_obs and buf.shape[1] >= 16:
                    assert buf.stride(0) % 4 == 0 and buf.stride(0) - buf.shape[1] < 4, (D, Dc, A, k, buf.stride())
                else:
                    assert buf.is_contiguous(), (D, Dc, A, k, tuple(buf.shape), buf.stride())
