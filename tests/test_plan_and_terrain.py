"""CPU: host logic -- name resolution, grid pattern, height-field mesh (pinned by reference fixtures), term compiler."""

import os

import numpy as np
import pytest

from _util import GOLDEN, TASKS, Golden
from isaaclab_amd import plan as planmod
from isaaclab_amd.robots import ANYMAL_C, G1, resolve_matching_names
from isaaclab_amd.terrain import height_field_to_mesh, make_rough_terrain


def test_resolve_matching_names_semantics():
    # examples from the reference docstring (isaaclab/utils/string.py:178-271)
    names = ["a", "b", "c", "d", "e"]
    assert resolve_matching_names(["a|c", "b"], names) == ([0, 1, 2], ["a", "b", "c"])
    assert resolve_matching_names(["a|c", "b"], names, preserve_order=True) == ([0, 2, 1], ["a", "c", "b"])
    with pytest.raises(ValueError):
        resolve_matching_names(["a", "a|b"], names)  # multiple matches
    with pytest.raises(ValueError):
        resolve_matching_names(["zzz"], names)  # unmatched key
    assert resolve_matching_names(".*FOOT", ANYMAL_C.body_names)[0] == [13, 14, 15, 16]
    assert len(G1.joint_names) == 37 and len(set(G1.joint_names)) == 37


def test_grid_pattern_matches_reference():
    z = np.load(os.path.join(GOLDEN, "hf_mesh.npz"))
    s, d = planmod.grid_pattern(0.1, [1.6, 1.0], (0.0, 0.0, -1.0), "xy")
    assert s.shape == (187, 3)
    # torch.arange(float32) evaluates vector lanes as fp32(base) + k*fp32(step): values depend on the SIMD width of
    # the machine that ran the reference and differ from the correctly rounded start + i*step by <= 1 ulp (1.5e-8 m)
    assert np.abs(s - z["grid_xy_starts"]).max() <= 3e-8 and np.array_equal(d, z["grid_xy_dirs"])
    s, d = planmod.grid_pattern(0.25, [1.0, 0.5], (0.0, 0.0, -1.0), "yx")
    assert np.abs(s - z["grid_yx_starts"]).max() <= 3e-8 and np.array_equal(d, z["grid_yx_dirs"])
    with pytest.raises(ValueError):
        planmod.grid_pattern(0.1, [1, 1], ordering="zz")
    with pytest.raises(ValueError):
        planmod.grid_pattern(0.0, [1, 1])


def test_height_field_mesh_matches_reference():
    z = np.load(os.path.join(GOLDEN, "hf_mesh.npz"))
    for thr, key in ((None, "none"), (0.75, "thr")):
        v, t = height_field_to_mesh(z["hf"], 0.1, 0.005, thr)
        assert np.array_equal(v, z[f"v_{key}"])
        assert np.array_equal(t.astype(np.int64), z[f"t_{key}"])


def test_rough_terrain_is_seeded_and_sized():
    v1, t1, ext = make_rough_terrain(2, 2, tile=4.0, border=2.0, seed=5)
    v2, t2, _ = make_rough_terrain(2, 2, tile=4.0, border=2.0, seed=5)
    assert np.array_equal(v1, v2) and np.array_equal(t1, t2)
    assert ext == (4.0, 4.0) and t1.max() < len(v1) and t1.shape[1] == 3
    # composition of ROUGH_TERRAINS_CFG: mesh stairs / boxes (12 triangles per box) + height-field tiles
    v3, t3, _ = make_rough_terrain(10, 20, seed=0)
    assert 1.0e6 < len(t3) < 1.4e6  # ~1.1-1.3 M triangles


@pytest.mark.parametrize("task", TASKS)
def test_term_compiler_matches_reference_term_tables(task):
    g = Golden(task)
    p = planmod.compile_plan(g.fixture["env"], g.robot)
    assert [t.name for t in p.reward_terms] == g.meta["reward_terms"]
    assert [t.name for t in p.termination_terms] == g.meta["termination_terms"]
    assert [t.name for t in p.obs_terms] == g.meta["obs_terms"]
    assert [list(d) for d in p.obs_term_dims] == g.meta["obs_term_dims"]
    assert p.obs_dim == g.meta["obs_dim"] and p.action_dim == g.meta["action_dim"]
    assert p.max_episode_length == g.meta["max_episode_length"]
    assert abs(p.step_dt - g.meta["step_dt"]) < 1e-12
    assert p.n_ext_rew == p.n_ext_term == p.n_ext_obs == 0  # everything in the target configs is fused


def test_group_shapes_follow_the_reference():
    """group_obs_term_dim / concatenate flags of a cfg with a dict-of-terms group and un-flattened history, against what the reference's
    ObservationManager reported when the fixture was generated; terms of different rank cannot be concatenated (observation_manager.py:89-99)."""
    import copy

    from _util import SHAPES

    g = Golden(SHAPES)
    p = planmod.compile_plan(g.fixture["env"], g.robot)
    assert [gr.name for gr in p.obs_groups] == g.meta["obs_groups"]
    for gr in p.obs_groups:
        assert [list(d) for d in gr.term_dims] == g.meta["obs_group_term_shapes"][gr.name]
        assert gr.concatenate == g.meta["obs_group_concatenate"][gr.name] and gr.dim == sum(gr.term_widths)
        assert [t.name for t in gr.terms] == g.meta["obs_group_terms"][gr.name]
    assert [gr.dim for gr in p.obs_groups] == g.meta["obs_group_dims"]
    assert [gr.is_flat for gr in p.obs_groups] == [True, False, False]
    bad = copy.deepcopy(g.fixture["env"])
    bad["observations"]["terms"]["concatenate_terms"] = True  # (2, 3) next to (12,)
    with pytest.raises(RuntimeError, match="Unable to concatenate observation terms in group 'terms'"):
        planmod.compile_plan(bad, g.robot)


def test_term_compiler_errors_follow_the_reference():
    g = Golden("Isaac-Velocity-Flat-Anymal-C-v0")
    import copy

    cfg = copy.deepcopy(g.fixture["env"])
    cfg["rewards"]["undesired_contacts"]["params"]["sensor_cfg"]["body_names"] = ".*ELBOW"
    with pytest.raises(ValueError):
        planmod.compile_plan(cfg, g.robot)  # regex matches nothing
    cfg = copy.deepcopy(g.fixture["env"])
    cfg["rewards"]["lin_vel_z_l2"]["weight"] = "heavy"
    with pytest.raises(TypeError):
        planmod.compile_plan(cfg, g.robot)  # reward_manager.py:231-236
    cfg = copy.deepcopy(g.fixture["env"])
    cfg["rewards"]["custom"] = {"func": "my_pkg.mdp:my_reward", "params": {}, "weight": 1.0}
    p = planmod.compile_plan(cfg, g.robot)
    assert p.n_ext_rew == 1 and p.reward_terms[-1].external == "my_pkg.mdp:my_reward"
    # zero-weight terms keep their slot AND their record (plan v3): the kernel skips them at run time (reward_manager.py:145) and
    # set_term_cfg can switch them on without changing the plan's shape
    assert p.blob[planmod.H["NREW"]] == p.blob[planmod.H["NREW_ALL"]] == len(p.reward_terms)
    zero = [i for i, t in enumerate(p.reward_terms) if t.weight == 0.0]
    assert zero and all(p.blob[p.blob[planmod.H["REW_OFF"]] + i * planmod.REC_WORDS + planmod.R["WEIGHT"]] == 0 for i in zero)


def test_observation_modifiers_compile_to_programs_and_the_library_validates_them():
    """ObservationTermCfg.modifiers (utils/modifiers/modifier.py): program encoding, state sizing, and plan validation in the
    C library (loads without a GPU; no compute)."""
    import copy
    import ctypes

    import numpy as np

    from isaaclab_amd._lib import check, lib, ImxError

    g = Golden("Isaac-Velocity-Flat-Anymal-C-v0-mod")
    p = planmod.compile_plan(g.fixture["env"], g.robot)
    # unit delay 3 x (2+1), integrator 12 x 2, IIR 12 x (3+2), low-pass 12 x (1+1)
    assert p.mod_state_dim == 9 + 24 + 60 + 24 and p.n_ext_obs == 0
    prog, slots = planmod.compile_modifiers(g.fixture["env"]["observations"]["policy"]["base_lin_vel"]["modifiers"])
    assert slots == 0 and prog[0::4] == [planmod.M_OPS["SCALE"], planmod.M_OPS["BIAS"], planmod.M_OPS["CLIP"]]
    assert prog[10] == planmod._f2w(float("inf"))  # clip upper bound None
    prog, slots = planmod.compile_modifiers(g.fixture["env"]["observations"]["policy"]["joint_vel"]["modifiers"])
    assert slots == 5 and prog[:4] == [planmod.M_OPS["DIGITAL_FILTER"], 2, 3, 0] and len(prog) == 4 + 5 + 4
    L = lib()

    def create(blob):
        blob = np.ascontiguousarray(blob, np.int32)
        h = ctypes.c_void_p()
        check(L.imx_plan_create(blob.ctypes.data, blob.size, ctypes.byref(h)))
        L.imx_plan_destroy(h)

    create(p.blob)
    bad = p.blob.copy()
    bad[planmod.H["MOD_STATE"]] -= 1  # last term's state no longer fits the row
    with pytest.raises(ImxError, match="modifier state"):
        create(bad)
    cfg = copy.deepcopy(g.fixture["env"])
    cfg["observations"]["policy"]["joint_vel"]["modifiers"][0]["A"] = None
    with pytest.raises(ValueError, match="coefficients A and B"):
        planmod.compile_plan(cfg, g.robot)  # modifier.py:131-132
    cfg = copy.deepcopy(g.fixture["env"])
    cfg["observations"]["policy"]["joint_vel"]["modifiers"] = [{"func": "my_pkg.mods:wobble", "params": {}}]
    with pytest.raises(NotImplementedError, match="_dim"):
        planmod.compile_plan(cfg, g.robot)  # unknown modifier: the term (function + modifier chain) has to be evaluated in Python
    cfg["observations"]["policy"]["joint_vel"]["_dim"] = 12
    p2 = planmod.compile_plan(cfg, g.robot)
    t = [t for t in p2.obs_terms if t.name == "joint_vel"][0]
    assert p2.n_ext_obs == 12 and t.external is not None and [m[0] for m in t.py_modifiers] == ["my_pkg.mods:wobble"]
    cfg["observations"]["policy"]["joint_vel"]["modifiers"].append({"func": "isaaclab.utils.modifiers.modifier:DigitalFilter", "A": [0.0], "B": [1.0]})
    with pytest.raises(NotImplementedError, match="class-based"):
        planmod.compile_plan(cfg, g.robot)  # a stateful modifier class behind a foreign one cannot run anywhere
    cfg = copy.deepcopy(g.fixture["env"])
    cfg["observations"]["policy"]["joint_vel"]["noise"] = {"func": "isaaclab.utils.noise.noise_model:gaussian_noise", "mean": 0.0, "std": 1.0, "operation": "add"}
    assert planmod.compile_plan(cfg, g.robot).enable_corruption  # gaussian / constant / uniform noise with scalar parameters are fused
    cfg["observations"]["policy"]["joint_vel"]["noise"]["std"] = [1.0] * 12  # per-element parameters are not
    with pytest.raises(NotImplementedError, match="noise model"):
        planmod.compile_plan(cfg, g.robot)  # never dropped silently
    cfg["observations"]["policy"]["joint_vel"]["noise"] = {"func": "my_pkg.noise:pink_noise", "alpha": 1.0}
    with pytest.raises(NotImplementedError, match="noise model"):
        planmod.compile_plan(cfg, g.robot)
    cfg["observations"]["policy"]["joint_vel"]["noise"] = {"func": "isaaclab.utils.noise.noise_model:constant_noise", "bias": 0.1, "operation": "mul"}
    with pytest.raises(ValueError, match="Unknown operation in noise"):
        planmod.compile_plan(cfg, g.robot)  # noise_model.py:38


def test_policy_is_exportable_like_the_reference_exporter(tmp_path):
    """SURVEY 8f row 3, export compatibility: what isaaclab_rl/rsl_rl/exporter.py::_TorchPolicyExporter does with a policy
    (deepcopy of ``policy.actor``, ``policy.is_recurrent``, normalizer in front, torch.jit.script, save / load) works on this
    ActorCritic after its parameters were re-homed into the flat bucket, and the normalizer module scripts too."""
    import copy

    import torch

    from isaaclab_amd.rsl_rl.actor_critic import ActorCritic
    from isaaclab_amd.rsl_rl.ppo import FlatParams

    torch.manual_seed(0)
    pol = ActorCritic(48, 48, 12, actor_hidden_dims=[128, 128, 128], critic_hidden_dims=[128, 128, 128])
    FlatParams(pol)  # parameters become views of one bucket, as inside PPO
    assert not pol.is_recurrent and hasattr(pol, "actor")

    class Exporter(torch.nn.Module):  # same structure as the reference's exporter module (non-recurrent branch)
        def __init__(self, policy):
            super().__init__()
            self.actor = copy.deepcopy(policy.actor)
            self.normalizer = torch.nn.Identity()

        def forward(self, x):
            return self.actor(self.normalizer(x))

    ex = Exporter(pol)
    ex.to("cpu")
    scripted = torch.jit.script(ex)
    path = str(tmp_path / "policy.pt")
    scripted.save(path)
    loaded = torch.jit.load(path)
    x = torch.randn(5, 48)
    assert torch.equal(loaded(x), pol.actor(x)) and torch.equal(pol.act_inference(x), pol.actor(x))


_LIVE_CFGS = {  # task id -> (module, class) of the reference cfg its gym registry entry names (isaaclab_tasks/.../__init__.py)
    "Isaac-Cartpole-v0": ("isaaclab_tasks.manager_based.classic.cartpole.cartpole_env_cfg", "CartpoleEnvCfg"),
    "Isaac-Velocity-Flat-Anymal-C-v0": ("isaaclab_tasks.manager_based.locomotion.velocity.config.anymal_c.flat_env_cfg", "AnymalCFlatEnvCfg"),
    "Isaac-Velocity-Rough-Anymal-C-v0": ("isaaclab_tasks.manager_based.locomotion.velocity.config.anymal_c.rough_env_cfg", "AnymalCRoughEnvCfg"),
    "Isaac-Velocity-Rough-G1-v0": ("isaaclab_tasks.manager_based.locomotion.velocity.config.g1.rough_env_cfg", "G1RoughEnvCfg"),
}

_LIVE_SCRIPT = r"""
import importlib, json, sys
import numpy as np
sys.path.insert(0, {root!r})
from oracle import ref_import
ref_import.install()
from isaaclab_amd.env import load_task_cfg
from isaaclab_amd.plan import compile_plan
from isaaclab_amd.robots import ROBOTS
out = {{}}
for task, (mod, cls) in {cfgs!r}.items():
    cfg = getattr(importlib.import_module(mod), cls)()        # the LIVE reference cfg object (configclass instance)
    fx = load_task_cfg(task)                                   # the committed cfg.to_dict() dump
    robot = ROBOTS[fx["robot"]]
    live = compile_plan(cfg, robot)
    dump = compile_plan(fx["env"], robot)
    out[task] = bool(np.array_equal(np.asarray(live.blob), np.asarray(dump.blob))) and live.obs_dim == dump.obs_dim
print("RESULT " + json.dumps(out))
"""


def test_live_reference_cfg_objects_compile_to_the_same_plan():
    """'Configs drop in unchanged' (isaaclab_tasks/utils/parse_cfg.py:20-100): the reference's own cfg OBJECTS -- instantiated from
    /root/reference with the simulator packages stubbed, in a child process so the stubs do not leak into this one -- compile to the
    very plan blob the committed JSON dumps give.  Skipped where the reference is absent (the GPU box)."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.isdir("/root/reference/source/isaaclab"):
        pytest.skip("the reference tree is not present")
    r = subprocess.run([sys.executable, "-c", _LIVE_SCRIPT.format(root=root, cfgs=_LIVE_CFGS)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1]
    res = json.loads(line[len("RESULT "):])
    assert set(res) == set(_LIVE_CFGS) and all(res.values()), res


def test_terrain_importer_state_follows_the_reference_layout():
    """``TerrainImporterState.from_generator_cfg`` = ``TerrainImporter._compute_env_origins_curriculum`` (terrains/terrain_importer.py:328-347):
    types = floor(arange(N) / (N / num_cols)), levels in [0, max_init_terrain_level], env_origins = origins[level, type]; sub-terrain origins
    at the tile centres of the generator grid."""
    import torch

    from isaaclab_amd.events import TerrainImporterState

    cfg = {"num_rows": 10, "num_cols": 20, "size": [8.0, 8.0]}
    ti = TerrainImporterState.from_generator_cfg(4096, cfg, "cpu", max_init_terrain_level=5)
    assert ti.terrain_origins.shape == (10, 20, 3) and ti.max_terrain_level == 10 and ti.size_x == 8.0
    assert torch.equal(ti.terrain_types, torch.div(torch.arange(4096), 4096 / 20, rounding_mode="floor").long())
    assert int(ti.terrain_levels.min()) >= 0 and int(ti.terrain_levels.max()) <= 5
    assert torch.equal(ti.env_origins, ti.terrain_origins[ti.terrain_levels, ti.terrain_types])
    assert torch.allclose(ti.terrain_origins[0, 0, :2], torch.tensor([-36.0, -76.0])) and torch.allclose(ti.terrain_origins[9, 19, :2], torch.tensor([36.0, 76.0]))


def test_task_cfg_merges_the_managers_side_file():
    """The headline task's events / curriculum / robot init state travel in ``<task>.managers.json`` (dumped from the UNMODIFIED reference cfg by
    oracle/gen_golden_orchestration.py): ``load_task_cfg`` merges it, other tasks are untouched."""
    from isaaclab_amd.env import load_task_cfg

    fx = load_task_cfg("Isaac-Velocity-Rough-Anymal-C-v0")["env"]
    assert set(fx["events"]) == {"base_external_force_torque", "reset_base", "reset_robot_joints", "push_robot"}
    assert fx["events"]["push_robot"]["interval_range_s"] == [10.0, 15.0] and fx["events"]["push_robot"]["mode"] == "interval"
    assert list(fx["curriculum"]) == ["terrain_levels"] and fx["scene"]["robot"]["init_state"]["pos"] == [0.0, 0.0, 0.6]
    assert "events" not in load_task_cfg("Isaac-Cartpole-v0")["env"]
