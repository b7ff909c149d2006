"""Random env cfgs through the term compiler and the HIP env step, against the CPU oracle consuming the same cfg: the rough-terrain
Anymal-C scene with random subsets (order kept) of the reward / termination / observation terms of the kitchen-sink fixture (every mdp
term the path knows, two observation groups), random weights (incl. 0), scale / clip / uniform, gaussian and constant noise (add, scale, abs), modifier chains (scale, bias, clip,
DigitalFilter, Integrator), per-term and
per-group history, episode length, the six joint action classes in random combinations (processed actions compared).  Masks / ids bit-exact, floats 1e-5.  Test infrastructure, run on
the GPU box:  python tools/fuzz_cfg.py [cases] [first_seed]"""
import copy
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

from _util import FLOAT_TOL, assert_close
from isaaclab_amd.env import ManagerBasedRLEnv
from isaaclab_amd.robots import ROBOTS
from isaaclab_amd.state_feed import StateFeed
from isaaclab_amd.terrain import make_rough_terrain
from oracle.mdp_oracle import OracleEnv

CFG = os.path.join(ROOT, "isaaclab_amd", "configs")
BASE = json.load(open(os.path.join(CFG, "Isaac-Velocity-Rough-Anymal-C-v0.json")))
POOL = json.load(open(os.path.join(CFG, "Isaac-Velocity-Rough-Anymal-C-v0-kitchen.json")))["env"]
NOISE = "isaaclab.utils.noise.noise_model:uniform_noise"
ACTIONS = json.load(open(os.path.join(CFG, "Isaac-Velocity-Flat-Anymal-C-v0-actions.json")))["env"]["actions"]


def mutate(rng):
    fx = copy.deepcopy(BASE)
    env = fx["env"]
    what = []
    names = list(POOL["terminations"])
    keep_t = [n for n in names if n == "time_out" or rng.random() < 0.5]
    env["terminations"] = {n: copy.deepcopy(POOL["terminations"][n]) for n in keep_t}
    names = list(POOL["rewards"])
    keep = [n for n in names if rng.random() < 0.6] or ["alive"]
    env["rewards"] = {}
    for n in keep:
        t = copy.deepcopy(POOL["rewards"][n])
        tk = (t.get("params") or {}).get("term_keys")
        if tk is not None and any(k not in env["terminations"] for k in ([tk] if isinstance(tk, str) else tk)):
            continue
        if rng.random() < 0.4:
            t["weight"] = float(rng.choice([0.0, 1.0, -0.5, 2.5e-5, -3.0]))
        env["rewards"][n] = t
    if not env["rewards"]:
        env["rewards"] = {"alive": copy.deepcopy(POOL["rewards"]["alive"])}
    what.append(f"{len(env['rewards'])} rewards, {len(env['terminations'])} terminations")
    env["observations"] = {}
    for gname in (["policy", "critic"] if rng.random() < 0.5 else ["policy"]):
        src = POOL["observations"][gname]
        terms = [k for k, v in src.items() if isinstance(v, dict)]
        keep_o = [k for k in terms if rng.random() < 0.6] or [terms[int(rng.integers(0, len(terms)))]]
        grp = {k: copy.deepcopy(v) for k, v in src.items() if not isinstance(v, dict)}
        grp["enable_corruption"] = bool(rng.random() < 0.6)
        if rng.random() < 0.3:
            grp["history_length"] = int(rng.choice([2, 3]))
            grp["flatten_history_dim"] = True
        for k in keep_o:
            t = copy.deepcopy(src[k])
            if rng.random() < 0.25:
                t["scale"] = float(rng.choice([0.25, 2.0, -1.0]))
            if rng.random() < 0.25:
                lo = -float(rng.choice([0.5, 1.0, 3.0]))
                t["clip"] = [lo, -lo * float(rng.choice([1.0, 0.5]))]
            r = rng.random()
            op = str(rng.choice(["add", "add", "scale", "abs"]))
            if r < 0.25:
                a = float(rng.choice([0.01, 0.1, 0.5]))
                t["noise"] = {"func": NOISE, "operation": op, "n_min": -a, "n_max": a}
            elif r < 0.35:
                t["noise"] = {"func": NOISE.replace("uniform_noise", "gaussian_noise"), "operation": op, "mean": float(rng.choice([0.0, 0.01, 1.0])), "std": float(rng.choice([0.05, 0.3]))}
            elif r < 0.42:
                t["noise"] = {"func": NOISE.replace("uniform_noise", "constant_noise"), "operation": op, "bias": float(rng.choice([0.05, 0.9, 0.25]))}
            elif r < 0.55:
                t["noise"] = None
            if rng.random() < 0.25:  # a chain of modifiers (isaaclab.utils.modifiers): stateless ones and the two stateful classes
                M = "isaaclab.utils.modifiers.modifier:"
                chain = []
                for _ in range(int(rng.integers(1, 4))):
                    kind = str(rng.choice(["scale", "bias", "clip", "clip1", "DigitalFilter", "Integrator"]))
                    if kind == "scale":
                        chain.append({"func": M + "scale", "params": {"multiplier": float(rng.choice([2.0, 0.5, -1.0]))}})
                    elif kind == "bias":
                        chain.append({"func": M + "bias", "params": {"value": float(rng.choice([0.25, -0.1]))}})
                    elif kind == "clip":
                        chain.append({"func": M + "clip", "params": {"bounds": [-float(rng.choice([0.01, 0.8])), float(rng.choice([0.015, 1.0]))]}})
                    elif kind == "clip1":
                        chain.append({"func": M + "clip", "params": {"bounds": [-0.8, None] if rng.random() < 0.5 else [None, 0.5]}})
                    elif kind == "DigitalFilter":
                        na, nb = int(rng.integers(1, 3)), int(rng.integers(1, 4))
                        chain.append({"func": M + "DigitalFilter", "params": {}, "A": [float(x) for x in rng.uniform(-0.5, 0.6, na).round(2)],
                                      "B": [float(x) for x in rng.uniform(0.0, 1.0, nb).round(2)]})
                    else:
                        chain.append({"func": M + "Integrator", "params": {}, "dt": float(rng.choice([0.02, 0.005]))})
                t["modifiers"] = chain
            if grp.get("history_length") is None and rng.random() < 0.2:
                t["history_length"] = int(rng.choice([2, 3]))
                t["flatten_history_dim"] = True
            grp[k] = t
        env["observations"][gname] = grp
        what.append(f"{gname}: {len(keep_o)} terms, group history {grp.get('history_length')}, corruption {grp['enable_corruption']}")
    env["episode_length_s"] = float(rng.choice([20.0, 5.0, 0.5]))
    # (decimation stays: with a shorter env step the height scanner's update_period gates its refresh -- SensorBase, reproduced by the
    #  product and pinned by the kitchen fixture -- while this harness hands the oracle every step's hits)
    if rng.random() < 0.6:
        act = env["actions"]["joint_pos"]
        act["scale"] = float(rng.choice([0.5, 0.25, 1.0]))
        if rng.random() < 0.3:
            act["clip"] = {".*": [-1.0, 1.0]}
        what.append("JointPositionAction")
    else:  # a random combination of the other joint action classes (the `-actions` fixture's terms with other numbers)
        pool = copy.deepcopy(ACTIONS)
        keep_a = [n for n in pool if rng.random() < 0.6] or ["all_ema"]
        env["actions"] = {}
        for n in keep_a:
            a = pool[n]
            if isinstance(a.get("scale"), float):
                a["scale"] = float(rng.choice([0.3, 0.9, 2.0]))
            if "offset" in a and isinstance(a["offset"], float):
                a["offset"] = float(rng.choice([0.0, 0.7, -0.2]))
            if n == "all_ema":
                a["alpha"] = {".*HAA": float(rng.choice([0.3, 1.0])), ".*HFE": float(rng.choice([0.75, 0.1])), ".*KFE": 1.0} if rng.random() < 0.7 else float(rng.choice([0.5, 1.0]))
            if rng.random() < 0.3:
                a["clip"] = None
            env["actions"][n] = a
        what.append("actions " + "+".join(keep_a))
    return fx, "; ".join(what)


def one_case(case_seed: int) -> str:
    rng = np.random.default_rng(case_seed)
    fx, what = mutate(rng)
    N = int(rng.choice([1, 63, 64, 65, 257, 1000, 2049]))
    steps = int(rng.integers(2, 6))
    robot = ROBOTS[fx["robot"]]
    v, t, e = make_rough_terrain(2, 3, tile=8.0, border=5.0, seed=int(rng.integers(0, 100)))
    terrain, ext = (v, t), (e[0] - 1.0, e[1] - 1.0)
    cpu_feed = StateFeed(robot, N, "cpu", seed=int(rng.integers(0, 10000)), num_snapshots=3, extent_xy=ext)
    gpu_feed = StateFeed.from_tensors(robot, [cpu_feed.snapshot(i) for i in range(3)], "cuda:0", cpu_feed.gravity_dir)
    try:
        env = ManagerBasedRLEnv(fx, state_feed=gpu_feed, terrain=terrain, terrain_cell=0.1)
    except NotImplementedError as exc:  # combinations the product refuses by name at construction
        return f"REFUSED ({str(exc)[:90]})"
    env.materialize_ray_hits = True
    orc = OracleEnv(fx["env"], robot.joint_names, robot.body_names, N, cpu_feed.__getitem__, cpu_feed.gravity_dir)
    gen = torch.Generator().manual_seed(int(rng.integers(0, 10000)))
    ep = torch.randint(0, env.max_episode_length, (N,), generator=gen)
    ep[::5] = env.max_episode_length - 1
    groups = list(fx["env"]["observations"])
    W = sum(int(np.prod(env.observation_manager.group_obs_dim[g])) for g in groups)  # wide enough for every group's columns
    env._noise_u = torch.rand(N, W, generator=gen).cuda()
    obs0, _ = env.reset()
    orc.reset_action_terms()  # ActionManager.reset of env.reset(): the EMA term's previous applied action <- the joint positions
    if env.plan.num_rays:
        orc.ray_hits_w = env._ray_hits.cpu()
    ref0 = orc.compute_observation_groups(env._noise_u.cpu())
    for gname in groups:
        assert_close(obs0[gname], ref0[gname], FLOAT_TOL, f"reset obs[{gname}]")
    env.episode_length_buf = ep
    orc.episode_length_buf[:] = ep
    nreset = 0
    for _ in range(steps):
        a = torch.randn(N, env.plan.action_dim, generator=gen).clamp(-3, 3)
        u = torch.rand(N, W, generator=gen)
        env._noise_u.copy_(u)
        obs_dict, rew, term, tout, extras = env.step(a.cuda())
        orc.process_action(a)
        pa, po = env._processed_action.cpu(), orc.processed_actions
        badc = ((pa - po).abs() > FLOAT_TOL * po.abs().clamp(min=1.0)).any(0).nonzero().flatten().tolist()
        if badc:
            raise AssertionError(f"processed actions step {_}: columns {badc} differ (max {float((pa - po).abs().max()):.3e}); terms "
                                 + json.dumps({n: {k: v for k, v in a.items() if k in ('class_type', 'scale', 'offset', 'clip', 'alpha', 'joint_names', 'use_zero_offset', 'use_default_offset', 'rescale_to_limits')}
                                               for n, a in fx["env"]["actions"].items()}))
        cpu_feed.advance()
        if env.plan.num_rays:
            orc.ray_hits_w = env._ray_hits.cpu()
        out = orc.post_physics_step(u)
        assert torch.equal(term.cpu(), out["terminated"]) and torch.equal(tout.cpu(), out["time_outs"]), "masks"
        assert torch.equal(env.reset_env_ids.cpu(), out["reset_env_ids"]), "reset ids"
        nreset += len(out["reset_env_ids"])
        # the reward is a sum of weighted terms that the random weights can make large and cancelling (seen: +8608 - 2485 ... -> 122.46): the
        # fp32 rounding of the TERMS (1e-7 relative each, agreed to the last digits) bounds the error of the sum, not the size of the sum
        scale = (out["step_reward"].abs().sum(1) * float(env.step_dt)).clamp(min=1.0)
        rerr = (rew.cpu() - out["reward"]).abs() / scale
        assert float(rerr.max()) <= FLOAT_TOL, f"reward: {float(rerr.max()):.2e} of the summed term magnitudes (env {int(rerr.argmax())})"
        assert_close(env.reward_manager._step_reward, out["step_reward"], FLOAT_TOL, "per-term step reward")
        for gname in groups:
            got, ref = obs_dict[gname].cpu(), out["obs_groups"][gname]
            bad = ((got - ref).abs() > FLOAT_TOL * ref.abs().clamp(min=1.0)).any(0).nonzero().flatten().tolist()
            if bad:
                grp = fx["env"]["observations"][gname]
                desc = [(k, (v.get("noise") or {}).get("operation"), v.get("history_length"), v.get("scale"), v.get("clip")) for k, v in grp.items() if isinstance(v, dict)]
                raise AssertionError(f"obs[{gname}] step {_}: columns {bad[:6]}..{bad[-1]} ({len(bad)} of {got.shape[1]}) differ, max {float((got - ref).abs().max()):.3e}; "
                                     f"group history {grp.get('history_length')} corruption {grp.get('enable_corruption')} terms {desc}")
        assert torch.equal(env.episode_length_buf.cpu(), orc.episode_length_buf), "episode length"
        for key, val in out["log"].items():
            assert abs(float(extras["log"][key]) - val) <= 1e-5 * max(1.0, abs(val)), key
    env.close()
    return f"N={N} steps={steps} resets={nreset}: {what}"


if __name__ == "__main__":
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    bad = 0
    for c in range(first, first + cases):
        try:
            print(f"case {c}: ok   {one_case(c)}", flush=True)
        except (AssertionError, NotImplementedError, ValueError, KeyError, RuntimeError) as e:
            bad += 1
            print(f"case {c}: FAIL {type(e).__name__}: {str(e)[:1500]}", flush=True)
    print(f"{cases - bad} / {cases} cases agree")
    sys.exit(1 if bad else 0)
