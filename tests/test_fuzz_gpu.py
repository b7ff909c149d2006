"""A few cases of every random sweep under tools/fuzz_*.py on each test run (the full campaigns -- 1500 env-step cases, 300 per kernel, 400
rollouts, 300 cfgs -- are recorded in profiles/r03_fuzz_*.txt; two of them found real defects in round 3).  The seeds differ per sweep and
are fixed, so a failure reproduces with `python tools/fuzz_<name>.py <cases> <seed>`."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.mark.parametrize("kind", ["gae", "fwd_elu", "infer", "dw", "lstm", "infer_act", "ppo_loss", "update", "update_graph"])
def test_kernel_sweep(libimx, kind):
    import fuzz_kernels as fk

    rng = np.random.default_rng(2024)
    fn = getattr(fk, "case_" + kind)
    for c in range(8):
        ok, msg = fn(rng)
        assert ok, f"{kind} case {c} (python tools/fuzz_kernels.py 8 2024): {msg}"


@pytest.mark.parametrize("seed", [5000, 5001, 5002, 5003, 5004, 5005])
def test_env_step_parity_sweep(seed):
    import fuzz_parity

    fuzz_parity.one_case(seed)


@pytest.mark.parametrize("seed", [7000, 7001, 7002, 7003, 7004, 7005])
def test_random_cfg_sweep(seed):
    import fuzz_cfg

    fuzz_cfg.one_case(seed)


@pytest.mark.parametrize("seed", [9000, 9001, 9002, 9003])
def test_rollout_sweep(seed):
    import fuzz_rollout

    fuzz_rollout.one_case(seed)


@pytest.mark.parametrize("kind", ["contact", "command", "pd_actuator", "articulation"])
def test_producer_sweep(kind):
    import fuzz_producers as fp

    rng = np.random.default_rng(31)
    for _ in range(8):
        getattr(fp, "case_" + kind)(rng)


@pytest.mark.parametrize("seed", [300, 301, 302])
def test_orchestration_sweep(seed):
    import fuzz_orchestration

    fuzz_orchestration.one_case(seed)


@pytest.mark.parametrize("seed", [400, 401, 402])
def test_raycast_sweep(seed):
    import fuzz_raycast

    fuzz_raycast.one_case(seed)
