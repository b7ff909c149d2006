"""TEST INFRASTRUCTURE (build container only): tests/golden/actuators.npz from the REAL reference actuator models
``IdealPDActuator`` and ``DCMotor`` (isaaclab/actuators/actuator_pd.py:148-286; ``ImplicitActuator.compute`` :115-145 evaluates
the same PD law for reporting), objects created with ``__new__`` and their gain / limit buffers set by hand (the constructors
need the PhysX-parsed joint properties).  See oracle/gen_golden.py for the import stub."""

from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_import  # noqa: E402

ref_import.install()

from isaaclab.actuators.actuator_pd import DCMotor, DelayedPDActuator, IdealPDActuator, RemotizedPDActuator  # noqa: E402
from isaaclab.utils.buffers import DelayBuffer  # noqa: E402
from isaaclab.utils.interpolation import LinearInterpolation  # noqa: E402
from isaaclab.utils.types import ArticulationActions  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def main():
    N, J = 40, 12
    g = torch.Generator().manual_seed(31)
    rec = {}
    stiff = torch.rand(N, J, generator=g) * 60 + 20
    damp = torch.rand(N, J, generator=g) * 3 + 0.5
    elim = torch.rand(N, J, generator=g) * 40 + 20
    vlim = torch.rand(N, J, generator=g) * 5 + 5
    for name, t in (("stiffness", stiff), ("damping", damp), ("effort_limit", elim), ("velocity_limit", vlim)):
        rec[name] = t.numpy().copy()
    sat = 120.0
    for tag, cls in (("ideal", IdealPDActuator), ("dc", DCMotor)):
        a = cls.__new__(cls)
        a.stiffness, a.damping, a.effort_limit, a.velocity_limit = stiff, damp, elim, vlim
        a.computed_effort = torch.zeros(N, J)
        a.applied_effort = torch.zeros(N, J)
        if cls is DCMotor:
            a._saturation_effort = sat
            a._joint_vel = torch.zeros(N, J)
            a._zeros_effort = torch.zeros(N, J)
        q_des = torch.randn(N, J, generator=g)
        qd_des = torch.randn(N, J, generator=g) * 0.5
        ff = torch.randn(N, J, generator=g) * 5
        q = q_des + torch.randn(N, J, generator=g) * 0.6   # errors large enough to hit the effort limits
        qd = torch.randn(N, J, generator=g) * 6             # velocities beyond the velocity limit too (DC motor saturation)
        act = ArticulationActions(joint_positions=q_des.clone(), joint_velocities=qd_des.clone(), joint_efforts=ff.clone())
        out = a.compute(act, q, qd)
        assert out.joint_positions is None and out.joint_velocities is None
        for k, t in (("q_des", q_des), ("qd_des", qd_des), ("ff", ff), ("q", q), ("qd", qd), ("computed", a.computed_effort),
                     ("applied", a.applied_effort)):
            rec[f"{tag}/{k}"] = t.numpy().copy()
        assert torch.equal(out.joint_efforts, a.applied_effort)
    # ---- DelayedPDActuator (:289-346) and RemotizedPDActuator (:349-412): 9 physics steps, partial resets (new random lags) at
    # steps 0 (all), 4 and 6; the lags drawn by the reference's torch.randint are read back from the DelayBuffer and stored
    class _Cfg:
        min_delay, max_delay = 0, 4

    lookup = torch.tensor([[-1.2, 1.0, 30.0], [-0.4, 1.1, 55.0], [0.1, 1.2, 48.0], [0.9, 1.3, 22.0], [1.6, 1.4, 35.0]])
    rec["remotized/lookup"] = lookup.numpy().copy()
    for tag, cls in (("delayed", DelayedPDActuator), ("remotized", RemotizedPDActuator)):
        a = cls.__new__(cls)
        a.cfg, a._num_envs, a._device = _Cfg, N, "cpu"
        inf = torch.full((N, J), float("inf"))
        a.stiffness, a.damping, a.velocity_limit = stiff, damp, vlim
        a.effort_limit = inf if cls is RemotizedPDActuator else elim
        a.computed_effort, a.applied_effort = torch.zeros(N, J), torch.zeros(N, J)
        a.positions_delay_buffer = DelayBuffer(_Cfg.max_delay, N, device="cpu")
        a.velocities_delay_buffer = DelayBuffer(_Cfg.max_delay, N, device="cpu")
        a.efforts_delay_buffer = DelayBuffer(_Cfg.max_delay, N, device="cpu")
        a._ALL_INDICES = torch.arange(N)
        if cls is RemotizedPDActuator:
            a._joint_parameter_lookup = lookup
            a._torque_limit = LinearInterpolation(a.angle_samples, a.max_torque_samples, device="cpu")
        torch.manual_seed(77)
        steps = 9
        for t in range(steps):
            ids = None
            if t == 0:
                ids = torch.arange(N)
            elif t == 4:
                ids = torch.arange(0, N, 3)
            elif t == 6:
                ids = torch.tensor([1, 2, 5, 11, 30])
            if ids is not None:
                a.reset(ids)
                rec[f"{tag}/step{t}/reset_ids"] = ids.numpy().copy()
                rec[f"{tag}/step{t}/time_lags"] = a.positions_delay_buffer.time_lags[ids].numpy().copy()
            q_des = torch.randn(N, J, generator=g)
            qd_des = torch.randn(N, J, generator=g) * 0.5
            ff = torch.randn(N, J, generator=g) * 5
            q = q_des * 0.5 + torch.randn(N, J, generator=g) * 0.8   # spans the lookup table and beyond both ends
            qd = torch.randn(N, J, generator=g) * 3
            act = ArticulationActions(joint_positions=q_des.clone(), joint_velocities=qd_des.clone(), joint_efforts=ff.clone())
            out = a.compute(act, q, qd)
            for k, v in (("q_des", q_des), ("qd_des", qd_des), ("ff", ff), ("q", q), ("qd", qd), ("computed", a.computed_effort),
                         ("applied", a.applied_effort)):
                rec[f"{tag}/step{t}/{k}"] = v.numpy().copy()
            assert torch.equal(out.joint_efforts, a.applied_effort)
    rec["meta"] = np.array(json.dumps(dict(N=N, J=J, saturation_effort=sat, max_delay=_Cfg.max_delay, delayed_steps=9)))
    np.savez_compressed(os.path.join(GOLDEN, "actuators.npz"), **rec)
    clipped = float((torch.from_numpy(rec["dc/applied"]) != torch.from_numpy(rec["dc/computed"])).float().mean())
    print(f"[golden] actuators: {len(rec)} arrays; DC motor clipped fraction {clipped:.2f}")


if __name__ == "__main__":
    main()
