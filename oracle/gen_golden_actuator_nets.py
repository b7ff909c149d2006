"""TEST INFRASTRUCTURE (build container only): tests/golden/actuator_nets.npz from the REAL reference classes ``ActuatorNetLSTM`` and
``ActuatorNetMLP`` (isaaclab/actuators/actuator_net.py:29-195), constructed through their own ``__init__`` from the reference's cfg
classes.  The network files the reference downloads from Nucleus are absent (no network), so the two TorchScript networks are synthetic:
random weights in the architecture the classes expect -- ``network.lstm`` = nn.LSTM(2, 8, 2, batch_first=True) + a softsign head for the
LSTM model (the ANYdrive-3 layout), a 6-32-32-1 softsign MLP for the history model -- scripted and saved by THIS script into a temp
directory, read back by the classes' own ``read_file`` / ``torch.jit.load``.  The fixture keeps the weights as arrays (no TorchScript
file is committed), the inputs of every step, the resets, and what the classes computed.  See oracle/gen_golden.py for the import stub."""

from __future__ import annotations

import json
import os
import sys
import tempfile

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_import  # noqa: E402

ref_import.install()

from isaaclab.actuators.actuator_cfg import ActuatorNetLSTMCfg, ActuatorNetMLPCfg  # noqa: E402
from isaaclab.actuators.actuator_net import ActuatorNetLSTM, ActuatorNetMLP  # noqa: E402
from isaaclab.utils.types import ArticulationActions  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


class SeaLstm(nn.Module):
    """(N*J, 1, 2) + (h, c) -> (N*J, 1) torque + (h, c): what ActuatorNetLSTM.compute calls (actuator_net.py:83-87)."""

    def __init__(self, hidden=8, layers=2, width=16):
        super().__init__()
        self.lstm = nn.LSTM(2, hidden, layers, batch_first=True)
        self.fc1 = nn.Linear(hidden, width)
        self.act = nn.Softsign()
        self.fc2 = nn.Linear(width, 1)

    def forward(self, x: torch.Tensor, hc: tuple[torch.Tensor, torch.Tensor]):
        y, (h, c) = self.lstm(x, hc)
        return self.fc2(self.act(self.fc1(y[:, -1]))), (h, c)


def main():
    N, J = 24, 12
    g = torch.Generator().manual_seed(41)
    torch.manual_seed(41)
    rec = {}
    tmp = tempfile.mkdtemp()
    joint_names = [f"j{k}" for k in range(J)]
    sat, elim, vlim = 120.0, 80.0, 7.5  # ANYDRIVE_3_LSTM_ACTUATOR_CFG (isaaclab_assets/robots/anymal.py:45-51)
    steps = 8

    def drive(a, tag):
        for t in range(steps):
            if t in (3, 6):
                ids = torch.arange(t % 4, N, 4)
                a.reset(ids)
                rec[f"{tag}/step{t}/reset_ids"] = ids.numpy().copy()
            q_des = torch.randn(N, J, generator=g) * 0.6
            q = q_des + torch.randn(N, J, generator=g) * 0.3
            qd = torch.randn(N, J, generator=g) * 5.0  # beyond the velocity limit too (DC-motor window)
            out = a.compute(ArticulationActions(joint_positions=q_des.clone(), joint_velocities=None, joint_efforts=None), q, qd)
            assert out.joint_positions is None and torch.equal(out.joint_efforts, a.applied_effort)
            for k, v in (("q_des", q_des), ("q", q), ("qd", qd), ("computed", a.computed_effort), ("applied", a.applied_effort)):
                rec[f"{tag}/step{t}/{k}"] = v.detach().numpy().copy()

    # ---- ActuatorNetLSTM
    net = SeaLstm()
    with torch.no_grad():
        for p_ in net.parameters():
            p_.mul_(2.5)  # livelier than the default init: gates leave their linear range, torques reach the clip
        net.fc2.weight.mul_(40.0)
    path = os.path.join(tmp, "sea_lstm.pt")
    torch.jit.script(net).save(path)
    cfg = ActuatorNetLSTMCfg(joint_names_expr=[".*"], network_file=path, saturation_effort=sat, effort_limit=elim, velocity_limit=vlim)
    a = ActuatorNetLSTM(cfg, joint_names=joint_names, joint_ids=slice(None), num_envs=N, device="cpu")
    for k, v in net.state_dict().items():
        rec[f"lstm/net/{k}"] = v.numpy().copy()
    drive(a, "lstm")
    rec["lstm/final_hidden"] = a.sea_hidden_state.numpy().copy()
    rec["lstm/final_cell"] = a.sea_cell_state.numpy().copy()

    # ---- ActuatorNetMLP (history model): entries 0, 2, 4 of the queues, velocities first
    mlp = nn.Sequential(nn.Linear(6, 32), nn.Softsign(), nn.Linear(32, 32), nn.Softsign(), nn.Linear(32, 1))
    with torch.no_grad():
        for p_ in mlp.parameters():
            p_.mul_(2.0)
    path = os.path.join(tmp, "sea_mlp.pt")
    torch.jit.script(mlp).save(path)
    mcfg = ActuatorNetMLPCfg(joint_names_expr=[".*"], network_file=path, saturation_effort=sat, effort_limit=elim, velocity_limit=vlim,
                             pos_scale=-1.0, vel_scale=1.0, torque_scale=60.0, input_idx=[0, 2, 4], input_order="vel_pos")
    m = ActuatorNetMLP(mcfg, joint_names=joint_names, joint_ids=slice(None), num_envs=N, device="cpu")
    for k, v in mlp.state_dict().items():
        rec[f"mlp/net/{k}"] = v.numpy().copy()
    drive(m, "mlp")
    rec["mlp/final_pos_hist"] = m._joint_pos_error_history.numpy().copy()
    rec["mlp/final_vel_hist"] = m._joint_vel_history.numpy().copy()
    rec["meta"] = np.array(json.dumps(dict(N=N, J=J, steps=steps, saturation_effort=sat, effort_limit=elim, velocity_limit=vlim,
                                           lstm=dict(hidden=8, layers=2, head_activation="softsign"),
                                           mlp=dict(input_idx=[0, 2, 4], input_order="vel_pos", pos_scale=-1.0, vel_scale=1.0,
                                                    torque_scale=60.0, activation="softsign"))))
    np.savez_compressed(os.path.join(GOLDEN, "actuator_nets.npz"), **rec)
    for tag in ("lstm", "mlp"):
        c = np.concatenate([rec[f"{tag}/step{t}/computed"].ravel() for t in range(steps)])
        ap = np.concatenate([rec[f"{tag}/step{t}/applied"].ravel() for t in range(steps)])
        print(f"[golden] actuator_nets/{tag}: |torque| max {np.abs(c).max():.1f}, clipped fraction {(c != ap).mean():.2f}")


if __name__ == "__main__":
    main()
