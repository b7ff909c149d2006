"""Library-GEMM algorithm selection for the actor-critic MLPs (they stay in torch / hipBLASLt / rocBLAS).

PyTorch's TunableOp picks, per GEMM shape, the fastest hipBLASLt/rocBLAS solution.  The selections for the shapes of
the BASELINE.json configs were recorded on an MI355X (``isaaclab_amd/tuning/tunableop_gfx950_0.csv``; the file carries
validators for the torch / hipBLASLt / rocBLAS versions and is ignored when they do not match) and are only *read*
here: no tuning happens at run time unless ``IMX_GEMM_TUNE=1`` (then results go to ``IMX_GEMM_TUNE_FILE``).
fp32 in, fp32 accumulate -- numerics unchanged up to the summation order inside the library.
"""

from __future__ import annotations

import os

import torch

_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tuning")
RECORDED = os.path.join(_DIR, "tunableop_gfx950_0.csv")
_done = False


def enable_recorded_gemm_tuning() -> bool:
    global _done
    if _done or not torch.cuda.is_available():
        return _done
    if os.environ.get("IMX_GEMM_TUNING", "1") == "0":
        return False
    try:
        import torch.cuda.tunable as tunable

        tune = os.environ.get("IMX_GEMM_TUNE", "0") == "1"
        tunable.enable(True)
        tunable.tuning_enable(tune)
        if tune:
            tunable.set_filename(os.environ.get("IMX_GEMM_TUNE_FILE", os.path.join("gpurun_out", "tunableop_gfx950.csv")), True)
        elif os.path.exists(RECORDED):
            if hasattr(tunable, "write_file_on_exit"):  # older torch rewrote the file at exit; 2.10 appends only while tuning
                tunable.write_file_on_exit(False)  # read-only: eight ranks exiting together must never rewrite the in-tree table
            tunable.set_filename(RECORDED, False)  # every rank reads the same recorded table
            tunable.read_file(RECORDED)
        _done = True
    except Exception as exc:  # an optimisation of library calls, never a correctness dependency
        print(f"[isaaclab_amd] GEMM tuning not enabled: {exc}")
        _done = False
    return _done
