"""isaaclab_amd -- MI355X-native post-physics env-step + RSL-RL rollout hot path for IsaacLab task configs.

Product code.  Never imports ``oracle`` (the CPU restatement is test infrastructure).
"""

import os as _os

# Effective when this package is imported before the first HIP call of the process (import it before touching torch.cuda): the
# update runs on three streams and a distributed run adds RCCL's; with ROCm's default of 4 hardware queues two of them end up sharing
# one once the process group is created first (+14 % update time, tools/dist_overhead.py).  A value set by the user wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

__version__ = "0.1.0"
