"""isaaclab_amd -- MI355X-native post-physics env-step + RSL-RL rollout hot path for IsaacLab task configs.

Product code.  Never imports ``oracle`` (the CPU restatement is test infrastructure).
"""

__version__ = "0.1.0"
