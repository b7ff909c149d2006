"""TEST INFRASTRUCTURE (build container only) -- import the Python reference with the simulator stubbed.

The reference (``/root/reference``) is pure Python but imports Omniverse / Isaac Sim / Warp / gymnasium at module
import time.  None of those exist here, so this module installs a ``sys.meta_path`` finder that serves ``MagicMock``
modules for the missing roots, a tiny real ``gymnasium`` (``Env``/``Wrapper``/``register``) and a ``toml`` shim over
``tomli``; then it puts the reference's source roots on ``sys.path``.  Nothing is written to ``/root/reference``
(``sys.dont_write_bytecode``).  This is used ONLY by ``oracle/gen_golden.py`` to produce the committed fixtures under
``tests/golden`` and by ``oracle/time_reference.py`` (reference vs restatement timing); it never travels to the GPU box as a dependency of anything
(the reference itself does not exist there).
"""

from __future__ import annotations

import importlib.abc
import importlib.machinery
import os
import sys
import types
from unittest import mock

REFERENCE_ROOT = os.environ.get("IMX_REFERENCE_ROOT", "/root/reference")

_MOCK_ROOTS = (
    "omni", "isaacsim", "pxr", "carb", "warp", "prettytable", "trimesh", "h5py", "hid", "pink", "pinocchio",
    "dex_retargeting", "usdrt", "nvidia", "flatdict", "onnx", "torchvision", "tensordict", "hydra", "pyglet",
    "websockets", "Semantics", "omegaconf", "rsl_rl",
)
# roots that are mocked only when genuinely absent from this interpreter
_MOCK_IF_MISSING = ("matplotlib", "mpl_toolkits", "PIL", "cv2", "pyperclip", "scipy")


class _MockModule(mock.MagicMock):
    # make ``from x import *`` and ``__path__`` lookups behave
    __all__: list = []
    __path__: list = []


class _MockLoader(importlib.abc.Loader):
    def create_module(self, spec):
        m = _MockModule(name=spec.name)
        m.__name__ = spec.name
        m.__spec__ = spec
        m.__loader__ = self
        m.__path__ = []
        m.__file__ = "<mock %s>" % spec.name
        return m

    def exec_module(self, module):
        pass


_probe_cache: dict = {}


def _really_importable(root: str) -> bool:
    if root not in _probe_cache:
        _probe_cache[root] = any(
            f is not None
            for f in (
                importlib.machinery.PathFinder.find_spec(root, None),
            )
        )
    return _probe_cache[root]


class _MockFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path, target=None):
        root = fullname.split(".", 1)[0]
        if root in _MOCK_ROOTS or (root in _MOCK_IF_MISSING and not _really_importable(root)):
            return importlib.machinery.ModuleSpec(fullname, _MockLoader(), is_package=True)
        return None


def _install_gymnasium():
    if "gymnasium" in sys.modules:
        return
    gym = types.ModuleType("gymnasium")

    class Env:  # minimal stand-in: ManagerBasedRLEnv multiply-inherits gym.Env
        metadata: dict = {}
        render_mode = None

        @property
        def unwrapped(self):
            return self

    class Wrapper(Env):
        def __init__(self, env):
            self.env = env

        @property
        def unwrapped(self):
            return self.env.unwrapped

    _registry: dict = {}

    def register(id, entry_point=None, disable_env_checker=True, kwargs=None, **kw):  # noqa: A002
        _registry[id] = dict(entry_point=entry_point, kwargs=kwargs or {})

    gym.Env = Env
    gym.Space = type("Space", (), {})
    gym.Wrapper = Wrapper
    gym.register = register
    gym.registry = _registry
    gym.spaces = _MockModule(name="gymnasium.spaces")
    gym.vector = _MockModule(name="gymnasium.vector")
    gym.envs = _MockModule(name="gymnasium.envs")
    gym.__path__ = []
    sys.modules["gymnasium"] = gym
    sys.modules["gymnasium.spaces"] = gym.spaces
    sys.modules["gymnasium.vector"] = gym.vector
    sys.modules["gymnasium.vector.utils"] = gym.vector.utils
    sys.modules["gymnasium.envs"] = gym.envs
    sys.modules["gymnasium.envs.registration"] = gym.envs.registration


def _install_toml():
    if "toml" in sys.modules:
        return
    import tomli

    toml = types.ModuleType("toml")
    toml.load = lambda f: tomli.loads(open(f).read() if isinstance(f, str) else f.read())
    toml.loads = tomli.loads
    sys.modules["toml"] = toml


_installed = False


def install():
    """Make ``import isaaclab``, ``isaaclab_tasks``, ``isaaclab_assets`` resolve to the reference sources."""
    global _installed
    if _installed:
        return
    if not os.path.isdir(REFERENCE_ROOT):
        raise RuntimeError(f"reference not present at {REFERENCE_ROOT} (this only works in the build container)")
    sys.dont_write_bytecode = True
    sys.meta_path.insert(0, _MockFinder())
    _install_gymnasium()
    _install_toml()
    for pkg in ("isaaclab", "isaaclab_tasks", "isaaclab_rl", "isaaclab_assets"):
        p = os.path.join(REFERENCE_ROOT, "source", pkg)
        if p not in sys.path:
            sys.path.insert(0, p)
    _installed = True


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "source", "isaaclab"))
