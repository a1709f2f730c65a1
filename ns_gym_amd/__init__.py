"""ns_gym_amd — MI355X-native vectorised non-stationary environment stepper.

Drop-in for the hot path of scope-lab-vu/ns_gym: NSClassicControlWrapper /
NSFrozenLakeWrapper step()/reset() + schedulers + update_functions, executed by
hand-written HIP kernels (gfx950) behind a C-ABI (include/nsgym_hip.h).
"""
from . import _abi, base, envs, schedulers, spec, update_functions  # noqa: F401
from .base import Reward, Scheduler, UpdateDistributionFn, UpdateFn  # noqa: F401
from .envs import TUNABLE_PARAMS, make, register, registry  # noqa: F401

__version__ = "0.1.0"

_LAZY = {"VecNSEnv": "vec_env", "NSClassicControlWrapper": "wrappers", "NSFrozenLakeWrapper": "wrappers", "NSCliffWalkingWrapper": "wrappers", "NSBridgeWrapper": "wrappers",
         "ConstraintViolationWarning": "wrappers", "functional": None, "vec_env": None, "wrappers": None,
         "distributed": None, "utils": None, "evaluate": None, "policies": None, "planning": None}


def __getattr__(name):
    if name in _LAZY:
        import importlib

        mod = importlib.import_module(f".{_LAZY[name] or name}", __name__)
        return mod if _LAZY[name] is None else getattr(mod, name)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
