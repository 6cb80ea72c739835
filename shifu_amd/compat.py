"""Run UNMODIFIED reference user code on this backend.

    import shifu_amd.compat; shifu_amd.compat.install()
    from isaacgym import gymapi            # -> shifu_amd.isaacgym.gymapi
    from shifu.gym import ShifuVecEnv      # -> shifu_amd.gym.ShifuVecEnv
    from shifu.units import LeggedRobot    # -> shifu_amd.units.LeggedRobot

install() registers aliases in sys.modules for the package names the reference's
examples import (examples/a1_conditional/a1_conditional.py:3-17).  It never shadows a
real installation: a name that is already importable is left alone unless force=True."""
import importlib
import importlib.util
import sys
import types

_ALIASES = {
    "isaacgym": "shifu_amd.isaacgym",
    "isaacgym.gymapi": "shifu_amd.isaacgym.gymapi",
    "isaacgym.gymtorch": "shifu_amd.isaacgym.gymtorch",
    "isaacgym.gymutil": "shifu_amd.isaacgym.gymutil",
    "isaacgym.torch_utils": "shifu_amd.isaacgym.torch_utils",
    "isaacgym.terrain_utils": "shifu_amd.isaacgym.terrain_utils",
    "shifu": "shifu_amd",
    "shifu.configs": "shifu_amd.configs",
    "shifu.gym": "shifu_amd.gym",
    "shifu.units": "shifu_amd.units",
    "shifu.runner": "shifu_amd.runner",
    "shifu.utils": "shifu_amd.utils",
    "shifu.utils.train": "shifu_amd.utils.train",
    "shifu.utils.terrain": "shifu_amd.utils.terrain",
    "shifu.utils.torch_utils": "shifu_amd.utils.torch_utils",
}


def install(force: bool = False):
    for alias, target in _ALIASES.items():
        top = alias.split(".")[0]
        if not force and top not in ("isaacgym", "shifu"):
            continue
        if not force and alias not in sys.modules and top not in sys.modules:
            try:
                if importlib.util.find_spec(top) is not None and not top.startswith("shifu_amd"):
                    # a real package of that name exists: do not shadow it
                    real = importlib.util.find_spec(top)
                    if real.origin and "shifu_amd" not in real.origin:
                        continue
            except (ImportError, ValueError):
                pass
        sys.modules[alias] = importlib.import_module(target)
    # rsl_rl.env.VecEnv is only used as a base class by the reference (env.py:6,18)
    if "rsl_rl" not in sys.modules and importlib.util.find_spec("rsl_rl") is None:
        rsl = types.ModuleType("rsl_rl")
        env = types.ModuleType("rsl_rl.env")
        env.VecEnv = type("VecEnv", (), {})
        rsl.env = env
        sys.modules["rsl_rl"], sys.modules["rsl_rl.env"] = rsl, env
