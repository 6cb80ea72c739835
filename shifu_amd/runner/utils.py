"""Run directories, checkpoint lookup, config flattening and seeding for `run_policy`.

Behavioural contract taken from the reference (shifu/runner/utils.py:8-73) so that run folders written by either code
base load in the other: a run lives in `<log_root>/<YYYYmmdd-HH:MM:SS>_<run_name>`, its checkpoints are `model_<it>.pt`,
"latest" means last in lexical order of the folder names / highest iteration of the checkpoints, seed -1 draws one.
The implementation is this repo's own.
"""
import datetime as _dt
import os
import random
import re

import numpy as np
import torch

_STAMP = "%Y%m%d-%H:%M:%S"
_MODEL_RE = re.compile(r"model_(\d+)\.pt$")


def datetime_logdir(log_root, run_name):
    """Path of a new run folder under `log_root` (not created here)."""
    return os.path.join(log_root, f"{_dt.datetime.now().strftime(_STAMP)}_{run_name}")


def latest_logdir(log_root, run_name=''):
    """The newest run folder under `log_root` whose name contains `run_name` (the time stamp sorts lexically)."""
    candidates = [d for d in os.listdir(log_root) if run_name in d]
    if not candidates:
        raise FileNotFoundError(f"no run containing {run_name!r} under {log_root}")
    newest = max(candidates)
    print(f'found the latest logdir: {newest}')
    return os.path.join(log_root, newest)


def class_to_dict(obj) -> dict:
    """Config object -> nested dict of its public attributes (what the runner's constructor consumes).  Anything without
    attributes of its own (numbers, strings, tuples, None) is returned as is; lists are converted element-wise."""
    if isinstance(obj, list):
        return [class_to_dict(v) for v in obj]
    if not hasattr(obj, "__dict__"):
        return obj
    public = (k for k in dir(obj) if not k.startswith("_"))
    return {k: class_to_dict(getattr(obj, k)) for k in public}


def _iteration(name: str) -> int:
    m = _MODEL_RE.search(name)
    return int(m.group(1)) if m else -1


def get_load_path(root, load_run=-1, checkpoint=-1):
    """`<root>/<run>/model_<checkpoint>.pt`; run -1 = the newest run folder, checkpoint -1 = the highest iteration saved."""
    if load_run == -1:
        try:
            run = latest_logdir(root)
        except (FileNotFoundError, NotADirectoryError) as exc:
            raise ValueError("No runs in this directory: " + root) from exc
    else:
        run = os.path.join(root, load_run)
    if checkpoint != -1:
        return os.path.join(run, f"model_{checkpoint}.pt")
    saved = [f for f in os.listdir(run) if _iteration(f) >= 0]
    if not saved:
        raise ValueError("No checkpoints in this run: " + run)
    return os.path.join(run, max(saved, key=_iteration))


def set_seed(seed):
    """Seed Python, NumPy and torch (all devices); -1 draws a seed in [0, 10000) first.  Returns nothing, prints the seed."""
    if seed == -1:
        seed = int(np.random.randint(0, 10000))
    print(f"Setting seed: {seed}")
    os.environ['PYTHONHASHSEED'] = str(seed)
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)          # seeds the CPU generator and, lazily, every CUDA device's
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
