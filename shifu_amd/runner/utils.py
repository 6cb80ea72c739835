"""Run-directory / seeding helpers (reference shifu/runner/utils.py:8-73)."""
import os
import random
from datetime import datetime

import numpy as np
import torch


def datetime_logdir(log_root, run_name):
    return os.path.join(log_root, datetime.now().strftime('%Y%m%d-%H:%M:%S') + '_' + run_name)


def latest_logdir(log_root, run_name=''):
    runs = sorted(r for r in os.listdir(log_root) if run_name in r)
    print(f'found the latest logdir: {runs[-1]}')
    return os.path.join(log_root, runs[-1])


def class_to_dict(obj) -> dict:
    """Config object -> nested dict of its public attributes (what rsl_rl consumes)."""
    if not hasattr(obj, "__dict__"):
        return obj
    out = {}
    for key in dir(obj):
        if key.startswith("_"):
            continue
        val = getattr(obj, key)
        out[key] = [class_to_dict(v) for v in val] if isinstance(val, list) else class_to_dict(val)
    return out


def get_load_path(root, load_run=-1, checkpoint=-1):
    try:
        last_run = latest_logdir(root)
    except Exception:
        raise ValueError("No runs in this directory: " + root)
    run = last_run if load_run == -1 else os.path.join(root, load_run)
    if checkpoint == -1:
        models = sorted((f for f in os.listdir(run) if 'model' in f), key=lambda m: '{0:0>15}'.format(m))
        model = models[-1]
    else:
        model = "model_{}.pt".format(checkpoint)
    return os.path.join(run, model)


def set_seed(seed):
    if seed == -1:
        seed = np.random.randint(0, 10000)
    print("Setting seed: {}".format(seed))
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    os.environ['PYTHONHASHSEED'] = str(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
        torch.cuda.manual_seed_all(seed)
