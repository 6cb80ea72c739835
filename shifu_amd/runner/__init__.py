from .utils import datetime_logdir, latest_logdir
from .policy_runner import run_policy, load_policy
