from ripor_amd.utils.metrics import *  # noqa: F401,F403
from ripor_amd.utils.metrics import evaluate, load_and_evaluate, mrr_k, truncate_run  # noqa: F401
