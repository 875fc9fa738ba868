from ripor_amd.utils.utils import *  # noqa: F401,F403
from ripor_amd.utils.utils import convert_ptsmtids_to_strsmtid, get_dataset_name, is_first_worker, makedir  # noqa: F401
