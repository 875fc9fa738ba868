from ripor_amd.dataset.lng_knp import *  # noqa: F401,F403
