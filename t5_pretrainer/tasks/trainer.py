from ripor_amd.tasks.trainer import *  # noqa: F401,F403
