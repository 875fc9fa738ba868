from ripor_amd.tasks.generation import *  # noqa: F401,F403
from ripor_amd.tasks.generation import (BeamSearchEncoderDecoderOutput, PrefixConstrainLogitProcessorFastSparse,  # noqa: F401
                                        generate_for_constrained_prefix_beam_search)
