from ripor_amd.modeling.t5_generative_retriever import (T5forDocIDConfig, T5ForDocIDGeneration,  # noqa: F401
                                                          T5SeqAQEncoder, T5SeqAQEncoderForLngKnpMarginMSE)
