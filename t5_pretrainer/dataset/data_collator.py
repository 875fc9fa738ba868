from ripor_amd.dataset.lng_knp import LngKnpMarginMSEforT5SeqAQCollator  # noqa: F401
