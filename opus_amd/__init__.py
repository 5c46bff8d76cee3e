"""MI355X-native multi-scale multi-view feature sampling for OPUS (the msmv_sampling op).

``opus_amd.csrc.wrapper`` mirrors the reference's ``models/csrc/wrapper.py``;
``opus_amd.opus_sampling`` mirrors the caller ``models/opus_sampling.py``.
"""
__version__ = "0.1.0"
