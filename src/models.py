"""``src.models`` of the reference (``src/models.py:7-193``) served by the HIP path."""
from robust_speech_analysis_framework_amd.cnnlstm import (  # noqa: F401
    AttentionPooling, CNNLSTM, ResidualBlock, get_activation_fn)
