"""Drop-in for the reference's ``src/utils.py``: same names and signatures, arithmetic on the MI355X."""
from robust_speech_analysis_framework_amd.aggregate import (  # noqa: F401
    aggregate_clip_features,
    aggregate_interview_sequences,
)
