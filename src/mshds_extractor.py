"""``src.mshds_extractor`` of the reference (``src/mshds_extractor.py:379-459``) on the HIP path."""
from robust_speech_analysis_framework_amd.mshds import extract_mshds_features  # noqa: F401
