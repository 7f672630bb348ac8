"""``src.opensmile_extractor`` of the reference (``src/opensmile_extractor.py:9-103``) on the HIP path."""
from robust_speech_analysis_framework_amd.smile import extract_opensmile_features  # noqa: F401
