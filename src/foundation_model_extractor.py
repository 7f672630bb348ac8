"""``src.foundation_model_extractor`` of the reference (``:37-166``) on the HIP path."""
from robust_speech_analysis_framework_amd.w2v2 import (  # noqa: F401
    extract_wav2vec2_embeddings, extract_wav2vec2_sequences)


def _safe_cuda_cleanup(*tensors):
    """Kept for signature compatibility (``src/foundation_model_extractor.py:12-35``).  The reference
    flushes the allocator after every chunk; the HIP path reuses one workspace, so this is a no-op."""
    return None
