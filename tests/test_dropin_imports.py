"""The reference's import paths and signatures resolve to the MI355X implementation (no GPU needed)."""
import inspect


def test_src_models_exports():
    from src.models import AttentionPooling, CNNLSTM, ResidualBlock, get_activation_fn
    sig = inspect.signature(CNNLSTM.__init__)
    assert list(sig.parameters)[1:] == ["input_dim", "num_classes", "cnn_out_channels", "lstm_hidden_dim",
                                        "lstm_layers", "dropout_rate", "activation_fn"]
    d = {k: v.default for k, v in sig.parameters.items() if k != "self"}
    assert d == {"input_dim": 768, "num_classes": 2, "cnn_out_channels": 128, "lstm_hidden_dim": 128,
                 "lstm_layers": 2, "dropout_rate": 0.5, "activation_fn": "silu"}
    assert callable(get_activation_fn("gelu")) and ResidualBlock and AttentionPooling


def test_extractor_signatures_match_reference():
    from src.foundation_model_extractor import extract_wav2vec2_embeddings, extract_wav2vec2_sequences
    from src.opensmile_extractor import extract_opensmile_features
    p = inspect.signature(extract_opensmile_features).parameters
    assert list(p)[:5] == ["input_df", "opensmile_exe_path", "config_file_path", "audio_file_column", "verbose"]
    assert p["audio_file_column"].default == "filepath" and p["verbose"].default is True
    q = inspect.signature(extract_wav2vec2_sequences).parameters
    assert list(q)[:6] == ["input_df", "model_name", "audio_file_column", "chunk_seconds", "overlap_seconds", "verbose"]
    assert (q["model_name"].default, q["chunk_seconds"].default, q["overlap_seconds"].default) == \
        ("facebook/wav2vec2-base-960h", 5, 1)
    assert list(inspect.signature(extract_wav2vec2_embeddings).parameters) == ["input_df", "kwargs"]


def test_product_never_imports_oracle():
    """oracle/ is test infrastructure: no module of the product may import it."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bad = []
    for d in ("robust_speech_analysis_framework_amd", "src"):
        for dirpath, _, files in os.walk(os.path.join(root, d)):
            for f in files:
                if f.endswith(".py"):
                    txt = open(os.path.join(dirpath, f)).read()
                    if re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M):
                        bad.append(os.path.join(dirpath, f))
    assert not bad, bad
