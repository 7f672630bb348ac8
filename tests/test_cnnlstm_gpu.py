"""HIP CNN-LSTM forward vs golden vectors captured from the reference module and vs the oracle."""
import glob
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from weights import synth_input, synth_state_dict  # noqa: E402

from oracle import cnnlstm_oracle as co

CASES = sorted(glob.glob(os.path.join(HERE, "golden", "cnnlstm_d*.npz")))
TOL = 1e-4   # north_star: <= 1e-4 relative for float outputs


def _rel(a, b):
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)


def _model(D, C, H, act, sd):
    import torch
    from robust_speech_analysis_framework_amd.cnnlstm import CNNLSTM
    m = CNNLSTM(input_dim=D, cnn_out_channels=C, lstm_hidden_dim=H, activation_fn=act)
    missing, unexpected = m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    assert not unexpected and all(k.endswith("num_batches_tracked") for k in missing)
    return m.cuda().eval()


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[8:-4] for p in CASES])
def test_logits_match_reference_golden(rsaf_lib, path):
    import torch
    z = np.load(path)
    D, C, H, B, T, seed = [int(v) for v in z["meta"]]
    act = str(z["act"])
    m = _model(D, C, H, act, synth_state_dict(D, C, H, seed))
    x = torch.from_numpy(synth_input(B, T, D, seed + 1000)).cuda()
    got = m(x)
    torch.cuda.synchronize()
    assert got.shape == (B, 2)
    assert _rel(got.cpu().numpy(), z["logits"]) < TOL


def test_ragged_zero_padded_batch_matches_reference(rsaf_lib):
    import torch
    from robust_speech_analysis_framework_amd.cnnlstm import collate_zero_pad, eval_outputs
    z = np.load(os.path.join(HERE, "golden", "cnnlstm_ragged_pad.npz"))
    D, C, H, B, T, seed = [int(v) for v in z["meta"]]
    m = _model(D, C, H, "silu", synth_state_dict(D, C, H, seed))
    a, b = synth_input(1, 37, D, 2001)[0], synth_input(1, 64, D, 2002)[0]
    xp = collate_zero_pad([a, b])
    got = m(xp)
    alone = m(torch.from_numpy(a[None]).cuda())
    torch.cuda.synchronize()
    assert _rel(got.cpu().numpy(), z["logits_padded"]) < TOL
    assert _rel(alone.cpu().numpy(), z["logits_alone"]) < TOL
    prob, pred = eval_outputs(got)
    rp, rpred = co.eval_outputs(z["logits_padded"])
    assert np.array_equal(pred.cpu().numpy(), rpred) and np.allclose(prob.cpu().numpy(), rp, atol=1e-5)


def test_shipped_reading_checkpoint(rsaf_lib):
    """Trained weights of the reference (models/final_tuned_cnn_lstm_reading.pt, exported as data)."""
    import torch
    z = np.load(os.path.join(HERE, "golden", "cnnlstm_shipped_ckpt_logits.npz"))
    sd = dict(np.load(os.path.join(HERE, "golden", "cnnlstm_ckpt_reading_state.npz")))
    C, H = [int(v) for v in z["reading_dims"]]
    m = _model(768, C, H, str(z["reading_act"]), sd)
    got = m(torch.from_numpy(synth_input(2, 300, 768, 3000)).cuda())
    torch.cuda.synchronize()
    assert _rel(got.cpu().numpy(), z["reading_logits"]) < TOL


@pytest.mark.parametrize("B,T,C,H,act", [(17, 150, 128, 128, "silu"), (33, 75, 64, 64, "gelu"), (1, 2, 32, 64, "silu")])
def test_matches_oracle_on_larger_batches(rsaf_lib, B, T, C, H, act):
    import torch
    sd = synth_state_dict(768, C, H, 900 + B)
    x = synth_input(B, T, 768, 901 + B)
    m = _model(768, C, H, act, sd)
    got = m(torch.from_numpy(x).cuda())
    torch.cuda.synchronize()
    ref = co.forward_torch(sd, x, act)
    assert _rel(got.cpu().numpy(), ref) < TOL


def test_default_geometry_t1500(rsaf_lib):
    """BASELINE config 4 geometry (C=H=128, T=1500) at a batch the oracle finishes in seconds."""
    import torch
    sd = synth_state_dict(768, 128, 128, 4242)
    x = synth_input(4, 1500, 768, 4243)
    m = _model(768, 128, 128, "silu", sd)
    got = m(torch.from_numpy(x).cuda())
    torch.cuda.synchronize()
    assert _rel(got.cpu().numpy(), co.forward_torch(sd, x, "silu")) < TOL


def test_tap_shared_panel_image_matches_the_row_major_image(rsaf_lib, monkeypatch):
    """conv1 and the 1x1 shortcut read the input's fp16 planes as ONE k16-panel image of the zero-padded rows (tap k of output
    row t = image row t + k: csrc/cnnlstm.hip split_padded_panels_kernel, gemm_f16x3's a_tap_panels).  RSAF_CNN_ROWMAJOR=1 is
    the row-major image it replaced: the same products in the same order, so the logits must agree to rounding - on a ragged
    zero-padded batch whose sequence boundaries sit inside GEMM tiles, and both against the oracle."""
    import torch
    sd = synth_state_dict(768, 128, 128, 5151)
    x = synth_input(5, 700, 768, 5152)
    x[1, 333:] = 0.0
    x[3, 17:] = 0.0
    m = _model(768, 128, 128, "gelu", sd)
    outs = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("RSAF_CNN_ROWMAJOR", mode)
        outs[mode] = m(torch.from_numpy(x).cuda()).cpu().numpy()
    monkeypatch.delenv("RSAF_CNN_ROWMAJOR", raising=False)
    assert np.array_equal(outs["0"], outs["1"]) or _rel(outs["0"], outs["1"]) < 1e-6, _rel(outs["0"], outs["1"])
    assert _rel(outs["0"], co.forward_torch(sd, x, "gelu")) < TOL


def test_weights_repacked_after_update_and_errors(rsaf_lib):
    import torch
    from robust_speech_analysis_framework_amd import _lib
    from robust_speech_analysis_framework_amd.cnnlstm import CNNLSTM
    sd = synth_state_dict(16, 32, 64, 5)
    m = _model(16, 32, 64, "silu", sd)
    x = torch.from_numpy(synth_input(2, 20, 16, 6)).cuda()
    a = m(x).clone()
    with torch.no_grad():
        m.fc.bias.add_(1.0)
    b = m(x)
    torch.cuda.synchronize()
    assert np.allclose((b - a).cpu().numpy(), 1.0, atol=1e-5)
    with pytest.raises(ValueError):
        CNNLSTM(activation_fn="relu")
    out = m.train()(x)                       # training mode runs the HIP training step (tests/test_cnnlstm_train_gpu.py)
    assert out.requires_grad and out.shape == (2, 2)
    with pytest.raises(_lib.RsafError):
        m.train()(x.cpu())
    with pytest.raises(_lib.RsafError):
        m.eval()(x.cpu())


D16_CASES = [p for p in CASES if "_d16_" in os.path.basename(p)]


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[8:-4] for p in CASES])
def test_stage_outputs_match_reference_golden(rsaf_lib, path):
    """Every tensor a sub-module of the reference returns (forward hooks on res_block1 / res_block2 / lstm /
    attention_pooling of src/models.py, captured by make_cnnlstm_golden.py) against the HIP path's stage taps."""
    import torch
    from robust_speech_analysis_framework_amd.cnnlstm import cnnlstm_forward_stages
    z = np.load(path)
    D, C, H, B, T, seed = [int(v) for v in z["meta"]]
    m = _model(D, C, H, str(z["act"]), synth_state_dict(D, C, H, seed))
    st = cnnlstm_forward_stages(m, torch.from_numpy(synth_input(B, T, D, seed + 1000)).cuda())
    torch.cuda.synchronize()
    checked = 0
    for k in ("res1", "res2", "lstm", "pooled", "logits"):
        if k in z.files:
            assert st[k].shape == z[k].shape, k
            assert _rel(st[k].cpu().numpy(), z[k]) < TOL, k
            checked += 1
    assert checked >= 2


@pytest.mark.parametrize("path", D16_CASES, ids=[os.path.basename(p)[8:-4] for p in D16_CASES])
def test_standalone_blocks_match_reference_golden(rsaf_lib, path):
    """ResidualBlock.forward ([B, C, T] in and out, src/models.py:64-76) and AttentionPooling.forward (:94-107) called
    on their own, as a notebook could call the reference's sub-modules."""
    import torch
    z = np.load(path)
    D, C, H, B, T, seed = [int(v) for v in z["meta"]]
    m = _model(D, C, H, str(z["act"]), synth_state_dict(D, C, H, seed))
    x = torch.from_numpy(synth_input(B, T, D, seed + 1000)).cuda()
    r1 = m.res_block1(x.permute(0, 2, 1))                                  # conv1x1 + BN shortcut
    assert r1.shape == (B, C, T)
    assert _rel(r1.permute(0, 2, 1).cpu().numpy(), z["res1"]) < TOL
    pooled_in = torch.nn.functional.max_pool1d(torch.from_numpy(z["res1"]).permute(0, 2, 1), 2).cuda()
    r2 = m.res_block2(pooled_in)                                           # identity shortcut
    assert _rel(r2.permute(0, 2, 1).cpu().numpy(), z["res2"]) < TOL
    if 2 * H in (128, 256):
        p = m.attention_pooling(torch.from_numpy(z["lstm"]).cuda())
        assert _rel(p.cpu().numpy(), z["pooled"]) < TOL
    with pytest.raises(Exception):
        m.res_block1(x.permute(0, 2, 1).cpu())                             # no CPU fallback


def test_config_c4_full_size_batch_256(rsaf_lib):
    """BASELINE config C4 at its full size: x = randn(256, 1500, 768) seed 1234, default CNNLSTM() under seed 0.
    Four sampled rows against the REFERENCE module's outputs (make_cnnlstm_c4_golden.py) and the oracle; every row
    bit-identical to the same row computed in a sub-batch (eval-mode rows are independent)."""
    import torch
    from robust_speech_analysis_framework_amd.cnnlstm import CNNLSTM, cnnlstm_forward_stages
    z = np.load(os.path.join(HERE, "golden", "cnnlstm_c4_rows.npz"))
    rows = [int(r) for r in z["rows"]]
    xh = torch.randn(256, 1500, 768, generator=torch.Generator().manual_seed(1234))
    assert np.array_equal(xh[rows][:, :2, :8].numpy(), z["x_probe"])      # same synthetic input as the generator saw
    torch.manual_seed(0)
    m = CNNLSTM().cuda().eval()
    x = xh.cuda()
    st = cnnlstm_forward_stages(m, x)
    logits = m(x)
    torch.cuda.synchronize()
    assert logits.shape == (256, 2) and torch.equal(logits, st["logits"])
    got = logits.cpu().numpy()
    assert _rel(got[rows], z["logits"]) < TOL
    assert _rel(st["pooled"][rows].cpu().numpy(), z["pooled"]) < TOL
    assert _rel(st["lstm"][rows][:, ::50].cpu().numpy(), z["lstm_t"]) < TOL
    assert _rel(st["res2"][rows][:, ::50].cpu().numpy(), z["res2_t"]) < TOL
    assert _rel(st["res1"][rows][:, ::100].cpu().numpy(), z["res1_t"]) < TOL
    sd = {k: v.cpu().numpy() for k, v in m.state_dict().items()}
    assert _rel(got[rows], co.forward_torch(sd, xh[rows].numpy(), "silu")) < TOL
    for b0 in (0, 64, 128, 192):
        part = m(x[b0:b0 + 64])
        torch.cuda.synchronize()
        assert torch.equal(part, logits[b0:b0 + 64]), b0
    assert np.isfinite(got).all()
