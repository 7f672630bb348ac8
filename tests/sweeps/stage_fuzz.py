"""Randomized parity sweeps of the openSMILE-chain and Wav2Vec2 -> CNN-LSTM HIP paths against the CPU oracles on random
clip lengths (tests/ holds the fixed cases and the tolerances reused here).  ``run()`` is called by
tests/test_sweeps_gpu.py (small) and by ``python tests/sweeps/stage_fuzz.py FIRST COUNT`` (large)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np


def run(first=8000, count=20, w2v2=True):
    import torch
    from oracle import smile_oracle as so, w2v2_oracle, cnnlstm_oracle
    from robust_speech_analysis_framework_amd import smile, synth
    from tests.test_smile_gpu import _check_rows, _check_pitch_chain, _assert_all_912
    rng = np.random.Generator(np.random.PCG64(first))
    durs = [float(rng.choice([rng.uniform(0.02, 0.2), rng.uniform(0.2, 2.0), rng.uniform(2.0, 12.0)])) for _ in range(count)]
    clips = [synth.synth_clip(first + k, d) for k, d in enumerate(durs)]

    # ---- openSMILE chain: all 38 rows and all 912 functionals (positions exact) ----
    p = smile.pack_clips(clips)
    lld, octv, cand = smile.smile_lld(p, octave_spectrum=True, return_candidates=True)
    f = smile.smile_functionals(lld, p)
    torch.cuda.synchronize()
    assert p.frames == [so.n_frames(len(c)) for c in clips]
    live = [c for c in clips if so.n_frames(len(c)) > 0]
    ref = np.concatenate([so.lld(c) for c in live], axis=1)
    P = so.Params(16000)
    g = lld.cpu().numpy()
    _check_pitch_chain(clips, p, g, octv.cpu().numpy(), cand.cpu().numpy(), P, min_voiced=0.0)
    _check_rows(g, ref, P)
    keep = [i for i, c in enumerate(clips) if so.n_frames(len(c)) > 0]
    llds = [so.lld(c) for c in live]
    _assert_all_912(f.cpu().numpy()[keep], np.stack([so.functionals(x) for x in llds]), lld_ref=llds)
    print(f"smile: {count} clips, {sum(p.frames)} frames: 38 rows + 912 functionals ok (tolerances of tests/test_smile_gpu.py)", flush=True)
    out = {"clips": count, "smile_frames": int(sum(p.frames))}
    if not w2v2:
        return out

    # ---- Wav2Vec2 (seeded random base weights) -> CNN-LSTM ----
    from robust_speech_analysis_framework_amd.w2v2 import W2V2Engine
    from robust_speech_analysis_framework_amd.w2v2_config import W2V2Config, random_state_dict
    from robust_speech_analysis_framework_amd.cnnlstm import CNNLSTM
    cfg = W2V2Config()
    sd = random_state_dict(cfg, 0)
    eng = W2V2Engine(cfg, sd, torch.device("cuda:0"))
    lens = [len(c) for c in clips]
    offs = np.concatenate([[0], np.cumsum(lens)])[:-1]
    wav = torch.from_numpy(np.concatenate(clips)).cuda()
    seq, frame_off = eng.extract_packed(wav, offs, lens)
    torch.cuda.synchronize()
    host = seq.cpu().numpy()
    torch.manual_seed(0)
    model = CNNLSTM().cuda().eval()
    sdm = {k: v.cpu().numpy() for k, v in model.state_dict().items()}
    worst, worst_l = 0.0, 0.0
    seqs = []
    for i, c in enumerate(clips):
        r = w2v2_oracle.extract_sequence(sd, cfg, c)
        a, b = int(frame_off[i]), int(frame_off[i + 1])
        if r is None:
            assert b == a, (i, durs[i])
            continue
        assert b - a == r.shape[0], (i, durs[i], b - a, r.shape)
        worst = max(worst, float(np.abs(host[a:b] - r).max() / np.abs(r).max()))
        seqs.append((host[a:b], r))
    print(f"w2v2: frame counts exact, worst rel-to-absmax {worst:.2e}", flush=True)
    assert worst <= 1e-4
    if seqs:
        x = cnnlstm_oracle.collate_zero_pad([r for _, r in seqs])
        ref_logits = np.asarray(cnnlstm_oracle.forward_torch(sdm, x, "silu"))
        got = model(torch.from_numpy(cnnlstm_oracle.collate_zero_pad([g_ for g_, _ in seqs])).cuda()).cpu().numpy()
        worst_l = float(np.abs(got - ref_logits).max() / np.abs(ref_logits).max())
        print(f"cnnlstm: ragged batch of {len(seqs)} (T {min(len(r) for _, r in seqs)}..{max(len(r) for _, r in seqs)}), logits rel {worst_l:.2e}", flush=True)
        assert worst_l <= 1e-4
    out.update({"w2v2_worst": worst, "logits_worst": worst_l})
    print("SUMMARY ok")
    return out


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 8000, int(sys.argv[2]) if len(sys.argv) > 2 else 20)
