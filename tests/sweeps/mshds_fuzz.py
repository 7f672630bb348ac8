"""Randomized parity sweep of the MSHDS HIP path against the CPU oracle: random clip lengths, seeds outside the fixed
test set.  ``run()`` is called by tests/test_sweeps_gpu.py (small) and by ``python tests/sweeps/mshds_fuzz.py FIRST COUNT``
(large, on the GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np


def _oracle(args):
    k, seconds = args
    from oracle import mshds_oracle
    from robust_speech_analysis_framework_amd import synth
    r, rng = mshds_oracle.extract(synth.synth_clip(k, seconds))
    return r, tuple(rng)


def run(first=5000, count=16, min_s=1.2, max_s=3.5, verbose=True, workers=8):
    import concurrent.futures as cf
    import multiprocessing as mp
    import torch
    from robust_speech_analysis_framework_amd import synth
    from robust_speech_analysis_framework_amd.mshds import MshdsEngine, FEATURE_NAMES
    rng = np.random.Generator(np.random.PCG64(first))
    secs = [float(rng.uniform(min_s, max_s)) for _ in range(count)]
    clips = [synth.synth_clip(first + k, secs[k]) for k in range(count)]
    lens = [len(c) for c in clips]
    offs = np.concatenate([[0], np.cumsum(lens)])[:-1]
    wav = torch.from_numpy(np.concatenate(clips)).cuda()
    out, ranges = MshdsEngine().extract_packed(wav, [int(o) for o in offs], lens)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    with cf.ProcessPoolExecutor(max_workers=min(workers, count), mp_context=mp.get_context("spawn")) as ex:
        refs = list(ex.map(_oracle, [(first + k, secs[k]) for k in range(count)]))
    worst, bad = 0.0, 0
    for i, (ref, rng_ref) in enumerate(refs):
        ok_rng = rng_ref == tuple(ranges[i])
        nanok = np.array_equal(np.isnan(got[i]), np.isnan(ref))
        m = ~np.isnan(ref) & ~np.isnan(got[i])
        rel = np.abs(got[i][m] - ref[m]) / np.maximum(np.abs(ref[m]), 1e-3)
        w = float(rel.max()) if m.any() else 0.0
        worst = max(worst, w)
        good = ok_rng and nanok and w <= 1e-4
        bad += not good
        if verbose or not good:
            names = [n for n, mm in zip(FEATURE_NAMES, m) if mm]
            j = int(np.argmax(rel)) if m.any() else -1
            print(f"clip {first + i} ({secs[i]:.2f} s): range {ranges[i]} vs {rng_ref}, nan pattern {nanok}, worst rel {w:.3e}"
                  f"{'' if good else ' at ' + (names[j] if j >= 0 else '-') + '  <-- MISMATCH'}", flush=True)
    print(f"SUMMARY clips {count} mismatches {bad} worst_rel {worst:.3e}")
    return {"clips": count, "mismatches": bad, "worst_rel": worst}


if __name__ == "__main__":
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    r = run(first, count, float(os.environ.get("FUZZ_MIN_S", "1.2")), float(os.environ.get("FUZZ_MAX_S", "3.5")))
    sys.exit(1 if r["mismatches"] else 0)
