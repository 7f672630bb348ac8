"""Time of the float64 openSMILE frame kernel after each phase (env RSAF_SMILE_STOP = k leaves every frame after phase k;
results are then wrong, the timing is valid).  One process per k (the value is read once per process).

    python tools/smile_phase.py            # prints ms per 1 000 x 30 s clips for k = 1..9 and the full kernel
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PHASES = {1: "window + FFT + magnitudes", 2: "spectral sums, 3 reductions", 3: "roll-off prefix sums", 4: "mel + MFCC",
          5: "peak enhancement + smoothing", 6: "spline (two affine scans)", 7: "octave-axis evaluation",
          8: "sub-harmonic summation", 9: "peaks + exp2 + compaction", 0: "full kernel (ranking, finaliser)"}

if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    from robust_speech_analysis_framework_amd import smile, synth
    n = int(os.environ.get("SMILE_PHASE_CLIPS", "1000"))
    base = np.stack([synth.synth_clip(k, 30.0) for k in range(16)])
    wav = torch.from_numpy(base[np.arange(n) % 16]).cuda()
    p = smile.pack_clips(wav)
    lib = smile._lib.load()
    lld = torch.empty((38, p.total_frames), dtype=torch.float64, device="cuda")
    cand = torch.zeros((p.total_frames, 6, 2), dtype=torch.float64, device="cuda")

    def run():
        smile._lib.check(lib.rsaf_smile_lld_batch(smile._lib.ptr(p.wav), smile._lib.ptr(p.clip_off), smile._lib.ptr(p.frame_off),
                                                  p.n_clips, p.max_frames, p.total_frames, p.fs, smile._lib.ptr(lld),
                                                  smile._lib.ptr(cand), None, None), "lld")
    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        run()
    e1.record()
    torch.cuda.synchronize()
    print(f"{e0.elapsed_time(e1) / 3:.2f}")
    sys.exit(0)

prev = 0.0
for k in list(range(1, 10)) + [0]:
    env = dict(os.environ, RSAF_SMILE_STOP=str(k))
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], capture_output=True, text=True, env=env)
    if r.returncode != 0:
        print(r.stderr[-2000:])
        sys.exit(1)
    ms = float(r.stdout.strip().splitlines()[-1])
    print(f"stop {k}: {ms:8.2f} ms  (+{ms - prev:6.2f})  {PHASES[k]}", flush=True)
    prev = ms
