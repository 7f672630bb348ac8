"""Workload for the PMC passes of the Praat low-pass kernels: the formant branch's whole-clip low-pass + 10 kHz interpolation of
256 x 30 s clips, twice (tools/pmc_summary.py turns the passes into per-kernel means)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from robust_speech_analysis_framework_amd import _lib, synth
from robust_speech_analysis_framework_amd.mshds import LP_SIG, _dev

clips = int(sys.argv[1]) if len(sys.argv) > 1 else 256
lib = _lib.load()
dev = torch.device("cuda:0")
wav = torch.from_numpy(synth.synth_batch(clips, 30.0, pool=8)).to(dev).reshape(-1)
n = 480000
lg = max(11, int(n + 2000 - 1).bit_length())
sigs = np.zeros(clips, dtype=LP_SIG)
for i in range(clips):
    sigs[i] = (i * n, i * n, i << (lg - 1), n, lg)
work = torch.empty(2 * (clips << (lg - 1)), dtype=torch.float64, device=dev)
lp = torch.empty(clips * n, dtype=torch.float64, device=dev)
sd = _dev(sigs, dev)
for _ in range(2):
    _lib.check(lib.rsaf_praat_lowpass_batch(_lib.ptr(wav), _lib.ptr(sd), clips, lg, 0.625, _lib.ptr(work), clips << (lg - 1), _lib.ptr(lp),
                                            _lib.stream_ptr(None)), "rsaf_praat_lowpass_batch")
torch.cuda.synchronize()
print(f"done: {clips} clips, nfft 2^{lg}, algorithmic bytes per launch set {clips * (1 << (lg - 1)) * 96:.3e}")
