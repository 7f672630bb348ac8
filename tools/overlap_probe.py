"""Do MFMA-bound GEMM blocks and fp64 pitch-frame blocks co-reside on a CU?  (tool, not product code)

Times a burst of Wav2Vec2-FFN GEMMs on one stream, a burst of pitch analyses on another, alone and together.
together ~ max(alone) -> the dispatcher co-schedules them;  together ~ sum -> they serialise (LDS / VGPR limits).
"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from robust_speech_analysis_framework_amd import _lib, ops, synth
from robust_speech_analysis_framework_amd.mshds import MshdsEngine

dev = torch.device("cuda:0")
_lib.load()
M, N, K = 256 * 249, 3072, 768
A = torch.randn((M, K), device=dev); W = torch.randn((N, K), device=dev); out = torch.empty((M, N), device=dev)
clips = 64
wav = torch.from_numpy(synth.synth_batch(clips, 30.0, pool=8)).to(dev).reshape(-1)
n_s = wav.numel() // clips
offs = np.arange(clips, dtype=np.int64) * n_s; lens = [n_s] * clips
eng = MshdsEngine(dev)
gpeak = eng.clip_peaks(wav, offs, lens)
cfgs = {
    "cc_pulses(21KB LDS)": dict(time_step=0.005, floor=75.0, ceiling=500.0, periods=1.0, is_cc=True),
    "ac_75_500(~30KB)": dict(time_step=0.005, floor=75.0, ceiling=500.0),
    "ac_wide(~35KB)": dict(time_step=0.005, floor=50.0, ceiling=600.0),
}
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

def gemm_burst(n=24):
    with torch.cuda.stream(s1):
        for _ in range(n):
            ops.linear(A, W, out=out)

def pitch_burst(kw, n=4):
    torch.cuda.set_device(dev)
    with torch.cuda.stream(s2):
        for _ in range(n):
            eng.pitch(wav, offs, lens, gpeak, stream=s2, **kw)

def timed(fns):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=f) for f in fns]
    [t.start() for t in th]; [t.join() for t in th]
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3

for name, kw in cfgs.items():
    pitch_burst(kw, 1); gemm_burst(2); torch.cuda.synchronize()
    tg = timed([gemm_burst]); tp = timed([lambda: pitch_burst(kw)]); tb = timed([gemm_burst, lambda: pitch_burst(kw)])
    print(f"{name:22s} gemm {tg:7.1f} ms  pitch {tp:7.1f} ms  together {tb:7.1f} ms  (sum {tg+tp:7.1f}, max {max(tg,tp):7.1f})", flush=True)
