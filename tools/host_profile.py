"""Host-side profile of one pipeline step (profiling aid; not part of the product path): cProfile over `Pipeline.run`
with the GPU work queued asynchronously, so that what shows up is the Python / numpy / ctypes time that has to stay
below the GPU time for the step to be GPU-bound.

  python tools/host_profile.py [stages] [clips]       e.g.  python tools/host_profile.py mshds,smile 1000
"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from robust_speech_analysis_framework_amd import pipeline, synth

stages = pipeline.resolve_stages(sys.argv[1]) if len(sys.argv) > 1 else ["mshds", "smile"]
clips = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
dev = torch.device("cuda:0")
base = [synth.synth_clip(m, 30.0) for m in range(64)]
wav = torch.from_numpy(np.stack([base[i % 64] for i in range(clips)])).to(dev)
pipe = pipeline.Pipeline(stages, device=dev, seconds=30.0)
pipe.run(wav)
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter()
    pipe.run(wav)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"step {rep}: host returned after {1e3 * (t1 - t0):.1f} ms, GPU drained {1e3 * (t2 - t1):.1f} ms later", flush=True)
pr = cProfile.Profile()
pr.enable()
pipe.run(wav)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
st.sort_stats("tottime").print_stats(30)
