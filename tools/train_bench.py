"""Time one CNN-LSTM training step (zero_grad / forward / CrossEntropy / backward / Adam.step) on the HIP path, with the
per-kernel-family event breakdown of rsaf_prof_*.  Shapes follow the reference's loops: batch 4 (Optuna inner loop,
src/dl_cv_strategies.py:234) and batch 8 (final training, :265), one vstack-ed session per participant.

    python tools/train_bench.py [--torch]     # --torch: also time plain PyTorch-ROCm (nn.LSTM / F.conv1d) on the same GPU
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from robust_speech_analysis_framework_amd import _lib
from robust_speech_analysis_framework_amd.cnnlstm import CNNLSTM

ap = argparse.ArgumentParser()
ap.add_argument("--torch", action="store_true")
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--only", type=int, default=-1, help="index of the single shape to run")
args = ap.parse_args()
_lib.load()

SHAPES = [(4, 4378, "reading task, batch 4"), (4, 20000, "interview sessions, batch 4"), (8, 20000, "interview sessions, batch 8")]


def torch_model():
    """The same architecture from stock torch.nn modules (what the reference runs on a ROCm PyTorch)."""
    import torch.nn as nn
    import torch.nn.functional as F

    class Block(nn.Module):
        def __init__(self, cin, cout):
            super().__init__()
            self.conv1, self.bn1 = nn.Conv1d(cin, cout, 3, padding=1), nn.BatchNorm1d(cout)
            self.conv2, self.bn2 = nn.Conv1d(cout, cout, 3, padding=1), nn.BatchNorm1d(cout)
            self.drop = nn.Dropout(0.2)
            self.sc = nn.Sequential(nn.Conv1d(cin, cout, 1), nn.BatchNorm1d(cout)) if cin != cout else nn.Sequential()

        def forward(self, x):
            o = self.drop(F.silu(self.bn1(self.conv1(x))))
            return F.silu(self.bn2(self.conv2(o)) + self.sc(x))

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.b1, self.b2 = Block(768, 128), Block(128, 128)
            self.lstm = nn.LSTM(128, 128, 2, batch_first=True, bidirectional=True, dropout=0.5)
            self.att, self.drop, self.fc = nn.Linear(256, 1), nn.Dropout(0.5), nn.Linear(256, 2)

        def forward(self, x):
            x = self.b2(F.max_pool1d(self.b1(x.permute(0, 2, 1)), 2)).permute(0, 2, 1)
            o, _ = self.lstm(x)
            return self.fc(self.drop(torch.sum(o * F.softmax(self.att(o), dim=1), dim=1)))

    return Net()


def run(model, B, T, steps, prof):
    x = torch.randn((B, T, 768), device="cuda")
    y = torch.randint(0, 2, (B,), device="cuda")
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    loss_fn = torch.nn.CrossEntropyLoss()

    def one():
        opt.zero_grad()
        loss = loss_fn(model(x), y)
        loss.backward()
        opt.step()
        return loss

    one()
    torch.cuda.synchronize()
    if prof:
        _lib.prof_begin()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return dt, (_lib.prof_end() if prof else None)


for B, T, tag in (SHAPES if args.only < 0 else [SHAPES[args.only]]):
    m = CNNLSTM().to("cuda").train()
    dt, rec = run(m, B, T, args.steps, True)
    print(f"== {tag}: B={B} T={T}: {dt * 1e3:.1f} ms per step (HIP path)", flush=True)
    for k, v in sorted(rec.items(), key=lambda kv: -kv[1]["ms"]):
        extra = f"  {v['flops'] / (v['ms'] * 1e-3) / 1e12:6.1f} TFLOP/s" if v["flops"] > 0 else ""
        print(f"   {k:26s} {v['launches'] / args.steps:6.0f} launches  {v['ms'] / args.steps:9.2f} ms{extra}", flush=True)
    if args.torch:
        tm = torch_model().to("cuda").train()
        dt2, _ = run(tm, B, T, args.steps, False)
        print(f"   plain PyTorch-ROCm (MIOpen LSTM) on the same GPU: {dt2 * 1e3:.1f} ms per step  ->  x{dt2 / dt:.2f}", flush=True)
