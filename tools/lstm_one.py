"""One CNN-LSTM training step at batch 4 x 20 000 frames in a short loop, for `rocprofv3 --pmc` passes over the two
4-row recurrence kernels (lstm_rec4_kernel, lstm_bwd4_kernel)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from robust_speech_analysis_framework_amd import _lib
from robust_speech_analysis_framework_amd.cnnlstm import CNNLSTM

_lib.load()
torch.manual_seed(0)
m = CNNLSTM().to("cuda").train()
x = torch.randn((4, int(sys.argv[1]) if len(sys.argv) > 1 else 20000, 768), device="cuda")
y = torch.randint(0, 2, (4,), device="cuda")
for _ in range(2):
    m.zero_grad()
    torch.nn.CrossEntropyLoss()(m(x), y).backward()
torch.cuda.synchronize()
print("done")
