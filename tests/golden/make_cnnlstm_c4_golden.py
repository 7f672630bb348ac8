"""Generate tests/golden/cnnlstm_c4_rows.npz from the REFERENCE module (src/models.py).

BASELINE config C4: x = torch.randn(256, 1500, 768, generator=manual_seed(1234)), default CNNLSTM()
weights under torch.manual_seed(0), eval mode.  Batch rows are independent in eval mode, so the reference is run on
four sampled rows only; the committed data are those rows' per-stage outputs, their indices and a fingerprint of the
default-initialised state_dict (per-tensor float64 sum and sum of squares) that pins "CNNLSTM() under seed 0".
Run in the build container only (needs /root/reference):  python tests/golden/make_cnnlstm_c4_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
from src.models import CNNLSTM  # noqa: E402  (the reference)

torch.set_num_threads(8)
ROWS = [0, 85, 170, 255]
x = torch.randn(256, 1500, 768, generator=torch.Generator().manual_seed(1234))
torch.manual_seed(0)
m = CNNLSTM().eval()
st = {}
hooks = [m.res_block1.register_forward_hook(lambda mod, i, o: st.__setitem__("res1", o.permute(0, 2, 1).numpy().copy())),
         m.res_block2.register_forward_hook(lambda mod, i, o: st.__setitem__("res2", o.permute(0, 2, 1).numpy().copy())),
         m.lstm.register_forward_hook(lambda mod, i, o: st.__setitem__("lstm", o[0].numpy().copy())),
         m.attention_pooling.register_forward_hook(lambda mod, i, o: st.__setitem__("pooled", o.numpy().copy()))]
with torch.no_grad():
    logits = m(x[ROWS]).numpy()
names, fp = [], []
for k, v in m.state_dict().items():
    if k.endswith("num_batches_tracked"):
        continue
    a = v.double().numpy()
    names.append(k)
    fp.append([a.sum(), (a * a).sum()])
# res1 of row 0 only (0.77 MB) and strided samples of the others keep the fixture small
np.savez_compressed(os.path.join(HERE, "cnnlstm_c4_rows.npz"), rows=np.array(ROWS), logits=logits, pooled=st["pooled"],
                    lstm_t=st["lstm"][:, ::50], res2_t=st["res2"][:, ::50], res1_t=st["res1"][:, ::100],
                    x_probe=x[ROWS][:, :2, :8].numpy(), sd_names=np.array(names), sd_fingerprint=np.array(fp))
print(logits)
