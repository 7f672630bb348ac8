"""Generate tests/golden/cnnlstm_train_*.npz from the REFERENCE module (src/models.py) in training mode.

Run in the build container only (needs /root/reference):  python tests/golden/make_cnnlstm_train_golden.py
One step of the reference's inner training loop (src/dl_cv_strategies.py:241-243): model.train(), CrossEntropyLoss,
backward, Adam(lr).step().  The dropout probabilities are set to 0 so that the step is deterministic; BatchNorm runs
on batch statistics.  What is committed is data: seeds, loss, logits and, per tensor, evenly spaced samples + sum + sum of squares
(weights.sample_tensor) of the gradients, the updated buffers and the updated parameters.
"""
import os
import sys

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference")
from src.models import CNNLSTM  # noqa: E402  (the reference)
from weights import sample_tensor, synth_input, synth_state_dict  # noqa: E402

torch.set_num_threads(4)
torch.use_deterministic_algorithms(True)

# (name, input_dim, C, H, act, B, T, seed, lr)
CASES = [
    ("d16_c32_h64_silu", 16, 32, 64, "silu", 3, 20, 201, 1e-3),
    ("d16_c64_h128_gelu", 16, 64, 128, "gelu", 2, 13, 202, 3e-4),
    ("d768_c128_h128_silu", 768, 128, 128, "silu", 4, 33, 203, 1e-3),
    ("d32_c32_h64_identity_shortcut", 32, 32, 64, "silu", 2, 16, 204, 1e-3),
]

for name, D, C, H, act, B, T, seed, lr in CASES:
    m = CNNLSTM(input_dim=D, cnn_out_channels=C, lstm_hidden_dim=H, activation_fn=act, dropout_rate=0.0).double()
    full = m.state_dict()
    for k, v in synth_state_dict(D, C, H, seed).items():
        full[k] = torch.from_numpy(v).double()
    m.load_state_dict(full)
    for mod in m.modules():
        if isinstance(mod, nn.Dropout):
            mod.p = 0.0
    m.train()
    x = torch.from_numpy(synth_input(B, T, D, seed + 1000)).double()
    labels = torch.from_numpy(np.random.Generator(np.random.PCG64(seed + 2000)).integers(0, 2, B))
    opt = torch.optim.Adam(m.parameters(), lr=lr)
    opt.zero_grad()
    out = m(x)
    loss = nn.CrossEntropyLoss()(out, labels)
    loss.backward()
    grads = {"grad/" + k: sample_tensor(p.grad.numpy()) for k, p in m.named_parameters()}
    opt.step()
    after = {"after/" + k: sample_tensor(v.detach().numpy()) for k, v in m.state_dict().items()}
    np.savez_compressed(os.path.join(HERE, f"cnnlstm_train_{name}.npz"), meta=np.array([D, C, H, B, T, seed]),
                        act=np.array(act), lr=np.array(lr), labels=labels.numpy(), loss=np.array(loss.item()),
                        logits=out.detach().numpy(), **grads, **after)
    print(name, float(loss), out.detach().numpy().ravel()[:4])
