"""Generate tests/golden/cnnlstm_*.npz from the REFERENCE module (src/models.py).

Run in the build container only (needs /root/reference):  python tests/golden/make_cnnlstm_golden.py
The reference never travels; what is committed is data: seeds, inputs, expected outputs.
"""
import hashlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference")
from src.models import CNNLSTM  # noqa: E402  (the reference)
from weights import synth_input, synth_state_dict  # noqa: E402

torch.set_num_threads(4)


def run_ref(model, x):
    st = {}
    hooks = [
        model.res_block1.register_forward_hook(lambda m, i, o: st.__setitem__("res1", o.permute(0, 2, 1).numpy().copy())),
        model.res_block2.register_forward_hook(lambda m, i, o: st.__setitem__("res2", o.permute(0, 2, 1).numpy().copy())),
        model.lstm.register_forward_hook(lambda m, i, o: st.__setitem__("lstm", o[0].numpy().copy())),
        model.attention_pooling.register_forward_hook(lambda m, i, o: st.__setitem__("pooled", o.numpy().copy())),
    ]
    with torch.no_grad():
        st["logits"] = model(torch.from_numpy(x)).numpy().copy()
    for h in hooks:
        h.remove()
    return st


def load_synth(model, sd):
    full = model.state_dict()
    for k, v in sd.items():
        full[k] = torch.from_numpy(v)
    model.load_state_dict(full)
    model.eval()


# (name, input_dim, C, H, act, B, T, seed); the state_dict is regenerated from the seed by the tests
CASES = [
    ("d16_c32_h64_silu", 16, 32, 64, "silu", 3, 64, 101),
    ("d16_c64_h128_gelu", 16, 64, 128, "gelu", 2, 37, 102),
    ("d16_c32_h64_silu_odd", 16, 32, 64, "silu", 1, 7, 103),
    ("d768_c32_h128_silu", 768, 32, 128, "silu", 2, 301, 104),
    ("d768_c128_h128_silu", 768, 128, 128, "silu", 2, 64, 105),
    ("d768_c128_h64_gelu", 768, 128, 64, "gelu", 3, 50, 106),
]

for name, D, C, H, act, B, T, seed in CASES:
    m = CNNLSTM(input_dim=D, cnn_out_channels=C, lstm_hidden_dim=H, activation_fn=act)
    load_synth(m, synth_state_dict(D, C, H, seed))
    x = synth_input(B, T, D, seed + 1000)
    st = run_ref(m, x)
    keep = {k: v for k, v in st.items() if D == 16 or k in ("pooled", "logits")}
    np.savez_compressed(os.path.join(HERE, f"cnnlstm_{name}.npz"), meta=np.array([D, C, H, B, T, seed]),
                        act=np.array(act), **keep)
    print(name, st["logits"].ravel()[:4])

# ragged batch, zero-padded without mask (collate_fn, src/dl_cv_strategies.py:81-84)
D, C, H, seed = 16, 32, 64, 107
m = CNNLSTM(input_dim=D, cnn_out_channels=C, lstm_hidden_dim=H)
load_synth(m, synth_state_dict(D, C, H, seed))
a, b = synth_input(1, 37, D, 2001)[0], synth_input(1, 64, D, 2002)[0]
xp = np.zeros((2, 64, D), np.float32)
xp[0, :37] = a
xp[1] = b
st_pad = run_ref(m, xp)
st_alone = run_ref(m, a[None])
np.savez_compressed(os.path.join(HERE, "cnnlstm_ragged_pad.npz"), meta=np.array([D, C, H, 2, 64, seed]),
                    logits_padded=st_pad["logits"], logits_alone=st_alone["logits"])
print("ragged", st_pad["logits"][0], st_alone["logits"][0])

# shipped checkpoints: logits on a seeded input; the 'reading' tensors are exported as data
out = {}
for tag in ("combined", "reading"):
    path = f"/root/reference/models/final_tuned_cnn_lstm_{tag}.pt"
    ck = torch.load(path, map_location="cpu", weights_only=True)
    hp = ck["hyperparameters"]
    m = CNNLSTM(input_dim=768, cnn_out_channels=hp["cnn_out_channels"], lstm_hidden_dim=hp["lstm_hidden_dim"],
                activation_fn=hp["activation_fn"], dropout_rate=hp["dropout_rate"])
    m.load_state_dict(ck["model_state_dict"])
    m.eval()
    x = synth_input(2, 300, 768, 3000)
    with torch.no_grad():
        lg = m(torch.from_numpy(x)).numpy()
    out[f"{tag}_logits"] = lg
    out[f"{tag}_sha256"] = np.array(hashlib.sha256(open(path, "rb").read()).hexdigest())
    out[f"{tag}_dims"] = np.array([hp["cnn_out_channels"], hp["lstm_hidden_dim"]])
    out[f"{tag}_act"] = np.array(hp["activation_fn"])
    print(tag, lg)
    if tag == "reading":
        sd = {k: v.numpy() for k, v in ck["model_state_dict"].items() if not k.endswith("num_batches_tracked")}
        np.savez_compressed(os.path.join(HERE, "cnnlstm_ckpt_reading_state.npz"), **sd)
np.savez_compressed(os.path.join(HERE, "cnnlstm_shipped_ckpt_logits.npz"), **out)
