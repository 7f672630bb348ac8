"""Turn two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE; they do not fit one pass on gfx950) into
profiles/<round>/pmc_bench_traffic.json, the file bench.py reads for `roofline.traffic`, with the HBM traffic of the
Wav2Vec2 GEMM kernel PER SHAPE beside the algorithmic bytes of that shape.

  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_fetch --output-format csv -- python3 bench.py ...
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc_write --output-format csv -- python3 bench.py ...
  python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r04/pmc_bench_traffic.json \
         --config e2e --clips 1000 --seconds 30 --windows 2048 --command "<command>"

Shapes: gemm_f16x3 runs persistent workgroups (one per CU), so the grid size no longer tells its shapes apart; every dispatch
is labelled by its template variant (activation / outputs / residual, in the kernel name) and its position in the forward
call: [conv1..5 (GELU -> planes), conv6 (GELU -> fp32)] per window group, feature projection, then per layer qkv (-> planes),
out-projection (+R), ffn1 (GELU -> planes), ffn2 (+R).  The 512 x 128 tile configuration (the CNN-LSTM's GEMMs) is left out.
The calls of a step and their window counts are replayed from the engine's own batching (w2v2.forward_windows).

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): both counters are in KB;
on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes, so wide coalesced reads are doubled; WRITE_SIZE is exact.
"""
import argparse
import csv
import glob
import hashlib
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def kernel_sha():
    h = hashlib.sha256()
    for f in ("gemm_f16x3.hip", "gemm_f16x3.h", "w2v2.hip"):
        with open(os.path.join(ROOT, "robust_speech_analysis_framework_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def variant(name):
    """(tile config 'A' | 'N', act, fp32 out, planes out, residual) from the kernel's template arguments."""
    m = re.search(r"H3Cfg<\s*(\d+),\s*(\d+),\s*(\d+),\s*(\d+),\s*(\d+),\s*(\d+)\s*>,\s*(\d+),\s*(true|false|[01]),\s*(true|false|[01]),\s*(true|false|[01])", name)
    if not m:
        return None
    tb = lambda v: v in ("true", "1")                                       # noqa: E731
    cfg = "N" if m.group(2) == "1" else ("M" if m.group(3) == "4" else "A")  # 256 x 64 / 512 x 64 | 512 x 128 (CNN-LSTM) | 256 x 256
    return (cfg, int(m.group(7)), tb(m.group(8)), tb(m.group(9)), tb(m.group(10)))


def dispatches(root, counter):
    """[(dispatch id, kernel name, counter value)] of every gemm_f16x3 dispatch, in dispatch order."""
    files = glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {root}")
    out = []
    for fn in files:
        with open(fn, newline="") as f:
            for r in csv.DictReader(f):
                if r.get("Counter_Name") == counter and "gemm_f16x3_kernel" in r["Kernel_Name"]:
                    v = variant(r["Kernel_Name"])
                    if v is not None and v[0] == "M":                      # the CNN-LSTM's convolutions / input projections
                        continue
                    out.append((int(r.get("Dispatch_Id", 0) or 0), r["Kernel_Name"], float(r["Counter_Value"])))
    out.sort()
    return out


def plan_calls(clips, seconds, max_chunks):
    """Window lengths of every rsaf_w2v2_forward_ragged call of one step (the engine's batching, replayed)."""
    import numpy as np
    from robust_speech_analysis_framework_amd.w2v2_config import chunk_plan
    lens = []
    for _ in range(clips):
        lens += [l for _, l in chunk_plan(int(round(seconds * 16000)), 5, 1)]
    lens = np.sort(np.asarray(lens))[::-1]
    n_total = len(lens)
    n_calls = max(1, -(-n_total // max_chunks))
    per = min(max_chunks, ((-(-n_total // n_calls)) + 3) & ~3)
    return [lens[b0:b0 + per].tolist() for b0 in range(0, n_total, per)]


def frames(n):
    T = []
    for k, s in zip((10, 3, 3, 3, 3, 2, 2), (5, 2, 2, 2, 2, 2, 2)):
        n = (n - k) // s + 1 if n >= k else 0
        T.append(n)
    return T


def call_sequence(lens, conv_group=512, C=512, H=768, inter=3072, layers=12, pos_k=128, pos_g=16):
    """[(label, variant, algorithmic bytes)] of the gemm_f16x3 dispatches of one forward call, in launch order.
    Algorithmic bytes: every operand element once as two fp16 planes (4 B), outputs as fp32 (4 B) or planes (4 B), fp32
    residual 4 B."""
    n = len(lens)
    Tw = [frames(l) for l in lens]
    rows = sum(t[6] for t in Tw)
    seq = []
    n_groups = -(-n // conv_group)
    gstep = -(-n // n_groups)
    for g0 in range(0, n, gstep):
        grp = Tw[g0:g0 + gstep]
        for i in range(1, 7):
            k = 3 if i <= 4 else 2
            a = 4.0 * C * sum(t[i - 1] for t in grp)
            o = 4.0 * C * sum(t[i] for t in grp)
            seq.append((f"conv{i} (K = {k * C})", ("A", 1, i == 6, i != 6, False), a + o + 4.0 * C * k * C))
    alg = lambda nn, kk, resid: 4.0 * (rows * kk + nn * kk) + rows * nn * (4.0 + 4.0 * resid)      # noqa: E731
    seq.append(("feature projection", ("A", 0, True, False, False), alg(H, C, 0)))
    # (the positional convolution is its own kernel since round 4: posconv_f16x3_kernel, listed under other_kernels)
    for _ in range(layers):
        seq.append(("qkv (-> planes)", ("A", 0, False, True, False), alg(3 * H, H, 0)))
        seq.append(("attention out-proj (+residual)", ("A", 0, True, False, True), alg(H, H, 1)))
        seq.append(("ffn1 (GELU -> planes)", ("A", 1, False, True, False), alg(inter, H, 0)))
        seq.append(("ffn2 (+residual)", ("A", 0, True, False, True), alg(H, inter, 1)))
    return seq, rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_dir")
    ap.add_argument("write_dir")
    ap.add_argument("out")
    ap.add_argument("--config", default="e2e")
    ap.add_argument("--clips", type=int, default=1000)
    ap.add_argument("--seconds", type=float, default=30.0)
    ap.add_argument("--windows", type=int, default=2048)            # --w2v2-chunks-per-call of the run (recorded; bench.py matches it)
    ap.add_argument("--kernel", default="w2v2_gemm")
    ap.add_argument("--command", default="")
    ap.add_argument("--also", default="smile_lld_kernel,pitch_cand_kernel,pitch_cell_coef_kernel,pitch_brent_kernel,posconv_f16x3_kernel,lp_rows_kernel,layernorm_kernel,conv0_kernel,attn_f16x3_kernel",
                    help="other kernels (name substrings) whose FETCH / WRITE totals per launch are recorded beside the GEMM")
    a = ap.parse_args()
    fe, wr = dispatches(a.fetch_dir, "FETCH_SIZE"), dispatches(a.write_dir, "WRITE_SIZE")
    if len(fe) != len(wr):
        print(f"warning: {len(fe)} dispatches in the FETCH pass, {len(wr)} in the WRITE pass", file=sys.stderr)
    calls = plan_calls(a.clips, a.seconds, a.windows)
    step_seq = []
    for ci, lens in enumerate(calls):
        seq, rows = call_sequence(lens)
        step_seq += [(f"call {ci + 1}/{len(calls)} ({len(lens)} windows, {rows} rows): {lab}", var, alg, lab, ci) for lab, var, alg in seq]
    per = len(step_seq)
    n_steps = len(fe) // per if per else 0
    leftover = len(fe) - n_steps * per
    agg = defaultdict(lambda: [0.0, 0.0, 0, 0.0, None])             # label -> [fetch KB, write KB, launches, alg bytes, variant]
    mism = 0
    for k, (did, name, val) in enumerate(fe[:n_steps * per]):
        lab_full, var, alg, lab, ci = step_seq[k % per]
        if variant(name) != var:
            mism += 1
        key = (ci, lab)
        agg[key][0] += val
        agg[key][2] += 1
        agg[key][3] = alg
        agg[key][4] = var
    for k, (did, name, val) in enumerate(wr[:n_steps * per]):
        _, var, alg, lab, ci = step_seq[k % per]
        agg[(ci, lab)][1] += val
    per_shape, tot_f, tot_w, tot_l, tot_alg = [], 0.0, 0.0, 0, 0.0
    for (ci, lab), (f, w, n, alg, var) in sorted(agg.items(), key=lambda kv: -(2.0 * kv[1][0] + kv[1][1])):
        traffic = (2.0 * f + w) * 1024.0 / max(n, 1)
        per_shape.append({"call": ci + 1, "windows": len(calls[ci]), "shape": lab, "variant_act_f32_planes_residual": var[1:],
                          "launches": n, "hbm_bytes_per_launch_fetch_x2_plus_write": traffic,
                          "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": traffic / alg if alg else None})
        tot_f += f
        tot_w += w
        tot_l += n
        tot_alg += alg * n
    doc = {"source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over: {a.command}",
           "run": {"kernel": a.kernel, "config": a.config, "clips": a.clips, "clip_seconds": a.seconds, "w2v2_windows_per_call": a.windows,
                   "calls_per_step": [len(c) for c in calls], "kernel_sha": kernel_sha()},
           "labelling": {"gemm_dispatches_fetch_pass": len(fe), "dispatches_per_step": per, "steps_matched": n_steps,
                         "leftover_dispatches": leftover, "variant_mismatches": mism,
                         "note": "a dispatch is labelled by its position in the forward call; variant_mismatches counts positions "
                                 "whose kernel template arguments differ from the expected variant (0 = the replay matches the run)"},
           "gemm_launches": tot_l, "fetch_KB_raw": tot_f, "write_KB": tot_w,
           "traffic_bytes_per_launch_fetch_x2_plus_write": (2.0 * tot_f + tot_w) * 1024.0 / max(tot_l, 1),
           "algorithmic_bytes_per_launch": tot_alg / max(tot_l, 1),
           "traffic_over_algorithmic": ((2.0 * tot_f + tot_w) * 1024.0) / tot_alg if tot_alg else None,
           "per_shape": per_shape}
    others = {}
    for sub in [x for x in a.also.split(",") if x]:
        tot = {"FETCH_SIZE": [0.0, 0], "WRITE_SIZE": [0.0, 0]}
        for root, counter in ((a.fetch_dir, "FETCH_SIZE"), (a.write_dir, "WRITE_SIZE")):
            for fn in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
                with open(fn, newline="") as f:
                    for r in csv.DictReader(f):
                        if r.get("Counter_Name") == counter and sub in r["Kernel_Name"]:
                            tot[counter][0] += float(r["Counter_Value"])
                            tot[counter][1] += 1
        n = max(tot["FETCH_SIZE"][1], tot["WRITE_SIZE"][1])
        if n:
            others[sub] = {"launches": n, "fetch_KB_raw_per_launch": tot["FETCH_SIZE"][0] / max(tot["FETCH_SIZE"][1], 1),
                           "write_KB_per_launch": tot["WRITE_SIZE"][0] / max(tot["WRITE_SIZE"][1], 1),
                           "hbm_bytes_per_launch_fetch_x2_plus_write": (2.0 * tot["FETCH_SIZE"][0] / max(tot["FETCH_SIZE"][1], 1)
                                                                        + tot["WRITE_SIZE"][0] / max(tot["WRITE_SIZE"][1], 1)) * 1024.0}
    doc["other_kernels"] = others
    with open(a.out, "w") as f:
        json.dump(doc, f, indent=1)
    print(json.dumps({k: doc[k] for k in ("gemm_launches", "traffic_bytes_per_launch_fetch_x2_plus_write", "traffic_over_algorithmic", "labelling")}))
    for r in per_shape[:16]:
        print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items()})


if __name__ == "__main__":
    main()
