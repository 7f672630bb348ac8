"""Turn two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE; they do not fit one pass on gfx950) into
profiles/<round>/pmc_bench_traffic.json, the file bench.py reads for `roofline.traffic`, with the HBM traffic of the
Wav2Vec2 GEMM kernel PER SHAPE (dispatches grouped by grid size) beside the algorithmic bytes of that shape.

  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_fetch --output-format csv -- python3 bench.py ...
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc_write --output-format csv -- python3 bench.py ...
  python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r02/pmc_bench_traffic.json \
         --config e2e --clips 1000 --windows 2048 --command "<command>"

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): both counters are in KB;
on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes, so wide coalesced reads are doubled; WRITE_SIZE is exact.
"""
import argparse
import csv
import glob
import hashlib
import json
import os
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_sha():
    h = hashlib.sha256()
    for f in ("gemm_f16x3.hip", "gemm_f16x3.h", "w2v2.hip"):
        with open(os.path.join(ROOT, "robust_speech_analysis_framework_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def collect(root, counter):
    """{kernel name: {shape key: [sum of counter, dispatches]}}.  Shape key = grid size, and for the three GEMMs that
    share the N = 768 grid (feature projection, attention out-projection, ffn2) the launch that preceded them in
    dispatch order (the encoder issues qkv, out-proj, ffn1, ffn2 per layer): grid size alone cannot tell them apart."""
    agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    files = glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {root}")
    for fn in files:
        with open(fn, newline="") as f:
            rows = [r for r in csv.DictReader(f) if r.get("Counter_Name") == counter]
        rows.sort(key=lambda r: int(r.get("Dispatch_Id", 0) or 0))
        prev = None
        for row in rows:
            k = row["Kernel_Name"]
            g = int(row.get("Grid_Size", 0) or 0)
            key = g
            if "gemm_f16x3" in k:
                key = (g, prev)
                prev = g
            a = agg[k][key]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
    return agg


def w2v2_shapes(n_windows, chunk_len=80000, conv_group=512):
    """Wav2Vec2-base GEMMs of one sub-batch of n equal windows: {grid size (threads): (label, algorithmic bytes per launch)}.
    gemm_f16x3: 256 x 256 tiles, 512 threads per workgroup; the conv GEMMs are batched over groups of 512 windows
    (grid.y), the encoder GEMMs run on all rows at once.  Algorithmic bytes: every operand element once as two fp16
    planes (6 B), outputs as fp32 (4 B) or planes (6 B), fp32 residual 4 B."""
    T, t = [], chunk_len
    for k, s in zip((10, 3, 3, 3, 3, 2, 2), (5, 2, 2, 2, 2, 2, 2)):
        t = (t - k) // s + 1
        T.append(t)
    C, H, inter = 512, 768, 3072
    out = {}
    tiles = lambda m, n: ((m + 255) // 256) * ((n + 255) // 256)           # noqa: E731
    conv_grid = {}
    n_groups = (n_windows + conv_group - 1) // conv_group
    gw = (n_windows + n_groups - 1) // n_groups                  # balanced window groups (w2v2.hip)
    for i in range(1, 7):
        k = 3 if i <= 4 else 2
        conv_grid[i] = tiles(T[i], C) * gw * 512
        out[conv_grid[i]] = (f"conv{i} [{T[i]} x {k * C}] x [{C}] x {gw} windows",
                             gw * (6.0 * T[i - 1] * C + (4.0 if i == 6 else 6.0) * T[i] * C) + 6.0 * C * k * C)
    rows = n_windows * T[6]
    g = {n: tiles(rows, n) * 512 for n in (H, 3 * H, inter)}
    alg = lambda n, k, out_b, resid: 6.0 * (rows * k + n * k) + rows * n * (out_b + 4.0 * resid)      # noqa: E731
    out[g[3 * H]] = (f"qkv [{rows} x {H}] x [{3 * H}]", alg(3 * H, H, 4.0, 0))
    out[g[inter]] = (f"ffn1 (GELU -> planes) [{rows} x {H}] x [{inter}]", alg(inter, H, 6.0, 0))
    out[(g[H], g[3 * H])] = (f"attention out-proj (+residual) [{rows} x {H}] x [{H}]", alg(H, H, 4.0, 1))
    out[(g[H], g[inter])] = (f"ffn2 (+residual) [{rows} x {inter}] x [{H}]", alg(H, inter, 4.0, 1))
    out[(g[H], conv_grid[6])] = (f"feature projection [{rows} x {C}] x [{H}]", alg(H, C, 4.0, 0))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_dir")
    ap.add_argument("write_dir")
    ap.add_argument("out")
    ap.add_argument("--config", default="e2e")
    ap.add_argument("--clips", type=int, default=1000)
    ap.add_argument("--windows", type=int, default=2048)            # --w2v2-chunks-per-call of the run (recorded; bench.py matches it)
    ap.add_argument("--shape-windows", type=int, default=0)         # windows per call the engine actually used (balanced sub-batches)
    ap.add_argument("--kernel", default="w2v2_gemm")
    ap.add_argument("--command", default="")
    a = ap.parse_args()
    fe, wr = collect(a.fetch_dir, "FETCH_SIZE"), collect(a.write_dir, "WRITE_SIZE")
    gemm = [k for k in set(fe) | set(wr) if "gemm_f16x3" in k]
    shapes = w2v2_shapes(a.shape_windows or a.windows)
    per_shape, tot_f, tot_w, tot_l = [], 0.0, 0.0, 0
    grids = sorted({g for k in gemm for g in list(fe.get(k, {})) + list(wr.get(k, {}))}, key=str)
    # collapse (grid, previous grid) keys that the shape table does not distinguish
    def canon(key):
        return key if key in shapes else (key[0] if isinstance(key, tuple) else key)
    merged_f, merged_w = defaultdict(lambda: [0.0, 0]), defaultdict(lambda: [0.0, 0])
    for src, dst in ((fe, merged_f), (wr, merged_w)):
        for k in gemm:
            for key, (v, n) in src.get(k, {}).items():
                dst[canon(key)][0] += v
                dst[canon(key)][1] += n
    fe = {"gemm": merged_f}
    wr = {"gemm": merged_w}
    gemm_names = gemm
    gemm = ["gemm"]
    grids = sorted(set(merged_f) | set(merged_w), key=str)
    for g in grids:
        f = sum(fe[k][g][0] for k in gemm if g in fe.get(k, {}))
        w = sum(wr[k][g][0] for k in gemm if g in wr.get(k, {}))
        n = max(sum(fe[k][g][1] for k in gemm if g in fe.get(k, {})), sum(wr[k][g][1] for k in gemm if g in wr.get(k, {})))
        traffic = (2.0 * f + w) * 1024.0 / max(n, 1)
        label, alg = shapes.get(g, (None, None))
        per_shape.append({"grid_threads": g[0] if isinstance(g, tuple) else g, "launches": n, "hbm_bytes_per_launch_fetch_x2_plus_write": traffic,
                          "shape": label, "algorithmic_bytes_per_launch": alg,
                          "traffic_over_algorithmic": (traffic / alg) if alg else None})
        tot_f += f
        tot_w += w
        tot_l += n
    per_shape.sort(key=lambda r: -r["launches"] * r["hbm_bytes_per_launch_fetch_x2_plus_write"])
    doc = {"source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over: {a.command}",
           "run": {"kernel": a.kernel, "config": a.config, "clips": a.clips, "w2v2_windows_per_call": a.windows, "w2v2_windows_per_call_actual": a.shape_windows or a.windows,
                   "kernel_sha": kernel_sha()},
           "gemm_kernels": gemm_names, "gemm_launches": tot_l, "fetch_KB_raw": tot_f, "write_KB": tot_w,
           "traffic_bytes_per_launch_fetch_x2_plus_write": (2.0 * tot_f + tot_w) * 1024.0 / max(tot_l, 1),
           "note": "all gemm_f16x3 dispatches of the run (full-window and tail-window sub-batches); "
                   "per_shape separates them by grid size, shapes of the full-window sub-batches are labelled",
           "per_shape": per_shape}
    with open(a.out, "w") as f:
        json.dump(doc, f, indent=1)
    print(json.dumps({k: doc[k] for k in ("gemm_launches", "traffic_bytes_per_launch_fetch_x2_plus_write")}))
    for r in per_shape[:14]:
        print(r)


if __name__ == "__main__":
    main()
