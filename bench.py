#!/usr/bin/env python3
"""Headline benchmark: audio-seconds processed per second on synthetic 16 kHz mono 30 s clips.

A "step" is one pass of the hot path over one batch of clips that is already resident in HBM:
every built stage of  extract (openSMILE-style chain, MSHDS, Wav2Vec2 frames) -> CNN-LSTM forward.
One process per GPU; clips shard across ranks with no data-path collective, the per-clip result
rows are all-gathered once per step (RCCL) when N > 1.  Rank 0 prints ONE JSON line.

    python bench.py --gpus 1 --steps 2 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense fp32 matrix peak


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--clips", type=int, default=1000, help="clips per GPU (weak scaling)")
    ap.add_argument("--seconds", type=float, default=30.0)
    ap.add_argument("--pool", type=int, default=8, help="distinct synthetic clips tiled to --clips")
    ap.add_argument("--stages", type=str, default="all")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-clips", type=int, default=0, help="0 = auto (about 10-30 s of CPU work)")
    ap.add_argument("--w2v2-chunks-per-call", type=int, default=2048,
                    help="Wav2Vec2 windows per sub-batch (workspace 118 GiB of the 288 GB at 2048; the larger GEMMs lose less to the "
                         "last partial wave of tiles: stage time -2.8 %% against 256)")
    ap.add_argument("--overlap", action="store_true",
                    help="run the MSHDS stage on a second HIP stream beside Wav2Vec2 (+6 %% throughput, but per-kernel "
                         "event times then include time-sharing, so the roofline object is only clean without it)")
    return ap.parse_args()


def usable_cpus() -> int:
    """CPUs this process may actually use: affinity mask capped by the cgroup CPU quota (the GPU box
    shows 256 host CPUs but grants a share of them; oversubscribing torch threads stalls for minutes)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // per))
        except (OSError, ValueError, IndexError):
            continue
    return max(1, min(n, int(os.environ.get("RSAF_CPU_THREADS", "16"))))


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def cpu_baseline(stages, seconds, sample_clips):
    """Time the CPU oracle (kind "port") on a bounded sample of the same synthetic workload.

    This is the only place outside tests/ and smoke() that touches oracle/: it is the reported
    baseline, never the product path.  DSP stages are numpy (one core), the model stages are
    torch-CPU float32 with every host core, exactly the ops the reference's CPU path dispatches."""
    import numpy as np
    import torch
    from robust_speech_analysis_framework_amd import synth
    cores = usable_cpus()
    torch.set_num_threads(cores)
    n = sample_clips or 1
    log(f"cpu_baseline: {n} clip(s) on {cores} thread(s)")
    clips = [synth.synth_clip(900000 + k, seconds) for k in range(n)]
    parts, total, ref = {}, 0.0, {"clip0": clips[0]}
    if "smile" in stages:
        from oracle import smile_oracle
        t0 = time.perf_counter()
        ref["smile"] = [smile_oracle.extract(c) for c in clips][0]
        parts["smile"] = time.perf_counter() - t0
        log(f"cpu_baseline smile {parts['smile']:.2f} s")
    if "mshds" in stages:
        from oracle import mshds_oracle
        sub = clips[0][:int(16000 * min(seconds, 5.0))]          # bounded: 5 s of one clip (Python oracle)
        t0 = time.perf_counter()
        ref["mshds"], _ = mshds_oracle.extract(sub)
        ref["mshds_input"] = sub
        dt_m = time.perf_counter() - t0
        parts["mshds"] = dt_m * (n * seconds) / (len(sub) / 16000.0)   # scaled to the sample's audio-seconds
        log(f"cpu_baseline mshds {dt_m:.2f} s for {len(sub) / 16000.0:g} audio-s (scaled to {parts['mshds']:.1f} s)")
    seqs = None
    if "w2v2" in stages:
        from oracle import w2v2_oracle
        from robust_speech_analysis_framework_amd.w2v2_config import W2V2Config, random_state_dict
        cfg = W2V2Config()
        sd = random_state_dict(cfg, 0)
        w2v2_oracle.extract_sequence(sd, cfg, clips[0][:16000])          # warm-up
        t0 = time.perf_counter()
        seqs = [w2v2_oracle.extract_sequence(sd, cfg, c) for c in clips]
        parts["w2v2"] = time.perf_counter() - t0
        log(f"cpu_baseline w2v2 {parts['w2v2']:.2f} s")
    if "cnnlstm" in stages and seqs is not None:
        from oracle import cnnlstm_oracle
        from robust_speech_analysis_framework_amd.cnnlstm import CNNLSTM
        torch.manual_seed(0)
        sdm = {k: v.numpy() for k, v in CNNLSTM().state_dict().items()}
        x = cnnlstm_oracle.collate_zero_pad(seqs)
        t0 = time.perf_counter()
        ref["logits"] = np.asarray(cnnlstm_oracle.forward_torch(sdm, x, "silu"))[0]
        parts["cnnlstm"] = time.perf_counter() - t0
    total = sum(parts.values())
    cpu_model = ""
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                cpu_model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    model_stages = [s for s in stages if s in ("w2v2", "cnnlstm")]
    return ref, {"value": round(n * seconds / total, 2), "unit": "audio-s/s", "cores": cores if model_stages else 1,
            "kind": "port",
            "sample": f"{n} x {seconds:g} s clips through stages {stages}: oracle/ (numpy float64 DSP on 1 core, "
                      f"MSHDS timed on 5 s and scaled linearly; torch-CPU float32 models on {cores} threads, "
                      f"batch-1 windows like the reference)",
            "host_cpus": os.cpu_count(), "cpu_model": cpu_model,
            "seconds_per_stage": {k: round(v, 3) for k, v in parts.items()}}


def parity_vs_cpu(pipe, ref, stages, dev):
    """BASELINE.json's second half of the metric ("feature max-abs-err vs CPU"): the HIP path on the very clip the
    cpu_baseline leg just pushed through oracle/ (the oracle is the checker here, as in tests/ and smoke()).
    rel = |gpu - cpu| / max(|cpu|, 1e-3 * max|cpu| of the stage's vector); NaN patterns must coincide."""
    import numpy as np
    import torch
    out = {}
    clip = torch.from_numpy(np.ascontiguousarray(ref["clip0"])).to(dev)[None, :]
    row = pipe.run(clip)[0].double().cpu().numpy()
    torch.cuda.synchronize()
    col = 0

    def cmp(name, got, want):
        got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
        both = ~np.isnan(got) & ~np.isnan(want)
        d = np.abs(got[both] - want[both])
        scale = np.maximum(np.abs(want[both]), 1e-3 * np.abs(want[both]).max()) if both.any() else np.ones(0)
        own = d / np.maximum(np.abs(want[both]), 1e-30)                 # per column, no floor (harsh on ~0 columns)
        out[name] = {"columns": int(got.size), "compared": int(both.sum()),
                     "nan_pattern_equal": bool(np.array_equal(np.isnan(got), np.isnan(want))),
                     "max_abs_err": float(d.max()) if d.size else None,
                     "max_rel_err": float((d / scale).max()) if d.size else None,
                     "median_rel_err_per_column": float(np.median(own)) if d.size else None,
                     "columns_within_1e-4_per_column_rel": int((own <= 1e-4).sum()) if d.size else 0}
    if "mshds" in stages:
        if "mshds" in ref:                                   # the oracle ran on a 5 s excerpt: run the HIP path on the same
            sub = torch.from_numpy(np.ascontiguousarray(ref["mshds_input"])).to(dev)
            got, _ = pipe.mshds.extract_packed(sub, [0], [int(sub.numel())])
            torch.cuda.synchronize()
            cmp("mshds_25_features_5s_excerpt", got[0].cpu().numpy(), ref["mshds"])
        col += 25
    if "smile" in stages:
        if "smile" in ref:
            cmp("opensmile_912_functionals_30s", row[col:col + 912], ref["smile"])
        col += 912
    if "cnnlstm" in stages and "logits" in ref:
        cmp("wav2vec2_to_cnnlstm_logits_30s", row[col:col + 2], ref["logits"])
    return out


def main():
    args = parse_args()
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from robust_speech_analysis_framework_amd import _lib, pipeline, synth
    from robust_speech_analysis_framework_amd.dist import gather_rows
    _lib.load()
    torch.set_num_threads(min(usable_cpus(), 8))

    stages = pipeline.resolve_stages(args.stages)
    # synthetic shard of this rank: clip index = rank*clips + i (pool-tiled), resident in HBM
    host = synth.synth_batch(args.clips, args.seconds, pool=args.pool, first=rank * args.pool)
    wav = torch.from_numpy(host).to(dev)
    pipe = pipeline.Pipeline(stages, device=dev, seconds=args.seconds,
                             w2v2_chunks_per_call=args.w2v2_chunks_per_call, overlap=args.overlap)
    audio_s_per_step = args.clips * args.seconds * world

    def step():
        rows = pipe.run(wav)                       # [clips, row_width] float32 on device
        return gather_rows(rows, args.clips * world)   # one RCCL all-gather of the result rows

    if rank == 0:
        log(f"setup done: stages {stages}, {args.clips} clips x {args.seconds:g} s per GPU, world {world}")
    for i in range(args.warmup):
        step()
        torch.cuda.synchronize()
        if rank == 0:
            log(f"warmup step {i + 1}/{args.warmup} done")
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    _lib.prof_begin()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof = _lib.prof_end()
    if rank == 0:
        log(f"timed region: {args.steps} steps in {dt:.3f} s")
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert torch.isfinite(out[:, pipe.finite_cols]).all(), "non-finite results in the timed region"
    # size-independent property at full size: the batch is a pool of distinct clips tiled to --clips, and a
    # clip's row must not depend on its position or on what else is in the batch -> duplicates bit-identical
    local = out[rank * args.clips:(rank + 1) * args.clips] if world > 1 else out
    uniq = min(args.pool, args.clips)
    dup_ok = True
    if args.clips > uniq:
        ref_rows = local[:uniq]
        reps = local[: (args.clips // uniq) * uniq].view(-1, uniq, local.shape[1])
        same = (reps == ref_rows[None]) | (torch.isnan(reps) & torch.isnan(ref_rows[None]))
        dup_ok = bool(same.all().item())
    assert dup_ok, "rows of duplicated clips differ: results depend on batch position"

    if rank == 0:
        value = audio_s_per_step * args.steps / dt
        roof = pipeline.roofline(prof, pipe, args.clips, args.seconds, args.steps,
                                 HBM_PEAK_GBS, MFMA_F32_PEAK_TFLOPS)
        # HBM bytes per launch of the dominant kernel come from separate rocprofv3 --pmc passes
        # (FETCH_SIZE x2 on gfx950 + WRITE_SIZE, MI355X_MICROARCH.md §HBM); PMC cannot run inside this process
        tpath = os.path.join(ROOT, "profiles", "r01", "pmc_bench_traffic.json")
        if roof and roof.get("kernel") == "w2v2_gemm" and os.path.exists(tpath):
            with open(tpath) as f:
                tj = json.load(f)
            roof["traffic"] = round(tj["traffic_bytes_per_launch_fetch_x2_plus_write"])
            roof["traffic_source"] = "profiles/r01/pmc_bench_traffic.json (separate --pmc FETCH_SIZE / WRITE_SIZE passes)"
        line = {
            "metric": "audio-seconds processed/sec (extract+CNN-LSTM fwd)",
            "value": round(value, 2), "unit": "audio-s/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": pipe.describe(args.clips, args.seconds),
                       "clips_per_gpu": args.clips, "clip_seconds": args.seconds,
                       "stages": [s for s in stages], "sharding": f"clips/{world} ranks, all_gather of result rows",
                       "w2v2_windows_per_call": args.w2v2_chunks_per_call},
            "roofline": roof,
            "checks": {"finite": True, "duplicate_clips_bit_identical": dup_ok,
                       "note": "parity vs the CPU oracle is asserted by tests/ (-m gpu) and smoke(); parity_vs_cpu below reports it for the cpu_baseline sample, outside the timed run"},
            "kernels": {k: {"launches": v["launches"], "ms": round(v["ms"], 3),
                            **({"tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2)} if v["flops"] > 0 and v["ms"] > 0 else {})}
                        for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])},
        }
        if world == 1 and not args.no_cpu_baseline:
            ref, line["cpu_baseline"] = cpu_baseline(stages, args.seconds, args.cpu_sample_clips)
            line["parity_vs_cpu"] = parity_vs_cpu(pipe, ref, stages, dev)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
