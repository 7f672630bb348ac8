#!/usr/bin/env python3
"""Benchmark of the hot path: audio-seconds processed per second on synthetic 16 kHz mono 30 s clips.

A "step" is one pass of the hot path over one batch that is already resident in HBM.  One process per GPU; clips
shard across ranks with no data-path collective, the per-clip result rows are all-gathered once per step (RCCL) when
N > 1.  Rank 0 prints ONE JSON line.

    python bench.py                                   # e2e: MSHDS + openSMILE-style + Wav2Vec2 -> CNN-LSTM, 1 000 clips
    python bench.py --config C2|C3|C4                 # BASELINE configs 2-4 on their own (own roofline object)
    python bench.py --total-clips 10000 --gpus 8      # BASELINE config C5 (strong scaling: 1 250 clips per rank)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...                      # no torchrun around it: starts that torchrun itself (launch_ranks)
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense fp32 matrix peak
F64_VECTOR_PEAK_TFLOPS = 78.6  # MI355X_MICROARCH.md: fp64 vector (= fp64 MFMA issue) peak


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=str, default="e2e", choices=["e2e", "C2", "C3", "C4"],
                    help="BASELINE config: e2e = extract -> CNN-LSTM (C5's per-GPU work), C2 = MSHDS + openSMILE-style, "
                         "C3 = Wav2Vec2 frames, C4 = CNN-LSTM forward on randn(256, 1500, 768)")
    ap.add_argument("--clips", type=int, default=None, help="clips per GPU (weak scaling); default 1000 (C4: batch 256)")
    ap.add_argument("--total-clips", type=int, default=None,
                    help="strong scaling: a fixed total sharded over the ranks (C5: 10000)")
    ap.add_argument("--seconds", type=float, default=30.0)
    ap.add_argument("--pool", type=int, default=64, help="distinct synthetic clips; global clip g plays member g mod pool")
    ap.add_argument("--stages", type=str, default=None, help="override the config's stage list (comma separated)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-inclusive", action="store_true", help="skip the PCM -> H2D -> ... -> D2H pass after the timed region")
    ap.add_argument("--cpu-sample-clips", type=int, default=0, help="0 = auto")
    ap.add_argument("--w2v2-chunks-per-call", type=int, default=2048,
                    help="Wav2Vec2 windows per sub-batch (workspace 118 GiB of the 288 GB at 2048)")
    ap.add_argument("--overlap", action="store_true",
                    help="run the MSHDS stage on a second HIP stream beside Wav2Vec2 (per-kernel event times then include "
                         "time-sharing, so the roofline object is only clean without it)")
    ap.add_argument("--no-per-config", action="store_true",
                    help="skip the short BASELINE config C2 / C3 / C4 measurements that the e2e line carries in `per_config`")
    ap.add_argument("--launcher-selftest", action="store_true",
                    help="no GPU work and NOT a measurement: every rank fabricates its shard's rows on the CPU (gloo), the ranks "
                         "run the sharding plan, the all-gather, the barriers and the MAX-over-ranks timing of the real run; "
                         "exercises `bench.py --gpus N` -> torchrun -> rank 0's JSON line where no GPU exists (tests/)")
    return ap.parse_args()


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` outside any launcher: start the N ranks as ONE child process tree
    (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py ...`,
    the command the driver itself uses) BEFORE this process has made any GPU call, let the child own stdout / stderr
    (rank 0 prints the JSON line) and return its exit code.  Under an existing torchrun (WORLD_SIZE set) this is not called."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    log(f"--gpus {args.gpus} without a launcher: starting {' '.join(cmd)}")
    return subprocess.run(cmd, env=env).returncode


def launcher_selftest(args) -> None:
    """The control flow of main() around a fabricated CPU 'pipeline' (row of global clip g = g * 10 + column): sharding plan,
    warm-up, barrier-bracketed timed steps, all-gather of the rows in global order, MAX over ranks, rank 0's line."""
    import torch
    import torch.distributed as dist
    from robust_speech_analysis_framework_amd import benchlib
    from robust_speech_analysis_framework_amd.dist import gather_rows
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    first, n_local, n_total, scaling = benchlib.shard_plan(rank, world, args.clips if args.clips is not None else 5, args.total_clips)
    width = 940

    pool = 4                                                   # fabricated rows depend on the pool member only: g mod pool

    def step():
        idx = torch.as_tensor(benchlib.pool_members(first, n_local, pool), dtype=torch.float32)[:, None]
        rows = idx * 10.0 + torch.arange(width, dtype=torch.float32)[None, :]
        return gather_rows(benchlib.tag_rows(rows, rank), n_total)
    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    rows, gathered = benchlib.check_gathered(out, world, args.clips if args.clips is not None else 5, args.total_clips, pool)
    want = torch.as_tensor(benchlib.pool_members(0, n_total, pool), dtype=torch.float32)[:, None] * 10.0 + torch.arange(width, dtype=torch.float32)[None, :]
    assert torch.equal(rows, want), "gathered rows are not in global clip order"
    assert gathered["rows_at_their_global_position_from_their_owner_rank"] and gathered["duplicate_clips_bit_identical_across_the_gathered_table"]
    assert gathered["ranks_contributing"] == gathered["ranks_with_a_shard"], "a rank's rows did not arrive"
    if rank == 0:
        print(json.dumps({"metric": "launcher selftest: NOT a measurement (no GPU work, fabricated rows on the CPU, gloo)",
                          "value": None, "unit": None, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(1e3 * dt / max(args.steps, 1), 3), "scaling": scaling, "data": "none",
                          "config": {"workload": "launcher selftest", "clips_total": n_total, "clips_per_rank": n_local},
                          "checks": {"gathered": gathered}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


# ---- CPU baseline leg (the only place outside tests/ and smoke() that touches oracle/) -------------------------------
def _dsp_worker(args):
    """One clip through one numpy oracle in its own process (1 thread): (stage, clip id, seconds, result)."""
    stage, k, seconds = args
    from robust_speech_analysis_framework_amd import synth
    x = synth.synth_clip(k, seconds)
    t0 = time.perf_counter()
    if stage == "smile":
        from oracle import smile_oracle
        res = smile_oracle.extract(x)
    else:
        from oracle import mshds_oracle
        res, _ = mshds_oracle.extract(x)
    return stage, k, time.perf_counter() - t0, res


def _dsp_pool(stage, clip_ids, seconds, workers):
    """The single-threaded numpy restatement on ``workers`` cores, one process per clip."""
    import concurrent.futures as cf
    import multiprocessing as mp
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    t0 = time.perf_counter()
    with cf.ProcessPoolExecutor(max_workers=workers, mp_context=mp.get_context("spawn")) as ex:
        out = list(ex.map(_dsp_worker, [(stage, k, seconds) for k in clip_ids]))
    return time.perf_counter() - t0, [o[2] for o in out], {o[1]: o[3] for o in out}


def cpu_baseline(config, stages, seconds, sample_clips):
    """The CPU restatement (kind "port": oracle/) on a bounded sample of the same synthetic workload, on the host cores of
    the GPU box: all usable cores AND one thread.  DSP oracles are single-threaded numpy: one process per
    clip, as many processes as clips (<= cores); model stages are torch-CPU float32 (the ops the reference's CPU path
    dispatches): 1 warm-up + median of 5."""
    import numpy as np
    import torch
    from robust_speech_analysis_framework_amd import benchlib, synth
    cores = benchlib.usable_cpus()
    n = sample_clips or min(4, cores)
    ids = [900000 + k for k in range(n)]
    ref = {"clip_id": ids[0]}
    all_cores, one_thread, notes = {}, {}, []
    if "smile" in stages:
        wall, per, res = _dsp_pool("smile", ids, seconds, min(n, cores))
        all_cores["smile"] = wall / n
        one_thread["smile"] = float(np.median(per))
        ref["smile"] = res[ids[0]]
        log(f"cpu_baseline smile: {n} clips on {min(n, cores)} processes {wall:.1f} s wall, {np.median(per):.2f} s per clip on one core")
    if "mshds" in stages:
        wall, per, res = _dsp_pool("mshds", ids, seconds, min(n, cores))     # full 30 s clips, not an excerpt
        all_cores["mshds"] = wall / n
        one_thread["mshds"] = float(np.median(per))
        ref["mshds"] = res[ids[0]]
        log(f"cpu_baseline mshds: {n} clips on {min(n, cores)} processes {wall:.1f} s wall, {np.median(per):.1f} s per clip on one core")
        notes.append(f"DSP oracles: {n} full {seconds:g} s clips, one process (1 thread) per clip")
    clips = [synth.synth_clip(k, seconds) for k in ids]
    ref["clip0"] = clips[0]
    seqs = None
    if "w2v2" in stages:
        from oracle import w2v2_oracle
        from robust_speech_analysis_framework_amd.w2v2_config import W2V2Config, random_state_dict
        cfg = W2V2Config()
        sd = random_state_dict(cfg, 0)
        torch.set_num_threads(cores)
        it = iter(range(10 ** 9))
        seqs = [w2v2_oracle.extract_sequence(sd, cfg, c) for c in clips[:1]]           # also the warm-up
        med, _ = benchlib.median_time(lambda: w2v2_oracle.extract_sequence(sd, cfg, clips[next(it) % n]), 0, 5)
        all_cores["w2v2"] = med
        torch.set_num_threads(1)
        w5 = clips[0][:80000]                                                             # one 5 s window, batch 1 like the reference
        w2v2_oracle.extract_sequence(sd, cfg, w5)
        m1, _ = benchlib.median_time(lambda: w2v2_oracle.extract_sequence(sd, cfg, w5), 0, 3)
        # a 30 s clip = 7 full windows + one 2 s tail: 529.64 / 71.66 full-window equivalents of work
        one_thread["w2v2"] = m1 * (529.64 / 71.66) * (seconds / 30.0)
        notes.append("Wav2Vec2 on 1 thread: median of 3 runs of ONE 5 s window, scaled by the clip's FLOP ratio 529.64 / 71.66")
        torch.set_num_threads(cores)
        log(f"cpu_baseline w2v2: {med:.2f} s per clip on {cores} threads, {one_thread['w2v2']:.1f} s on 1 thread (scaled)")
    if "cnnlstm" in stages or "cnnlstm_only" in stages:
        from oracle import cnnlstm_oracle
        from robust_speech_analysis_framework_amd.cnnlstm import CNNLSTM
        torch.manual_seed(0)
        sdm = {k: v.numpy() for k, v in CNNLSTM().state_dict().items()}
        if seqs is not None:
            x = cnnlstm_oracle.collate_zero_pad(seqs)
            per_clip = 1.0
        else:                                                                             # C4: rows of the seed-1234 batch
            xs = torch.randn(256, 1500, 768, generator=torch.Generator().manual_seed(1234))[:4].numpy()
            x, per_clip = xs, 4.0
            ref["c4_rows"] = x
        torch.set_num_threads(cores)
        ref["logits"] = np.asarray(cnnlstm_oracle.forward_torch(sdm, x, "silu"))
        med, _ = benchlib.median_time(lambda: cnnlstm_oracle.forward_torch(sdm, x, "silu"), 1, 5)
        all_cores["cnnlstm"] = med / per_clip
        torch.set_num_threads(1)
        m1, _ = benchlib.median_time(lambda: cnnlstm_oracle.forward_torch(sdm, x, "silu"), 1, 5)
        one_thread["cnnlstm"] = m1 / per_clip
        torch.set_num_threads(cores)
    tot_all, tot_one = sum(all_cores.values()), sum(one_thread.values())
    return ref, {"value": round(seconds / tot_all, 3), "unit": "audio-s/s", "cores": cores, "kind": "port",
                 "one_thread": {"value": round(seconds / tot_one, 4), "cores": 1,
                                "seconds_per_clip_per_stage": {k: round(v, 3) for k, v in one_thread.items()}},
                 "seconds_per_clip_per_stage": {k: round(v, 3) for k, v in all_cores.items()},
                 "sample": ("4 rows of the randn(256, 1500, 768) seed-1234 batch through oracle/cnnlstm_oracle (torch-CPU float32, "
                            f"{cores} threads and 1 thread; 1 warm-up + median of 5)") if config == "C4" else
                           f"{n} x {seconds:g} s synthetic clips through {stages} of oracle/ (numpy float64 DSP restatements: one "
                           f"single-threaded process per clip, {min(n, cores)} at a time; torch-CPU float32 models on {cores} threads, "
                           f"batch-1 windows like the reference; 1 warm-up + median of 5).  " + "  ".join(notes),
                 "host_cpus": os.cpu_count(), "usable_cpus": cores, "cpu_model": benchlib.cpu_model(),
                 "note": "reported baseline, not the target: the restatement is a numpy port, not Praat / openSMILE C++"}


def parity_vs_oracle(pipe, ref, stages, dev, model=None):
    """BASELINE.json's second half of the metric ("feature max-abs-err vs CPU"): the HIP path on the first clip the
    cpu_baseline leg pushed through oracle/ (the oracle is the checker here, as in tests/ and smoke()).  The MSHDS and
    openSMILE oracles are this repository's own restatements (parity UNPINNED: no Praat / SMILExtract output exists);
    the CNN-LSTM oracle is pinned by vectors captured from the reference module."""
    import numpy as np
    import torch
    out = {}

    def cmp(name, got, want, pinned):
        got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
        both = ~np.isnan(got) & ~np.isnan(want)
        d = np.abs(got[both] - want[both])
        scale = np.maximum(np.abs(want[both]), 1e-3 * np.abs(want[both]).max()) if both.any() else np.ones(0)
        own = d / np.maximum(np.abs(want[both]), 1e-30)
        out[name] = {"oracle_pinned_by": pinned, "columns": int(got.size), "compared": int(both.sum()),
                     "nan_pattern_equal": bool(np.array_equal(np.isnan(got), np.isnan(want))),
                     "max_abs_err": float(d.max()) if d.size else None,
                     "max_rel_err": float((d / scale).max()) if d.size else None,
                     "median_rel_err_per_column": float(np.median(own)) if d.size else None,
                     "columns_within_1e-4_per_column_rel": int((own <= 1e-4).sum()) if d.size else 0}
    if "cnnlstm_only" in stages:
        got = model(torch.from_numpy(ref["c4_rows"]).to(dev))
        torch.cuda.synchronize()
        cmp("cnnlstm_logits_4_rows_of_the_seed_1234_batch", got.cpu().numpy(), ref["logits"], "reference module (tests/golden)")
        return out
    clip = torch.from_numpy(np.ascontiguousarray(ref["clip0"])).to(dev)[None, :]
    row = pipe.run(clip)[0].double().cpu().numpy()
    torch.cuda.synchronize()
    col = 0
    if "mshds" in stages:
        if "mshds" in ref:
            cmp("mshds_25_features_30s", row[col:col + 25], ref["mshds"], "none (own restatement of Praat)")
        col += 25
    if "smile" in stages:
        if "smile" in ref:
            names = __import__("oracle.smile_oracle", fromlist=["x"]).feature_names()
            pitch = np.array([n.split("_sma")[0] in ("F0final", "voicingFinalUnclipped", "jitterLocal", "jitterDDP",
                                                     "shimmerLocal", "logHNR") for n in names])
            g, r = row[col:col + 912], np.asarray(ref["smile"])
            cmp("opensmile_768_functionals_of_the_32_frame_local_LLDs_30s", g[~pitch], r[~pitch], "none (own restatement of openSMILE)")
            cmp("opensmile_144_functionals_of_the_pitch_chain_LLDs_30s", g[pitch], r[pitch],
                "none (own restatement; float64 on both sides since round 3: tests/ assert all 912 columns end to end, positions exact)")
        col += 912
    if "cnnlstm" in stages and "logits" in ref:
        cmp("wav2vec2_to_cnnlstm_logits_30s", row[col:col + 2], ref["logits"][0], "reference module (CNN-LSTM) / transformers (Wav2Vec2 arithmetic)")
    return out


def kernel_sha():
    """Identity of the GEMM / Wav2Vec2 kernel sources: a stored PMC traffic figure is attached only to the code it measured."""
    h = hashlib.sha256()
    for f in ("gemm_f16x3.hip", "gemm_f16x3.h", "w2v2.hip"):
        with open(os.path.join(ROOT, "robust_speech_analysis_framework_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def attach_traffic(roof, args, n_local):
    """roofline.traffic = HBM bytes per launch from separate rocprofv3 --pmc passes (FETCH_SIZE x2 on gfx950 + WRITE_SIZE,
    MI355X_MICROARCH.md §HBM); PMC cannot run inside this process, so the figure comes from a committed profile and is
    attached only when that profile was taken with this run's shape and kernel sources."""
    if not roof:
        return
    for rnd in ("r04", "r03", "r02", "r01"):
        path = os.path.join(ROOT, "profiles", rnd, "pmc_bench_traffic.json")
        if not os.path.exists(path):
            continue
        with open(path) as f:
            tj = json.load(f)
        want = tj.get("run", {})
        if (roof.get("kernel") == want.get("kernel") and want.get("clips") == n_local and want.get("config") == args.config
                and want.get("w2v2_windows_per_call") == args.w2v2_chunks_per_call and want.get("kernel_sha") == kernel_sha()):
            roof["traffic"] = round(tj["traffic_bytes_per_launch_fetch_x2_plus_write"])
            roof["traffic_source"] = f"profiles/{rnd}/pmc_bench_traffic.json ({tj.get('source', '')})"
            return
    roof["traffic_note"] = "no committed PMC profile matches this run's shape and kernel sources"


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        sys.exit(launch_ranks(args))                         # nothing has touched the GPU (or imported torch) yet
    if args.launcher_selftest:
        return launcher_selftest(args)
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    force_collective = os.environ.get("RSAF_FORCE_COLLECTIVE", "0") == "1"
    if world > 1 or force_collective:
        # RSAF_FORCE_COLLECTIVE=1 at N = 1: a one-rank RCCL group, so that the terminal all-gather really executes
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        else:
            dist.init_process_group("nccl", device_id=dev)

    from robust_speech_analysis_framework_amd import _lib, benchlib, pipeline, synth
    from robust_speech_analysis_framework_amd.dist import gather_rows
    _lib.load()
    torch.set_num_threads(min(benchlib.usable_cpus(), 8))

    stages = pipeline.resolve_stages(args.stages) if args.stages else list(benchlib.CONFIGS[args.config])
    c4 = stages == ["cnnlstm_only"]
    if c4:
        args.pool = max(args.pool, 256)                      # the config's input is the whole randn(256, 1500, 768) batch
    clips_per_gpu = args.clips if args.clips is not None else (256 if c4 else 1000)
    first, n_local, n_total, scaling = benchlib.shard_plan(rank, world, clips_per_gpu, args.total_clips)
    members = benchlib.pool_members(first, n_local, args.pool)
    pipe = model = None
    if c4:
        from robust_speech_analysis_framework_amd.cnnlstm import CNNLSTM
        torch.manual_seed(0)
        model = CNNLSTM().to(dev).eval()                     # reference defaults C = H = 128, silu, seed 0
        xh = torch.randn(256, 1500, 768, generator=torch.Generator().manual_seed(1234))
        x = xh[torch.as_tensor([m % 256 for m in members])].to(dev) if members != list(range(256)) else xh.to(dev)
        frames_seconds = 30.0                                # 1 500 frames ~ 30 s of audio per sequence

        def run_local():
            return model(x)
        audio_s_per_step = n_total * frames_seconds
    else:
        uniq = sorted(set(members))
        base = {m: synth.synth_clip(m, args.seconds) for m in uniq}
        host = np.stack([base[m] for m in members]) if n_local else np.zeros((0, int(args.seconds * 16000)), np.float32)
        wav = torch.from_numpy(host).to(dev)
        pipe = pipeline.Pipeline(stages, device=dev, seconds=args.seconds,
                                 w2v2_chunks_per_call=args.w2v2_chunks_per_call, overlap=args.overlap)

        def run_local():
            return pipe.run(wav)
        audio_s_per_step = n_total * args.seconds

    def step():
        # one RCCL all-gather of the result rows; the producing rank rides in a last column (benchlib.tag_rows)
        return gather_rows(benchlib.tag_rows(run_local(), rank), n_total)

    if rank == 0:
        log(f"setup done: config {args.config}, stages {stages}, {n_local} clips on this rank of {n_total} ({scaling}), "
            f"{len(set(members))} distinct, world {world}")
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
    out, gathered = benchlib.check_gathered(out, world, clips_per_gpu, args.total_clips, args.pool)
    assert gathered["rows_at_their_global_position_from_their_owner_rank"], "gathered rows are not at their global positions"
    assert gathered["ranks_contributing"] == gathered["ranks_with_a_shard"], "a rank's rows did not arrive"
    assert gathered["duplicate_clips_bit_identical_across_the_gathered_table"], "rows of one clip differ between ranks / batch positions"
    local = out[first:first + n_local] if world > 1 else out
    m_cols = 25 if (not c4 and "mshds" in stages) else 0    # MSHDS cells may be NaN by contract (failed helper -> NaN)
    assert torch.isfinite(local[:, m_cols:]).all(), "non-finite results in the timed region"
    assert m_cols == 0 or (torch.isfinite(local[:, :m_cols]).sum(dim=1) >= 20).all(), "MSHDS rows mostly NaN"
    dup_ok = benchlib.duplicates_bit_identical(local, members)
    assert dup_ok, "rows of duplicated clips differ: results depend on batch position"

    # inclusive pass (outside the timed region): 16-bit PCM in pinned host memory -> H2D -> decode on the device -> hot
    # path -> D2H of the result rows; reported beside the resident-in-HBM number, never as `value`
    inclusive = None
    if rank == 0 and not c4 and not args.no_inclusive and n_local:
        import ctypes as C
        pcm = torch.from_numpy(np.round(host * 32768.0).astype(np.int16)).pin_memory()
        lib = _lib.load()
        times = []
        for _ in range(2):
            torch.cuda.synchronize()
            ti = time.perf_counter()
            d_pcm = pcm.to(dev, non_blocking=True)
            w2 = torch.empty(pcm.shape, dtype=torch.float32, device=dev)
            _lib.check(lib.rsaf_pcm_to_mono_f32(_lib.ptr(d_pcm), 2, 1, C.c_int64(pcm.numel()), _lib.ptr(w2), _lib.stream_ptr(None)),
                       "rsaf_pcm_to_mono_f32")
            rows = pipe.run(w2).cpu()
            torch.cuda.synchronize()
            times.append(time.perf_counter() - ti)
        assert torch.equal(torch.nan_to_num(rows), torch.nan_to_num(local.cpu())), "decoded PCM path differs from the resident path"
        inclusive = {"value": round(n_local * args.seconds / min(times), 2), "unit": "audio-s/s", "ms_per_step": round(1e3 * min(times), 3),
                     "h2d_bytes": int(pcm.numel() * 2), "d2h_bytes": int(rows.numel() * 4), "ranks": 1,
                     "what": "rank 0's shard: pinned int16 PCM -> H2D -> rsaf_pcm_to_mono_f32 -> hot path -> D2H rows (best of 2)"}

    # MSHDS runs its analyses on three HIP streams in the product (mshds.py), so a family's event bracket in the timed region
    # contains the time its kernels waited for the others: those brackets are not kernel times.  One more pass of the
    # DSP stages, AFTER the timed region, with every analysis on ONE stream and the event brackets on: the MSHDS / smile /
    # resampling families of `other_rooflines` and the C2-level entry come from this pass, the Wav2Vec2 / CNN-LSTM families
    # (one stream anyway) from the timed region itself.
    prof_one, one_wall = None, None
    dsp = tuple(st for st in ("mshds", "smile") if st in stages)
    if rank == 0 and world == 1 and pipe is not None and dsp and n_local:
        keep_streams = pipe.mshds.n_streams if pipe.mshds is not None else None
        if pipe.mshds is not None:
            pipe.mshds.n_streams = 1
        try:
            pipe.run(wav, only=dsp)
            torch.cuda.synchronize()
            _lib.prof_begin()
            t1 = time.perf_counter()
            pipe.run(wav, only=dsp)
            torch.cuda.synchronize()
            one_wall = time.perf_counter() - t1
            prof_one = _lib.prof_end()
        finally:
            if pipe.mshds is not None:
                pipe.mshds.n_streams = keep_streams
        log(f"one-stream pass of {dsp}: {1e3 * one_wall:.1f} ms, {sum(v['ms'] for v in prof_one.values()):.1f} ms of family brackets")

    # BASELINE configs C2 / C3 / C4 on their own, same process, after the timed region (N = 1, e2e only): the same code
    # paths as `--config C2|C3|C4`, 1 warm-up + 2 timed steps each, so that the per-config table is driver-visible
    per_config = None
    if rank == 0 and world == 1 and args.config == "e2e" and not args.stages and not args.no_per_config and n_local:
        per_config = {}

        def timed(fn, audio_s):
            fn()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            el = (time.perf_counter() - t1) / 2
            return {"value": round(audio_s / el, 2), "unit": "audio-s/s", "ms_per_step": round(1e3 * el, 3), "steps": 2, "warmup": 1}
        per_config["C2"] = {**timed(lambda: pipe.run(wav, only=("mshds", "smile")), n_local * args.seconds),
                            "workload": f"MSHDS 25/25 + openSMILE-style 38 LLD / 912 functionals on {n_local} x {args.seconds:g} s clips"}
        per_config["C3"] = {**timed(lambda: pipe.run(wav, only=("w2v2",)), n_local * args.seconds),
                            "workload": f"Wav2Vec2-base frame embeddings on {n_local} x {args.seconds:g} s clips"}
        x4 = torch.randn(256, 1500, 768, generator=torch.Generator().manual_seed(1234)).to(dev)
        per_config["C4"] = {**timed(lambda: pipe.model(x4), 256 * 30.0),
                            "workload": "CNN-LSTM-attn forward on randn(256, 1500, 768) seed 1234 (7 680 audio-s per batch)"}
        del x4
        # C3 on a corpus-shaped batch: clip lengths U[5, 60] s (seeded), about the same audio as the equal-length C3, so every
        # clip ends in a tail window of its own length (the reference's loop, src/foundation_model_extractor.py:103-108); all
        # windows of all clips go through rsaf_w2v2_forward_ragged in balanced sub-batches
        if pipe.w2v2 is not None:
            rng = np.random.default_rng(20261005)
            secs_r = []
            while sum(secs_r) < n_local * args.seconds:
                secs_r.append(float(rng.uniform(5.0, 60.0)))
            long_pool = {m: synth.synth_clip(m, 60.0) for m in range(min(args.pool, 16))}
            lens_r = [int(round(sr_ * 16000)) for sr_ in secs_r]
            offs_r = np.concatenate([[0], np.cumsum(lens_r)])
            wav_r = torch.from_numpy(np.concatenate([long_pool[k % len(long_pool)][:ln] for k, ln in enumerate(lens_r)])).to(dev)
            audio_r = float(offs_r[-1]) / 16000.0
            n_win = sum(len(pl) for pl in pipe.w2v2.plan(lens_r)[0])
            tails = len({pl[-1][1] for pl in pipe.w2v2.plan(lens_r)[0]})
            per_config["C3_ragged"] = {**timed(lambda: pipe.w2v2.extract_packed(wav_r, offs_r[:-1], lens_r), audio_r),
                                       "workload": f"Wav2Vec2-base frame embeddings on {len(lens_r)} clips of U[5, 60] s (seed 20261005, "
                                                   f"{audio_r:.0f} audio-s, {n_win} windows, {tails} distinct tail-window lengths), all windows "
                                                   "through rsaf_w2v2_forward_ragged"}
            per_config["C3_ragged"]["relative_to_equal_length_C3"] = round(per_config["C3_ragged"]["value"] / per_config["C3"]["value"], 4)
            del wav_r
        # The drop-in path a notebook gets: `src.*` on 16-bit WAV files with the default batch sizes (32 files per MSHDS batch,
        # 64 files / 256 windows per Wav2Vec2 sub-batch), file read + decode + H2D + extraction + D2H of the sequences included
        if pipe.w2v2 is not None and pipe.mshds is not None:
            import tempfile
            import wave
            import pandas as pd
            n_files = min(64, n_local)
            with tempfile.TemporaryDirectory(prefix="rsaf_dropin_") as td:
                paths = []
                for k in range(n_files):
                    pth = os.path.join(td, f"clip_{k:03d}.wav")
                    with wave.open(pth, "wb") as wf:
                        wf.setnchannels(1); wf.setsampwidth(2); wf.setframerate(16000)
                        wf.writeframes(np.round(base[uniq[k % len(uniq)]] * 32767.0).astype(np.int16).tobytes())
                    paths.append(pth)
                df = pd.DataFrame({"filepath": paths})
                os.environ.setdefault("RSAF_W2V2_RANDOM_SEED", "0")
                from src.foundation_model_extractor import extract_wav2vec2_sequences
                from src.mshds_extractor import extract_mshds_features

                def dropin():
                    f = extract_mshds_features(df, verbose=False)
                    q = extract_wav2vec2_sequences(df, model_name="seeded-random-base", verbose=False)
                    assert len(f) == n_files and len(q) == n_files
                per_config["dropin_mshds_w2v2"] = {
                    **timed(dropin, n_files * args.seconds),
                    "workload": f"src.mshds_extractor.extract_mshds_features + src.foundation_model_extractor.extract_wav2vec2_sequences on "
                                f"{n_files} x {args.seconds:g} s 16-bit WAV files (default batch sizes; file read, decode, H2D, D2H of the "
                                f"{n_files} x 1842 x 768 float32 sequences included)"}
        per_config["note"] = "measured after the timed region in the same process (resident inputs, no CPU baseline); the headline value is the e2e line"
        log("per-config lines: " + ", ".join(f"{k} {v['value']}" for k, v in per_config.items() if k != "note"))

    if rank == 0:
        value = audio_s_per_step * args.steps / dt
        frames = {"mshds_pitch_path": 6000.0 * args.seconds / 30.0, "smile_viterbi": 2998.0 * args.seconds / 30.0}
        main_prof = {k: v for k, v in prof.items() if prof_one is None or k not in prof_one}
        roofs = pipeline.rooflines(main_prof, stages, n_local, args.seconds, args.steps, HBM_PEAK_GBS, MFMA_F32_PEAK_TFLOPS,
                                   F64_VECTOR_PEAK_TFLOPS, wall_ms=1e3 * dt, frames_per_launch=frames)
        c2_level = None
        if prof_one is not None:
            one = pipeline.rooflines(prof_one, stages, n_local, args.seconds, 1, HBM_PEAK_GBS, MFMA_F32_PEAK_TFLOPS,
                                     F64_VECTOR_PEAK_TFLOPS, wall_ms=1e3 * one_wall, frames_per_launch=frames)
            for r in one:
                r["timed_in"] = "one-stream pass after the timed region (share_of_step_wall = share of that pass)"
            roofs += one
            c2_level = pipeline.c2_level_roofline(prof_one, n_local, args.seconds, one_wall, HBM_PEAK_GBS, F64_VECTOR_PEAK_TFLOPS)
        roofs = [r for r in roofs if r.get("bound") != "latency"] + [r for r in roofs if r.get("bound") == "latency"]
        roof = roofs[0] if roofs else None
        for r in roofs:                                      # the recurrence is latency-bound: report the time per step
            if r["kernel"] == "lstm_recurrent":
                tp = 750 if c4 else (pipe.last_frames // 2 if pipe is not None and pipe.last_frames else None)
                if tp:
                    r["us_per_step"] = round(1e3 * r["avg_launch_ms"] / tp, 3)
                    r["steps_per_launch"] = tp
        attach_traffic(roof, args, n_local)
        workload = (f"CNN-LSTM-attn forward (C=H=128, default init seed 0) on randn({n_local}, 1500, 768) seed 1234 per GPU"
                    if c4 else pipe.describe(n_local, args.seconds))
        line = {
            "metric": "audio-seconds processed/sec (extract+CNN-LSTM fwd)",
            "value": round(value, 2), "unit": "audio-s/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "dtype_note": "fp32 results throughout; the dense Wav2Vec2 layers obtain them from two-way fp16 splits of both "
                          "operands under power-of-two row scales (3 fp16 MFMA products per term, fp32 accumulation): error against "
                          "float64 at or below that of an fp32 FMA chain (tests/test_gemm_gpu.py); MSHDS in f64, CNN-LSTM and the "
                          "remaining GEMMs on the fp32 MFMA",
            "config": {"workload": workload, "baseline_config": args.config, "clips_per_gpu": n_local, "clips_total": n_total,
                       "distinct_clips": len(set(members)), "clip_seconds": args.seconds, "stages": list(stages),
                       "sharding": f"clips/{world} ranks, all_gather of result rows",
                       "w2v2_windows_per_call": args.w2v2_chunks_per_call},
            "roofline": roof, "other_rooflines": roofs[1:], "c2_level_roofline": c2_level, "per_config": per_config,
            "inclusive_of_pcie_and_decode": inclusive,
            "checks": {"finite": True, "duplicate_clips_bit_identical": dup_ok, "gathered": gathered,
                       "rccl_all_gather_executed": bool(world > 1 or force_collective),
                       "note": "parity vs the CPU oracle is asserted by tests/ (-m gpu) and smoke(); parity_vs_oracle below reports it "
                               "for the cpu_baseline sample, outside the timed run"},
            "kernel_time_note": "per-family HIP-event brackets on the launch stream.  `kernels` = the timed region: Wav2Vec2 / "
                                "CNN-LSTM families run on one stream (bracket = kernel time); the MSHDS analyses run on three "
                                "streams beside each other there, so THOSE brackets include waiting and are flagged multi_stream; "
                                "`kernels_one_stream_pass` = the DSP stages once more on ONE stream after the timed region "
                                "(bracket = kernel time), which is what other_rooflines and c2_level_roofline use",
            "kernels": {k: {"launches": v["launches"], "ms": round(v["ms"], 3), "share_of_step_wall": round(v["ms"] / (1e3 * dt), 4),
                            **({"multi_stream": True} if prof_one is not None and k in prof_one and k.startswith(("mshds", "praat")) else {}),
                            **({"tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2)} if v["flops"] > 0 and v["ms"] > 0 else {})}
                        for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])},
            "kernels_one_stream_pass": None if prof_one is None else {
                "wall_ms": round(1e3 * one_wall, 3), "sum_of_family_ms": round(sum(v["ms"] for v in prof_one.values()), 3),
                "families": {k: {"launches": v["launches"], "ms": round(v["ms"], 3)}
                             for k, v in sorted(prof_one.items(), key=lambda kv: -kv[1]["ms"])}},
            "family_time_sums": {"timed_region_wall_ms": round(1e3 * dt, 3),
                                 "single_stream_families_ms": round(sum(v["ms"] for k, v in prof.items()
                                                                        if not k.startswith(("mshds", "praat"))), 3)},
        }
        if world == 1 and not args.no_cpu_baseline:
            ref, line["cpu_baseline"] = cpu_baseline(args.config, stages, args.seconds, args.cpu_sample_clips)
            line["parity_vs_oracle"] = parity_vs_oracle(pipe, ref, stages, dev, model)
        print(json.dumps(line), flush=True)
    if world > 1 or force_collective:
        dist.barrier()                                      # rank 0's post-run passes are done: every rank leaves together
        torch.cuda.synchronize()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
