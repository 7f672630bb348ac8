"""Idle gaps of the GPU in a `rocprofv3 --kernel-trace --output-format csv` run: the union of all kernel intervals of the last
step, its largest holes and the kernels either side of them (how a host-side stall or a lost overlap between the two MSHDS
streams shows up when the per-family event times do not explain a step time).

  rocprofv3 --kernel-trace -d gpurun_out/kt -o kt --output-format csv -- python3 bench.py --config C2 --steps 1 --warmup 1
  python tools/trace_gaps.py gpurun_out/kt/kt_kernel_trace.csv [marker-kernel-substring]

The marker (default `clip_peak`, the first MSHDS kernel of a step) splits warm-up from the timed step.
"""
import csv
import sys


def main():
    path = sys.argv[1]
    marker = sys.argv[2] if len(sys.argv) > 2 else "clip_peak"
    with open(path, newline="") as f:
        ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(f))
    marks = [i for i, e in enumerate(ev) if marker in e[2]]
    sub = ev[marks[len(marks) // 2]:] if len(marks) >= 2 else ev
    t0, t1 = sub[0][0], max(e[1] for e in sub)
    cur, last, gaps, busy = sub[0][0], None, [], 0
    for s, e, n in sub:
        if s > cur:
            gaps.append((s - cur, (cur - t0) / 1e6, last, n))
            cur = s
        if e > cur:
            busy += e - cur
            cur, last = e, n
    print(f"{len(sub)} kernels, span {(t1 - t0) / 1e6:.1f} ms, busy {busy / 1e6:.1f} ms, idle {sum(g[0] for g in gaps) / 1e6:.1f} ms")
    for g in sorted(gaps, reverse=True)[:20]:
        print(f"{g[0] / 1e6:8.2f} ms at {g[1]:9.1f} ms  after {(g[2] or '-')[:48]:48s} before {g[3][:48]}")


if __name__ == "__main__":
    main()
