"""Turn two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE; they do not fit one pass on gfx950) into
profiles/<round>/pmc_bench_traffic.json, the file bench.py reads for `roofline.traffic`.

  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_fetch -o f --output-format csv -- python3 bench.py ...
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc_write -o w --output-format csv -- python3 bench.py ...
  python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01/pmc_bench_traffic.json "<command>"

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): both counters are in KB;
on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes, so wide coalesced reads are doubled; WRITE_SIZE is exact.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def collect(root, counter):
    agg = defaultdict(lambda: [0.0, 0])
    files = glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {root}")
    for fn in files:
        with open(fn, newline="") as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") != counter:
                    continue
                a = agg[row["Kernel_Name"]]
                a[0] += float(row["Counter_Value"])
                a[1] += 1
    return agg


def main():
    fetch_dir, write_dir, out, cmd = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else ""
    fe, wr = collect(fetch_dir, "FETCH_SIZE"), collect(write_dir, "WRITE_SIZE")
    per = {}
    for k in sorted(set(fe) | set(wr)):
        per[k] = {"fetch_raw_KB": fe[k][0] if k in fe else 0.0, "write_KB": wr[k][0] if k in wr else 0.0,
                  "launches": fe[k][1] if k in fe else wr[k][1]}
    gemm = [k for k in per if "gemm_f32" in k]
    gl = sum(per[k]["launches"] for k in gemm)
    gf = sum(per[k]["fetch_raw_KB"] for k in gemm)
    gw = sum(per[k]["write_KB"] for k in gemm)
    doc = {"source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over: {cmd}",
           "gemm_kernels": gemm, "gemm_launches": gl, "fetch_KB_raw": gf, "write_KB": gw,
           "traffic_bytes_per_launch_fetch_x2_plus_write": (2.0 * gf + gw) * 1024.0 / max(gl, 1),
           "per_kernel_KB": per}
    with open(out, "w") as f:
        json.dump(doc, f, indent=1)
    print(json.dumps({k: doc[k] for k in ("gemm_launches", "fetch_KB_raw", "write_KB",
                                          "traffic_bytes_per_launch_fetch_x2_plus_write")}))


if __name__ == "__main__":
    main()
