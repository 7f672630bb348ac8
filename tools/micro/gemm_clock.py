"""In-kernel clock and matrix-pipe occupancy of the gemm_f16x3 dispatches of a rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE
SQ_VALU_MFMA_BUSY_CYCLES run: clock = GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / dispatch duration from the kernel trace.

  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES -d DIR --output-format csv -- python3 tools/gemm3_bench.py
  python tools/micro/gemm_clock.py DIR
"""
import csv, glob, os, sys
from collections import defaultdict

root = sys.argv[1]
cc = glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)
kt = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
dur = {}
for fn in kt:
    for r in csv.DictReader(open(fn, newline="")):
        dur[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
cnt = defaultdict(dict)
meta = {}
for fn in cc:
    for r in csv.DictReader(open(fn, newline="")):
        if "gemm_f16x3" not in r["Kernel_Name"]:
            continue
        d = int(r["Dispatch_Id"])
        cnt[d][r["Counter_Name"]] = float(r["Counter_Value"])
        meta[d] = (r["Kernel_Name"].split("H3Cfg")[1][:48], int(r["Grid_Size"]))
groups = defaultdict(list)
for d, c in cnt.items():
    if d in dur and "GRBM_GUI_ACTIVE" in c:
        clk = c["GRBM_GUI_ACTIVE"] / 8.0 / dur[d]
        busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (c["GRBM_GUI_ACTIVE"] * 128.0)
        groups[meta[d]].append((dur[d] * 1e3, clk / 1e9, busy))
for k, v in sorted(groups.items()):
    n = len(v)
    print(f"{k[0]} grid {k[1]}: {n} dispatches, {sum(x[0] for x in v) / n:.3f} ms, in-kernel clock {sum(x[1] for x in v) / n:.3f} GHz, "
          f"matrix pipe busy {100 * sum(x[2] for x in v) / n:.1f} % of the active cycles")
