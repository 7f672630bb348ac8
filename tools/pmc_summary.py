"""Summarise rocprofv3 --pmc passes: per kernel (name substring filter) the mean of every counter per dispatch.

  python tools/pmc_summary.py OUT.json KERNEL_SUBSTRING DIR [DIR ...]     (each DIR = one rocprofv3 -d directory)
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    out, needle, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
    agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    meta = {}
    for d in dirs:
        for fn in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(fn, newline="") as f:
                for row in csv.DictReader(f):
                    k = row["Kernel_Name"]
                    if needle not in k:
                        continue
                    a = agg[k][row["Counter_Name"]]
                    a[0] += float(row["Counter_Value"])
                    a[1] += 1
                    meta[k] = {x: row.get(x) for x in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size",
                                                       "Scratch_Size", "Workgroup_Size", "Grid_Size") if x in row}
    doc = {}
    for k, cs in agg.items():
        doc[k] = {"dispatch": meta.get(k, {}),
                  "mean_per_dispatch": {c: v[0] / max(v[1], 1) for c, v in sorted(cs.items())},
                  "dispatches": {c: v[1] for c, v in sorted(cs.items())}}
    with open(out, "w") as f:
        json.dump(doc, f, indent=1)
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main()
