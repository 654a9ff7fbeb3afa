#!/usr/bin/env python3
"""Print the finest-level Jacobi kernel's counter values from tools/pmc_diag.sh outputs."""
import csv, glob, os, sys, collections
for tag in sys.argv[1:]:
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join("gpurun_out", "diag", tag, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if ("sdia_kernel<2" in n or "sdia_tab_kernel<2" in n or "sell_kernel<2" in n or "csr_block_kernel<2" in n or "csr_wave_kernel<2" in n) and ", 1>" in n:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(tag)
    for k in sorted(acc):
        v = acc[k]
        print(f"  {k:32s} {sum(v) / len(v):16.1f}  (n={len(v)})")
