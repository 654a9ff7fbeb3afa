#!/usr/bin/env python3
"""What the multi-rank tuner would choose for the 216^3 hierarchy (host-only; no device, no transport): Engine::decide_comm_schedule fed
with assumed transport numbers instead of measured ones -- a latency sweep at xGMI-like bandwidth (neighbour link ~ 50 GB/s effective = 20 us/MB,
all-gather over 7 links ~ 5 us/MB) and the device numbers of this repo's profiles (launch floor 3.3 us, 5.3 TB/s = 0.19 us/MB).
Usage: python tools/comm_schedule_whatif.py > profiles/r03_comm_schedule_whatif_216.json"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparsh_amg_amd as sa
from sparsh_amg_amd import problems

grid = int(sys.argv[1]) if len(sys.argv) > 1 else 216
rp, ci, v = problems.poisson3d(grid)
A = sa.sp_matrix_mg(rp, ci, v).setup(sa.default_params(print_setup=0, print_solve=0), host_only=True)
out = {"_how": __doc__.split("Usage")[0].strip(), "levels": [A.level_info(l)["nrow"] for l in range(A.nlevels)], "cases": []}
for G in (2, 4, 8):
    for lat in (3.0, 10.0, 25.0, 60.0):
        t = A.plan_comm_schedule(G, exchange_us=lat, exchange_us_per_MB=20.0, allreduce_us=lat, allgather_us=1.5 * lat, allgather_us_per_MB=5.0,
                                 sweep_floor_us=3.3, sweep_us_per_MB=0.19)
        nu = 7
        part = [c for c in t if c["partitioned"]]
        total = sum((c["model_us_deep_halo"] if c["deep_halo"] else c["model_us_exchange_per_sweep"]) if c["partitioned"] else c["model_us_replicated"] for c in t[:-1])
        one = sum(c["model_us_replicated"] for c in t[:-1])
        out["cases"].append({"ranks": G, "exchange_latency_us": lat, "partitioned_levels": len(part), "deep_halo": bool(part and part[0]["deep_halo"]),
                             "modelled_vcycle_us_without_coarsest_solve": round(total, 1), "one_gpu_model_us": round(one, 1),
                             "modelled_speedup_of_the_smoothing_part": round(one / total, 2), "table": t})
print(json.dumps(out, indent=1))
