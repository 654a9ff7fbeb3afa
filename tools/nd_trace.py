#!/usr/bin/env python3
"""Per-launch picture of one nested-dissection coarse solve (run ON the GPU box under the kernel tracer):
  cd /tmp && rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 tools/nd_trace.py --run [--case 100|2d|fem]
  python3 tools/nd_trace.py --summarize OUT > profiles/r03_nd_solve_trace_<case>.txt
--run sets the hierarchy up with the default (reference) level policy and issues 60 coarse solves back to back; --summarize
takes the LAST solve's launches of nd_gdot_kernel from the trace: duration of each launch and the gap to the previous one."""
import argparse
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(case):
    import sparsh_amg_amd as sa
    from sparsh_amg_amd import problems

    rp, ci, v = {"100": lambda: problems.poisson3d(100), "2d": lambda: problems.poisson2d(1000), "fem": lambda: problems.fem_unstructured()}[case]()
    A = sa.sp_matrix_mg(rp, ci, v)
    if os.environ.get("SPARSH_COARSE_FORM"):
        f = os.environ["SPARSH_COARSE_FORM"].split(",")
        A.set_coarse_form(f[0], int(f[1]) if len(f) > 1 else 0, int(f[2]) if len(f) > 2 else -1, int(f[3]) if len(f) > 3 else -1)
    A.setup(sa.default_params(print_setup=0, print_solve=0))
    info = A.coarse_info()
    t = A.bench_op("coarse", A.nlevels - 1, 57)  # 3 warm-up + 57
    info["coarse_solve_us_hip_events"] = round(t * 1e6, 2)
    with open("/tmp/nd_trace_info.json", "w") as f:
        json.dump(info, f)


def summarize(d):
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "nd_gdot_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    info = json.load(open("/tmp/nd_trace_info.json")) if os.path.exists("/tmp/nd_trace_info.json") else {}
    per = info.get("nd_launches_per_solve") or 1
    last = rows[-per:]
    prev_end = int(rows[-per - 1]["End_Timestamp"]) if len(rows) > per else None
    print(f"# coarsest level: {json.dumps(info)}")
    print(f"# last of {len(rows) // per} solves: {per} launches of nd_gdot_kernel (forward levels 1..L-1, then backward L-1..0)")
    print("# launch  pass      grid   duration_us  gap_before_us")
    t0 = int(last[0]["Start_Timestamp"])
    tot = 0.0
    gkey = next((k for k in ("Grid_Size", "Grid_Size_X", "Grid_X") if last and k in last[0]), None)
    wkey = next((k for k in ("Workgroup_Size", "Workgroup_Size_X", "Workgroup_X") if last and k in last[0]), None)
    for i, r in enumerate(last):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        grid = int(r[gkey]) // max(1, int(r[wkey]) if wkey else 256) if gkey else -1
        gap = (s - prev_end) / 1e3 if prev_end is not None else float("nan")
        prev_end = e
        tot += (e - s) / 1e3
        print(f"{i:6d}  {'forward ' if 'true' in r['Kernel_Name'] else 'backward'}  {grid:6d}  {(e - s) / 1e3:10.2f}  {gap:10.2f}")
    print(f"# first start -> last end: {(prev_end - t0) / 1e3:.2f} us; sum of kernel durations {tot:.2f} us")


def summarize_pmc(fetch_dir, write_dir):
    """HBM bytes of one coarse solve from two counter passes of --run (FETCH_SIZE, WRITE_SIZE): sum over the nd_gdot_kernel dispatches of the
    last 20 solves.  FETCH_SIZE x 1.9997 (8-byte per-lane loads on gfx950: calibrated in tools/pmc_traffic.py), WRITE_SIZE exact; KiB -> bytes."""
    info = json.load(open("/tmp/nd_trace_info.json")) if os.path.exists("/tmp/nd_trace_info.json") else {}
    per = info.get("nd_launches_per_solve") or 1
    out = {}
    for name, d, corr in (("read", fetch_dir, 1.9997), ("write", write_dir, 1.0)):
        f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
        rows = [r for r in csv.DictReader(open(f)) if "nd_gdot_kernel" in r["Kernel_Name"]]
        rows.sort(key=lambda r: int(r["Dispatch_Id"]))
        last = rows[-20 * per:]
        out[name] = sum(float(r["Counter_Value"]) for r in last) * 1024.0 * corr / 20.0
    total = out["read"] + out["write"]
    print(json.dumps({"coarsest_level": info, "hbm_read_bytes_per_solve": out["read"], "hbm_write_bytes_per_solve": out["write"],
                      "hbm_bytes_per_solve": total, "factor_and_index_bytes": info.get("bytes"),
                      "traffic_over_bytes": total / info["bytes"] if info.get("bytes") else None,
                      "GBps_at_back_to_back_rate": total / (info["coarse_solve_us_hip_events"] * 1e-6) / 1e9 if info.get("coarse_solve_us_hip_events") else None}, indent=1))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--run", action="store_true")
    ap.add_argument("--case", default="100")
    ap.add_argument("--summarize")
    ap.add_argument("--summarize-pmc", nargs=2, metavar=("FETCH_DIR", "WRITE_DIR"))
    a = ap.parse_args()
    if a.run:
        run(a.case)
    elif a.summarize_pmc:
        summarize_pmc(*a.summarize_pmc)
    else:
        summarize(a.summarize)
