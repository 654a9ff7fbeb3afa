#!/usr/bin/env python3
"""HBM traffic of the dominant kernel from rocprofv3 PMC counters (run ON the GPU box).

Usage (two separate counter passes, as MI355X_MICROARCH.md prescribes; no trace domains mixed in):
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d OUT/fetch -- python3 tools/pmc_traffic.py --run
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d OUT/write -- python3 tools/pmc_traffic.py --run
  python3 tools/pmc_traffic.py --summarize OUT/fetch OUT/write --out profiles/pmc_latest.json

--run launches, on the 216^3 hierarchy with the default kernel policy, the finest-level fused Jacobi
sweep plus three calibration kernels of known byte counts: fp64 axpby (8-B/lane loads), fp64 dot,
int32 copy (4-B/lane loads).  --summarize converts counter values (KiB) to bytes, derives the
FETCH_SIZE correction factor of each access width from the calibration kernels (the gfx950 note:
wide coalesced reads are tallied at half their size) and applies them to the Jacobi kernel's streams.
"""
import argparse
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

N_GRID = 216


def run(grid=N_GRID, cfg=None, no_fold=False, layout_out=None, fem=0, idx16=None, iters=0, coarse_limit=None):
    import numpy as np

    import sparsh_amg_amd as sa
    from sparsh_amg_amd import problems

    rp, ci, v = problems.fem_unstructured(fem) if fem else problems.poisson3d(grid)
    A = sa.sp_matrix_mg(rp, ci, v)
    if no_fold:
        A.set_const_slots(False)
    if idx16 is not None:
        A.set_index_compression(idx16)
    # Default parameters, as the timed run.  (Round 2 forced coarse_limit = 8192 here: the block-tridiagonal factorisation
    # issues ~32 k launches and rocprofv3's counter-collection mode segfaults beyond a few ten thousand dispatches of ANY kernel
    # -- tools/micro/many_launches.hip: 10 000 trivial launches pass, 40 000 crash inside the tool,
    # profiles/r03_pmc/rocprofv3_pmc_dispatch_limit.txt.  The nested-dissection factorisation issues a few hundred.)
    A.setup(sa.default_params(print_setup=0, print_solve=0, **({} if coarse_limit is None else {"coarse_limit": coarse_limit})))
    cfg = cfg or os.environ.get("SPARSH_PMC_CFG")  # "kind,vec,nt,remap": override the per-operator policy (diagnostics)
    if cfg:
        A.set_kernel_config(*[int(t) for t in cfg.split(",")])
    double = A.level_double_sweep(0)["on"]  # the finest level's smoothing legs run double sweeps: that kernel is the one measured
    for op in ("axpby", "dot", "copy_int", "jacobi_double" if double else "jacobi"):
        A.bench_op(op, 0, 4)
    if iters > 0:
        # whole AMG-PCG iterations between marker launches (SURVEY 8d metric (ii): HBM bytes of the whole V-cycle + Krylov
        # step): the markers are single launches of `copy_int` (used by nothing in the solver; identical to the calibration
        # launches above, so the calibration average is unaffected): marker, warm-up, marker, K steps, marker
        n = len(rp) - 1
        bd = A.dev_alloc(8 * n)
        xd = A.dev_alloc(8 * n)
        A.h2d(bd, np.ones(n))
        A.h2d(xd, np.zeros(n))
        A.set_stopping(0.0, 100000, 1 << 30)
        A.bench_op("copy_int", 0, 1)
        A.krylov_init_dev("pcg", bd, xd)
        A.krylov_step_dev(2)
        A.sync()
        A.bench_op("copy_int", 0, 1)
        A.krylov_step_dev(iters)
        A.sync()
        A.bench_op("copy_int", 0, 1)
    slots, vblocks, meta = A.level_layout(0)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if layout_out is None:
        os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
        layout_out = os.path.join(root, "gpurun_out", "pmc_layout.json")
    with open(layout_out, "w") as f:
        b16, nblk = A.level_index16(0)
        json.dump({"slots": slots, "value_blocks": vblocks, "descriptor_bytes": meta, "kernel": "sdia_box2_kernel" if double else A.level_kernel(0), "grid": grid,
                   "nrow": len(rp) - 1, "nnz": int(rp[-1]), "stored_entries": A.level_format(0)[1], "iters": iters,
                   "index16_blocks": b16, "row_blocks": nblk}, f)


def collect(d, jacobi_kernel=None):
    """jacobi_kernel: count only launches of this kernel as the measured sweep (the setup's placement search and the
    solver launch other Jacobi kernels in the same process)."""
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    out = {}
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        key = None
        if "axpby_kernel" in name:
            key = "axpby"
        elif "dot_kernel" in name and "dot2" not in name:
            key = "dot"
        elif "copy_int_kernel" in name:
            key = "copy_int"
        elif jacobi_kernel == "sdia_box2_kernel" and "sdia_box2_kernel<" in name and ", 1, false>" in name:  # finest level, not the from-zero variant
            key = "jacobi"
            out["_kind"] = [4.0]
        elif ("sdia_kernel<2" in name or "sdia_tab_kernel<2" in name or "sdia_ord_kernel<2" in name or "sell_kernel<2" in name
              or "csr_block_kernel<2" in name or "csr_rowlane_kernel<2" in name or "csr_rowlane16_kernel<2" in name) and ", 1>" in name \
                and (jacobi_kernel is None or (jacobi_kernel + "<") in name):
            key = "jacobi"
            out["_kind"] = [3.0 if "sdia_" in name else (2.0 if "sell_kernel" in name else (1.0 if "rowlane16" in name else 0.0))]
        if key:
            out.setdefault(key, []).append(float(r["Counter_Value"]) * 1024.0)
    return {k: sum(v) / len(v) for k, v in out.items()}, {k: len(v) for k, v in out.items()}


def collect_iterations(d):
    """Sum of the counter over every dispatch between the last two `copy_int` marker launches of run(iters=K)."""
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Dispatch_Id"]))
    marks = [i for i, r in enumerate(rows) if "copy_int_kernel" in r["Kernel_Name"]]
    # a marker is one bench_op call = a run of consecutive copy_int launches (3 warm-up + reps): group them
    groups = []
    for i in marks:
        if groups and i == groups[-1][1] + 1:
            groups[-1][1] = i
        else:
            groups.append([i, i])
    if len(groups) < 3:
        return None
    a, b = groups[-2][1], groups[-1][0]
    total = 0.0
    by_kernel = {}
    for r in rows[a + 1:b]:
        val = float(r["Counter_Value"]) * 1024.0
        total += val
        k = r["Kernel_Name"].replace("void ", "").replace("sparsh::", "").replace("(anonymous namespace)::", "").split("(")[0]
        e = by_kernel.setdefault(k, [0, 0.0])
        e[0] += 1
        e[1] += val
    return total, b - a - 1, by_kernel


def summarize(fetch_dir, write_dir, out_path, grid=N_GRID, layout_path=None, quiet=False):
    n = grid ** 3
    nnz = 7 * n - 6 * grid ** 2
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lay_path = layout_path or os.path.join(root, "gpurun_out", "pmc_layout.json")
    layout = json.load(open(lay_path)) if os.path.exists(lay_path) else None
    jk = layout.get("kernel") if layout else None
    fetch, cnt = collect(fetch_dir, jk)
    write, _ = collect(write_dir, jk)
    known_read = {"axpby": 16 * n, "dot": 16 * n, "copy_int": 4 * n}
    known_write = {"axpby": 8 * n, "dot": 0, "copy_int": 4 * n}
    f8 = known_read["axpby"] / fetch["axpby"]      # 8-byte-per-lane loads
    f8b = known_read["dot"] / fetch["dot"]
    f4 = known_read["copy_int"] / fetch["copy_int"]  # 4-byte-per-lane loads
    wf8 = known_write["axpby"] / write["axpby"]
    wf4 = known_write["copy_int"] / write["copy_int"]
    # Jacobi kernel streams.  sliced ELL: 4-byte loads = column indices (padded) + rowptr; sliced
    # diagonals: no per-entry index (offsets/masks are per slot: < 1.5 % of the bytes, counted with
    # the 8-byte streams).  raw = bytes4/f4 + bytes8/f8  ->  solve for bytes8.
    pad_nnz = 7 * n
    if layout and layout.get("nrow"):  # the matrix the --run pass actually used (e.g. the unstructured stand-in)
        n, nnz = layout["nrow"], layout["nnz"]
        known_read = {"axpby": 16 * n, "dot": 16 * n, "copy_int": 4 * n}
        known_write = {"axpby": 8 * n, "dot": 0, "copy_int": 4 * n}
        f8 = known_read["axpby"] / fetch["axpby"]
        f8b = known_read["dot"] / fetch["dot"]
        f4 = known_read["copy_int"] / fetch["copy_int"]
        wf8 = known_write["axpby"] / write["axpby"]
        wf4 = known_write["copy_int"] / write["copy_int"]
        pad_nnz = layout.get("stored_entries", nnz)
    kind = int(fetch.pop("_kind", 2.0))
    cnt.pop("_kind", None)
    # 4-byte-per-lane streams: column indices (+ rowptr); the 16-bit delta form reads one 4-byte word per entry PAIR
    bytes4 = (4 * pad_nnz + 4 * n) if kind == 2 else ((4 * nnz + 4 * n) if kind == 0 else ((2 * nnz + 4 * n) if kind == 1 else 0))
    bytes8 = (fetch["jacobi"] - bytes4 / f4) * f8
    read_total = bytes4 + bytes8
    # bytes of the value stream: sliced diagonals store 64-value blocks only for non-constant slots
    val_bytes = 8 * pad_nnz
    meta_bytes = 0
    if kind == 3 and layout is not None:
        val_bytes = 512 * layout["value_blocks"]
        meta_bytes = layout.get("descriptor_bytes", 24 * layout["slots"])
    alg = 12 * nnz + 36 * n
    if kind == 4:  # double sweep on a box grid: no matrix stream at all; what one launch (= two sweeps) has to move: x and b in, y out
        val_bytes = 0
        meta_bytes = 0
        alg = 24 * n
    res = {
        "_how": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes); counter KiB -> bytes; per-width correction from "
                "calibration kernels of known size run in the same process (tools/pmc_traffic.py)",
        "raw_fetch_bytes": fetch, "raw_write_bytes": write, "dispatches": cnt,
        "fetch_correction": {"8B_per_lane(axpby)": f8, "8B_per_lane(dot)": f8b, "4B_per_lane(copy_int)": f4},
        "write_correction": {"8B_per_lane": wf8, "4B_per_lane": wf4},
        "jacobi_fine_read_bytes": read_total,
        "jacobi_fine_write_bytes": write["jacobi"] * wf8,
        "jacobi_fine_bytes_per_launch": read_total + write["jacobi"] * wf8,
        "jacobi_fine_algorithmic_bytes": alg,
        "ratio_traffic_over_algorithmic": (read_total + write["jacobi"] * wf8) / alg,
        "kernel_family": {3: "sdia_kernel (sliced diagonals)", 2: "sell_kernel (sliced ELL)", 0: "CSR-stream (csr_block_kernel / csr_rowlane_kernel)",
                          1: "CSR-stream, 16-bit delta column indices (csr_rowlane16_kernel)",
                          4: "sdia_box2_kernel (two Jacobi sweeps per launch on a box grid; algorithmic bytes = 24 n per LAUNCH)"}.get(kind),
        "layout": layout,
        "x_vector_fetches_per_entry": (bytes8 - val_bytes - meta_bytes - 8 * n) / (8 * n),
    }
    if layout and layout.get("iters"):
        K = layout["iters"]
        fi = collect_iterations(fetch_dir)
        wi = collect_iterations(write_dir)
        if fi and wi:
            # every load stream of these kernels is 4- or 8-byte-per-lane (both calibrated, both x2 within 0.1 %): the 8-byte factor is applied to the sum
            rd = fi[0] * f8 / K
            wr = wi[0] * wf8 / K
            res["iteration"] = {
                "iterations_measured": K, "dispatches_per_iteration": fi[1] / K,
                "hbm_read_bytes_per_iteration": rd, "hbm_write_bytes_per_iteration": wr, "hbm_bytes_per_iteration": rd + wr,
                "top_kernels_by_read_bytes": {k: {"launches_per_iteration": v[0] / K, "read_bytes_per_iteration": v[1] * f8 / K}
                                              for k, v in sorted(fi[2].items(), key=lambda kv: -kv[1][1])[:8]},
            }
    if out_path:
        with open(out_path, "w") as f:
            json.dump(res, f, indent=1)
    if not quiet:
        print(json.dumps(res, indent=1))
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--run", action="store_true")
    ap.add_argument("--summarize", nargs=2)
    ap.add_argument("--out", default="profiles/pmc_latest.json")
    ap.add_argument("--grid", type=int, default=N_GRID)
    ap.add_argument("--kcfg", default=None, help="kind,vec,nt,remap")
    ap.add_argument("--no-fold", action="store_true")
    ap.add_argument("--layout", default=None, help="where --run writes / --summarize reads the layout description")
    ap.add_argument("--fem", type=int, default=0, help="--run on the unstructured P1-FEM stand-in with this many points instead of the grid")
    ap.add_argument("--idx16", type=int, default=None, help="--run: sparsh_set_index_compression mode (2 = every operator)")
    ap.add_argument("--iters", type=int, default=0, help="--run: also K whole AMG-PCG iterations between marker launches (bytes per iteration)")
    ap.add_argument("--coarse-limit", type=int, default=None, help="--run: sparsh_params.coarse_limit (default: the library's)")
    a = ap.parse_args()
    if a.run:
        run(a.grid, a.kcfg, a.no_fold, a.layout, a.fem, a.idx16, a.iters, a.coarse_limit)
    elif a.summarize:
        summarize(a.summarize[0], a.summarize[1], a.out, a.grid, a.layout)
