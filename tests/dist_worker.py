"""Worker of tests/test_dist_gloo.py: one of WORLD_SIZE gloo ranks.  Each rank plans its own
block of A/P/R with the product's host logic (C ABI, no GPU), performs the halo exchange the
plans describe over torch.distributed (gloo), applies its local operator with numpy/scipy and
checks the result against the rows of the global product."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sparsh_amg_amd as sa  # noqa: E402
from sparsh_amg_amd import problems  # noqa: E402


def exchange(plan, vec_local, rank):
    """vec_local: own entries; returns [own | halo] after the planned sends/receives."""
    out = np.concatenate([vec_local, np.zeros(plan["nhalo"])])
    reqs, bufs = [], []
    for peer, off, cnt in plan["send"]:
        t = torch.from_numpy(np.ascontiguousarray(vec_local[plan["send_idx"][off:off + cnt]]))
        bufs.append(t)
        reqs.append(dist.isend(t, int(peer)))
    recvs = []
    for peer, off, cnt in plan["recv"]:
        t = torch.zeros(int(cnt), dtype=torch.float64)
        recvs.append((t, off, cnt))
        reqs.append(dist.irecv(t, int(peer)))
    for r in reqs:
        r.wait()
    for t, off, cnt in recvs:
        out[plan["nloc"] + off: plan["nloc"] + off + cnt] = t.numpy()
    return out


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    case = sys.argv[1]
    rp, ci, v = problems.poisson3d(20) if case == "p3d" else problems.random_spd(9000, 8, seed=21)
    A = sa.sp_matrix_mg(rp, ci, v).setup(sa.default_params(print_setup=0, limit_upper=500, limit_lower=250), host_only=True)
    rng = np.random.default_rng(5)  # same stream on every rank
    assert A.nlevels >= 3
    for level in range(min(3, A.nlevels - 1)):
        G = {"A": A.level_scipy(level, "A"), "P": A.level_scipy(level, "P")}
        G["R"] = G["P"].T.tocsr()
        for which in ("A", "P", "R"):
            M, plan = A.dist_local_op(level, which, rank, world)
            ncols = G[which].shape[1]
            x = rng.standard_normal(ncols)
            # which global entries of the input vector do I own?  (own block = contiguous range)
            counts = torch.zeros(world, dtype=torch.int64)
            counts[rank] = plan["nloc"]
            dist.all_reduce(counts)
            assert int(counts.sum()) == ncols, (which, level, counts)
            # ownership ranges may be handed out in reverse rank order on odd levels: find mine by
            # asking every rank for its first halo-free global column via the row offset of A
            Mcol, pcol = (M, plan) if which == "A" else A.dist_local_op(level if which == "R" else level + 1, "A", rank, world)
            lo = pcol["row0"]
            xin = exchange(plan, x[lo:lo + plan["nloc"]], rank)
            # halo entries must equal the global vector at halo_global
            assert np.array_equal(xin[plan["nloc"]:], x[plan["halo_global"]]), (which, level)
            y = M @ xin
            ref = (G[which] @ x)[plan["row0"]: plan["row0"] + M.shape[0]]
            assert np.array_equal(y, ref) or np.allclose(y, ref, rtol=1e-15, atol=0), (which, level, np.abs(y - ref).max())
            # rows are covered exactly once
            rows = torch.zeros(world, dtype=torch.int64)
            rows[rank] = M.shape[0]
            dist.all_reduce(rows)
            assert int(rows.sum()) == G[which].shape[0]
    dist.barrier()
    if rank == 0:
        print("DIST_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
