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


def deep_exchange(plan, vec_ext, rank):
    """Fill the ghost layers (<= the plan's depth) of vec_ext from their owners: the staged exchange of csrc/comm.cpp."""
    reqs, bufs = [], []
    for peer, off, cnt in plan["send"]:
        t = torch.from_numpy(np.ascontiguousarray(vec_ext[plan["send_idx"][off:off + cnt]]))
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
        vec_ext[plan["recv_pos"][off:off + cnt]] = t.numpy()


def deep_halo_leg(A, level, rank, world, rng, nu):
    """One deep-halo smoothing leg on CPU (numpy), exactly as Engine::smooth runs it on a deep level: ONE exchange of the
    K = nu + 1 ghost layers of x and of the K - 1 layers of b, then nu Jacobi sweeps over a shrinking row prefix -- own rows
    and the first ghost layer must equal nu global sweeps bit for bit."""
    G = A.level_scipy(level, "A")
    n = G.shape[0]
    K = nu + 1
    M, plx = A.dist_deep_op(level, rank, world, K, K)
    _, plb = A.dist_deep_op(level, rank, world, K, K - 1) if K > 1 else (None, None)
    omega = 0.66667
    x = rng.standard_normal(n)
    b = rng.standard_normal(n)
    d = G.diagonal()
    ref = x.copy()
    for _ in range(nu):   # global sweeps, row sums in stored order (scipy's csr matvec adds in stored order)
        ref = ref + omega * (b - G @ ref) / d
    gof, le, nloc = plx["global_of"], plx["layer_end"], plx["nloc"]
    own = gof[:nloc]
    xe = np.zeros(plx["nall"])
    be = np.zeros(plx["nall"])
    de = np.ones(plx["nall"])
    xe[:nloc] = x[own]
    be[:nloc] = b[own]
    valid = gof >= 0
    de[valid] = d[gof[valid]]
    deep_exchange(plx, xe, rank)
    if plb is not None:
        deep_exchange(plb, be, rank)
    # what arrived is the global vector on the ghost layers
    assert np.array_equal(xe[plx["npad"]:], x[gof[plx["npad"]:]]), "ghost layers of x"
    nb = plb["layer_end"][K - 1] if plb is not None else plx["npad"]
    assert np.array_equal(be[plx["npad"]:nb], b[gof[plx["npad"]:nb]]), "ghost layers of b"
    for s in range(1, nu + 1):
        rows = le[K - s]
        y = M[:rows] @ xe
        xe[:rows] = xe[:rows] + omega * (be[:rows] - y) / de[:rows]
    upto = le[1]
    got, want = xe[:upto][gof[:upto] >= 0], ref[gof[:upto][gof[:upto] >= 0]]
    assert np.array_equal(got, want) or np.allclose(got, want, rtol=1e-15, atol=0), np.abs(got - want).max()


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
    # deep-halo (communication-avoiding) smoothing legs on the two finest levels
    for level in range(min(2, A.nlevels - 1)):
        for nu in (1, 2, 7):
            deep_halo_leg(A, level, rank, world, rng, nu)
    dist.barrier()
    if rank == 0:
        print("DIST_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
