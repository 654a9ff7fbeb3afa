import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sparsh_amg_amd as sa
from sparsh_amg_amd import problems
m = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
t0 = time.time()
rp, ci, v = problems.poisson2d(m)
print("generated", len(rp) - 1, "rows in %.1f s" % (time.time() - t0), flush=True)
for kw in (dict(coarse_limit=1 << 30), dict()):
    t0 = time.time()
    A = sa.sp_matrix_mg(rp, ci, v).setup(sa.default_params(print_setup=1, print_solve=0, **kw))
    print("setup %.2f s wall, %.2f s reported" % (time.time() - t0, A.setup_seconds), A.coarse_info(), flush=True)
    print("coarse solve us", A.bench_op("coarse", A.nlevels - 1, 10) * 1e6, flush=True)
    n = len(rp) - 1
    b = np.ones(n)
    bd, xd = A.dev_alloc(8 * n), A.dev_alloc(8 * n)
    A.h2d(bd, b)
    for mth, cap in (("pcg", 500), ("amg", 300)):
        A.h2d(xd, np.zeros(n))
        A.set_stopping(1e-8, cap, 1)
        t0 = time.time()
        h, it, sec, rc = A.solve_dev(mth, bd, xd)
        print(mth, "iterations", it, "rc", rc, "seconds %.3f" % sec, "first/last residual", h[0], h[-1], "wall %.1f" % (time.time() - t0), flush=True)
    A.close()
