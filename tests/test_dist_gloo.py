"""N>1 host logic on CPU: world_size-2 and -3 gloo runs of tests/dist_worker.py."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT


@pytest.mark.parametrize("world,case", [(2, "p3d"), (3, "ragged"), (2, "ragged")])
def test_partition_and_halo_plans_gloo(world, case):
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", "2")
    port = 29500 + (os.getpid() % 2000) + world
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "dist_worker.py"), case]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    assert "DIST_OK" in r.stdout
