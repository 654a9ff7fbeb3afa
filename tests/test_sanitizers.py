"""AddressSanitizer + UBSan on the CPU builds (GPU ASan is not available on the pool): the product's
host setup / partition code and the oracle, each through a small native driver."""
import os
import subprocess

import pytest

from conftest import ROOT

ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
           OMP_NUM_THREADS="4")


def _build_and_run(cmd, exe, marker):
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, env=ENV, timeout=600)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    assert marker in r.stdout
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr


@pytest.mark.slow
def test_host_setup_and_partition_under_asan(tmp_path):
    exe = str(tmp_path / "asan_host")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fopenmp", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
           "-ffp-contract=off", os.path.join(ROOT, "tests", "cpp", "asan_host.cpp"), "-o", exe]
    _build_and_run(cmd, exe, "ASAN_HOST_OK")


@pytest.mark.slow
def test_oracle_under_asan(tmp_path):
    exe = str(tmp_path / "asan_oracle")
    cmd = ["gcc", "-std=gnu11", "-O1", "-g", "-fopenmp", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
           "-ffp-contract=off", f"-I{os.path.join(ROOT, 'oracle')}", os.path.join(ROOT, "oracle", "asan_driver.c"),
           os.path.join(ROOT, "oracle", "amg_oracle.c"), "-o", exe, "-lm"]
    _build_and_run(cmd, exe, "ASAN_ORACLE_OK")
