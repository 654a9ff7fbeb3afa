#!/bin/bash
# Round-end verification on the GPU box (one gpurun call): GPU tests, smoke, bench, per-level kernel
# bench, rocprofv3 kernel stats of the bench, PMC traffic of the dominant kernel.  Outputs under
# gpurun_out/ ; copy what should be judged into profiles/.
set -o pipefail
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1 || { tail -30 gpurun_out/gpu_tests.log; exit 1; }
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1 || exit 2
python bench.py > gpurun_out/bench_latest.json 2> gpurun_out/bench_latest.err || exit 3
python tools/kernel_bench.py --levels 13 > gpurun_out/kb_final.txt 2>&1 || exit 4
python tools/config_bench.py > gpurun_out/configs.json 2> gpurun_out/configs.err || exit 9
cd /tmp
rm -rf $R/gpurun_out/prof_stats $R/gpurun_out/pmc_run
# the same bench command under the kernel tracer (no counter passes / family runs / CPU leg inside a traced process)
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -- python3 $R/bench.py --no-cpu --no-pmc --no-families > $R/gpurun_out/prof_stats_bench.json 2> $R/gpurun_out/prof_stats.err || exit 5
cd $R
find gpurun_out/prof_stats -name "*kernel_trace.csv" -size +4M -delete   # the per-dispatch trace is large; the stats summary is what is kept
# (HBM traffic of the dominant kernel is measured by bench.py itself: gpurun_out/pmc_bench_run.json)
tail -2 gpurun_out/gpu_tests.log; tail -1 gpurun_out/smoke.log
