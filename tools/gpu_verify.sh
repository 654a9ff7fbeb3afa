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
SPARSH_BENCH_NO_CPU=1 SPARSH_BENCH_NO_GENERAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -- python3 $R/bench.py > $R/gpurun_out/prof_stats_bench.json 2> $R/gpurun_out/prof_stats.err || exit 5
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_run/fetch -- python3 $R/tools/pmc_traffic.py --run > $R/gpurun_out/pmc_fetch.log 2>&1 || exit 6
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_run/write -- python3 $R/tools/pmc_traffic.py --run > $R/gpurun_out/pmc_write.log 2>&1 || exit 7
cd $R
python3 tools/pmc_traffic.py --summarize gpurun_out/pmc_run/fetch gpurun_out/pmc_run/write --out gpurun_out/pmc_latest.json > gpurun_out/pmc_summary.txt 2>&1 || exit 8
tail -2 gpurun_out/gpu_tests.log; tail -1 gpurun_out/smoke.log
