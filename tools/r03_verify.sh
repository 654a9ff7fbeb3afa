#!/bin/bash
# Round-3 verification on the GPU box (one gpurun call): GPU tests, smoke, bench (with PMC + CPU leg), bench under the kernel tracer,
# the other BASELINE configurations, the RCCL path with one rank.  Outputs under gpurun_out/r03_final/ ; copy what should be judged into profiles/.
set -o pipefail
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/r03_final
rm -rf $O; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout -k 10 2400 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1 || { tail -30 $O/gpu_tests.log; echo "GPU TESTS FAILED"; }
tail -3 $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || echo "SMOKE FAILED"; tail -1 $O/smoke.log
python bench.py > $O/bench.json 2> $O/bench.err || echo "BENCH FAILED"
python tools/config_bench.py > $O/configs.json 2> $O/configs.err || echo "CONFIG BENCH FAILED"; grep -E "^\[" $O/configs.err | tail -40
python bench.py --rccl --no-cpu --no-pmc --no-families --steps 5 > $O/bench_rccl_1rank.json 2> $O/bench_rccl_1rank.err || echo "RCCL 1-rank path FAILED"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python3 $R/bench.py --no-cpu --no-pmc --no-families > $O/bench_under_rocprof_trace.json 2> $O/prof_stats.err || echo "TRACED BENCH FAILED"
cd $R
find $O/prof_stats -name "*kernel_trace.csv" -size +4M -delete
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("bench", d["value"], "it/s", d["ms_per_step"], "ms  roofline frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"], "whole", json.dumps(d["config"]["whole_iteration"])[:300])
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["hierarchy"][:120])
PY
