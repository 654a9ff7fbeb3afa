#!/bin/bash
# stability checks: long timed region, the torchrun launcher with one rank (+ RCCL transport installed), the graph-replay test
set -o pipefail
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/r03_final
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "graph" 2>&1 | tail -3
python bench.py --steps 200 --warmup 5 --no-cpu --no-pmc --no-families > $O/bench_200steps.json 2> $O/bench_200steps.err; echo "bench 200 rc=$?"
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu --no-pmc --no-families --rccl > $O/bench_torchrun_1rank.json 2> $O/bench_torchrun_1rank.err; echo "torchrun 1 rank rc=$?"
python - <<PY
import json
for f in ("bench_200steps.json","bench_torchrun_1rank.json"):
    d=json.loads(open("$O/"+f).read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d["n_gpus"], d["config"]["parallelism"][:60], d.get("failed"))
PY
