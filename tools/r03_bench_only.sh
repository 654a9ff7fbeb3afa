#!/bin/bash
# One default bench.py run on the GPU box (PMC + CPU leg included), output under gpurun_out/r03_final/ :  gpurun -- 'bash tools/r03_bench_only.sh'
mkdir -p gpurun_out/r03_final && python bench.py > gpurun_out/r03_final/bench.json 2> gpurun_out/r03_final/bench.err; tail -c 200 gpurun_out/r03_final/bench.err
