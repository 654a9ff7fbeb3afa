mkdir -p gpurun_out/r03_final && python bench.py > gpurun_out/r03_final/bench.json 2> gpurun_out/r03_final/bench.err; tail -c 200 gpurun_out/r03_final/bench.err
