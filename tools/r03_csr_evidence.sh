#!/bin/bash
# Round-3 evidence for the CSR-stream family at 216^3: rocprofv3 kernel stats of the bench with kind 0 forced
# (32-bit and 16-bit column indices), FETCH_SIZE / WRITE_SIZE passes of the finest-level sweep of both kernels, whole-iteration
# HBM bytes of the default path, and the counter pass with the block-tridiagonal coarse solver active (crash under --pmc).
set -o pipefail
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/r03_csr
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for v in 32 16; do
  if [ $v = 16 ]; then export SPARSH_BENCH_IDX16=2; else unset SPARSH_BENCH_IDX16; fi
  SPARSH_BENCH_KCFG=0,3,-1,-1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_idx$v -- python3 $R/bench.py --no-cpu --no-pmc --no-families \
     > $O/bench_idx$v.json 2> $O/bench_idx$v.err || exit 1
  find $O/stats_idx$v -name "*kernel_trace.csv" -size +4M -delete
  echo "stats idx$v done"
done
unset SPARSH_BENCH_IDX16
for v in 32 16; do
  X=""; [ $v = 16 ] && X="--idx16 2"
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_idx${v}_$c -- python3 $R/tools/pmc_traffic.py --run --kcfg 0,3,-1,-1 $X --layout $O/layout_idx$v.json \
      > $O/pmc_idx${v}_$c.log 2>&1 || exit 2
  done
  python3 $R/tools/pmc_traffic.py --summarize $O/pmc_idx${v}_FETCH_SIZE $O/pmc_idx${v}_WRITE_SIZE --layout $O/layout_idx$v.json --out $O/pmc_csr_stream_idx$v.json > /dev/null || exit 3
  echo "pmc idx$v done"
done
# whole-iteration HBM bytes, default path (table kernels), 10 iterations
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_iter_$c -- python3 $R/tools/pmc_traffic.py --run --iters 10 --layout $O/layout_iter.json \
    > $O/pmc_iter_$c.log 2>&1 || exit 4
done
python3 $R/tools/pmc_traffic.py --summarize $O/pmc_iter_FETCH_SIZE $O/pmc_iter_WRITE_SIZE --layout $O/layout_iter.json --out $O/pmc_iteration_default.json > /dev/null || exit 5
echo "pmc iteration done"
find $O -name "*counter_collection.csv" -size +8M -delete
# counter pass with the block-tridiagonal coarse factorisation active (100^3, default coarse_limit)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_bt -- python3 $R/tools/pmc_traffic.py --run --grid 100 --coarse-limit 40000 --layout $O/layout_bt.json \
   > $O/pmc_bt.log 2>&1
echo "pmc with block-tridiagonal coarse solver: exit code $?" | tee $O/pmc_bt_exit.txt
tail -20 $O/pmc_bt.log
find $O -name "*counter_collection.csv" -size +8M -delete
