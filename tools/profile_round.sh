#!/bin/bash
# Collects the round's profiles on the GPU box: kernel-trace stats of the default bench command, then HBM
# traffic counters (FETCH_SIZE, WRITE_SIZE in separate --pmc passes, as MI355X_MICROARCH.md prescribes) for
# the fused-mask stage, the full pipeline and the JPEG path.  Writes small summaries under gpurun_out/prof/.
set -e
export TMPDIR=/tmp
OUT=gpurun_out/prof
rm -rf $OUT /tmp/prof && mkdir -p $OUT /tmp/prof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/stats -o bench -- python3 bench.py --steps 20 --warmup 3 > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
f=$(find /tmp/prof/stats -name '*kernel_stats.csv' | head -1)
head -1 "$f" > $OUT/bench_kernel_stats.csv
grep -i 'melf' "$f" >> $OUT/bench_kernel_stats.csv || true
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv --kernel-include-regex 'melf' -d /tmp/prof/pmc_fused_$c -- python3 tools/run_stage.py fused --iters 5 > /dev/null 2>&1
  rocprofv3 --pmc $c --output-format csv --kernel-include-regex 'melf' -d /tmp/prof/pmc_full_$c -- python3 tools/run_stage.py full --iters 5 > /dev/null 2>&1
  rocprofv3 --pmc $c --output-format csv --kernel-include-regex 'melf' -d /tmp/prof/pmc_jpeg_$c -- python3 tools/jpeg_timing.py sample-images1 1024 > /dev/null 2>&1
done
python3 tools/pmc_summary.py /tmp/prof/pmc_fused_FETCH_SIZE /tmp/prof/pmc_fused_WRITE_SIZE > $OUT/pmc_fused.txt
python3 tools/pmc_summary.py /tmp/prof/pmc_full_FETCH_SIZE /tmp/prof/pmc_full_WRITE_SIZE > $OUT/pmc_full.txt
python3 tools/pmc_summary.py /tmp/prof/pmc_jpeg_FETCH_SIZE /tmp/prof/pmc_jpeg_WRITE_SIZE > $OUT/pmc_jpeg.txt
cat $OUT/pmc_fused.txt $OUT/pmc_full.txt $OUT/pmc_jpeg.txt
