#!/bin/bash
# Re-runs step 2 of tools/profile_round.sh alone (the default bench command under rocprofv3 --kernel-trace --stats), e.g.
# when the profiler stalled inside the 6 ms timed region of the first attempt:   tools/profile_bench_only.sh r03
set -e
export TMPDIR=/tmp
R=${1:-r03}
OUT=gpurun_out/prof_$R
mkdir -p $OUT
rm -rf /tmp/prof/stats_bench
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/stats_bench -o run -- python3 bench.py --skip twostream > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
f=$(find /tmp/prof/stats_bench -name '*kernel_stats.csv' | head -1)
head -1 "$f" > $OUT/bench_kernel_stats.csv
grep -i 'melf' "$f" >> $OUT/bench_kernel_stats.csv || true
echo "python3 bench.py --skip twostream" > $OUT/bench_command.txt
python3 tools/print_bench.py $OUT/bench_under_rocprof.json
