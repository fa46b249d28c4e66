set -e
export TMPDIR=/tmp
rm -rf /tmp/prof/stats_c5 && mkdir -p /tmp/prof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/stats_c5 -o run -- python3 tools/run_stage.py fused --iters 80 --hw 1080x1920 --batch 512 --nbuf 1 --profiling 0 > /dev/null 2>&1
f=$(find /tmp/prof/stats_c5 -name '*kernel_stats.csv' | head -1)
head -1 "$f" > gpurun_out/r05_bench_kernel_stats_config5_90launches.csv
grep -i 'k_fused_mask_lut' "$f" >> gpurun_out/r05_bench_kernel_stats_config5_90launches.csv
cut -c1-60,220-330 gpurun_out/r05_bench_kernel_stats_config5_90launches.csv
python3 bench.py --steps 20 --warmup 5 --only config5 > gpurun_out/r05_bench_config5_same_lease.json 2>/dev/null
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_bench_config5_same_lease.json').read().strip().splitlines()[-1])
print(d['config5']['fused_mask']['roofline'])
PY
