# Clean per-kernel table of the full path (single stream, no event records, no resident hint), rocprofv3 kernel trace:
#   bash tools/clean_kernel_table.sh          (GPU box)
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-.}
for sd in sample-images2 sample-images1; do
rm -rf /tmp/c4 && mkdir -p /tmp/c4
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c4 -o run -- python3 tools/run_stage.py full --iters 60 --sample-dir $sd --device-records --profiling 0 > /dev/null 2>&1
f=$(find /tmp/c4 -name '*kernel_stats.csv' | head -1)
echo "== $sd (1024 frames per launch, 60 launches, 4 batches in rotation)"
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'melf' in r['Name']:
        print('%-60s calls %5s  avg %9.1f ns  min %9s  max %9s' % (r['Name'][:60], r['Calls'], float(r['AverageNs']), r['MinNs'], r['MaxNs']))
PY
done
