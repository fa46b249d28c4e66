#!/bin/bash
# Per-kernel durations of one command under rocprofv3 (kernel trace + stats): tools/kstats.sh OUTNAME -- python3 ...
set -e
export TMPDIR=/tmp
name=$1; shift; shift
rm -rf /tmp/kstats_$name
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kstats_$name -o t -- "$@" > gpurun_out/kstats_$name.out 2> gpurun_out/kstats_$name.err || true
f=$(find /tmp/kstats_$name -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r['Name']
    if 'melf' in n:
        print('%-60s calls %5s  avg %10.1f ns  total %10.3f ms  %5s%%' % (n[:60], r['Calls'], float(r['AverageNs']), float(r['TotalDurationNs']) / 1e6, r['Percentage']))
PY
