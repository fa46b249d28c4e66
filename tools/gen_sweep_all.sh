#!/bin/bash
# k_match_gen shape sweeps over the batch sizes and crops the planner's cost model is fitted to (GPU box):
#   tools/gen_sweep_all.sh OUTDIR   -> OUTDIR/sweep_<sample dir>_<batch>.txt + summary.txt (planner's choice against the best shape)
out=${1:-gpurun_out/gen_sweep}
mkdir -p $out
: > $out/summary.txt
for cfg in "sample-images2 64" "sample-images2 320" "sample-images2 512" "sample-images2 1024" "sample-images2 2048" "sample-images2 4096" \
           "sample-images1 64 nx2" "sample-images1 128 nx2" "sample-images1 256 nx2" "sample-images1 288 nx2"; do
  set -- $cfg
  timeout -k 10 300 python3 tools/gen_shape_sweep.py $cfg 2>&1 | grep -v '^/opt' > $out/sweep_$1_$2.txt
  python3 - $out/sweep_$1_$2.txt "$cfg" >> $out/summary.txt <<'PY'
import re, sys
rows = []
for ln in open(sys.argv[1]):
    m = re.match(r'pass \d+ (\S+)\s+k_match (\S+) ms\s+(\S+)\s+waves\s+(\d+).*records (.*)', ln)
    if m:
        rows.append((m.group(1), float(m.group(2)), m.group(3), m.group(5).startswith('same')))
if rows:
    d = [r for r in rows if r[0] == 'default'][0]
    best = min((r for r in rows if r[0] != 'default'), key=lambda r: r[1])
    bad = [r[0] for r in rows if not r[3]]
    print('%-24s planner %-12s %.4f ms | best %-12s %.4f ms | planner / best %.3f%s' % (sys.argv[2], d[2], d[1], best[2], best[1], d[1] / best[1],
                                                                                      '  RECORDS DIFFER: %s' % bad if bad else ''))
PY
done
cat $out/summary.txt
