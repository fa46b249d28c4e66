#!/bin/bash
# Collects a round's profiles on the GPU box:  tools/profile_round.sh r04
# Order matters: the PMC passes and traffic.json FIRST, so that the bench run under rocprofv3 finds a fresh traffic.json
# (round 2 ran the bench first and its line carried `traffic: null (stale)`).
#  1. HBM traffic counters, FETCH_SIZE and WRITE_SIZE in separate --pmc passes (MI355X_MICROARCH.md: they do not fit one
#     pass; FETCH_SIZE is doubled for wide coalesced reads on gfx950), for each workload bench.py reports a roofline on:
#       config3  full path, sample-images1, 1024 frames per launch, four batches in rotation
#       config4  full path, sample-images2
#       config2  fused mask, B=256 640x480, four buffer pairs in rotation
#       config5  fused mask, 1080p, B=512
#       jpeg     1024 fixture files
#     -> traffic.json: bytes per launch for bench.py, stamped with the hash of the kernel sources they were measured on
#  2. rocprofv3 --kernel-trace --stats of the bench command on one lane and without its two_streams blocks (`--skip twostream --no-resident-hint`:
#     launches that share the chip with another stream's kernels have stretched durations and would blur the per-kernel
#     averages the roofline lines are checked against) -> bench_kernel_stats.csv + the JSON line of that run
#  3. one --kernel-trace --stats run PER WORKLOAD whose roofline the line quotes, so that every roofline.frac can be
#     recomputed from this directory alone: bench_kernel_stats_config2.csv (fused mask B=256 640x480 rotating),
#     _config5.csv (fused mask 1080p B=512), _config4.csv (full path, sample-images2, single stream, NO resident hint,
#     no event records: the clean per-kernel table of the 8-GPU headline workload), _config3.csv likewise
# Everything lands in gpurun_out/prof_<round>/; copy it to profiles/<round>/ to commit it.
set -e
export TMPDIR=/tmp
R=${1:-r04}
OUT=gpurun_out/prof_$R
rm -rf $OUT /tmp/prof && mkdir -p $OUT /tmp/prof
pmc() {  # name, command...
  name=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv --kernel-include-regex 'melf' -d /tmp/prof/pmc_${name}_$c -- "$@" > /dev/null 2>&1
  done
  python3 tools/pmc_summary.py /tmp/prof/pmc_${name}_FETCH_SIZE /tmp/prof/pmc_${name}_WRITE_SIZE > $OUT/pmc_$name.txt
  echo "pmc $name done" >> $OUT/progress.txt
}
pmc config3 python3 tools/run_stage.py full --iters 8
pmc config4 python3 tools/run_stage.py full --iters 8 --sample-dir sample-images2
# k_dials is bound by the vector units (issue, then the last waves' dependency chains): issued vector instructions and the cycles the vector units were busy (SQ_ACTIVE_INST_VALU counts
# quad-cycles per SIMD, MI355X_MICROARCH.md), the waves' lifetime, and the chip's active cycles (GRBM_GUI_ACTIVE: the sum over
# the 8 XCDs) for the clock the launch ran at -- one pass, SQ has eight slots and GRBM two
valu() {  # name, command...
  name=$1; shift
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv \
      --kernel-include-regex 'k_dials' -d /tmp/prof/valu_$name -- "$@" > /dev/null 2>&1
  python3 tools/pmc_summary.py /tmp/prof/valu_$name > $OUT/pmc_dials_$name.txt
  echo "valu $name done" >> $OUT/progress.txt
}
valu config3 python3 tools/run_stage.py full --iters 8
valu config4 python3 tools/run_stage.py full --iters 8 --sample-dir sample-images2
# k_match_mfma's matrix pipe (round 6: after the sparse pairing): matrix instructions issued, cycles the pipe was busy, wave lifetime
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv \
    --kernel-include-regex 'k_match_mfma' -d /tmp/prof/mfma_busy -- python3 tools/run_stage.py full --iters 8 > /dev/null 2>&1 || true
python3 tools/pmc_summary.py /tmp/prof/mfma_busy > $OUT/pmc_match_mfma_busy.txt || true
echo "mfma busy done" >> $OUT/progress.txt
pmc config2 python3 tools/run_stage.py fused --iters 8
pmc config5 python3 tools/run_stage.py fused --iters 4 --hw 1080x1920 --batch 512 --nbuf 1
# the fused mask needs the vector units too (12-15 instructions per pixel): their busy share, counters as for k_dials
for cfg in config2 config5; do
  if [ $cfg = config2 ]; then args="--iters 8"; else args="--iters 4 --hw 1080x1920 --batch 512 --nbuf 1"; fi
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv \
      --kernel-include-regex 'k_fused_mask' -d /tmp/prof/valu_fused_$cfg -- python3 tools/run_stage.py fused $args > /dev/null 2>&1 || true
  python3 tools/pmc_summary.py /tmp/prof/valu_fused_$cfg > $OUT/pmc_fused_valu_$cfg.txt || true
done
echo "valu fused done" >> $OUT/progress.txt
pmc jpeg python3 tools/jpeg_timing.py sample-images1 1024
# k_jpeg_huff is a latency chain (dependent decode steps of a few waves per workgroup): what its launches issue and how busy
# they keep the vector units, counters as for k_dials
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv \
    --kernel-include-regex 'k_jpeg_huff' -d /tmp/prof/valu_jpeg -- python3 tools/jpeg_timing.py sample-images1 1024 > /dev/null 2>&1 || true
python3 tools/pmc_summary.py /tmp/prof/valu_jpeg > $OUT/pmc_huff_jpeg.txt || true
echo "valu jpeg done" >> $OUT/progress.txt
python3 - "$OUT" <<'PY'
import json, os, re, sys
sys.path.insert(0, os.getcwd())
import bench
out = sys.argv[1]
def load(name):
    d = {}
    for ln in open(os.path.join(out, 'pmc_%s.txt' % name)):
        m = re.match(r'(\S+)\s+(\S+)\s+n=\d+\s+mean=(\S+)', ln)
        if m:
            d.setdefault(m.group(1), {})[m.group(2)] = float(m.group(3))
    return d
def bytes_of(k):  # KiB counters; FETCH_SIZE doubled (gfx950 wide coalesced reads)
    return int(round((2 * k.get('FETCH_SIZE', 0.0) + k.get('WRITE_SIZE', 0.0)) * 1024))
per = {}
detail = {}
for (cfg, kernels) in (('config3', ('k_match_mfma', 'k_match_gen', 'k_prep_lplane', 'k_dials')),
                       ('config4', ('k_match_mfma', 'k_match_gen', 'k_prep_lplane', 'k_dials')),
                       ('config2', ('k_fused_mask',)), ('config5', ('k_fused_mask',)),
                       ('jpeg', ('k_jpeg_huff', 'k_jpeg_idct', 'k_jpeg_color'))):
    d = load(cfg)
    for k in kernels:
        if k in d:
            short = 'k_match' if k.startswith('k_match') else k
            detail['%s:%s' % (cfg, k)] = bytes_of(d[k])
            per['%s:%s' % (cfg, short)] = bytes_of(d[k])
valu = {}
for (cfg, fname, kern) in (('config3', 'dials_config3', 'k_dials'), ('config4', 'dials_config4', 'k_dials'), ('jpeg', 'huff_jpeg', 'k_jpeg_huff'),
                           ('config2', 'fused_valu_config2', 'k_fused_mask'), ('config5', 'fused_valu_config5', 'k_fused_mask')):
    try:
        d = load(fname).get(kern, {})
    except OSError:
        d = {}
    if d.get('SQ_ACTIVE_INST_VALU') and d.get('GRBM_GUI_ACTIVE'):
        # vector units busy: quad-cycles x 4, summed over the 1024 SIMDs / (1024 SIMDs x the launch's cycles); GRBM_GUI_ACTIVE is
        # the sum over the 8 XCDs of the cycles the launch kept the chip active
        cyc = d['GRBM_GUI_ACTIVE'] / 8.0
        valu[cfg + ':' + kern] = {'insts_valu_per_launch': d.get('SQ_INSTS_VALU'), 'active_inst_valu_quadcycles': d['SQ_ACTIVE_INST_VALU'],
                                  'wave_quadcycles': d.get('SQ_WAVE_CYCLES'), 'waves': d.get('SQ_WAVES'), 'busy_cycles': d.get('SQ_BUSY_CYCLES'),
                                  'wait_inst_any_quadcycles': d.get('SQ_WAIT_INST_ANY'), 'gui_active_cycles_per_xcd': cyc,
                                  'valu_busy_frac': 4.0 * d['SQ_ACTIVE_INST_VALU'] / (1024.0 * cyc)}
json.dump({'_note': 'HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 from separate rocprofv3 --pmc passes '
                    '(tools/profile_round.sh; raw counters: pmc_*.txt next to this file); valu: k_dials vector-unit counters of one more pass',
           'kernel_sources_sha16': bench.kernel_sources_sha(), 'per_launch_bytes': per, 'per_kernel': detail, 'valu': valu},
          open(os.path.join(out, 'traffic.json'), 'w'), indent=1)
print(json.dumps(per, indent=1))
PY
# bench.py looks for profiles/r*/traffic.json: put the fresh one where it will be found (and committed)
mkdir -p profiles/$R && cp $OUT/traffic.json profiles/$R/traffic.json
stats() {  # name, command...
  name=$1; shift
  rm -rf /tmp/prof/stats_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/stats_$name -o run -- "$@" > $OUT/${name}_stdout.txt 2> $OUT/${name}_stderr.txt
  f=$(find /tmp/prof/stats_$name -name '*kernel_stats.csv' | head -1)
  head -1 "$f" > $OUT/bench_kernel_stats_$name.csv
  grep -i 'melf' "$f" >> $OUT/bench_kernel_stats_$name.csv || true
  echo "$*" > $OUT/${name}_command.txt
  echo "stats $name done" >> $OUT/progress.txt
}
# (--no-resident-hint: every timed region on ONE caller stream, so that no launch of the trace shares the chip with another lane's kernels)
stats bench python3 bench.py --skip twostream --no-resident-hint
mv $OUT/bench_kernel_stats_bench.csv $OUT/bench_kernel_stats.csv
mv $OUT/bench_stdout.txt $OUT/bench_under_rocprof.json; mv $OUT/bench_stderr.txt $OUT/bench_under_rocprof.err
stats config2 python3 tools/run_stage.py fused --iters 60 --hw 640x480 --batch 256 --nbuf 4 --profiling 0
stats config5 python3 tools/run_stage.py fused --iters 80 --hw 1080x1920 --batch 512 --nbuf 1 --profiling 0   # 90 launches: a 26-launch trace is 5 % slow from its cold start
stats config4 python3 tools/run_stage.py full --iters 60 --sample-dir sample-images2 --device-records --profiling 0
stats config3 python3 tools/run_stage.py full --iters 60 --device-records --profiling 0
rm -f $OUT/*_stderr.txt
cut -c1-160 $OUT/bench_kernel_stats.csv
for n in config2 config5 config4 config3; do echo "== $n"; cut -c1-160 $OUT/bench_kernel_stats_$n.csv; done
