#!/bin/bash
# k_match_gen: tile shape / slice experiments (MELF_GEN_SHAPE=rows,colblocks,slices)
export MELF_GEN_TRACE=1 MELF_MATCH=gen
i=0
for cfg in "$@"; do
  set -- $cfg
  i=$((i+1))
  export MELF_GEN_SHAPE=$3
  timeout -k 10 120 tools/kstats.sh gs$i -- python3 tools/run_stage.py full --iters 12 --sample-dir $1 --batch $2 > gpurun_out/gen_s_$i.txt 2>&1
  echo "== $cfg"; grep "melf gen" gpurun_out/kstats_gs$i.err | head -1; grep "match" gpurun_out/gen_s_$i.txt
done
