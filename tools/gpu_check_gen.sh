#!/bin/bash
# general matrix-core kernel: parity tests first, then per-kernel times; every step under its own timeout, and no step
# after a failed one
set -e
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "general_matrix or either_matrix or match" > gpurun_out/gen_tests.log 2>&1 || { tail -30 gpurun_out/gen_tests.log; exit 1; }
tail -3 gpurun_out/gen_tests.log
export MELF_GEN_TRACE=1
for cfg in ${CFGS:-"f4 sample-images2 1024 gen" "f3g sample-images1 1024 gen" "f3g5 sample-images1 512 gen" "f3g2 sample-images1 256 gen" "f3g1 sample-images1 64 gen" "f3f1 sample-images1 64 fast"}; do
  set -- $cfg
  if [ "$4" = auto ]; then unset MELF_MATCH; else export MELF_MATCH=$4; fi
  timeout -k 10 120 tools/kstats.sh $1 -- python3 tools/run_stage.py full --iters 12 --sample-dir $2 --batch $3 > gpurun_out/gen_k_$1.txt 2>&1
  echo "== $cfg"; grep "melf gen" gpurun_out/kstats_$1.err | head -1; grep "match" gpurun_out/gen_k_$1.txt
done
