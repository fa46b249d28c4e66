# A/B of the fused-mask launch configurations (MELF_FUSED_CONFIG) on both frame orientations.
set -e
for hw in 480x640 640x480; do for cfg in ${CFGS:-0 1 2 3}; do
echo "== hw=$hw cfg=$cfg"; MELF_FUSED_CONFIG=$cfg timeout -k 10 120 python3 tools/run_stage.py fused --iters 40 --hw $hw | grep fused
done; done
for cfg in 0 2; do
echo "== memonly 480x640 cfg=$cfg"; MELF_FUSED_CONFIG=$cfg MELF_FUSED_VARIANT=memonly timeout -k 10 120 python3 tools/run_stage.py fused --iters 40 --hw 480x640 | grep fused
done
