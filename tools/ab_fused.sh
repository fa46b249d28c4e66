# A/B of the fused-mask launch configurations (MELF_FUSED_CONFIG) with the launches rotating over 4 buffer pairs
# (HBM, not Infinity Cache), plus the timing-only twin (all traffic and barriers, no pixel math).
set -e
for hw in ${HWS:-480x640 640x480 1080x1920}; do for cfg in ${CFGS:-0 1 2 3}; do
b=256; nb=4; if [ $hw = 1080x1920 ]; then b=128; nb=2; fi
echo "== hw=$hw cfg=$cfg"; MELF_FUSED_CONFIG=$cfg timeout -k 10 120 python3 tools/run_stage.py fused --iters 40 --hw $hw --batch $b --nbuf $nb | grep fused
done; done
for cfg in 0 1; do
echo "== memonly 640x480 cfg=$cfg"; MELF_FUSED_CONFIG=$cfg MELF_FUSED_VARIANT=memonly timeout -k 10 120 python3 tools/run_stage.py fused --iters 40 --hw 640x480 | grep fused
done
