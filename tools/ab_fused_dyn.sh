# A/B of the fused-mask launch's work queue (MELF_FUSED_DYN = small segments per workgroup aimed at, 0 = the static split of rounds
# 1-4; MELF_FUSED_BIG = percent of the rows dealt as one big first segment per workgroup) at config 2 (B = 256, 640 x 480, four
# buffer pairs in rotation) and config 5 (B = 512, 1080p).
set -e
for cfg in ${CFGS:-0,0 3,60 2,70 3,70 4,70 3,80 5,80 2,50 0,0}; do
dyn=${cfg%,*}; big=${cfg#*,}
echo "== config 2 (640x480 B=256 nbuf=4) dyn=$dyn big=$big"; MELF_FUSED_DYN=$dyn MELF_FUSED_BIG=$big timeout -k 10 120 python3 tools/run_stage.py fused --iters 60 --hw 640x480 --batch 256 --nbuf 4 | grep fused
done
for cfg in ${CFGS5:-0,0 3,60 3,70 6,70 4,80 8,80 0,0}; do
dyn=${cfg%,*}; big=${cfg#*,}
echo "== config 5 (1080x1920 B=512 nbuf=1) dyn=$dyn big=$big"; MELF_FUSED_DYN=$dyn MELF_FUSED_BIG=$big timeout -k 10 120 python3 tools/run_stage.py fused --iters 16 --hw 1080x1920 --batch 512 --nbuf 1 | grep fused
done
