for ps in 0 1; do for sm in 1 2 4; do for cfg in 0 4; do
echo "== plain=$ps segmult=$sm cfg=$cfg"; MELF_FUSED_PLAINSTORE=$ps MELF_FUSED_SEGMULT=$sm MELF_FUSED_CONFIG=$cfg timeout -k 10 120 python3 tools/run_stage.py fused --iters 40 --hw 640x480 --batch 256 --nbuf 4 2>&1 | grep fused
done; done; done
