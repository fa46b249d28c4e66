for sm in 1 2 4 8 16 32; do
echo "== 640x480 B=256 nbuf=4 segmult=$sm"; MELF_FUSED_SEGMULT=$sm timeout -k 10 120 python3 tools/run_stage.py fused --iters 40 --hw 640x480 --batch 256 --nbuf 4 | grep fused
done
for sm in 1 2 4 8 16 32; do
echo "== 1080x1920 B=512 nbuf=1 segmult=$sm"; MELF_FUSED_SEGMULT=$sm timeout -k 10 120 python3 tools/run_stage.py fused --iters 12 --hw 1080x1920 --batch 512 --nbuf 1 | grep fused
done
