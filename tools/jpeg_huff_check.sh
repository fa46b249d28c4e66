#!/bin/bash
# k_jpeg_huff after a change: JPEG parity tests, the stamp build's phase / round times, call and kernel times.
set -e
cd "$(dirname "$0")/.."
OUT=gpurun_out/jpeg_huff_check; mkdir -p $OUT; rm -f $OUT/*.txt
python3 -m pytest tests/test_jpeg.py -x -q -m gpu > $OUT/parity.txt 2>&1 || { tail -40 $OUT/parity.txt; exit 1; }
tail -3 $OUT/parity.txt
python3 tools/jpeg_rounds.py sample-images1 crop > $OUT/rounds.txt 2>&1 || { tail -20 $OUT/rounds.txt; exit 1; }
for n in 512 1024; do
  echo "== serial n=$n" >> $OUT/timing.txt
  MELF_JPEG_SERIAL=1 python3 tools/jpeg_timing.py sample-images1 $n >> $OUT/timing.txt 2>&1
done
echo "== pipelined n=1024" >> $OUT/timing.txt
python3 tools/jpeg_timing.py sample-images1 1024 >> $OUT/timing.txt 2>&1
echo "== pipelined n=1024 sample-images2" >> $OUT/timing.txt
python3 tools/jpeg_timing.py sample-images2 1024 >> $OUT/timing.txt 2>&1
head -40 $OUT/rounds.txt; cat $OUT/timing.txt
