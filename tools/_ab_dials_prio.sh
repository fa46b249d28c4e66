C=meterelf_amd/csrc
for rep in 1 2; do
for v in key2 base; do
  if [ $v = base ]; then L=meterelf_amd/libmeterelf_hip.so; else L=$C/libmeterelf_hip_$v.so; fi
  for sd in sample-images1 sample-images2; do
    echo "== $v $sd"; MELF_LIB_PATH=$PWD/$L bash tools/kstats.sh dp_${v}_${sd}_$rep -- python3 tools/run_stage.py full --iters 100 --profiling 0 --device-records --sample-dir $sd 2>&1 | grep "k_dials"
  done
done
done
python tools/dials_clock.py sample-images1 2>&1 | grep -v amdgpu.ids | tail -16
python tools/dials_clock.py sample-images1 256 2>&1 | grep -v amdgpu.ids | tail -16
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "dial or full or golden or record or stress" 2>&1 | tail -3
