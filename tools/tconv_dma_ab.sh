#!/bin/bash
# Same-box A/B of the transposed conv with its footprint staged by LDS-DMA: exp_build/libmgf_old.so (HEAD) against libmgf_dma.so
D=gpurun_out/dma; mkdir -p $D
python -m pytest tests/test_hip_ops.py -m gpu -x -q -k "tconv or transposed or conv_taps or resample" > $D/tests.log 2>&1 || { tail -20 $D/tests.log; exit 1; }
tail -2 $D/tests.log
export MGF_MICRO_N=32
for v in old dma old dma; do
  echo "== $v" | tee -a $D/ab.txt
  MGF_LIB_PATH=$PWD/exp_build/libmgf_$v.so python tools/conv_micro.py r64_tconv r128_tconv r256_tconv r512_tconv r1024_tconv 2>>$D/err.txt | tee -a $D/ab.txt
done
