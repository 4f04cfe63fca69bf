#!/bin/bash
# Soak: the operator-level fuzz file under shifted seeds (other shapes, other data than the committed cases):  bash tools/fuzz_soak.sh OUT "1 2 3 4 5 6" [-k expr]
# The LPIPS-GRADIENT cases are kink-sensitive by construction (random vgg / alex backbones: the float32 oracle moves its own gradient by 2 - 8 % under a 1e-5
# input perturbation; tools/soak_lpips_probe.py tells a kink from a defect) -- a failure there needs that probe, any other failure is a defect.
D=${1:-gpurun_out/soak}; mkdir -p $D
for k in ${2:-1 2 3}; do
  echo "== MGF_FUZZ_OFFSET=$k"
  MGF_FUZZ_OFFSET=$k python -m pytest tests/test_hip_fuzz.py -m gpu -q --no-header -p no:cacheprovider "${@:3}" > $D/soak_$k.log 2>&1; rc=$?
  tail -1 $D/soak_$k.log
  if [ $rc -ne 0 ]; then grep -n "Error\|assert\|FAILED" $D/soak_$k.log | head -20; fi
done
exit 0
