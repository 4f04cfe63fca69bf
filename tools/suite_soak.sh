#!/bin/bash
# Soak of the GPU suite under shifted torch seeds (tests/conftest.py: MGF_SOAK_OFFSET): every HIP-vs-oracle comparison on other random inputs.
#   bash tools/suite_soak.sh OUT "1 2" [pytest args]        Tests that re-draw a committed fixture's inputs by seed fail on every offset by a lot: not findings.
D=${1:-gpurun_out/ssoak}; mkdir -p $D
shift; OFFS=${1:-1 2}; shift
for k in $OFFS; do
  echo "== MGF_SOAK_OFFSET=$k"
  MGF_SOAK_OFFSET=$k python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --deselect tests/test_hip_fuzz.py "$@" > $D/soak_$k.log 2>&1
  tail -1 $D/soak_$k.log; grep "^FAILED" $D/soak_$k.log | cut -c1-230
done
exit 0
