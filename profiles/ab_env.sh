#!/bin/bash
# Same-box A/B of library switches given as environment assignments, e.g.
#     bash profiles/ab_env.sh "PH_TAP4=0 PH_BST=0" "PH_TAP4=1 PH_BST=0" "PH_TAP4=1 PH_BST=1"
# alternates the arms three times (box-to-box spread is +-5 %: never compare numbers from different gpurun calls).
ARGS="--steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timer --no-parity-mode --no-variants --no-north-star-block"
for i in 1 2 3; do
  for arm in "$@"; do
    env $arm python bench.py $ARGS $AB_EXTRA 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$arm', d['ms_per_step'])"
  done
done
