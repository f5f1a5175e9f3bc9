#!/bin/bash
# Same-box A/B of library builds AND environment switches: each arm = "VARIANT|ENV=.. ENV=.." (VARIANT = suffix of
# libpathomic_hip<suffix>.so, empty = the default build), three alternations.
#     bash profiles/scripts/ab_lib_env.sh "_old|" "|PH_EW_ITEMS=4" "|PH_EW_ITEMS=8"
ARGS="--steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timer --no-parity-mode --no-variants --no-north-star-block"
for i in 1 2 3; do
  for arm in "$@"; do
    v="${arm%%|*}"; e="${arm#*|}"
    env PH_LIB_VARIANT="$v" $e python bench.py $ARGS $AB_EXTRA 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$arm]', d['ms_per_step'])"
  done
done
