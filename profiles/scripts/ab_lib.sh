#!/bin/bash
# same-box A/B of two builds of the library: libpathomic_hip.so against libpathomic_hip<tag>.so
#     bash profiles/scripts/ab_lib.sh _b [extra bench flags]
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=$1; shift
run() { python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timer --no-parity-mode --no-variants "$@" 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for i in 1 2 3; do
  echo "default lib   $(run "$@")"
  echo "variant $TAG   $(PH_LIB_VARIANT=$TAG run "$@")"
done
