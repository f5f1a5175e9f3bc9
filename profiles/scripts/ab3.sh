#!/bin/bash
# same-box comparison of wgrad workgroup targets x backward overlap
cd "$GRAFT_REPO_ROOT" || exit 1
run() { python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timer --no-parity-mode --no-variants "$@" 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for i in 1 2 3; do
  echo "512 serial  $(PH_WG_WANT=512 run --no-bwd-overlap)"
  echo "256 overlap $(PH_WG_WANT=256 run)"
  echo "256 overlap nosettle $(PH_BWD_NO_SETTLE=1 PH_WG_WANT=256 run)"
  echo "192 overlap $(PH_WG_WANT=192 run)"
  echo "320 overlap $(PH_WG_WANT=320 run)"
done
