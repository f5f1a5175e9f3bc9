#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
run() { python bench.py --no-cpu-baseline --no-kernel-timer --no-parity-mode --no-variants "$@" 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for i in 1 2; do
  echo "K10 W3  $(run --steps 10 --warmup 3)"
  echo "K10 W8  $(run --steps 10 --warmup 8)"
  echo "K20 W3  $(run --steps 20 --warmup 3)"
  echo "K20 W5  $(run --steps 20 --warmup 5)"
  echo "K30 W5  $(run --steps 30 --warmup 5)"
done
