#!/bin/bash
# Same-box A/B of a bench.py switch (box-to-box spread is +-5 %: never compare numbers from different gpurun calls):
#     bash profiles/ab.sh --no-fuse          # alternates `bench.py <switch>` and `bench.py` three times
SW="$@"
for i in 1 2 3; do
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timer --no-parity-mode --no-variants $SW 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('with    $SW', d['ms_per_step'])"
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timer --no-parity-mode --no-variants 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('without $SW', d['ms_per_step'])"
done
