#!/bin/bash
# round 6, first call: same-box baseline + trunk-only overlap A/B at B=256 and B=64 + where the D2D copies come from
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_first; mkdir -p $O
for i in 1 2; do
  PH_TRUNK_NO_OVERLAP=0 python bench.py --trunk-only --steps 8 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('trunk256 overlap', d['ms_per_step'])"
  PH_TRUNK_NO_OVERLAP=1 python bench.py --trunk-only --steps 8 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('trunk256 serial ', d['ms_per_step'])"
done
for i in 1 2; do
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timer --no-parity-mode --no-variants --no-north-star-block 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('step64 overlap', d['ms_per_step'])"
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timer --no-parity-mode --no-variants --no-north-star-block --no-bwd-overlap 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('step64 nobwdovl', d['ms_per_step'])"
  python bench.py --steps 10 --warmup 5 --north-star --no-cpu-baseline --no-kernel-timer --no-parity-mode --no-variants --no-north-star-block 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('step256 overlap', d['ms_per_step'])"
  python bench.py --steps 10 --warmup 5 --north-star --no-cpu-baseline --no-kernel-timer --no-parity-mode --no-variants --no-north-star-block --no-bwd-overlap 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('step256 nobwdovl', d['ms_per_step'])"
done
python bench.py --steps 20 --warmup 5 --precision fp16x3/x1 --no-cpu-baseline --no-kernel-timer --no-parity-mode --no-variants --no-north-star-block 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp16x3/x1', d['ms_per_step'])"
python tests/prof_copies_gpu.py > $O/copies.txt 2>&1; tail -40 $O/copies.txt
