# Reliability check of the data-parallel code path on one GPU: `bench.py --force-dist` (a one-rank RCCL group, collectives captured
# inside the step graph) eight times; every run must print its JSON line.  bash profiles/scripts/fd_loop.sh
ok=0; bad=0
for i in 1 2 3 4 5 6 7 8; do
  python bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-kernel-timer --no-parity-mode --no-variants --no-north-star-block --force-dist > /tmp/fd_$i.log 2>&1
  if grep -q '^{' /tmp/fd_$i.log; then ok=$((ok+1)); else bad=$((bad+1)); grep -m2 -i "error\|what()" /tmp/fd_$i.log | cut -c1-200; fi
done
echo "force-dist runs ok=$ok bad=$bad"
