#!/bin/bash
# gpurun helper: rocprofv3 kernel stats of a short bench run -> gpurun_out/$1/ and the top of the table to stdout
TAG=${1:-ks}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity-mode --no-variants --no-kernel-timer "$@" > gpurun_out/$TAG/bench.log 2>&1
tail -1 gpurun_out/$TAG/bench.log | cut -c1-200
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/$TAG/stats/*/*kernel_stats.csv")[0]
rows=list(csv.DictReader(open(f)))
tot=sum(int(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot/1e6)
for r in rows[:40]:
    print(f"{r['Name'][:90]:<90s} {r['Calls']:>6s} {int(r['TotalDurationNs'])/1e6:9.3f} {float(r['AverageNs'])/1e3:9.2f} {float(r['Percentage']):6.2f}")
PY
