#!/bin/bash
# One serial kernel-trace pass of the headline step -> gpurun_out/<tag>_kernel_stats.txt (top of the rocprofv3 stats table)
#     bash profiles/quick_stats.sh tag [extra bench args]
TAG=${1:-q}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
T=gpurun_out/$TAG
mkdir -p "$T"
rocprofv3 --kernel-trace --stats --output-format csv -d "$T/stats" -- python3 bench.py --serial --steps 10 --warmup 3 --no-cpu-baseline --no-parity-mode --no-variants --no-north-star-block "$@" > "$T/bench_stats.log" 2>&1
F=$(find "$T/stats" -name "*kernel_stats.csv" | head -1)
python3 - "$F" <<'PY' > gpurun_out/${TAG}_kernel_stats.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("# sum of kernel durations %.1f ms" % (tot / 1e6))
print("%-110s %7s %10s %9s %6s" % ("kernel", "calls", "total_ms", "avg_us", "pct"))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:45]:
    print("%-110s %7d %10.3f %9.2f %6.2f" % (r["Name"][:110], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
head -30 gpurun_out/${TAG}_kernel_stats.txt | cut -c1-160
