cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
T=gpurun_out/kt
rm -rf $T; mkdir -p $T
rocprofv3 --kernel-trace --output-format csv -d $T -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --eager --no-kernel-timer > $T/log 2>&1
f=$(find $T -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
d = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if "tapconv2" in n:
        d[n[:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    v5 = v[-16*1:] if "l1" in k else v[-36:]
    print(k, len(v), "last step launches (us):", " ".join(f"{x:.0f}" for x in v5))
PY
