#!/bin/bash
# gpurun helper: SQ counters of a micro-benchmark script ($1 = python file, rest = counters); prints per-kernel averages
PY=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf gpurun_out/pmcm && mkdir -p gpurun_out/pmcm
ONLY_MAIN=1 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmcm -- python3 $PY > gpurun_out/pmcm/log.txt 2>&1
python3 - <<'PY'
import csv,glob,collections
f=glob.glob("gpurun_out/pmcm/*/*counter_collection.csv")
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
for r in csv.DictReader(open(f[0])):
    agg[r["Kernel_Name"]][r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Kernel_Name"]].add(r["Dispatch_Id"])
for k,d in agg.items():
    if len(n[k])<5: continue
    print(k[:70], len(n[k]))
    for c,v in sorted(d.items()): print("   %-32s %.4g" % (c, v/len(n[k])))
PY
