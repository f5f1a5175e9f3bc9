#!/bin/bash
# HBM read bytes per launch of the layer-1 kernel (FETCH_SIZE x 2, the gfx950 correction of the guide), default library against
# libpathomic_hip<tag>.so on one box:   bash profiles/scripts/l1_traffic.sh _l1old
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=$1
OUT=gpurun_out/l1_traffic
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 2 --warmup 2 --no-cpu-baseline --no-parity-mode --no-variants --no-north-star-block --eager --no-kernel-timer"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/new -- python3 bench.py $ARGS > $OUT/new.log 2>&1
[ -n "$TAG" ] && PH_LIB_VARIANT=$TAG rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/old -- python3 bench.py $ARGS > $OUT/old.log 2>&1
python3 - <<PY
import csv, glob
for tag in ("new", "old"):
    fs = glob.glob("$OUT/%s/*/*counter_collection.csv" % tag)
    if not fs: continue
    rows = [r for r in csv.DictReader(open(fs[0])) if "tapconv2_l1" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
    v = [float(r["Counter_Value"]) * 2 * 1024 / 1e6 for r in rows]     # KiB, x 2: profiles/summarize.py
    v = v[-16:]
    print(tag, "layer-1 launches of the last step, read MB:", " ".join("%.0f" % x for x in v))
PY
