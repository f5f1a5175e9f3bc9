#!/bin/bash
# Same-box A/B of the whole step: current library vs a variant library (default: the one-group layer-1 kernel,
# built with -DPH_L1_ONE_GROUP into multimodal-learning_amd/libpathomic_hip_oldl1.so).  Run through gpurun.
VAR=${1:-libpathomic_hip_oldl1.so}
cd multimodal-learning_amd
cp libpathomic_hip.so /tmp/lib_new.so
show='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], [(k["kernel"][:28], k["total_ms"]) for k in d["roofline"]["all_kernels"]])'
for r in 1 2; do
  cp /tmp/lib_new.so libpathomic_hip.so; echo "== current"
  (cd .. && python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "$show")
  cp $VAR libpathomic_hip.so; echo "== variant $VAR"
  (cd .. && python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "$show")
done
cp /tmp/lib_new.so libpathomic_hip.so
