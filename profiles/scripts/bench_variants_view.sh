#!/bin/bash
# gpurun helper: default bench line (no parity-mode / CPU legs) and a readable dump of its `variants` block
python bench.py --no-parity-mode --no-cpu-baseline > gpurun_out/r03_a.json 2> gpurun_out/r03_a.err; tail -c 400 gpurun_out/r03_a.err
python - <<'PY'
import json
r=json.loads(open('gpurun_out/r03_a.json').read().strip().splitlines()[-1])
print(r['value'], r['ms_per_step'])
print(json.dumps(r.get('variants'), indent=1)[:3500])
PY
