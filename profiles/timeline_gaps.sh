#!/bin/bash
# Timeline analysis of one graph-replayed step (run through gpurun): device busy time (union of kernel intervals),
# idle gaps, and the kernels that precede the largest gaps.   bash profiles/timeline_gaps.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
T=gpurun_out/timeline
rm -rf $T; mkdir -p $T
rocprofv3 --kernel-trace --output-format csv -d $T -- python3 bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-kernel-timer > $T/log 2>&1
f=$(find $T -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
# the last replayed step = the kernels after the last occurrence of the first kernel of a step (stem pack_input)
idx = [i for i, r in enumerate(rows) if "adam_ema" in r[2]]
# a step ends with its last adam kernel; take the window between the 2nd-last and last step ends
ends = [rows[i][1] for i in idx]
last_end = ends[-1]
prev_end = max(e for e in ends if e < last_end - 5_000_000)
win = [r for r in rows if prev_end <= r[0] and r[1] <= last_end]
t0, t1 = prev_end, last_end
busy = 0; cur_s, cur_e = None, None
gaps = []
for s, e, n in win:
    if cur_e is None: cur_s, cur_e, last = s, e, n; gaps.append((s - t0, "(step start)")); continue
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append((s - cur_e, last)); cur_s, cur_e, last = s, e, n
    else:
        if e > cur_e: cur_e, last = e, n
busy += cur_e - cur_s
print(f"step window {(t1 - t0) / 1e6:.3f} ms, {len(win)} kernels, device busy (union) {busy / 1e6:.3f} ms, idle {(t1 - t0 - busy) / 1e6:.3f} ms in {len(gaps)} gaps")
print("sum of kernel durations %.3f ms" % (sum(e - s for s, e, n in win) / 1e6))
h = collections.Counter()
for g, n in gaps: h[min(int(g / 1000), 20)] += 1
print("gap histogram (us: count):", sorted(h.items()))
by = collections.defaultdict(lambda: [0, 0])
for g, n in gaps:
    by[n[:50]][0] += g; by[n[:50]][1] += 1
print("idle time by preceding kernel:")
for n, (g, c) in sorted(by.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f"   {g / 1e3:8.1f} us in {c:3d} gaps after {n}")
PY
