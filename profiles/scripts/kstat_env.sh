#!/bin/bash
# per-kernel durations of one kernel-name pattern under several environment arms, one box:
#     bash profiles/scripts/kstat_env.sh "<grep pattern>" "<bench args>" "ENV=.." "ENV=.." ...
PAT="$1"; ARGS="$2"; shift 2
i=0
for arm in "$@"; do
  i=$((i+1))
  env $arm bash profiles/quick_stats.sh ks$i $ARGS > /dev/null 2>&1
  echo "== $arm"; grep -h "$PAT" gpurun_out/ks${i}_kernel_stats.txt | cut -c1-70,100-150
done
