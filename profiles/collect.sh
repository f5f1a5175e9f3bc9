#!/bin/bash
# Collect the rocprofv3 evidence behind bench.py's roofline block (run on the GPU box through gpurun):
#     bash profiles/collect.sh r01           # -> gpurun_out/r01/{stats,sq,fetch,write}
# then, back in the repo:  python profiles/summarize.py gpurun_out/r01 r01
# Kernel trace + stats in one run; the PMC counters in their own runs (never combined with a trace domain), FETCH_SIZE
# and WRITE_SIZE in separate passes as /opt/skills/guides/MI355X_MICROARCH.md prescribes.
set -u
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
T=gpurun_out/$TAG
mkdir -p "$T"
# per-kernel durations: the step on ONE stream (--serial).  In the product step three forwards run on three streams and the
# weight gradients beside the BatchNorm-backward passes, and a launch's wall duration includes the time it shares the chip;
# that run is traced too (stats_concurrent) - its total is the honest one, its per-kernel averages are not kernel quality.
rocprofv3 --kernel-trace --stats --output-format csv -d "$T/stats" -- python3 bench.py --serial --steps 10 --warmup 3 --no-cpu-baseline --no-parity-mode --no-variants --no-north-star-block > "$T/bench_stats.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$T/stats_concurrent" -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity-mode --no-variants --no-north-star-block > "$T/bench_stats_concurrent.log" 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 \
  --output-format csv -d "$T/sq" -- python3 bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-parity-mode --no-variants --no-north-star-block --eager --no-kernel-timer > "$T/bench_sq.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$T/fetch" -- python3 bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-parity-mode --no-variants --no-north-star-block --eager --no-kernel-timer > "$T/bench_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$T/write" -- python3 bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-parity-mode --no-variants --no-north-star-block --eager --no-kernel-timer > "$T/bench_write.log" 2>&1
# L2 view of the same step (do the re-reads of the layer-1 kernel's halos reach the fabric?): hits / misses / fabric read requests
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum --output-format csv -d "$T/tcc" -- python3 bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-parity-mode --no-variants --no-north-star-block --eager --no-kernel-timer > "$T/bench_tcc.log" 2>&1
# CRD kernels at the bank sizes of BASELINE configs[3] / configs[4] (counter bytes for the `variants` block of the bench line)
for V in mia2022 mia2023; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$T/fetch_$V" -- python3 bench.py --variant $V --steps 3 --warmup 3 --eager > "$T/bench_fetch_$V.log" 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$T/write_$V" -- python3 bench.py --variant $V --steps 3 --warmup 3 --eager > "$T/bench_write_$V.log" 2>&1
done
# the north-star's single-GPU point (batch 256): per-kernel durations and counter traffic of its own launches
rocprofv3 --kernel-trace --stats --output-format csv -d "$T/stats_b256" -- python3 bench.py --north-star --serial --steps 5 --warmup 3 --no-cpu-baseline --no-parity-mode --no-variants --no-north-star-block > "$T/bench_stats_b256.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$T/fetch_b256" -- python3 bench.py --north-star --steps 2 --warmup 3 --no-cpu-baseline --no-parity-mode --no-variants --no-north-star-block --eager --no-kernel-timer > "$T/bench_fetch_b256.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$T/write_b256" -- python3 bench.py --north-star --steps 2 --warmup 3 --no-cpu-baseline --no-parity-mode --no-variants --no-north-star-block --eager --no-kernel-timer > "$T/bench_write_b256.log" 2>&1
# the north-star's literal quantity: the student's ResNet forward + backward alone at batch 256 (bench.py --trunk-only)
rocprofv3 --kernel-trace --stats --output-format csv -d "$T/stats_trunk_b256" -- python3 bench.py --trunk-only --steps 5 > "$T/bench_stats_trunk_b256.log" 2>&1
# the tolerance-compliant arithmetic (fp16x3/x1): where its step spends the time; the MIA-2023 leg with its full-bank KNN kernels
rocprofv3 --kernel-trace --stats --output-format csv -d "$T/stats_fp16x3" -- python3 bench.py --precision fp16x3/x1 --serial --steps 5 --warmup 3 --no-cpu-baseline --no-parity-mode --no-variants --no-north-star-block > "$T/bench_stats_fp16x3.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$T/stats_mia2023" -- python3 bench.py --variant mia2023 --steps 5 --warmup 3 --eager > "$T/bench_stats_mia2023.log" 2>&1
python3 bench.py > "$T/bench_default.log" 2>&1
tail -1 "$T/bench_default.log" | cut -c1-300
python3 bench.py --north-star --no-parity-mode --no-cpu-baseline --no-variants > "$T/bench_b256.log" 2>&1
tail -1 "$T/bench_b256.log" | cut -c1-200
