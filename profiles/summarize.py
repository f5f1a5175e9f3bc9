#!/usr/bin/env python3
"""Turn the rocprofv3 CSVs that `gpurun` merged into gpurun_out/<run>/ into the compact summaries committed
under profiles/ (kernel stats table, PMC table, per-launch HBM traffic of the MFMA kernels).

    python profiles/summarize.py gpurun_out/r01 r01

HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE come from separate
--pmc passes; both are in KiB; on gfx950 FETCH_SIZE reports exactly half of the bytes of wide coalesced streaming
reads, so it is doubled; WRITE_SIZE is exact for 16-B-per-lane stores.
"""
import collections
import csv
import glob


def newest(pattern):
    """files matching `pattern`, newest first (a second collection into one tag leaves the earlier run's files beside it)"""
    return sorted(glob.glob(pattern), key=os.path.getmtime, reverse=True)
import json
import os
import sys


def short(k):
    for a, b in (("(anonymous namespace)::", ""), ("void ", ""), ("_ZN12_GLOBAL__N_1", "")):
        k = k.replace(a, b)
    return k


def cls_of(name):
    """kernel name -> class index of ph_prof_summary / bench.py CLS_NAMES"""
    n = name
    if "tapconv3_kernel" in n or "tapconv7_kernel" in n:      # dense 3x3 kernels (conv_tap3.hip; round 6: conv_tap7.hip for its plain perf-mode form): class 6
        return 6
    if "tapconv5_kernel" in n:      # half-pair layer-1 kernel (conv_tap5.hip, round 6): the layer-1 class
        return 7
    if "tapconv6_kernel" in n:      # half-pair stride-2 forward kernel (conv_tap6.hip, round 6): the stride-2 class
        return 2
    if "tapconv2_l1_kernel" in n or "tapconv4_kernel" in n:   # layer 1 (Cin = Cout = 64): conv_tap4.hip (round 5), before it the two-group kernel
        return 7
    if "tapconv2_kernel" in n:   # second-generation 3x3 stride-1 kernel, Cout >= 128; <..., true>: masked stride-2 grid
        if "false, true" in n or "Lb0ELb1E" in n:
            return 12
        return 7 if ("4, 1, 2" in n or "Li4ELi1ELi2E" in n) else 6
    if "tapconv_kernel" in n:
        if "Li2ELi8ELi128" in n or ", 2, 8, 128" in n:
            return 2
        if "16, 64, 4, 1" in n or "Li16ELi64ELi4ELi1" in n:
            return 0
        if "16, 128, 4, 2" in n or "Li16ELi128ELi4ELi2" in n:
            return 1
        return None
    if "stem_wgrad_kernel" in n:
        return 5
    if "stem_fwd_kernel" in n:
        return 4
    if "wgrad_kernel" in n and "reduce" not in n:
        return 3
    return None


dur = {}     # kernel name -> summed duration (ns) of its profiled dispatches (last load_pmc call)


def load_pmc(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(set)
    seen = set()
    dur.clear()
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k].add(r["Dispatch_Id"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            dur[k] = dur.get(k, 0) + int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return agg, {k: len(v) for k, v in cnt.items()}


def main():
    src, tag = sys.argv[1], sys.argv[2]
    out = os.path.dirname(os.path.abspath(__file__))
    # ---- kernel stats: the one-stream run (per-kernel durations) and the product run (three forward streams, weight
    # gradients beside the BatchNorm-backward passes: a launch's wall duration includes time it shares the chip)
    for sub, suffix, cmd in (("stats", "", "python3 bench.py --serial --steps 10 --warmup 3 --no-cpu-baseline --no-parity-mode --no-variants"),
                             ("stats_concurrent", "_concurrent", "python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity-mode --no-variants"),
                             ("stats_b256", "_b256", "python3 bench.py --north-star --serial --steps 5 --warmup 3 ... (batch 256 on one GPU)"),
                             ("stats_trunk_b256", "_trunk_b256", "python3 bench.py --trunk-only --steps 5 (the north-star's literal quantity: the student's ResNet forward + backward alone at batch 256, one captured HIP graph)"),
                             ("stats_fp16x3", "_fp16x3", "python3 bench.py --precision fp16x3/x1 --serial --steps 5 --warmup 3 ... (the tolerance-compliant arithmetic)"),
                             ("stats_mia2023", "_mia2023", "python3 bench.py --variant mia2023 --steps 5 --warmup 3 --eager (BASELINE configs[4] single-GPU leg: the full-bank KNN kernels)")):
        f = newest(os.path.join(src, sub, "*", "*kernel_stats.csv"))
        if not f:
            continue
        rows = list(csv.DictReader(open(f[0])))
        tot = sum(int(r["TotalDurationNs"]) for r in rows)
        with open(os.path.join(out, f"{tag}_kernel_stats{suffix}.txt"), "w") as o:
            o.write(f"# rocprofv3 --kernel-trace --stats -- {cmd}   (sum of kernel durations {tot/1e6:.1f} ms"
                    + (": kernels overlap, durations include shared time" if suffix else "") + ")\n")
            o.write(f"{'kernel':<100s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'pct':>6s}\n")
            for r in rows[:(70 if suffix == "_mia2023" else 45)]:
                o.write(f"{short(r['Name'])[:100]:<100s} {r['Calls']:>6s} {int(r['TotalDurationNs'])/1e6:10.3f} "
                        f"{float(r['AverageNs'])/1e3:10.2f} {float(r['Percentage']):6.2f}\n")
    # ---- SQ counters
    f = newest(os.path.join(src, "sq", "*", "*counter_collection.csv"))
    if f:
        agg, cnt = load_pmc(f[0])
        with open(os.path.join(out, f"{tag}_pmc_sq.txt"), "w") as o:
            o.write("# rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES "
                    "SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 (eager bench, 6 steps)\n")
            # MFMA utilisation at the ACTUAL shader clock: SQ_VALU_MFMA_BUSY_CYCLES sums the busy cycles of all 1024 SIMDs
            # (32 per v_mfma_f32_32x32x16_bf16: the tapconv2 launch of 2 359 296 MFMAs reads 75 497 472), SQ_BUSY_CYCLES sums
            # the kernel's duration in shader cycles over the 32 shader engines (81.7 us x 1.87 GHz x 32 = 4.89e6), so the
            # fraction of matrix-pipe cycles in use is mfma_busy / (busy / 32 * 1024).  (r01 divided by 4 instead of 32:
            # its column was 8x too large.)
            o.write(f"{'kernel':<80s} {'n':>5s} {'lds_conflict/lds_active':>24s} {'wait_any/wave_cycles':>22s} {'mfma_util':>10s} {'clock_GHz':>10s}\n")
            tot_m = tot_b = 0.0
            for k in sorted(agg, key=lambda k: -agg[k].get("SQ_WAVE_CYCLES", 0))[:14]:
                d = agg[k]
                o.write(f"{short(k)[:80]:<80s} {cnt[k]:5d} "
                        f"{d.get('SQ_LDS_BANK_CONFLICT',0)/max(d.get('SQ_LDS_IDX_ACTIVE',0),1):24.3f} "
                        f"{d.get('SQ_WAIT_ANY',0)/max(d.get('SQ_WAVE_CYCLES',0),1):22.3f} "
                        f"{d.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/max(d.get('SQ_BUSY_CYCLES',0),1)/32:10.3f} "
                        f"{d.get('SQ_BUSY_CYCLES',0)/32/max(dur.get(k,0),1):10.2f}\n")
            for k, d in agg.items():
                tot_m += d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0); tot_b += d.get("SQ_BUSY_CYCLES", 0)
            o.write(f"# all kernels of the profiled steps: MFMA busy {tot_m:.4g} SIMD-cycles / ({tot_b:.4g} / 32 x 1024) = "
                    f"{tot_m / max(tot_b, 1) / 32:.3f} of the matrix-pipe cycles while a kernel runs\n")
    # ---- HBM traffic per launch of the MFMA kernel classes
    ff = newest(os.path.join(src, "fetch", "*", "*counter_collection.csv"))
    fw = newest(os.path.join(src, "write", "*", "*counter_collection.csv"))
    if ff and fw:
        fa, fc = load_pmc(ff[0])
        wa, wc = load_pmc(fw[0])
        traffic = {}
        lines = []
        per = collections.defaultdict(lambda: [0.0, 0.0, 0, 0])
        for k in fa:
            c = cls_of(k)
            if c is not None:
                per[c][0] += fa[k].get("FETCH_SIZE", 0.0); per[c][2] += fc[k]
        for k in wa:
            c = cls_of(k)
            if c is not None:
                per[c][1] += wa[k].get("WRITE_SIZE", 0.0); per[c][3] += wc[k]
        for c, (fe, wr, nf, nw) in sorted(per.items()):
            rd = 2.0 * fe * 1024 / max(nf, 1)        # gfx950: FETCH_SIZE counts 64 B per 128-B request
            ww = wr * 1024 / max(nw, 1)
            traffic[str(c)] = {"read_bytes_per_launch": round(rd), "write_bytes_per_launch": round(ww),
                               "bytes_per_launch": round(rd + ww), "launches_sampled": nf}
            lines.append(f"class {c}: read {rd/1e6:9.2f} MB  write {ww/1e6:9.2f} MB  per launch ({nf} launches)")
        # whole-step total over ALL kernels of the profiled run (bench.py --steps 3 --warmup 3 --eager: 6 steps; the few
        # initialisation kernels in front of them are negligible beside 6 x ~35 GB)
        nsteps = 6
        tot_r = sum(2.0 * d.get("FETCH_SIZE", 0.0) * 1024 for d in fa.values())
        tot_w = sum(d.get("WRITE_SIZE", 0.0) * 1024 for d in wa.values())
        traffic["step_total_bytes"] = round((tot_r + tot_w) / nsteps)
        traffic["step_read_bytes"] = round(tot_r / nsteps)
        traffic["step_write_bytes"] = round(tot_w / nsteps)
        lines.append(f"whole step (all kernels / {nsteps} steps): read {tot_r/nsteps/1e9:.2f} GB  write {tot_w/nsteps/1e9:.2f} GB")
        json.dump(traffic, open(os.path.join(out, f"{tag}_traffic.json"), "w"), indent=1)
        open(os.path.join(out, f"{tag}_traffic.txt"), "w").write(
            "# HBM traffic per launch (FETCH_SIZE x2 correction, WRITE_SIZE exact; separate --pmc passes)\n" + "\n".join(lines) + "\n")
    # ---- the same at batch 256 (the north-star's single-GPU point): profiles/<tag>_b256_traffic.json
    ff = newest(os.path.join(src, "fetch_b256", "*", "*counter_collection.csv"))
    fw = newest(os.path.join(src, "write_b256", "*", "*counter_collection.csv"))
    if ff and fw:
        fa, fc = load_pmc(ff[0])
        wa, wc = load_pmc(fw[0])
        traffic = {}
        per = collections.defaultdict(lambda: [0.0, 0.0, 0, 0])
        for k in fa:
            c = cls_of(k)
            if c is not None:
                per[c][0] += fa[k].get("FETCH_SIZE", 0.0); per[c][2] += fc[k]
        for k in wa:
            c = cls_of(k)
            if c is not None:
                per[c][1] += wa[k].get("WRITE_SIZE", 0.0); per[c][3] += wc[k]
        for c, (fe, wr, nf, nw) in sorted(per.items()):
            rd = 2.0 * fe * 1024 / max(nf, 1)
            ww = wr * 1024 / max(nw, 1)
            traffic[str(c)] = {"read_bytes_per_launch": round(rd), "write_bytes_per_launch": round(ww),
                               "bytes_per_launch": round(rd + ww), "launches_sampled": nf}
        json.dump(traffic, open(os.path.join(out, f"{tag}_b256_traffic.json"), "w"), indent=1)
    # ---- L2 (TCC) view per MFMA kernel class: hit rate and fabric read requests
    ft = newest(os.path.join(src, "tcc", "*", "*counter_collection.csv"))
    if ft:
        ta, tc = load_pmc(ft[0])
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        for k in ta:
            c = cls_of(k)
            if c is not None:
                for name, v in ta[k].items():
                    per[c][name] += v
                per[c]["n"] += tc[k]
        with open(os.path.join(out, f"{tag}_tcc.txt"), "w") as o:
            o.write("# rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum (eager bench); per launch of each kernel class\n")
            o.write("# l2_hit = HIT / (HIT + MISS) over 128-B request granules; rdreq = read requests the L2 sent to the fabric "
                    "(Infinity Cache / HBM); rdreq_dram = those addressed to device memory\n")
            for c, d in sorted(per.items()):
                n = max(d["n"], 1)
                o.write(f"class {c}: l2_hit {d['TCC_HIT_sum'] / max(d['TCC_HIT_sum'] + d['TCC_MISS_sum'], 1):.3f}  "
                        f"hit {d['TCC_HIT_sum'] / n:12.0f}  miss {d['TCC_MISS_sum'] / n:12.0f}  rdreq {d['TCC_EA0_RDREQ_sum'] / n:12.0f}  "
                        f"rdreq_dram {d['TCC_EA0_RDREQ_DRAM_sum'] / n:12.0f}  ({int(n)} launches)\n")
    # ---- CRD kernels of the variant steps (bench.py --variant mia2022 / mia2023): counter bytes per CALL of each kernel family
    crd = {}
    for var in ("mia2022", "mia2023"):
        ff = newest(os.path.join(src, f"fetch_{var}", "*", "*counter_collection.csv"))
        fw = newest(os.path.join(src, f"write_{var}", "*", "*counter_collection.csv"))
        if not (ff and fw):
            continue
        fa, fc = load_pmc(ff[0])
        fdur = dict(dur)
        wa, wc = load_pmc(fw[0])
        fam = {"crd_score": ("crd_score_kernel",), "crd_loss_grad": ("crd_loss_grad_kernel", "crd_loss_grad_reduce_kernel"),
               "crd_bank_topk": ("crd_bank_knn_kernel", "crd_knn_thr_kernel", "crd_knn_merge_kernel")}
        crd[var] = {}
        for name, kerns in fam.items():
            rd = wr = t = 0.0
            calls = 0
            for k in fa:
                if any(q in k for q in kerns):
                    rd += 2.0 * fa[k].get("FETCH_SIZE", 0.0) * 1024
                    t += fdur.get(k, 0)
                    if kerns[-1] in k:       # (a kernel launched once per call of the family)
                        calls = fc[k]
            for k in wa:
                if any(q in k for q in kerns):
                    wr += wa[k].get("WRITE_SIZE", 0.0) * 1024
            if calls:
                crd[var][name] = {"read_bytes_per_launch": round(rd / calls), "write_bytes_per_launch": round(wr / calls),
                                  "bytes_per_launch": round((rd + wr) / calls), "launches_sampled": calls,
                                  "avg_us_under_profiler": round(t / calls / 1e3, 2)}
    if crd:
        json.dump(crd, open(os.path.join(out, f"{tag}_crd_traffic.json"), "w"), indent=1)
    print("wrote", sorted(os.listdir(out)))


if __name__ == "__main__":
    main()
