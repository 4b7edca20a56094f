#!/usr/bin/env python3
"""Condense rocprofv3 outputs (kernel trace stats + FETCH_SIZE / WRITE_SIZE / MFMA-busy passes) into one per-kernel table.
usage: profile_summary.py <dir with trace/ fetch/ write/ mfma/> <tag>
Matrix-pipe busy fraction of a kernel = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs): the
numerator counts, per SIMD, the cycles an MFMA holds the matrix pipe (16 per v_mfma_f32_16x16x32_bf16, guide: cycle
constants), the denominator is the dispatch's shader-clock cycles (rocprofv3 sums GRBM_GUI_ACTIVE over the 8 XCDs) times
256 CUs x 4 SIMDs.  It is a fraction of the clock the chip actually HELD, so it reads higher than FLOP / 2.5 PF x time
whenever the chip ran below 2.4 GHz; both are reported.
HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE are in KiB;
on gfx950 FETCH_SIZE counts wide coalesced reads at half their bytes, so reads = 2 x FETCH_SIZE."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def find(d, pat):
    fs = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return fs[0] if fs else None


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    if name.endswith(")"):          # drop the argument list, keep the template arguments
        depth = 0
        for i in range(len(name) - 1, -1, -1):
            depth += name[i] == ")"
            depth -= name[i] == "("
            if depth == 0:
                name = name[:i]
                break
    for key in ("gemm8p_kernel", "gemm256_kernel", "gemm_kernel", "conv3x3_gn_kernel", "attn_q_kernel", "attn_dkv_kernel"):
        if key in name:
            # keep the template arguments: they tell layouts / modes apart
            i = name.index(key)
            return name[i:][:90]
    return name[-90:]


def main():
    root, tag = sys.argv[1], sys.argv[2]
    rows = defaultdict(lambda: dict(calls=0, ns=0.0, fetch_kib=0.0, write_kib=0.0, mfma_busy=0.0, gui_active=0.0,
                                    sq_busy=0.0))
    kt = find(os.path.join(root, "trace"), "*kernel_trace.csv")
    if kt:
        for r in csv.DictReader(open(kt)):
            k = short(r["Kernel_Name"])
            rows[k]["calls"] += 1
            rows[k]["ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    for sub, field in (("fetch", "fetch_kib"), ("write", "write_kib")):
        cc = find(os.path.join(root, sub), "*counter_collection.csv")
        if not cc:
            continue
        for r in csv.DictReader(open(cc)):
            rows[short(r["Kernel_Name"])][field] += float(r["Counter_Value"])
    cc = find(os.path.join(root, "mfma"), "*counter_collection.csv")
    have_mfma = False
    if cc:
        field = {"SQ_VALU_MFMA_BUSY_CYCLES": "mfma_busy", "GRBM_GUI_ACTIVE": "gui_active", "SQ_BUSY_CYCLES": "sq_busy"}
        for r in csv.DictReader(open(cc)):
            f = field.get(r.get("Counter_Name", ""))
            if f:
                rows[short(r["Kernel_Name"])][f] += float(r["Counter_Value"])
                have_mfma = True

    def busy(v):
        return v["mfma_busy"] / (v["gui_active"] / 8.0 * 1024.0) if v["gui_active"] > 0 else None

    tot_ns = sum(v["ns"] for v in rows.values()) or 1.0
    table = sorted(rows.items(), key=lambda kv: -kv[1]["ns"])
    out_csv = os.path.join(root, f"summary_{tag}.csv")
    with open(out_csv, "w") as f:
        f.write("kernel,calls,total_ms,avg_us,pct_time,hbm_read_MB(2xFETCH_SIZE),hbm_write_MB(WRITE_SIZE),"
                "mfma_busy_frac(SQ_VALU_MFMA_BUSY_CYCLES/(GRBM_GUI_ACTIVE/8*1024)),eff_clock_GHz(GRBM_GUI_ACTIVE/8/time)\n")
        for k, v in table:
            b = busy(v)
            bs = "" if b is None else "%.4f" % b
            clk = "" if v["gui_active"] <= 0 or v["ns"] <= 0 else "%.3f" % (v["gui_active"] / 8.0 / v["ns"])
            f.write(f"\"{k}\",{v['calls']},{v['ns'] / 1e6:.3f},{v['ns'] / 1e3 / max(v['calls'], 1):.2f},"
                    f"{100 * v['ns'] / tot_ns:.2f},{2 * v['fetch_kib'] / 1024:.1f},{v['write_kib'] / 1024:.1f},{bs},{clk}\n")
    gem = [v for k, v in table if "gemm" in k or "conv3x3_gn" in k]
    js = dict(tag=tag, gemm_family_ms=sum(v["ns"] for v in gem) / 1e6,
              gemm_family_hbm_read_MB=sum(2 * v["fetch_kib"] for v in gem) / 1024,
              gemm_family_hbm_write_MB=sum(v["write_kib"] for v in gem) / 1024, all_kernels_ms=tot_ns / 1e6,
              gemm_family_mfma_busy=(round(sum(v["mfma_busy"] for v in gem) / (sum(v["gui_active"] for v in gem) / 8.0 * 1024.0), 4)
                                     if have_mfma and sum(v["gui_active"] for v in gem) > 0 else None),
              mfma_busy_by_kernel=({k: round(busy(v), 4) for k, v in table
                                    if busy(v) is not None and busy(v) > 0.005 and v["ns"] > 0.002 * tot_ns}
                                   if have_mfma else None),
              workload=("gpt_vae_xl" if "gpt_vae_xl" in os.environ.get("PROFILE_CMD", "") else "class_gpt"),
              steps_profiled=int(os.environ.get("PROFILE_STEPS", "5")),  # tools/profile_round.sh: --steps 3 --warmup 2
              command="rocprofv3 --kernel-trace --stats / --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc SQ_VALU_MFMA_BUSY_CYCLES "
                      "SQ_BUSY_CYCLES GRBM_GUI_ACTIVE (four separate runs) -- " + os.environ.get(
                          "PROFILE_CMD", "python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline"),
              note="hbm_read = 2 x FETCH_SIZE (gfx950 counts wide coalesced reads at half their bytes), hbm_write = "
                   "WRITE_SIZE; KiB -> MB; GEMM family = gemm8p_kernel + gemm256_kernel + gemm_kernel + conv3x3_gn*_kernel")
    # attention launches of the step (HBM-bound at T = 265, hs = 64): counted bytes per launch, for bench.py's attn_frac_hbm
    att = {}
    for kind, keys in (("fwd", ("attn_q_kernel", "attn_fwd32_kernel")), ("bwd", ("attn_bwd1_kernel",))):
        sel = [v for k, v in table if any(x in k for x in keys) and v["calls"] > 0]
        calls = sum(v["calls"] for v in sel)
        if calls and sum(v["fetch_kib"] + v["write_kib"] for v in sel) > 0:
            att[kind] = round(sum(2 * v["fetch_kib"] + v["write_kib"] for v in sel) * 1024 / 1e6 / calls, 1)
            att[kind + "_avg_us"] = round(sum(v["ns"] for v in sel) / 1e3 / calls, 2)
    if att:
        js["attention_hbm_MB_per_launch"] = att
    # in-kernel clocks (tools/lab/clock_lab.py on the stamped diagnostic build of the same sources), when collected
    cl = os.path.join(root, f"clock_lab_{tag}.jsonl")
    if os.path.exists(cl):
        clocks = {}
        for ln in open(cl):
            try:
                r = json.loads(ln)
            except ValueError:
                continue
            if "in_kernel_clock_GHz" in r:
                clocks[r["kernel"]] = {"GHz": r["in_kernel_clock_GHz"], "TFLOPs": r.get("TFLOPs"),
                                       "frac_of_peak_at_that_clock": r.get("frac_of_peak_at_that_clock")}
        js["in_kernel_clock_GHz"] = clocks
        js["in_kernel_clock_source"] = ("d(s_memtime) / d(s_memrealtime) x 100 MHz per workgroup, median over workgroups, after 2 s of "
                                        "back-to-back launches on random data (tools/lab/clock_lab.py, -DMELGPT_CLOCK_STAMPS build)")
    json.dump(js, open(os.path.join(root, f"summary_{tag}.json"), "w"), indent=1)
    print(open(out_csv).read()[:4000])
    print(json.dumps(js))


if __name__ == "__main__":
    main()
