#!/usr/bin/env python3
"""Condense rocprofv3 outputs (kernel trace stats + FETCH_SIZE / WRITE_SIZE passes) into one per-kernel table.
usage: profile_summary.py <dir with trace/ fetch/ write/> <tag>
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
    for key in ("gemm256_kernel", "gemm_kernel", "conv3x3_gn_kernel", "attn_q_kernel", "attn_dkv_kernel"):
        if key in name:
            # keep the template arguments: they tell layouts / modes apart
            i = name.index(key)
            return name[i:][:90]
    return name[-90:]


def main():
    root, tag = sys.argv[1], sys.argv[2]
    rows = defaultdict(lambda: dict(calls=0, ns=0.0, fetch_kib=0.0, write_kib=0.0))
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
    tot_ns = sum(v["ns"] for v in rows.values()) or 1.0
    table = sorted(rows.items(), key=lambda kv: -kv[1]["ns"])
    out_csv = os.path.join(root, f"summary_{tag}.csv")
    with open(out_csv, "w") as f:
        f.write("kernel,calls,total_ms,avg_us,pct_time,hbm_read_MB(2xFETCH_SIZE),hbm_write_MB(WRITE_SIZE)\n")
        for k, v in table:
            f.write(f"\"{k}\",{v['calls']},{v['ns'] / 1e6:.3f},{v['ns'] / 1e3 / max(v['calls'], 1):.2f},"
                    f"{100 * v['ns'] / tot_ns:.2f},{2 * v['fetch_kib'] / 1024:.1f},{v['write_kib'] / 1024:.1f}\n")
    gem = [v for k, v in table if "gemm" in k or "conv3x3_gn" in k]
    js = dict(tag=tag, gemm_family_ms=sum(v["ns"] for v in gem) / 1e6,
              gemm_family_hbm_read_MB=sum(2 * v["fetch_kib"] for v in gem) / 1024,
              gemm_family_hbm_write_MB=sum(v["write_kib"] for v in gem) / 1024, all_kernels_ms=tot_ns / 1e6,
              steps_profiled=int(os.environ.get("PROFILE_STEPS", "5")),  # tools/profile_round.sh: --steps 3 --warmup 2
              command="rocprofv3 --kernel-trace --stats / --pmc FETCH_SIZE / --pmc WRITE_SIZE (three separate runs) -- "
                      "python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline",
              note="hbm_read = 2 x FETCH_SIZE (gfx950 counts wide coalesced reads at half their bytes), hbm_write = "
                   "WRITE_SIZE; KiB -> MB; GEMM family = gemm256_kernel + gemm_kernel + conv3x3_gn*_kernel")
    json.dump(js, open(os.path.join(root, f"summary_{tag}.json"), "w"), indent=1)
    print(open(out_csv).read()[:4000])
    print(json.dumps(js))


if __name__ == "__main__":
    main()
