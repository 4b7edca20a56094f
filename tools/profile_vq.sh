#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel trace of the VQ lookup alone (tools/lab/vq_lab.py, bf16 lane) at
# B = 64 and B = 4096, plus FETCH_SIZE / WRITE_SIZE passes at B = 4096 (separate runs, as the guide prescribes).
# usage: tools/profile_vq.sh <tag> [image]     image: the prepared-image kernel (vq_image_kernel) instead of vq_bf16_kernel
set -u
TAG=${1:-rXX}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_vq
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export VQ_LAB_BF16_ONLY=1
export VQ_KERNEL=vq_bf16_kernel
if [ "${2:-}" = "image" ]; then export VQ_LAB_IMAGE=1 VQ_KERNEL="vq_image_kernel<false, false>"; fi
for B in 64 4096; do
  timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace$B" -o trace --output-format csv -- python3 $REPO/tools/lab/vq_lab.py $B > "$OUT/trace$B.log" 2>&1 || exit 1
done
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/fetch" -o fetch --output-format csv -- python3 $REPO/tools/lab/vq_lab.py 4096 > "$OUT/fetch.log" 2>&1 || exit 1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$OUT/write" -o write --output-format csv -- python3 $REPO/tools/lab/vq_lab.py 4096 > "$OUT/write.log" 2>&1 || exit 1
cd "$REPO"
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, json, os, sys
out, tag = sys.argv[1], sys.argv[2]
KERNEL = os.environ.get("VQ_KERNEL", "vq_bf16_kernel")
res = {"tag": tag, "command": "rocprofv3 --kernel-trace --stats / --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate runs) -- python3 tools/lab/vq_lab.py <B>"}
for B in (64, 4096):
    f = glob.glob(f"{out}/trace{B}/**/*kernel_stats.csv", recursive=True)[0]
    for r in csv.DictReader(open(f)):
        if KERNEL in r["Name"]:
            res[f"B{B}"] = {"kernel": r["Name"][:80], "calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                            "min_us": float(r["MinNs"]) / 1e3, "max_us": float(r["MaxNs"]) / 1e3}
    open(f"{out}/{tag}_vq_kernel_stats_B{B}.csv", "w").write(open(f).read())
for name in ("fetch", "write"):
    f = glob.glob(f"{out}/{name}/**/*counter_collection.csv", recursive=True)[0]
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if KERNEL in r["Kernel_Name"]]
    res[f"{name.upper()}_SIZE_KiB_per_launch"] = sum(vals) / max(1, len(vals))
    res[f"{name}_launches"] = len(vals)
n = 4096 * 265
res["algorithmic_bytes_per_launch_B4096"] = n * 520 + (131072 if KERNEL == "vq_bf16_kernel" else 66048)
res["kernel_filter"] = KERNEL
if "FETCH_SIZE_KiB_per_launch" in res:
    res["hbm_read_bytes_per_launch_B4096(2xFETCH_SIZE)"] = 2 * 1024 * res["FETCH_SIZE_KiB_per_launch"]
    res["hbm_write_bytes_per_launch_B4096"] = 1024 * res["WRITE_SIZE_KiB_per_launch"]
json.dump(res, open(f"{out}/{tag}_vq_summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
find "$OUT" -name "*.csv" -size +2M -delete
