mkdir -p gpurun_out/r05l
for i in 1 2; do
timeout -k 10 200 python bench.py --no-extras --no-cpu-baseline --steps 12 2>/dev/null > gpurun_out/r05l/bench_nodp_$i.json
MELGPT_BENCH_FORCE_DP=1 timeout -k 10 200 python bench.py --no-extras --no-cpu-baseline --steps 12 2>/dev/null > gpurun_out/r05l/bench_dp_$i.json
done
MELGPT_BENCH_FORCE_DP=1 MELGPT_RCCL_CHANNELS=16 timeout -k 10 200 python bench.py --no-extras --no-cpu-baseline --steps 12 2>/dev/null > gpurun_out/r05l/bench_dp_pin16.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05l/*.json')):
    o=json.loads(open(f).read().strip().splitlines()[-1]); c=o['config']
    print(f.split('/')[-1], o['ms_per_step'], c['gemm_launches_per_step'], c.get('dp_tiles'), c.get('reserved_cus'), c.get('rccl_channels_pinned'), o.get('exposed_comm_ms'))
PY
