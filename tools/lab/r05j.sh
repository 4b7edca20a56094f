mkdir -p gpurun_out/r05j
timeout -k 10 300 python -m pytest tests/test_vqvae_gpu.py -q -x 2>&1 | tail -2
for r in 1 2; do for m in 1 0; do echo "M16=$m: $(MELGPT_CONV_WS_M16=$m timeout -k 10 120 python tools/lab/convw_ab.py 2>&1 | tail -1)"; done; done | tee gpurun_out/r05j/convw_ab.txt
for m in 1 0; do MELGPT_CONV_WS_M16=$m timeout -k 10 200 python bench.py --no-extras --no-cpu-baseline --steps 10 2>/dev/null | python -c "
import sys,json
o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('M16=$m', o['value'], o['ms_per_step'], [(r['shape'],r['ms_per_step'],r['tflops']) for r in o['roofline']['per_shape'] if 'conv3x3+gn' in r['shape']], o.get('config2_vq_encode',{}).get('ms'))
"; done | tee gpurun_out/r05j/bench_ab.txt
