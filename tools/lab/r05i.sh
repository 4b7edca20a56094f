mkdir -p gpurun_out/r05i
timeout -k 10 120 python -m pytest tests/test_rowops_attn_gpu.py -q -x -k "attention" 2>&1 | tail -1
timeout -k 10 200 python tools/lab/attn32_ab.py 2>&1 | tee gpurun_out/r05i/attn32_ab.jsonl | grep -v amdgpu
MELGPT_LAB_LIB=$PWD/tools/lab/bin/libmelgpt_clock.so CLOCK_ONLY="attention forward" timeout -k 10 400 python tools/lab/clock_lab.py > gpurun_out/r05i/clock_lab.jsonl 2> gpurun_out/r05i/clock_lab.err; cat gpurun_out/r05i/clock_lab.jsonl | cut -c1-330
