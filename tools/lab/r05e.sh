mkdir -p gpurun_out/r05e
timeout -k 10 120 python -m pytest tests/test_rowops_attn_gpu.py -q -x -k "attention" > gpurun_out/r05e/attn_tests.log 2>&1; tail -2 gpurun_out/r05e/attn_tests.log
for st in 0 6000 10000 14000 20000; do echo "stagger $st"; MELGPT_ATTN_STAGGER=$st timeout -k 10 200 python tools/lab/attn32_ab.py 2>&1 | head -2 | tee -a gpurun_out/r05e/attn32_ab_stagger$st.jsonl; done
MELGPT_LAB_LIB=$PWD/tools/lab/bin/libmelgpt_clock.so CLOCK_ONLY=attn timeout -k 10 400 python tools/lab/clock_lab.py > gpurun_out/r05e/clock_lab.jsonl 2> gpurun_out/r05e/clock_lab.err; grep attn gpurun_out/r05e/clock_lab.jsonl
