mkdir -p gpurun_out/r05f
MELGPT_LAB_LIB=$PWD/tools/lab/bin/libmelgpt_clock.so CLOCK_ONLY="attention forward" timeout -k 10 400 python tools/lab/clock_lab.py > gpurun_out/r05f/clock_lab.jsonl 2> gpurun_out/r05f/clock_lab.err; cat gpurun_out/r05f/clock_lab.jsonl; tail -3 gpurun_out/r05f/clock_lab.err
