mkdir -p gpurun_out/r05k; export MELGPT_REPORT_DIR=$PWD/gpurun_out/r05k
timeout -k 10 800 python -m pytest tests -m gpu -q --durations=6 > gpurun_out/r05k/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r05k/tests.log; tail -12 gpurun_out/r05k/tests.log
timeout -k 10 400 python bench.py > gpurun_out/r05k/bench.json 2> gpurun_out/r05k/bench.err; echo bench rc=$?; tail -c 300 gpurun_out/r05k/bench.json
