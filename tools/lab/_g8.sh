timeout -k 10 400 python -m pytest tests/test_gemm8p_gpu.py tests/test_gemm_gpu.py tests/test_race_screens_gpu.py -q -x 2>&1 | tail -3
timeout -k 10 120 python tools/lab/epi_ab.py 2>&1 | grep -v amdgpu
export MELGPT_LAB_LIB=$PWD/tools/lab/bin/libmelgpt_p8lab.so
for cfg in "33920,4096,1024 0 plain" "33920,4096,1024 0 gelu_dact" "33920,1024,1024 0 drop_res" "33920,4096,1024 1 mul"; do set -- $cfg; SHAPE=$1 B_KMAJOR=$2 MODE=$3 timeout -k 10 60 python tools/lab/p8_stamps.py 2>&1 | grep -v amdgpu | head -4 | cut -c1-140; done
