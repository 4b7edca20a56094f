for st in 0 50 100 0 150 75; do MELGPT_GEMM_STAGGER=$st timeout -k 10 120 python tools/lab/epi_ab.py 2>&1 | grep -v amdgpu; done
