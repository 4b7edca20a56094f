timeout -k 10 200 python -m pytest tests/test_step_kernels_gpu.py tests/test_vqvae_gpu.py -q -rs 2>&1 | tail -6
