timeout -k 10 300 python -m pytest tests/test_vqvae_gpu.py tests/test_race_screens_gpu.py -q -x -k "conv or vqvae or drain" 2>&1 | tail -2
for r in 1 2; do for m in 1 0; do echo "M16=$m: $(MELGPT_CONV_WS_M16=$m timeout -k 10 120 python tools/lab/convw_ab.py 2>&1 | tail -1)"; done; done
