"""The fp16 flavour of the library (libmelgpt_hip_fp16.so: the same kernels with IEEE half as the 16-bit storage format,
csrc/common.h) - BASELINE configs[4] names fp16.  The format is a property of the process (MELGPT_HALF=fp16 before the
package is imported), so the checks run in a child process; this test asserts the figures it reports and records them."""
import json
import os
import subprocess
import sys

import pytest

from util import report

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_fp16_flavour_kernels_and_end_to_end_chain():
    env = dict(os.environ, MELGPT_HALF="fp16")
    r = subprocess.run([sys.executable, os.path.join(HERE, "fp16_lane_worker.py")], env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("FP16_LANE_RESULT ")][-1]
    o = json.loads(line[len("FP16_LANE_RESULT "):])
    report("fp16_lane", **{k: v for k, v in o.items() if not isinstance(v, (list, str))})
    assert o["lib"] == "libmelgpt_hip_fp16.so" and o["bf16_refused"]
    assert o["gemm_rel_err_f32_out"] < 1e-5                      # exact products, f32 accumulation
    assert o["gelu_epilogue_rel_err"] < 2 ** -9                  # one rounding to 11 significant bits
    assert o["attn_fwd_rel_err"] < 5e-3 and o["attn_bwd_rel_err"] < 1e-2
    assert o["gpt_logits_rel_err_vs_reference"] < 1e-2 and o["gpt_loss_abs_err_vs_reference"] < 5e-3
    assert o["gpt_loss_after_one_step"][1] < o["gpt_loss_after_one_step"][0]
    assert o["all_finite"] and o["chain_finite"] and o["chain_output_shape"] == [2, 1, 80, 848]
    assert o["vqvae_latent_rel_err_vs_reference"] < 2e-2 and o["vqvae_code_agreement_vs_reference"] > 0.95
    assert o["vqvae_rec_rel_err_vs_reference"] < 5e-2
    assert o["mel_tile_fp16_abs_err"] < 2e-3 and o["greedy16_token_agreement_vs_reference"] > 0.9
