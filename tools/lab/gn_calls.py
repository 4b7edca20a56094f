import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
import bench
from melspec_gpt_vqvae_amd import ops
dev = torch.device("cuda", 0)
gpt, vqvae = bench.build_models(dev, torch.bfloat16, bench.vas_args())
x_mel, c = bench.synthetic_batch(16, 0, dev)
orig_gn, orig_stats = ops.groupnorm, ops.groupnorm_stats
def gn(x, *a, **k):
    print("groupnorm apply", tuple(x.shape), k.get("swish"))
    return orig_gn(x, *a, **k)
def st(x, *a, **k):
    print("groupnorm stats", tuple(x.shape))
    return orig_stats(x, *a, **k)
ops.groupnorm, ops.groupnorm_stats = gn, st
import melspec_gpt_vqvae_amd.vqvae.big_model_attn_gan as m
with torch.no_grad():
    vqvae.encode_to_codes(x_mel)
