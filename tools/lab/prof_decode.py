import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "tests", "golden"))
import torch, synth
from melspec_gpt_vqvae_amd.transformer.minGPT import Lit_minGPT, set_compute_dtype
args = synth.gpt_args(n_layer=24, n_head=16, n_embd=1024, reconstruct_spec="", device="cuda:0", batch_size=2, learning_rate=1e-6)
lit = Lit_minGPT(args).to("cuda:0").eval()
set_compute_dtype(lit.transformer, torch.bfloat16)
c = torch.randint(0, 8, (1, 1), device="cuda:0")
x0 = torch.zeros(1, 0, dtype=torch.int64, device="cuda:0")
lit.sample(x0, c, steps=64, sample=True, top_k=64)
torch.cuda.synchronize()
