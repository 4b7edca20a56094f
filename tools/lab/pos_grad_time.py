import sys, torch
sys.path.insert(0, '.')
from melspec_gpt_vqvae_amd import ops
B,T,C=128,266,1024
g=torch.Generator(device='cuda').manual_seed(1)
dx=(torch.randn(B,T,C,device='cuda',generator=g)).to(torch.bfloat16)
idx=torch.zeros(B,T,dtype=torch.long,device='cuda')
pg=torch.zeros(T,C,device='cuda')
def f(): ops.embed_bwd(dx, idx, pos_grad=pg, drop_p=0.5, seed=3, stream_id=1)
for _ in range(3): f()
s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20): f()
e.record(); torch.cuda.synchronize()
print("embed_bwd pos us", s.elapsed_time(e)/20*1e3)
ops.embed_bwd(dx, idx, pos_grad=pg)
ref=dx.float().sum(0)
print("max err", float((pg-ref).abs().max()), float(ref.abs().max()))
