"""A/B inside one job: melgpt_linear_skinny vs melgpt_gemv_rows for decode-shaped layers (graph of 48 calls)."""
import sys, time, torch
sys.path.insert(0, '/root/repo')
from melspec_gpt_vqvae_amd import ops
from melspec_gpt_vqvae_amd._ffi import call, ptr, stream, dtype_code

def run(fn, n=48, reps=50):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps / n * 1e6

for M in (8, 16, 32, 64, 128):
    for N, K in ((1024, 4096), (1024, 1024), (4096, 1024), (3072, 1024)):
        x = torch.randn(M, K, device='cuda:0').bfloat16(); w = (0.02 * torch.randn(N, K, device='cuda:0')).bfloat16()
        b = torch.zeros(N, device='cuda:0'); r = torch.randn(M, N, device='cuda:0').bfloat16(); y = torch.empty(M, N, device='cuda:0', dtype=torch.bfloat16)
        sk = lambda: call("melgpt_linear_skinny", ptr(x), K, ptr(w), K, ptr(b), ptr(r), N, ptr(y), N, M, N, K, 0, dtype_code(x.dtype), 0, None, None, 0.0, stream())
        gv = lambda: call("melgpt_gemv_rows", ptr(x), K, ptr(w), K, ptr(b), ptr(r), N, ptr(y), N, M, N, K, 0, dtype_code(x.dtype), 0, None, None, 0.0, stream())
        print(f"M={M:4d} N={N:5d} K={K:5d}  skinny {run(sk):6.2f} us   gemv_rows {run(gv):6.2f} us", flush=True)
