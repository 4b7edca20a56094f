import sys, time, torch
sys.path.insert(0, '/root/repo')
from melspec_gpt_vqvae_amd import ops
c = torch.zeros(1, dtype=torch.int32, device='cuda:0')
x = torch.zeros(256*256, device='cuda:0')
def body(n, kind):
    for _ in range(n):
        if kind == 0: ops.incr_i32(c)
        else: x.add_(1.0)
for kind in (0, 1):
    body(10, kind); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body(124, kind)
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): g.replay()
    torch.cuda.synchronize()
    print("kind", kind, "us per kernel in graph", (time.perf_counter() - t0) / 200 / 124 * 1e6)
    t0 = time.perf_counter()
    for _ in range(200): body(124, kind)
    torch.cuda.synchronize()
    print("kind", kind, "us per kernel eager", (time.perf_counter() - t0) / 200 / 124 * 1e6)
