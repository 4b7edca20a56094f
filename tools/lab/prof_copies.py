import os, sys, collections
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests", "golden"))
import torch
import bench
from types import SimpleNamespace
a = SimpleNamespace(layers=24, batch=128)
dev = torch.device("cuda", 0)
job = bench.ClassGPTStep(a, dev, torch.bfloat16, 0, 1)
import time
for _ in range(2): job.step(time.perf_counter)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    job.step(time.perf_counter)
torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if e.name in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::contiguous", "aten::clone", "aten::_to_copy", "aten::cat", "aten::zeros", "aten::add_", "aten::mul", "aten::add"):
        st = [s for s in (e.stack or []) if "melspec_gpt_vqvae_amd" in s or "bench.py" in s]
        cnt[(e.name, str(e.input_shapes)[:60], st[0][-70:] if st else "?")] += 1
for k, v in cnt.most_common(40):
    print(v, k)
