"""Child rank of tests/test_dp_gpu.py (test infrastructure): several ranks share cuda:0, rendezvous over gloo, run one
forward/backward of a small model through dp.DataParallel and dump the reduced flat gradient.
usage: python tests/dp_worker.py <gptclass|vae> <outdir>     (RANK / WORLD_SIZE / MASTER_* in the environment)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)

import torch
import torch.distributed as dist

import dp_models


def main():
    which, outdir = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)          # before anything touches the GPU
    torch.cuda.set_device(0)
    from melspec_gpt_vqvae_amd.dp import DataParallel

    model, batch, loss_fn = dp_models.build(which, "cuda:0")
    from melspec_gpt_vqvae_amd import _ffi

    # the window in which RCCL kernels may share the chip (claimed tiles + reserved CUs on) is what an RCCL run gets by
    # default; asked for explicitly here because this rendezvous is gloo
    dp = DataParallel(model, dynamic_tiles=True, reserve_cus=8)
    L = _ffi.lib()
    window = [(L.melgpt_get_dynamic_tiles(), L.melgpt_get_reserved_cus())]        # before the backward pass: (0, 0)
    assert dp.world == world and len(dp.blocks) == (2 if which == "gptclass" else 4)
    n = dp_models.BATCH // world
    local = {k: v[rank * n:(rank + 1) * n] for k, v in batch.items()}
    loss = loss_fn(model, local)
    loss.backward()
    launched_early = len(dp.ex._done)
    window.append((L.melgpt_get_dynamic_tiles(), L.melgpt_get_reserved_cus()))    # after the first hook: (1, 8)
    dp.finish()
    window.append((L.melgpt_get_dynamic_tiles(), L.melgpt_get_reserved_cus()))    # after finish(): (0, 0)
    torch.cuda.synchronize()
    m = dp.reduce_metrics(loss, float(rank), 3.0)
    torch.save({"grad": dp.fp.grad.cpu(), "hook_calls": dp.hook_calls, "launched_early": launched_early,
                "loss": float(loss), "metrics": [float(v) for v in m], "names": dp.fp.names, "window": window,
                "offsets": dp.fp.offsets}, os.path.join(outdir, f"rank{rank}.pt"))
    # a second backward without finish() in between must be refused (partial sums would be reduced twice)
    refused = False
    loss2 = loss_fn(model, local)
    loss2.backward()
    try:
        loss3 = loss_fn(model, local)
        loss3.backward()
    except RuntimeError as e:
        refused = "launched twice" in str(e)
    for w in dp.ex._works:
        w.wait()
    dp.ex._works, dp.ex._done = [], []
    dist.barrier()
    torch.save({"refused": refused}, os.path.join(outdir, f"rank{rank}_guard.pt"))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
