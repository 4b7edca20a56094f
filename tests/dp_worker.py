"""Child rank of tests/test_dp_gpu.py (test infrastructure): several ranks share cuda:0, rendezvous over gloo, run one
forward/backward of a small model through dp.DataParallel and dump the reduced flat gradient.  DP_BACKEND=nccl: ONE rank
over a real RCCL process group (RCCL refuses two ranks on one device) with MELGPT_DP_FORCE_EXCHANGE=1, so that the
all-reduces are really issued on RCCL's stream beside the backward GEMMs, with DataParallel's RCCL defaults.
usage: python tests/dp_worker.py <gptclass|vae|gptclass_vas16> <outdir>     (RANK / WORLD_SIZE / MASTER_* in the environment)"""
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
    backend = os.environ.get("DP_BACKEND", "gloo")
    if backend == "nccl":
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)      # before anything touches the GPU
        torch.cuda.set_device(0)
    from melspec_gpt_vqvae_amd.dp import DataParallel

    model, batch, loss_fn = dp_models.build(which, "cuda:0")
    from melspec_gpt_vqvae_amd import _ffi

    # the window in which RCCL kernels may share the chip (reserved CUs on): nothing is reserved by default over RCCL
    # (checked below); asked for explicitly when this rendezvous is gloo
    if backend == "nccl":
        dp = DataParallel(model)      # no reserved CUs (no channel pin in this child): they change the weight gradients' split-K factor, hence the bits
        assert dp.ex.active and dp.reserve_cus == 0, "over RCCL the exchange is live on the whole chip by default"
    else:
        dp = DataParallel(model, reserve_cus=8)
    L = _ffi.lib()
    window = [L.melgpt_get_reserved_cus()]        # before the backward pass: 0
    assert dp.world == world and len(dp.blocks) == (4 if which == "vae" else 2)
    n = next(iter(batch.values())).shape[0] // world
    local = {k: v[rank * n:(rank + 1) * n] for k, v in batch.items()}
    if os.environ.get("DP_MISMATCH") == "1":
        # at step DP_MISMATCH_STEP rank 1 ends its backward with one parameter (outside the Blocks) WITHOUT a gradient,
        # rank 0 with all of them: finish() must refuse on BOTH ranks at the SAME step - that very step when it is the
        # first (checked synchronously), exactly one step later otherwise (the verdict rides a pinned flag one step old);
        # no host synchronisation between the steps, as in the training loop
        at = int(os.environ.get("DP_MISMATCH_STEP", "1"))
        refused_at = None
        for step in range(1, at + 3):
            for p in model.parameters():
                p.grad = None
            dp.fp.zero_grad()
            loss = loss_fn(model, local)
            loss.backward()
            if rank == 1 and step == at:
                victim = [p for nm, p in model.named_parameters() if nm.endswith("pos_emb") or nm.endswith("head.weight")][0]
                victim.grad = None
            try:
                dp.finish()
            except RuntimeError as e:
                if "disagree on which parameters received a gradient" in str(e):
                    refused_at = step
                    break
                raise
        torch.cuda.synchronize()
        dist.barrier()
        torch.save({"refused": refused_at is not None, "refused_at": refused_at},
                   os.path.join(outdir, f"rank{rank}_mismatch.pt"))
        dist.destroy_process_group()
        return
    loss = loss_fn(model, local)
    loss.backward()
    launched_early = len(dp.ex._done)
    window.append(L.melgpt_get_reserved_cus())    # after the first hook: 8 (gloo rig) / 0 (RCCL default)
    dp.finish()
    window.append(L.melgpt_get_reserved_cus())    # after finish(): 0
    torch.cuda.synchronize()
    m = dp.reduce_metrics(loss, float(rank), 3.0)
    torch.save({"grad": dp.fp.grad.cpu(), "hook_calls": dp.hook_calls, "launched_early": launched_early,
                "loss": float(loss), "metrics": [float(v) for v in m], "names": dp.fp.names, "window": window,
                "offsets": dp.fp.offsets, "backend": dist.get_backend(), "n_works": len(dp.ex._done)}, os.path.join(outdir, f"rank{rank}.pt"))
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
