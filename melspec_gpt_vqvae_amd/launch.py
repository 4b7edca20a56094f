"""One process per GPU, started from a plain `python script.py --gpus N`: the place where Lightning's DDP launcher
stands in the reference (GPT_VAE_train.py:166-190 - `pl.Trainer(devices=args.gpus, strategy="ddp...")` re-runs the script
once per GPU with the rank in the environment).  `spawn_ranks` starts N FRESH children of the same script with
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, passes their output through, and returns the worst
exit status.  It must be called before the parent has touched the GPU (a process that has initialised HIP must neither
fork GPU work nor be replaced): nothing here calls into torch.cuda except device_count(), which does not initialise it.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import time


def launched_by_a_launcher() -> bool:
    """True inside a rank (torch.distributed.run or spawn_ranks set WORLD_SIZE)."""
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def visible_gpus() -> int:
    import torch

    return torch.cuda.device_count()   # counts devices without creating a HIP context on this image


def spawn_ranks(argv, nproc: int, *, share_gpu: bool = False, env_extra: dict | None = None, timeout: float | None = None) -> int:
    """Run `python argv...` as `nproc` ranks on this node.  share_gpu: every rank uses device 0 (debug aid on a 1-GPU
    box; the ranks then rendezvous over gloo - RCCL refuses two ranks on one device).  Rank 0 inherits stdout, the other
    ranks' stdout goes to stderr (a bench prints its one JSON line on rank 0).  Returns the largest exit status; when a
    rank fails the others are terminated (by PID) instead of waiting for a rendezvous that cannot complete."""
    if nproc < 1:
        raise ValueError("nproc must be >= 1")
    if not share_gpu:
        have = visible_gpus()
        if have < nproc:
            raise SystemExit(f"--gpus {nproc}: this node shows {have} GPU(s) (set MELGPT_BENCH_SHARE_GPU=1 to let the "
                             "ranks share cuda:0 over gloo - a control-flow rehearsal, not a measurement)")
    port = free_port()
    procs = []
    for r in range(nproc):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(nproc), LOCAL_WORLD_SIZE=str(nproc),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.update(env_extra or {})
        procs.append(subprocess.Popen([sys.executable] + list(argv), env=env, stdout=None if r == 0 else sys.stderr))
    t0 = time.monotonic()
    rcs: list = [None] * nproc
    try:
        while any(rc is None for rc in rcs):
            for i, p in enumerate(procs):
                if rcs[i] is None:
                    rcs[i] = p.poll()
            failed = [rc for rc in rcs if rc not in (None, 0)]
            timed_out = timeout is not None and time.monotonic() - t0 > timeout
            if failed or timed_out:
                for i, p in enumerate(procs):          # the ranks still running wait for a peer that is gone
                    if rcs[i] is None:
                        p.terminate()
                for i, p in enumerate(procs):
                    if rcs[i] is None:
                        try:
                            p.wait(timeout=20)
                        except subprocess.TimeoutExpired:
                            p.kill()
                            p.wait()
                        rcs[i] = 0                     # ended by this launcher: not that rank's own verdict
                if timed_out and not failed:
                    return 124
                break
            time.sleep(0.2)
    except KeyboardInterrupt:
        for p in procs:
            p.terminate()
        raise
    return max((abs(rc) for rc in rcs if rc), default=0)
