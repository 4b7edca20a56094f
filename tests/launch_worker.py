"""Child rank of tests/test_host_cpu.py::test_spawn_ranks_* (test infrastructure): a world-size-N gloo rendezvous from
the environment melspec_gpt_vqvae_amd.launch.spawn_ranks sets, one all-reduce, one line on stdout from rank 0.
usage: python tests/launch_worker.py [--fail-rank R]"""
import os
import sys

import torch
import torch.distributed as dist


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if "--fail-rank" in sys.argv and rank == int(sys.argv[sys.argv.index("--fail-rank") + 1]):
        raise SystemExit(7)                      # before the rendezvous: the other ranks would wait for ever
    dist.init_process_group("gloo", rank=rank, world_size=world)
    v = torch.tensor([float(rank + 1)])
    dist.all_reduce(v)
    if rank == 0:
        print(f"LAUNCH_OK world={world} local_rank={os.environ['LOCAL_RANK']} sum={v.item():.0f}", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
