"""dp.DataParallel with world = 2 on ONE GPU: two fresh child processes (started before they touch the GPU) share
cuda:0 and rendezvous over gloo - the same wrapper, Block hooks, segments and finish() that run over RCCL on an
8-GPU node; only the transport differs.  Checked: the reduced flat gradient x 1/world == the single-process gradient
of the concatenated batch (1e-6), the Block hooks fired (early launches happened), the fused metric all-reduce, and
the double-backward guard.  Covers a GPTClass and a GPT_VAE (two transformers in one flat buffer - the model the
reference trains under DDP, GPT_VAE_train.py:166-190)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

import dp_models

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("which", ["gptclass", "vae"])
def test_two_ranks_share_one_gpu_gradients_equal_single_process(which, tmp_path):
    from melspec_gpt_vqvae_amd.flat import ensure_flat

    world, port = 2, _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "dp_worker.py"), which, str(tmp_path)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    # the single-process answer, computed here while the children run
    model, batch, loss_fn = dp_models.build(which, "cuda:0")
    loss = loss_fn(model, batch)
    loss.backward()
    fp = ensure_flat(model)
    want = fp.grad.cpu()
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    res = [torch.load(os.path.join(str(tmp_path), f"rank{r}.pt")) for r in range(world)]
    n_blocks = 2 if which == "gptclass" else 4
    for r in res:
        assert r["names"] == fp.names and r["offsets"] == fp.offsets
        assert r["hook_calls"] == n_blocks and r["launched_early"] == 2 * n_blocks     # two slices per Block, before finish()
        # the CU reservation is on exactly while an all-reduce can be in flight (first hook .. finish())
        assert list(r["window"]) == [0, 8, 0]
        got = r["grad"] / world                                                        # the optimizer's grad_scale
        err = float((got - want).abs().max() / want.abs().max())
        assert err < 1e-6, err
        mean_loss = sum(x["loss"] for x in res) / world
        assert abs(r["metrics"][0] - mean_loss) <= 2e-7 * max(1.0, abs(mean_loss))         # (the VAE loss is ~1.3e3: one f32 ulp is 1.2e-4)
        assert r["metrics"][1] == 0.5 and r["metrics"][2] == 3.0                      # mean of the ranks' values
    assert torch.equal(res[0]["grad"], res[1]["grad"])                                 # both ranks hold the same sum
    mean_loss = sum(x["loss"] for x in res) / world
    assert abs(mean_loss - float(loss)) <= 1e-6 * max(1.0, abs(float(loss)))           # (two half batches against one full batch)
    for rnk in range(world):
        assert torch.load(os.path.join(str(tmp_path), f"rank{rnk}_guard.pt"))["refused"]


def test_one_rank_over_a_real_rccl_group_bit_identical_to_no_dp(tmp_path):
    """The `nccl` backend (= RCCL) itself: a child rank initialises a real RCCL process group (size 1 - RCCL does not
    allow two ranks on one device), forces the exchange on (MELGPT_DP_FORCE_EXCHANGE=1) and runs the 16-bit-lane backward of
    a VAS-width model through DataParallel with its RCCL defaults: every Block's two slices are all-reduced on RCCL's
    stream while the earlier Blocks' GEMMs are still running (the same ping-pong GEMM on static tile lists as a single GPU).
    The flat gradient must equal the no-DP run of this process bit for bit and the hooks fire
    (no reserved CUs here: a smaller grid changes the weight gradients' split-K factor and with it the summation order)."""
    from melspec_gpt_vqvae_amd.flat import ensure_flat

    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY="0", DP_BACKEND="nccl", MELGPT_DP_FORCE_EXCHANGE="1")
    env.pop("NCCL_MAX_NCHANNELS", None)
    env.pop("MELGPT_RESERVE_CUS", None)
    proc = subprocess.Popen([sys.executable, os.path.join(HERE, "dp_worker.py"), "gptclass_vas16", str(tmp_path)], env=env,
                            stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    model, batch, loss_fn = dp_models.build("gptclass_vas16", "cuda:0")
    loss = loss_fn(model, batch)
    loss.backward()
    fp = ensure_flat(model)
    torch.cuda.synchronize()
    want = fp.grad.cpu()
    out = proc.communicate(timeout=900)[0]
    assert proc.returncode == 0, out
    r = torch.load(os.path.join(str(tmp_path), "rank0.pt"))
    assert r["backend"] == "nccl"
    assert r["hook_calls"] == 2 and r["launched_early"] == 4
    assert list(r["window"]) == [0, 0, 0]
    assert r["names"] == fp.names and r["offsets"] == fp.offsets
    assert float(loss) == r["loss"]
    assert torch.equal(r["grad"], want), float((r["grad"] - want).abs().max())
    assert torch.load(os.path.join(str(tmp_path), "rank0_guard.pt"))["refused"]


def _run_child(argv, extra_env, timeout):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", MELGPT_BENCH_SHARE_GPU="1", **extra_env)
    r = subprocess.run([sys.executable] + argv, env=env, capture_output=True, text=True, timeout=timeout,
                       cwd=os.path.dirname(HERE))
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, lines


def test_bench_self_launches_two_ranks_and_prints_one_self_describing_line():
    """The multi-rank rehearsal a 1-GPU box allows, in the driver-run suite: `python bench.py --gpus 2` (no launcher
    around it) starts its two ranks itself (launch.spawn_ranks) before touching the GPU; under MELGPT_BENCH_SHARE_GPU=1
    they share cuda:0 and rendezvous over gloo.  Checked: exit status 0, exactly ONE JSON line (rank 0's), the whole-job
    fields (n_gpus, global_batch = 2 x per-GPU batch, parallelism dp2), the run flagged INVALID_debug_shared_gpu, and the
    fields a SCALE record needs to explain itself (exchange bytes / format, the reserved-CU switch,
    exposed_comm_ms); then the same with the 16-bit wire format."""
    import json

    base = [os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "2", "--layers", "2", "--steps", "2", "--warmup",
            "1", "--batch", "32", "--no-cpu-baseline", "--no-extras"]
    for wire, nbytes_per_param in (("f32", 4), ("bf16", 2)):
        r, lines = _run_child(base + ["--grad-dtype", wire], {}, 900)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        assert len(lines) == 1, r.stdout
        o = json.loads(lines[0])
        c = o["config"]
        assert o["n_gpus"] == 2 and o["scaling"] == "weak" and o["steps"] == 2 and o["value"] > 0
        assert c["batch_per_gpu"] == 32 and c["global_batch"] == 64 and c["parallelism"] == "dp2"
        assert c["INVALID_debug_shared_gpu"] is True and c["INVALID_debug_layers"] == 2
        assert abs(o["value"] - 64 / (o["ms_per_step"] * 1e-3)) <= 1e-3 * o["value"]
        assert c["exchange_dtype"] == ("float32" if wire == "f32" else "bfloat16") and c["backend"] == "gloo"
        assert c["exchange_bytes"] % nbytes_per_param == 0 and c["exchange_bytes"] // nbytes_per_param > 25_000_000
        assert c["dp_tiles"] == "static" and c["reserved_cus"] == 0 and c["overlap"] is True
        assert o["exposed_comm_ms"] >= 0.0 and np.isfinite(c["final_loss"])


def test_bench_e2e_self_launches_two_ranks_clips_sharded_no_collective():
    """tools/bench_e2e.py --gpus 2 under MELGPT_BENCH_SHARE_GPU=1: two ranks, clips r::2, no collective on the data path;
    the parent prints one line per batch size with the ranks' clips/s summed."""
    import json

    r, lines = _run_child([os.path.join(os.path.dirname(HERE), "tools", "bench_e2e.py"), "--gpus", "2", "--batches", "1"],
                          {}, 900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert len(lines) == 1, r.stdout
    o = json.loads(lines[0])
    assert o["n_gpus"] == 2 and o["batch_per_gpu"] == 1 and o["INVALID_debug_shared_gpu"] is True
    assert len(o["clips_per_s_by_rank"]) == 2 and abs(sum(o["clips_per_s_by_rank"]) - o["clips_per_s"]) < 0.02
    assert o["clip_ids_by_rank"] == [[0, 0], [1, 1]]


@pytest.mark.parametrize("at", [1, 3])
def test_ranks_that_disagree_on_which_parameters_got_a_gradient_are_refused(tmp_path, at):
    """One rank ends step `at` with a parameter that got no gradient, the other with all of them: torch's DDP - the
    reference - raises; here finish() raises on BOTH ranks at the SAME step instead of letting the replicas diverge - step
    1 is checked synchronously, a later step's verdict is read (waited for, not polled) at the next finish(), so a
    mismatch that first appears at step 3 is refused at step 4 on every rank even when the host runs a step ahead."""
    world, port = 2, _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0", DP_MISMATCH="1", DP_MISMATCH_STEP=str(at))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "dp_worker.py"), "gptclass", str(tmp_path)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    for rnk in range(world):
        o = torch.load(os.path.join(str(tmp_path), f"rank{rnk}_mismatch.pt"))
        assert o["refused"] and o["refused_at"] == (1 if at == 1 else at + 1), o
