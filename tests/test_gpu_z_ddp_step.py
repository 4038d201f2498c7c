"""the training step is data-parallel over the batch -- one process per GPU, DDP gradient all-reduce
of the CNN / head parameters, no collective inside the renderer.  Two ranks (gloo, sharing the one GPU of the test
box; on a node the same code runs with backend "nccl" = RCCL) each train on half of a batch with the alternate
optimisation of unicorn.py:226-230 (unused parameters every step); the result must equal single-process training on
the whole batch."""
import os
import sys
import tempfile

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STEPS, PER_RANK, IMG = 4, 2, 32


def _make(device):
    sys.path.insert(0, ROOT)
    from examples.train_step import TinyUnicorn, synthetic_targets
    torch.manual_seed(0)
    model = TinyUnicorn(IMG, level=2, faces_per_pixel=10, tex_size=16, latent=32).to(device)
    imgs = synthetic_targets(2 * PER_RANK, IMG, seed=5, device=device)
    return model, imgs


def _log_dir():
    """where the ranks' progress notes and stack dumps go: under gpurun_out/ on the GPU box (merged back after the
    call, so they survive a kill of the whole run), else the system's temporary directory"""
    root = os.environ.get("GRAFT_REPO_ROOT")
    d = os.path.join(root, "gpurun_out") if root and os.path.isdir(root) else tempfile.gettempdir()
    os.makedirs(d, exist_ok=True)
    return d


def _rank_main(rank, world, rdzv_path, out_path, log_dir):
    import datetime
    import faulthandler
    import time
    # (child only) c10d / gloo say what they are doing; every collective names itself if it gives up
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), TORCH_DISTRIBUTED_DEBUG="INFO", GLOO_LOG_LEVEL="INFO",
                      TORCH_CPP_LOG_LEVEL="INFO")
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from examples.train_step import train_steps
    from unicorn_amd import dist as udist
    # Everything a rank can say about where it stands goes to ONE file per rank that outlives the process: progress
    # notes as they happen (line buffered), this rank's stderr (c10d / gloo / MIOpen messages), and -- if the rank is
    # still busy after 60 s -- the stacks of all its threads (again every 20 s after that).
    log = open(os.path.join(log_dir, f"ddp_step_rank{rank}.log"), "w", buffering=1)
    os.dup2(log.fileno(), 2)
    faulthandler.dump_traceback_later(60, repeat=True, file=log, exit=False)
    t_start = time.perf_counter()

    def mark(what):
        log.write(f"[{time.perf_counter() - t_start:7.1f} s] rank {rank}: {what}\n")
    mark("started")
    udist.isolate_miopen_cache(rank)  # the ranks share the box: each gets its own MIOpen database / kernel cache
    udist.gloo_on_loopback()          # no host-name lookups for gloo's sockets (the box's name does not resolve)
    t0 = time.perf_counter()
    # rendezvous through a file in the test's own temporary directory (no store port another process could hold); a
    # collective that cannot complete RAISES after 45 s, naming itself, instead of sitting there until the parent
    # kills the rank
    dist.init_process_group("gloo", init_method="file://" + rdzv_path, rank=rank, world_size=world,
                            timeout=datetime.timedelta(seconds=45))
    mark("process group up")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    model, imgs = _make(dev)
    mark("model and target images on the GPU (first renderer call done)")
    assert model.pose_step is True  # both ranks start in the same phase of the alternate optimisation: the sets of
    #                                 unused parameters DDP sees agree across ranks in every iteration
    ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0], find_unused_parameters=True)
    mark("DDP wrapper built (parameter broadcast over gloo done)")
    losses = []
    for i in range(STEPS):
        losses += train_steps(ddp, imgs[rank * PER_RANK:(rank + 1) * PER_RANK], 1)
        torch.cuda.synchronize()
        mark(f"training step {i} done (loss {losses[-1]:.5f})")
    print(f"rank {rank}: {STEPS} DDP steps (MIOpen convolutions) after {time.perf_counter() - t0:.1f} s", flush=True)
    t = torch.tensor(losses, device=dev)
    dist.all_reduce(t)
    if rank == 0:
        torch.save({"params": {k: v.detach().cpu() for k, v in model.state_dict().items()},
                    "losses": (t / world).cpu()}, out_path)
    dist.barrier()
    mark("results saved, leaving")
    dist.destroy_process_group()
    faulthandler.cancel_dump_traceback_later()


def test_two_rank_ddp_step_equals_single_process_training():
    import torch.multiprocessing as mp
    from examples.train_step import train_steps
    with tempfile.TemporaryDirectory() as tmp:
        out_path = os.path.join(tmp, "rank0.pt")
        log_dir = _log_dir()
        ctx = mp.spawn(_rank_main, args=(2, os.path.join(tmp, "rendezvous"), out_path, log_dir), nprocs=2, join=False)
        for _ in range(100):                       # bounded wait: never hang the suite on a stuck rank
            if ctx.join(timeout=1.0):
                break
        else:
            for p in ctx.processes:
                p.kill()
            notes = []
            for r in range(2):
                try:
                    notes.append(f"rank {r}:\n" + open(os.path.join(log_dir, f"ddp_step_rank{r}.log")).read()[-6000:])
                except OSError:
                    notes.append(f"rank {r}: no progress note")
            pytest.fail(f"DDP ranks did not finish within 100 s (full notes: {log_dir}/ddp_step_rank*.log)\n"
                        + "\n".join(notes))
        # the same training in this process on the whole batch (after the ranks), MIOpen convolutions on both sides
        model, imgs = _make(torch.device("cuda", 0))
        losses = train_steps(model, imgs, STEPS)
        got = torch.load(out_path, weights_only=True)
    np.testing.assert_allclose(got["losses"].numpy(), np.array(losses), rtol=1e-4)
    assert losses[-1] < losses[0]
    ref = model.state_dict()
    for k, v in got["params"].items():
        a, b = v.numpy(), ref[k].detach().cpu().numpy()
        assert np.abs(a - b).max() <= 1e-4 * max(np.abs(b).max(), 1e-3), k
    # both halves of the alternate optimisation moved their own parameters
    torch.manual_seed(0)
    from examples.train_step import TinyUnicorn
    init = TinyUnicorn(IMG, level=2, faces_per_pixel=10, tex_size=16, latent=32).state_dict()
    assert not torch.equal(init["pose.weight"], got["params"]["pose.weight"])
    assert not torch.equal(init["texture.weight"], got["params"]["texture.weight"])
