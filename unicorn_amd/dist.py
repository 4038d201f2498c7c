"""Data-parallel plumbing of the render benchmark / a DDP training step: one process per GPU, launched by
`torch.distributed.run` (reference: mp.spawn + NCCL DDP, trainer.py:260-264,284-290).  The renderer itself needs no
collective -- images are independent units, each rank renders its own shard of the batch -- so the only
communication here is the barrier / max-reduce that brackets a timed region."""
import os
import time

import torch
import torch.distributed as dist


def env_world():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def isolate_miopen_cache(rank):
    """Give this process its own MIOpen user database and kernel cache directory (unless the caller already chose
    them).  MIOpen keeps both under ~/.config/miopen and ~/.cache/miopen behind file locks; ranks that meet a new
    convolution shape at the same moment -- every rank of a DDP job does, on a box that has never run it -- then queue
    on those locks while each tunes / builds the same kernels.  Must run before the process's first convolution."""
    base = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"miopen-{os.getuid()}", f"rank{int(rank)}")
    for var, sub in (("MIOPEN_USER_DB_PATH", "db"), ("MIOPEN_CUSTOM_CACHE_DIR", "cache")):
        if var not in os.environ:
            path = os.path.join(base, sub)
            os.makedirs(path, exist_ok=True)
            os.environ[var] = path


def gloo_on_loopback():
    """Single-node gloo jobs (CPU tests, the one-GPU rehearsal of `bench.py --gpus N`): pin gloo's sockets to the
    loopback interface unless the caller chose one.  Left alone gloo looks its interface up through the HOST NAME, and
    a container's host name may not resolve (c10d warns "hostname of the client socket cannot be retrieved"): each
    failed lookup waits on the resolver.  Does nothing when `lo` does not exist.  RCCL is left alone."""
    if "GLOO_SOCKET_IFNAME" not in os.environ and os.path.isdir("/sys/class/net/lo"):
        os.environ["GLOO_SOCKET_IFNAME"] = "lo"


def init(backend=None):
    """-> (rank, local_rank, world).  backend: "nccl" (= RCCL on ROCm) on GPUs, "gloo" on CPU."""
    rank, local, world = env_world()
    if world > 1 and not dist.is_initialized():
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "gloo" and os.environ.get("MASTER_ADDR", "127.0.0.1") in ("127.0.0.1", "localhost", "::1"):
            gloo_on_loopback()
        kw = {"device_id": torch.device("cuda", local)} if backend == "nccl" else {}
        dist.init_process_group(backend, **kw)
    return rank, local, world


def shard_seed(base_seed, rank):
    """every rank draws its own synthetic shard (the reference seeds with seed + rank, trainer.py:263)"""
    return base_seed + 100 * rank


def fence(world, cuda):
    if cuda:
        torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    if cuda:
        torch.cuda.synchronize()


_last_local = [0.0]


def timed_steps(step, steps, warmup, world, cuda):
    """warmup untimed steps, then exactly `steps` timed ones between two fences; returns the MAX over ranks of the
    elapsed seconds (identical on every rank)."""
    for _ in range(warmup):
        step()
    fence(world, cuda)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    if cuda:
        torch.cuda.synchronize()
    _last_local[0] = time.perf_counter() - t0  # this rank's own time, before waiting for the others
    fence(world, cuda)
    return max_over_ranks(time.perf_counter() - t0, world, cuda)


def timed_blocks(step, steps, warmup, world, cuda, min_seconds=0.3, max_blocks=200):
    """`warmup` untimed steps, then BLOCKS of exactly `steps` timed steps, each block between two fences and timed as
    the MAX over ranks, repeated until the blocks add up to `min_seconds` (a region of a few milliseconds is at the
    mercy of one scheduler hiccup).  Returns (block times in seconds, this rank's own times), identical lengths on
    every rank: the stop decision is taken on the max-over-ranks times, which every rank holds."""
    for _ in range(warmup):
        step()
    blocks, local = [], []
    while True:
        fence(world, cuda)
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        if cuda:
            torch.cuda.synchronize()
        local.append(time.perf_counter() - t0)
        fence(world, cuda)
        blocks.append(max_over_ranks(time.perf_counter() - t0, world, cuda))
        if sum(blocks) >= min_seconds or len(blocks) >= max_blocks:
            return blocks, local


def median_block(blocks, local=None):
    """the median block's time (the lower middle one of an even count) and, when given, the same block's local time"""
    order = sorted(range(len(blocks)), key=lambda i: blocks[i])
    i = order[(len(order) - 1) // 2]
    return (blocks[i], local[i]) if local is not None else blocks[i]


def last_local_elapsed():
    """this rank's own elapsed seconds in the last timed_steps (without the wait for the slowest rank)"""
    return _last_local[0]


def gather_floats(value, world, cuda):
    """[value of rank 0, ..., value of rank world-1] on every rank"""
    if world == 1:
        return [float(value)]
    t = torch.tensor([value], dtype=torch.float64, device="cuda" if cuda else "cpu")
    out = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]


def sum_over_ranks(value, world, cuda):
    if world == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device="cuda" if cuda else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def max_over_ranks(value, world, cuda):
    if world == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device="cuda" if cuda else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def job_throughput(units_per_rank, steps, elapsed_max, world):
    """whole-job units/s: every rank processed units_per_rank * steps units in (at most) elapsed_max seconds"""
    return world * units_per_rank * steps / elapsed_max


def finish(world):
    if world > 1 and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
