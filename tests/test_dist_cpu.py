"""CPU, world_size 2 on gloo: the N>1 plumbing of bench.py (rank-sharded synthetic batches, barrier-bracketed timing,
max over ranks, whole-job throughput).  The render kernels themselves need a GPU and are not called here."""
import os
import socket
import time

import torch
import torch.multiprocessing as mp

from unicorn_amd import dist as udist
from unicorn_amd import synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, local, w = udist.init("gloo")
    assert (r, w) == (rank, world)
    sc = synth.make_scene(2, level=1, tex_size=4, seed=udist.shard_seed(0, rank))
    calls = []

    def step():  # rank 1 is slower: the job is as slow as its slowest rank
        calls.append(1)
        time.sleep(0.01 * (1 + 2 * rank))

    elapsed = udist.timed_steps(step, steps=3, warmup=1, world=w, cuda=False)
    out[rank] = dict(elapsed=elapsed, calls=len(calls), R00=float(sc["R"][0, 0, 0]),
                     value=udist.job_throughput(2, 3, elapsed, w))
    udist.finish(w)


def test_two_rank_sharding_and_timing():
    world, port = 2, _free_port()
    out = mp.Manager().dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    a, b = out[0], out[1]
    assert a["calls"] == b["calls"] == 4                      # 1 warmup + 3 timed
    assert a["elapsed"] == b["elapsed"]                       # max over ranks, same on both
    assert a["elapsed"] >= 3 * 0.03                           # the slow rank's time
    assert abs(a["value"] - 2 * 2 * 3 / a["elapsed"]) < 1e-9  # whole-job units / max time
    assert a["R00"] != b["R00"]                               # ranks render different shards


def test_single_process_defaults():
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        os.environ.pop(k, None)
    assert udist.env_world() == (0, 0, 1)
    assert udist.max_over_ranks(1.5, 1, False) == 1.5
    assert udist.job_throughput(32, 10, 2.0, 1) == 160.0
