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
    n_before = len(calls)
    # bench.py's loop: blocks of exactly `steps` steps until they add up to min_seconds; both ranks must stop together
    blocks, local = udist.timed_blocks(step, steps=2, warmup=0, world=w, cuda=False, min_seconds=0.25)
    out[rank] = dict(elapsed=elapsed, calls=n_before, R00=float(sc["R"][0, 0, 0]), blocks=blocks, block_local=local,
                     block_calls=len(calls) - n_before, median=udist.median_block(blocks, local),
                     value=udist.job_throughput(2, 3, elapsed, w), local=udist.last_local_elapsed(),
                     rates=udist.gather_floats(10.0 + rank, w, cuda=False), ones=udist.sum_over_ranks(1.0, w, False))
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
    assert a["local"] < b["local"] <= b["elapsed"]            # each rank's own time, before the wait for the slowest
    assert a["rates"] == b["rates"] == [10.0, 11.0] and a["ones"] == b["ones"] == 2.0  # per-rank line of bench.py
    # repeated blocks: same count and same (max-over-ranks) times on both ranks, at least min_seconds in total,
    # every block exactly 2 steps of the slow rank, the median one reported
    assert a["blocks"] == b["blocks"] and len(a["blocks"]) >= 2 and sum(a["blocks"]) >= 0.25
    assert a["block_calls"] == b["block_calls"] == 2 * len(a["blocks"])
    assert all(t >= 2 * 0.03 for t in a["blocks"]) and sum(a["blocks"][:-1]) < 0.25
    assert a["median"][0] == sorted(a["blocks"])[(len(a["blocks"]) - 1) // 2] == b["median"][0]
    assert a["median"][1] < b["median"][1] <= b["median"][0]


def test_median_block_picks_the_lower_middle():
    assert udist.median_block([3.0, 1.0, 2.0]) == 2.0
    assert udist.median_block([4.0, 1.0, 3.0, 2.0]) == 2.0
    assert udist.median_block([5.0, 1.0], [50.0, 10.0]) == (1.0, 10.0)


def test_bench_starts_its_own_ranks(monkeypatch):
    """`python bench.py --gpus N` without RANK in the environment (how the driver may call it) must start N ranks
    itself (reference: trainer.py:284-290 spawns its ranks from the entry script) -- as a CHILD process, before
    anything touches the GPU -- and pass the child's exit code on."""
    import importlib
    import subprocess
    import sys
    bench = importlib.import_module("bench")
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "3"])
    assert bench.launch_own_ranks(2) == 7
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=2" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "2", "--steps", "3"]
    assert cmd[cmd.index("--master-port") + 2].endswith("bench.py")
    assert seen["env"].get("UNICORN_BENCH_BACKEND") == "gloo"  # no GPUs here: the ranks would share devices
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    with __import__("pytest").raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7


def test_single_process_defaults():
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        os.environ.pop(k, None)
    assert udist.env_world() == (0, 0, 1)
    assert udist.max_over_ranks(1.5, 1, False) == 1.5
    assert udist.job_throughput(32, 10, 2.0, 1) == 160.0
