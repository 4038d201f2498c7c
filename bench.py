"""bench.py -- rendered images/s, forward + backward, through the `Renderer` drop-in.

Workload (BASELINE.json metric): B=32 textured level-3 icospheres (642 V / 1280 F) per GPU,
128x128, faces_per_pixel K=50, sigma=1e-4, 64x64 UV textures, FoV-30 camera, fixed random cotangent; gradients to
vertices, textures, R and T.  One "step" = one Renderer.forward + one backward of that batch.  Inputs live in HBM
before the timed region.  With N>1 GPUs every rank renders its own batch (weak scaling, no collective on the data
path -- images are independent); the timed region is bracketed by barrier + synchronize and the max over ranks is used.

`python bench.py --gpus N` with N > 1 and no RANK in the environment starts its own N ranks (a child
`python -m torch.distributed.run`, the reference's trainer.py:284-290 spawns its ranks from the entry script too); under
an external torch.distributed.run it just joins.  On a box with fewer GPUs than ranks the ranks share the GPUs over
gloo (a rehearsal of the launch path; the line says so).

Timing: after the warmup, blocks of EXACTLY --steps steps, each between barrier + synchronize fences and timed as the
max over ranks, are repeated until they add up to 0.3 s; `ms_per_step` / `value` are the MEDIAN block's (a single block
of 20 steps is 2 ms: one scheduler hiccup would move it by several percent).  `value` is always the hipGraph replay of
the captured step (one fixed loop, not the better of two); the eager loop's rate is a side field.

Prints ONE JSON line.  `roofline` prices the dominant kernel against the VALU-issue roof -- these kernels are per-pixel
geometry tests, sorting and scatter, not streaming -- with the kernel time measured live (HIP events on
the launch stream), the per-launch instruction counts from the committed rocprofv3 --pmc passes (profiles/
r03_counters.json, stamped with the hash of the kernel sources they were taken on: a stale file nulls the fraction) and
the roof from the committed micro-benchmark of the issue rate (profiles/r03_valu_issue.json, tools/micro/valu_issue.hip);
`path_hbm` keeps SURVEY 8(d)'s byte model beside it; `cpu_baseline` times the oracle's CPU restatement (oracle/, test
infrastructure) on the host cores for a bounded sample of the same workload.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from unicorn_amd import dist as udist  # noqa: E402
from unicorn_amd import native, synth  # noqa: E402
from unicorn_amd.renderer import Renderer  # noqa: E402
from unicorn_amd.structures import Meshes, TexturesUV, repeat  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is what a float4 copy achieves
# VALU issue roof: 256 CUs x 4 SIMDs at the 2.4 GHz nominal shader clock (MI355X_MICROARCH.md), one wave64 VALU
# instruction per CLK_PER_WAVE_VALU clocks per SIMD.  That constant is MEASURED (tools/micro/valu_issue.hip, result
# committed as profiles/r03_valu_issue.json): independent v_add_f32 / v_mul_f32 / v_add_u32 streams issue every 4.63
# clocks from one wave alone and every 2.31 clocks once two or more waves share the SIMD (the guide's SIMD-32 figure;
# round 2 priced the kernels at 4).  It is the roof of the CHEAPEST instructions: v_min_f32 / v_max_f32 / v_mad_u32_u24
# and the packed-f32 forms stay at ~4.6 clocks however many waves there are, transcendentals at 8.6.
SIMDS, CLOCK_GHZ = 1024, 2.4
VALU_ISSUE = os.path.join(ROOT, "profiles", "r03_valu_issue.json")
COUNTERS = os.path.join(ROOT, "profiles", "r03_counters.json")


def valu_issue_clocks():
    """clocks per wave64 VALU instruction per SIMD, from the committed micro-benchmark: the best sustained rate of a
    plain full-rate instruction (v_add_f32) with the SIMD shared by two waves"""
    try:
        rows = json.load(open(VALU_ISSUE))["results"]
        return min(r["clk_per_instr_simd"] for r in rows if r["op"] == "v_add_f32" and r["waves_per_simd"] == 2)
    except (OSError, KeyError, ValueError):
        return 2.314  # the committed measurement's value (file not shipped)


CLK_PER_WAVE_VALU = valu_issue_clocks()
VALU_PEAK_GINSTR = SIMDS * CLOCK_GHZ / CLK_PER_WAVE_VALU  # ~1062 G wave-instructions / s

WORKLOADS = {
    # name: batch per GPU, icosphere level, image, K, texture
    "C4_B32_128x128_K50_ico3": dict(B=32, level=3, img=128, K=50, tex=64),
    "C2_B32_64x64_K50_ico3": dict(B=32, level=3, img=64, K=50, tex=64),
    "C4b_B64_128x128_K50_ico4": dict(B=64, level=4, img=128, K=50, tex=64),
    "C2p_B192_64x64_K50_ico3": dict(B=192, level=3, img=64, K=50, tex=64),      # pose step of cfg2: 32 x 6 candidates
    # the same pose step called the way unicorn.py:226-233 calls it: 32 meshes (detached), repeat(meshes, 6), 192
    # poses; gradients to R and T only.  The tiling is an index (structures.repeat -> n_meshes of the C ABI), no copies.
    "C2pt_B32x6_64x64_K50_ico3": dict(B=192, poses=6, level=3, img=64, K=50, tex=64, grads="R+T"),
    "C3_B32_64x64_K25_ico3": dict(B=32, level=3, img=64, K=25, tex=64),         # cfg3: faces_per_pixel 25
    "C3p_B192_64x64_K25_ico3": dict(B=192, level=3, img=64, K=25, tex=64),
    "C5_B16_256x256_K100_ico5": dict(B=16, level=5, img=256, K=100, tex=64),
    "C5_B128_256x256_K100_ico5": dict(B=128, level=5, img=256, K=100, tex=64),  # cfg[4] at its per-GPU shard size
}


def path_bytes_per_image(H, W, K, F, Ht, Wt):
    """fragments written once (24 B) and read once (20 B) + RGBA out/grad + faces + texture."""
    return 44 * H * W * K + 32 * H * W + 108 * F + 24 * Ht * Wt


def kernel_bytes_per_image(call, H, W, K, F, Ht, Wt):
    """The forward / backward launch's share of SURVEY.md 8(d)'s per-image figure: the forward owns the fragment
    write (24 B per (pixel, k)), the RGBA image, one pass over the faces and the texture read; the backward owns the
    fragment read (20 B), the RGBA cotangent, faces read + gradient write and the texture-gradient write."""
    if "backward" in call:
        return 20 * H * W * K + 16 * H * W + 72 * F + 12 * Ht * Wt
    return 24 * H * W * K + 16 * H * W + 36 * F + 12 * Ht * Wt


def cpu_baseline(cfg, seconds_budget=12.0):
    """The oracle's whole path (forward + backward) on the host cores: repeated passes over a small resident sample
    (the oracle materialises fragments: a few images per thread keep it within a few GB) until about
    `seconds_budget` seconds of CPU work are spent, half on one thread, half on all of the job's cores."""
    from oracle import orc
    s = float(1.0 / np.tan(np.deg2rad(15.0)))
    H, K = cfg["img"], cfg["K"]

    def rate(n_images, n_threads, budget):
        sc = synth.make_scene(n_images, level=cfg["level"], tex_size=cfg["tex"], seed=0)
        cot = np.random.RandomState(0).randn(n_images, 4, H, H).astype(np.float32)
        args = (sc["verts"].numpy(), sc["faces"].numpy(), sc["R"].numpy(), sc["T"].numpy(), sc["maps"].numpy(),
                sc["verts_uvs"][None].expand(n_images, -1, -1).numpy())
        done, t0 = 0, time.perf_counter()
        while True:
            orc.render_fwd_bwd(*args, H=H, W=H, K=K, s=s, grad_image=cot, n_threads=n_threads)
            done += n_images
            dt = time.perf_counter() - t0
            if dt >= budget or done >= 100 * n_images:
                return done / dt, done

    single, n_single = rate(2, 1, seconds_budget / 2)
    cores = min(16, os.cpu_count() or 1)  # the GPU box gives a 1-GPU job a 16-core share
    multi, n_multi = rate(cores * 2, cores, seconds_budget / 2)
    return {"value": round(multi, 3), "unit": "images/s", "cores": cores, "kind": "port",
            "single_thread_value": round(single, 3),
            "sample": f"{n_multi} images of the same workload ({H}x{H}, K={K}, ico{cfg['level']}) fwd+bwd on {cores} "
                      f"threads ({cores * 2} per pass, meshes split over threads); {n_single} image(s) on 1 thread for "
                      f"single_thread_value; oracle/ C++ restatement of the pytorch3d-style naive CPU rasterizer "
                      f"(pytorch3d itself is not available offline)"}


def launch_own_ranks(n):
    """`python bench.py --gpus N` started directly: run the N ranks as a child torch.distributed.run and pass its exit
    code on.  Nothing here touches the GPU (device_count() does not initialise it on this image)."""
    env = dict(os.environ)
    if torch.cuda.device_count() < n:
        env["UNICORN_BENCH_BACKEND"] = "gloo"  # fewer GPUs than ranks: the ranks share them (rehearsal)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def kernel_roofline(name, ms, counters, alg_bytes):
    """one kernel against the VALU-issue roof (and its measured HBM traffic against the HBM roof).  `counters`: mean
    per launch from rocprofv3 --pmc (SQ_INSTS_VALU in wave-instructions; FETCH_SIZE / WRITE_SIZE in KB)."""
    out = {"kernel": name, "kernel_ms": round(ms, 4), "algorithmic_bytes_per_launch": alg_bytes,
           "algorithmic_GBps": round(alg_bytes / (ms * 1e-3) / 1e9, 1)}
    if counters and "SQ_INSTS_VALU" in counters:
        ginstr = counters["SQ_INSTS_VALU"] / (ms * 1e-3) / 1e9
        out.update(valu_wave_instructions=int(counters["SQ_INSTS_VALU"]), valu_Ginstr_per_s=round(ginstr, 1),
                   valu_issue_frac=round(ginstr / VALU_PEAK_GINSTR, 4))
    if counters and "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
        # gfx950: FETCH_SIZE tallies half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section)
        traffic = int((2 * counters["FETCH_SIZE"] + counters["WRITE_SIZE"]) * 1024)
        out.update(traffic=traffic, traffic_GBps=round(traffic / (ms * 1e-3) / 1e9, 1),
                   hbm_frac=round(traffic / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4))
    return out


def replay_rate(workload, dev, steps, warmup, min_seconds=0.2):
    """graph-replay rate of another workload on this GPU (forward + backward, all gradients), timed like the headline:
    blocks of `steps` steps until `min_seconds`, median block.  A side field: errors are reported, not raised."""
    try:
        from unicorn_amd.graphs import GraphedStep
        cfg = WORKLOADS[workload]
        B, H, K = cfg["B"], cfg["img"], cfg["K"]
        sc = synth.make_scene(B, level=cfg["level"], tex_size=cfg["tex"], seed=0, device=dev)
        rend = Renderer(H, cameras={"name": "fov", "fov": 30}, faces_per_pixel=K, sigma=1e-4).to(dev)
        leaves = [sc[k].requires_grad_(True) for k in ("verts", "maps", "R", "T")]
        fb, ub = sc["faces"][None].expand(B, -1, -1), sc["verts_uvs"][None].expand(B, -1, -1)
        cot = torch.randn(B, 4, H, H, device=dev, generator=torch.Generator(device=dev).manual_seed(0))

        def body():
            img = rend(Meshes(verts=leaves[0], faces=fb, textures=TexturesUV(leaves[1], fb, ub)), leaves[2], leaves[3])
            img.backward(cot)
            return img
        run = GraphedStep(body, grads_of=tuple(leaves))
        blocks, _ = udist.timed_blocks(run, steps, warmup, 1, cuda=True, min_seconds=min_seconds)
        t = udist.median_block(blocks)
        return {"images_per_s": round(B * steps / t, 1), "ms_per_step": round(t / steps * 1e3, 4), "batch": B,
                "image": [H, H], "faces_per_pixel": K, "launch": "hipGraph replay", "blocks": len(blocks)}
    except Exception as e:
        return {"error": f"{type(e).__name__}: {e}"[:200]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS),
                    help="default: C4_B32_128x128_K50_ico3, the configuration BASELINE.json's metric is quoted on")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--min-seconds", type=float, default=0.3, help="blocks of --steps timed steps are repeated until "
                    "they add up to this long (0: one block; the counter passes of tools/prof_workload.sh use that)")
    ap.add_argument("--eager", action="store_true", help="time eager Python steps instead of replaying the step's "
                    "hipGraph (the eager rate is reported either way, as eager_images_per_s)")
    ap.add_argument("--grads", default=None, help="which inputs require grad, e.g. verts+texture+R+T (ablation only; "
                    "the reported metric uses the workload's default = all of them, R+T for the tiled pose step)")
    a = ap.parse_args()
    # the plain default run (what the driver starts) also times the reference's own training size, as a side field
    side_workloads = ["C2_B32_64x64_K50_ico3"] if (a.workload is None and a.gpus == 1 and not a.eager) else []
    a.workload = a.workload or "C4_B32_128x128_K50_ico3"
    if a.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_own_ranks(a.gpus))

    rank, local, world = udist.env_world()
    # rehearsal on a box with fewer GPUs than ranks: UNICORN_BENCH_BACKEND=gloo lets the ranks share a device
    backend = os.environ.get("UNICORN_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    udist.init(backend)
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    dev = torch.device("cuda", local)
    native.lib()

    cfg = WORKLOADS[a.workload]
    B, H, K = cfg["B"], cfg["img"], cfg["K"]
    a.grads = a.grads or cfg.get("grads", "verts+texture+R+T")
    poses = cfg.get("poses", 1)       # candidate poses per mesh (K-major tiling, utils/mesh.py:43-71)
    M = B // poses                    # distinct meshes
    sc = synth.make_scene(B, level=cfg["level"], tex_size=cfg["tex"], seed=udist.shard_seed(0, rank), device=dev)
    F = sc["faces"].shape[0]
    rend = Renderer(H, cameras={"name": "fov", "fov": 30}, faces_per_pixel=K, sigma=1e-4).to(dev)
    want = set(a.grads.split("+"))
    verts = sc["verts"][:M].contiguous().requires_grad_("verts" in want)
    maps = sc["maps"][:M].contiguous().requires_grad_("texture" in want)
    R = sc["R"].requires_grad_("R" in want)
    T = sc["T"].requires_grad_("T" in want)
    fb, ub = sc["faces"][None].expand(M, -1, -1), sc["verts_uvs"][None].expand(M, -1, -1)
    cot = torch.randn(B, 4, H, H, device=dev, generator=torch.Generator(device=dev).manual_seed(0))

    def build_meshes():
        meshes = Meshes(verts=verts, faces=fb, textures=TexturesUV(maps, fb, ub))  # unicorn.py:262-266,283
        return repeat(meshes, poses) if poses > 1 else meshes                      # unicorn.py:232

    def step():
        for t in (verts, maps, R, T):
            t.grad = None
        img = rend(build_meshes(), R, T)
        img.backward(cot)
        return img

    # Eager steps cost 0.08-0.15 ms of Python / autograd on the host (box dependent) against ~0.1 ms of kernels, so the
    # reported loop is the replay of the step's hipGraph: the same kernels on the same buffers, one launch per step.
    # The eager loop is timed too, first (capturing afterwards leaves the leaves' AccumulateGrad nodes tied to the
    # capture stream, which would slow later eager steps down with stream syncs), and reported as a side field.
    eager_blocks, eager_local = udist.timed_blocks(step, a.steps, a.warmup, world, cuda=True,
                                                   min_seconds=a.min_seconds)
    eager_elapsed, eager_local_med = udist.median_block(eager_blocks, eager_local)
    launch, blocks, elapsed, local_elapsed, replay_elapsed = "eager", eager_blocks, eager_elapsed, eager_local_med, None
    if not a.eager:
        try:
            from unicorn_amd.graphs import GraphedStep

            def body():
                img = rend(build_meshes(), R, T)
                img.backward(cot)
                return img
            run = GraphedStep(body, grads_of=tuple(t for t in (verts, maps, R, T) if t.requires_grad))
            blocks, replay_local = udist.timed_blocks(run, a.steps, a.warmup, world, cuda=True,
                                                       min_seconds=a.min_seconds)
            replay_elapsed, local_elapsed = udist.median_block(blocks, replay_local)
            elapsed = replay_elapsed
            launch = "hipGraph replay of one captured Renderer forward + backward"
        except Exception as e:  # capture unsupported: the eager loop is the reported one, and the line says so
            blocks = eager_blocks
            launch = f"eager (graph capture failed: {type(e).__name__}: {e})"[:200]
    # every rank's own rate and a one-tensor all-reduce (RCCL under "nccl") that proves `world` ranks took part
    per_rank = udist.gather_floats(B * a.steps / local_elapsed, world, cuda=True)
    n_ranks_seen = udist.sum_over_ranks(1.0, world, cuda=True)
    # roofline leg (rank 0 only): the same steps again, eagerly, with (i) HIP events recorded by the library on the
    # launch stream right around the two dominant kernels (ur_profile_begin/end) and (ii) events around every C-ABI call
    result = None
    if rank == 0:
        native.enable_timing(True)
        native.profile_begin(2 * a.steps)
        for _ in range(a.steps):
            step()
        torch.cuda.synchronize()
        kernel_ms = {k: sum(v) / len(v) for k, v in native.profile_end(2 * a.steps).items()}
        per_call = native.timing_summary()
        native.enable_timing(False)
        dom = max(kernel_ms, key=kernel_ms.get)  # the dominant kernel: k_raster<2> (fused forward)
        calls = {"k_raster<2>": "ur_render_meshes_forward", "k_render_bwd": "ur_render_meshes_backward"}
        counters, counters_note = {}, None
        if os.path.exists(COUNTERS):  # PMC counters cannot be read from inside the process: last committed passes
            doc = json.load(open(COUNTERS))
            if doc.get("kernel_source_sha256") == native.source_hash():
                counters = doc.get("workloads", {}).get(a.workload, {})
            else:  # taken on other kernels than the ones that just ran: no fraction rather than a stale one
                counters_note = ("profiles/r03_counters.json was taken on other kernel sources (sha256 differs from "
                                 "unicorn_amd/csrc): instruction counts withheld, re-run tools/prof_workload.sh")
        roofs = {k: kernel_roofline(k, ms, counters.get(k), B * kernel_bytes_per_image(calls[k], H, H, K, F, cfg["tex"],
                                                                                       cfg["tex"]))
                 for k, ms in kernel_ms.items() if k in calls}
        rd = roofs[dom]
        ips = udist.job_throughput(B, a.steps, elapsed, world)
        path_gbps = ips / world * path_bytes_per_image(H, H, K, F, cfg["tex"], cfg["tex"]) / 1e9
        result = {
            "metric": "rendered images/sec fwd+bwd (B=32, 128x128, K=50)", "value": round(ips, 2),
            "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "timing": {"blocks": len(blocks), "steps_per_block": a.steps, "timed_seconds": round(sum(blocks), 4),
                       "reported": "median block", "ms_per_step_min": round(min(blocks) / a.steps * 1e3, 4),
                       "ms_per_step_max": round(max(blocks) / a.steps * 1e3, 4)},
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": a.workload, "batch_per_gpu": B, "meshes_per_gpu": M, "image": [H, H],
                       "faces_per_pixel": K,
                       "faces_per_mesh": F, "texture": [cfg["tex"], cfg["tex"]], "sigma": 1e-4,
                       "grads": a.grads, "launch": launch},
            "eager_images_per_s": round(udist.job_throughput(B, a.steps, eager_elapsed, world), 2),
            "graph_replay_images_per_s": (None if replay_elapsed is None else
                                          round(udist.job_throughput(B, a.steps, replay_elapsed, world), 2)),
            "roofline": {"bound": "valu_issue", "kernel": dom, "achieved": rd.get("valu_Ginstr_per_s"),
                         "peak": VALU_PEAK_GINSTR, "unit": "G wave-instr/s", "frac": rd.get("valu_issue_frac"),
                         "traffic": rd.get("traffic"), "kernel_ms": rd["kernel_ms"], "dominant_call": calls[dom],
                         "clock_GHz": CLOCK_GHZ, "simds": SIMDS, "clk_per_wave_valu_instr": CLK_PER_WAVE_VALU,
                         "counters_from": os.path.relpath(COUNTERS, ROOT) if counters else counters_note,
                         "issue_rate_from": os.path.relpath(VALU_ISSUE, ROOT),
                         "kernels": [roofs[k] for k in sorted(roofs)],
                         "note": "achieved = SQ_INSTS_VALU per launch (rocprofv3 --pmc, committed under profiles/, "
                                 "stamped with the kernel sources' hash) / the kernel's duration measured live with "
                                 "HIP events on its launch stream; peak = 1024 SIMDs x 2.4 GHz / the measured clocks "
                                 "per wave64 VALU instruction of a SIMD shared by >= 2 waves (tools/micro/"
                                 "valu_issue.hip: 2.31 for plain f32 / int ops -- the roof of the cheapest "
                                 "instructions; min / max / packed ops cost twice that).  The kernels are per-pixel "
                                 "geometry tests, sorts and scatters: not streaming, no MFMA (DESIGN.md 6); SURVEY "
                                 "8(d)'s byte model is kept as path_hbm / algorithmic_GBps (fragments counted as if "
                                 "materialised; the fused kernels never store them, so that figure is not a bound)"},
            "path_hbm": {"bytes_per_image": path_bytes_per_image(H, H, K, F, cfg["tex"], cfg["tex"]),
                         "achieved_GBps_per_gpu": round(path_gbps, 2),
                         "frac_of_peak": round(path_gbps / HBM_PEAK_GBPS, 4)},
            "per_call_ms": {k: round(v[1], 4) for k, v in sorted(per_call.items())},
            "ranks": {"backend": backend if world > 1 else None, "images_per_s": [round(v, 1) for v in per_rank],
                      "allreduce_sum_of_ones": n_ranks_seen,
                      "shared_gpus": bool(world > torch.cuda.device_count())},
        }
        if side_workloads:
            # (after everything the headline line is made of; its kernels are other instantiations -- eight-wave bins
            # -- than the headline's, so a rocprofv3 summary of this command keeps the two apart by name)
            result["other_workloads"] = {w: replay_rate(w, dev, a.steps, a.warmup) for w in side_workloads}
        if not a.no_cpu_baseline and world == 1:
            result["cpu_baseline"] = cpu_baseline(cfg)
            result["gpu_over_cpu"] = round(ips / result["cpu_baseline"]["value"], 1)
    udist.finish(world)
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
