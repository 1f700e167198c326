#!/usr/bin/env python3
"""bench.py — Mrays/s and frame ms of the MI355X render path (BASELINE.json metric).

    python bench.py [--gpus N --steps K --warmup W] [--workload NAME --width W --height H]

A "step" is ONE FRAME of the workload: Camera::render of the scene at the config's resolution,
scene tables and output image resident in HBM.  Default workload = BASELINE.json configs[1]:
examples/striped_sphere_multiple_lights at 1920x1080 (as the example file is written: two AREA
lights, 8x8 and 4x4 cells, jittered with the counter RNG, seed 0).  With N > 1 (launched by
torch.distributed.run, one rank per GPU) the SAME frame is split into interleaved bands of 8 pixel
rows (band b -> rank b mod N), every rank renders its bands with the scene replicated, and the frame
is assembled in rank 0's HBM inside the timed region ("strong" scaling):
  * "p2p"   : rank 0 owns ONE shared frame (rt_frame_create), the other ranks map it
              (rt_frame_open, peer access over xGMI) and their kernels store their rows straight into
              it (RT_FLAG_GLOBAL_ROWS) — no collective and no staging copy on the data path; ranks
              only meet at the barriers that bracket the timed region;
  * "copy2d": the same shared frame, but every rank renders into its own HBM and then moves its bands
              with ONE strided device-to-device copy (hipMemcpy2DAsync, bands are 8 rows apart by N)
              — a bulk transfer instead of the kernel's 12-byte stores over the fabric.  Which of the
              two is faster depends on the links; a few untimed frames of each decide (both leave the
              same frame, and it is verified bit for bit either way);
  * "gather": if the shared frame cannot be set up on every rank, the packed bands go through an RCCL
              gather to rank 0 and are re-interleaved there (the round-1 path, kept as plumbing
              fallback — both paths run the same kernels).
`RT_BENCH_ASSEMBLY=p2p|copy2d` overrides that choice.  `RT_BENCH_SINGLE_DEVICE=1` (rehearsal on a one-GPU box) puts every rank on GPU 0 with a gloo control
plane; it exercises the same inter-process frame sharing without xGMI.

Rank 0 prints ONE JSON line.  `value` = unique rays traced per second over the whole job
(primary + reflection + refraction + shadow; the reference's own count is the same for this
workload because nothing in it is reflective).  `roofline` compares the ALGORITHMIC FP32 flop rate of
the render kernel (raytracer_amd/flopmodel.py) with the FP32 vector peak; the kernel
time is measured with events on the stream the kernel is launched on.  `cpu_baseline` is the f64
oracle (a restatement of the reference algorithm, NOT the Rust binary) timed on this host.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from raytracer_amd import RT_FLAG_GLOBAL_ROWS, ExampleScene, RtOptions, Scene, SharedFrame  # noqa: E402
from raytracer_amd import partition  # noqa: E402
from raytracer_amd import flopmodel  # noqa: E402

WORKLOADS = {  # BASELINE.json configs (the scene files as written)
    "striped_sphere_multiple_lights": (1920, 1080),
    "checkered_walls_metallic_sphere": (1280, 720),
    "glass_sphere_checkered_floor_aerial_view": (1920, 1080),
    "multiple_glass_and_metallic_spheres": (3840, 2160),
    "cover_image": (3840, 2160),
    "cover_image_area": (3840, 2160),
    "shapes_zoo": (1920, 1080),
    "obj_mesh": (1920, 1080),
}
BAND_ROWS = partition.BAND_ROWS


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="striped_sphere_multiple_lights", choices=sorted(WORKLOADS))
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--jitter", default="counter", choices=["counter", "center"])
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the oracle sample")
    return ap.parse_args()


def cpu_baseline(name: str, width: int, height: int, jitter: int, seed: int, target_s: float):
    """The f64 oracle on this host: bounded sample of the same workload (checker/baseline only)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers as H

    cores = int(os.environ.get("OMP_NUM_THREADS", "0")) or (os.cpu_count() or 1)
    cores = max(1, min(cores, os.cpu_count() or 1))
    mid = height // 2
    _, st, sec = H.oracle_render(name, width, height, jitter=jitter, seed=seed, threads=cores,
                                 rows=(mid - 4, mid + 4), want_stats=True)  # calibration band
    per_row = max(sec / 8.0, 1e-6)
    rows = int(min(height, max(8, target_s / per_row)))
    r0 = max(0, mid - rows // 2)
    r1 = min(height, r0 + rows)
    _, st, sec = H.oracle_render(name, width, height, jitter=jitter, seed=seed, threads=cores, rows=(r0, r1),
                                 want_stats=True)
    rays = st["primary"] + st["reflect"] + st["refract"] + st["shadow"]
    return {
        "value": rays / sec / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
        "sample": f"rows {r0}..{r1} of {height} at {width}x{height} ({rays} reference-equivalent rays in {sec:.2f} s; "
                  f"f64 C++ restatement of the reference algorithm, RENDER_THREADS={cores})",
        "frame_ms_extrapolated": sec * 1e3 * height / max(1, r1 - r0),
    }, st


def traffic_from_profile(args, world):
    """HBM bytes per launch of the render kernel, measured with PMC counters in their own runs
    (scripts/traffic.sh) and committed as profiles/traffic.json — valid for the default workload on
    one GPU only; anything else reports null."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    default = args.workload == "striped_sphere_multiple_lights" and not args.width and not args.height and \
        args.jitter == "counter" and args.seed == 0
    if world != 1 or not default or not os.path.exists(path):
        return None
    with open(path) as f:
        return json.load(f).get("hbm_bytes_per_launch")


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world == 1 and args.gpus > 1:
        print(f"bench.py: --gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks", file=sys.stderr)
        sys.exit(2)
    rehearsal = os.environ.get("RT_BENCH_SINGLE_DEVICE") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    ctl = torch.device("cpu") if rehearsal else dev  # where the control-plane tensors live
    dist = None
    if world > 1:
        import torch.distributed as dist_mod

        dist = dist_mod
        if rehearsal:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)

    width, height = WORKLOADS[args.workload]
    width, height = args.width or width, args.height or height
    ex = ExampleScene(args.workload, width, height)
    scene = Scene(ex.scene_desc, local_rank)
    n_lights = int(ex.scene_desc.contents.n_lights)

    base = RtOptions()
    base.jitter_mode = 1 if args.jitter == "counter" else 0
    base.seed = args.seed
    opt = partition.rank_options(base, rank, world)
    max_rows = partition.padded_rows(height, world)
    assert scene.output_rows(ex.camera_desc, opt) <= max_rows
    local = torch.zeros((max_rows, width, 3), dtype=torch.float32, device=dev)

    # Frame assembly for N > 1: one shared frame in rank 0's HBM, or the gather fallback.
    shared = None
    mode = "single"
    if world > 1:
        mode = "p2p"
        box = [None, None]
        if rank == 0:
            try:
                shared = SharedFrame(height, width, local_rank)
                box = [shared.handle, local_rank]
            except Exception as e:  # noqa: BLE001 - any failure means: use the gather path
                print(f"bench.py: shared frame not available on rank 0 ({e})", file=sys.stderr)
        dist.broadcast_object_list(box, src=0)
        ok = 1
        if box[0] is None:
            ok = 0
        elif rank != 0:
            try:
                shared = SharedFrame(height, width, local_rank, owner_device=box[1], handle=box[0])
            except Exception as e:  # noqa: BLE001
                print(f"bench.py: rank {rank} cannot map the shared frame ({e})", file=sys.stderr)
                ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device=ctl)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            mode = "gather"
            if shared is not None:
                shared.close()
                shared = None
    opt_frame = opt
    if mode == "p2p":
        opt_frame = partition.rank_options(base, rank, world)
        opt_frame.flags = RT_FLAG_GLOBAL_ROWS
    final = perm = gathered = None
    if mode == "gather" and rank == 0:
        gathered = [torch.empty_like(local) for _ in range(world)]
        perm = torch.from_numpy(partition.gather_permutation(height, world)).to(dev)
        final = torch.empty((height, width, 3), dtype=torch.float32, device=dev)

    hip = None
    if mode == "p2p":
        try:
            hip = C.CDLL("libamdhip64.so")
            hip.hipMemcpy2DAsync.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int,
                                             C.c_void_p]
            hip.hipMemcpy2DAsync.restype = C.c_int
        except (OSError, AttributeError):
            hip = None
    my_rows = int(scene.output_rows(ex.camera_desc, opt))
    def move_bands(stream):
        """This rank's packed bands -> their rows of the shared frame: one strided copy (+ a last partial band)."""
        for dst, dpitch, src, spitch, wbytes, rows in partition.band_copy_plan(width, rank, world, my_rows):
            rc = hip.hipMemcpy2DAsync(shared.ptr.value + dst, dpitch, local.data_ptr() + src, spitch, wbytes, rows, 4, stream)
            if rc != 0:
                raise RuntimeError(f"hipMemcpy2DAsync failed ({rc})")

    def frame():
        stream = torch.cuda.current_stream().cuda_stream
        if mode == "p2p":
            scene.render_device(ex.camera_desc, shared.ptr.value, opt_frame, stream)
            return
        scene.render_device(ex.camera_desc, local.data_ptr(), opt, stream)
        if mode == "copy2d":
            move_bands(stream)
        if mode == "gather":
            dist.gather(local, gathered, dst=0)
            if rank == 0:
                torch.index_select(torch.cat(gathered, 0), 0, perm, out=final)

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # ray / work counters of this rank's share (untimed, instrumented kernel variant)
    _, st = scene.render(ex.camera_desc, opt, stats=True)
    counts = torch.tensor([st.primary_rays, st.reflection_rays, st.refraction_rays, st.shadow_rays, st.shaded_hits,
                           st.light_cells, st.pattern_lookups, st.container_passes], dtype=torch.int64, device=ctl)
    if world > 1:
        dist.all_reduce(counts)
    names = ["primary_rays", "reflection_rays", "refraction_rays", "shadow_rays", "shaded_hits", "light_cells",
             "pattern_lookups", "container_passes"]
    total = dict(zip(names, (int(v) for v in counts.tolist())))
    mine = dict(zip(names, (st.primary_rays, st.reflection_rays, st.refraction_rays, st.shadow_rays, st.shaded_hits,
                            st.light_cells, st.pattern_lookups, st.container_passes)))
    total_rays = total["primary_rays"] + total["reflection_rays"] + total["refraction_rays"] + total["shadow_rays"]

    # direct stores or one strided copy per rank?  A few untimed frames of each; every rank takes the
    # same decision because it is taken on the all-reduced times.
    calibration = None
    if mode == "p2p" and hip is not None:
        took = {}
        for m in ("p2p", "copy2d"):
            mode = m
            ok = 1
            try:
                frame()
                sync_all()
                t0 = time.perf_counter()
                for _ in range(5):
                    frame()
                sync_all()
                took[m] = time.perf_counter() - t0
            except RuntimeError as e:
                print(f"bench.py: rank {rank}: {m} not usable ({e})", file=sys.stderr)
                ok, took[m] = 0, float("inf")
            flag = torch.tensor([took[m] if ok else 1e30], dtype=torch.float64, device=ctl)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            took[m] = float(flag.item())
        mode = "copy2d" if took["copy2d"] < took["p2p"] else "p2p"
        forced = os.environ.get("RT_BENCH_ASSEMBLY")  # "p2p" / "copy2d": skip the choice (rehearsals)
        if forced in took and took[forced] < 1e29:
            mode = forced
        calibration = {k: (v / 5 * 1e3 if v < 1e29 else None) for k, v in took.items()}

    for _ in range(args.warmup):
        frame()
    sync_all()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        stream = torch.cuda.current_stream().cuda_stream
        ev[k][0].record()
        if mode == "p2p":
            scene.render_device(ex.camera_desc, shared.ptr.value, opt_frame, stream)
        else:
            scene.render_device(ex.camera_desc, local.data_ptr(), opt, stream)
        ev[k][1].record()
        if mode == "copy2d":
            move_bands(stream)
        if mode == "gather":
            dist.gather(local, gathered, dst=0)
            if rank == 0:
                torch.index_select(torch.cat(gathered, 0), 0, perm, out=final)
    sync_all()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=ctl)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))  # this rank's render kernel

    # the assembled frame must be the frame one GPU renders alone, bit for bit (untimed check)
    verified = None
    if world > 1 and rank == 0:
        whole, _ = scene.render(ex.camera_desc, base)
        got = shared.download() if mode in ("p2p", "copy2d") else final.cpu().numpy()
        verified = bool(np.array_equal(whole, got))

    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        cpu, ost = (None, None)
        if world == 1 and not args.no_cpu_baseline:
            cpu, ost = cpu_baseline(args.workload, width, height, opt.jitter_mode, opt.seed, args.cpu_seconds)
        cost = flopmodel.ray_cost(ex.scene_desc, ost)
        flops = flopmodel.frame_flops(mine, n_lights, cost)  # rank 0's launch
        achieved = flops / (kernel_ms * 1e-3) / 1e12
        out = {
            "metric": "Mrays/sec (primary+secondary) and frame ms at 1080p/4K, 1 and 8 MI355X",
            "value": total_rays / (ms_per_step * 1e-3) / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"examples/{args.workload} at {width}x{height}, jitter={args.jitter} seed={args.seed}, "
                            f"RECURSION_DEPTH=5",
                "partition": "whole frame on one GPU" if world == 1 else (
                    f"interleaved {BAND_ROWS}-row bands over {world} GPUs, every rank's kernel stores its rows straight "
                    f"into ONE frame in rank 0's HBM (peer access over xGMI, no collective)" if mode == "p2p" else
                    f"interleaved {BAND_ROWS}-row bands over {world} GPUs, every rank moves its bands with one strided "
                    f"device-to-device copy into ONE frame in rank 0's HBM (peer access over xGMI, no collective)"
                    if mode == "copy2d" else
                    f"interleaved {BAND_ROWS}-row bands over {world} GPUs + RCCL gather to rank 0"),
                "assembly": mode, "assembly_calibration_ms_per_frame": calibration,
                "assembled_frame_equals_single_gpu_frame": verified,
                "rays_per_frame": total_rays, "rays_by_kind": total,
            },
            "frame_ms": ms_per_step,
            "kernel_ms": kernel_ms,
            "roofline": {
                "bound": "valu_fp32", "achieved": achieved, "peak": flopmodel.PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / flopmodel.PEAK_FP32_TFLOPS,
                # HBM bytes per launch come from SEPARATE rocprofv3 --pmc runs of this same command
                # (scripts/traffic.sh -> profiles/traffic.json, default workload only); null otherwise
                "traffic": traffic_from_profile(args, world),
                "algorithmic_hbm_bytes_per_launch": int(scene.output_rows(ex.camera_desc, opt)) * width * 12,
                "algorithmic_flop_per_launch": flops, "flop_per_ray_intersection": cost,
                "kernel": "render_kernel<false, false, false, true>", "kernel_ms": kernel_ms,
            },
        }
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out), flush=True)
    if world > 1:
        sync_all()
        if shared is not None and rank != 0:
            shared.close()
        dist.barrier()
        if shared is not None and rank == 0:
            shared.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
