#!/usr/bin/env python3
"""bench.py — Mrays/s and frame ms of the MI355X render path (BASELINE.json metric).

    python bench.py [--gpus N --steps K --warmup W] [--workload NAME --width W --height H] [--no-also]

A "step" is ONE FRAME of the workload: Camera::render of the scene at the config's resolution,
scene tables and output image resident in HBM.  Default workload = BASELINE.json configs[1]:
examples/striped_sphere_multiple_lights at 1920x1080 (as the example file is written: two AREA
lights, 8x8 and 4x4 cells, jittered with the counter RNG, seed 0), rendered with the f64 chain
(raytracer_amd/csrc/rt_exact.h) — the mode whose output meets the parity bar.  With N > 1 (launched
by torch.distributed.run, one rank per GPU) the SAME frame is split into interleaved bands of 8
pixel rows (band b -> rank b mod N), every rank renders its bands with the scene replicated, and
the frame is assembled in rank 0's HBM inside the timed region ("strong" scaling):
  * "p2p"   : rank 0 owns ONE shared frame (rt_frame_create), the other ranks map it
              (rt_frame_open, peer access over xGMI) and their kernels store their rows straight into
              it (RT_FLAG_GLOBAL_ROWS) — no collective and no staging copy on the data path; ranks
              only meet at the barriers that bracket the timed region;
  * "copy2d": the same shared frame, but every rank renders into its own HBM and then moves its bands
              with ONE strided device-to-device copy (hipMemcpy2DAsync, bands are 8 rows apart by N)
              — a bulk transfer instead of the kernel's 12-byte stores over the fabric.  Which of the
              two is faster depends on the links; a few untimed frames of each decide;
  * "gather": if the shared frame cannot be set up on every rank, the packed bands go through an RCCL
              gather to rank 0 and are re-interleaved there (plumbing fallback, same kernels).
`RT_BENCH_ASSEMBLY=p2p|copy2d` overrides that choice.  `RT_BENCH_SINGLE_DEVICE=1` (rehearsal on a
one-GPU box) puts every rank on GPU 0 with a gloo control plane; it exercises the same inter-process
frame sharing without xGMI.  After the timed region the assembled frame is checked: the shared frame is
filled with a sentinel, ONE frame is rendered in the chosen mode, and the result must equal the frame one
GPU renders alone, bit for bit.

Rank 0 prints ONE JSON line.
  value     REFERENCE-EQUIVALENT rays per second over the whole job: every ray of the reference's accounting
            for this frame (primary + reflection + refraction + one shadow ray per light cell of every
            shaded hit, light.rs:192-209) divided by the frame time.  The kernel ANSWERS most shadow rays
            without casting them (shaft tests, lights behind the surface): `config.rays_cast` is what it
            really traced, `config.cast_share` the ratio, `value_cast` the rate of traced rays.
  roofline  `frac` = EXECUTED FP32 vector flop/s over the 157.3 TFLOP/s peak: FP32 add/mul/fma/
            transcendental wave-instructions counted by the hardware (rocprofv3 --pmc, scripts/
            roofline_pmc.sh -> profiles/pmc.json, per launch of this workload's kernel) x 64 lanes
            (x 2 for fma), divided by the kernel time measured live here with events on the launch
            stream.  `valu_issue_frac`, `waves_per_simd`, `wait_frac`, `lane_util` come from the same
            counters; `reference_equivalent` is the round-1 figure (the reference algorithm's flop for the
            frame over the kernel time: credit for avoided work, not a hardware fraction).
  also      the other BASELINE configs a multi-GPU run is about (C4 and cover-image at 3840x2160), timed in
            the same run with the same partition and assembly.
  sustained back-to-back frames for >= 0.3 s after the timed steps.
  preroll_frames  untimed frames rendered BEFORE the W warm-up steps (--settle seconds of them, default 0.25 s): the GPU's
            clocks need a few hundred 0.15 ms frames to settle, and a short timed region straight after an idle period
            measured the ramp, not the kernel.
  first_frame_kernel_ms  the kernel time of the FIRST frame of a fresh scene object: scenes with glass start the
            expensive tiles of a frame first once the previous frame has measured them; a single-shot
            render gets this number, the timed loop the steady state.
  render_call_ms  the reference's own call shape, `camera.render(&world)` of a caller that keeps nothing: wall clock of
            rt_scene_create + rt_render (or rt_render_u8, `.to_image()` included) + rt_scene_destroy on rank 0, split
            into create / render (kernel inside) / destroy, and the same render call repeated on the kept scene.
  cpu_baseline  the f64 oracle (a restatement of the reference algorithm, NOT the Rust binary) on this host.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import math
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
ALSO = ["multiple_glass_and_metallic_spheres", "cover_image", "cover_image_area"]  # BASELINE configs 4 and 5
BAND_ROWS = partition.BAND_ROWS
COUNTER_NAMES = ["primary_rays", "reflection_rays", "refraction_rays", "shadow_rays", "shaded_hits", "light_cells",
                 "pattern_lookups", "container_passes", "shadow_rays_cast", "exact_nodes"]
N_CU, N_SIMD, CLOCK_HZ = 256, 4, 2.4e9  # MI355X_MICROARCH.md


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)  # (a frame is 0.16 ms: 20 steps end before the clocks have settled)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="striped_sphere_multiple_lights", choices=sorted(WORKLOADS))
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--jitter", default="counter", choices=["counter", "center"])
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--f32-only", action="store_true", help="RT_FLAG_F32_ONLY: without the f64 chain")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the C4 / cover-image 4K frames")
    ap.add_argument("--settle", type=float, default=0.25, help="seconds of untimed frames before the warm-up steps (clock settling); 0 = none")
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


def pmc_of(workload: str, kernel: str):
    """Per-launch hardware counters of this workload's kernel, collected in their own rocprofv3 --pmc runs
    (scripts/roofline_pmc.sh, merged into profiles/pmc.json) — valid for the config's full frame on one GPU."""
    path = os.path.join(ROOT, "profiles", "pmc.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        table = json.load(f).get(workload, {})
    for k, v in table.items():
        if kernel in k:
            return v
    return None


def hardware_roofline(pmc: dict | None, kernel_ms: float) -> dict:
    """Executed FP32 flop/s and issue statistics from the committed counters and the live kernel time."""
    if not pmc:
        return {"executed_tflops": None, "frac": None, "valu_issue_frac": None, "waves_per_simd": None, "wait_frac": None,
                "lane_util": None, "traffic": None, "counters": "no PMC run of this workload/kernel in profiles/pmc.json"}
    g = pmc.get
    f32 = g("SQ_INSTS_VALU_ADD_F32", 0) + g("SQ_INSTS_VALU_MUL_F32", 0) + 2 * g("SQ_INSTS_VALU_FMA_F32", 0) + g("SQ_INSTS_VALU_TRANS_F32", 0)
    f64 = g("SQ_INSTS_VALU_ADD_F64", 0) + g("SQ_INSTS_VALU_MUL_F64", 0) + 2 * g("SQ_INSTS_VALU_FMA_F64", 0) + g("SQ_INSTS_VALU_TRANS_F64", 0)
    sec = kernel_ms * 1e-3
    executed = 64.0 * f32 / sec / 1e12  # every counted instruction a full wave: an upper bound on executed FP32
    out = {
        "executed_tflops": executed, "frac": executed / flopmodel.PEAK_FP32_TFLOPS,
        "executed_f64_tflops": 64.0 * f64 / sec / 1e12,
        # wave-instructions issued on the vector pipes over what the chip can issue (one wave64 op per 2 cycles per SIMD)
        "valu_issue_frac": g("SQ_INSTS_VALU", 0) / sec / flopmodel.PEAK_VALU_WAVE_INSTS_PER_S,
        "fp32_share_of_valu": (g("SQ_INSTS_VALU_ADD_F32", 0) + g("SQ_INSTS_VALU_MUL_F32", 0) + g("SQ_INSTS_VALU_FMA_F32", 0) +
                               g("SQ_INSTS_VALU_TRANS_F32", 0)) / max(1.0, g("SQ_INSTS_VALU", 0)),
        # SQ_WAVE_CYCLES counts quad-cycles summed over waves; GRBM_GUI_ACTIVE sums the 8 XCDs' busy cycles
        "waves_per_simd": 4.0 * g("SQ_WAVE_CYCLES", 0) / (g("GRBM_GUI_ACTIVE", 1) / 8.0) / (N_CU * N_SIMD) if g("GRBM_GUI_ACTIVE") else None,
        "wait_frac": g("SQ_WAIT_ANY", 0) / g("SQ_WAVE_CYCLES", 1) if g("SQ_WAVE_CYCLES") else None,
        "lane_util": g("SQ_THREAD_CYCLES_VALU", 0) / (64.0 * g("SQ_ACTIVE_INST_VALU", 1)) if g("SQ_ACTIVE_INST_VALU") else None,
        # HBM bytes per launch: FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE doubled per the gfx950 correction
        "traffic": int(1024 * (2 * g("FETCH_SIZE", 0) + g("WRITE_SIZE", 0))) if g("WRITE_SIZE") else None,
        "counters": "profiles/pmc.json (rocprofv3 --pmc, scripts/roofline_pmc.sh), per launch; kernel time measured live",
    }
    return out


class Job:
    """Process-wide state of one bench run: ranks, devices, control plane."""

    def __init__(self, args):
        self.args = args
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if args.gpus != self.world and self.world == 1 and args.gpus > 1:
            print(f"bench.py: --gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks", file=sys.stderr)
            sys.exit(2)
        self.rehearsal = os.environ.get("RT_BENCH_SINGLE_DEVICE") == "1"
        if self.rehearsal:
            self.local_rank = 0
        torch.cuda.set_device(self.local_rank)
        self.dev = torch.device("cuda", self.local_rank)
        self.ctl = torch.device("cpu") if self.rehearsal else self.dev  # where the control-plane tensors live
        self.dist = None
        if self.world > 1:
            import torch.distributed as dist_mod

            self.dist = dist_mod
            if self.rehearsal:
                self.dist.init_process_group(backend="gloo")
            else:
                self.dist.init_process_group(backend="nccl", device_id=self.dev)
        self.hip = None
        try:
            self.hip = C.CDLL("libamdhip64.so")
            self.hip.hipMemcpy2DAsync.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int,
                                                  C.c_void_p]
            self.hip.hipMemcpy2DAsync.restype = C.c_int
            self.hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
            self.hip.hipMemset.restype = C.c_int
        except (OSError, AttributeError):
            self.hip = None

    def sync_all(self):
        torch.cuda.synchronize()
        if self.world > 1:
            self.dist.barrier()
            torch.cuda.synchronize()

    def agree(self, ok: bool) -> bool:
        """True iff every rank says ok (the same collective on every rank, whatever happened locally)."""
        if self.world == 1:
            return ok
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self.ctl)
        self.dist.all_reduce(flag, op=self.dist.ReduceOp.MIN)
        return int(flag.item()) == 1

    def max_over_ranks(self, v: float) -> float:
        if self.world == 1:
            return v
        t = torch.tensor([v], dtype=torch.float64, device=self.ctl)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())


def one_shot_calls(ex: ExampleScene, device: int, opt: RtOptions) -> dict:
    """`camera.render(&world)` as ONE call of a caller that keeps nothing (the reference's API, camera.rs:108): wall clock of
    rt_scene_create + rt_render + rt_scene_destroy on a whole frame, split into its parts, for the float canvas
    (rt_render) and for the byte image (`.to_image()` included, rt_render_u8); `again_*` = the same call repeated on the
    scene it already has (tile costs known, device image in place).  Median of 3, whole frames on this GPU."""
    def once(u8: bool):
        t0 = time.perf_counter()
        sc = Scene(ex.scene_desc, device)
        t1 = time.perf_counter()
        _, st = (sc.render_u8 if u8 else sc.render)(ex.camera_desc, opt)
        t2 = time.perf_counter()
        _, st2 = (sc.render_u8 if u8 else sc.render)(ex.camera_desc, opt)
        t3 = time.perf_counter()
        sc.close()
        t4 = time.perf_counter()
        return {"total_ms": (t2 - t0 + t4 - t3) * 1e3, "create_ms": (t1 - t0) * 1e3, "render_ms": (t2 - t1) * 1e3,
                "kernel_ms": float(st.kernel_ms), "destroy_ms": (t4 - t3) * 1e3, "again_render_ms": (t3 - t2) * 1e3,
                "again_kernel_ms": float(st2.kernel_ms)}

    out = {}
    for key, u8 in (("f32_canvas", False), ("u8_image", True)):
        runs = [once(u8) for _ in range(3)]
        out[key] = {k: float(np.median([r[k] for r in runs])) for k in runs[0]}
    return out


def run_workload(job: Job, name: str, width: int, height: int, steps: int, warmup: int, sustain_s: float, settle_s: float = 0.0) -> dict:
    """Times `steps` frames of one workload on this job's partition; returns rank 0's report."""
    args, rank, world, dist = job.args, job.rank, job.world, job.dist
    ex = ExampleScene(name, width, height)
    scene = Scene(ex.scene_desc, job.local_rank)
    n_lights = int(ex.scene_desc.contents.n_lights)

    base = RtOptions()
    base.jitter_mode = 1 if args.jitter == "counter" else 0
    base.seed = args.seed
    base.flags = 2 if args.f32_only else 0
    opt = partition.rank_options(base, rank, world)
    max_rows = partition.padded_rows(height, world)
    assert scene.output_rows(ex.camera_desc, opt) <= max_rows
    local = torch.zeros((max_rows, width, 3), dtype=torch.float32, device=job.dev)

    # Frame assembly for N > 1: one shared frame in rank 0's HBM, or the gather fallback.
    shared = None
    mode = "single"
    if world > 1:
        mode = "p2p"
        box = [None, None]
        if rank == 0:
            try:
                shared = SharedFrame(height, width, job.local_rank)
                box = [shared.handle, job.local_rank]
            except Exception as e:  # noqa: BLE001 - any failure means: use the gather path
                print(f"bench.py: shared frame not available on rank 0 ({e})", file=sys.stderr)
        dist.broadcast_object_list(box, src=0)
        ok = box[0] is not None
        if ok and rank != 0:
            try:
                shared = SharedFrame(height, width, job.local_rank, owner_device=box[1], handle=box[0])
            except Exception as e:  # noqa: BLE001
                print(f"bench.py: rank {rank} cannot map the shared frame ({e})", file=sys.stderr)
                ok = False
        if not job.agree(ok):
            mode = "gather"
            if shared is not None:
                shared.close()
                shared = None
    opt_frame = opt
    if mode == "p2p":
        opt_frame = partition.rank_options(base, rank, world)
        opt_frame.flags = base.flags | RT_FLAG_GLOBAL_ROWS
    final = perm = gathered = None
    if mode == "gather" and rank == 0:
        gathered = [torch.empty_like(local) for _ in range(world)]
        perm = torch.from_numpy(partition.gather_permutation(height, world)).to(job.dev)
        final = torch.empty((height, width, 3), dtype=torch.float32, device=job.dev)

    my_rows = int(scene.output_rows(ex.camera_desc, opt))

    def move_bands(stream):
        """This rank's packed bands -> their rows of the shared frame: one strided copy (+ a last partial band)."""
        for dst, dpitch, src, spitch, wbytes, rows in partition.band_copy_plan(width, rank, world, my_rows):
            rc = job.hip.hipMemcpy2DAsync(shared.ptr.value + dst, dpitch, local.data_ptr() + src, spitch, wbytes, rows, 4, stream)
            if rc != 0:
                raise RuntimeError(f"hipMemcpy2DAsync failed ({rc})")

    def frame(m):
        stream = torch.cuda.current_stream().cuda_stream
        if m == "p2p":
            scene.render_device(ex.camera_desc, shared.ptr.value, opt_frame, stream)
            return
        scene.render_device(ex.camera_desc, local.data_ptr(), opt, stream)
        if m == "copy2d":
            move_bands(stream)
        if m == "gather":
            dist.gather(local, gathered, dst=0)
            if rank == 0:
                torch.index_select(torch.cat(gathered, 0), 0, perm, out=final)

    # ray / work counters of this rank's share (untimed, instrumented kernel variant)
    _, st = scene.render(ex.camera_desc, opt, stats=True)
    mine = {k: int(getattr(st, k)) for k in COUNTER_NAMES}
    counts = torch.tensor([mine[k] for k in COUNTER_NAMES], dtype=torch.int64, device=job.ctl)
    if world > 1:
        dist.all_reduce(counts)
    total = dict(zip(COUNTER_NAMES, (int(v) for v in counts.tolist())))
    total_rays = total["primary_rays"] + total["reflection_rays"] + total["refraction_rays"] + total["shadow_rays"]
    rays_cast = total["primary_rays"] + total["reflection_rays"] + total["refraction_rays"] + total["shadow_rays_cast"]

    # direct stores or one strided copy per rank?  A few untimed frames of each; every rank runs the SAME
    # sequence of barriers and collectives whatever happens locally, and takes the same decision (it is taken
    # on the all-reduced times).
    calibration = None
    if mode == "p2p" and job.hip is not None:
        took = {}
        for m in ("p2p", "copy2d"):
            ok, t0 = True, 0.0
            try:
                frame(m)
            except RuntimeError as e:
                print(f"bench.py: rank {rank}: {m} not usable ({e})", file=sys.stderr)
                ok = False
            job.sync_all()
            t0 = time.perf_counter()
            if ok:
                try:
                    for _ in range(5):
                        frame(m)
                except RuntimeError as e:
                    print(f"bench.py: rank {rank}: {m} failed ({e})", file=sys.stderr)
                    ok = False
            job.sync_all()
            dt = time.perf_counter() - t0
            took[m] = job.max_over_ranks(dt if ok else 1e30)
        mode = "copy2d" if took["copy2d"] < took["p2p"] else "p2p"
        forced = os.environ.get("RT_BENCH_ASSEMBLY")  # "p2p" / "copy2d": skip the choice (rehearsals)
        if forced in took and took[forced] < 1e29:
            mode = forced
        calibration = {k: (v / 5 * 1e3 if v < 1e29 else None) for k, v in took.items()}

    # Pre-roll (untimed, before the W warm-up steps): a frame is 0.15 ms and the GPU's clocks take a few hundred of them
    # to settle — a 20-step timed region straight after an idle period read 11 % slower than a 200-step one (round 2).
    # Every rank runs the same number of frames (the count comes from an all-reduced time).
    preroll = 0
    if settle_s > 0:
        job.sync_all()
        t0 = time.perf_counter()
        for _ in range(5):
            frame(mode)
        job.sync_all()
        per_frame = job.max_over_ranks(time.perf_counter() - t0) / 5
        preroll = int(min(5000, max(0, math.ceil(settle_s / max(per_frame, 1e-6)))))
        for _ in range(preroll):
            frame(mode)
        job.sync_all()
    for _ in range(warmup):
        frame(mode)
    job.sync_all()
    # Kernel time by HIP events on the launch stream.  One GPU: ONE pair around the K launches of the timed region
    # (average launch-to-launch time of back-to-back launches; a pair per launch puts a barrier packet between the
    # kernels and cost the headline 0.012 ms a frame).  Several GPUs: a pair per launch, so that the copies and
    # collectives that follow a kernel stay outside its time.
    per_launch = world > 1
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps if per_launch else 1)]
    t0 = time.perf_counter()
    if not per_launch:
        ev[0][0].record()
    for k in range(steps):
        stream = torch.cuda.current_stream().cuda_stream
        if per_launch:
            ev[k][0].record()
        if mode == "p2p":
            scene.render_device(ex.camera_desc, shared.ptr.value, opt_frame, stream)
        else:
            scene.render_device(ex.camera_desc, local.data_ptr(), opt, stream)
        if per_launch:
            ev[k][1].record()
        if mode == "copy2d":
            move_bands(stream)
        if mode == "gather":
            dist.gather(local, gathered, dst=0)
            if rank == 0:
                torch.index_select(torch.cat(gathered, 0), 0, perm, out=final)
    if not per_launch:
        ev[0][1].record()
    job.sync_all()
    elapsed = job.max_over_ranks(time.perf_counter() - t0)
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev])) / (1 if per_launch else steps)  # this rank's render kernel

    # sustained: back-to-back frames for >= sustain_s (every rank runs the same count)
    sustained = None
    if sustain_s > 0:
        n = max(steps, int(math.ceil(1.3 * sustain_s / max(elapsed / steps, 1e-6))))  # frames without event records run faster
        job.sync_all()
        t0 = time.perf_counter()
        for _ in range(n):
            frame(mode)
        job.sync_all()
        dt = job.max_over_ranks(time.perf_counter() - t0)
        sustained = {"frames": n, "seconds": dt, "ms_per_frame": dt * 1e3 / n}

    # The assembled frame must be the frame one GPU renders alone, bit for bit (untimed).  The frame is first
    # filled with a sentinel, then ONE frame is rendered in the chosen mode: stale contents cannot pass.
    verified = None
    if world > 1:
        job.sync_all()
        if rank == 0:
            if mode in ("p2p", "copy2d"):
                if job.hip is None or job.hip.hipMemset(shared.ptr, 0xFF, shared.nbytes) != 0:
                    print("bench.py: could not fill the shared frame with the sentinel", file=sys.stderr)
            else:
                final.fill_(float("nan"))
        job.sync_all()
        frame(mode)
        job.sync_all()
        if rank == 0:
            whole, _ = scene.render(ex.camera_desc, base)
            got = shared.download() if mode in ("p2p", "copy2d") else final.cpu().numpy()
            verified = bool(np.array_equal(whole, got))

    # The FIRST frame of a scene runs in raster order (no measured tile costs yet): a fresh scene object,
    # one frame of this rank's share, untimed by the loop above and reported beside the steady state.
    first_kernel_ms = None
    render_call = None
    if rank == 0:
        fresh = Scene(ex.scene_desc, job.local_rank)
        _, st_first = fresh.render(ex.camera_desc, opt)
        first_kernel_ms = float(st_first.kernel_ms)
        fresh.close()
        render_call = one_shot_calls(ex, job.local_rank, base)

    rep = None
    if rank == 0:
        ms = elapsed * 1e3 / steps
        rep = {
            "workload": f"examples/{name} at {width}x{height}, jitter={args.jitter} seed={args.seed}, RECURSION_DEPTH=5, "
                        f"{'FP32 only' if args.f32_only else 'f64 chain on'}",
            "frame_ms": ms, "kernel_ms": kernel_ms, "first_frame_kernel_ms": first_kernel_ms, "render_call_ms": render_call, "assembly": mode, "assembly_calibration_ms_per_frame": calibration,
            "assembled_frame_equals_single_gpu_frame": verified, "sustained": sustained, "preroll_frames": preroll,
            "rays_per_frame": total_rays, "rays_cast": rays_cast, "cast_share": rays_cast / max(1, total_rays),
            "rays_by_kind": total, "mrays_per_s": total_rays / (ms * 1e-3) / 1e6, "mrays_cast_per_s": rays_cast / (ms * 1e-3) / 1e6,
            "kernel": scene.kernel_name(ex.camera_desc, opt), "mine": mine, "n_lights": n_lights,
            "rows_of_rank0": my_rows, "example": ex,  # ex owns the arrays scene_desc points to
        }
    if world > 1:
        job.sync_all()
        if shared is not None and rank != 0:
            shared.close()
        dist.barrier()
        if shared is not None and rank == 0:
            shared.close()
    scene.close()
    return rep


def main():
    args = parse_args()
    job = Job(args)
    world = job.world
    width, height = WORKLOADS[args.workload]
    width, height = args.width or width, args.height or height
    head = run_workload(job, args.workload, width, height, args.steps, args.warmup, sustain_s=0.3, settle_s=args.settle)

    also = {}
    if not args.no_also:
        for name in ALSO:
            if name == args.workload:
                continue
            w, h = WORKLOADS[name]
            r = run_workload(job, name, w, h, steps=5, warmup=1, sustain_s=0.0)
            if r is not None:
                also[name] = {k: r[k] for k in ("workload", "frame_ms", "kernel_ms", "first_frame_kernel_ms", "render_call_ms", "assembly", "assembled_frame_equals_single_gpu_frame",
                                                "rays_per_frame", "rays_cast", "mrays_per_s", "mrays_cast_per_s", "kernel")}

    if job.rank == 0:
        opt_jitter = 1 if args.jitter == "counter" else 0
        cpu, ost = (None, None)
        if world == 1 and not args.no_cpu_baseline:
            cpu, ost = cpu_baseline(args.workload, width, height, opt_jitter, args.seed, args.cpu_seconds)
        cost = flopmodel.ray_cost(head["example"].scene_desc, ost)
        flops = flopmodel.frame_flops(head["mine"], head["n_lights"], cost)  # rank 0's launch
        kernel_ms = head["kernel_ms"]
        ref_equiv = flops / (kernel_ms * 1e-3) / 1e12
        default_frame = world == 1 and not args.width and not args.height and args.jitter == "counter" and args.seed == 0
        hw = hardware_roofline(pmc_of(args.workload, head["kernel"]) if default_frame else None, kernel_ms)
        out = {
            "metric": "Mrays/sec (primary+secondary) and frame ms at 1080p/4K, 1 and 8 MI355X",
            "value": head["mrays_per_s"],
            "unit": "Mrays/s",
            "value_counts": "reference-equivalent rays (every ray of the reference's accounting; see config.rays_cast)",
            "value_cast": head["mrays_cast_per_s"],
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": head["frame_ms"],
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": head["workload"],
                "partition": "whole frame on one GPU" if world == 1 else (
                    f"interleaved {BAND_ROWS}-row bands over {world} GPUs, every rank's kernel stores its rows straight "
                    f"into ONE frame in rank 0's HBM (peer access over xGMI, no collective)" if head["assembly"] == "p2p" else
                    f"interleaved {BAND_ROWS}-row bands over {world} GPUs, every rank moves its bands with one strided "
                    f"device-to-device copy into ONE frame in rank 0's HBM (peer access over xGMI, no collective)"
                    if head["assembly"] == "copy2d" else
                    f"interleaved {BAND_ROWS}-row bands over {world} GPUs + RCCL gather to rank 0"),
                "assembly": head["assembly"], "assembly_calibration_ms_per_frame": head["assembly_calibration_ms_per_frame"],
                "assembled_frame_equals_single_gpu_frame": head["assembled_frame_equals_single_gpu_frame"],
                "rays_per_frame": head["rays_per_frame"], "rays_cast": head["rays_cast"], "cast_share": head["cast_share"],
                "rays_by_kind": head["rays_by_kind"],
            },
            "frame_ms": head["frame_ms"],
            "kernel_ms": kernel_ms,
            "first_frame_kernel_ms": head["first_frame_kernel_ms"],
            "render_call_ms": head["render_call_ms"],
            "sustained": head["sustained"],
            "preroll_frames": head["preroll_frames"],
            "roofline": {
                "bound": "valu_fp32", "achieved": hw["executed_tflops"], "peak": flopmodel.PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                "frac": hw["frac"], "traffic": hw["traffic"],
                "executed_tflops": hw["executed_tflops"], "executed_f64_tflops": hw.get("executed_f64_tflops"),
                "valu_issue_frac": hw["valu_issue_frac"], "fp32_share_of_valu": hw.get("fp32_share_of_valu"),
                "waves_per_simd": hw["waves_per_simd"], "wait_frac": hw["wait_frac"], "lane_util": hw["lane_util"],
                "counters": hw["counters"],
                "reference_equivalent": {"tflops": ref_equiv, "frac_of_peak": ref_equiv / flopmodel.PEAK_FP32_TFLOPS,
                                         "flop_per_launch": flops, "flop_per_ray_intersection": cost,
                                         "note": "the REFERENCE algorithm's flop for this frame over the kernel time: credit for "
                                                 "avoided work (culling, shadow rays answered without casting), not a hardware fraction"},
                "algorithmic_hbm_bytes_per_launch": head["rows_of_rank0"] * width * 12,
                "kernel": head["kernel"], "kernel_ms": kernel_ms,
            },
            "also": also,
        }
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out), flush=True)
    if world > 1:
        job.dist.destroy_process_group()


if __name__ == "__main__":
    main()
