"""Two PROCESSES assemble one frame: the parent owns it (rt_frame_create), the child maps it
(rt_frame_open) and renders its row bands straight into it (RT_FLAG_GLOBAL_ROWS) while the parent
renders the others.  Run by tests/test_gpu_parity.py::test_shared_frame_across_processes in a fresh
interpreter, the way bench.py's ranks start; prints "ok" when the assembled frame equals the frame one
process renders alone, bit for bit.  Both processes use GPU `device` (the box has one)."""
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

NAME, W, H, SEED = "striped_sphere_multiple_lights", 320, 180, 4


def _options(rank, world):
    from raytracer_amd import RT_FLAG_GLOBAL_ROWS, RtOptions, partition
    base = RtOptions()
    base.jitter_mode, base.seed = 1, SEED
    o = partition.rank_options(base, rank, world)
    o.flags = RT_FLAG_GLOBAL_ROWS
    return base, o


def child(handle, rank, world, q):
    try:
        from raytracer_amd import ExampleScene, Scene, SharedFrame
        ex = ExampleScene(NAME, W, H)
        sc = Scene(ex.scene_desc, 0)
        fr = SharedFrame(H, W, 0, owner_device=0, handle=handle)
        _, o = _options(rank, world)
        sc.render_device(ex.camera_desc, fr.ptr.value, o, 0)
        fr.close()  # waits for the device
        sc.close()
        q.put("ok")
    except Exception as e:  # noqa: BLE001
        q.put(f"child failed: {e!r}")


def main():
    from raytracer_amd import ExampleScene, Scene, SharedFrame
    world = 3
    ex = ExampleScene(NAME, W, H)
    sc = Scene(ex.scene_desc, 0)
    base, o = _options(0, world)
    whole, _ = sc.render(ex.camera_desc, base)
    fr = SharedFrame(H, W, 0)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    kids = [ctx.Process(target=child, args=(fr.handle, r, world, q)) for r in range(1, world)]
    for k in kids:
        k.start()
    sc.render_device(ex.camera_desc, fr.ptr.value, o, 0)
    msgs = [q.get(timeout=100) for _ in kids]
    for k in kids:
        k.join(timeout=60)
    got = fr.download()
    fr.close()
    sc.close()
    if any(m != "ok" for m in msgs):
        print(msgs)
        return 1
    if not np.array_equal(got, whole):
        print("assembled frame differs from the single-process frame")
        return 1
    print("ok")
    return 0


if __name__ == "__main__":
    sys.exit(main())
