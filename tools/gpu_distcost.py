"""Host-side cost of the N>1 frame loop, measured on ONE GPU with a world_size-1 RCCL group
(development aid): how many microseconds of Python/launch work does one frame cost when the
collective and the assemble are enqueued, and does anything in the loop block the host?"""
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tray_racing_amd as T  # noqa: E402
from tray_racing_amd import dist as D  # noqa: E402

dist.init_process_group("nccl", rank=0, world_size=1, init_method="tcp://127.0.0.1:29533",
                        device_id=torch.device("cuda", 0))
w, h = 1920, 1080
sim = int(os.environ.get("SIM_SHARDS", "8"))
name = sys.argv[1] if len(sys.argv) > 1 else "bistro"
verts, counts = T.gen_scene(name, 0, 1)
flat = T.flat_build(verts, counts)
eye, look, fov = T.scene_camera(name)
view = T.view_from_camera(eye, look, fov, w, h)
scene = T.Scene(flat)
records = D.shard_tiles(w, h, 0, sim) * 64
n_streams = int(sys.argv[3]) if len(sys.argv) > 3 else 8
n_groups = int(sys.argv[4]) if len(sys.argv) > 4 else 1
groups = [dist.group.WORLD] + [dist.new_group([0]) for _ in range(n_groups - 1)]
mask_mode = sys.argv[2] if len(sys.argv) > 2 else "none"


def masked_stream(words):
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    h = C.c_void_p()
    arr = (C.c_uint32 * len(words))(*words)
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(h), C.c_uint32(len(words)), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(h.value)


if mask_mode == "none":
    streams = [torch.cuda.Stream() for _ in range(n_streams)]
else:
    bits = [1] * 256
    if mask_mode == "low8":          # CUs 0..7 reserved
        for i in range(8):
            bits[i] = 0
    elif mask_mode == "stride32":    # one CU in every group of 32 reserved
        for i in range(31, 256, 32):
            bits[i] = 0
    elif mask_mode == "low16":
        for i in range(16):
            bits[i] = 0
    words = [sum(bits[32 * k + b] << b for b in range(32)) for k in range(8)]
    streams = [masked_stream(words) for _ in range(n_streams)]
print("stream mask mode:", mask_mode, "streams", n_streams, "groups", n_groups, flush=True)
gathered = [torch.zeros((1, records), dtype=torch.int64, device="cuda") for _ in range(n_streams)]
frames = [torch.zeros(records, dtype=torch.int64, device="cuda") for _ in range(n_streams)]
inv = torch.randperm(records, device="cuda")


def run(mode, steps=400):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        j = k % n_streams
        s = streams[j]
        with torch.cuda.stream(s):
            scene.trace_primary_dev(view, w, h, gathered[j].data_ptr(), sem=3, shard=(0, sim, 1), stream=s.cuda_stream)
            if mode >= 1:
                work = dist.all_gather_into_tensor(gathered[j].view(-1), gathered[j][0], async_op=True, group=groups[j % n_groups])
                work.wait()
            if mode >= 2:
                torch.index_select(gathered[j].view(-1), 0, inv, out=frames[j])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) / steps * 1e6, (t2 - t0) / steps * 1e6


for mode, label in ((0, "trace only"), (1, "trace + all_gather(world 1, in place)"), (2, "trace + all_gather + index_select assemble")):
    run(mode, 50)
    host_us, total_us = run(mode)
    print("%-48s host enqueue %.1f us/frame, wall %.1f us/frame" % (label, host_us, total_us), flush=True)
dist.destroy_process_group()

# --- batches of F frames per stream: F kernels back to back on one stream, then ONE in-place all-gather (world-1
# stand-in) and ONE assemble on the same stream; n_streams batches in flight ---
dist.init_process_group("nccl", rank=0, world_size=1, init_method="tcp://127.0.0.1:29534", device_id=torch.device("cuda", 0))
for F in (2, 4, 8):
    fgs = [D.FrameGather(w, h, 0, sim, "cuda", batch=F) for _ in range(n_streams)]
    outs = [torch.empty(F * w * h, dtype=torch.int64, device="cuda") for _ in range(n_streams)]

    def run_batched(steps, with_gather=True, with_assemble=True):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        k = 0
        b = 0
        while k < steps:
            m = min(F, steps - k)
            j = b % n_streams
            fg = fgs[j]
            s = streams[j]
            with torch.cuda.stream(s):
                if one_launch:
                    scene.trace_primary_batch_dev([view] * m, w, h, fg.slot(0, m).data_ptr(), fg.records, sem=3,
                                                  shard=(0, sim, 1), stream=s.cuda_stream)
                else:
                    for f in range(m):
                        scene.trace_primary_dev(view, w, h, fg.slot(f, m).data_ptr(), sem=3, shard=(0, sim, 1),
                                                stream=s.cuda_stream)
                k += m
                if with_gather:
                    n = m * fg.records
                    dist.all_gather_into_tensor(fg.flat[:n], fg.flat[:n], async_op=True).wait()
                if with_assemble:
                    fg.assemble(outs[j][: m * w * h], m=m)
            b += 1
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        return (t1 - t0) / steps * 1e6, (t2 - t0) / steps * 1e6

    for one_launch in (False, True):
        for wg, wa, label in ((False, False, "trace"), (True, True, "trace + gather + assemble")):
            run_batched(64, wg, wa)
            hu, tu = run_batched(480, wg, wa)
            print("per-stream batches F=%d %s %-28s host %.1f us/frame, wall %.1f us/frame" % (
                F, "one launch " if one_launch else "F launches ", label, hu, tu), flush=True)
dist.destroy_process_group()
