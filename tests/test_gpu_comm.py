"""The C-ABI side of the multi-GPU path (include/trx.h "multi-GPU"): trx_assemble_frames against the index-based
de-interleave of tray_racing_amd/dist.py for several world sizes and ragged images, and a real RCCL communicator
(world size 1: the only size a one-GPU box can form) driving trx_gather_shards."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _scene(trx, name="kitchen", n=20000):
    verts, counts = trx.gen_scene(name, n, 1)
    flat = trx.flat_build(verts, counts)
    eye, look, fov = trx.scene_camera(name)
    return flat, (eye, look, fov)


@pytest.mark.parametrize("w,h,world,m", [(64, 40, 1, 1), (100, 37, 2, 1), (131, 67, 8, 3), (256, 144, 4, 8), (9, 9, 8, 2)])
def test_assemble_frames_equals_image_layout_traces(trx, w, h, world, m):
    import torch
    from tray_racing_amd import dist as D
    from tray_racing_amd import _lib as L
    flat, (eye, look, fov) = _scene(trx)
    sc = trx.Scene(flat)
    views = []
    for f in range(m):   # a different camera per frame of the batch
        e = (eye[0] + 0.05 * f, eye[1], eye[2] - 0.03 * f)
        views.append(trx.view_from_camera(e, look, fov, w, h))
    want = torch.empty(m * w * h, dtype=torch.int64, device="cuda")
    for f in range(m):
        sc.trace_primary_dev(views[f], w, h, want[f * w * h:(f + 1) * w * h].data_ptr(), sem=3)
    R = D.max_shard_tiles(w, h, world) * 64
    flat_buf = torch.full((world * m * R,), -1, dtype=torch.int64, device="cuda")
    for r in range(world):   # every "rank" traces its shard of the batch straight into its block, in ONE launch
        block = flat_buf[r * m * R:(r + 1) * m * R]
        sc.trace_primary_batch_dev(views, w, h, block.data_ptr(), R, sem=3, shard=(r, world, 1))
    sc.check()
    got = torch.empty(m * w * h, dtype=torch.int64, device="cuda")
    L.check(L.load().trx_assemble_frames(C.c_void_p(flat_buf.data_ptr()), R, w, h, world, m, C.c_void_p(got.data_ptr()), None))
    torch.cuda.synchronize()
    assert (got == want).all()
    # and the torch statement of the same de-interleave agrees
    fg = D.FrameGather(w, h, 0, world, "cuda", batch=m)
    fg.flat[: world * m * R].copy_(flat_buf)
    assert (fg.assemble(m=m) == want).all()
    sc.close()
    with pytest.raises(trx.TrxError, match="smaller than"):
        L.check(L.load().trx_assemble_frames(C.c_void_p(flat_buf.data_ptr()), 64, w if w > 16 else 64, h if h > 16 else 64, 1, 1,
                                             C.c_void_p(got.data_ptr()), None))


def test_rccl_communicator_through_the_abi(trx):
    """ncclGetUniqueId / ncclCommInitRank / ncclAllGather / ncclCommDestroy driven by libtrx.so (RCCL loaded on
    first use).  A one-GPU box can only form world size 1; the in-place all-gather must then leave the block as it
    is, on the caller's stream, and the frame must come out equal to an image-layout trace."""
    import torch
    from tray_racing_amd import dist as D
    w, h, m = 200, 120, 4
    flat, (eye, look, fov) = _scene(trx, "bistro", 60000)
    sc = trx.Scene(flat)
    view = trx.view_from_camera(eye, look, fov, w, h)
    ident = D.AbiFrameGather.unique_id()
    assert len(ident) == 128 and any(ident)
    fg = D.AbiFrameGather(w, h, 0, 1, torch.device("cuda", 0), ident, batch=m)
    assert fg.world_size() == 1
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        sc.trace_primary_batch_dev([view] * m, w, h, fg.slot(0, m).data_ptr(), fg.records, sem=3, shard=(0, 1, 1),
                                   stream=s.cuda_stream)
        fg.gather(m=m)
        frames = fg.assemble(m=m)
    s.synchronize()
    sc.check(s.cuda_stream)
    want, _ = sc.trace_primary(view, w, h, sem=3)
    got = D.int64_to_hits(frames).reshape(m, w * h)
    for f in range(m):
        assert (got[f]["prim"] == want["prim"]).all() and (got[f]["t"].view(np.uint32) == want["t"].view(np.uint32)).all()
    fg.close()
    sc.close()


def test_torch_nccl_in_place_all_gather_as_bench_calls_it(trx, tmp_path):
    """bench.py's N > 1 loop ends in dist.all_gather_into_tensor(flat[:N*n], flat[r*n:(r+1)*n], async_op=True) on RCCL.
    A one-GPU box can only form a one-rank group, so this checks the exact call (in place, asynchronous, waited on by
    the stream) through torch's RCCL rather than its scaling: a child process initialises the NCCL backend, runs one
    batch of the loop and compares the assembled frames with image-layout traces."""
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    code = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
import tray_racing_amd as T
from tray_racing_amd import dist as D
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
w, h, m = 160, 96, 3
verts, counts = T.gen_scene("kitchen", 20000, 1)
flat = T.flat_build(verts, counts)
eye, look, fov = T.scene_camera("kitchen")
view = T.view_from_camera(eye, look, fov, w, h)
sc = T.Scene(flat)
fg = D.FrameGather(w, h, 0, 1, "cuda", batch=m)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    sc.trace_primary_batch_dev([view] * m, w, h, fg.slot(0, m).data_ptr(), fg.records, sem=3, shard=(0, 1, 1), stream=s.cuda_stream)
    n = m * fg.records
    work = dist.all_gather_into_tensor(fg.flat[:n], fg.flat[:n], async_op=True)   # world 1: FrameGather.gather returns early, so call it here
    work.wait()
    frames = fg.assemble(m=m)
s.synchronize()
want, _ = sc.trace_primary(view, w, h, sem=3)
got = D.int64_to_hits(frames).reshape(m, w * h)
ok = all((got[f]["prim"] == want["prim"]).all() and (got[f]["t"].view(np.uint32) == want["t"].view(np.uint32)).all() for f in range(m))
print("NCCL_OK" if ok and dist.get_backend() == "nccl" else "NCCL_BAD")
sc.close()
dist.destroy_process_group()
''' % root
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert out.returncode == 0 and "NCCL_OK" in out.stdout, (out.stdout[-500:], out.stderr[-1500:])
