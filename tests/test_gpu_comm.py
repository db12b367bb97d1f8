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
