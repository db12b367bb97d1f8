"""world_size-2 coverage of the multi-GPU path on CPU (gloo): tile sharding, the one
all-gather of compact hit shards, and re-assembly into the row-major frame.  The shards
are produced by the oracle here (no GPU); on the GPU box the same FrameGather code moves
device tensors over RCCL (tests/test_gpu_parity.py checks the kernel's shard layout)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H = 52, 44  # not multiples of 8: edge tiles are partial


def oracle_compact_shard(osc, ov, w, h, rank, world, records):
    """What trx_trace_primary_dev writes with TRX_LAYOUT_SHARD, computed by the oracle."""
    from oracle import binding as O
    full = np.zeros(w * h, dtype=O.HIT_DTYPE)
    osc.trace_primary(ov, w, h, sem=3, shard=(rank, world), out=full)
    out = np.zeros(records, dtype=O.HIT_DTYPE)
    out["t"] = np.inf
    out["prim"] = 0xFFFFFFFF
    tx = (w + 7) // 8
    n_tiles = tx * ((h + 7) // 8)
    for lt in range(records // 64):
        tile = lt * world + rank
        if tile >= n_tiles:
            break
        for k in range(64):
            px, py = (tile % tx) * 8 + (k & 7), (tile // tx) * 8 + (k >> 3)
            if px < w and py < h:
                out[lt * 64 + k] = full[py * w + px]
    return out


def worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import tray_racing_amd as T
        from oracle import binding as O
        from tray_racing_amd import dist as D
        from helpers import make_scene
        _flat, _v, osc, ov = make_scene(T, O, "cornell", 0, W, H)
        fg = D.FrameGather(W, H, rank, world, "cpu")
        local = fg.new_local()
        shard = oracle_compact_shard(osc, ov, W, H, rank, world, fg.records)
        local.copy_(D.hits_to_int64(shard))
        work = fg.gather(local, async_op=True)
        work.wait()
        frame = D.int64_to_hits(fg.assemble())
        want, _ = osc.trace_primary(ov, W, H, sem=3)
        ok = bool((frame["t"].view(np.uint32) == want["t"].view(np.uint32)).all() and
                  (frame["prim"] == want["prim"]).all())
        # a batch of 3 frames (of a buffer sized for 4) completed by ONE in-place all-gather
        fgb = D.FrameGather(W, H, rank, world, "cpu", batch=4)
        eye, look, fov = T.scene_camera("cornell")
        wants = []
        for f in range(3):
            vf = O.view_from_bytes(T.view_from_camera((eye[0] + 0.3 * f, eye[1], eye[2]), look, fov, W, H))
            fgb.slot(f, 3).copy_(D.hits_to_int64(oracle_compact_shard(osc, vf, W, H, rank, world, fgb.records)))
            wants.append(osc.trace_primary(vf, W, H, sem=3)[0])
        fgb.gather(m=3, async_op=True).wait()
        got = D.int64_to_hits(fgb.assemble(m=3)).reshape(3, W * H)
        for f in range(3):
            ok = ok and bool((got[f]["t"].view(np.uint32) == wants[f]["t"].view(np.uint32)).all() and
                             (got[f]["prim"] == wants[f]["prim"]).all())
        ok = ok and not np.array_equal(wants[0]["prim"], wants[2]["prim"])
        # timing reduce used by bench.py: max over ranks
        t = torch.tensor([float(rank + 1)])
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        q.put((rank, ok, float(t[0]), D.shard_tiles(W, H, rank, world)))
    finally:
        dist.destroy_process_group()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 8])
def test_tile_sharding_and_gather(world):
    """world 2, and world 8: the geometry of the first 8-GPU run (one rank per GPU of a node), every rank a process."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=480) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [r[1] for r in results] == [True] * world
    assert [r[2] for r in results] == [float(world)] * world
    assert sum(r[3] for r in results) == ((W + 7) // 8) * ((H + 7) // 8)


@pytest.mark.timeout(900)
def test_bench_set_up_at_world_8_without_a_gpu(tmp_path):
    """`bench.py --gpus 8` up to the point where it needs a device (--setup-only, gloo): the rendezvous of eight ranks, ONE
    scene build by rank 0 handed to the other seven through the node's temporary directory, the gather geometry of a
    1920x1080 frame dealt to eight ranks - and a second run that finds rank 0's copy instead of building."""
    import json
    import subprocess
    env = dict(os.environ, TMPDIR=str(tmp_path), MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "5",
           "--tris", "200000", "--dist-backend", "gloo", "--setup-only"]
    lines = []
    for run in range(2):
        cmd[cmd.index("--master-port") + 1] = str(free_port())
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=800, cwd=ROOT, env=env)
        assert out.returncode == 0, out.stderr[-2000:]
        lines.append(json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1]))
    first, second = lines
    for d in lines:
        assert d["setup_only"] is True and d["n_gpus"] == 8 and d["rccl_world"] == 8
        assert len(d["setup_seconds"]) == 8 and max(d["setup_seconds"]) < 240          # the driver's budget is 600 s for the whole run
        assert len(set(d["nodes"])) == 1 and len(set(d["tris"])) == 1 and d["tris"][0] == 200000   # every rank holds the same scene
        assert d["records_per_rank"] == ((240 * 135 + 7) // 8) * 64 and d["build"]["lib_sha16"]
    assert first["scene_cache"] == "built" and first["build_seconds"][0] > 0 and first["build_seconds"][1:] == [0.0] * 7
    assert second["scene_cache"] == "hit" and second["build_seconds"] == [0.0] * 8


def test_record_index_table_covers_every_pixel_once():
    sys.path.insert(0, ROOT)
    from tray_racing_amd import dist as D
    for w, h, world in [(52, 44, 2), (1920, 1080, 8), (64, 64, 3), (8, 8, 4)]:
        idx = D.pixel_index_of_records(w, h, world).view(-1)
        valid = idx[idx >= 0]
        assert valid.numel() == w * h and torch.unique(valid).numel() == w * h
        assert idx.numel() == world * D.max_shard_tiles(w, h, world) * 64
