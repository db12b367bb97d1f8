#!/usr/bin/env python3
"""bench.py — Mrays/s, primary rays, bistro-class scene, 1920x1080, CWBVH, N x MI355X.

A step is one pass of the hot path over one 1920x1080 frame of primary rays
(2,073,600 rays generated in-kernel; scene resident in HBM before the timed
region).  With N > 1 (one process per GPU under torch.distributed.run) the
frame's 8x8 tiles are dealt round-robin to the ranks (tile % N == rank), each
rank traces its tiles straight into its block of a gather buffer and ONE in-place
all-gather (RCCL) per batch of `--gather-batch` frames brings the hit records to
every rank, which de-interleaves them into row-major frames (all inside the
timed region).  Total work is fixed as N grows: "scaling": "strong".

Frames are independent units of work, so `--streams` frames are kept in flight
(frame k on HIP stream k % streams with its own buffers): the tail of a frame —
a few slow tiles, ~0.4 ms on this scene however many GPUs share it — overlaps
the next frames instead of idling the GPU.  The `roofline` object is measured
in a separate leg with ONE frame in flight (the kernel's own speed);
`in_flight_kernel_ms` is the average launch duration while frames overlap.

The reference's Bistro asset is absent (assets/large_obj is git-ignored), so
the workload is the seeded procedural bistro-class stand-in with Bistro's
triangle count (3,872,303) and the camera of assets/scenes/bistro.ron.
"""
import argparse
import json
import os
import sys
import time

# concurrent kernels from several HIP streams need as many hardware queues (read at HIP init)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
TRI_BYTES = 48         # device triangle record actually fetched per test
NODE_BYTES = 80
HIT_BYTES = 8


def baseline_metric():
    """The metric string exactly as BASELINE.json spells it."""
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:  # noqa: BLE001
        return "Mrays/s primary rays, Bistro 1920\u00d71080 CWBVH, 1/2/4/8 MI355X"


def usable_cores():
    """Cores this process may actually use: CPU affinity capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except Exception:  # noqa: BLE001
        pass
    return n


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--scene", default="bistro")
    ap.add_argument("--tris", type=int, default=0, help="0 = the scene's reference triangle count")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--preset", default="medium_build",
                    help="builder preset (the reference's --preset names); BASELINE.json's config is medium_build")
    ap.add_argument("--sem", type=int, default=3, help="trx_semantics bits (3 = TRX_SEM_CPU)")
    ap.add_argument("--streams", type=int, default=0,
                    help="frames in flight; 0 = 4 for 1-2 GPUs, 8 beyond (a rank's shard shrinks with N, its "
                         "slowest tile does not); 1 = strictly one frame at a time")
    ap.add_argument("--gather-batch", type=int, default=0,
                    help="N > 1: frames completed by one all-gather; 0 = 8 (a 1080p frame is only 2 MB per rank at N = 8)")
    ap.add_argument("--frames-per-launch", type=int, default=0,
                    help="frames submitted per kernel launch (1..8); 0 = 1 on one GPU, the gather batch beyond")
    ap.add_argument("--roofline-launches", type=int, default=40, help="un-overlapped launches timed for `roofline`")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sim-shards", type=int, default=1, help=argparse.SUPPRESS)
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo stages the shard gather through host memory (lets N ranks share one GPU in tests)")
    ap.add_argument("--dump-frame", default="", help="rank 0 writes the last timed frame (int64 {t, prim} records, "
                                                      ".npy) here, with the scene's flat buffers next to it (tests)")
    return ap.parse_args()


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist

    import tray_racing_amd as T
    from tray_racing_amd import dist as D

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs `python -m torch.distributed.run --nproc-per-node %d bench.py ...`"
                             % (args.gpus, args.gpus))
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    lib = T.load()
    if args.dist_backend == "gloo":
        local_rank = 0  # test mode: every rank drives GPU 0
    if lib.trx_device_count() <= local_rank:
        raise SystemExit("no HIP device %d (libtrx.so has no CPU fallback)" % local_rank)
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")

    w, h = args.width, args.height
    threads = max(1, usable_cores() // world)
    verts, counts = T.gen_scene(args.scene, args.tris, 1)
    t0 = time.time()
    flat = T.flat_build(verts, counts, use_tlas=False, threads=threads, preset=args.preset)
    build_s = time.time() - t0
    eye, look, fov = T.scene_camera(args.scene)
    view = T.view_from_camera(eye, look, fov, w, h)
    scene = T.Scene(flat, device=local_rank)

    n_streams = args.streams if args.streams > 0 else (4 if max(world, args.sim_shards) <= 2 else 8)
    streams = [torch.cuda.Stream() for _ in range(n_streams)]
    shard_img = (rank, world, 0)
    shard_cmp = (rank, world, 1)
    if args.sim_shards > 1:  # development aid: time one rank's shard of an N-way split on one GPU
        shard_img = (0, args.sim_shards, 0)
    n_rays_total = w * h
    # N > 1: frames are gathered in batches of F (one in-place all-gather per F frames: fewer, larger
    # collectives).  Batch b runs on stream b % streams: F kernels back to back, the all-gather, the
    # de-interleave; `streams` batches are in flight, so a gather overlaps the tracing of other batches.
    F = 1 if world == 1 else (args.gather_batch if args.gather_batch > 0 else 8)
    fgs = [D.FrameGather(w, h, rank, world, "cuda", batch=F) for _ in range(n_streams)] if world > 1 else []
    frames = [torch.empty(F * n_rays_total, dtype=torch.int64, device="cuda") for _ in range(n_streams)]
    state = {"k": 0, "batch": 0, "last": None}

    # algorithmic bytes of ONE launch on this rank (counting kernel = the reference's PROFILE_RT counters)
    st = scene.count_primary(view, w, h, sem=args.sem, shard=shard_img)
    launch_bytes = NODE_BYTES * st.n_node + TRI_BYTES * st.n_tri + HIT_BYTES * st.n_rays

    L_launch = args.frames_per_launch if args.frames_per_launch > 0 else (1 if world == 1 else F)
    L_launch = max(1, min(L_launch, 8, F if world > 1 else 8))
    if world == 1 and L_launch > 1:
        frames = [torch.empty(L_launch * n_rays_total, dtype=torch.int64, device="cuda") for _ in range(n_streams)]

    def trace(s, ptr, shard, n, stride):
        """n frames in one launch (n == 1: the plain entry point), bracketed by events on the launching stream."""
        ev0 = torch.cuda.Event(enable_timing=True)
        ev1 = torch.cuda.Event(enable_timing=True)
        ev0.record(s)
        if n == 1:
            scene.trace_primary_dev(view, w, h, ptr, sem=args.sem, shard=shard, stream=s.cuda_stream)
        else:
            scene.trace_primary_batch_dev([view] * n, w, h, ptr, stride, sem=args.sem, shard=shard, stream=s.cuda_stream)
        ev1.record(s)
        return ev0, ev1, n

    def run_frames(n, events):
        """Enqueue n frames (nothing here waits on the GPU)."""
        done = 0
        if world == 1:
            while done < n:
                m = min(L_launch, n - done)
                j = state["k"] % n_streams
                state["k"] += 1
                with torch.cuda.stream(streams[j]):
                    # one GPU owns every tile: the kernel writes the row-major frame(s) directly
                    events.append(trace(streams[j], frames[j].data_ptr(), shard_img, m, n_rays_total))
                state["last"] = frames[j][(m - 1) * n_rays_total: m * n_rays_total]
                done += m
            return
        while done < n:
            m = min(F, n - done)
            j = state["batch"] % n_streams
            fg, s = fgs[j], streams[j]
            with torch.cuda.stream(s):
                for f0 in range(0, m, L_launch):
                    mm = min(L_launch, m - f0)
                    events.append(trace(s, fg.slot(f0, m).data_ptr(), shard_cmp, mm, fg.records))
                if args.dist_backend == "nccl":
                    work = fg.gather(m=m, async_op=True)         # the one collective: 8 B/ray, m frames at once
                    work.wait()                                   # stream s (not the host) waits for it
                else:  # test mode: the same gather staged through host memory
                    s.synchronize()
                    nrec = m * fg.records
                    host = torch.empty(world * nrec, dtype=torch.int64)
                    dist.all_gather_into_tensor(host, fg.flat[rank * nrec:(rank + 1) * nrec].cpu())
                    fg.flat[: world * nrec].copy_(host)
                fg.assemble(frames[j][: m * n_rays_total], m=m)
            state["last"] = frames[j][(m - 1) * n_rays_total: m * n_rays_total]
            state["batch"] += 1
            done += m

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if world > 1:  # gather indices for every batch size the loops below will meet, built outside the timed region
        fgs[0].prepare(min(F, args.steps), args.steps % F, min(F, max(args.warmup, 1)), args.warmup % F)
    # set-up, like the scene upload: every stream's launch slot sees the frame geometry once, so that the
    # tile-order feedback is in its steady state whatever --warmup is
    sync_all()
    run_frames(n_streams * (1 if world == 1 else F), [])
    sync_all()
    run_frames(args.warmup, [])
    sync_all()
    t0 = time.perf_counter()
    events = []
    run_frames(args.steps, events)
    sync_all()
    elapsed = time.perf_counter() - t0
    for s in streams:
        scene.check(s.cuda_stream)
    frame = state["last"]
    in_flight_ms = sum(a.elapsed_time(b) for a, b, _ in events) / len(events)   # per launch

    # roofline leg: the same launch with ONE frame in flight, hipEvents on the launching stream
    s0 = streams[0]
    single = []
    with torch.cuda.stream(s0):
        for _ in range(max(1, args.roofline_launches)):
            ev0 = torch.cuda.Event(enable_timing=True)
            ev1 = torch.cuda.Event(enable_timing=True)
            ev0.record(s0)
            scene.trace_primary_dev(view, w, h, (frames[0] if world == 1 else fgs[0].slot(0, 1)).data_ptr(), sem=args.sem,
                                    shard=(shard_img if world == 1 else shard_cmp), stream=s0.cuda_stream)
            ev1.record(s0)
            single.append((ev0, ev1))
    sync_all()
    kernel_ms = sum(a.elapsed_time(b) for a, b in single) / len(single)

    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax[0])
    rays_per_step = st.n_rays if args.sim_shards > 1 else n_rays_total
    value = rays_per_step * args.steps / elapsed / 1e6

    out = None
    if rank == 0:
        achieved = launch_bytes / (kernel_ms * 1e-3) / 1e9
        traffic, valu_insts = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath) and world == 1 and args.scene == "bistro" and (w, h) == (1920, 1080):
            try:  # PMC counters cannot be read live; these come from the committed rocprofv3 --pmc passes
                tj = json.load(open(tpath))
                traffic, valu_insts = tj.get("hbm_bytes_per_launch"), tj.get("valu_wave_insts_per_launch")
            except Exception:  # noqa: BLE001
                traffic = None
        out = {
            "metric": baseline_metric(),
            "value": round(value, 2),
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "%s-class procedural stand-in, %d tris, %d CWBVH nodes, primary rays %dx%d "
                            "(BASELINE.json configs[2])" % (args.scene, flat.n_tris, flat.n_nodes, w, h),
                "semantics": "TRX_SEM_CPU" if args.sem == 3 else "bits=%d" % args.sem,
                "builder": "binned-SAH BVH2 -> reinsertion pass -> SAH-optimal BVH8 collapse (stands in for obvhs "
                           "ploc_cwbvh), preset %s" % args.preset,
                "parallelism": ("one GPU owns every 8x8 tile" if world == 1 else
                                "8x8 tiles round-robin over %d ranks; hit shards (8 B/ray) all-gathered in place, %d frames "
                                "per collective, and de-interleaved to row-major frames on every rank" % (world, F)),
                "frames_in_flight": n_streams,   # kernels in flight (one per stream)
                "frames_per_launch": L_launch,
                "frames_per_gather": F,
                "build_seconds": round(build_s, 2),
            },
            "roofline": {
                "bound": "hbm",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "kernel_ms": round(kernel_ms, 4),
                "launches": len(single),
                "frames_in_flight": 1,
                "in_flight_kernel_ms": round(in_flight_ms, 4),
                "bytes_per_launch": int(launch_bytes),
                "nodes_per_ray": round(st.n_node / max(st.n_rays, 1), 3),
                "tris_per_ray": round(st.n_tri / max(st.n_rays, 1), 3),
                "valu_issue_frac": (round(valu_insts * 2.0 / (1024 * 2.4e9 * kernel_ms * 1e-3), 3) if valu_insts else None),
                "note": "algorithmic (requested) bytes; coherent rays are served by L1/L2/Infinity Cache, "
                        "see `traffic` (measured HBM bytes per launch) and DESIGN.md section 4",
            },
        }

    if rank == 0 and args.dump_frame:
        np.save(args.dump_frame, frame.detach().cpu().numpy())
        np.savez(args.dump_frame + ".scene.npz", nodes=flat.nodes, tri_verts=flat.tri_verts,
                 instance_offsets=flat.instance_offsets, tlas_start=np.uint32(flat.tlas_start),
                 view=np.frombuffer(bytes(view), dtype=np.uint8), width=np.uint32(w), height=np.uint32(h))
    # CPU baseline: the oracle (a port, not the reference binary) on the host cores, rank 0 at N=1 only; it is
    # the only place this file touches oracle/ (as the thing timed beside the GPU, and as the frame's checker)
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.sim_shards == 1:
        from oracle import binding as O
        osc = O.Scene.from_flat(flat)
        ov = O.view_from_bytes(view)
        cores = usable_cores()
        hits, ost = osc.trace_primary(ov, w, h, sem=args.sem, threads=cores)  # one full frame
        n_frames, secs = 1, ost.seconds
        while secs < args.cpu_seconds and n_frames < 64:
            _, s2 = osc.trace_primary(ov, w, h, sem=args.sem, threads=cores, out=hits)
            n_frames += 1
            secs += s2.seconds
        gpu = D.int64_to_hits(frame)
        parity = bool((gpu["t"].view(np.uint32) == hits["t"].view(np.uint32)).all() and
                      (gpu["prim"] == hits["prim"]).all())
        # the reference's own CPU figure is a whole frame: ray generation + primary + one AO ray per hit pixel +
        # shading (src/rt_cpu/rt_cpu.rs:35-92,98,113); two frames of that, with the GPU's primary + AO frame beside it
        frame_s = min(osc.render_frame(ov, w, h, sem=args.sem, frame=f, ao_eps=0.01, threads=cores) for f in range(2))
        gpu_frame_ms = min(scene.trace_primary_ao(view, w, h, sem=args.sem, frame=f, ao_eps=0.01)[2] for f in range(4))
        out["cpu_baseline"] = {
            "value": round(n_rays_total * n_frames / secs / 1e6, 3),
            "unit": "Mrays/s",
            "cores": cores,
            "kind": "port",
            "sample": "%d full %dx%d frame(s) of the same workload, %.1f s, OpenMP over 8x8 tiles" % (
                n_frames, w, h, secs),
            "reference_style_frame_ms": round(frame_s * 1e3, 2),       # primary + AO + shade, wall clock
            "gpu_primary_ao_frame_ms": round(gpu_frame_ms, 3),          # the same rays on the GPU (kernel time)
        }
        out["parity_vs_oracle_full_frame"] = parity
        if not parity:
            print("WARNING: GPU frame differs from the oracle frame", file=sys.stderr)
    if rank == 0:
        print(json.dumps(out), flush=True)
    scene.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
