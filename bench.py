#!/usr/bin/env python3
"""bench.py — Mrays/s, primary rays, bistro-class scene, 1920x1080, CWBVH, N x MI355X.

A step is one pass of the hot path over one 1920x1080 frame of primary rays (2,073,600 rays generated
in-kernel; scene resident in HBM before the timed region).

N = 1 measures what SURVEY.md 8(d) defines, the way the reference measures its GPU path
(src/rt_gpu/rt_gpu_software.rs:289-302,339-344,376): ONE frame in flight — the K timed launches go back to back
on one HIP stream, ONE hipEvent pair on that stream around all of them — after `--wake-frames` untimed frames that take
the GPU out of its idle clocks and the W warm-up launches.  `value` is rays * K / wall-clock of the timed region;
`kernel_ms_mean` is that event pair / K (the average launch duration, gaps included); `kernel_ms_min` and
`kernel_ms_per_step` are per-launch event pairs of the same frames run once more right after the timed region (an event
pair per launch inside it costs about 5 us of idle GPU per frame).  Beside it, as separate legs that never enter
`value`: the reference's own protocol (3 passes x [3 discarded + 20 timed frames], min and mean), the first
frame of a geometry (tile order not yet learnt: `cold_order_ms`), the literal-HLSL arithmetic
(`sem_hlsl_ms`), frames overlapped on 4 streams (`pipelined_mrays`), a measured HBM copy ceiling, live
rocprofv3 counter passes of the same workload (child processes) and the CPU port on the host cores.

N > 1 (one process per GPU; under `python -m torch.distributed.run` as the driver launches it, or plain
`python bench.py --gpus N`, which starts those ranks itself before it touches the GPU): the frame's 8x8 tiles are dealt round-robin to the
ranks (tile % N == rank), each rank traces its tiles straight into its block of a gather buffer and ONE
in-place all-gather (RCCL) per batch of `--gather-batch` frames brings the hit records to every rank, which
de-interleaves them into row-major frames — all inside the timed region.  Total work is fixed as N grows:
"scaling": "strong".  Per-phase times (trace / gather / assemble) are reported per frame.

The reference's Bistro asset is absent (assets/large_obj is git-ignored), so the workload is the seeded
procedural bistro-class stand-in with Bistro's triangle count (3,872,303) and the camera of
assets/scenes/bistro.ron.
"""
import argparse
import csv
import glob
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

# read by the HIP / HSA runtimes when they initialise, so set before anything touches the GPU
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")          # concurrent kernels from several streams (pipelined leg)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # RCCL / cross-process sharing on this driver

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec); 6.29 TB/s measured there
TRI_BYTES = 48         # device triangle record actually fetched per test
NODE_BYTES = 80
HIT_BYTES = 8
SIMDS = 1024           # 256 CUs x 4 SIMD-32
CLOCK_GHZ = 2.4        # max engine clock
VALU_CYCLES = 2.0      # a wave64 VALU instruction issues over 2 cycles on a SIMD-32 (same guide, "Wave scheduling")
# measured on this part (tools/ubench/valu_rates2.hip, profiles/r02_ubench_valu_rates2.log): at 4 waves per SIMD the
# instruction classes the node test is made of (v_cvt_f32_ubyte, v_pk_mul/add_f32, v_max3/min3, v_cndmask, shifts,
# SDWA, and scalar ALU alike) issue at one per 2.63 cycles per SIMD; only v_fma/mul/add_f32, v_and, v_add_u32 reach 1.62
ISSUE_CYCLES_MEASURED = 2.63
VARIANT_COLD = 1 << 20  # trx_set_kernel_variant: tile-order feedback off
VARIANT_CUT = 1 << 7    # ... every frame runs as the first frame of its geometry: natural order while the tiles are measured
# static VALU instruction counts of the two tests in the shipped primary kernel (llvm-objdump of k_trace<0,false,1,...>:
# the node-test block is 213 vector instructions, one per-lane triangle round - bit select, address, test, commit - 70);
# roofline.useful_frac prices the COUNTED lane-level tests at these, i.e. what a divergence-free walk would issue
NODE_TEST_VALU = 213
TRI_TEST_VALU = 70
L1_BYTES_PER_CLK_CU = 64.0  # vector L1 (TCP) data path per CU and clock (guide, cache hierarchy)
CUS = 256


def baseline_metric():
    """The metric string exactly as BASELINE.json spells it."""
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:  # noqa: BLE001
        return "Mrays/s primary rays, Bistro 1920×1080 CWBVH, 1/2/4/8 MI355X"


def build_stamp():
    """What produced this line: a hash of the library that ran (tray_racing_amd/libtrx.so travels to the GPU box as built),
    of this file, and the commit the library was built at (`make` writes tray_racing_amd/.build_id where git is available -
    the GPU box's snapshot has no .git).  tests/test_bench_contract.py holds the committed lines of a round to ONE library."""
    def sha16(path):
        try:
            h = hashlib.sha256()
            with open(path, "rb") as f:
                for blk in iter(lambda: f.read(1 << 20), b""):
                    h.update(blk)
            return h.hexdigest()[:16]
        except OSError:
            return None
    lib_path = os.environ.get("TRX_LIB") or os.path.join(ROOT, "tray_racing_amd", "libtrx.so")
    head = None
    try:
        head = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True, timeout=10).stdout.strip() or None
    except Exception:  # noqa: BLE001
        head = None
    built_at = None
    try:
        built_at = open(os.path.join(ROOT, "tray_racing_amd", ".build_id")).read().strip() or None
    except OSError:
        pass
    return {"lib_sha16": sha16(lib_path), "bench_py_sha16": sha16(os.path.abspath(__file__)), "git_head": head, "lib_built_at": built_at}


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


def cpu_model():
    """Host CPU as /proc/cpuinfo names it (the cpu_baseline leg says what it was timed on)."""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--scene", default="bistro")
    ap.add_argument("--input", default="", help="a scene file of the reference (assets/scenes/<name>.ron) or a model (.obj / .json "
                                                 "triangle list) instead of the procedural stand-in: `data` becomes \"real\"; the "
                                                 "camera comes from the scene file (a bare model uses --scene's camera)")
    ap.add_argument("--builder", default="preset", choices=["preset", "ploc_gpu"],
                    help="preset: binned-SAH BVH2 -> reinsertion -> collapse under --preset (the default); ploc_gpu: the "
                         "reference-default ploc_cwbvh pipeline (BvhBuildParams of src/main.rs:571-585) with its stages on the GPU")
    ap.add_argument("--no-scene-cache", action="store_true",
                    help="N > 1: every rank builds the scene itself (default: rank 0 builds with all cores and the others read "
                         "its flat buffers from the temporary directory)")
    ap.add_argument("--setup-only", action="store_true",
                    help="N > 1 rehearsal without a GPU: rendezvous, scene build / cache hand-over, gather geometry, then one "
                         "JSON line with the per-rank set-up times (tests/test_dist_gloo.py runs it at world 8)")
    ap.add_argument("--tris", type=int, default=0, help="0 = the scene's reference triangle count")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--preset", default="medium_build",
                    help="builder preset (the reference's --preset names); BASELINE.json's config is medium_build")
    ap.add_argument("--sem", type=int, default=3, help="trx_semantics bits (3 = TRX_SEM_CPU)")
    ap.add_argument("--streams", type=int, default=0,
                    help="frames in flight inside the timed region; 0 = 1 on one GPU (the SURVEY 8(d) metric), 4 for "
                         "2 GPUs, 8 beyond (a rank's shard shrinks with N, its slowest tile does not)")
    ap.add_argument("--gather-batch", type=int, default=0,
                    help="N > 1: frames completed by one all-gather; 0 = 8 (a 1080p frame is only 2 MB per rank at N = 8)")
    ap.add_argument("--frames-per-launch", type=int, default=0,
                    help="frames submitted per kernel launch (1..8); 0 = 1 on one GPU, the gather batch beyond")
    ap.add_argument("--wake-frames", type=int, default=64,
                    help="untimed frames of the same workload BEFORE the warm-up steps: the GPU idles while the host builds the "
                         "scene and its clocks take about 20 ms of load to come back up (profiles/r04_clock_ramp.log: the same "
                         "kernel runs 5-7 %% slower for its first ~40 frames after an idle second, whatever the tile order); 0 = "
                         "none, the warm-up steps then run on a GPU at idle clocks")
    ap.add_argument("--repeats", type=int, default=7,
                    help="the timed region is run this many times in all; `value` is the FIRST (the contract's K steps), "
                         "legs.timed_region_repeats carries median / min / max over all of them")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the live rocprofv3 counter passes (child processes)")
    ap.add_argument("--no-legs", action="store_true", help="skip the secondary legs (cold order, HLSL, pipelined, HBM copy)")
    ap.add_argument("--sim-shards", type=int, default=1, help=argparse.SUPPRESS)
    ap.add_argument("--gather", default="torch", choices=["torch", "abi"],
                    help="N > 1: the shard gather through torch.distributed (default) or through the C ABI "
                         "(trx_comm_* / trx_gather_shards / trx_assemble_frames: RCCL driven by libtrx.so itself)")
    ap.add_argument("--gather-to", default="all", choices=["all", "root"],
                    help="N > 1: every rank gets every frame (in-place all-gather, default) or only rank 0 does (ncclSend / "
                         "ncclRecv through trx_gather_shards_root with --gather abi, torch.distributed.gather otherwise: "
                         "1/N of the bytes when one GPU consumes the frames)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo stages the shard gather through host memory (lets N ranks share one GPU in tests)")
    ap.add_argument("--kernel-variant", default="0", help="trx_set_kernel_variant word (tuning / A-B runs; 0 = the product's defaults)")
    ap.add_argument("--dump-frame", default="", help="rank 0 writes the last timed frame (int64 {t, prim} records, "
                                                      ".npy) here, with the scene's flat buffers next to it (tests)")
    return ap.parse_args()


# ---- live counters: rocprofv3 --pmc over a child process that replays the same scene -------------------------

PMC_GROUPS = [
    "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS",
    "FETCH_SIZE",
    "WRITE_SIZE",
    "VALUBusy VALUUtilization SALUBusy",   # rocprofv3's derived metrics (per cent), reported as they come
    "TA_TA_BUSY_sum GRBM_GUI_ACTIVE",      # texture-address / L1 front end: busy cycles summed over the CUs
    "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum",   # L1 tag look-ups and what they sent on to L2
]


def pmc_passes(scene_npz, frames=10, timeout_s=150):
    """One rocprofv3 --pmc pass per counter group (separate passes, --kernel-trace only beside them), averaged
    per launch of the traversal kernel.  Returns ({counter: value per launch}, kernel_ms under the profiler)
    or (None, reason)."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    out, dur = {}, []
    tmp = tempfile.mkdtemp(prefix="trx_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        for i, grp in enumerate(PMC_GROUPS):
            d = os.path.join(tmp, "g%d" % i)
            cmd = [exe, "--kernel-trace", "--pmc"] + grp.split() + ["--output-format", "csv", "-d", d, "--",
                                                                     sys.executable, os.path.join(ROOT, "tools", "pmc_child.py"),
                                                                     scene_npz, str(frames)]
            try:
                r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
            except subprocess.TimeoutExpired:
                if i >= 3:
                    continue   # the derived metrics are an extra: the roofline does not depend on them
                return None, "rocprofv3 pass %d timed out" % i
            if r.returncode != 0:
                if i >= 3:
                    continue
                return None, "rocprofv3 pass %d rc %d: %s" % (i, r.returncode, r.stderr[-200:])
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if not files:
                if i >= 3:
                    continue
                return None, "rocprofv3 pass %d wrote no counter csv" % i
            tot, seen = {}, {}
            for f in files:
                for row in csv.DictReader(open(f)):
                    if "k_trace<0" not in row["Kernel_Name"]:
                        continue
                    c = tot.setdefault(row["Counter_Name"], [0.0, 0])
                    c[0] += float(row["Counter_Value"])
                    c[1] += 1
                    seen[row["Dispatch_Id"]] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-6
            # the first dispatches are warm-up (cold tile order); every dispatch counts the same instructions, so
            # averaging over all of them is exact for instruction counts and a slight over-estimate for bytes
            for k, (s, n) in tot.items():
                out[k] = s / max(n, 1)
            if i == 0 and seen:
                v = sorted(seen.values())
                dur = v[: max(1, len(v) // 2)]  # the faster half = the steady state
        return out, (sum(dur) / len(dur) if dur else None)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def entry_nodes(flat):
    """The entry-node table of a re-braided TLAS (None for every BLAS-only scene) travels with the flat buffers: a scene
    restored without it would enter every BLAS at its root (same hits, many more node visits)."""
    ent = getattr(flat, "instance_entry", None)
    return {} if ent is None else {"instance_entry": ent}


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves.  This process has not touched the
    GPU (nothing below `import` level does; libtrx.so is not even loaded yet), so the ranks are FRESH children of
    `python -m torch.distributed.run` - never an exec of a process that initialised HIP - one per GPU, rendezvous on
    127.0.0.1.  Their stdout / stderr are ours (rank 0 prints the JSON line); the launcher's return code is ours."""
    env = dict(os.environ)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    # (--standalone: the launcher binds its own rendezvous port - a port picked here and closed again could be taken by
    # another bench starting at the same moment)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           "--nproc-per-node", str(n), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args.gpus))
    import numpy as np
    import torch
    import torch.distributed as dist

    import tray_racing_amd as T
    from tray_racing_amd import dist as D

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    lib = T.load()
    lib.trx_set_kernel_variant(int(args.kernel_variant, 0))
    if args.dist_backend == "gloo":
        local_rank = 0  # test mode: every rank drives GPU 0
    if not args.setup_only:
        if lib.trx_device_count() <= local_rank:
            raise SystemExit("no HIP device %d (libtrx.so has no CPU fallback)" % local_rank)
        torch.cuda.set_device(local_rank)
    elif world == 1 or args.dist_backend != "gloo":
        raise SystemExit("--setup-only rehearses the N > 1 set-up on CPUs: it needs --gpus N > 1 and --dist-backend gloo")
    if world > 1:
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")

    w, h = args.width, args.height
    threads = max(1, usable_cores() // world)
    t_setup0 = time.time()

    def load_input():
        """(verts, counts, camera) of --input or of the procedural stand-in."""
        if args.input:
            if args.input.endswith(".ron"):
                v, c, e, l, f = T.load_scene(args.input)
                return v, c, (e, l, f)
            v, c = T.load_meshs(args.input)
            return v, c, T.scene_camera(args.scene)
        v, c = T.gen_scene(args.scene, args.tris, 1)
        return v, c, T.scene_camera(args.scene)

    def build_flat(v, c, nthreads):
        if args.builder == "ploc_gpu":   # the preset's build with every stage on the device (trx_flat_build_preset_device)
            return T.flat_build_preset_device(v, c, preset=args.preset, device=local_rank, threads=nthreads)
        return T.flat_build(v, c, use_tlas=False, threads=nthreads, preset=args.preset)

    stamp = build_stamp()
    verts = counts = None
    scene_cache = "off"
    if world == 1 or args.no_scene_cache:
        verts, counts, (eye, look, fov) = load_input()
        t0 = time.time()
        flat = build_flat(verts, counts, threads)
        build_s = time.time() - t0
    else:
        # N > 1: ONE build, by rank 0 on every core this job may use (eight identical builds on an eighth of the cores each
        # is what rounds 1-5 did), handed to the other ranks as flat buffers in the temporary directory of the node - keyed
        # by input, builder and library, so the next run of the same job on this node (the driver goes N = 2, 4, 8) finds it
        ident = json.dumps([args.input or args.scene, os.path.getmtime(args.input) if args.input else 0, args.tris, args.preset,
                            args.builder, stamp["lib_sha16"]])
        cdir = os.path.join(tempfile.gettempdir(), "trx_bench_scene_" + hashlib.sha256(ident.encode()).hexdigest()[:20])
        build_s = 0.0
        if rank == 0:
            if os.path.exists(os.path.join(cdir, "meta.json")):
                scene_cache = "hit"
            else:
                verts, counts, cam = load_input()
                t0 = time.time()
                flat0 = build_flat(verts, counts, usable_cores())
                build_s = time.time() - t0
                tmp = tempfile.mkdtemp(prefix="trx_bench_scene_tmp_", dir=tempfile.gettempdir())
                np.save(os.path.join(tmp, "nodes.npy"), flat0.nodes)
                np.save(os.path.join(tmp, "tri_verts.npy"), flat0.tri_verts)
                np.save(os.path.join(tmp, "instance_offsets.npy"), flat0.instance_offsets)
                json.dump({"tlas_start": int(flat0.tlas_start), "camera": [list(map(float, cam[0])), list(map(float, cam[1])), float(cam[2])],
                           "build_seconds": build_s}, open(os.path.join(tmp, "meta.json"), "w"))
                try:
                    os.replace(tmp, cdir)
                except OSError:   # another job of the same kind got there first: its copy is as good
                    shutil.rmtree(tmp, ignore_errors=True)
                scene_cache = "built"
                del flat0
        dist.barrier()
        meta = json.load(open(os.path.join(cdir, "meta.json")))
        flat = T.FlatScene(np.load(os.path.join(cdir, "nodes.npy")), np.load(os.path.join(cdir, "tri_verts.npy")),
                           np.load(os.path.join(cdir, "instance_offsets.npy")), meta["tlas_start"], np.zeros(0, np.uint32), np.zeros(1, np.uint32))
        eye, look, fov = meta["camera"]
        if rank != 0:
            scene_cache = "read"
    view = T.view_from_camera(eye, look, fov, w, h)
    if args.setup_only:
        # the rehearsal of the N-rank set-up on hosts without a GPU: the gather geometry for this world, then the times
        fg = D.FrameGather(w, h, rank, world, "cpu", batch=args.gather_batch if args.gather_batch > 0 else 8)
        fg.prepare(min(fg.batch, args.steps), args.steps % fg.batch)
        mine = [time.time() - t_setup0, build_s, float(flat.n_nodes), float(flat.n_tris)]
        allv = [None] * world
        dist.all_gather_object(allv, mine)
        if rank == 0:
            print(json.dumps({"setup_only": True, "n_gpus": world, "rccl_world": dist.get_world_size(), "scene_cache": scene_cache,
                              "setup_seconds": [round(x[0], 2) for x in allv], "build_seconds": [round(x[1], 2) for x in allv],
                              "nodes": [int(x[2]) for x in allv], "tris": [int(x[3]) for x in allv], "records_per_rank": fg.records,
                              "build": stamp}), flush=True)
        dist.destroy_process_group()
        return
    scene = T.Scene(flat, device=local_rank)

    shards = max(world, args.sim_shards)
    n_streams = args.streams if args.streams > 0 else (1 if shards == 1 else 4 if shards == 2 else 8)
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
    if world > 1 and args.gather == "abi" and args.dist_backend == "nccl":
        # ONE communicator per rank: the streams' buffers share it, their collectives chained by an event (a communicator's
        # collectives must be issued in one order on every rank; the batches are dealt to the streams in that order anyway).
        # Rounds 1-5 created one communicator per stream - eight per rank at N = 8 - for an overlap of gathers that are
        # ~50 us each: never measured, and eight times the set-up.
        fg0 = D.AbiFrameGather.from_process_group(w, h, torch.device("cuda", local_rank), batch=F)
        fgs = [fg0] + [D.AbiFrameGather.sharing(fg0) for _ in range(n_streams - 1)]
    elif world > 1 and args.gather == "abi":
        # test mode (gloo): the shards are staged through host memory below, the frames are assembled by the ABI's kernel
        fgs = [D.AbiFrameGather.without_communicator(w, h, rank, world, torch.device("cuda", local_rank), batch=F)
               for _ in range(n_streams)]
    else:
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

    def ev():
        return torch.cuda.Event(enable_timing=True)

    def trace(s, ptr, shard, n, stride, timed=True):
        """n frames in one launch (n == 1: the plain entry point), bracketed by events on the launching stream when `timed`."""
        ev0, ev1 = (ev(), ev()) if timed else (None, None)
        if timed:
            ev0.record(s)
        if n == 1:
            scene.trace_primary_dev(view, w, h, ptr, sem=args.sem, shard=shard, stream=s.cuda_stream)
        else:
            scene.trace_primary_batch_dev([view] * n, w, h, ptr, stride, sem=args.sem, shard=shard, stream=s.cuda_stream)
        if timed:
            ev1.record(s)
        return ev0, ev1, n

    def run_frames(n, events, phases=None):
        """Enqueue n frames on the chosen streams (nothing here waits on the GPU)."""
        done = 0
        if world == 1:
            while done < n:
                m = min(L_launch, n - done)
                j = state["k"] % n_streams
                state["k"] += 1
                with torch.cuda.stream(streams[j]):
                    # one GPU owns every tile: the kernel writes the row-major frame(s) directly
                    e = trace(streams[j], frames[j].data_ptr(), shard_img, m, n_rays_total, timed=events is not None)
                    if events is not None:
                        events.append(e)
                state["last"] = frames[j][(m - 1) * n_rays_total: m * n_rays_total]
                done += m
            return
        while done < n:
            m = min(F, n - done)
            j = state["batch"] % n_streams
            fg, s = fgs[j], streams[j]
            with torch.cuda.stream(s):
                if args.dist_backend == "gloo":   # test mode: a launch that leaves a record unwritten must not go unnoticed
                    fg.flat[rank * m * fg.records:(rank + 1) * m * fg.records].fill_(-1)
                for f0 in range(0, m, L_launch):
                    mm = min(L_launch, m - f0)
                    events.append(trace(s, fg.slot(f0, m).data_ptr(), shard_cmp, mm, fg.records))
                e_g0, e_g1, e_a1 = ev(), ev(), ev()
                e_g0.record(s)
                to_root = args.gather_to == "root"
                nrec = m * fg.records
                if args.dist_backend == "nccl" and to_root and args.gather == "abi":
                    fg.gather(m=m, root=0)                        # world - 1 sends into rank 0, nothing comes back
                elif args.dist_backend == "nccl" and to_root:
                    mine = fg.flat[rank * nrec:(rank + 1) * nrec]
                    dist.gather(mine, [fg.flat[r * nrec:(r + 1) * nrec] for r in range(world)] if rank == 0 else None, dst=0)
                elif args.dist_backend == "nccl":
                    work = fg.gather(m=m, async_op=True)         # the one collective: 8 B/ray, m frames at once
                    if work is not None:
                        work.wait()                               # stream s (not the host) waits for it
                else:  # test mode: the same gather staged through host memory
                    s.synchronize()
                    mine = fg.flat[rank * nrec:(rank + 1) * nrec].cpu()
                    if (w % 8 == 0 and h % 8 == 0 and (w // 8) * (h // 8) % world == 0) and bool((mine == -1).any()):
                        raise SystemExit("rank %d batch %d (%d frames, stream %d): %d of %d records of the shard were never "
                                         "written" % (rank, state["batch"], m, j, int((mine == -1).sum()), nrec))
                    if to_root:
                        parts = [torch.empty(nrec, dtype=torch.int64) for _ in range(world)] if rank == 0 else None
                        dist.gather(mine, parts, dst=0)
                        if rank == 0:
                            fg.flat[: world * nrec].copy_(torch.cat(parts))
                    else:
                        parts = [torch.empty(nrec, dtype=torch.int64) for _ in range(world)]
                        dist.all_gather(parts, mine)
                        fg.flat[: world * nrec].copy_(torch.cat(parts))
                    s.synchronize()   # the staging tensors are pageable and die with this scope: the copy must have left them
                e_g1.record(s)
                if not to_root or rank == 0:
                    fg.assemble(frames[j][: m * n_rays_total], m=m)
                e_a1.record(s)
                if phases is not None:
                    phases.append((e_g0, e_g1, e_a1, m))
            state["last"] = frames[j][(m - 1) * n_rays_total: m * n_rays_total]
            state["batch"] += 1
            done += m

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if world > 1:  # gather indices for every batch size the loops below will meet, built outside the timed region
        fgs[0].prepare(min(F, args.steps), args.steps % F, min(F, max(args.warmup, 1)), args.warmup % F,
                       min(F, max(args.wake_frames, 1)), args.wake_frames % F)
    # set-up, like the scene upload: every stream's launch slot sees the frame geometry once (its first frame
    # runs in natural tile order and measures the tiles: reported separately as cold_order_ms)
    sync_all()
    run_frames(n_streams * (1 if world == 1 else F), [])
    sync_all()
    setup_s = time.time() - t_setup0   # scene (generated / loaded / built / read), upload, communicators, first frames
    if args.wake_frames > 0:   # bring the GPU out of its idle clocks (see --wake-frames); reported in config.wake_frames
        run_frames(args.wake_frames, [])
        sync_all()
    run_frames(args.warmup, [])
    sync_all()
    # One GPU, one frame in flight (the default): the K launches go back to back with ONE hipEvent pair around all of them
    # on their stream - an event pair per launch puts two markers between consecutive kernels and costs the timed region
    # about 5 us of idle GPU per frame (measured: 0.4293 ms per step against 0.4224 ms of kernel).  The per-launch series
    # (kernel_ms_min, kernel_ms_per_step) comes from a second, untimed pass of the same frames right after.
    one_pair = world == 1 and n_streams == 1
    t0 = time.perf_counter()
    events, phases = [], []
    if one_pair:
        r0, r1 = ev(), ev()
        r0.record(streams[0])
        run_frames(args.steps, None)
        r1.record(streams[0])
    else:
        run_frames(args.steps, events, phases)
    sync_all()
    elapsed = time.perf_counter() - t0
    for s in streams:
        scene.check(s.cuda_stream)
    frame = state["last"]
    region_kernel_ms = None
    if one_pair:
        region_kernel_ms = r0.elapsed_time(r1) / args.steps   # average launch duration over the timed region, gaps included
        run_frames(min(args.steps, 64), events)               # untimed: the same frames again, an event pair each
        sync_all()
    launch_ms = [a.elapsed_time(b) / n for a, b, n in events]   # per frame of each launch
    # The timed region again, REPEATS - 1 more times (same K steps, same barriers; never part of `value`): a K = 20 region at
    # N = 8 is about a millisecond of wall clock, so the line carries the median and the spread of the repetitions next to
    # the one region the contract defines.
    repeat_s = [elapsed]
    for _ in range(max(args.repeats, 1) - 1):
        sync_all()
        tr = time.perf_counter()
        run_frames(args.steps, None if one_pair else [])
        sync_all()
        repeat_s.append(time.perf_counter() - tr)

    if world > 1:
        tmax = torch.tensor(repeat_s, dtype=torch.float64, device="cuda" if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        repeat_s = [float(x) for x in tmax]
        elapsed = repeat_s[0]
    # N > 1 runs more frames in flight and more frames per launch than the N = 1 metric does, and that protocol alone is
    # worth something on ONE GPU (independent frames overlap each other's tails): rank 0 traces WHOLE frames on its GPU under
    # this run's streams / frames-per-launch, no gather, while the other ranks wait - the figure a scaling factor has to be
    # quoted against (`scaling_vs_same_protocol`), next to the driver's own N = 1 line.
    n1_same = None
    if world > 1 and args.sim_shards == 1:
        if rank == 0:
            solo = [torch.empty(L_launch * n_rays_total, dtype=torch.int64, device="cuda") for _ in range(n_streams)]

            def run_solo(n):
                done, k = 0, 0
                while done < n:
                    m = min(L_launch, n - done)
                    j = k % n_streams
                    k += 1
                    with torch.cuda.stream(streams[j]):
                        trace(streams[j], solo[j].data_ptr(), (0, 1, 0), m, n_rays_total, timed=False)
                    done += m
            run_solo(n_streams * L_launch + max(args.warmup, 16))   # every slot learns the whole-frame geometry; warm-up
            torch.cuda.synchronize()
            ts = []
            for _ in range(3):
                tr = time.perf_counter()
                run_solo(args.steps)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - tr)
            n1_same = n_rays_total * args.steps / sorted(ts)[1] / 1e6
            for s in streams:
                scene.check(s.cuda_stream)
            del solo
        sync_all()
    kernel_ms_ranks = None
    if world > 1:  # every rank's mean kernel time per frame: the spread says how even the tile deal was
        mine = torch.tensor([sum(launch_ms) / max(len(launch_ms), 1)], dtype=torch.float64,
                            device="cuda" if args.dist_backend == "nccl" else "cpu")
        allk = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allk, mine)
        kernel_ms_ranks = [float(x[0]) for x in allk]
    setup_ranks, build_ranks, cache_ranks = [setup_s], [build_s], [scene_cache]
    if world > 1:
        allv = [None] * world
        dist.all_gather_object(allv, [setup_s, build_s, scene_cache, time.time() - t_setup0])
        setup_ranks, build_ranks, cache_ranks = [x[0] for x in allv], [x[1] for x in allv], [x[2] for x in allv]
        wall_ranks = [x[3] for x in allv]
    else:
        wall_ranks = [time.time() - t_setup0]
    rays_per_step = st.n_rays if args.sim_shards > 1 else n_rays_total
    value = rays_per_step * args.steps / elapsed / 1e6
    kernel_ms = region_kernel_ms if region_kernel_ms is not None else sum(launch_ms) / len(launch_ms)

    out = None
    legs = {}
    if rank == 0 and world == 1 and args.sim_shards == 1 and not args.no_legs:
        ctx = {}  # what later legs need of earlier ones (the bench frame's primary hits on the device, ...)
        variant0 = int(args.kernel_variant, 0)

        def run_leg(name, fn):
            """Every leg on its own: one that fails - a scene that does not fit, a build stage that errors - leaves
            legs[name] = {"error": ...} and neither the headline nor the legs behind it; the process-wide switches a leg may
            have set (kernel variant, builder device / batching / preset) are put back whatever happened."""
            try:
                fn()
            except Exception as e:  # noqa: BLE001
                legs[name] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
            finally:
                lib.trx_set_kernel_variant(variant0)
                lib.trx_set_build_device(-1)
                lib.trx_set_build_reinsertion_batches(0)
                lib.trx_set_build_preset(args.preset.encode())

        def timed(fn, reps=12, skip=3):
            """hipEvent time of fn(i) on the default stream, one call in flight: (min, mean) over `reps` after `skip`."""
            ts = []
            for i in range(reps + skip):
                a, b = ev(), ev()
                a.record()
                fn(i)
                b.record()
                torch.cuda.synchronize()
                if i >= skip:
                    ts.append(a.elapsed_time(b))
            return min(ts), sum(ts) / len(ts)

        def fetch_vs_random(sc, n_node, n_tri, ms):
            """north_star's "node-fetch loop against a MEASURED roofline" for an incoherent pass: what the pass asks for
            per second (counted node steps x 80 B + triangle tests x 48 B over its kernel time) against what this GPU
            serves when every lane of the same grid fetches UNIFORMLY RANDOM nodes and triangle records of the same
            scene in the same proportion and does nothing else (trx_debug_fetch_rate).  Not an upper bound: a walk's
            upper tree levels stay in L1 / L2 and a leaf's triangles share lines, the probe's fetches do neither - a
            ratio above 1 says the pass runs beyond the no-locality rate of the memory system, i.e. on its caches."""
            nps, tps = sc.fetch_rate(tris_per_node=n_tri / max(n_node, 1))
            rnd = NODE_BYTES * nps + TRI_BYTES * tps
            ach = (NODE_BYTES * n_node + TRI_BYTES * n_tri) / (ms * 1e-3)
            return {"requested_gbs": round(ach / 1e9, 1), "random_fetch_gbs": round(rnd / 1e9, 1), "ratio": round(ach / rnd, 3),
                    "nodes_per_s_g": round(n_node / (ms * 1e-3) / 1e9, 2), "tris_per_s_g": round(n_tri / (ms * 1e-3) / 1e9, 2),
                    "random_nodes_per_s_g": round(nps / 1e9, 2), "random_tris_per_s_g": round(tps / 1e9, 2)}

        def primary_hits():
            """The bench frame's primary hit records on the device (shared by the AO legs) and how many of them are hits."""
            if "d_prim" not in ctx:
                d = torch.empty(n_rays_total, dtype=torch.int64, device="cuda")
                scene.trace_primary_dev(view, w, h, d.data_ptr(), sem=args.sem)
                torch.cuda.synchronize()
                ctx["d_prim"] = d
                ctx["n_ao"] = int(((d & 0xffffffff) != 0x7f800000).sum().item())   # low word = t bits; +inf = miss
                ctx["d_ao"] = torch.empty(n_rays_total, dtype=torch.int64, device="cuda")
            return ctx["d_prim"], ctx["d_ao"], ctx["n_ao"]

        def leg_reference_protocol():
            # (a) the reference's protocol: 3 passes x [3 discarded + 20 frames], hipEvent pair per frame, min and mean
            passes = [scene.bench_primary(view, w, h, sem=args.sem, warmup=3, frames=20) for _ in range(3)]
            legs["reference_protocol"] = {
                "passes": 3, "discarded_frames": 3, "frames": 20,
                "min_ms": round(sum(p[0] for p in passes) / 3, 4), "mean_ms": round(sum(p[1] for p in passes) / 3, 4),
                "mrays_at_min": round(n_rays_total / (sum(p[0] for p in passes) / 3) / 1e3, 1),
            }

        def leg_cold_order():
            # (b) tile-order feedback off: what the first frame of a geometry (or a caller that never repeats one) gets
            lib.trx_set_kernel_variant(VARIANT_COLD)
            cmin, cmean = scene.bench_primary(view, w, h, sem=args.sem, warmup=3, frames=20)
            legs["cold_order_ms"] = {"min": round(cmin, 4), "mean": round(cmean, 4)}

        def leg_sem_hlsl():
            # (c) the literal HLSL arithmetic (per-node IEEE divides, tt <= t)
            hmin, hmean = scene.bench_primary(view, w, h, sem=0, warmup=3, frames=20)
            legs["sem_hlsl_ms"] = {"min": round(hmin, 4), "mean": round(hmean, 4)}

        def leg_first_frame():
            # (c') every frame runs as the first frame of its image geometry (natural order while its tiles are measured; the
            #      probe pass that once predicted an order here was measured and removed)
            lib.trx_set_kernel_variant(VARIANT_CUT)
            fmin, fmean = scene.bench_primary(view, w, h, sem=args.sem, warmup=3, frames=20)
            legs["first_frame_ms"] = {"min": round(fmin, 4), "mean": round(fmean, 4)}

        def leg_ao_pass():
            # (c") the AO pass over this frame's primary hits (the reference's second ray per pixel, rt_gpu_software.hlsl:105-128):
            #      one cosine-weighted ray per primary hit, a new noise seed every pass; default stream, hipEvents per launch
            d_prim, d_ao, n_ao = primary_hits()
            ao_ev = []
            for k in range(3 + 12):
                a, b = ev(), ev()
                a.record()
                scene.trace_ao_dev(view, w, h, d_prim.data_ptr(), d_ao.data_ptr(), sem=args.sem, frame=k, ao_eps=0.01)
                b.record()
                ao_ev.append((a, b))
            torch.cuda.synchronize()
            at = [a.elapsed_time(b) for a, b in ao_ev][3:]
            legs["ao_pass_ms"] = {"rays": n_ao, "frames": len(at), "min": round(min(at), 4), "mean": round(sum(at) / len(at), 4),
                                  "mrays_at_mean": round(n_ao / (sum(at) / len(at)) / 1e3, 1)}
            ao_st = scene.count_ao(view, w, h, d_prim.data_ptr(), d_ao.data_ptr(), sem=args.sem, frame=0, ao_eps=0.01)
            legs["ao_pass_ms"]["nodes_per_ray"] = round(int(ao_st.n_node) / max(n_ao, 1), 2)
            legs["ao_pass_ms"]["tris_per_ray"] = round(int(ao_st.n_tri) / max(n_ao, 1), 2)
            legs["ao_pass_ms"]["fetch_vs_random"] = fetch_vs_random(scene, int(ao_st.n_node), int(ao_st.n_tri), min(at))

        def leg_random_rays():
            # random rays: origins spread over the scene's box, uniformly random directions - incoherent by construction
            _, d_ao, _ = primary_hits()
            rng_r = np.random.default_rng(5)
            tv = flat.tri_verts.reshape(-1, 3)
            blo, bhi = tv.min(axis=0), tv.max(axis=0)
            rr = np.zeros(n_rays_total, dtype=T.RAY_DTYPE)
            rr["origin"] = (blo + (bhi - blo) * rng_r.random((n_rays_total, 3))).astype(np.float32)
            rd = rng_r.normal(size=(n_rays_total, 3))
            rr["direction"] = (rd / np.linalg.norm(rd, axis=1, keepdims=True)).astype(np.float32)
            rr["tmax"] = 3.4028234663852886e38
            d_rr = torch.from_numpy(rr.view(np.uint8).reshape(-1)).cuda()
            rr_ev = []
            for k in range(3 + 12):
                a, b = ev(), ev()
                a.record()
                scene.trace_rays_dev(d_rr.data_ptr(), n_rays_total, d_ao.data_ptr(), sem=args.sem)
                b.record()
                rr_ev.append((a, b))
            torch.cuda.synchronize()
            rt = [a.elapsed_time(b) for a, b in rr_ev][3:]
            rr_st = scene.count_rays(d_rr.data_ptr(), n_rays_total, d_ao.data_ptr(), sem=args.sem)
            legs["random_rays_ms"] = {"rays": n_rays_total, "min": round(min(rt), 4), "mean": round(sum(rt) / len(rt), 4),
                                      "mrays_at_mean": round(n_rays_total / (sum(rt) / len(rt)) / 1e3, 1),
                                      "nodes_per_ray": round(int(rr_st.n_node) / n_rays_total, 2),
                                      "tris_per_ray": round(int(rr_st.n_tri) / n_rays_total, 2),
                                      "fetch_vs_random": fetch_vs_random(scene, int(rr_st.n_node), int(rr_st.n_tri), min(rt))}

        def leg_ao_4spp():
            # (c"') BASELINE.json's "4 spp" = AO frames with seeds 0..3 (src/rt_cpu/rt_cpu.rs:95-97): ONE launch
            #       (trx_trace_ao_batch_dev) - the four passes share one drain
            d_prim, _, n_ao = primary_hits()
            d_ao4 = torch.empty(4 * n_rays_total, dtype=torch.int64, device="cuda")
            a4 = timed(lambda i: scene.trace_ao_batch_dev(view, w, h, d_prim.data_ptr(), d_ao4.data_ptr(), n_rays_total, 4,
                                                          sem=args.sem, frame0=4 * i, ao_eps=0.01))
            legs["ao_4spp_ms"] = {"rays": 4 * n_ao, "launches": 1, "min": round(a4[0], 4), "mean": round(a4[1], 4),
                                  "mrays_at_mean": round(4 * n_ao / a4[1] / 1e3, 1)}

        def leg_frame_primary_ao():
            # (c"") the reference-style frame, device-resident: primary + AO as two launches back to back on one stream, and as
            #       ONE launch (trx_trace_frame_dev: the reference's single dispatch, a lane whose primary ray hits goes on as
            #       the pixel's AO ray) - same records either way
            d_prim, d_ao, _ = primary_hits()
            f2 = timed(lambda i: (scene.trace_primary_dev(view, w, h, d_prim.data_ptr(), sem=args.sem),
                                  scene.trace_ao_dev(view, w, h, d_prim.data_ptr(), d_ao.data_ptr(), sem=args.sem, frame=i % 4, ao_eps=0.01)))
            f1 = timed(lambda i: scene.trace_frame_dev(view, w, h, d_prim.data_ptr(), d_ao.data_ptr(), sem=args.sem, frame=i % 4, ao_eps=0.01))
            legs["frame_primary_ao_ms"] = {"two_launches": {"min": round(f2[0], 4), "mean": round(f2[1], 4)},
                                           "one_launch": {"min": round(f1[0], 4), "mean": round(f1[1], 4)}}

        def frame_loop_leg(sc, vw, frames=48):
            """The reference's frame loop (src/rt_gpu/rt_gpu_software.rs:271-361: primary pass, AO pass, next frame), device
            resident, `frames` frames with the noise seed advancing (--animate): serial on one stream, and with frame i's AO
            pass on a second stream under frame i + 1's primary pass (trx_frame_loop); the last frame's records of both modes
            are compared."""
            sc.frame_loop(vw, w, h, sem=args.sem, frames=8, overlap=False)          # (the loop's two launch slots learn their orders)
            sc.frame_loop(vw, w, h, sem=args.sem, frames=8, overlap=True)
            ser = [sc.frame_loop(vw, w, h, sem=args.sem, frames=frames, overlap=False, fetch=(k == 0)) for k in range(3)]
            ovl = [sc.frame_loop(vw, w, h, sem=args.sem, frames=frames, overlap=True, fetch=(k == 0)) for k in range(3)]
            same = bool((ser[0][1] == ovl[0][1]).all() and (ser[0][2] == ovl[0][2]).all())
            s_ms, o_ms = min(x[0] for x in ser) / frames, min(x[0] for x in ovl) / frames
            return {"frames": frames, "serial_ms_per_frame": round(s_ms, 4), "overlapped_ms_per_frame": round(o_ms, 4),
                    "gain": round(1.0 - o_ms / s_ms, 4), "records_identical": same}

        def leg_frame_loop_overlapped():
            legs["frame_loop_overlapped_ms"] = frame_loop_leg(scene, view)

        def leg_hairball():
            # (c5) BASELINE.json configs[3] itself: the hairball-class stand-in, primary frame, one AO pass, and the 4 spp in one launch
            if not (args.scene == "bistro" and args.tris == 0 and not args.input):
                return
            hv, hc = T.gen_scene("hairball", 0, 1)
            hflat = T.flat_build(hv, hc, use_tlas=False, threads=threads, preset=args.preset)
            hscene = T.Scene(hflat, device=local_rank)
            try:
                he, hl, hf = T.scene_camera("hairball")
                hview = T.view_from_camera(he, hl, hf, w, h)
                hp = torch.empty(n_rays_total, dtype=torch.int64, device="cuda")
                ha = torch.empty(4 * n_rays_total, dtype=torch.int64, device="cuda")
                hprim = [hscene.bench_primary(hview, w, h, sem=args.sem, warmup=3, frames=20) for _ in range(2)]
                hscene.trace_primary_dev(hview, w, h, hp.data_ptr(), sem=args.sem)
                torch.cuda.synchronize()
                h_ao = int(((hp & 0xffffffff) != 0x7f800000).sum().item())
                h1 = timed(lambda i: hscene.trace_ao_dev(hview, w, h, hp.data_ptr(), ha.data_ptr(), sem=args.sem, frame=i % 4, ao_eps=0.01))
                h4 = timed(lambda i: hscene.trace_ao_batch_dev(hview, w, h, hp.data_ptr(), ha.data_ptr(), n_rays_total, 4, sem=args.sem,
                                                               frame0=4 * i, ao_eps=0.01))
                h_st = hscene.count_ao(hview, w, h, hp.data_ptr(), ha.data_ptr(), sem=args.sem, frame=0, ao_eps=0.01)
                legs["hairball_4spp"] = {
                    "scene": "hairball", "tris": int(hflat.n_tris), "ao_rays_per_frame": h_ao,
                    "ao_pass_fetch_vs_random": fetch_vs_random(hscene, int(h_st.n_node), int(h_st.n_tri), h1[0]),
                    "primary_ms": round(sum(q[1] for q in hprim) / 2, 4),
                    "ao_pass_ms": {"min": round(h1[0], 4), "mean": round(h1[1], 4), "mrays_at_mean": round(h_ao / h1[1] / 1e3, 1)},
                    "ao_4spp_one_launch_ms": {"min": round(h4[0], 4), "mean": round(h4[1], 4),
                                              "mrays_at_mean": round(4 * h_ao / h4[1] / 1e3, 1)},
                }
                del hp, ha
                legs["hairball_4spp"]["frame_loop"] = frame_loop_leg(hscene, hview, frames=24)
            finally:
                hscene.close()

        def leg_pipelined():
            # (d) frames overlapped on 4 streams (independent frames; the tail of one overlaps the next)
            ps = [torch.cuda.Stream() for _ in range(4)]
            pbuf = [torch.empty(n_rays_total, dtype=torch.int64, device="cuda") for _ in ps]

            def pipelined(n):
                for k in range(n):
                    with torch.cuda.stream(ps[k % 4]):
                        scene.trace_primary_dev(view, w, h, pbuf[k % 4].data_ptr(), sem=args.sem, stream=ps[k % 4].cuda_stream)
            pipelined(16)
            torch.cuda.synchronize()
            tp = time.perf_counter()
            pipelined(200)
            torch.cuda.synchronize()
            legs["pipelined_mrays"] = round(n_rays_total * 200 / (time.perf_counter() - tp) / 1e6, 1)
            legs["pipelined_frames_in_flight"] = 4

        def leg_hbm_copy():
            # (e) measured HBM ceiling: the float4 copy kernel of the platform guide over 2 x 1 GiB (read + written bytes per
            #     second, trx_debug_copy_rate), and hipMemcpyDtoD of the same size beside it (what rounds 1-5 quoted: it runs
            #     about a fifth below the kernel on this part)
            legs["hbm_copy_gbs"] = round(T.copy_rate(local_rank, 1 << 30, reps=5) / 1e9, 1)
            n64 = (1 << 30) // 8
            src = torch.empty(n64, dtype=torch.int64, device="cuda").fill_(1)
            dst = torch.empty_like(src)
            for _ in range(3):
                dst.copy_(src)
            e0, e1 = ev(), ev()
            e0.record()
            for _ in range(10):
                dst.copy_(src)
            e1.record()
            torch.cuda.synchronize()
            legs["hbm_memcpy_dtod_gbs"] = round(10 * 2 * (1 << 30) / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)

        def leg_moving_camera():
            # (h) a camera that moves: every frame a new view (the eye and its target advance 5 cm along the street per frame),
            #     the tile order learnt from the PREVIOUS view; what a renderer with temporal coherence gets, between the
            #     static-camera figure and the cold one
            mv = []
            mbuf = torch.empty(n_rays_total, dtype=torch.int64, device="cuda")
            for f in range(48):
                off = 0.05 * f
                v = T.view_from_camera((eye[0] + off, eye[1], eye[2]), (look[0] + off, look[1], look[2]), fov, w, h)
                a, b = ev(), ev()
                a.record()
                scene.trace_primary_dev(v, w, h, mbuf.data_ptr(), sem=args.sem)
                b.record()
                mv.append((a, b))
            torch.cuda.synchronize()
            mt = [a.elapsed_time(b) for a, b in mv][8:]
            legs["moving_camera_ms"] = {"step_m": 0.05, "frames": len(mt), "min": round(min(mt), 4), "mean": round(sum(mt) / len(mt), 4)}

        def leg_dense_scene():
            # (g) second headline row: the denser bistro-class stand-in built to the reference's PROFILE_RT legend
            #     (about 30 node visits / 15 triangle tests per primary ray, rt_gpu_software.hlsl:95,102), same protocol
            if not (args.scene == "bistro" and args.tris == 0 and not args.input):
                return
            dv, dc = T.gen_scene("bistro_dense", 0, 1)
            dflat = T.flat_build(dv, dc, use_tlas=False, threads=threads, preset=args.preset)
            dscene = T.Scene(dflat, device=local_rank)
            try:
                dst = dscene.count_primary(view, w, h, sem=args.sem)
                dp = [dscene.bench_primary(view, w, h, sem=args.sem, warmup=3, frames=20) for _ in range(3)]
                legs["dense_scene"] = {
                    "scene": "bistro_dense", "tris": int(dflat.n_tris), "nodes_per_ray": round(dst.n_node / dst.n_rays, 2),
                    "tris_per_ray": round(dst.n_tri / dst.n_rays, 2),
                    "min_ms": round(sum(p[0] for p in dp) / 3, 4), "mean_ms": round(sum(p[1] for p in dp) / 3, 4),
                    "mrays_at_mean": round(n_rays_total / (sum(p[1] for p in dp) / 3) / 1e3, 1),
                }
            finally:
                dscene.close()

        def tree_leg(pflat, pbuild, params):
            pscene = T.Scene(pflat, device=local_rank)
            try:
                pst = pscene.count_primary(view, w, h, sem=args.sem)
                pp = [pscene.bench_primary(view, w, h, sem=args.sem, warmup=3, frames=20) for _ in range(3)]
                return {
                    "params": params, "build_seconds": round(pbuild, 2), "nodes": int(pflat.n_nodes),
                    "nodes_per_ray": round(pst.n_node / pst.n_rays, 2), "tris_per_ray": round(pst.n_tri / pst.n_rays, 2),
                    "min_ms": round(sum(q[0] for q in pp) / 3, 4), "mean_ms": round(sum(q[1] for q in pp) / 3, 4),
                    "mrays_at_mean": round(n_rays_total / (sum(q[1] for q in pp) / 3) / 1e3, 1),
                }
            finally:
                pscene.close()

        def leg_ploc_pipeline():
            # (i) the same frame over a tree from the ploc_cwbvh pipeline with the reference's command-line defaults for
            #     BvhBuildParams (src/main.rs:85-124,571-585: search distance 14, depth threshold 2, 64-bit codes, reinsertion
            #     0.15, 3 primitives per leaf) - obvhs' own values for the preset name are not in the reference tree, so the
            #     headline runs this library's medium_build; this leg says what the other builder's tree costs to traverse
            if not (args.scene == "bistro" and args.tris == 0 and not args.input):
                return
            tp0 = time.time()
            pflat = T.flat_build_params(verts, counts, T.build_params(), use_tlas=False, threads=threads)
            legs["ploc_pipeline"] = tree_leg(pflat, time.time() - tp0,
                                             "reference command-line defaults (ploc_search_distance 14, search_depth_threshold 2, "
                                             "sort_precision 64, reinsertion_batch_ratio 0.15, max_prims_per_leaf 3)")

        def leg_ploc_pipeline_gpu_stages():
            # (i') the same pipeline with its GPU stages (round 5): Morton sort + PLOC rounds, and the reinsertion pass with one
            #      batch per iteration - candidates chosen (area keys + radix sort) and searched as kernels, one thread per
            #      search, the moves chosen and applied there too - and the BVH2 -> CWBVH collapse + node encoding as kernels; every stage
            #      byte-identical to its host twin (tests/test_gpu_builder.py)
            if not (args.scene == "bistro" and args.tris == 0 and not args.input):
                return
            lib.trx_set_build_device(local_rank)
            lib.trx_set_build_reinsertion_batches(1)
            lib.trx_set_build_reinsertion(0.02, 8)   # (the ratio comes from the build parameters: 0.15; 8 iterations)
            tg0 = time.time()
            gflat = T.flat_build_params(verts, counts, T.build_params(), use_tlas=False, threads=threads)
            gbuild = time.time() - tg0
            legs["ploc_pipeline_gpu_stages"] = tree_leg(gflat, gbuild,
                                                        "the same build parameters; reinsertion in 8 whole-iteration batches (ratio 0.15); BVH2 stage, "
                                                        "reinsertion (searches and moves) and collapse + encoding on the GPU")

        def leg_medium_build_gpu():
            # (i") review item 8: the headline's own preset built with every stage on the device - trx_flat_build_preset_device:
            #      the ploc_cwbvh pipeline under the preset's reinsertion budget - against the host preset's build time and walk
            if not (args.scene == "bistro" and args.tris == 0 and not args.input):
                return
            ts = []
            for _ in range(2):   # (the second build finds the device buffers and the module in place)
                tg0 = time.time()
                gflat = T.flat_build_preset_device(verts, counts, preset=args.preset, device=local_rank, threads=threads)
                ts.append(time.time() - tg0)
            legs["preset_build_gpu"] = tree_leg(gflat, min(ts), "preset %s, every stage on the GPU" % args.preset)
            legs["preset_build_gpu"]["build_seconds_first"] = round(ts[0], 2)
            legs["preset_build_gpu"]["host_preset_build_seconds"] = round(build_s, 2)
            legs["preset_build_gpu"]["host_preset_nodes_per_ray"] = round(st.n_node / max(st.n_rays, 1), 2)

        def leg_no_wake():
            # (j) the timed region WITHOUT the wake frames: the GPU idles for a second (as it does while a host builds a scene),
            #     then the W warm-up steps and K timed steps run straight away, on clocks that are still coming up - what the
            #     round-3 protocol measured (profiles/r04_clock_ramp.log); one event pair around the K launches, like `value`
            if not one_pair:
                return
            torch.cuda.synchronize()
            time.sleep(1.0)
            run_frames(args.warmup, None)
            n0, n1 = ev(), ev()
            n0.record(streams[0])
            run_frames(args.steps, None)
            n1.record(streams[0])
            torch.cuda.synchronize()
            nw_ms = n0.elapsed_time(n1) / args.steps
            legs["no_wake"] = {"idle_s": 1.0, "warmup": args.warmup, "steps": args.steps, "kernel_ms_mean": round(nw_ms, 4),
                               "mrays": round(n_rays_total / (nw_ms * 1e-3) / 1e6, 1)}

        def leg_traverse1_threads():
            # (k) Traversable::traverse called the way the reference's CPU loop calls it (src/rt_cpu/rt_cpu.rs:35-57): 16 host
            #     threads, ONE ray per call, every call blocking for its RayHit.  The scene's resident ray service answers (no launch
            #     per ray); a caller still waits for its ray's own walk, so this is a latency figure - threads / call time - next to
            #     which trx_traverse_batch (the same rays in one call) is the throughput one
            rng = np.random.default_rng(11)
            n_t1 = 16 * 1500
            px = rng.integers(0, n_rays_total, n_t1)
            fx = (px % w + 0.5) / w * 2.0 - 1.0
            fy = 1.0 - (px // w + 0.5) / h * 2.0
            fwd = np.array(look, dtype=np.float64) - np.array(eye, dtype=np.float64)
            fwd /= np.linalg.norm(fwd)
            right = np.cross(fwd, [0.0, 1.0, 0.0])
            right /= np.linalg.norm(right)
            up = np.cross(right, fwd)
            th = np.tan(np.radians(fov) / 2.0)
            dirs = fwd[None, :] + (fx * th * w / h)[:, None] * right[None, :] + (fy * th)[:, None] * up[None, :]
            dirs /= np.linalg.norm(dirs, axis=1)[:, None]
            t1_rays = np.zeros(n_t1, dtype=T.RAY_DTYPE)
            t1_rays["origin"] = np.array(eye, dtype=np.float32)
            t1_rays["direction"] = dirs.astype(np.float32)
            t1_rays["tmax"] = 3.4028234663852886e38
            scene.traverse_threads(t1_rays[:512], threads=16, sem=args.sem)           # (warms the launch slots)
            t1_hits, t1_s, t1_launches = scene.traverse_threads(t1_rays, threads=16, sem=args.sem)
            tb_hits, tb_ms = scene.traverse_batch(t1_rays, sem=args.sem)
            one_hits, one_s, _ = scene.traverse_threads(t1_rays[:2000], threads=1, sem=args.sem)
            # ... and with the reference's own access pattern (`(0..w*h).into_par_iter()`, rt_cpu.rs:35: rayon hands every worker a
            # CONTIGUOUS run of pixel indices): thread k walks pixels k m .. (k + 1) m - 1 of the frame's middle rows
            m_run = 1500
            idx = np.arange(16 * m_run) + (h // 2) * w
            fx = (idx % w + 0.5) / w * 2.0 - 1.0
            fy = 1.0 - (idx // w + 0.5) / h * 2.0
            dirs = fwd[None, :] + (fx * th * w / h)[:, None] * right[None, :] + (fy * th)[:, None] * up[None, :]
            dirs /= np.linalg.norm(dirs, axis=1)[:, None]
            runs = np.zeros(16 * m_run, dtype=T.RAY_DTYPE)
            runs["origin"] = np.array(eye, dtype=np.float32)
            runs["direction"] = dirs.astype(np.float32)
            runs["tmax"] = 3.4028234663852886e38
            dealt = np.empty_like(runs)
            for k in range(16):
                dealt[k::16] = runs[k * m_run:(k + 1) * m_run]   # (the helper gives thread k the rays k, k + 16, ...)
            run_hits, run_s, _ = scene.traverse_threads(dealt, threads=16, sem=args.sem)
            runb_hits, _ = scene.traverse_batch(dealt, sem=args.sem)
            st = scene.service_stats()
            legs["traverse1_threads"] = {
                "threads": 16, "rays": n_t1, "mrays": round(n_t1 / t1_s / 1e6, 4),
                # (single-level scenes since round 6: a resident kernel answers the calls - `service_starts` kernel launches for
                # all of them instead of one per batch of callers)
                "service_starts": t1_launches, "us_per_call_and_thread": round(t1_s / n_t1 * 16 * 1e6, 2),
                "equals_traverse_batch": bool((t1_hits == tb_hits).all() and (one_hits == tb_hits[:2000]).all() and (run_hits == runb_hits).all()),
                "pixel_runs_mrays": round(len(dealt) / run_s / 1e6, 4), "pixel_runs_us_per_call_and_thread": round(run_s / len(dealt) * 16 * 1e6, 2),
                "gpu_us_per_call": round(st["gpu_us_per_call"], 2), "trips_per_call": round(st["trips_per_call"], 1),
                "traverse_batch_kernel_mrays": round(n_t1 / (tb_ms * 1e-3) / 1e6, 1),
                "one_thread_mrays": round(2000 / one_s / 1e6, 4), "one_thread_us_per_call": round(one_s / 2000 * 1e6, 2),
                "note": "one blocking trx_traverse1 call per ray from 16 host threads (and from one): Traversable::traverse, literally; `mrays`: random pixels, `pixel_runs_mrays`: every thread a contiguous run of pixels (rayon's split of the reference's loop)",
            }

        def leg_traverse1_two_level():
            # (k2) the same literal traverse over a TWO-LEVEL scene (the reference's CwBvhTlasScene, src/rt_cpu/mod.rs:48-60): answered by
            #      the resident service as well since the thin walk learned its two levels (late round 6; the launch combiner of round 5
            #      before that) - a reduced san-miguel-class scene, so that the default run stays short
            v2, c2 = T.gen_scene("san_miguel", 600000, 1)
            f2 = T.flat_build(v2, c2, use_tlas=True)
            e2, l2, fov2 = T.scene_camera("san_miguel")
            w2, h2 = 640, 360
            sc2 = T.Scene(f2, device=local_rank)
            try:
                rng = np.random.default_rng(13)
                px = rng.integers(0, w2 * h2, 16 * 600)
                fx = (px % w2 + 0.5) / w2 * 2.0 - 1.0
                fy = 1.0 - (px // w2 + 0.5) / h2 * 2.0
                fwd = np.array(l2, dtype=np.float64) - np.array(e2, dtype=np.float64)
                fwd /= np.linalg.norm(fwd)
                right = np.cross(fwd, [0.0, 1.0, 0.0])
                right /= np.linalg.norm(right)
                up = np.cross(right, fwd)
                th = np.tan(np.radians(fov2) / 2.0)
                dirs = fwd[None, :] + (fx * th * w2 / h2)[:, None] * right[None, :] + (fy * th)[:, None] * up[None, :]
                dirs /= np.linalg.norm(dirs, axis=1)[:, None]
                r2 = np.zeros(len(px), dtype=T.RAY_DTYPE)
                r2["origin"] = np.array(e2, dtype=np.float32)
                r2["direction"] = dirs.astype(np.float32)
                r2["tmax"] = 3.4028234663852886e38
                sc2.traverse_threads(r2[:512], threads=16, sem=args.sem)
                g16, s16, starts = sc2.traverse_threads(r2, threads=16, sem=args.sem)
                g1, s1, _ = sc2.traverse_threads(r2[:1500], threads=1, sem=args.sem)
                gb, _ = sc2.traverse_batch(r2, sem=args.sem)
                st = sc2.service_stats()
                legs["traverse1_two_level"] = {
                    "scene": "san_miguel (600 000 triangles, %d TLAS primitives)" % int(f2.instance_offsets.shape[0]), "threads": 16, "rays": int(len(px)),
                    "mrays": round(len(px) / s16 / 1e6, 4), "us_per_call_and_thread": round(s16 / len(px) * 16 * 1e6, 2),
                    "one_thread_mrays": round(1500 / s1 / 1e6, 4), "one_thread_us_per_call": round(s1 / 1500 * 1e6, 2),
                    "service_starts": starts, "gpu_us_per_call": round(st["gpu_us_per_call"], 2), "trips_per_call": round(st["trips_per_call"], 1),
                    "equals_traverse_batch": bool((g16 == gb).all() and (g1 == gb[:1500]).all()),
                }
            finally:
                sc2.close()

        def leg_footprint():
            # (f) compulsory footprint: distinct nodes / triangles one frame touches
            fn, ft = scene.footprint(view, w, h, sem=args.sem)
            legs["footprint"] = {"nodes": fn, "tris": ft, "bytes": NODE_BYTES * fn + TRI_BYTES * ft + HIT_BYTES * n_rays_total}

        for name, fn in (("reference_protocol", leg_reference_protocol), ("cold_order_ms", leg_cold_order), ("sem_hlsl_ms", leg_sem_hlsl),
                         ("first_frame_ms", leg_first_frame), ("ao_pass_ms", leg_ao_pass), ("random_rays_ms", leg_random_rays),
                         ("ao_4spp_ms", leg_ao_4spp), ("frame_primary_ao_ms", leg_frame_primary_ao),
                         ("frame_loop_overlapped_ms", leg_frame_loop_overlapped), ("hairball_4spp", leg_hairball),
                         ("pipelined_mrays", leg_pipelined), ("hbm_copy_gbs", leg_hbm_copy), ("moving_camera_ms", leg_moving_camera),
                         ("dense_scene", leg_dense_scene), ("ploc_pipeline", leg_ploc_pipeline),
                         ("ploc_pipeline_gpu_stages", leg_ploc_pipeline_gpu_stages), ("preset_build_gpu", leg_medium_build_gpu),
                         ("no_wake", leg_no_wake),
                         ("traverse1_threads", leg_traverse1_threads), ("traverse1_two_level", leg_traverse1_two_level),
                         ("footprint", leg_footprint)):
            run_leg(name, fn)
            ctx.pop("tmp", None)
        ctx.clear()
        failed = [k for k, v in legs.items() if isinstance(v, dict) and "error" in v]
        if failed:
            print("WARNING: bench legs failed: %s" % ", ".join(failed), file=sys.stderr)

    pmc, pmc_src, pmc_ms = None, None, None
    if rank == 0 and world == 1 and args.sim_shards == 1:
        if not args.no_pmc:
            npz = os.path.join(tempfile.gettempdir(), "trx_bench_scene_%d.npz" % os.getpid())
            np.savez(npz, nodes=flat.nodes, tri_verts=flat.tri_verts, instance_offsets=flat.instance_offsets,
                     tlas_start=np.uint32(flat.tlas_start), view=np.frombuffer(bytes(view), dtype=np.uint8),
                     width=np.uint32(w), height=np.uint32(h), sem=np.uint32(args.sem), **entry_nodes(flat))
            try:
                pmc, pmc_ms = pmc_passes(npz)
                pmc_src = "live: rocprofv3 --pmc child passes of this workload" if pmc else "live passes failed (%s)" % pmc_ms
            finally:
                try:
                    os.remove(npz)
                except OSError:
                    pass
        if not pmc:
            tpath = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tpath) and args.scene == "bistro" and (w, h) == (1920, 1080):
                try:
                    tj = json.load(open(tpath))
                    pmc = {"SQ_INSTS_VALU": tj.get("valu_wave_insts_per_launch"),
                           "_hbm_bytes": tj.get("hbm_bytes_per_launch")}
                    pmc_src = "profiles/traffic.json (committed rocprofv3 passes, not this run)" + (
                        "; " + pmc_src if pmc_src else "")
                except Exception:  # noqa: BLE001
                    pmc = None

    if rank == 0:
        req_gbs = launch_bytes / (kernel_ms * 1e-3) / 1e9
        traffic = None
        valu = None
        if pmc:
            if pmc.get("_hbm_bytes") is not None:
                traffic = pmc["_hbm_bytes"]
            elif "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
                # rocprofv3 reports KiB; gfx950 FETCH_SIZE tallies 128-B requests at 64 B: doubled (guide, HBM section)
                traffic = int((2.0 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024)
            valu = pmc.get("SQ_INSTS_VALU")
        peak_ginstr = SIMDS * CLOCK_GHZ / VALU_CYCLES
        ach_ginstr = (valu / (kernel_ms * 1e-3) / 1e9) if valu else None
        # algorithmic VALU work = 250 lane-operations per node step + 50 per triangle test, 64 lanes
        # to a wave-instruction - a property of the rays and the tree (the counts are the oracle's exactly), independent of
        # this kernel's instruction stream, so the fraction rises only when the frame gets faster.  (Until round 4 `frac`
        # divided the ISSUED instructions instead, a figure that goes up when the kernel wastes instructions: that one is
        # `issued_frac` now.)
        algo = (st.n_node * 250.0 + st.n_tri * 50.0) / 64.0
        algo_ginstr = algo / (kernel_ms * 1e-3) / 1e9
        roof = {
            # the kernel is bound by vector-instruction issue, not by bytes: achieved = algorithmic
            # wave-instructions per second, peak = one wave64 instruction per 2 cycles per SIMD-32 x 1024 SIMDs x 2.4 GHz
            "bound": "valu",
            "achieved": round(algo_ginstr, 1),
            "peak": round(peak_ginstr, 1),
            "unit": "Ginstr/s",
            "frac": round(algo_ginstr / peak_ginstr, 4),
            "traffic": traffic,   # HBM bytes per launch from FETCH_SIZE (doubled) + WRITE_SIZE
            "source": pmc_src,
            "kernel_ms": round(kernel_ms, 4),
            "algorithmic_valu_wave_insts_per_launch": int(algo),
            "algorithmic_note": "(node steps x 250 + triangle tests x 50) / 64 wave-instructions per launch (SURVEY.md 8(d)), "
                                "node steps and triangle tests counted by the COUNT kernel = the oracle's counts",
            # what the kernel actually issued (SQ_INSTS_VALU per launch, live rocprofv3 child passes) over the same peak
            "valu_wave_insts_per_launch": int(valu) if valu else None,
            "issued_ginstr_s": round(ach_ginstr, 1) if ach_ginstr else None,
            "issued_frac": round(ach_ginstr / peak_ginstr, 4) if ach_ginstr else None,
            "issued_over_algorithmic": round(valu / algo, 3) if valu else None,
        }
        # what a divergence-free walk of the same rays would issue with THIS kernel's tests: every COUNTED lane-level node
        # step and triangle test (the reference's PROFILE_RT counters, rt_gpu_software_query.hlsl:377-379,407-409) at the
        # static instruction count of the test, 64 lanes to a wave-instruction; issued / useful = divergence + bookkeeping
        useful = (st.n_node * NODE_TEST_VALU + st.n_tri * TRI_TEST_VALU) / 64.0
        if args.sim_shards == 1:
            roof["useful_valu_wave_insts_per_launch"] = int(useful)
            roof["useful_frac"] = round(useful / (kernel_ms * 1e-3) / 1e9 / peak_ginstr, 4)
            roof["issued_over_useful"] = round(valu / useful, 3) if valu else None
            roof["useful_note"] = ("(node steps x %d + triangle tests x %d) / 64 wave-instructions per launch over the "
                                   "same peak as `frac`" % (NODE_TEST_VALU, TRI_TEST_VALU))
        # the vector-memory front end: every lane requests its node / triangle bytes through TA / L1 whatever the caches
        # then serve, so the requested bytes price the L1 data path (64 B per CU and clock), and TA_TA_BUSY says how
        # long the address unit was occupied
        l1_peak = CUS * L1_BYTES_PER_CLK_CU * CLOCK_GHZ
        roof["l1"] = {
            "bound": "l1", "unit": "GB/s", "achieved": round(req_gbs, 1), "peak": round(l1_peak, 1),
            "frac": round(req_gbs / l1_peak, 4),
            "ta_busy_frac": (round(pmc["TA_TA_BUSY_sum"] / (CUS * pmc["GRBM_GUI_ACTIVE"] / 8.0), 4)
                             if pmc and pmc.get("TA_TA_BUSY_sum") and pmc.get("GRBM_GUI_ACTIVE") else None),
            "l1_hit_rate": (round(1.0 - pmc["TCP_TCC_READ_REQ_sum"] / pmc["TCP_TOTAL_CACHE_ACCESSES_sum"], 4)
                            if pmc and pmc.get("TCP_TOTAL_CACHE_ACCESSES_sum") else None),
            "note": "achieved = requested (algorithmic) bytes per second, all of which cross TA / L1; peak = 256 CUs x 64 B "
                    "per clock x 2.4 GHz; ta_busy_frac = TA_TA_BUSY summed over the CUs / (256 x GPU-active cycles per XCD)",
        }
        if pmc and pmc.get("SQ_INSTS"):
            # every instruction class shares the SIMD's issue stage: all wave-instructions per second
            # against the issue rate measured for this instruction mix at this occupancy
            peak_meas = SIMDS * CLOCK_GHZ / ISSUE_CYCLES_MEASURED
            roof["issue_stage"] = {
                "all_wave_insts_per_launch": int(pmc["SQ_INSTS"]),
                "achieved_ginstr_s": round(pmc["SQ_INSTS"] / (kernel_ms * 1e-3) / 1e9, 1),
                "peak_measured_ginstr_s": round(peak_meas, 1),
                "frac": round(pmc["SQ_INSTS"] / (kernel_ms * 1e-3) / 1e9 / peak_meas, 4),
                "valu_frac_of_measured": round(ach_ginstr / peak_meas, 4) if ach_ginstr else None,
                "note": "peak = one instruction per 2.63 cycles per SIMD at 4 waves per SIMD, measured "
                        "(profiles/r02_ubench_valu_rates2.log); the 2-cycle figure in `peak` is the nominal SIMD-32 rate",
            }
        if pmc and "SQ_WAVE_CYCLES" in pmc:
            wc = pmc["SQ_WAVE_CYCLES"]
            roof["wave_cycle_split"] = {k: round(pmc[c] / wc, 3) for k, c in (
                ("issuing", "SQ_ACTIVE_INST_ANY"), ("issuing_valu", "SQ_ACTIVE_INST_VALU"), ("waitcnt", "SQ_WAIT_ANY"),
                ("issue_stall", "SQ_WAIT_INST_ANY")) if c in pmc}
            roof["kernel_ms_under_profiler"] = round(pmc_ms, 4) if pmc_ms else None
        if pmc and "VALUBusy" in pmc:
            # rocprofv3's own derived metrics: VALUBusy = VALU-active cycles x 4 / SIMDs / GPU-active cycles (it prices a
            # wave64 instruction at 4 cycles, so it reads about twice `frac` and can pass 100 on the TLAS kernel);
            # VALUUtilization = active lanes per VALU instruction
            roof["rocprof_derived_pct"] = {k: round(pmc[k], 1) for k in ("VALUBusy", "VALUUtilization", "SALUBusy") if k in pmc}
        hbm_gbs = (traffic / (kernel_ms * 1e-3) / 1e9) if traffic else None
        hbm = {
            # the byte side, measured: HBM / fabric bytes per launch from the counters over the kernel time.  (The REQUESTED
            # bytes of SURVEY 8(d) - what the rays ask for - are mostly served by L1 / L2 / Infinity Cache and exceed the
            # HBM peak on coherent frames, so they are reported as `requested_gbs`, not as a fraction of anything.)
            "bound": "hbm",
            "achieved": round(hbm_gbs, 1) if hbm_gbs else None,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(hbm_gbs / HBM_PEAK_GBS, 4) if hbm_gbs else None,
            "traffic": traffic,
            # the float4 copy kernel of the platform guide (read + written GB/s, trx_debug_copy_rate); hipMemcpyDtoD, which
            # rounds 1-5 quoted here, is legs.hbm_memcpy_dtod_gbs
            "peak_measured": legs.get("hbm_copy_gbs") if isinstance(legs.get("hbm_copy_gbs"), float) else None,
            "requested_gbs": round(req_gbs, 1),
            "bytes_per_launch": int(launch_bytes),
            "compulsory_bytes": legs.get("footprint", {}).get("bytes"),
            "nodes_per_ray": round(st.n_node / max(st.n_rays, 1), 3),
            "tris_per_ray": round(st.n_tri / max(st.n_rays, 1), 3),
            "note": "requested (algorithmic) bytes = 80 x node fetches + 48 x triangle tests + 8 x rays per launch; "
                    "achieved / frac = MEASURED HBM traffic (FETCH_SIZE doubled + WRITE_SIZE), the requested bytes are "
                    "served by the caches because coherent rays share lines",
        }
        rep_ms = sorted(x / args.steps * 1e3 for x in repeat_s)
        legs["timed_region_repeats"] = {
            "n": len(rep_ms), "steps_each": args.steps, "ms_per_step_median": round(rep_ms[len(rep_ms) // 2], 4),
            "ms_per_step_min": round(rep_ms[0], 4), "ms_per_step_max": round(rep_ms[-1], 4),
            "mrays_median": round(rays_per_step / (rep_ms[len(rep_ms) // 2] * 1e-3) / 1e6, 1),
            "note": "`value` / `ms_per_step` are the FIRST region (the contract's K steps); the others follow it, each behind a barrier",
        }
        # the collective's world as the communicator itself reports it (ncclCommCount through trx_comm_world_size, or
        # torch.distributed's own count): 1 at N = 1, where no communicator exists
        rccl_world = 1
        if world > 1:
            rccl_world = fgs[0].world_size() if hasattr(fgs[0], "world_size") else dist.get_world_size()
        out = {
            "metric": baseline_metric(),
            "value": round(value, 2),
            # the same frame under the literal HLSL arithmetic (TRX_SEM_HLSL: per-node IEEE divides, tt <= t), the only
            # semantics the reference's tree states in full; `value` runs the CPU-path preset (TRX_SEM_CPU)
            "value_sem_hlsl": (round(n_rays_total / (legs["sem_hlsl_ms"]["mean"] * 1e-3) / 1e6, 2)
                               if "mean" in legs.get("sem_hlsl_ms", {}) else None),
            # the same frame over the tree of the reference-default ploc_cwbvh pipeline (BvhBuildParams of src/main.rs:571-585,
            # every stage on the GPU): what `--build ploc_cwbvh` names; `value` walks this library's --preset tree
            "value_ploc_tree": (legs["ploc_pipeline_gpu_stages"]["mrays_at_mean"]
                                if "mrays_at_mean" in legs.get("ploc_pipeline_gpu_stages", {}) else None),
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "real" if args.input else "synthetic",
            "config": {
                "workload": ("%s, %d tris, %d CWBVH nodes, primary rays %dx%d" % (os.path.basename(args.input), flat.n_tris, flat.n_nodes, w, h)
                             if args.input else
                             "%s-class procedural stand-in, %d tris, %d CWBVH nodes, primary rays %dx%d "
                             "(BASELINE.json configs[2])" % (args.scene, flat.n_tris, flat.n_nodes, w, h)),
                "semantics": "TRX_SEM_CPU" if args.sem == 3 else "bits=%d" % args.sem,
                "builder": ("PLOC BVH2 (the reference's search parameters) -> reinsertion in whole-iteration batches -> SAH-optimal BVH8 "
                            "collapse, every stage on the GPU, budget of preset %s (trx_flat_build_preset_device)" % args.preset
                            if args.builder == "ploc_gpu" else
                            "binned-SAH BVH2 -> reinsertion pass -> SAH-optimal BVH8 collapse (stands in for obvhs "
                            "ploc_cwbvh), preset %s" % args.preset),
                "parallelism": ("one GPU owns every 8x8 tile" if world == 1 else
                                "8x8 tiles round-robin over %d ranks; hit shards (8 B/ray) all-gathered in place, %d frames "
                                "per collective, and de-interleaved to row-major frames on every rank" % (world, F)),
                "frames_in_flight": n_streams,   # kernels in flight inside the timed region (one per stream)
                "frames_per_launch": L_launch,
                "frames_per_gather": F,
                # untimed frames ahead of the `warmup` steps that take the GPU out of its idle clocks (--wake-frames 0: none)
                "wake_frames": args.wake_frames,
                "build_seconds": round(build_s, 2),
                "scene_cache": cache_ranks[0],
                "tile_order": "filed by the first frame of the view on the stream, then replayed unchanged (static camera, as "
                              "the reference benches); a first frame (natural order while the tiles are measured) in "
                              "legs.first_frame_ms, feedback off in legs.cold_order_ms",
            },
            # round 5: roofline.frac is the algorithmic fraction (issued_frac = the old figure), roofline_hbm carries the
            # measured traffic, timed-region repeats, same-protocol N = 1 figure at N > 1, no-wake leg (the round-3 / round-4
            # lines differ from each other by the wake frames: README "Bench protocol")
            "protocol_version": 6,
            "build": stamp,
            "rccl_world": rccl_world,
            # per rank: wall clock from the start of main() to the end of the set-up frames (scene generated / loaded / built
            # or read from rank 0's copy, upload, communicators, first frames), the build's share of it, and the whole run
            "setup_seconds": [round(x, 2) for x in setup_ranks],
            "build_seconds": [round(x, 2) for x in build_ranks],
            "scene_cache": cache_ranks,
            "wall_seconds": [round(x, 2) for x in wall_ranks],
            "n1_same_protocol_mrays": round(n1_same, 2) if n1_same else None,
            "scaling_vs_same_protocol": round(value / n1_same, 4) if n1_same else None,
            "kernel_ms_mean": round(kernel_ms, 4),
            "kernel_ms_min": round(min(launch_ms), 4),
            "kernel_ms_min_source": ("per-launch hipEvent pairs of an untimed REPLAY of the same frames right after the timed "
                                     "region" if region_kernel_ms is not None else "per-launch hipEvent pairs of the timed region"),
            # per-launch event times in order (for one frame in flight: of the untimed pass that follows the timed region;
            # kernel_ms_mean is then the timed region's own event pair / steps): shows a schedule or a clock still settling
            "kernel_ms_per_step": [round(x, 4) for x in launch_ms] if len(launch_ms) <= 64 else None,
            "kernel_ms_mean_source": ("one hipEvent pair around the K timed launches / K" if region_kernel_ms is not None else
                                      "mean of the per-launch hipEvent pairs of the timed region"),
            "timed_region_ms": round(elapsed * 1e3, 3),
            "roofline": roof,
            "roofline_hbm": hbm,
            "legs": legs,
        }
        if world > 1 and phases:
            torch.cuda.synchronize()
            nf = sum(p[3] for p in phases)
            out["phases_ms_per_frame"] = {
                "trace": round(kernel_ms, 4),
                "trace_ranks_min": round(min(kernel_ms_ranks), 4) if kernel_ms_ranks else None,
                "trace_ranks_max": round(max(kernel_ms_ranks), 4) if kernel_ms_ranks else None,
                "gather_to": args.gather_to,
                "gather": round(sum(a.elapsed_time(b) for a, b, _, _ in phases) / nf, 4),
                "assemble": round(sum(b.elapsed_time(c) for _, b, c, _ in phases) / nf, 4),
                "collective_world_size": fgs[0].world_size() if hasattr(fgs[0], "world_size") else dist.get_world_size(),
                "backend": dist.get_backend(),
                "gather_via": args.gather,
            }

    if rank == 0 and args.dump_frame:
        np.save(args.dump_frame, frame.detach().cpu().numpy())
        np.savez(args.dump_frame + ".scene.npz", nodes=flat.nodes, tri_verts=flat.tri_verts,
                 instance_offsets=flat.instance_offsets, tlas_start=np.uint32(flat.tlas_start),
                 view=np.frombuffer(bytes(view), dtype=np.uint8), width=np.uint32(w), height=np.uint32(h),
                 **entry_nodes(flat))
    # CPU baseline: the oracle (a port, not the reference binary) on the host cores, rank 0 at N=1 only; it is
    # the only place this file touches oracle/ (as the thing timed beside the GPU, and as the frame's checker)
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.sim_shards == 1:
        from oracle import binding as O
        osc = O.Scene.from_flat(flat)
        ov = O.view_from_bytes(view)
        cores = usable_cores()
        # two implementations of the same restatement: the scalar one (the checker) and the one whose node test takes
        # the eight children at once in AVX2 registers - the reference's CPU node test (obvhs) is SIMD as well, so that
        # is the one `value` quotes; both produce the same bits (asserted here on the timed frame, and in tests/test_oracle.py)
        O.set_simd(False)
        hits, ost = osc.trace_primary(ov, w, h, sem=args.sem, threads=cores)  # one full frame, scalar
        scalar_mrays = n_rays_total / ost.seconds / 1e6
        simd = O.set_simd(True)
        hits_simd, sst = osc.trace_primary(ov, w, h, sem=args.sem, threads=cores)
        simd_equal = bool((hits_simd["t"].view(np.uint32) == hits["t"].view(np.uint32)).all() and
                          (hits_simd["prim"] == hits["prim"]).all())
        n_frames, secs = 1, sst.seconds
        while secs < args.cpu_seconds and n_frames < 64:
            _, s2 = osc.trace_primary(ov, w, h, sem=args.sem, threads=cores, out=hits_simd)
            n_frames += 1
            secs += s2.seconds
        gpu = D.int64_to_hits(frame)
        parity = bool((gpu["t"].view(np.uint32) == hits["t"].view(np.uint32)).all() and
                      (gpu["prim"] == hits["prim"]).all())
        # the reference's own CPU figure is a whole frame: ray generation + primary + one AO ray per hit pixel +
        # shading (src/rt_cpu/rt_cpu.rs:35-92,98,113); two frames of that, with the GPU's primary + AO frame beside it
        frame_s = min(osc.render_frame(ov, w, h, sem=args.sem, frame=f, ao_eps=0.01, threads=cores) for f in range(2))
        O.set_simd(False)
        # (kernel time of the same rays on the GPU: the device-resident two-launch frame of legs.frame_primary_ao_ms where the
        # legs ran, else the host-buffer entry point's event time)
        if "frame_primary_ao_ms" in legs:
            gpu_frame_ms = legs["frame_primary_ao_ms"]["two_launches"]["min"]
        else:
            gpu_frame_ms = min(scene.trace_primary_ao(view, w, h, sem=args.sem, frame=f, ao_eps=0.01)[2] for f in range(4))
        out["cpu_baseline"] = {
            "value": round(n_rays_total * n_frames / secs / 1e6, 3),
            "unit": "Mrays/s",
            "cores": cores,
            "cpu_model": cpu_model(),
            "kind": "port",
            "implementation": ("AVX2 node test (8 children per instruction), scalar triangle test" if simd else
                               "scalar (this CPU has no AVX2 + FMA)"),
            "value_scalar": round(scalar_mrays, 3),                    # the scalar restatement, one frame
            "simd_equals_scalar": simd_equal,
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
