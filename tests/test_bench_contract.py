"""The bench line's contract, checked on the committed line of the newest round (profiles/rNN_bench_default.json is the
verbatim output of `python bench.py` on the GPU box) and on bench.py's own argument defaults: the keys the driver
reads, BASELINE.json's metric spelled exactly, the roofline and cpu_baseline objects, and the internal consistency
the review asked for (kernel time x steps ~ timed region, fractions below 1 where they must be) - and, from round 6 on,
that every committed artefact of the round's profile set was produced by ONE library build (round 5 shipped a bench line
of an older build beside the final profiles)."""
import glob
import importlib.util
import json
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def newest_round():
    rounds = [int(re.search(r"r(\d+)_bench_default", f).group(1)) for f in glob.glob(os.path.join(ROOT, "profiles", "r*_bench_default.json"))]
    return max(rounds)


TAG = "r%02d" % newest_round()


@pytest.fixture(scope="module")
def line():
    path = os.path.join(ROOT, "profiles", TAG + "_bench_default.json")
    return json.loads(open(path).read().strip().splitlines()[-1])


def test_the_rounds_artefacts_come_from_one_library(line):
    """Round 6 on: bench.py stamps the hash of the libtrx.so that ran (and the commit it was built at) into its line, the
    profile targets into theirs; every committed bench line and per-config summary of the round names the same library."""
    if line.get("protocol_version", 5) < 6:
        pytest.skip("round %s predates the build stamp" % TAG)
    lib = line["build"]["lib_sha16"]
    assert lib and len(lib) == 16 and line["build"]["bench_py_sha16"]
    seen = 0
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", TAG + "_bench_*.json"))):
        for l in open(f).read().strip().splitlines():
            assert json.loads(l)["build"]["lib_sha16"] == lib, f
            seen += 1
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", TAG + "_traffic_*.json"))):
        assert json.load(open(f))["lib_sha16"] == lib, f
        seen += 1
    assert seen >= 2


def test_driver_keys_and_metric(line):
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert line["metric"] == base["metric"] and line["unit"] == "Mrays/s"
    for key, typ in [("value", float), ("n_gpus", int), ("steps", int), ("warmup", int), ("ms_per_step", float),
                     ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict)]:
        assert isinstance(line[key], typ), key
    assert line["n_gpus"] == 1 and line["higher_is_better"] is True and line["scaling"] in ("weak", "strong")
    assert line["vs_baseline"] is None and base["published"] == {}      # no published number for this metric
    assert line["data"] == "synthetic" and "workload" in line["config"] and "model" not in line["config"]
    assert "configs[2]" in line["config"]["workload"] and line["dtype"] == "f32"


def test_value_is_the_one_frame_in_flight_metric(line):
    rays = 1920 * 1080
    assert line["config"]["frames_in_flight"] == 1 and line["config"]["frames_per_launch"] == 1
    assert line["value"] == pytest.approx(rays / (line["ms_per_step"] * 1e-3) / 1e6, rel=1e-3)
    # hipEvent time per launch x steps accounts for the timed region (launch gaps are the rest)
    assert 0.95 < line["kernel_ms_mean"] * line["steps"] / line["timed_region_ms"] <= 1.0
    assert line["kernel_ms_min"] <= line["kernel_ms_mean"] <= line["ms_per_step"]
    assert line["parity_vs_oracle_full_frame"] is True


def test_roofline_objects(line):
    r = line["roofline"]
    assert r["bound"] == "valu" and r["unit"] == "Ginstr/s" and r["source"].startswith("live")
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-3) and 0 < r["frac"] < 1
    # round 5: `frac` is the ALGORITHMIC fraction of SURVEY.md 8(d) - (250 lane-operations per node step + 50 per triangle
    # test) / 64 per launch over the kernel time - a figure of the rays and the tree that only rises when the frame gets
    # faster; what the kernel issued (SQ_INSTS_VALU, live counter passes) sits beside it as issued_frac
    h0 = line["roofline_hbm"]
    algo = (h0["nodes_per_ray"] * 250 + h0["tris_per_ray"] * 50) * 1920 * 1080 / 64
    assert r["algorithmic_valu_wave_insts_per_launch"] == pytest.approx(algo, rel=2e-3)
    assert r["achieved"] == pytest.approx(algo / (r["kernel_ms"] * 1e-3) / 1e9, rel=2e-3)
    assert r["issued_ginstr_s"] == pytest.approx(r["valu_wave_insts_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9, rel=1e-3)
    assert r["frac"] < r["issued_frac"] < 1 and r["issued_over_algorithmic"] == pytest.approx(r["issued_frac"] / r["frac"], rel=1e-2)
    assert "algorithmic_frac" not in r and line["protocol_version"] in (5, 6)
    assert 0 < r["issue_stage"]["frac"] < 1 and r["traffic"] > 0
    # what a divergence-free walk would issue with this kernel's tests: counted node steps x 213 + triangle tests x 70
    useful = (h0["nodes_per_ray"] * 213 + h0["tris_per_ray"] * 70) * 1920 * 1080 / 64
    assert r["useful_valu_wave_insts_per_launch"] == pytest.approx(useful, rel=2e-3)
    assert 0 < r["useful_frac"] < r["issued_frac"] and r["issued_over_useful"] == pytest.approx(r["issued_frac"] / r["useful_frac"], rel=1e-2)
    # the vector-memory front end: requested bytes over the L1 data path, TA busy share, L1 hit rate
    l1 = r["l1"]
    assert l1["bound"] == "l1" and l1["peak"] == pytest.approx(256 * 64 * 2.4, rel=1e-6)
    assert l1["frac"] == pytest.approx(l1["achieved"] / l1["peak"], rel=1e-3) and 0 < l1["frac"] < 1
    assert 0 < l1["ta_busy_frac"] < 1 and 0.9 < l1["l1_hit_rate"] < 1
    # the byte side is MEASURED traffic over the HBM peak (a roofline fraction: below 1); the requested bytes of SURVEY 8(d),
    # which coherent rays share through the caches, are reported as a rate and priced against nothing
    h = line["roofline_hbm"]
    assert h["bound"] == "hbm" and h["unit"] == "GB/s" and h["peak"] == 8000.0
    assert "requested_over_hbm_peak" not in h and h["frac"] == pytest.approx(h["achieved"] / h["peak"], abs=1e-4) and 0 < h["frac"] < 0.1
    assert h["achieved"] == pytest.approx(h["traffic"] / (line["kernel_ms_mean"] * 1e-3) / 1e9, rel=2e-3)
    per_ray = 80 * h["nodes_per_ray"] + 48 * h["tris_per_ray"] + 8       # algorithmic bytes per ray
    assert h["bytes_per_launch"] == pytest.approx(per_ray * 1920 * 1080, rel=1e-3)
    assert h["requested_gbs"] == pytest.approx(h["bytes_per_launch"] / (line["kernel_ms_mean"] * 1e-3) / 1e9, rel=1e-3)
    assert h["traffic"] < 0.1 * h["bytes_per_launch"]                    # served by the caches, not HBM
    assert h["compulsory_bytes"] < h["traffic"] * 4 and h["peak_measured"] > 3000
    if line["protocol_version"] >= 6:
        # round 6: the measured ceiling is a float4 copy KERNEL (the fastest of several shapes), not hipMemcpyDtoD
        assert h["peak_measured"] > 4800 and line["legs"]["hbm_memcpy_dtod_gbs"] <= h["peak_measured"] * 1.02


def test_repeats_no_wake_and_protocol_fields(line):
    """Round 5: the timed region is repeated (median / min / max beside the contract's first region), the no-wake figure sits
    in the same line, kernel_ms_min says where it comes from, rccl_world is reported (1: no communicator at N = 1), and the
    same-protocol N = 1 figure exists only at N > 1."""
    rep = line["legs"]["timed_region_repeats"]
    assert rep["n"] >= 7 and rep["steps_each"] == line["steps"]
    assert rep["ms_per_step_min"] <= rep["ms_per_step_median"] <= rep["ms_per_step_max"]
    assert rep["ms_per_step_min"] <= line["ms_per_step"] <= rep["ms_per_step_max"] * 1.001
    assert rep["mrays_median"] == pytest.approx(1920 * 1080 / (rep["ms_per_step_median"] * 1e-3) / 1e6, rel=1e-3)
    nw = line["legs"]["no_wake"]
    assert nw["idle_s"] == 1.0 and nw["steps"] == line["steps"] and nw["mrays"] > 0.85 * line["value"]
    assert "replay" in line["kernel_ms_min_source"].lower()
    assert line["rccl_world"] == 1 and line["n1_same_protocol_mrays"] is None and line["scaling_vs_same_protocol"] is None


def test_cpu_baseline_and_legs(line):
    # (every leg runs on its own since round 6: one that failed says so in place)
    assert not [k for k, v in line["legs"].items() if isinstance(v, dict) and "error" in v] and "error" not in line["legs"]
    c = line["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "Mrays/s" and c["cores"] >= 1 and c["value"] > 0
    assert "frame" in c["sample"] and c["cpu_model"]
    # the SIMD node test is what `value` quotes, the scalar restatement sits beside it, and they agreed bit for bit
    assert c["simd_equals_scalar"] is True and c["value"] > 2 * c["value_scalar"] > 0 and "AVX2" in c["implementation"]
    assert c["value"] >= 40.0                                            # the review's bar for a SIMD baseline on 16 cores
    # the literal HLSL arithmetic next to the headline
    assert 0 < line["value_sem_hlsl"] < line["value"]
    legs = line["legs"]
    assert legs["reference_protocol"]["passes"] == 3 and legs["reference_protocol"]["frames"] >= 20
    assert legs["cold_order_ms"]["mean"] > line["kernel_ms_mean"]         # the learnt order is what the steady state gains
    assert legs["first_frame_ms"]["mean"] > line["kernel_ms_mean"]        # a camera cut: natural order + measuring
    assert legs["moving_camera_ms"]["mean"] < legs["first_frame_ms"]["mean"]
    assert legs["ploc_pipeline"]["nodes_per_ray"] > 0 and legs["dense_scene"]["nodes_per_ray"] > 25
    # the AO pass over the bench frame's primary hits: one ray per hit, timed per launch
    assert 0 < legs["ao_pass_ms"]["rays"] <= 1920 * 1080 and legs["ao_pass_ms"]["mean"] >= legs["ao_pass_ms"]["min"] > 0
    # round 4: BASELINE's "4 spp" as ONE launch (the four AO frames share a drain: more rays per second than one pass),
    # the reference-style frame as two launches and as one, configs[3]'s own scene
    a4 = legs["ao_4spp_ms"]
    assert a4["launches"] == 1 and a4["rays"] == 4 * legs["ao_pass_ms"]["rays"] and a4["mrays_at_mean"] > legs["ao_pass_ms"]["mrays_at_mean"]
    fr = legs["frame_primary_ao_ms"]
    assert 0 < fr["two_launches"]["min"] <= fr["two_launches"]["mean"] and 0 < fr["one_launch"]["min"] <= fr["one_launch"]["mean"]
    assert fr["two_launches"]["mean"] < legs["ao_pass_ms"]["mean"] + 1.2 * line["kernel_ms_mean"]   # the sum of its passes, no more
    hb = legs["hairball_4spp"]
    assert hb["tris"] == 2880000 and hb["ao_4spp_one_launch_ms"]["mrays_at_mean"] > 1.3 * hb["ao_pass_ms"]["mrays_at_mean"]
    # round 5: the PLOC pipeline with its GPU stages (BVH2 + reinsertion selection and searches as kernels) builds several times
    # faster than on the host cores and its tree is walked with no more node visits; the single-ray Traversable path under 16
    # callers shares launches and answers exactly what the batch entry point answers
    g, p = legs["ploc_pipeline_gpu_stages"], legs["ploc_pipeline"]
    assert g["build_seconds"] < 0.5 * p["build_seconds"] and g["build_seconds"] <= 2.0 and g["nodes_per_ray"] <= p["nodes_per_ray"]
    t1 = legs["traverse1_threads"]
    assert t1["threads"] == 16 and t1["equals_traverse_batch"] is True and t1["mrays"] > 0.1
    if line["protocol_version"] >= 6:   # the resident ray service: a handful of kernel starts, four times round 5's rate
        assert t1["service_starts"] < 8 and t1["mrays"] > 0.6 and t1["one_thread_mrays"] > 0.045
        # ... every thread a contiguous run of pixels, as rayon deals out the reference's loop (rt_cpu.rs:35); on the GPU a call is
        # its ray's trips, well under a microsecond each since the porter polls through the scalar cache (EXPERIMENTS 6.7)
        assert t1["pixel_runs_mrays"] > 0.6 and 0.4 < t1["gpu_us_per_call"] / t1["trips_per_call"] < 1.0
        # ... and over a two-level scene (the reference's CwBvhTlasScene): the service as well, several times round 5's combiner
        t2 = legs["traverse1_two_level"]
        assert t2["equals_traverse_batch"] is True and t2["service_starts"] < 8 and t2["mrays"] > 0.25 and t2["one_thread_mrays"] > 0.02
        assert t2["trips_per_call"] > t1["trips_per_call"]   # (two levels: the longer walk)
    # round 5, second half: the incoherent passes against the measured no-locality fetch rate of the same scene (trx_debug_fetch_rate)
    for leg in (legs["ao_pass_ms"]["fetch_vs_random"], legs["random_rays_ms"]["fetch_vs_random"], hb["ao_pass_fetch_vs_random"]):
        assert leg["random_fetch_gbs"] > 1000 and leg["requested_gbs"] > 0.4 * leg["random_fetch_gbs"]   # north_star's ">= 40 % of the measured roofline"
        assert leg["ratio"] == pytest.approx(leg["requested_gbs"] / leg["random_fetch_gbs"], rel=2e-3)
    assert 4 < legs["random_rays_ms"]["nodes_per_ray"] < 10
    # ... and the literal-HLSL arithmetic within 7 % of the CPU preset since its divisions went (4 789 of 5 230 before)
    assert line["value_sem_hlsl"] > 0.93 * line["value"]
    if line["protocol_version"] >= 6:
        # round 6: the tree `--build ploc_cwbvh` names (reference-default parameters, GPU stages) beside the headline's; the
        # reference's frame loop with frame i's AO pass under frame i + 1's primary pass - same records, less time per frame
        assert line["value_ploc_tree"] == g["mrays_at_mean"] and 0.8 * line["value"] < line["value_ploc_tree"] < 1.1 * line["value"]
        pb = legs["preset_build_gpu"]   # review item 8: the headline's preset built on the device
        assert pb["build_seconds"] <= 0.45 and pb["nodes_per_ray"] <= 17.9 and pb["build_seconds"] < 0.5 * pb["host_preset_build_seconds"]
        fl = legs["frame_loop_overlapped_ms"]
        assert fl["records_identical"] is True and 0 < fl["overlapped_ms_per_frame"] < fl["serial_ms_per_frame"]
        assert hb["frame_loop"]["records_identical"] is True and hb["frame_loop"]["overlapped_ms_per_frame"] < hb["frame_loop"]["serial_ms_per_frame"]
        assert len(line["setup_seconds"]) == 1 and line["build_seconds"][0] == pytest.approx(line["config"]["build_seconds"], abs=0.011)


def test_the_drivers_protocol_lines_of_the_round():
    """`python bench.py --steps 20 --warmup 5` twice on one box (profiles/rNN_bench_driver_protocol.json), and once with
    --wake-frames 0 on a GPU at idle clocks (…_no_wake.json): the contract's keys, and what the wake frames are worth."""
    lines = [json.loads(l) for l in open(os.path.join(ROOT, "profiles", TAG + "_bench_driver_protocol.json")).read().strip().splitlines()]
    cold = json.loads(open(os.path.join(ROOT, "profiles", TAG + "_bench_driver_protocol_no_wake.json")).read().strip().splitlines()[-1])
    for d in lines + [cold]:
        assert d["steps"] == 20 and d["warmup"] == 5 and d["n_gpus"] == 1 and d["config"]["frames_in_flight"] == 1
        assert d["value"] == pytest.approx(1920 * 1080 / (d["ms_per_step"] * 1e-3) / 1e6, rel=1e-3)
        assert len(d["kernel_ms_per_step"]) == 20
    assert all(d["config"]["wake_frames"] == 64 and d["value"] > 4800 for d in lines)
    assert cold["config"]["wake_frames"] == 0 and cold["value"] < min(d["value"] for d in lines)
    # a GPU that has not been woken: the step costs over a per cent more, and the series' first launch is not its fastest
    # (how the series falls differs from box to box - on some the clocks are up after the scene's build, and what is left is
    # launch gaps)
    assert cold["ms_per_step"] > min(d["ms_per_step"] for d in lines) * 1.01
    assert cold["kernel_ms_per_step"][0] > min(cold["kernel_ms_per_step"]) * 1.01


def test_default_arguments_finish_in_minutes():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    import sys
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        a = mod.parse()
    finally:
        sys.argv = argv
    assert a.gpus == 1 and a.steps * 0.5e-3 < 5 and a.warmup < a.steps and a.cpu_seconds <= 30
    assert mod.baseline_metric() == json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]


def test_bench_starts_its_own_ranks_and_relays_their_failure():
    """`python bench.py --gpus 2` with no launcher around it: the parent starts two ranks through torch.distributed.run
    before it touches the GPU and relays the launcher's return code.  Without a GPU every rank stops at "no HIP device"
    (libtrx.so has no CPU fallback), so here the relayed code is a failure - and the parent neither hangs nor prints a
    bench line.  (With a GPU the same command is tests/test_gpu_parity.py::test_bench_launches_its_own_ranks.)"""
    import subprocess
    import sys
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("a GPU is present: covered by the GPU suite")
    except ImportError:
        pytest.skip("no torch")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--tris", "2000",
                          "--width", "64", "--height", "64", "--dist-backend", "gloo"], capture_output=True, text=True, timeout=600,
                         cwd=ROOT, env=env)
    assert out.returncode != 0
    assert "no HIP device" in out.stderr and not [l for l in out.stdout.splitlines() if l.startswith("{")]

