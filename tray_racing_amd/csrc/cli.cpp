// cli.cpp — `tray_racing_hip`: the reference's command line (src/main.rs:65-171) over the C ABI.
//
// Same flag names and the same 4-column result table (src/main.rs:634-640, printed once after
// `--passes` passes, averaged: :188-207).  What differs is the backend: every frame is the primary +
// AO frame of src/rt_gpu/rt_gpu_software.hlsl:47-144 traced by libtrx.so on an MI355X; the BVH is
// built on the CPU by this repo's stand-in builder (the reference builds with OBVHS).  Links against
// include/trx.h only.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include <zlib.h>

#include "../../include/trx.h"

namespace {

struct Options { // src/main.rs:65-171 (flags this backend cannot honour are rejected loudly)
    std::string input;
    bool benchmark = false;
    float render_time = 1.0f;
    std::string build = "ploc_cwbvh";
    unsigned max_prims_per_leaf = 3;
    bool cpu = false, hardware = false, verbose = false, animate = false, png = false, tlas = false, flatten_blas = false;
    unsigned width = 1920, height = 1080;
    float collapse_traversal_cost = 1.0f;
    unsigned passes = 3;
    std::string preset;
    float reinsertion_batch_ratio = 0.15f; // -r (src/main.rs:113-118)
    unsigned search_distance = 14, search_depth_threshold = 2, sort_precision = 64; // src/main.rs:85-98,110-112
    float post_collapse_multiplier = 0.0f;
    bool dry_run = false; // load + build only (no device needed)
    bool gpu_build = false; // --gpu-build: Morton sort + PLOC rounds of the build on the device (trx_set_build_device)
    bool split = false;   // --split: pre-splitting of large triangles
    bool overlap = false; // --overlap: frame i's AO pass under frame i + 1's primary pass (trx_frame_loop, two streams)
    int device = 0;
    unsigned semantics = TRX_SEM_HLSL; // the GPU path of the reference is the HLSL text
};

struct Camera {
    float eye[3] = {0, 0, 0}, look_at[3] = {0, 0, -1}, fov = 90.0f;
};

struct Stats {
    std::string name;
    double traversal_ms = 0, blas_build_time_s = 0, tlas_build_time_ms = 0;
};

[[noreturn]] void die(const std::string &msg) {
    std::fprintf(stderr, "%s\n", msg.c_str());
    std::exit(1);
}

void check(int rc, const char *what) {
    if (rc != TRX_OK) die(std::string(what) + ": " + trx_last_error());
}

void usage() {
    std::puts("tray_racing_hip -i <scene.ron|demoscene|standin:<name>>[,...] [--benchmark] [--render-time s]\n"
              "  [--build ploc_cwbvh] [--max-prims-per-leaf 1..3] [--collapse-traversal-cost c] [--preset p]\n"
              "  [--width w] [--height h] [--animate] [--tlas] [--flatten-blas] [--passes n] [--verbose] [--device d]\n"
              "  [--png] [--cpu-semantics] [--dry-run (load + build only, needs no GPU)] [-r reinsertion_batch_ratio]\n"
              "  [--search-distance d] [--search-depth-threshold n] [--sort-precision 64|128] [--split] [--gpu-build]\n"
              "  [--overlap (frames stay on the device, frame i's AO pass under frame i+1's primary pass; reports the average)]\n"
              "stand-in names: cornell demoscene kitchen bistro hairball san_miguel (seeded procedural scenes)");
}

Options parse_args(int argc, char **argv) {
    Options o;
    auto need = [&](int &i) -> const char * {
        if (i + 1 >= argc) die(std::string("missing value for ") + argv[i]);
        return argv[++i];
    };
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        if (a == "-i") o.input = need(i);
        else if (a == "--benchmark") o.benchmark = true;
        else if (a == "--render-time") o.render_time = (float)std::atof(need(i));
        else if (a == "--build") o.build = need(i);
        else if (a == "--max-prims-per-leaf") o.max_prims_per_leaf = (unsigned)std::atoi(need(i));
        else if (a == "--collapse-traversal-cost") o.collapse_traversal_cost = (float)std::atof(need(i));
        else if (a == "--preset") o.preset = need(i);
        else if (a == "--width") o.width = (unsigned)std::atoi(need(i));
        else if (a == "--height") o.height = (unsigned)std::atoi(need(i));
        else if (a == "--passes") o.passes = (unsigned)std::atoi(need(i));
        else if (a == "--device") o.device = std::atoi(need(i));
        else if (a == "--cpu") o.cpu = true;
        else if (a == "--hardware") o.hardware = true;
        else if (a == "--verbose") o.verbose = true;
        else if (a == "--animate") o.animate = true;
        else if (a == "--png") o.png = true;
        else if (a == "--tlas") o.tlas = true;
        else if (a == "--flatten-blas") o.flatten_blas = true;
        else if (a == "--cpu-semantics") o.semantics = TRX_SEM_CPU;
        else if (a == "--dry-run") o.dry_run = true;
        else if (a == "--gpu-build") o.gpu_build = true;
        else if (a == "--overlap") o.overlap = true;
        else if (a == "-h" || a == "--help") {
            usage();
            std::exit(0);
        } else if (a == "-r") {
            o.reinsertion_batch_ratio = (float)std::atof(need(i));
        } else if (a == "--search-distance") {
            o.search_distance = (unsigned)std::atoi(need(i));
        } else if (a == "--search-depth-threshold") {
            o.search_depth_threshold = (unsigned)std::atoi(need(i));
        } else if (a == "--sort-precision") {
            o.sort_precision = (unsigned)std::atoi(need(i));
        } else if (a == "--post-collapse-reinsertion-batch-ratio-multiplier") {
            o.post_collapse_multiplier = (float)std::atof(need(i)); // "For BVH2 only" (src/main.rs:119-123)
        } else if (a == "--split") {
            o.split = true;
        } else if (a == "--auto-tune" || a == "--disable-auto-tune-model-cache") {
            // accepted for command-line compatibility; no effect here
        } else {
            die("unknown argument: " + a);
        }
    }
    if (o.input.empty()) {
        usage();
        die("error: -i <input> is required");
    }
    if (o.build.find("cwbvh") != std::string::npos && o.max_prims_per_leaf > 3)
        die("CWBVH only supports a maximum of 3 primitives per leaf."); // src/main.rs:176-178
    if (o.build != "ploc_cwbvh") die("NO BVH BUILDER SPECIFIED (this backend serves --build ploc_cwbvh)"); // src/cwbvh.rs:99
    if (o.cpu) die("--cpu is the reference's own rt_cpu path; the HIP backend has no CPU traversal");
    if (o.hardware) die("--hardware needs ray-tracing hardware; MI355X (CDNA4) has none");
    if (o.passes == 0) o.passes = 1;
    return o;
}

// 8-bit RGBA PNG, one zlib stream, filter 0 on every scanline
void put_be32(std::vector<unsigned char> &v, uint32_t x) {
    for (int k = 3; k >= 0; k--) v.push_back((unsigned char)(x >> (8 * k)));
}
void png_chunk(std::ofstream &f, const char *tag, const std::vector<unsigned char> &body) {
    std::vector<unsigned char> head, crc_in(tag, tag + 4);
    put_be32(head, (uint32_t)body.size());
    crc_in.insert(crc_in.end(), body.begin(), body.end());
    std::vector<unsigned char> tail;
    put_be32(tail, (uint32_t)crc32(0L, crc_in.data(), (uInt)crc_in.size()));
    f.write((const char *)head.data(), 4);
    f.write((const char *)crc_in.data(), (std::streamsize)crc_in.size());
    f.write((const char *)tail.data(), 4);
}
bool write_png(const std::string &path, const std::vector<unsigned char> &rgba, unsigned w, unsigned h) {
    std::vector<unsigned char> raw;
    raw.reserve((size_t)h * (w * 4 + 1));
    for (unsigned y = 0; y < h; y++) {
        raw.push_back(0);
        raw.insert(raw.end(), rgba.begin() + (size_t)y * w * 4, rgba.begin() + (size_t)(y + 1) * w * 4);
    }
    uLongf zlen = compressBound((uLong)raw.size());
    std::vector<unsigned char> z(zlen);
    if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 6) != Z_OK) return false;
    z.resize(zlen);
    std::ofstream f(path, std::ios::binary);
    if (!f) return false;
    const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    f.write((const char *)sig, 8);
    std::vector<unsigned char> ihdr;
    put_be32(ihdr, w);
    put_be32(ihdr, h);
    const unsigned char fmt[5] = {8, 6, 0, 0, 0}; // 8 bit, RGBA
    ihdr.insert(ihdr.end(), fmt, fmt + 5);
    png_chunk(f, "IHDR", ihdr);
    png_chunk(f, "IDAT", z);
    png_chunk(f, "IEND", {});
    return (bool)f;
}

// The reference's image: AO term of the primary hit, 1/t where the primary ray missed, gamma 2.2 to u8
// (src/rt_cpu/rt_cpu.rs:57-85,102-112; the same shading ends the GPU shader, rt_gpu_software.hlsl:47-144).
void save_png(const Options &o, trx_scene *scene, const trx_view &view, unsigned frame_count, const std::string &name) {
    const size_t n = (size_t)o.width * o.height;
    std::vector<trx_hit> primary(n), ao(n);
    float ms = 0;
    check(trx_trace_primary_ao(scene, &view, o.width, o.height, o.semantics, frame_count, 0.0001f, primary.data(),
                               ao.data(), &ms), "png frame");
    std::vector<unsigned char> rgba(n * 4);
    for (size_t i = 0; i < n; i++) {
        float col = 1.0f / primary[i].t;
        if (primary[i].t < 3.4028234663852886e38f) col = ao[i].t < 3.4028234663852886e38f ? ao[i].t / (1.0f + ao[i].t) : 1.0f;
        const float g = std::pow(col, 2.2f) * 255.0f;
        const unsigned char c = (unsigned char)(uint32_t)(g < 0.f ? 0.f : g);
        rgba[4 * i + 0] = rgba[4 * i + 1] = rgba[4 * i + 2] = c;
        rgba[4 * i + 3] = 255;
    }
    const std::string path = name + "_rend.png";
    if (!write_png(path, rgba, o.width, o.height)) die("Failed to save image " + path);
    if (o.verbose) std::printf("saved %s\n", path.c_str());
}

// one input of one pass: src/main.rs:241-478 (load, build, trace) -> Stats
Stats render_input(const Options &o, const std::string &input) {
    Stats st;
    float *verts = nullptr;
    uint64_t n_tris = 0, *counts = nullptr;
    uint32_t n_objects = 0;
    Camera cam;
    if (input == "demoscene" || input.rfind("standin:", 0) == 0) {
        const std::string name = input == "demoscene" ? "demoscene" : input.substr(8);
        st.name = name;
        check(trx_gen_scene(name.c_str(), 0, 1, &verts, &n_tris, &counts, &n_objects), "scene");
        check(trx_scene_camera(name.c_str(), cam.eye, cam.look_at, &cam.fov), "camera");
    } else {
        size_t k = input.find_last_of('/');
        st.name = input.substr(k == std::string::npos ? 0 : k + 1);
        k = st.name.find_last_of('.');
        if (k != std::string::npos) st.name.erase(k);
        // scene file + model through the library's loader (src/main.rs:259-298: "Failed to load config" / the model's error)
        check(trx_load_scene(input.c_str(), &verts, &n_tris, &counts, &n_objects, cam.eye, cam.look_at, &cam.fov), "scene");
    }
    const bool tlas = o.tlas && !o.flatten_blas; // src/main.rs:300-308
    if (o.verbose) std::printf("%u objects \"%s\"\ntriangles %llu\n", n_objects, st.name.c_str(), (unsigned long long)n_tris);
    trx_flat *flat = nullptr;
    check(trx_set_build_device(o.gpu_build ? o.device : -1), "build device");
    if (o.preset.empty()) {
        // no preset: BvhBuildParams from the individual flags (src/main.rs:571-585), the ploc_cwbvh pipeline
        trx_build_params bp;
        trx_build_params_default(&bp);
        bp.pre_split = o.split ? 1u : 0u;
        bp.ploc_search_distance = o.search_distance;
        bp.search_depth_threshold = o.search_depth_threshold;
        bp.reinsertion_batch_ratio = o.reinsertion_batch_ratio;
        bp.sort_precision = o.sort_precision;
        bp.max_prims_per_leaf = o.max_prims_per_leaf;
        bp.post_collapse_reinsertion_batch_ratio_multiplier = o.post_collapse_multiplier;
        bp.collapse_traversal_cost = o.collapse_traversal_cost;
        check(trx_flat_build_params(verts, counts, n_objects, tlas ? 1 : 0, &bp, 0, &flat), "build");
    } else {
        // a named preset overrides the individual builder flags (src/main.rs:563-570); obvhs' preset values are not in
        // the reference tree, so the names select this library's own settings of matching cost
        check(trx_set_build_costs(o.collapse_traversal_cost, 0.3f), "build costs");
        check(trx_set_build_preset(o.preset.c_str()), "preset");
        if (o.gpu_build) // every stage of the preset's build on the device (PLOC + whole-iteration reinsertion + collapse)
            check(trx_flat_build_preset_device(verts, counts, n_objects, tlas ? 1 : 0, o.preset.c_str(), o.max_prims_per_leaf, 0, o.device,
                                               &flat), "build");
        else
            check(trx_flat_build(verts, counts, n_objects, tlas ? 1 : 0, o.max_prims_per_leaf, 0, &flat), "build");
    }
    st.blas_build_time_s = flat->blas_build_s;
    st.tlas_build_time_ms = flat->tlas_build_s * 1000.0;
    if (o.verbose) std::printf("nodes %llu tlas_start %u instances %u\n", (unsigned long long)flat->n_nodes, flat->tlas_start, flat->n_instances);
    if (o.dry_run) { // scene file, model and build only: nothing below this line works without a device
        if (o.verbose) std::printf("camera eye %g %g %g look_at %g %g %g fov %g\n", cam.eye[0], cam.eye[1], cam.eye[2],
                                   cam.look_at[0], cam.look_at[1], cam.look_at[2], cam.fov);
        trx_flat_destroy(flat);
        trx_free(verts);
        trx_free(counts);
        return st;
    }

    trx_scene *scene = nullptr;
    check(trx_scene_create(flat->bvh_bytes, flat->n_nodes, flat->tri_verts, flat->n_tris, TRX_TRI_VERTS_36,
                           flat->n_instances ? flat->instance_offsets : nullptr, flat->n_instances, flat->tlas_start,
                           o.device, &scene), "scene upload");
    // a re-braided TLAS (trx_set_build_rebraid, on by default) names the node of its BLAS at which each primitive's walk
    // starts; without the table every primitive would enter its BLAS at the root and a ray crossing k sub-boxes of one
    // BLAS would walk that BLAS k times (same hits, many more node visits)
    if (flat->instance_entry_nodes && flat->n_instances)
        check(trx_scene_set_instance_entry_nodes(scene, flat->instance_entry_nodes, flat->n_instances), "instance entry nodes");
    trx_view view;
    check(trx_view_from_camera(cam.eye, cam.look_at, cam.fov, (float)o.width, (float)o.height, &view), "camera");

    // frames until render_time is used up; --benchmark adds the untimed warm-up dispatch and the result
    // is the MINIMUM frame time (src/rt_gpu/rt_gpu_software.rs:289-302,339,376)
    double total_ms = 0, min_ms = 1e30;
    unsigned frames = 0, frame_count = 0;
    if (o.overlap) {
        // The same loop without the host between its frames, and with frame i's AO pass on a second stream under frame
        // i + 1's primary pass: batches of frames until render_time is used up.  A frame has no time of its own here, so the
        // result is the average over the last batch (with --benchmark: the best batch's average).
        unsigned batch = 16;
        float ms = 0;
        if (o.benchmark) check(trx_frame_loop(scene, &view, o.width, o.height, o.semantics, 0, 0, 0.0001f, 8, 1, nullptr, nullptr, &ms), "warm-up");
        do {
            check(trx_frame_loop(scene, &view, o.width, o.height, o.semantics, frame_count, o.animate ? 1 : 0, 0.0001f, batch, 1, nullptr,
                                 nullptr, &ms), "trace");
            total_ms += ms;
            min_ms = std::min(min_ms, (double)ms / batch);
            frames += batch;
            if (o.animate) frame_count = frames;
        } while (total_ms < o.render_time * 1000.0 && frames < 100000);
    } else {
        if (o.benchmark) {
            float ms = 0;
            check(trx_trace_primary_ao(scene, &view, o.width, o.height, o.semantics, 0, 0.0001f, nullptr, nullptr, &ms), "warm-up");
        }
        do {
            float ms = 0;
            check(trx_trace_primary_ao(scene, &view, o.width, o.height, o.semantics, frame_count, 0.0001f, nullptr, nullptr, &ms),
                  "trace");
            total_ms += ms;
            min_ms = std::min(min_ms, (double)ms);
            frames++;
            if (o.animate) frame_count = frames;
        } while (total_ms < o.render_time * 1000.0 && frames < 100000);
    }
    st.traversal_ms = o.benchmark ? min_ms : total_ms / frames;
    if (o.verbose) std::printf("%.2fms   avg render time over %u frames (min %.3fms)\n", total_ms / frames, frames, min_ms);
    if (o.png) save_png(o, scene, view, frame_count, st.name);

    trx_scene_destroy(scene);
    trx_flat_destroy(flat);
    trx_free(verts);
    trx_free(counts);
    return st;
}

void print_table(const std::vector<Stats> &rows) { // tabled Style::blank(), src/main.rs:207,634-640
    const char *hdr[4] = {"name", "traversal_ms", "blas_build_time_s", "tlas_build_time_ms"};
    std::vector<std::vector<std::string>> cells;
    for (const Stats &s : rows) {
        char a[64], b[64], c[64];
        std::snprintf(a, sizeof(a), "%g", s.traversal_ms);
        std::snprintf(b, sizeof(b), "%g", s.blas_build_time_s);
        std::snprintf(c, sizeof(c), "%g", s.tlas_build_time_ms);
        cells.push_back({s.name, a, b, c});
    }
    size_t wdt[4];
    for (int k = 0; k < 4; k++) {
        wdt[k] = std::strlen(hdr[k]);
        for (auto &r : cells) wdt[k] = std::max(wdt[k], r[k].size());
    }
    for (int k = 0; k < 4; k++) std::printf(" %-*s ", (int)wdt[k], hdr[k]);
    std::printf("\n");
    for (auto &r : cells) {
        for (int k = 0; k < 4; k++) std::printf(" %-*s ", (int)wdt[k], r[k].c_str());
        std::printf("\n");
    }
}

} // namespace

int main(int argc, char **argv) {
    Options o = parse_args(argc, argv);
    if (!o.dry_run && trx_device_count() <= o.device) die("no HIP device " + std::to_string(o.device) + " (libtrx.so has no CPU fallback)");
    std::vector<std::string> inputs;
    {
        std::stringstream ss(o.input);
        std::string item;
        while (std::getline(ss, item, ',')) if (!item.empty()) inputs.push_back(item);
    }
    std::vector<Stats> avg;
    for (unsigned pass = 0; pass < o.passes; pass++) {
        std::vector<Stats> stats;
        for (const std::string &in : inputs) stats.push_back(render_input(o, in));
        Stats a; // the "Avg" row, src/main.rs:479-488
        a.name = "Avg";
        for (const Stats &s : stats) {
            a.traversal_ms += s.traversal_ms / stats.size();
            a.blas_build_time_s += s.blas_build_time_s / stats.size();
            a.tlas_build_time_ms += s.tlas_build_time_ms / stats.size();
        }
        stats.push_back(a);
        if (avg.empty()) {
            avg = stats;
            for (Stats &s : avg) s.traversal_ms = s.blas_build_time_s = s.tlas_build_time_ms = 0;
        }
        for (size_t i = 0; i < stats.size(); i++) { // averaged over passes, src/main.rs:191-206
            avg[i].traversal_ms += stats[i].traversal_ms / o.passes;
            avg[i].blas_build_time_s += stats[i].blas_build_time_s / o.passes;
            avg[i].tlas_build_time_ms += stats[i].tlas_build_time_ms / o.passes;
        }
    }
    print_table(avg);
    return 0;
}
