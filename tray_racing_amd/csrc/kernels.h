// kernels.h — launch interface between the C-ABI layer (api_launch.cpp) and the
// gfx950 kernels (kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

#include "../../include/trx.h"

namespace trx {

constexpr int kWave = 64;          // gfx950 wavefront
constexpr int kMaxBlock = 256;     // up to 4 waves per workgroup
constexpr uint32_t kLptShards = 8; // appenders per tile-cost bucket (16 buckets x 8 shards = 128 lists)
#ifndef TRX_LDS_STACK
#define TRX_LDS_STACK 12
#endif
#ifndef TRX_MIN_WAVES
#define TRX_MIN_WAVES 4 // waves per SIMD the register allocator must leave room for
#endif
#ifndef TRX_TRI_BATCH
#define TRX_TRI_BATCH 1
#endif
#ifndef TRX_TRI_BATCH_TLAS
#define TRX_TRI_BATCH_TLAS 1
#endif
#ifndef TRX_TRI_BATCH_PIPE
#define TRX_TRI_BATCH_PIPE 1
#endif
// triangles of a lane requested together in a per-lane triangle round: BLAS-only walk / two-level walk / pipelined walk
constexpr int kTriBatch = TRX_TRI_BATCH, kTriBatchTlas = TRX_TRI_BATCH_TLAS, kTriBatchPipe = TRX_TRI_BATCH_PIPE;
constexpr int kLdsStack = TRX_LDS_STACK;        // traversal-stack entries per lane kept in LDS
constexpr int kSpillStack = 64 - TRX_LDS_STACK; // further entries per lane in HBM (total 64 = oracle's ORC_STACK_SIZE)
// A wave's private HBM area (TraceParams::spill, in uint2 entries): the stack entries past the LDS part, then the WORLD-SPACE
// rays of its lanes ([6][64] floats: origin, direction as given) - a two-level walk over instance transforms comes back
// to them when it leaves a BLAS, and six registers that are read on that path only are six registers the walk does not have
constexpr int kWaveScratch = kSpillStack * kWave + 3 * kWave;
// LDS per wave: stack + ray table (2 x float4 per lane) + triangle-phase tables (group, result, prefix, heads)
constexpr int kLptPend = 32; // tile-list appends a wave parks in LDS before issuing them together
constexpr int kLdsBytesPerWave = TRX_LDS_STACK * kWave * 8 + kWave * 32 + kWave * 8 + kWave * 8 + kWave * 4 + kWave * 4 + kLptPend * 8 + 6 * 8 * 4 + 64; // ... + the decoded child planes of a wave-uniform node step + the packet test's ray bounds (16 waves per CU = 160 KB exactly)
constexpr int kWaveTimeStride = 8; // diagnostics record per wave: start, end, then (TRX_STAMPS builds) phase cycles
constexpr uint32_t kMaxSteps = 1u << 22; // per-ray iteration cap: every wave reaches an exit

// kModeFused: the reference's whole pixel program in one launch (rt_gpu_software.hlsl:47-144): a lane whose primary
// ray ends in a hit becomes that pixel's AO ray in place
// kModeService: a resident kernel that answers single-ray requests out of a ring in pinned host memory (trx_traverse1:
// Traversable::traverse called ray by ray from a thread pool, src/rt_cpu/rt_cpu.rs:35-57) - no launch per ray, see below
enum TraceMode : int { kModePrimary = 0, kModeAo = 1, kModeRays = 2, kModeFused = 3, kModeService = 4 };

// The ray service (k_trace<kModeService>; api_traverse.cpp holds the host side).  A workgroup is two waves: the WALKER
// steps up to kSvcRays rays, eight lanes to a ray (the thin walk of an incoherent pass's last rays), and reads nothing
// from host memory - a load from it takes 1.5-2 us, and a wave's loads return in order, so a poll in flight would hold up
// every node fetch behind it; the PORTER polls the workgroup's kSvcRays request slots - through the SCALAR cache: a CU's
// vector-memory path makes the walker's requests wait for the porter's as well (trace_service.inc) - and hands complete
// requests to the walker through mailboxes in LDS; the walker stores its answers into the slots' answer words itself.
//   slot k of the ring = 128 bytes: [0..47] the request, three 16-byte granules {ox oy oz seq}{dx dy dz seq}{tmin tmax 0
// seq}, each written by ONE 16-byte store of the host (a request is complete when the three carry the same seq, and new
// when that differs from the slot's last answer); [64..79] the answer {t, prim, overflow, seq}, one 16-byte store of the
// GPU.  Host and GPU never write the same 64-byte line.
//   control words (pinned host memory): [0] != 0: leave (set when no caller is inside trx_traverse1); [1] a heartbeat
// the host bumps every few milliseconds while the service is up - a porter that sees it stand still for kSvcDeadTicks of
// the 100 MHz wall clock takes the host for dead and leaves as well: every wave of the grid reaches its exit.
constexpr uint32_t kSvcRays = 4;                 // request slots per workgroup (one ray per eight lanes of the walker's lower half: the porter reads a workgroup's requests in ONE look - 4 x 12 words are what its scalar registers hold)
constexpr uint32_t kSvcSlotWords = 32;           // 128 bytes per slot
constexpr unsigned long long kSvcDeadTicks = 200000000ull; // 2 s without a heartbeat

constexpr int kMaxBatchFrames = 8;

struct ViewDev {
    float view_inv[16];
    float proj_inv[16];
    float eye[3];
    float pad;
};

// Device-side counters of one launch slot.
struct alignas(128) QueueHead {
    unsigned int taken; // chunks handed out from this queue (self-resetting)
    unsigned int pad[31];
};

// Self-tuning of the frame schedule (k_trace exit protocol): start stamp of the running frame; the mode the frames run
// in (0 = whole tiles, heaviest first from the learnt order; 1 = whole tiles in natural order, no feedback machinery;
// 2 = natural order, finished rays replaced once kFbRefill lanes idle); frames run in the current phase; the phase
// (0..2 = measuring that mode, 3 = holding the winner); best frame time (100 MHz ticks) seen per mode.
struct FbState {
    unsigned long long t0;
    unsigned int mode, frames, phase, t[3];
};

struct SlotCounters {
    QueueHead heads[8];      // one work-queue head per XCD, each on its own 128-byte line
    unsigned int waves_done; // exit ticket (self-resetting)
    unsigned int overflow;   // rays whose stack overflowed or that hit the step cap (sticky)
    unsigned int pad;
    unsigned long long n_rays, n_node, n_tri, n_hits; // COUNT kernels only
    unsigned int max_stack;
    unsigned int pad2;
    unsigned long long n_wave_node, n_wave_tri; // wave-level node / triangle step executions
    // COUNT kernels: per triangle phase, histogram of the largest per-lane triangle count (0..15+) and of the
    // wave's pair total in units of 8 (0..15+)
    unsigned int hist_max[16], hist_total[16];
    // self-tuning of the frame schedule, one state per pass kind (0 = primary, 1 = AO: a frame loop runs both on one stream)
    FbState fb[2];
    unsigned int lpt_sel[2]; // which of a kind's two tile-list sets holds the order to read (TraceParams::lpt_sel)
};

struct TraceParams {
    const uint4 *nodes;
    const float4 *tris;
    const uint32_t *inst;
    const uint32_t *inst_entry; // node of its BLAS at which a TLAS primitive's walk starts (null: node 0, the reference's rule)
    const trx_ray *rays;
    const trx_hit *primary;
    trx_hit *out;
    trx_hit *out_ao;        // kModeFused: the AO pass's records (out = the primary pass's)
    uint32_t *out_ao_inst;  // ... and their instance ids (TLAS scenes), or null
    uint32_t thin_max;      // a dry wave with this many rays or fewer gives every ray several lanes (kernels.hip, thin_walk); 0 = never
    uint32_t pend_min;      // kModeFused, queues dry: convert finished primary rays to AO rays once this many lanes wait
    // instance transforms (TLAS scenes; all null = the reference's identity behaviour): world-to-object rows
    // {m0 m1 m2 t} x 3 per TLAS primitive; the instance each hit was found in, per record like `out`; the
    // primary pass's instance ids (AO mode: to take the hit triangle's normal into world space)
    const float4 *inst_xform;
    uint32_t *out_inst;
    const uint32_t *primary_inst;
    SlotCounters *ctr;
    uint2 *spill;
    uint32_t n_items;
    uint32_t tlas_start;
    uint32_t width, height, tiles_x;
    uint32_t shard_index, shard_count;
    uint32_t compact; // TRX_LAYOUT_SHARD: hit buffers indexed by local_tile*64 + pixel-in-tile
    uint32_t single_queue;  // tuning: one global queue instead of one per XCD
    // tile order feedback: 16 x kLptShards list counts + lists (lpt_cap entries each) read / written this frame
    // Two list sets of lpt_set_words words each ({16 x kLptShards counts}{16 x kLptShards lists of lpt_cap tile ids}) at
    // lpt_sets (null: no feedback); *lpt_sel (0 / 1, device side: the exit wave of a learning frame flips it) names the
    // set this frame reads, the other one is empty and takes what a learning frame files.
    uint32_t *lpt_sets;
    unsigned int *lpt_sel;
    uint32_t lpt_set_words;
    uint32_t lpt_cap;
    uint32_t same_view;     // the views of this launch are those of the previous launch of this kind on the slot: a complete
                            // order is replayed as it is - nothing is timed or filed (a FROZEN order, kernels.hip)
    uint32_t *cost;         // diagnostics: per-tile cost (wall-clock ticks), or null
    uint32_t prio_cut[3];   // chunks below these (heaviest-first) indices run at s_setprio 3 / 2 / 1
    uint8_t *touch_nodes, *touch_tris; // diagnostics (COUNT kernels): byte set per node fetched / triangle tested, or null
    uint32_t *tile_iters;   // diagnostics (COUNT kernels): per tile (node steps << 16) | triangle rounds
    uint32_t frame;
    float ao_eps;
    uint32_t tie_first;
    uint32_t refill_idle; // refill the wave when at least this many lanes are idle (1..64)
    uint32_t tri_compact_min; // consider spreading the wave's triangle tests over all lanes when a lane owns this many
    uint32_t tri_coop_ratio;  // ... and do it when the largest per-lane count exceeds this x the cooperative rounds
    uint32_t variant;
    uint32_t tune;                   // development switches (TRX_TUNE environment word), 0 in the product
    uint32_t n_tris, n_nodes;        // buffer extents (prefetch addresses are clamped to them)
    uint32_t waves_per_block;        // 1, 2 or 4
    uint32_t merge;                  // 2 waves per workgroup, incoherent BLAS pass: the second wave may hand its last rays to the first (kernels.hip, drain)
    unsigned long long *wave_times;  // diagnostics: [8*wave] start, [8*wave+1] end (wall_clock64), [+2..7] phase cycles in TRX_STAMPS builds; or null
    FbState *fb;          // image passes with tile-order feedback: the schedule tuner's state (null = feedback always on)
    uint32_t exp_exact;   // >= 1: every exponent byte of the scene's nodes is 0 or >= 21: e / d may be computed as e * (1/d) exactly; 2: and every node origin is +0 or within 2^-36 .. 2^59: (p - o) / d from 1/d by one correction step (kernels.hip, TRX_NODE_FRAME)
    uint32_t no_order;    // ignore the order on file (this frame files a new one): first frame of an image geometry
    uint32_t new_view;    // camera cut: the schedule tuner starts over
    uint32_t uni_decode;  // coherent primary walk: decode the child planes of a node step once per wave when every lane visits the same node
    uint32_t any_hit;     // explicit rays only: stop at the first accepted hit, write one byte (0/1) per ray
    // kModeService: the request / answer ring and the control words (all pinned host memory), see kSvcRays
    uint32_t *svc_ring;
    const uint32_t *svc_ctl;
    uint32_t *over_host;  // null, or a word in pinned host memory that is set when a ray of THIS launch overflows (trx_traverse1:
                          // the caller learns it from the word, without a device-to-host copy of the slot's sticky counter)
    // frames per launch: primary passes - frame f = local_tile / tiles_per_frame uses views[f]; AO passes - one view
    // (views[0]) and one primary buffer, frame f uses the noise seed frame + f (a tile's seeds are consecutive tickets of
    // one queue); either way frame f writes its records at out + f * frame_stride
    uint32_t n_frames, tiles_per_frame, frame_stride;
    // floor(2^32 / d) of the four launch-uniform divisors the refill divides by (kernels.hip, div_uniform): filled in by
    // enqueue(); left to the compiler each division keeps its reciprocal in a VECTOR register for the whole kernel
    uint32_t rcp_tiles_x, rcp_width, rcp_tiles_per_frame, rcp_n_frames;
    ViewDev views[kMaxBatchFrames];
};

// Resident waves the persistent kernel should be launched with on `device`.
int trace_grid_size(int device, int mode, bool tlas, uint32_t sem, bool count);

// Enqueues one traversal kernel.  sem: trx_semantics bits; pipe: the pipelined walk (BLAS-only scenes; ignored with a TLAS).
hipError_t launch_trace(const TraceParams &p, int mode, bool tlas, uint32_t sem, bool count, bool pipe, int grid,
                        hipStream_t stream);

} // namespace trx
