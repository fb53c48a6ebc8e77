// bmo_engine.hip — MI355X (gfx950) trace engine behind the C ABI of include/bmo.h.
//
// Execution model: wavefront tracing, bounce levels fused per wave.
//   * One launch of step_kernel / step_kernel_gauss advances every ACTIVE beam node by up to 32 (31) bounces
//     (tracing_step! + interact3d, System.jl:133-152).  Lane j works on record j of the
//     current step chunk; the segment log IS the sequence of step chunks (SoA planes), so a
//     bounce reads the 64 B it needs (pos, dir, n, hint) and writes intersection + next
//     segment once — SURVEY.md §8d's 184 B/bounce.  Inside a launch a beam that goes on writes
//     its next record in place (same slot of the next in-place chunk); every wave runs its
//     own level loop and notes how far it got (Chunk::wl).  A beam splitter met in the loop keeps
//     its transmitted child in the lane and pushes the reflected one to the next launch's chunk
//     (one reservation per wave).
//   * The scene tables (objects, shapes, triangles, n(lambda), candidate table) stay in global memory
//     and are read with SCALAR loads through constant-address-space pointers: the wave works on one
//     shape at a time, lanes that disagree on it take turns (bmo_lane.hpp "scalar scene access").
//     Rays stay in HBM as structure-of-arrays planes (coalesced 8 B/lane loads); LDS holds two
//     per-lane columns only (the union children's Lipschitz memory, the lane memory of tracing_step).
//   * Survivors of a launch's last fused level are compacted into the next chunk with a wave
//     ballot + prefix popcount and ONE atomic per workgroup.
//   * After the last step, nodes are put in the reference's order (bundle order x BFS order):
//     heap-index bitmaps per root for trees of up to 5 levels, a radix sort on (root, depth,
//     path) keys up to 26, level-by-level ranking beyond; detector hits are compacted in that
//     order (scan + gather) so the hit buffers equal the reference's push! order.
// No CPU fallback: every entry point that traces requires a HIP device.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cmath>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <unistd.h>
#include <memory>
#include <mutex>
#include <string>
#include <map>
#include <vector>

#include "bmo_lane.hpp"

using namespace bmo;

namespace {

thread_local std::string g_err;
bool dbg_on() {
    static int v = -1;
    if (v < 0) v = getenv("BMO_DEBUG") ? 1 : 0;
    return v == 1;
}
#define DBG(...)                          \
    do {                                  \
        if (dbg_on()) {                   \
            fprintf(stderr, "[bmo] " __VA_ARGS__); \
            fprintf(stderr, "\n");        \
            fflush(stderr);               \
        }                                 \
    } while (0)
int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
#define HIP_TRY(expr)                                                                                    \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess) {                                                                          \
            return fail(e_ == hipErrorOutOfMemory ? BMO_ERR_OOM : BMO_ERR_NO_DEVICE,                      \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                              \
        }                                                                                                \
    } while (0)

// ------------------------------------------------------------------ scene blob (LDS image)
struct BlobHeader {
    int32_t n_objects, n_shapes, n_children, n_tris, n_media, n_lambda, n_detectors, march_iters;
    double eps_srf, eps_ray, eps_ins, mt_keps, mt_leps, grad_h;
    uint32_t off_objects, off_shapes, off_children, off_tris, off_ntable, total;
    uint32_t has_splitter, off_coefs;
    uint32_t has_asphere, off_cands;
    int32_t n_cands, has_meniscus, pad[2];
};

// `h`: the blob's header; the kernels read it from their arguments (scalar loads the compiler may repeat instead of holding the
// values in registers — read from the LDS copy they would sit in vector registers for the whole kernel)
template <class CharPtr>
__host__ __device__ inline SceneView view_of(CharPtr blob, const BlobHeader* h) {
    SceneView S;
    S.objects = (CObject*)(blob + h->off_objects);
    S.shapes = (CShape*)(blob + h->off_shapes);
    S.children = (CInt*)(blob + h->off_children);
    S.tris = (CDouble*)(blob + h->off_tris);
    S.n_table = (CDouble*)(blob + h->off_ntable);
    S.coefs = (CDouble*)(blob + h->off_coefs);
    S.cands = (const BMO_KONST Cand*)(blob + h->off_cands);
    S.n_cands = h->n_cands;
    S.n_objects = h->n_objects;
    S.n_lambda = h->n_lambda;
    S.eps_srf = h->eps_srf;
    S.eps_ray = h->eps_ray;
    S.eps_ins = h->eps_ins;
    S.mt_keps = h->mt_keps;
    S.mt_leps = h->mt_leps;
    S.grad_h = h->grad_h;
    S.march_iters = h->march_iters;
    return S;
}

// ------------------------------------------------------------------ record layout
// double planes per record: ABI planes first (include/bmo.h), then accumulators
//   RAY:       0-6 px py pz dx dy dz n | 7-10 t nx ny nz | 11 opl_acc
//   POLARIZED: 0-10 as RAY | 11-16 Re/Im E0 | 17 opl_acc
// int planes: 0 node, 1 k, 2 hint_obj, 3 hint_shape, 4 obj, 5 shape, 6 flags
template <int KIND>
struct Layout;
template <>
struct Layout<BMO_BEAM_RAY> {
    static constexpr int ABI = 11, ND = 12, OPL = 11;
};
template <>
struct Layout<BMO_BEAM_POLARIZED> {
    static constexpr int ABI = 17, ND = 18, OPL = 17;
};
//   GAUSSIAN:  chief 0-10 | waist 11-21 | divergence 22-32 | 33 lenA 34 lenB 35 oplC 36 oplW 37 oplD
template <>
struct Layout<BMO_BEAM_GAUSSIAN> {
    static constexpr int ABI = 33, ND = 38, OPL = 35;
};
constexpr int NI = 7;
enum { I_NODE = 0, I_K = 1, I_HOBJ = 2, I_HSHAPE = 3, I_OBJ = 4, I_SHAPE = 5, I_FLAGS = 6 };
enum { F_DEAD = 1 };

struct Chunk {
    double* d;   // [ND][cap]
    int32_t* i;  // [NI][cap]
    int64_t cap;
    int64_t count;
    // In-place level `level` (1..) of a fused launch: the slots of wave w (64 consecutive slots) hold records or hole marks only if that
    // wave got as far as this level, wl[w] >= level; what lies beyond was never written.  wl == nullptr: every slot below count is a record.
    // A launch over few records spreads them thinly — wave w owns the 2^wl_shift consecutive slots from w << wl_shift (StepParams::lane_shift).
    const uint8_t* wl = nullptr;
    int32_t level = 0;
    int32_t wl_shift = 6;
};
// node id of record j of a chunk of the log, -1 where there is none (a beam that ended earlier, or a level its wave did not reach)
__device__ __forceinline__ int32_t chunk_node(const Chunk& c, int64_t j) {
    if (c.wl && c.level > (int32_t)c.wl[j >> c.wl_shift]) return -1;
    return c.i[I_NODE * c.cap + j];
}

// interact's sink for the ray that goes on (bmo_lane.hpp NextInOut): the seven slots of the lane memory
struct NextInLaneMem {
    const LaneMem& lm;
    __device__ void put(const d3& pos, const d3& dir, double n) const {
        lm.put3(0, pos);
        lm.put3(3, dir);
        lm.m[7 * lm.stride] = n;  // (slot 6 is tracing_step's plate mark)
    }
};

struct Counters {  // device-resident, one per trace
    // records written to the next chunk.  Two slots: step s accumulates into slot s & 1 and clears the other one for step s + 1
    // (the host has read it before launching step s), so no host->device reset sits between two launches.
    unsigned long long next_count[2];
    unsigned long long node_count;
    unsigned long long max_depth;  // deepest beam-tree level created so far (sizes the sort keys of the final ordering)
    unsigned long long overflow;
    unsigned long long max_level[2];  // deepest in-place level any wave of the launch reached (slots used like next_count's)
    unsigned long long inwave[2];     // reflected children pushed from inside the fused loops of the launch (slots used like next_count's)
};

struct NodeArrays {
    int32_t* root;
    int32_t* parent;
    int32_t* nseg;
    int32_t* status;
    int32_t* li;
    int32_t* hit_det;
    unsigned long long* key;  // depth<<32 | path: one bit per tree level, newest lowest (0 transmitted, 1 reflected), last 32 levels kept
    double* lambda;
    double* hit;  // [cap][9 * hit_sub]
    double* aux;  // [cap][4]: Gaussian l0 (length of parent chief), w0, Re(E0), Im(E0)
    int64_t cap;
    int32_t hit_sub;  // detector records per node: 1 (Ray / PolarizedRay), 3 (GaussianBeamlet: chief, waist, divergence)
    int32_t* old;     // retrace only: node of the previous solution this beam re-walks, -1 once it traces freshly
    // [roots] the heap indices (tree_bits below: 2^depth - 1 + path) of the nodes of every root's tree, one bit each, for trees of up to 5
    // levels: set where the nodes are made (init_roots_kernel, the step kernels' make_children) so that the final ordering needs no pass
    // over the nodes to collect them; nullptr when the scene has no beam splitter
    unsigned long long* tbits;
};

// Tables of the previous solution a retrace re-walks (System.jl:188-255), indexed by ITS node ids; all device pointers.
struct OldChunk {  // one chunk of the previous solution's segment log
    const double* d;
    int64_t cap;
};
struct OldSolution {
    const int32_t* nseg;         // stored rays per beam
    const int32_t* status;       // BMO_NODE_SPLIT <=> the beam has (two) children
    const int32_t* first_child;  // node id of the transmitted child; the reflected one follows
    const int32_t* rec_start;    // exclusive scan of nseg
    const int32_t* rec_obj;      // [rec_start[node] + k]: object of the stored intersection of ray k, -1 = none
    const double* aux;           // Gaussian: [node][4] = l0, w0, Re E0, Im E0
    // where the stored rays are: record k of beam `node` is slot (loc & 2^40 - 1) of chunk (loc >> 40), loc = rec_loc[rec_start[node] + k].
    // Two situations read stored RAYS, not only the stored objects: children the reference keeps although the re-walk of their parent
    // ended in a `nothing` interaction before the splitter (they start from their stored first ray, System.jl:232-240), and a beamlet
    // that splits before the end of its stored path (its children are sized with the stale tail attached, ThinBeamsplitter.jl:125).
    const int64_t* rec_loc;
    const OldChunk* chunks;
};
constexpr int OLD_LOC_SHIFT = 40;
// planes and slot of stored record k of beam `node` of the previous solution
__device__ __forceinline__ const double* old_record(const OldSolution& O, int32_t node, int32_t k, int64_t& cap, int64_t& slot) {
    const int64_t loc = O.rec_loc[(int64_t)O.rec_start[node] + k];
    const OldChunk c = O.chunks[loc >> OLD_LOC_SHIFT];
    cap = c.cap;
    slot = loc & (((int64_t)1 << OLD_LOC_SHIFT) - 1);
    return c.d;
}

#if !defined(BMO_MAX_FUSE)
#define BMO_MAX_FUSE 32
#endif
constexpr int MAX_FUSE = BMO_MAX_FUSE;
struct StepParams {
    BlobHeader hdr;  // copy of the scene blob's header
    const char* blob;
    uint32_t blob_bytes;
    int32_t use_lds;
    Chunk cur, nxt;
    Chunk inner[MAX_FUSE - 1];   // in-place levels 1..n_fuse-1 of this launch (capacity >= cur.count, same slot numbering as cur); the GaussianBeamlet
                                 // kernels use one more, inner[n_fuse - 1], for the rays of the last fused level (step_kernel_gauss)
    int32_t n_fuse;   // bounces per launch, 1..MAX_FUSE
    Counters* ctr;
    unsigned long long* call_shards;  // 64 counters, 128 B apart: reference intersect3d call count (metric numerator)
    NodeArrays nodes;
    int32_t r_max;
    int32_t parity;   // step & 1: which next_count slot this launch fills
    OldSolution old;  // RETR kernels only
    double* gstage;      // GaussianBeamlet kernels: [21][gstage_cap] staging planes of the reflected child's rays (GaussRecDev)
    int64_t gstage_cap;
    uint8_t* wave_last;  // [waves of the launch]: the last in-place level each wave reached (nullptr: nobody will read the log)
    // Records per wave = 2^lane_shift (6: every lane has one).  A level of a wave takes as long as its slowest lane, and the launches of a
    // deep beam tree's tail have far fewer records than the device has lanes: spread over more waves (the upper lanes idle) a slow march
    // holds up 2^lane_shift - 1 neighbours instead of 63.  Record j lives in lane j & (2^lane_shift - 1) of wave j >> lane_shift.
    int32_t lane_shift;
    // Room in P.nxt (and in the node arrays) for beam-splitter children made INSIDE the fused loop: a splitting lane goes on with its transmitted
    // child in place and pushes the reflected one to the next launch, as long as the launch has pushed fewer than this many; after that a
    // split ends the wave's loop as it always did (block_alloc has room for two records per lane whatever happened before).
    int64_t inwave_cap;
    int64_t nodes0;  // beams (nodes) in existence when the launch starts
    // Beam kernels with in-loop splitters: the reflected child of a splitting lane waits in slot j of this chunk (capacity >= cur.count) while
    // the lane goes on with the transmitted child; when that beam ends — and the wave's loop goes on — the lane takes its kept child up
    // itself, in the same launch (depth first).  The launch that used to follow for the reflected children alone lasted as long as the
    // slowest single march among them (SURVEY 8(d)'s config-2 bundle: two grazing marches of ~1 000 evaluations in 10^6 children, 1.3 ms
    // of a 5.5 ms solve with the device idle); inside the first launch those chains run beside everybody else's work.  A lane keeps ONE
    // child: a second split while one waits pushes the new reflected child to P.nxt as before; children still waiting when the wave's
    // loop ends are compacted into P.nxt behind the survivors.  d == nullptr: off.
    Chunk pend;
    double* gkeep;  // GaussianBeamlet kernels: the kept reflected child of lane j, [24][gstage_cap]: its rays as in gstage, then l0, oplC, flags
    // Workgroup b of the grid works on tile (grid - 1 - b) instead of tile b: the LAST records of the chunk start first.  The hardware hands
    // out workgroups in index order, a launch lasts until its slowest wave is done, and the slow waves of a bundle are its edge rays —
    // grazing exits through lens barrels, marches of 500 - 1 000 evaluations that are one dependent chain of ~1 ms each; a disc source
    // (BeamGroups.jl:232-243: radius grows with the index) has them at the END of the bundle, and children are queued in the order their
    // parents finished, the slow ones last again.  Started first, those chains run beside the bulk of the launch instead of behind it.
    int32_t reverse;
    // Longest-processing-time-first, from feedback (first launch of a solve over a batch that has been solved before): workgroup b works on
    // tile tile_order[b], the tiles sorted by the time they took in the previous solve of this batch, slowest first — the hardware hands out
    // workgroups in index order and a launch ends when its slowest wave does.  tile_cost[tile]: this launch's time of every tile
    // (s_memrealtime ticks), written for the next solve.  Both nullptr: `reverse` decides.  Results do not depend on any of it.
    const int32_t* tile_order;
    uint32_t* tile_cost;
#if defined(BMO_DEV_TIMELINE)
    unsigned long long* tl;  // developer builds: [2 * wave] start, [2 * wave + 1] end of every wave (wall_clock64, 100 MHz)
#endif
};

// Per-lane retrace context (shared by the Beam and the GaussianBeamlet step kernels; tests/emu walks the same rules)
struct RetraceLane {
    int32_t old = -1, old_n = 0, probe_obj = -1;
    bool probe = false, fresh_allowed = true, missed = false;
};
__device__ __forceinline__ RetraceLane retrace_lane(const StepParams& P, int32_t node, int32_t k) {
    RetraceLane r;
    r.old = P.nodes.old[node];
    if (r.old >= 0) {
        r.old_n = P.old.nseg[r.old];
        r.probe_obj = P.old.rec_obj[(int64_t)P.old.rec_start[r.old] + k];
        r.fresh_allowed = k + 1 < P.r_max;
        r.probe = r.probe_obj >= 0;
        r.missed = !r.probe;  // a stored ray without intersection: cleanup, trace_system! goes on without a hint
    }
    return r;
}

__device__ inline int lane_id() { return (int)(threadIdx.x & 63); }
__device__ inline int prefix_rank(unsigned long long mask) {
    return __popcll(mask & ((1ull << lane_id()) - 1ull));
}


// Block-aggregated slot allocation.  A single address sustains only ~88 returning atomics/us (MI355X_MICROARCH.md
// "dequeue"): one atomic per wave = 16 K per 1 M-ray launch put a ~0.3 ms floor under every launch.  Here the 4 waves
// of a block publish their ballot counts in LDS, thread 0 issues ONE atomic per counter, and every wave derives its
// own offsets.  Block layout in the next chunk: [survivors of wave 0..3][child pairs of wave 0..3].
#ifndef BMO_BLOCK
// Lanes per workgroup of the step kernels (a multiple of 64, at most 256).  Round 4: ONE wave per workgroup.  A workgroup keeps its LDS and
// its place among a CU's resident workgroups until its LAST wave is done, and the waves of a ragged launch end far apart (one grazing march
// holds a wave for milliseconds): with four waves per workgroup the ragged config-2 bundle ran with a third to two thirds of the wave slots
// EMPTY in the middle of its launch although thousands of workgroups were waiting (profiles/r04_timeline_c2v.txt) — 9.3 ms per solve, 7.5 ms
// with one wave per workgroup; the coherent bundles pay 0.5 % for four times as many workgroups (c2 3.42 -> 3.44 ms, c5 8.12 -> 8.17 ms;
// profiles/r04_ab_scheduling.txt item 9).  The slot allocation below is per workgroup either way; with one wave its barriers compile away.
#define BMO_BLOCK 64
#endif
struct SlotAlloc {
    unsigned long long surv_base, child_base, node_base;
    unsigned long long m_surv, m_split;
};
// PAIRS = false: the second channel counts single records that need no new nodes (reflected children a lane kept for itself and
// could not get to before its wave's loop ended, StepParams::pend): block layout [survivors of wave 0..3][kept children of wave 0..3].
template <bool PAIRS = true>
__device__ __forceinline__ SlotAlloc block_alloc(bool survive, bool split, uint32_t calls, const StepParams& P, char* scratch, int level = 0) {
    uint32_t* w32 = reinterpret_cast<uint32_t*>(scratch);                         // [0..3] surv, [4..7] split, [8..11] calls, [12..15] level
    unsigned long long* w64 = reinterpret_cast<unsigned long long*>(scratch + 32);  // [0] base, [1] nbase
    SlotAlloc a;
    a.m_surv = __ballot(survive);
    a.m_split = __ballot(split);
    const int wave = (int)(threadIdx.x >> 6);
    unsigned int c = calls;
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
    if (lane_id() == 0) {
        w32[wave] = (uint32_t)__popcll(a.m_surv);
        w32[4 + wave] = (uint32_t)__popcll(a.m_split);
        w32[8 + wave] = c;
        w32[12 + wave] = (uint32_t)level;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t ts = 0, tp = 0, tc = 0, ml = 0;
        for (int w = 0; w < BMO_BLOCK / 64; ++w) {
            ts += w32[w];
            tp += w32[4 + w];
            tc += w32[8 + w];
            ml = w32[12 + w] > ml ? w32[12 + w] : ml;
        }
        if (ml) atomicMax(&P.ctr->max_level[P.parity], (unsigned long long)ml);
        unsigned long long b = 0, nb = 0;
        if (ts + tp) b = atomicAdd(&P.ctr->next_count[P.parity], (unsigned long long)(ts + (PAIRS ? 2 : 1) * tp));
        if (PAIRS && tp) nb = atomicAdd(&P.ctr->node_count, (unsigned long long)(2 * tp));
        if (tc) atomicAdd(&P.call_shards[(blockIdx.x & 63u) * 16u], (unsigned long long)tc);  // sharded, no return value
        w64[0] = b;
        w64[1] = nb;
    }
    __syncthreads();
    uint32_t ps = 0, pp = 0, ts = 0;
    for (int w = 0; w < BMO_BLOCK / 64; ++w) {
        if (w < wave) {
            ps += w32[w];
            pp += w32[4 + w];
        }
        ts += w32[w];
    }
    a.surv_base = w64[0] + ps;
    a.child_base = w64[0] + ts + (PAIRS ? 2ull : 1ull) * pp;
    a.node_base = w64[1] + 2ull * pp;
    return a;
}

// Which tile of the chunk workgroup b of the grid works on (StepParams::reverse): 0 tile b; 1 the tiles from the end; 2 from both ends
// towards the middle (b even: tile b / 2, b odd: tile grid - 1 - b / 2).
__device__ __forceinline__ unsigned tile_of_block(int mode, unsigned b, unsigned grid) {
    if (mode == 1) return grid - 1u - b;
    if (mode == 2) return (b & 1u) ? grid - 1u - (b >> 1) : (b >> 1);
    return b;
}
// Waves per SIMD a step-kernel variant is compiled for = its register budget (3: 168 VGPRs, 4: 128).  Round 3: with the scene tables read by scalar
// loads tracing_step is spill-free at 168 registers, and 3 waves/SIMD beat 2 by 12 % on C2, 17 % on the vignetted bundle and 28 % on C5
// (profiles/r03_ab_scalar_scene.txt).  Round 4, after the leaf-dispatch work: at 128 registers the Ray kernel of the plain-shapes level spills
// 60 - 92 B per lane (100 - 132 before the selection-form dual rules), all of it at bounce-level depth and none inside a march, and the fourth wave is worth more than that — on LARGE launches
// (profiles/r04_ab_scheduling.txt item 14: config 2 - 4 %, config 5 - 9 %, the ragged bundle - 9 %).  A launch of a few thousand waves is as long as its slowest
// marches, and those run slower with three neighbours on their SIMD than with two (2^18 rays of config 2: + 2 %), so the Ray kernels of that level are
// compiled twice and the launch picks by its size (`wide_min_waves` below: 4 096 waves, the device filled once at 4 per SIMD).  The other variants are compiled once, for the count that measured faster
// (tools/ext_times.py; the extended-shape levels and the polarized kernels spill 216 - 540 B at 128 registers, partly inside the normals' code).
// -DBMO_MIN_WAVES=n compiles every variant for n (A/B builds).
template <int KIND, int EXT, bool RETR>
constexpr int step_waves() {
#if defined(BMO_MIN_WAVES)
    return BMO_MIN_WAVES;
#else
    if (KIND == BMO_BEAM_POLARIZED && EXT == 0) return 4;  // three singlets, 2^18 PolarizedRays: fresh 0.317 against 0.341 ms, retrace 0.297 against 0.327
    if (RETR && EXT == 2) return 4;                        // asphere objective: retrace 2.60 against 3.06 ms (Ray), 2.75 against 3.21 (PolarizedRay)
    return 3;
#endif
}
#ifndef BMO_MIN_WAVES_GAUSS
#define BMO_MIN_WAVES_GAUSS 3
#endif
// One launch advances every active beam by up to P.n_fuse bounces.  Bounce 0 reads its records from P.cur; a lane that goes
// on writes its next record IN PLACE (same slot j) into P.inner[b] and traces it in the same launch — no compaction, no host
// round trip, the ray stays in registers; lanes that ended leave an invalid record (node = -1) in the inner chunks.  The fused
// loop ends for a WAVE when one of its lanes splits (children need the slot allocation below) or when none goes on; the four waves
// of a workgroup run their loops independently and meet at the block-wide allocation.  Survivors of the last fused bounce and
// beam-splitter children are compacted into P.nxt as before.
// INW: beam splitters are handled inside the fused loop (the launches of a beam tree's tail: fewer launches, no launch waits for the
// slowest march of every generation); without it a split ends the wave's loop and both children wait for the next launch, which costs
// less register room — the large launches of the BASELINE configs run 2 - 7 % faster that way (profiles/r03_ab_inwave.txt).
template <int KIND, int EXT, bool RETR, bool INW, int WAVES = step_waves<KIND, EXT, RETR>()>
__global__ __launch_bounds__(BMO_BLOCK, WAVES) void step_kernel(StepParams P) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const SceneView S = view_of((const char*)P.blob, &P.hdr);  // scene tables: global memory, scalar loads (bmo_lane.hpp)
    char* scratch = lds;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        P.ctr->next_count[P.parity ^ 1] = 0;
        P.ctr->max_level[P.parity ^ 1] = 0;
        P.ctr->inwave[P.parity ^ 1] = 0;
    }
    using L = Layout<KIND>;
    const unsigned tile = P.tile_order ? (unsigned)P.tile_order[blockIdx.x] : tile_of_block(P.reverse, blockIdx.x, gridDim.x);
#if !defined(BMO_NO_TILE_COST)
    if (P.tile_cost && threadIdx.x == 0) P.tile_cost[tile] = (uint32_t)wall_clock64();  // start stamp; turned into the tile's time at the end
#endif
    const int64_t gwave = ((int64_t)tile * blockDim.x + threadIdx.x) >> 6;  // wave of the grid
    const int64_t j = (gwave << P.lane_shift) + lane_id();
    const int64_t m = P.cur.count;
    const bool valid = j < m && lane_id() < (1 << P.lane_shift);
#if defined(BMO_DEV_TIMELINE)
    if (P.tl && (threadIdx.x & 63) == 0 && (gwave << P.lane_shift) < P.cur.count) P.tl[2 * gwave] = wall_clock64();
#endif

    // A lane carries nothing but `alive` from one fused bounce to the next: it writes its next record and reads it back at the
    // top of the next iteration (its own store, served by L1/L2) — keeping the ray in registers across the march instead costs
    // ~300 B/lane of scratch spills, and carrying only what the next march needs (position, direction, header) from the in-place store
    // to the loop top still costs 64 B of them and 0.8 % (round 2).
    bool alive = valid;
    uint32_t calls = 0;
    auto write_ray = [&](const Chunk& C, int64_t slot, const RayS& r, int32_t nd, int32_t kk, int32_t ho, int32_t hs, int32_t fl, double opl) {
        const int64_t ncap = C.cap;
        double* D = C.d;
        int32_t* I = C.i;
        D[0 * ncap + slot] = r.pos.x;
        D[1 * ncap + slot] = r.pos.y;
        D[2 * ncap + slot] = r.pos.z;
        D[3 * ncap + slot] = r.dir.x;
        D[4 * ncap + slot] = r.dir.y;
        D[5 * ncap + slot] = r.dir.z;
        D[6 * ncap + slot] = r.n;
        if (KIND == BMO_BEAM_POLARIZED)
            for (int c = 0; c < 3; ++c) {
                D[(11 + 2 * c) * ncap + slot] = r.E0[c].re;
                D[(12 + 2 * c) * ncap + slot] = r.E0[c].im;
            }
        D[L::OPL * ncap + slot] = opl;
        I[I_NODE * ncap + slot] = nd;
        I[I_K * ncap + slot] = kk;
        I[I_HOBJ * ncap + slot] = ho;
        I[I_HSHAPE * ncap + slot] = hs;
        I[I_FLAGS * ncap + slot] = fl;
    };

    int b = 0;  // fused bounce index: records of bounce b live in C = (b == 0 ? P.cur : P.inner[b - 1]) at slot j
    // What a lane that goes on in place takes to the top of the next level: the header of the record it has just written (registers — nothing
    // else is live there) and, in the lane memory, the ray's position and direction where interact's sink left them.  Reading the record
    // back from the chunk cost eleven loads from L2 per level, each at the head of the level's dependency chain.
    int32_t c_node = -1, c_k = 0, c_ho = -1, c_hs = -1, c_fl = 0;
    int32_t pend_node = -1;  // node of the reflected child this lane keeps for itself in P.pend[j] (StepParams::pend), -1: none
#if defined(BMO_DEV_TIMELINE)
    unsigned long long tk0 = 0, tk1 = 0, tk2 = 0, tk_last = wall_clock64();  // time before / in / after tracing_step, summed over the levels
#endif
    for (;;) {
        const Chunk C = b == 0 ? P.cur : P.inner[b - 1];
        // ---- the tracing step.  Only what it needs is read before it, and nothing the interaction needs is computed before it: the
        //      record addresses, node id and segment index are derived again behind it from a copy of the slot index the optimiser
        //      cannot see through (`jj`), so no address or header value of this level is held in a register (or spilled) across the
        //      marches — the ray's origin comes back from the lane memory of tracing_step, the hit's normal too.
        RetraceLane rt;
        int32_t node = -1, k = 0;  // (two registers across the marches; read again behind them they came back from HBM: the level's records
        d3 dir{0, 0, 0};           //  do not stay in L2 that long.  The direction is live in tracing_step anyway.)
        int32_t x_obj = -1, x_shape = -1;
        double x_t = kinf();
        bool traced = false;  // the lane ran a tracing step at this level (its record is not a pushed-but-never-traced one)
        const LaneMem lm{reinterpret_cast<double*>(scratch + 64) + BMO_CC_MAX * BMO_BLOCK + threadIdx.x, BMO_BLOCK};
        const int32_t t_node = c_node, t_k = c_k, t_ho = c_ho, t_hs = c_hs, t_fl = c_fl;
        c_node = c_k = c_ho = c_hs = c_fl = 0;  // (dead from here to the next in-place write, in every lane: no register across the marches)
        if (alive) {
            const int64_t cap = C.cap;
            const double* D = C.d;
            const int32_t* I = C.i;
            int32_t flags = t_fl, ho = t_ho, hs = t_hs;
            node = t_node;
            k = t_k;
            if (b == 0) {  // (wave-uniform) the launch's input chunk
                flags = I[I_FLAGS * cap + j];
                ho = I[I_HOBJ * cap + j];
                hs = I[I_HSHAPE * cap + j];
                node = I[I_NODE * cap + j];
                k = I[I_K * cap + j];
            }
            if (RETR) {
                rt = retrace_lane(P, node, k);
                if (rt.old >= 0 && !rt.probe) ho = hs = -1;
            }
            traced = !((flags & F_DEAD) || (RETR && rt.old >= 0 && !rt.probe && !rt.fresh_allowed));  // else: pushed but never traced (System.jl:133)
            if (traced) {
                d3 pos;
                if (b == 0) {
                    pos = {D[0 * cap + j], D[1 * cap + j], D[2 * cap + j]};
                    dir = {D[3 * cap + j], D[4 * cap + j], D[5 * cap + j]};
                    // what the interaction will want behind the marches, into the lane memory with the same batch of loads (slots 7..10;
                    // from level 1 on they are there already: the beam's constants do not change along a lane, children included)
                    lm.m[7 * lm.stride] = D[6 * cap + j];
                    lm.m[8 * lm.stride] = D[L::OPL * cap + j];
                    lm.m[9 * lm.stride] = (double)P.nodes.li[node];
                    lm.m[10 * lm.stride] = P.nodes.lambda[node];
                } else {
                    pos = lm.get3(0);
                    dir = lm.get3(3);
                }
                // per-lane column of LDS behind the block_alloc scratch: Lipschitz memory of the union children (bmo_lane.hpp sdf_any)
                ChildCache cc{reinterpret_cast<double*>(scratch + 64) + threadIdx.x, BMO_BLOCK, 0};
#if defined(BMO_DEV_TIMELINE)
                {
                    const unsigned long long t = wall_clock64();
                    tk0 += t - tk_last;
                    tk_last = t;
                }
#endif
                const Hit X = tracing_step<EXT, RETR>(S, pos, dir, ho, hs, calls, cc, lm, rt.probe, rt.probe_obj, rt.fresh_allowed, &rt.missed);
#if defined(BMO_DEV_TIMELINE)
                {
                    const unsigned long long t = wall_clock64();
                    tk1 += t - tk_last;
                    tk_last = t;
                }
#endif
                x_t = X.t;
                x_obj = X.obj;
                x_shape = X.shape;
            }
        }
        // ---- the interaction and the record of this level
        int64_t jj = j;
        asm volatile("" : "+v"(jj));
        bool survive = false, split = false, still = false, old_kids = false, stale_kids = false;
        double opl_next = 0.0, lambda = 0.0;
        int32_t li = 0;
        StepOut o;
        o.outcome = OUT_MISS;
        o.status = 0;
        o.hint_obj = o.hint_shape = -1;
        o.det_slot = -1;
        o.det = nullptr;
        // header of the record that follows a surviving bounce: r_max flag and hint, with the retrace overrides
        auto next_header = [&](int32_t& fl, int32_t& ho, int32_t& hs) {
            fl = (k + 2 < P.r_max) ? 0 : F_DEAD;
            ho = o.hint_obj;
            hs = o.hint_shape;
            if (RETR && still) {
                if (k + 1 < rt.old_n) fl = 0;  // replace!: the next stored ray is re-walked whatever r_max says
                else ho = hs = -1;            // push!, then trace_system! starts over without a hint
            }
        };
        if (alive) {
            const int64_t cap = C.cap;
            double* D = C.d;
            int32_t* I = C.i;
            Hit X;
            X.t = x_t;
            X.obj = x_obj;
            X.shape = x_shape;
            X.n = {0, 0, 0};
            int status = 0;
            if (!traced) {
                status = BMO_NODE_RMAX;
            } else if (x_shape < 0) {
                status = (RETR && rt.old >= 0 && rt.missed && !rt.fresh_allowed) ? BMO_NODE_RMAX : BMO_NODE_MISS;
            } else {
                X.n = lm.get3(0);
                RayS ray;  // the interaction's share of the record
                ray.pos = lm.get3(3);
                ray.dir = dir;
                ray.n = lm.m[7 * lm.stride];
                if (KIND == BMO_BEAM_POLARIZED)
                    for (int c = 0; c < 3; ++c) ray.E0[c] = {D[(11 + 2 * c) * cap + jj], D[(12 + 2 * c) * cap + jj]};
                const double opl_acc = lm.m[8 * lm.stride];
                li = (int32_t)lm.m[9 * lm.stride];
                lambda = lm.m[10 * lm.stride];
                o.det = P.nodes.hit + (int64_t)node * 9;  // a detector hit ends the beam: its record goes straight to the node's slot
                // (the ray that goes on is put into the lane memory — hit normal, origin and plate mark there are spent — as soon as a
                //  branch of the interaction has it, and comes back from there when its record is written: bmo_lane.hpp NextInOut)
                interact<KIND>(S, ray, X, li, lambda, opl_acc, o, NextInLaneMem{lm});
                status = o.status;
                if (o.outcome == OUT_CONTINUE) {
                    survive = true;
                    opl_next = opl_acc + X.t * ray.n;
                } else if (o.outcome == OUT_SPLIT) {
                    split = true;
                    status |= BMO_NODE_SPLIT | BMO_NODE_STOPPED;
                    opl_next = opl_acc + X.t * ray.n;
                } else {
                    status |= BMO_NODE_STOPPED;
                    if (RETR && rt.old >= 0 && rt.probe && !rt.missed && P.old.first_child[rt.old] >= 0) {
                        // The re-walk ends here in a `nothing` interaction, before the splitter the stored beam ended on: the reference cuts the
                        // tail but KEEPS the children (cleanup_children stays false, System.jl:232-240); solve_system! then retraces each of them
                        // from its stored first ray (System.jl:446-458).  The two stored heads take the places a splitter's children have: the
                        // transmitted one in the lane memory (o.next for the field vector), the reflected one in o.refl.
                        stale_kids = true;
                        opl_next = opl_acc + X.t * ray.n;  // optical_path_length(parent): every ray of the cut beam up to this hit (Beam.jl:137-149)
                        const int32_t oc = P.old.first_child[rt.old];
                        int64_t hc, hs;
                        const double* H = old_record(P.old, oc, 0, hc, hs);
                        lm.put3(0, d3{H[0 * hc + hs], H[1 * hc + hs], H[2 * hc + hs]});
                        lm.put3(3, d3{H[3 * hc + hs], H[4 * hc + hs], H[5 * hc + hs]});
                        lm.m[7 * lm.stride] = H[6 * hc + hs];
                        if (KIND == BMO_BEAM_POLARIZED)
                            for (int c = 0; c < 3; ++c) o.next.E0[c] = {H[(11 + 2 * c) * hc + hs], H[(12 + 2 * c) * hc + hs]};
                        H = old_record(P.old, oc + 1, 0, hc, hs);
                        o.refl.pos = d3{H[0 * hc + hs], H[1 * hc + hs], H[2 * hc + hs]};
                        o.refl.dir = d3{H[3 * hc + hs], H[4 * hc + hs], H[5 * hc + hs]};
                        o.refl.n = H[6 * hc + hs];
                        if (KIND == BMO_BEAM_POLARIZED)
                            for (int c = 0; c < 3; ++c) o.refl.E0[c] = {H[(11 + 2 * c) * hc + hs], H[(12 + 2 * c) * hc + hs]};
                    }
                }
            }
            // intersection part of this record
            D[7 * cap + jj] = X.t;
            D[8 * cap + jj] = X.n.x;
            D[9 * cap + jj] = X.n.y;
            D[10 * cap + jj] = X.n.z;
            I[I_OBJ * cap + jj] = X.obj;
            I[I_SHAPE * cap + jj] = X.shape;
            if (RETR) {
                still = rt.old >= 0 && rt.probe && !rt.missed;  // the stored path held at this ray
                old_kids = still && P.old.first_child[rt.old] >= 0;  // (not the SPLIT status: kept stale children hang on a beam that did not split)
                if (stale_kids) {  // from here on like a splitter's children, but the beam did not split: no BMO_NODE_SPLIT
                    status |= BMO_NODE_RETRACE_STALE;
                    split = true;
                }
                if (rt.old >= 0 && !(survive && still && k + 1 < rt.old_n)) P.nodes.old[node] = -1;
            }
            if (!survive) {  // node ends here
                P.nodes.nseg[node] = k + 1;
                P.nodes.status[node] = status;
                if (o.det_slot >= 0) P.nodes.hit_det[node] = o.det_slot;
            }
        }
#if defined(BMO_DEV_TIMELINE)
        {
            const unsigned long long t = wall_clock64();
            tk2 += t - tk_last;
            tk_last = t;
        }
#endif
        const int64_t ncap = P.nxt.cap;
        auto write_next = [&](int64_t slot, const RayS& r, int32_t nd, int32_t kk, int32_t ho, int32_t hs, int32_t fl, double opl) {
            if (slot >= ncap) {
                atomicAdd(&P.ctr->overflow, 1ull);
                return;
            }
            write_ray(P.nxt, slot, r, nd, kk, ho, hs, fl, opl);
        };
        // the two children of a splitting lane: nodes cn (transmitted) and cn + 1 (reflected)
        auto make_children = [&](int64_t cn) {
            const unsigned long long pkey = P.nodes.key[node];
            const unsigned long long depth = pkey >> 32, path = pkey & 0xFFFFFFFFull;
            const int32_t root = P.nodes.root[node];
            atomicMax(&P.ctr->max_depth, depth + 1ull);
            if (P.nodes.tbits && depth < 5) atomicOr(&P.nodes.tbits[root], 3ull << ((2ull << depth) - 1ull + (path << 1)));  // both children's heap indices
            for (int w = 0; w < 2; ++w) {
                const int64_t c = cn + w;
                P.nodes.root[c] = root;
                P.nodes.parent[c] = node;
                P.nodes.nseg[c] = 1;
                P.nodes.status[c] = 0;
                P.nodes.li[c] = li;
                P.nodes.lambda[c] = lambda;
                P.nodes.hit_det[c] = -1;
                P.nodes.key[c] = ((depth + 1) << 32) | (((path << 1) | (unsigned long long)w) & 0xFFFFFFFFull);
                if (RETR) P.nodes.old[c] = old_kids ? P.old.first_child[rt.old] + w : -1;  // children!: the stored child is re-walked
            }
        };
        const int32_t child_flags = ((RETR && old_kids) || 1 < P.r_max) ? 0 : F_DEAD;
        auto next_ray = [&]() {
            RayS r = o.next;  // (E0 of a PolarizedRay stays where it is)
            r.pos = lm.get3(0);
            r.dir = lm.get3(3);
            r.n = lm.m[7 * lm.stride];
            return r;
        };
        bool go_on = b + 1 < P.n_fuse;
        // wave-uniform decisions: every wave runs its own fused loop (no workgroup barrier per level: the four waves of a workgroup used to
        // wait for the slowest of them at every bounce); the workgroup meets again at block_alloc below.
        // Beam splitters first, wherever in the loop they are met: one reservation per wave — a run of node pairs and a run of slots in
        // P.nxt — and, while the launch has room for it (inwave_cap), the splitting lane goes on with its transmitted child in place and
        // only the reflected one waits for the next launch; otherwise both wait there and the wave's loop ends.
        const unsigned long long m_split = INW ? __ballot(split) : 0ull;
        bool kid_here = false;  // this lane goes on with its transmitted child
#if defined(BMO_NO_KEEP)
        const bool keep = false;
#else
        const bool keep = INW && P.pend.d != nullptr;  // reflected children stay with their lane (StepParams::pend)
#endif
        if (!INW) {
            if (go_on) go_on = !__any(split ? 1 : 0);
        } else if (m_split) {
            const int ns = __popcll(m_split);
            // splitting lanes whose reflected child has to go to P.nxt at once: all of them without `keep`, else those that keep one already
            const unsigned long long m_push = keep ? (m_split & __ballot(pend_node >= 0)) : m_split;
            const int np = __popcll(m_push);
            unsigned long long r1 = 0, r2 = 0;
            // the node pairs first: their running count is also the number of in-loop splits of this launch so far (every node a launch of
            // this kernel makes is made here), which the launch's room bounds — one returning atomic instead of two in a row
            if (lane_id() == 0) r2 = atomicAdd(&P.ctr->node_count, 2ull * ns);
            r2 = __shfl(r2, 0);
            if (go_on && (int64_t)((r2 - (unsigned long long)P.nodes0) / 2ull + ns) > P.inwave_cap) go_on = false;  // (kept child or not: the bound also sizes the node table)
            if (!go_on || np) {
                if (lane_id() == 0) r1 = atomicAdd(&P.ctr->next_count[P.parity], (unsigned long long)(go_on ? np : 2 * ns));
                r1 = __shfl(r1, 0);
            }
            if (split) {
                const int r = prefix_rank(m_split);
                const int64_t cn = (int64_t)r2 + 2 * r;
                if (cn + 1 < P.nodes.cap) {
                    make_children(cn);
                    const Chunk T = go_on ? P.inner[b] : P.nxt;  // (wave-uniform)
                    const int64_t ts = go_on ? jj : (int64_t)r1 + 2 * r;
                    if (ts < T.cap) write_ray(T, ts, next_ray(), (int32_t)cn, 0, -1, -1, child_flags, opl_next);
                    else atomicAdd(&P.ctr->overflow, 1ull);
                    c_node = (int32_t)cn;
                    c_k = 0;
                    c_ho = c_hs = -1;
                    c_fl = child_flags;
                    lm.m[8 * lm.stride] = opl_next;
                    if (go_on && keep && pend_node < 0) {  // the reflected child waits for this lane
                        write_ray(P.pend, jj, o.refl, (int32_t)(cn + 1), 0, -1, -1, child_flags, opl_next);
                        pend_node = (int32_t)(cn + 1);
                    } else {
                        write_next(go_on ? (int64_t)r1 + prefix_rank(m_push) : (int64_t)r1 + 2 * r + 1, o.refl, (int32_t)(cn + 1), 0, -1, -1, child_flags, opl_next);
                    }
                    kid_here = go_on;
                } else {
                    atomicAdd(&P.ctr->overflow, 1ull);
                }
            }
        }
        // a lane whose beam ends here takes up the reflected child it kept, if the wave goes on
        const bool take_kept = INW && valid && alive && !survive && !kid_here && pend_node >= 0;
        if (go_on) go_on = __any((survive || kid_here || take_kept) ? 1 : 0) != 0;
        if (go_on) {
            // go on in place: the next record of a surviving lane is written to the same slot of the next inner chunk
            const Chunk N = P.inner[b];
            if (valid) {
                if (alive && survive) {
                    int32_t fl, ho, hs;
                    next_header(fl, ho, hs);
                    write_ray(N, jj, next_ray(), node, k + 1, ho, hs, fl, opl_next);
                    c_node = node;
                    c_k = k + 1;
                    c_ho = ho;
                    c_hs = hs;
                    c_fl = fl;
                    lm.m[8 * lm.stride] = opl_next;
                } else if (take_kept) {
                    // the kept child's first record moves into the log (this level, this slot); its ray goes where a lane that goes on in
                    // place expects it: position, direction, index and optical path in the lane memory, the header in registers
                    const int64_t pc = P.pend.cap, nc = N.cap;
                    BMO_NOUNROLL
                    for (int q = 0; q < L::ND; ++q) N.d[q * nc + jj] = P.pend.d[q * pc + jj];
                    c_fl = P.pend.i[I_FLAGS * pc + jj];
                    N.i[I_NODE * nc + jj] = pend_node;
                    N.i[I_K * nc + jj] = 0;
                    N.i[I_HOBJ * nc + jj] = -1;
                    N.i[I_HSHAPE * nc + jj] = -1;
                    N.i[I_FLAGS * nc + jj] = c_fl;
                    lm.put3(0, d3{P.pend.d[0 * pc + jj], P.pend.d[1 * pc + jj], P.pend.d[2 * pc + jj]});
                    lm.put3(3, d3{P.pend.d[3 * pc + jj], P.pend.d[4 * pc + jj], P.pend.d[5 * pc + jj]});
                    lm.m[7 * lm.stride] = P.pend.d[6 * pc + jj];
                    lm.m[8 * lm.stride] = P.pend.d[L::OPL * pc + jj];
                    c_node = pend_node;
                    c_k = 0;
                    c_ho = c_hs = -1;
                    pend_node = -1;
                } else if (!kid_here) {
                    N.i[I_NODE * N.cap + jj] = -1;  // no record of this beam at this level
                    alive = false;
                }
            }
            b += 1;
            continue;
        }
        // ---- last fused bounce of this wave (kept inside the loop so that o.next / o.refl die here instead of staying live
        //      across the march of the next iteration): note how far the wave got — the in-place levels beyond are never written and
        //      never read (Chunk::wl) —, then compact into the next launch's chunk
        if (P.wave_last && lane_id() == 0) P.wave_last[gwave] = (uint8_t)b;
#if defined(BMO_DEV_TIMELINE)
        if (P.tl && (threadIdx.x & 63) == 0 && (gwave << P.lane_shift) < P.cur.count) {  // (tail waves of the grid have no slot in the timeline)
            P.tl[2 * gwave + 1] = wall_clock64();  // before the workgroup barrier of block_alloc
            const int64_t nw = (P.cur.count + (1 << P.lane_shift) - 1) >> P.lane_shift;
            atomicAdd(&P.tl[2 * nw + 0], tk0);
            atomicAdd(&P.tl[2 * nw + 1], tk1);
            atomicAdd(&P.tl[2 * nw + 2], tk2);
        }
#endif
        const SlotAlloc al = INW ? block_alloc<false>(survive, pend_node >= 0, calls, P, scratch, b) : block_alloc<true>(survive, split, calls, P, scratch, b);
        if (survive) {  // survivors first
            const int64_t slot = (int64_t)al.surv_base + prefix_rank(al.m_surv);
            int32_t fl, ho, hs;
            next_header(fl, ho, hs);
            write_next(slot, next_ray(), node, k + 1, ho, hs, fl, opl_next);
        }
        if (INW && pend_node >= 0) {  // then the reflected children their lanes did not get to: whole records, P.pend -> P.nxt
            const int64_t slot = (int64_t)al.child_base + prefix_rank(al.m_split);
            if (slot < ncap) {
                const int64_t pc = P.pend.cap;
                BMO_NOUNROLL
                for (int q = 0; q < L::ND; ++q) P.nxt.d[q * ncap + slot] = P.pend.d[q * pc + jj];
                BMO_NOUNROLL
                for (int q = 0; q < NI; ++q) P.nxt.i[q * ncap + slot] = P.pend.i[q * pc + jj];
            } else {
                atomicAdd(&P.ctr->overflow, 1ull);
            }
        }
#if !defined(BMO_NO_TILE_COST)
        if (P.tile_cost && threadIdx.x == 0) P.tile_cost[tile] = (uint32_t)wall_clock64() - P.tile_cost[tile];  // (behind block_alloc's barrier: all four waves are done)
#endif
        if (!INW && split) {  // then 2 children per splitting lane
            const int r = prefix_rank(al.m_split);
            const int64_t slot = (int64_t)al.child_base + 2 * r;
            const int64_t cn = (int64_t)al.node_base + 2 * r;
            if (cn + 1 < P.nodes.cap) {
                make_children(cn);
                write_next(slot, next_ray(), (int32_t)cn, 0, -1, -1, child_flags, opl_next);
                write_next(slot + 1, o.refl, (int32_t)(cn + 1), 0, -1, -1, child_flags, opl_next);
            } else {
                atomicAdd(&P.ctr->overflow, 1ull);
            }
        }
        return;
    }
}


// ------------------------------------------------------------------ GaussianBeamlet step (System.jl:274-318)
// The beamlet record of lane j as gauss_step_rec sees it: rays come out of the chunk when a march starts, hits go back into it when
// the march ends (they are part of the log anyway), the accumulators are read after the marches.
struct GaussRecDev {
    double* D;
    const int32_t* I;
    const NodeArrays& nodes;
    int64_t cap, j;
    int32_t node;
    double* N;     // the next level's record planes (same slot j): the rays the step produces for the beamlet that goes on are written
    int64_t ncap;  // straight into their record, planes 11 r + c
    double* G;     // staging of the reflected child's rays: [21][gcap], planes 7 r + c
    int64_t gcap;
    // wavelength index and wavelength of the beamlet: constant along a lane (children included), carried in the lane memory from the level
    // that loaded them — the node tables are indexed by beam, and with the roots in coherence order neighbouring lanes hold beams far apart:
    // six scattered reads per lane and level were 1.5 GB of the Gaussian kernel's HBM reads per config-3 solve
    int32_t li;
    double lambda;
    __device__ void put_next(int r, const RayS& x) const {
        const int64_t b = 11 * (int64_t)r;
        N[(b + 0) * ncap + j] = x.pos.x;
        N[(b + 1) * ncap + j] = x.pos.y;
        N[(b + 2) * ncap + j] = x.pos.z;
        N[(b + 3) * ncap + j] = x.dir.x;
        N[(b + 4) * ncap + j] = x.dir.y;
        N[(b + 5) * ncap + j] = x.dir.z;
        N[(b + 6) * ncap + j] = x.n;
    }
    __device__ void put_refl(int r, const RayS& x) const {
        const int64_t b = 7 * (int64_t)r;
        G[(b + 0) * gcap + j] = x.pos.x;
        G[(b + 1) * gcap + j] = x.pos.y;
        G[(b + 2) * gcap + j] = x.pos.z;
        G[(b + 3) * gcap + j] = x.dir.x;
        G[(b + 4) * gcap + j] = x.dir.y;
        G[(b + 5) * gcap + j] = x.dir.z;
        G[(b + 6) * gcap + j] = x.n;
    }
    __device__ RayS ray(int r) const {
        const int64_t b = 11 * (int64_t)r;
        RayS x;
        x.pos = {D[(b + 0) * cap + j], D[(b + 1) * cap + j], D[(b + 2) * cap + j]};
        x.dir = {D[(b + 3) * cap + j], D[(b + 4) * cap + j], D[(b + 5) * cap + j]};
        x.n = D[(b + 6) * cap + j];
        return x;
    }
    __device__ void put_hit(int r, const Hit& X) const {
        const int64_t b = 11 * (int64_t)r;
        D[(b + 7) * cap + j] = X.t;
        D[(b + 8) * cap + j] = X.n.x;
        D[(b + 9) * cap + j] = X.n.y;
        D[(b + 10) * cap + j] = X.n.z;
    }
    __device__ Hit hit(int r) const {
        const int64_t b = 11 * (int64_t)r;
        Hit X;
        X.t = D[(b + 7) * cap + j];
        X.n = {D[(b + 8) * cap + j], D[(b + 9) * cap + j], D[(b + 10) * cap + j]};
        X.obj = X.shape = -1;
        return X;
    }
    __device__ void clear_hits() const {
        const Hit X = no_hit();
        put_hit(0, X);
        put_hit(1, X);
        put_hit(2, X);
    }
    __device__ int32_t hint_obj() const { return I[I_HOBJ * cap + j]; }
    __device__ int32_t hint_shape() const { return I[I_HSHAPE * cap + j]; }
    __device__ GaussAcc acc() const {
        GaussAcc a;
        a.lenA = D[33 * cap + j];
        a.lenB = D[34 * cap + j];
        a.oplC = D[35 * cap + j];
        a.oplW = D[36 * cap + j];
        a.oplD = D[37 * cap + j];
        a.li = li;
        a.lambda = lambda;
        a.l0 = a.w0 = 0.0;  // (the splitter branch takes l0, w0, E0 from load(); nothing else reads them)
        a.E0 = {0.0, 0.0};
        return a;
    }
    __device__ GaussIn load() const {
        GaussIn g;
        g.c = ray(0);
        g.w = ray(1);
        g.d = ray(2);
        g.hint_obj = g.hint_shape = -1;  // consumed before the marches
        g.lenA = D[33 * cap + j];
        g.lenB = D[34 * cap + j];
        g.oplC = D[35 * cap + j];
        g.oplW = D[36 * cap + j];
        g.oplD = D[37 * cap + j];
        g.li = li;
        g.lambda = lambda;
        g.l0 = nodes.aux[(int64_t)node * 4 + 0];
        g.w0 = nodes.aux[(int64_t)node * 4 + 1];
        g.E0 = {nodes.aux[(int64_t)node * 4 + 2], nodes.aux[(int64_t)node * 4 + 3]};
        return g;
    }
};
// retrace: the stored rays behind the one a re-walking beamlet is at (bmo_lane.hpp gauss_step_rec `tail`), read from the previous
// solution's log
struct GaussTailDev {
    const OldSolution& O;
    int32_t old, k, old_n;
    __device__ int more() const { return old >= 0 ? old_n - (k + 1) : 0; }
    __device__ double t(int q) const {
        int64_t c, s;
        const double* D = old_record(O, old, k + 1 + q, c, s);
        return D[7 * c + s];
    }
    __device__ RayS ray(int q, int r) const {
        int64_t c, s;
        const double* D = old_record(O, old, k + 1 + q, c, s);
        const int64_t b = 11 * (int64_t)r;
        RayS x;
        x.pos = {D[(b + 0) * c + s], D[(b + 1) * c + s], D[(b + 2) * c + s]};
        x.dir = {D[(b + 3) * c + s], D[(b + 4) * c + s], D[(b + 5) * c + s]};
        x.n = D[(b + 6) * c + s];
        return x;
    }
};
// retrace: a beamlet that re-walks its stored path without a probe starts without a hint
struct GaussRecDevNoHint : GaussRecDev {
    __device__ int32_t hint_obj() const { return -1; }
    __device__ int32_t hint_shape() const { return -1; }
};

// One launch advances every active beamlet by up to P.n_fuse bounces, like step_kernel: level b reads its records from C = (b == 0 ? P.cur :
// P.inner[b - 1]) at slot j and writes the rays of the beamlet that goes on IN PLACE into P.inner[b] at the same slot, as gauss_step_rec
// produces them (no staging copy, no compaction between the levels; P.inner has n_fuse entries here — the last one only holds the rays of
// the last fused level until they are compacted into P.nxt).  Beam splitters are handled in the loop: the transmitted child goes on in
// place, the reflected child's rays wait in the staging planes and are pushed to P.nxt (StepParams::inwave_cap).
template <int EXT, bool RETR>
__global__ __launch_bounds__(BMO_BLOCK, BMO_MIN_WAVES_GAUSS) void step_kernel_gauss(StepParams P) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const SceneView S = view_of((const char*)P.blob, &P.hdr);  // scene tables: global memory, scalar loads (bmo_lane.hpp)
    char* scratch = lds;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        P.ctr->next_count[P.parity ^ 1] = 0;
        P.ctr->max_level[P.parity ^ 1] = 0;
        P.ctr->inwave[P.parity ^ 1] = 0;
    }
    const unsigned tile = P.tile_order ? (unsigned)P.tile_order[blockIdx.x] : tile_of_block(P.reverse, blockIdx.x, gridDim.x);
#if !defined(BMO_NO_TILE_COST)
    if (P.tile_cost && threadIdx.x == 0) P.tile_cost[tile] = (uint32_t)wall_clock64();  // start stamp; turned into the tile's time at the end
#endif
    const int64_t gwave = ((int64_t)tile * blockDim.x + threadIdx.x) >> 6;  // wave of the grid
    const int64_t j = (gwave << P.lane_shift) + lane_id();
    const int64_t m = P.cur.count;
    const bool valid = j < m && lane_id() < (1 << P.lane_shift);
    bool alive = valid;
    uint32_t calls = 0;
    const int64_t ncap = P.nxt.cap;
    int32_t pend_node = -1;  // node of the reflected child this lane keeps for itself in P.gkeep (StepParams::pend), -1: none
    // (measured twice, profiles/r04_ab_scheduling.txt items 5 and 11: while the leaves' dispatch still cost this kernel 196 B of scratch the kept
    //  child made it 228 B and config 3 2 % slower, 10.56 against 10.34 ms; with the pinned dispatch — 140 B, 164 with the kept child — it is
    //  2 % FASTER, 9.41 against 9.62 ms, and one launch per solve.  -DBMO_NO_GAUSS_KEEP compiles it out.)
#if defined(BMO_NO_GAUSS_KEEP)
    const bool keep = false;
#else
    const bool keep = P.gkeep != nullptr;
#endif
    // accumulators and header of a record whose rays are in place already
    auto write_tail = [&](const Chunk& T, int64_t slot, int32_t nd, int32_t kk, int32_t ho, int32_t hs, int32_t fl, double lenA, double lenB, double oplC,
                          double oplW, double oplD) {
        const int64_t tcap = T.cap;
        double* D = T.d;
        int32_t* I = T.i;
        D[33 * tcap + slot] = lenA;
        D[34 * tcap + slot] = lenB;
        D[35 * tcap + slot] = oplC;
        D[36 * tcap + slot] = oplW;
        D[37 * tcap + slot] = oplD;
        I[I_NODE * tcap + slot] = nd;
        I[I_K * tcap + slot] = kk;
        I[I_HOBJ * tcap + slot] = ho;
        I[I_HSHAPE * tcap + slot] = hs;
        I[I_FLAGS * tcap + slot] = fl;
    };
    // a whole record in P.nxt: rays from `src` (plane r * rstride + c of capacity scap, at slot j), then accumulators and header
    auto write_next = [&](int64_t slot, const double* src, int64_t scap, int rstride, int32_t nd, int32_t kk, int32_t ho, int32_t hs, int32_t fl, double lenA,
                          double lenB, double oplC, double oplW, double oplD) {
        if (slot >= ncap) {
            atomicAdd(&P.ctr->overflow, 1ull);
            return;
        }
        double* D = P.nxt.d;
        BMO_NOUNROLL
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 7; ++c) D[(11 * r + c) * ncap + slot] = src[(int64_t)(rstride * r + c) * scap + j];
        write_tail(P.nxt, slot, nd, kk, ho, hs, fl, lenA, lenB, oplC, oplW, oplD);
    };
    int b = 0;
    for (;;) {
        const Chunk C = b == 0 ? P.cur : P.inner[b - 1];
        const Chunk N = P.inner[b];
        const int64_t cap = C.cap;
        bool survive = false, split = false;
        int32_t node = -1, k = 0;
        GaussOut o;
        o.outcome = OUT_MISS;
        RetraceLane rt;
        bool still = false, old_kids = false;
        if (alive) {
            double* D = C.d;
            int32_t* I = C.i;
            node = I[I_NODE * cap + j];
            k = I[I_K * cap + j];
            const int32_t flags = I[I_FLAGS * cap + j];
            int status = 0;
            o.hit_obj = o.hit_shape = -1;
            o.det_slot = -1;
            o.det = P.nodes.hit + (int64_t)node * 27;  // a detector hit ends the beamlet: its records go straight to the node's slot
            bool no_hint = false;
            if (RETR) {
                rt = retrace_lane(P, node, k);
                no_hint = rt.old >= 0 && !rt.probe;
            }
            // (lane memory slots 9, 10: wavelength index and wavelength; loaded where the lane's first record of this launch is read)
            const LaneMem lmw{reinterpret_cast<double*>(scratch + 64) + BMO_CC_MAX * BMO_BLOCK + threadIdx.x, BMO_BLOCK};
            if (b == 0) {
                lmw.m[9 * lmw.stride] = (double)P.nodes.li[node];
                lmw.m[10 * lmw.stride] = P.nodes.lambda[node];
            }
            GaussRecDev rec{D, I, P.nodes, cap, j, node, N.d, N.cap, P.gstage, P.gstage_cap, (int32_t)lmw.m[9 * lmw.stride], lmw.m[10 * lmw.stride]};
            if ((flags & F_DEAD) || (RETR && rt.old >= 0 && !rt.probe && !rt.fresh_allowed)) {
                status = BMO_NODE_RMAX;
                rec.clear_hits();
            } else {
                ChildCache cc{reinterpret_cast<double*>(scratch + 64) + threadIdx.x, BMO_BLOCK, 0};
                const LaneMem lm{reinterpret_cast<double*>(scratch + 64) + BMO_CC_MAX * BMO_BLOCK + threadIdx.x, BMO_BLOCK};
                if (RETR && no_hint) {
                    GaussRecDevNoHint rn{{D, I, P.nodes, cap, j, node, N.d, N.cap, P.gstage, P.gstage_cap, rec.li, rec.lambda}};
                    gauss_step_rec<EXT, RETR>(S, rn, o, calls, cc, lm, rt.probe, rt.probe_obj, rt.fresh_allowed, &rt.missed);
                } else if (RETR) {
                    const GaussTailDev tail{P.old, rt.old, k, rt.old_n};
                    gauss_step_rec<EXT, RETR, GaussRecDev, GaussTailDev>(S, rec, o, calls, cc, lm, rt.probe, rt.probe_obj, rt.fresh_allowed, &rt.missed, tail);
                } else {
                    gauss_step_rec<EXT, RETR>(S, rec, o, calls, cc, lm, rt.probe, rt.probe_obj, rt.fresh_allowed, &rt.missed);
                }
                status = o.status;
                if (o.outcome == OUT_CONTINUE) survive = true;
                else if (o.outcome == OUT_SPLIT) {
                    split = true;
                    status |= BMO_NODE_SPLIT | BMO_NODE_STOPPED;
                } else if (o.outcome == OUT_STOP) status |= BMO_NODE_STOPPED;
            }
            I[I_OBJ * cap + j] = o.hit_obj;
            I[I_SHAPE * cap + j] = o.hit_shape;
            if (RETR) {
                still = rt.old >= 0 && rt.probe && !rt.missed;
                old_kids = still && P.old.first_child[rt.old] >= 0;  // (not the SPLIT status: kept stale children hang on a beamlet that did not split)
                if (!survive && old_kids && !split && !(flags & F_DEAD)) {
                    // The re-walk ends in a `nothing` interaction before the splitter the stored beamlet ended on: the reference keeps the children
                    // (System.jl:393-400) and retraces each from its stored first rays.  They take a splitter's children's places: the transmitted
                    // one's rays in the next level's record, the reflected one's in the staging planes; w0 and E0 stay the stored ones.
                    status |= BMO_NODE_RETRACE_STALE;
                    const int32_t oc = P.old.first_child[rt.old];
                    BMO_NOUNROLL
                    for (int w = 0; w < 2; ++w) {
                        int64_t hc, hs;
                        const double* H = old_record(P.old, oc + w, 0, hc, hs);
                        BMO_NOUNROLL
                        for (int r = 0; r < 3; ++r) {
                            const int64_t b11 = 11 * (int64_t)r;
                            RayS x;
                            x.pos = {H[(b11 + 0) * hc + hs], H[(b11 + 1) * hc + hs], H[(b11 + 2) * hc + hs]};
                            x.dir = {H[(b11 + 3) * hc + hs], H[(b11 + 4) * hc + hs], H[(b11 + 5) * hc + hs]};
                            x.n = H[(b11 + 6) * hc + hs];
                            if (w == 0) rec.put_next(r, x);
                            else rec.put_refl(r, x);
                        }
                    }
                    o.child_l0 = o.lenA + P.nodes.aux[(int64_t)node * 4 + 0];  // length(parent chief): its rays up to this hit + its own parents (Beam.jl:125-130)
                    o.child_w0 = P.old.aux[(int64_t)oc * 4 + 1];
                    o.Et = {P.old.aux[(int64_t)oc * 4 + 2], P.old.aux[(int64_t)oc * 4 + 3]};
                    o.Er = {P.old.aux[(int64_t)(oc + 1) * 4 + 2], P.old.aux[(int64_t)(oc + 1) * 4 + 3]};
                    split = true;  // from here on like a splitter's children, but the beamlet did not split: no BMO_NODE_SPLIT
                }
                // a split before the end of the stored path: the reference sizes the children (w0, E0) with the stale tail still attached
                // to the beamlet (gauss_parameters(gauss, length(gauss)), ThinBeamsplitter.jl:125) — gauss_step_rec's `tail`; the flag stays as a note
                if ((status & BMO_NODE_SPLIT) && still && k + 1 < rt.old_n) status |= BMO_NODE_RETRACE_STALE;
                if (rt.old >= 0 && !(survive && still && k + 1 < rt.old_n)) P.nodes.old[node] = -1;
            }
            if (!survive) {
                P.nodes.nseg[node] = k + 1;
                P.nodes.status[node] = status;
                if (o.det_slot >= 0 && !(flags & F_DEAD)) P.nodes.hit_det[node] = o.det_slot;
            }
        }
        // header of the record that follows a surviving bounce
        auto next_header = [&](int32_t& fl, int32_t& ho, int32_t& hs) {
            fl = (k + 2 < P.r_max) ? 0 : F_DEAD;
            ho = o.hint_obj;
            hs = o.hint_shape;
            if (RETR && still) {
                if (k + 1 < rt.old_n) fl = 0;
                else ho = hs = -1;
            }
        };
        bool go_on = b + 1 < P.n_fuse;
        // beam splitters, wherever in the loop they are met (see step_kernel): one reservation per wave
        const unsigned long long m_split = __ballot(split);
        bool kid_here = false;
        if (m_split) {
            const int ns = __popcll(m_split);
            const unsigned long long m_push = keep ? (m_split & __ballot(pend_node >= 0)) : m_split;  // reflected children that go to P.nxt at once
            const int np = __popcll(m_push);
            unsigned long long r1 = 0, r2 = 0;
            if (lane_id() == 0) r2 = atomicAdd(&P.ctr->node_count, 2ull * ns);  // (also counts the launch's in-loop splits: see step_kernel)
            r2 = __shfl(r2, 0);
            if (go_on && (int64_t)((r2 - (unsigned long long)P.nodes0) / 2ull + ns) > P.inwave_cap) go_on = false;
            if (!go_on || np) {
                if (lane_id() == 0) r1 = atomicAdd(&P.ctr->next_count[P.parity], (unsigned long long)(go_on ? np : 2 * ns));
                r1 = __shfl(r1, 0);
            }
            if (split) {
                const int r = prefix_rank(m_split);
                const int64_t cn = (int64_t)r2 + 2 * r;
                if (cn + 1 < P.nodes.cap) {
                    const unsigned long long pkey = P.nodes.key[node];
                    const unsigned long long depth = pkey >> 32, path = pkey & 0xFFFFFFFFull;
                    const int32_t root = P.nodes.root[node];
                    atomicMax(&P.ctr->max_depth, depth + 1ull);
                    if (P.nodes.tbits && depth < 5) atomicOr(&P.nodes.tbits[root], 3ull << ((2ull << depth) - 1ull + (path << 1)));
                    for (int w = 0; w < 2; ++w) {
                        const int64_t c = cn + w;
                        P.nodes.root[c] = root;
                        P.nodes.parent[c] = node;
                        P.nodes.nseg[c] = 1;
                        P.nodes.status[c] = 0;
                        P.nodes.li[c] = P.nodes.li[node];
                        P.nodes.lambda[c] = P.nodes.lambda[node];
                        P.nodes.hit_det[c] = -1;
                        P.nodes.key[c] = ((depth + 1) << 32) | (((path << 1) | (unsigned long long)w) & 0xFFFFFFFFull);
                        P.nodes.aux[c * 4 + 0] = o.child_l0;
                        P.nodes.aux[c * 4 + 1] = o.child_w0;
                        P.nodes.aux[c * 4 + 2] = w == 0 ? o.Et.re : o.Er.re;
                        P.nodes.aux[c * 4 + 3] = w == 0 ? o.Et.im : o.Er.im;
                        if (RETR) {
                            const int32_t oc = old_kids ? P.old.first_child[rt.old] + w : -1;
                            P.nodes.old[c] = oc;
                            if (oc >= 0) P.nodes.aux[c * 4 + 1] = P.old.aux[(int64_t)oc * 4 + 1];  // _modify_beam_head! keeps the stored w0 (Gaussian.jl:154-161)
                        }
                    }
                    const int32_t fl = ((RETR && old_kids) || 1 < P.r_max) ? 0 : F_DEAD;
                    // children: chief inherits the parent chain (parent! Gaussian.jl:113-117); waist/div beams have no parent
                    if (go_on) write_tail(N, j, (int32_t)cn, 0, -1, -1, fl, 0.0, o.child_l0, o.oplC, 0.0, 0.0);  // (its rays are in place)
                    else write_next((int64_t)r1 + 2 * r, N.d, N.cap, 11, (int32_t)cn, 0, -1, -1, fl, 0.0, o.child_l0, o.oplC, 0.0, 0.0);
                    if (go_on && keep && pend_node < 0) {  // the reflected child waits for this lane (the staging planes serve the next split)
                        const int64_t gc = P.gstage_cap;
                        BMO_NOUNROLL
                        for (int q = 0; q < 21; ++q) P.gkeep[q * gc + j] = P.gstage[q * gc + j];
                        P.gkeep[21 * gc + j] = o.child_l0;
                        P.gkeep[22 * gc + j] = o.oplC;
                        P.gkeep[23 * gc + j] = (double)fl;
                        pend_node = (int32_t)(cn + 1);
                    } else {
                        write_next(go_on ? (int64_t)r1 + prefix_rank(m_push) : (int64_t)r1 + 2 * r + 1, P.gstage, P.gstage_cap, 7, (int32_t)(cn + 1), 0, -1, -1, fl, 0.0,
                                   o.child_l0, o.oplC, 0.0, 0.0);
                    }
                    kid_here = go_on;
                } else {
                    atomicAdd(&P.ctr->overflow, 1ull);
                }
            }
        }
        const bool take_kept = valid && alive && !survive && !kid_here && pend_node >= 0;  // this lane's beamlet ends here: it takes up the child it kept
        if (go_on) go_on = __any((survive || kid_here || take_kept) ? 1 : 0) != 0;
        if (go_on) {
            if (valid) {
                if (alive && survive) {
                    int32_t fl, ho, hs;
                    next_header(fl, ho, hs);
                    write_tail(N, j, node, k + 1, ho, hs, fl, o.lenA, o.lenB, o.oplC, o.oplW, o.oplD);
                } else if (take_kept) {
                    const int64_t gc = P.gstage_cap, nc = N.cap;
                    BMO_NOUNROLL
                    for (int r = 0; r < 3; ++r)
                        for (int c = 0; c < 7; ++c) N.d[(11 * r + c) * nc + j] = P.gkeep[(int64_t)(7 * r + c) * gc + j];
                    write_tail(N, j, pend_node, 0, -1, -1, (int32_t)P.gkeep[23 * gc + j], 0.0, P.gkeep[21 * gc + j], P.gkeep[22 * gc + j], 0.0, 0.0);
                    pend_node = -1;
                } else if (!kid_here) {
                    N.i[I_NODE * N.cap + j] = -1;  // no record of this beamlet at this level
                    alive = false;
                }
            }
            b += 1;
            continue;
        }
        // ---- last fused bounce of this wave
        if (P.wave_last && lane_id() == 0) P.wave_last[gwave] = (uint8_t)b;
        const SlotAlloc al = block_alloc<false>(survive, pend_node >= 0, calls, P, scratch, b);
        if (survive) {
            const int64_t slot = (int64_t)al.surv_base + prefix_rank(al.m_surv);
            int32_t fl, ho, hs;
            next_header(fl, ho, hs);
            write_next(slot, N.d, N.cap, 11, node, k + 1, ho, hs, fl, o.lenA, o.lenB, o.oplC, o.oplW, o.oplD);
        }
        if (pend_node >= 0) {  // the reflected children their lanes did not get to
            const int64_t gc = P.gstage_cap;
            write_next((int64_t)al.child_base + prefix_rank(al.m_split), P.gkeep, gc, 7, pend_node, 0, -1, -1, (int32_t)P.gkeep[23 * gc + j], 0.0, P.gkeep[21 * gc + j],
                       P.gkeep[22 * gc + j], 0.0, 0.0);
        }
        if (P.tile_cost && threadIdx.x == 0) P.tile_cost[tile] = (uint32_t)wall_clock64() - P.tile_cost[tile];
        return;
    }
}

// ------------------------------------------------------------------ small helper kernels
// Coherence key of a root ray: the set of candidates (first 64 of the candidate table) whose bounding sphere the ray's line meets —
// what the mask pre-pass of trace_all computes for the first bounce.  Rays with equal keys walk the same slots, so putting them next
// to each other makes waves whose lanes agree on the shape they march (bmo_lane.hpp "scalar scene access": one waterfall pass).
__global__ void root_key_kernel(const char* __restrict__ blob, BlobHeader hdr, const double* __restrict__ planes, int64_t n, unsigned long long* __restrict__ keys,
                                int32_t* __restrict__ ids) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const SceneView S = view_of(blob, &hdr);
    const d3 pos{planes[0 * n + j], planes[1 * n + j], planes[2 * n + j]}, dir{planes[3 * n + j], planes[4 * n + j], planes[5 * n + j]};
    unsigned long long mask = 0;
    const int nc = S.n_cands < 64 ? S.n_cands : 64;
    for (int i = 0; i < nc; ++i) {
        const BMO_KONST Cand& cd = S.cands[i];
        mask |= (unsigned long long)!cull_miss(cd.cx, cd.cy, cd.cz, cd.R, pos, dir) << i;
    }
    keys[j] = mask;
    ids[j] = (int32_t)j;
}

// Coherence order of the root rays (round 4).  A wave works on one shape at a time and a level of a wave lasts as long as its slowest
// lane, so the 64 rays of a wave should be NEIGHBOURS IN RAY SPACE — same elements, similar marches, ending together.  Bundle order does not
// give that: a disc source numbers its rays along a spiral (BeamGroups.jl:232-243: equal radius, every azimuth) and draws directions at random.
// A ray's line is described by the two points where it enters and leaves the scene's bounding sphere (the two-sphere parametrisation of a
// light field: position and direction weigh in with the leverage they have over the scene, no scale to choose); the rays are sorted by a
// Morton code over those six coordinates, scaled to the bundle's own extent.  In front of the code sits the ray's distance from the bundle's
// middle in that space, farthest first: the marginal rays are the ones that graze barrels and rims — marches of 500 - 1 000 evaluations,
// one dependent chain of ~1 ms each — and a launch ends when its slowest wave does, so they have to START first (the hardware hands out
// workgroups in index order).  Beam nodes keep the bundle's numbering: result order, detector order and retrace do not see any of this.
__global__ void root_chord_kernel(const double* __restrict__ planes, int64_t n, double cx, double cy, double cz, double R, double* __restrict__ chord) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const double px = planes[0 * n + j], py = planes[1 * n + j], pz = planes[2 * n + j], dx = planes[3 * n + j], dy = planes[4 * n + j], dz = planes[5 * n + j];
    const double ox = cx - px, oy = cy - py, oz = cz - pz;
    const double dd = dx * dx + dy * dy + dz * dz, b = ox * dx + oy * dy + oz * dz, cc = ox * ox + oy * oy + oz * oz - R * R;
    const double disc = b * b - dd * cc;
    double e[6] = {cx, cy, cz, cx, cy, cz};  // (a line that misses the sphere meets nothing: it sorts to the middle, where it disturbs nobody)
    if (dd > 0.0 && disc >= 0.0) {
        const double sq = sqrt(disc), t0 = (b - sq) / dd, t1 = (b + sq) / dd;
        e[0] = px + t0 * dx, e[1] = py + t0 * dy, e[2] = pz + t0 * dz;
        e[3] = px + t1 * dx, e[4] = py + t1 * dy, e[5] = pz + t1 * dz;
    }
    for (int q = 0; q < 6; ++q) chord[q * n + j] = (e[q] == e[q]) ? e[q] : (q % 3 == 0 ? cx : (q % 3 == 1 ? cy : cz));
}
// lim[0..5] minima, lim[6..11] maxima of the six chord coordinates over the bundle
__global__ void root_order_key_kernel(const double* __restrict__ chord, int64_t n, const double* __restrict__ lim, unsigned long long* __restrict__ keys, int32_t* __restrict__ ids) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    unsigned long long key = 0;
    double far2 = 0.0;
    for (int g = 0; g < 2; ++g) {  // entry points, exit points: one scale per point set (isotropic)
        double hw = 0.0, mid[3];
        for (int a = 0; a < 3; ++a) {
            const double lo = lim[3 * g + a], hi = lim[6 + 3 * g + a];
            mid[a] = 0.5 * (lo + hi);
            hw = fmax(hw, 0.5 * (hi - lo));
        }
        double r2 = 0.0;
        for (int a = 0; a < 3; ++a) {
            const double u = hw > 0.0 ? (chord[(3 * g + a) * n + j] - mid[a]) / hw : 0.0;  // [-1, 1]
            r2 += u * u;
            int q = (int)floor(u * 512.0 + 512.0);
            q = q < 0 ? 0 : (q > 1023 ? 1023 : q);
            unsigned long long v = (unsigned long long)q, sp = 0;
            for (int bit = 0; bit < 10; ++bit) sp |= ((v >> bit) & 1ull) << (6 * bit);
            key |= sp << (3 * g + a);
        }
        far2 = fmax(far2, r2);
    }
    // 16 coarse rings around the middle of the bundle in front of the code, outermost first (measured: the ragged config-2 bundle 9.0 ms
    // with them, 9.7 ms without; finer rings cost the multi-train scene 10 %): marginal rays — the ones that graze barrels — sit together
    // and start first when no feedback orders the tiles yet
    int ring = (int)(sqrt(far2) * 12.0);
    ring = ring > 15 ? 15 : ring;
    keys[j] = ((unsigned long long)(15 - ring) << 60) | key;
    ids[j] = (int32_t)j;
}
// Is the bundle's OWN numbering coherent already?  A disc source numbers its rays along a spiral (BeamGroups.jl:232-243): neighbours in the
// bundle sit at the same distance from the bundle's axis at every azimuth, and in a system that is close to rotationally symmetric about
// that axis such rays share their path exactly — 64 of them make a better wave than any patch of the Morton code (1/30 of the bundle wide in
// radius): SURVEY 8(d)'s collimated disc runs 40 % slower in Morton order.  Measure: the mean jump, from one ray to the next in bundle order,
// of the entry point's distance from the middle of the bundle's entry points (in units of the bundle's half width).  sum[0] += the jumps.
__global__ void root_ring_jump_kernel(const double* __restrict__ chord, int64_t n, const double* __restrict__ lim, double* __restrict__ sum) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double hw = 0.0, mid[3];
    for (int a = 0; a < 3; ++a) {
        mid[a] = 0.5 * (lim[a] + lim[6 + a]);
        hw = fmax(hw, 0.5 * (lim[6 + a] - lim[a]));
    }
    double jump = 0.0;
    if (j + 1 < n && hw > 0.0) {
        double r[2];
        for (int q = 0; q < 2; ++q) {
            double r2 = 0.0;
            for (int a = 0; a < 3; ++a) {
                const double u = (chord[a * n + j + q] - mid[a]) / hw;
                r2 += u * u;
            }
            r[q] = sqrt(r2);
        }
        jump = fabs(r[1] - r[0]);
    }
    for (int off = 32; off > 0; off >>= 1) jump += __shfl_down(jump, off);
    if ((threadIdx.x & 63) == 0 && jump != 0.0) atomicAdd(sum, jump);
}

// Slot s of the first chunk holds root ray perm[s] (perm == nullptr: ray s); `slot_planes` are the batch's planes in slot order (the
// batch itself when perm == nullptr, its binned copy otherwise), so every read and write here is coalesced.  Beam nodes keep the
// bundle's numbering (root i is node i): everything downstream — result order, detector order, retrace — is independent of where a
// ray sits in the chunk.  Thread i fills slot i of the chunk and node i of the node table.
template <int KIND>
__global__ void init_roots_kernel(const double* __restrict__ planes, const double* __restrict__ slot_planes, const int32_t* __restrict__ lambda_idx,
                                  const int32_t* __restrict__ perm, int64_t n, Chunk c0, NodeArrays nodes, int32_t r_max, int32_t n_planes) {
    using L = Layout<KIND>;
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int64_t cap = c0.cap;
    const double* Q = slot_planes;
    const bool cont = KIND == BMO_BEAM_GAUSSIAN && n_planes >= BMO_PLANES_GAUSSIAN_CONTINUED;
    if (KIND == BMO_BEAM_GAUSSIAN) {
        for (int b = 0; b < 3; ++b) {
            for (int p = 0; p < 6; ++p) c0.d[(11 * b + p) * cap + j] = Q[(6 * b + p) * n + j];
            c0.d[(11 * b + 6) * cap + j] = Q[19 * n + j];
        }
        // accumulated lengths: zero for a fresh beamlet; a batch that continues solved beamlets brings them along (include/bmo.h)
        c0.d[33 * cap + j] = cont ? Q[25 * n + j] : 0.0;  // lenA
        c0.d[34 * cap + j] = cont ? Q[26 * n + j] : 0.0;  // lenB
        c0.d[35 * cap + j] = cont ? Q[28 * n + j] : 0.0;  // oplC
        c0.d[36 * cap + j] = cont ? Q[29 * n + j] : 0.0;  // oplW
        c0.d[37 * cap + j] = cont ? Q[30 * n + j] : 0.0;  // oplD
    } else {
        for (int p = 0; p < 6; ++p) c0.d[p * cap + j] = Q[p * n + j];
        c0.d[6 * cap + j] = Q[7 * n + j];
        if (KIND == BMO_BEAM_POLARIZED)
            for (int p = 0; p < 6; ++p) c0.d[(11 + p) * cap + j] = Q[(8 + p) * n + j];
        c0.d[L::OPL * cap + j] = 0.0;
    }
    c0.i[I_NODE * cap + j] = perm ? perm[j] : (int32_t)j;
    c0.i[I_K * cap + j] = 0;
    c0.i[I_HOBJ * cap + j] = -1;
    c0.i[I_HSHAPE * cap + j] = -1;
    c0.i[I_FLAGS * cap + j] = (1 < r_max) ? 0 : F_DEAD;
    // node j = root ray j of the bundle
    if (KIND == BMO_BEAM_GAUSSIAN) {
        nodes.aux[j * 4 + 0] = cont ? planes[27 * n + j] : 0.0;  // l0
        nodes.aux[j * 4 + 1] = planes[20 * n + j];
        nodes.aux[j * 4 + 2] = planes[21 * n + j];
        nodes.aux[j * 4 + 3] = planes[22 * n + j];
    }
    nodes.root[j] = (int32_t)j;
    nodes.parent[j] = -1;
    nodes.nseg[j] = 1;
    nodes.status[j] = 0;
    nodes.li[j] = lambda_idx[j];
    nodes.lambda[j] = planes[(KIND == BMO_BEAM_GAUSSIAN ? 18 : 6) * n + j];
    nodes.hit_det[j] = -1;
    nodes.key[j] = 0;
    if (nodes.tbits) nodes.tbits[j] = 1ull;  // the root itself: heap index 0
}
// the batch's planes in slot order: out[p][s] = in[p][perm[s]]
__global__ void bin_planes_kernel(const double* __restrict__ in, const int32_t* __restrict__ perm, int64_t n, int n_planes, double* __restrict__ out) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const int64_t j = perm[s];
    for (int p = 0; p < n_planes; ++p) out[(int64_t)p * n + s] = in[(int64_t)p * n + j];
}

// retrace tables of a finished solution
__global__ void old_obj_scatter_kernel(Chunk c, int32_t chunk_index, const int32_t* __restrict__ rec_start, int32_t* __restrict__ rec_obj,
                                       int64_t* __restrict__ rec_loc) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= c.count) return;
    const int32_t node = chunk_node(c, j);
    if (node < 0) return;  // hole of a fused level
    const int32_t k = c.i[I_K * c.cap + j];
    rec_obj[(int64_t)rec_start[node] + k] = c.i[I_OBJ * c.cap + j];
    rec_loc[(int64_t)rec_start[node] + k] = ((int64_t)chunk_index << OLD_LOC_SHIFT) | j;
}
__global__ void old_first_child_kernel(const int32_t* __restrict__ parent, const unsigned long long* __restrict__ key, int64_t n, int32_t* __restrict__ first_child) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    const int32_t p = parent[c];
    if (p >= 0 && !(key[c] & 1ull)) first_child[p] = (int32_t)c;  // transmitted child (path bit 0); the reflected one is c + 1
}
// (root, depth, path) keys packed into as few bits as this solve needs: the radix sort of the final ordering runs 3-4 passes over
// 32-bit keys instead of 8 over 64-bit ones.  Trees of up to MAX_PATH_LEVELS levels.
constexpr int MAX_PATH_LEVELS = 26;
__global__ void pack_keys_kernel(const int32_t* __restrict__ root_of, const unsigned long long* __restrict__ key, int64_t n, int bits_depth, int bits_path,
                                 uint32_t* __restrict__ k32, unsigned long long* __restrict__ k64) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long k = key[i];
    const unsigned long long root = (unsigned long long)root_of[i], depth = k >> 32, path = k & ((1ull << bits_path) - 1ull);
    const unsigned long long packed = (root << (bits_depth + bits_path)) | (depth << bits_path) | path;
    if (k32) k32[i] = (uint32_t)packed;
    else k64[i] = packed;
}
// ---- ordering of shallow trees (up to MAX_HEAP_LEVELS levels below the root: what one to five splitters in a row give) without a
// sort: node (depth, path) sits at index 2^depth - 1 + path of its root's tree laid out as a heap, and heap order IS (depth, path)
// order.  Every root collects the heap indices of its nodes in one 64-bit word; a node's position in the canonical order is then
// (number of nodes of the roots before its own: one scan over the roots) + (number of set bits below its own).
constexpr int MAX_HEAP_LEVELS = 5;
__device__ __forceinline__ unsigned heap_index(unsigned long long key) {
    const unsigned depth = (unsigned)(key >> 32);
    return (1u << depth) - 1u + (unsigned)(key & ((1ull << depth) - 1ull));
}
__global__ void tree_bits_kernel(const int32_t* __restrict__ root_of, const unsigned long long* __restrict__ key, int64_t n, unsigned long long* __restrict__ bits) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    atomicOr(&bits[root_of[i]], 1ull << heap_index(key[i]));
}
__global__ void tree_count_kernel(const unsigned long long* __restrict__ bits, int64_t n_roots, int32_t* __restrict__ cnt) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n_roots) cnt[r] = __popcll(bits[r]);
}
__global__ void tree_rank_kernel(const int32_t* __restrict__ root_of, const unsigned long long* __restrict__ key, int64_t n, const unsigned long long* __restrict__ bits,
                                 const int32_t* __restrict__ base, int32_t* __restrict__ order) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t r = root_of[i];
    const unsigned h = heap_index(key[i]);
    order[base[r] + __popcll(bits[r] & ((1ull << h) - 1ull))] = (int32_t)i;
}
// ---- ordering of beam trees deeper than MAX_PATH_LEVELS (cavities: a splitter facing a mirror): the path no longer fits a key, so
// the rank of every node within its tree level is built level by level from its parent's rank (a splitting beam has exactly two
// children, transmitted first): rank(child) = 2 * #(splitting parents of smaller rank) + w.
__global__ void node_depth_kernel(const unsigned long long* __restrict__ key, int64_t n, uint32_t* __restrict__ depth) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) depth[i] = (uint32_t)(key[i] >> 32);
}
__global__ void level_starts_kernel(const uint32_t* __restrict__ sorted_depth, int64_t n, int64_t max_depth, int32_t* __restrict__ start) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t d = sorted_depth[i], dprev = i ? (int64_t)sorted_depth[i - 1] : -1;
    for (int64_t x = dprev + 1; x <= d; ++x) start[x] = (int32_t)i;
    if (i == n - 1)
        for (int64_t x = d + 1; x <= max_depth + 1; ++x) start[x] = (int32_t)n;
}
__global__ void level_root_rank_kernel(const int32_t* __restrict__ ids, int32_t cnt, int32_t* __restrict__ rank) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cnt) rank[ids[i]] = i;  // level 0: the roots, in node id = bundle order (the depth sort is stable)
}
__global__ void level_flag_kernel(const int32_t* __restrict__ ids, int32_t cnt, const int32_t* __restrict__ parent, const unsigned long long* __restrict__ key,
                                  const int32_t* __restrict__ rank, int32_t* __restrict__ flag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cnt) return;
    const int32_t nd = ids[i];
    if (!(key[nd] & 1ull)) flag[rank[parent[nd]]] = 1;
}
__global__ void level_rank_kernel(const int32_t* __restrict__ ids, int32_t cnt, const int32_t* __restrict__ parent, const unsigned long long* __restrict__ key,
                                  const int32_t* __restrict__ scan, int32_t* __restrict__ rank) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cnt) return;
    const int32_t nd = ids[i];
    rank[nd] = 2 * scan[rank[parent[nd]]] + (int32_t)(key[nd] & 1ull);
}
__global__ void gather_u32_kernel(const int32_t* __restrict__ idx, const int32_t* __restrict__ src, int64_t n, uint32_t* __restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (uint32_t)src[idx[i]];
}
__global__ void root_depth_key_kernel(const int32_t* __restrict__ idx, const int32_t* __restrict__ root_of, const unsigned long long* __restrict__ key, int64_t n,
                                      int bits_depth, unsigned long long* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t nd = idx[i];
    out[i] = ((unsigned long long)root_of[nd] << bits_depth) | (key[nd] >> 32);
}
__global__ void iota_kernel(int32_t* a, int64_t n) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) a[j] = (int32_t)j;
}
// flags[d*n + i] = 1 if canonical node i recorded a hit on detector d
__global__ void hit_flags_kernel(const int32_t* order, const int32_t* hit_det, int64_t n, int32_t n_det, int32_t nsub, int32_t* flags) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t d = hit_det[order[i]];
    for (int32_t q = 0; q < n_det; ++q) flags[(int64_t)q * n + i] = (q == d) ? nsub : 0;
}
// offsets of every detector's segment in the compacted hit list + the total, in one small array (one read-back instead of nd + 2)
__global__ void hit_offsets_kernel(const int32_t* __restrict__ offs, const int32_t* __restrict__ flags, int64_t n, int32_t n_det, int32_t* __restrict__ out) {
    const int d = (int)threadIdx.x;
    if (d < n_det) out[d] = offs[(int64_t)d * n];
    if (d == n_det) out[d] = offs[(int64_t)n_det * n - 1] + flags[(int64_t)n_det * n - 1];
}
// A workgroup takes 256 consecutive canonical nodes: their (id, destination) pairs go to LDS once, then the 72-byte records are copied
// as contiguous runs, one double per thread and turn (the first form of this kernel re-read the three index tables for each of
// the 9 doubles: 152 us per C2 solve for 0.3 GB of traffic).
__global__ void hit_gather_kernel(const int32_t* __restrict__ order, const int32_t* __restrict__ hit_det, const double* __restrict__ hit, int64_t n, int32_t nsub,
                                  const int32_t* __restrict__ offs, int64_t cap, double* __restrict__ out, int32_t* __restrict__ out_node,
                                  unsigned long long* __restrict__ overflow) {
    __shared__ int32_t s_nd[256], s_pos[256];
    const int64_t i0 = (int64_t)blockIdx.x * 256;
    {
        const int64_t i = i0 + threadIdx.x;
        int32_t nd = -1, pos = -1;
        if (i < n) {
            nd = order[i];
            const int32_t d = hit_det[nd];
            if (d >= 0) {
                pos = offs[(int64_t)d * n + i];
                if ((int64_t)pos + nsub > cap) {
                    atomicAdd(overflow, 1ull);
                    pos = -1;
                } else {
                    for (int q = 0; q < nsub; ++q) out_node[pos + q] = (int32_t)i;
                }
            }
        }
        s_nd[threadIdx.x] = nd;
        s_pos[threadIdx.x] = pos;
    }
    __syncthreads();
    const int width = 9 * nsub;
    // element t = e * width + k of the workgroup's 256 records; t advances by 256 per turn: (e, k) by (256 / width, 256 % width) with a
    // carry, no division per element
    const int de = 256 / width, dk = 256 % width;
    int e = (int)threadIdx.x / width, k = (int)threadIdx.x % width;
    for (int t = threadIdx.x; t < 256 * width; t += 256) {
        const int32_t pos = s_pos[e];
        if (pos >= 0) out[(int64_t)pos * 9 + k] = hit[(int64_t)s_nd[e] * width + k];
        e += de;
        k += dk;
        if (k >= width) {
            k -= width;
            e += 1;
        }
    }
}

// Node tables in the ABI's canonical order, built on the device (the host used to loop over 3 M nodes per C2 solve):
//   rank[order[i]] = i, then for canonical node i (id nd = order[i]): root, parent's rank, nseg, status, aux; the transmitted child
//   (path bit 0) writes itself as first_child of its parent — siblings are adjacent in BFS order, no atomics.
__global__ void rank_kernel(const int32_t* __restrict__ order, int64_t n, int32_t* __restrict__ rank) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) rank[order[i]] = (int32_t)i;
}
__global__ void canon_nodes_kernel(const int32_t* __restrict__ order, const int32_t* __restrict__ rank, int64_t n, int gaussian, const int32_t* __restrict__ root,
                                   const int32_t* __restrict__ parent, const int32_t* __restrict__ nseg, const int32_t* __restrict__ status,
                                   const unsigned long long* __restrict__ key, const double* __restrict__ lambda, const double* __restrict__ aux,
                                   int32_t* __restrict__ o_root, int32_t* __restrict__ o_parent, int32_t* __restrict__ o_first_child, int32_t* __restrict__ o_nseg,
                                   int32_t* __restrict__ o_status, double* __restrict__ o_aux) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t nd = order[i];
    const int32_t p = parent[nd];
    o_root[i] = root[nd];
    o_parent[i] = p < 0 ? -1 : rank[p];
    o_nseg[i] = nseg[nd];
    o_status[i] = status[nd];
    if (gaussian) {
        o_aux[4 * i + 0] = aux[4 * (int64_t)nd + 1];
        o_aux[4 * i + 1] = aux[4 * (int64_t)nd + 2];
        o_aux[4 * i + 2] = aux[4 * (int64_t)nd + 3];
        o_aux[4 * i + 3] = lambda[nd];
    } else {
        o_aux[4 * i + 0] = lambda[nd];
        o_aux[4 * i + 1] = o_aux[4 * i + 2] = o_aux[4 * i + 3] = 0.0;
    }
    if (p >= 0 && !(key[nd] & 1ull)) o_first_child[rank[p]] = (int32_t)i;
}
// dst_base[nd] = first record of node id nd in the node-major record order
__global__ void dst_base_kernel(const int32_t* __restrict__ order, const int32_t* __restrict__ first_rec, int64_t n, int32_t* __restrict__ dst_base) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst_base[order[i]] = first_rec[i];
}
// Last segment of every beam (last(rays(beam)), Beam.jl:79): record (node, k = nseg - 1) goes to column rank[node].
__global__ void last_records_kernel(Chunk c, int planes, const int32_t* __restrict__ rank, const int32_t* __restrict__ nseg, int64_t nn, double* __restrict__ rec,
                                    int32_t* __restrict__ obj, int32_t* __restrict__ shape) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= c.count) return;
    const int32_t nd = chunk_node(c, j);
    if (nd < 0) return;  // hole of a fused level
    if (c.i[I_K * c.cap + j] != nseg[nd] - 1) return;
    const int64_t dst = rank[nd];
    for (int p = 0; p < planes; ++p) rec[(int64_t)p * nn + dst] = c.d[(int64_t)p * c.cap + j];
    obj[dst] = c.i[I_OBJ * c.cap + j];
    shape[dst] = c.i[I_SHAPE * c.cap + j];
}
// the leading n_cols columns of a [n][9] hit table, packed (a Spotdetector keeps x, y only: 16 of the 72 bytes)
__global__ void hit_columns_kernel(const double* __restrict__ src, int64_t n, int n_cols, double* __restrict__ dst) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * n_cols) return;
    const int64_t i = t / n_cols;
    dst[t] = src[i * 9 + (t - i * n_cols)];
}

// Segment log -> the ABI's node-major order (bmo_trace_result_view.rec): record j of a step chunk goes to first_rec(node) + k.
// Done in HBM so that the host receives ONE contiguous copy per table instead of re-ordering 10^7 records itself.
__global__ void order_records_kernel(Chunk c, int planes, const int32_t* __restrict__ dst_base, int64_t nr, double* __restrict__ rec,
                                     int32_t* __restrict__ obj, int32_t* __restrict__ shape) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= c.count) return;
    const int32_t nd = chunk_node(c, j);
    if (nd < 0) return;  // hole of a fused level
    const int64_t dst = (int64_t)dst_base[nd] + c.i[I_K * c.cap + j];
    for (int p = 0; p < planes; ++p) rec[(int64_t)p * nr + dst] = c.d[(int64_t)p * c.cap + j];
    obj[dst] = c.i[I_OBJ * c.cap + j];
    shape[dst] = c.i[I_SHAPE * c.cap + j];
}

// ------------------------------------------------------------------ host-side objects
// Device memory pool: hipMalloc/hipFree of the multi-GB segment log cost more than the trace itself, so
// freed blocks are kept per device and reused by the next trace (same sizes every call for a fixed batch).
struct PoolBlock {
    void* p;
    size_t bytes;
    int device;
};
std::mutex g_pool_mu;
std::vector<PoolBlock> g_pool;
size_t g_pool_bytes = 0;

void* pool_take(size_t bytes, int device, size_t& got) {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    int best = -1;
    for (int i = 0; i < (int)g_pool.size(); ++i) {
        const PoolBlock& b = g_pool[i];
        if (b.device != device || b.bytes < bytes || b.bytes > bytes + bytes / 4 + 4096) continue;
        if (best < 0 || b.bytes < g_pool[best].bytes) best = i;
    }
    if (best < 0) return nullptr;
    void* p = g_pool[best].p;
    got = g_pool[best].bytes;
    g_pool_bytes -= got;
    g_pool.erase(g_pool.begin() + best);
    return p;
}
void pool_give(void* p, size_t bytes, int device) {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    g_pool.push_back({p, bytes, device});
    g_pool_bytes += bytes;
}
void pool_release_all() {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (auto& b : g_pool) {
        (void)hipSetDevice(b.device);
        (void)hipFree(b.p);
    }
    (void)hipSetDevice(cur);
    g_pool.clear();
    g_pool_bytes = 0;
}

// Deferred release (run_trace): device and pinned blocks released on this thread while a PoolHold lives are parked in it and go
// back to the pools when it dies, after the stream has been synchronised.  PoolHold::Now lifts the deferral inside a scope whose
// releases are known to be safe at once (a chunk whose launch has completed, record_segments = 0).
struct PoolHold;
thread_local PoolHold* g_hold = nullptr;
void host_pool_give(void* p, size_t bytes);
struct PoolHold {
    hipStream_t stream;
    std::vector<PoolBlock> dev;
    std::vector<std::pair<void*, size_t>> host;
    PoolHold* prev;
    explicit PoolHold(hipStream_t s) : stream(s), prev(g_hold) { g_hold = this; }
    ~PoolHold() {
        g_hold = prev;
        (void)hipStreamSynchronize(stream);
        for (auto& b : dev) pool_give(b.p, b.bytes, b.device);
        for (auto& h : host) host_pool_give(h.first, h.second);
    }
    PoolHold(const PoolHold&) = delete;
    PoolHold& operator=(const PoolHold&) = delete;
    struct Now {
        PoolHold* saved;
        Now() : saved(g_hold) { g_hold = nullptr; }
        ~Now() { g_hold = saved; }
    };
};

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    int device = 0;
    int alloc(size_t b) {
        release();
        if (b == 0) b = 16;
        (void)hipGetDevice(&device);
        size_t got = 0;
        if (void* q = pool_take(b, device, got)) {
            p = q;
            bytes = got;
            return BMO_OK;
        }
        hipError_t e = hipMalloc(&p, b);
        if (e != hipSuccess) {
            pool_release_all();  // give cached blocks back and retry once
            e = hipMalloc(&p, b);
        }
        if (e != hipSuccess) {
            p = nullptr;
            return fail(BMO_ERR_OOM, std::string("hipMalloc: ") + hipGetErrorString(e));
        }
        bytes = b;
        return BMO_OK;
    }
    void release() {
        if (p) {
            if (g_hold) g_hold->dev.push_back({p, bytes, device});
            else pool_give(p, bytes, device);
        }
        p = nullptr;
        bytes = 0;
    }
    ~DevBuf() { release(); }
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
};

// Page-locked host memory for the tables bmo_result_view hands out: a D2H copy into pinned memory runs at PCIe speed, into
// pageable memory at a fraction of it.  Pinning costs more than the copy, so freed blocks are kept (up to a cap) and reused
// by the next view of a similar size.
constexpr size_t HOST_POOL_CAP = (size_t)16 << 30;
std::mutex g_hpool_mu;
std::vector<PoolBlock> g_hpool;
size_t g_hpool_bytes = 0;

void host_pool_give(void* p, size_t bytes) {
    std::unique_lock<std::mutex> lk(g_hpool_mu);
    if (g_hpool_bytes + bytes <= HOST_POOL_CAP) {
        g_hpool.push_back({p, bytes, 0});
        g_hpool_bytes += bytes;
    } else {
        lk.unlock();
        (void)hipHostFree(p);
    }
}

struct HostBuf {
    void* p = nullptr;
    size_t bytes = 0;
    int alloc(size_t b) {
        release();
        if (b == 0) b = 16;
        {
            std::lock_guard<std::mutex> lk(g_hpool_mu);
            int best = -1;
            for (int i = 0; i < (int)g_hpool.size(); ++i) {
                const PoolBlock& k = g_hpool[i];
                if (k.bytes < b || k.bytes > b + b / 4 + 4096) continue;
                if (best < 0 || k.bytes < g_hpool[best].bytes) best = i;
            }
            if (best >= 0) {
                p = g_hpool[best].p;
                bytes = g_hpool[best].bytes;
                g_hpool_bytes -= bytes;
                g_hpool.erase(g_hpool.begin() + best);
                return BMO_OK;
            }
        }
        if (hipHostMalloc(&p, b, hipHostMallocDefault) != hipSuccess) {
            p = nullptr;
            return fail(BMO_ERR_OOM, "hipHostMalloc failed");
        }
        bytes = b;
        return BMO_OK;
    }
    void release() {
        if (!p) return;
        if (g_hold) g_hold->host.emplace_back(p, bytes);
        else host_pool_give(p, bytes);
        p = nullptr;
        bytes = 0;
    }
    template <class T>
    T* as() const { return static_cast<T*>(p); }
    ~HostBuf() { release(); }
    HostBuf() = default;
    HostBuf(const HostBuf&) = delete;
    HostBuf& operator=(const HostBuf&) = delete;
};

}  // namespace

struct bmo_scene {
    std::vector<char> blob;
    BlobHeader hdr;
    double bound[4] = {0, 0, 0, -1};  // a sphere around the bounding spheres of all candidates (centre, radius; radius < 0: none) — root_chord_kernel
    std::vector<std::pair<int, std::unique_ptr<DevBuf>>> dev;  // per-device copy of the blob
    std::mutex dev_mu;  // the handle is shared between host threads (include/bmo.h "Threading"): the lazy per-device upload is the one mutation
    const char* device_blob(int device, int& rc) {
        std::lock_guard<std::mutex> lk(dev_mu);
        for (auto& d : dev)
            if (d.first == device) return static_cast<const char*>(d.second->p);
        auto b = std::make_unique<DevBuf>();
        rc = b->alloc(blob.size());
        if (rc) return nullptr;
        if (hipMemcpy(b->p, blob.data(), blob.size(), hipMemcpyHostToDevice) != hipSuccess) {
            rc = fail(BMO_ERR_NO_DEVICE, "hipMemcpy(scene)");
            return nullptr;
        }
        const char* p = static_cast<const char*>(b->p);
        dev.emplace_back(device, std::move(b));
        return p;
    }
};

struct bmo_device_batch {
    int device = 0;
    int kind = 0;
    int64_t n = 0;
    int n_planes = 0;
    DevBuf planes, li;
    DevBuf perm;    // slot -> root ray, roots binned by coherence key (empty: bundle order is kept; root_key_kernel)
    DevBuf binned;  // the planes in slot order (only with perm)
    // feedback for the next solve of this batch (StepParams::tile_order): time of every tile of the first launch, the tiles sorted by it
    DevBuf tile_cost, tile_order, lpt_ids, lpt_keys, lpt_tmp;
    int64_t tile_n = 0;
    bool cost_valid = false;
};

struct bmo_trace_result {
    int device = 0, kind = 0, n_detectors = 0, n_objects = 0;
    int64_t n_roots = 0, n_nodes = 0, n_records = 0;
    unsigned long long calls = 0;
    int n_steps = 0;
    double kernel_ms = 0, total_ms = 0;
    int nd = 0;  // double planes per record (device layout)
    int abi_planes = 0;
    // device state
    std::vector<std::unique_ptr<DevBuf>> arena;  // chunk storage
    int64_t view_records = 0;
    bool has_log = true;                         // false: solved with record_segments = 0, only beams and detector hits were kept
    std::vector<Chunk> chunks;
    std::vector<std::unique_ptr<DevBuf>> wave_last;  // Chunk::wl of the fused launches
    DevBuf n_root, n_parent, n_nseg, n_status, n_li, n_hitdet, n_key, n_lambda, n_hit, n_aux, order, det_data, det_node;
    DevBuf n_old;  // retrace runs only (NodeArrays::old)
    DevBuf pre_start, pre_segs, pre_opl;  // earlier segments of continued root beamlets (bmo_result_set_gauss_prefix), empty otherwise
    int64_t pre_roots = 0, pre_total = 0;
    // tables a later bmo_retrace of THIS solution needs, built on first use (OldSolution)
    std::mutex rt_mu;
    bool rt_built = false;
    DevBuf rt_rec_start, rt_rec_obj, rt_first_child, rt_rec_loc, rt_chunks;
    // time of every tile of this solve's first launch (StepParams::tile_cost), copied from the batch: a retrace of this solution with a
    // freshly uploaded batch (bmo_retrace: the wrappers upload the root heads on every call) orders its tiles by it
    DevBuf tile_cost;
    int64_t tile_n = 0;
    std::vector<int64_t> det_count, det_offset;
    // host views (filled by bmo_result_view / bmo_result_view_select), all page-locked; each part is materialised on first request
    bool nodes_viewed = false, hits_viewed = false;
    int rec_mode = 0;  // records on the host: 0 none, 1 last segment of every beam, 2 the whole log
    HostBuf h_root, h_parent, h_first_child, h_first_rec, h_last_rec, h_nseg, h_status, h_aux;
    HostBuf h_rec, h_rec_obj, h_rec_shape, h_det, h_det_node;
    DevBuf c_rank, c_first_rec;  // canonical rank of every node id / first record of every canonical node (device, kept for record views)
};

namespace {

// OldSolution tables of `prev` (device arrays keyed by prev's node ids), built once per solution.
int build_retrace_tables(bmo_trace_result* prev, hipStream_t stream) {
    std::lock_guard<std::mutex> lk(prev->rt_mu);
    if (!prev->has_log) return fail(BMO_ERR_INVALID, "retrace: the previous solution was solved with record_segments = 0 (no segment log to re-walk)");
    if (prev->rt_built) return BMO_OK;
    const int64_t nn = prev->n_nodes, nr = prev->n_records;
    if (nr >= (int64_t)1 << 31) return fail(BMO_ERR_UNSUPPORTED, "retrace: previous solution has more than 2^31 segments");
    int rc;
    if ((rc = prev->rt_rec_start.alloc((size_t)std::max<int64_t>(nn, 1) * 4)) || (rc = prev->rt_rec_obj.alloc((size_t)std::max<int64_t>(nr, 1) * 4)) ||
        (rc = prev->rt_first_child.alloc((size_t)std::max<int64_t>(nn, 1) * 4)) || (rc = prev->rt_rec_loc.alloc((size_t)std::max<int64_t>(nr, 1) * 8)) ||
        (rc = prev->rt_chunks.alloc(std::max<size_t>(prev->chunks.size(), 1) * sizeof(OldChunk))))
        return rc;
    if (prev->chunks.size() >= ((size_t)1 << (63 - OLD_LOC_SHIFT))) return fail(BMO_ERR_UNSUPPORTED, "retrace: previous solution has too many log chunks");
    if (nn > 0) {
        DevBuf tmp;
        size_t tmp_bytes = 0;
        HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, (const int32_t*)prev->n_nseg.p, (int32_t*)prev->rt_rec_start.p, (int)nn, stream));
        if ((rc = tmp.alloc(tmp_bytes))) return rc;
        HIP_TRY(hipcub::DeviceScan::ExclusiveSum(tmp.p, tmp_bytes, (const int32_t*)prev->n_nseg.p, (int32_t*)prev->rt_rec_start.p, (int)nn, stream));
        std::vector<OldChunk> oc(prev->chunks.size());
        for (size_t q = 0; q < prev->chunks.size(); ++q) oc[q] = OldChunk{prev->chunks[q].d, prev->chunks[q].cap};
        if (!oc.empty()) HIP_TRY(hipMemcpyAsync(prev->rt_chunks.p, oc.data(), oc.size() * sizeof(OldChunk), hipMemcpyHostToDevice, stream));
        for (size_t q = 0; q < prev->chunks.size(); ++q) {
            const Chunk& c = prev->chunks[q];
            if (c.count > 0)
                hipLaunchKernelGGL(old_obj_scatter_kernel, dim3((unsigned)((c.count + 255) / 256)), dim3(256), 0, stream, c, (int32_t)q,
                                   (const int32_t*)prev->rt_rec_start.p, (int32_t*)prev->rt_rec_obj.p, (int64_t*)prev->rt_rec_loc.p);
        }
        HIP_TRY(hipMemsetAsync(prev->rt_first_child.p, 0xFF, (size_t)nn * 4, stream));
        hipLaunchKernelGGL(old_first_child_kernel, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, stream, (const int32_t*)prev->n_parent.p,
                           (const unsigned long long*)prev->n_key.p, nn, (int32_t*)prev->rt_first_child.p);
        HIP_TRY(hipStreamSynchronize(stream));  // tmp goes back to the pool
        HIP_TRY(hipGetLastError());
        if (dbg_on()) {
            const int q = (int)std::min<int64_t>(nn, 6), qr = (int)std::min<int64_t>(nr, 24);
            std::vector<int32_t> a(q), b(q), c(q), d(qr);
            (void)hipMemcpy(a.data(), prev->n_nseg.p, q * 4, hipMemcpyDeviceToHost);
            (void)hipMemcpy(b.data(), prev->rt_rec_start.p, q * 4, hipMemcpyDeviceToHost);
            (void)hipMemcpy(c.data(), prev->rt_first_child.p, q * 4, hipMemcpyDeviceToHost);
            (void)hipMemcpy(d.data(), prev->rt_rec_obj.p, qr * 4, hipMemcpyDeviceToHost);
            for (int i = 0; i < q; ++i) DBG("retrace table node %d: nseg %d rec_start %d first_child %d", i, a[i], b[i], c[i]);
            std::string line;
            for (int i = 0; i < qr; ++i) line += std::to_string(d[i]) + " ";
            DBG("retrace table rec_obj: %s", line.c_str());
        }
    }
    prev->rt_built = true;
    return BMO_OK;
}

template <int KIND>
int run_trace(bmo_scene* scene, bmo_device_batch* batch, const bmo_trace_opts* opts, bmo_trace_result* R, bmo_trace_result* prev = nullptr) {
    using L = Layout<KIND>;
    const int device = batch->device;
    if (prev) {
        if (prev->device != device) return fail(BMO_ERR_INVALID, "retrace: the previous solution lives on another device");
        if (prev->kind != KIND || prev->n_roots != batch->n)
            return fail(BMO_ERR_INVALID, "retrace: batch does not match the previous solution (root count / beam kind)");
    }
    HIP_TRY(hipSetDevice(device));
    int rc = BMO_OK;
    const char* dblob = scene->device_blob(device, rc);
    if (rc) return rc;
    const int64_t n = batch->n;
    const bool has_split = scene->hdr.has_splitter != 0;
    R->device = device;
    R->kind = KIND;
    R->n_roots = n;
    R->n_detectors = scene->hdr.n_detectors;
    R->nd = L::ND;
    R->abi_planes = L::ABI;

    auto wall0 = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!dbg_on()) return;
        auto t = std::chrono::steady_clock::now();
        DBG("phase %-10s %.3f ms", what, std::chrono::duration<double, std::milli>(t - wall0).count());
        wall0 = t;
    };
    // one trace stream + timing events per device, created once (hipStreamCreate costs ~1 ms)
    struct DevCtx {
        hipStream_t stream = nullptr;
        hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
        Counters* pinned = nullptr;
        std::vector<hipEvent_t> step_ev;
        std::mutex busy;  // one trace at a time per device: the stream, events and pinned counters are shared
    };
    static std::mutex ctx_mu;
    static std::vector<std::pair<int, std::unique_ptr<DevCtx>>> ctxs;
    DevCtx* ctxp = nullptr;
    {
        std::lock_guard<std::mutex> lk(ctx_mu);
        for (auto& c : ctxs)
            if (c.first == device) ctxp = c.second.get();
        if (!ctxp) {
            auto nc = std::make_unique<DevCtx>();
            // non-blocking: no implicit synchronisation with the NULL stream, so a collective or copy another library has in flight
            // there (RCCL all-gather of the previous solve's hits) overlaps this solve
            HIP_TRY(hipStreamCreateWithFlags(&nc->stream, hipStreamNonBlocking));
            for (int q = 0; q < 4; ++q) HIP_TRY(hipEventCreate(&nc->ev[q]));
            HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&nc->pinned), sizeof(Counters), hipHostMallocDefault));
            ctxp = nc.get();
            ctxs.emplace_back(device, std::move(nc));
        }
    }
    DevCtx& ctx = *ctxp;
    std::lock_guard<std::mutex> one_trace_per_device(ctx.busy);
    hipStream_t stream = ctx.stream;
    hipEvent_t ev_t0 = ctx.ev[2], ev_t1 = ctx.ev[3];
    HIP_TRY(hipEventRecord(ev_t0, stream));
    // Every exit path — the error returns too — waits for what has been queued BEFORE the temporaries of this function go back to the
    // shared pools: a kernel or copy still in flight must not write into a block another host thread has been handed.
    // While `hold` lives, blocks released on this thread are parked in it; its destructor (it is declared before every temporary, so it
    // runs after theirs) synchronises the stream and only then hands them to the pools.
    PoolHold hold(stream);

    const int nsub = KIND == BMO_BEAM_GAUSSIAN ? 3 : 1;
    // node arrays: roots + room for children (grown on demand)
    // Children need node slots, and every launch must have room for two per record it traces (any record may split).  One splitter per
    // ray (3 n beams) then asks for 7 n slots before the launch that follows the split; growing the tables there costs a device copy of
    // every node array (0.3 ms of a 6 ms solve of config C2), so moderate batches get that room up front.  Larger ones grow on demand.
    int64_t node_cap = has_split ? (n <= ((int64_t)1 << 22) ? 7 * n + 64 : 3 * n + 64) : n;
    auto alloc_nodes = [&](int64_t cap) -> int {
        int r;
        if ((r = R->n_root.alloc(cap * 4))) return r;
        if ((r = R->n_parent.alloc(cap * 4))) return r;
        if ((r = R->n_nseg.alloc(cap * 4))) return r;
        if ((r = R->n_status.alloc(cap * 4))) return r;
        if ((r = R->n_li.alloc(cap * 4))) return r;
        if ((r = R->n_hitdet.alloc(cap * 4))) return r;
        if ((r = R->n_key.alloc(cap * 8))) return r;
        if ((r = R->n_lambda.alloc(cap * 8))) return r;
        if ((r = R->n_hit.alloc(cap * 72 * nsub))) return r;
        if ((r = R->n_aux.alloc(cap * 32))) return r;
        if (prev && (r = R->n_old.alloc(cap * 4))) return r;
        return BMO_OK;
    };
    if ((rc = alloc_nodes(node_cap))) return rc;
    DevBuf tbits_buf;  // NodeArrays::tbits (allocated below when the scene has a splitter)
    auto node_arrays = [&]() {
        NodeArrays a;
        a.root = (int32_t*)R->n_root.p;
        a.parent = (int32_t*)R->n_parent.p;
        a.nseg = (int32_t*)R->n_nseg.p;
        a.status = (int32_t*)R->n_status.p;
        a.li = (int32_t*)R->n_li.p;
        a.hit_det = (int32_t*)R->n_hitdet.p;
        a.key = (unsigned long long*)R->n_key.p;
        a.lambda = (double*)R->n_lambda.p;
        a.hit = (double*)R->n_hit.p;
        a.aux = (double*)R->n_aux.p;
        a.cap = node_cap;
        a.hit_sub = nsub;
        a.old = (int32_t*)R->n_old.p;
        a.tbits = (unsigned long long*)tbits_buf.p;
        return a;
    };
    auto grow_nodes = [&](int64_t need) -> int {
        if (need <= node_cap) return BMO_OK;
        int64_t ncap = std::max(need, node_cap * 2);
        auto mv = [&](DevBuf& b, size_t elem) -> int {
            DevBuf nb;
            int r = nb.alloc((size_t)ncap * elem);
            if (r) return r;
            if (hipMemcpyAsync(nb.p, b.p, (size_t)node_cap * elem, hipMemcpyDeviceToDevice, stream) != hipSuccess)
                return fail(BMO_ERR_NO_DEVICE, "grow nodes");
            (void)hipStreamSynchronize(stream);
            std::swap(b.p, nb.p);
            std::swap(b.bytes, nb.bytes);
            PoolHold::Now at_once;  // the old table goes back right away (the copy above has completed)
            nb.release();
            return BMO_OK;
        };
        int r;
        if ((r = mv(R->n_root, 4)) || (r = mv(R->n_parent, 4)) || (r = mv(R->n_nseg, 4)) || (r = mv(R->n_status, 4)) || (r = mv(R->n_li, 4)) ||
            (r = mv(R->n_hitdet, 4)) || (r = mv(R->n_key, 8)) || (r = mv(R->n_lambda, 8)) || (r = mv(R->n_hit, 72 * (size_t)nsub)) || (r = mv(R->n_aux, 32)))
            return r;
        if (prev && (r = mv(R->n_old, 4))) return r;
        node_cap = ncap;
        return BMO_OK;
    };

    // chunk arena: bump allocation out of large blocks
    const size_t rec_bytes = (size_t)L::ND * 8 + (size_t)NI * 4;
    size_t block_bytes = std::max<size_t>((size_t)n * rec_bytes * 6, (size_t)1 << 20);
    size_t top = 0;  // offset in the last block
    // record_segments = 0: the log is not kept — every chunk gets its own pool block and goes back to the pool as soon as its level
    // is done, so a solve holds two levels instead of all of them (beams, detector hits and counts are kept as always)
    const bool keep_log = opts->record_segments != 0;
    R->has_log = keep_log;
    auto new_chunk = [&](int64_t cap, Chunk& c) -> int {
        cap = std::max<int64_t>(cap, 1);
        const size_t cap_al = ((size_t)cap + 1) & ~(size_t)1;  // keep int planes 8-byte aligned
        const size_t need = cap_al * rec_bytes;
        if (!keep_log) {
            auto b = std::make_unique<DevBuf>();
            int r = b->alloc(need);
            if (r) return r;
            char* base = static_cast<char*>(b->p);
            c.d = reinterpret_cast<double*>(base);
            c.i = reinterpret_cast<int32_t*>(base + cap_al * (size_t)L::ND * 8);
            c.cap = (int64_t)cap_al;
            c.count = 0;
            R->arena.push_back(std::move(b));
            return BMO_OK;
        }
        if (R->arena.empty() || top + need > R->arena.back()->bytes) {
            auto b = std::make_unique<DevBuf>();
            int r = b->alloc(std::max(block_bytes, need));
            if (r) return r;
            R->arena.push_back(std::move(b));
            top = 0;
        }
        char* base = static_cast<char*>(R->arena.back()->p) + top;
        c.d = reinterpret_cast<double*>(base);
        c.i = reinterpret_cast<int32_t*>(base + cap_al * (size_t)L::ND * 8);
        c.cap = (int64_t)cap_al;
        c.count = 0;
        top += need;
        return BMO_OK;
    };
    auto drop_chunk = [&](const Chunk& c) {  // record_segments = 0 only; called behind the synchronisation of the chunk's launch
        PoolHold::Now at_once;
        for (size_t q = 0; q < R->arena.size(); ++q)
            if (R->arena[q]->p == (void*)c.d) {
                R->arena.erase(R->arena.begin() + (long)q);
                return;
            }
    };
    auto shrink_last = [&](Chunk& c, int64_t used) {
        // planes are strided by cap, so the chunk keeps its footprint; nothing to return.
        c.count = used;
    };

    DevBuf ctr_buf, shard_buf;
    if ((rc = ctr_buf.alloc(sizeof(Counters))) || (rc = shard_buf.alloc(64 * 128))) return rc;
    HIP_TRY(hipMemsetAsync(shard_buf.p, 0, 64 * 128, stream));
    Counters* d_ctr = static_cast<Counters*>(ctr_buf.p);
    Counters* h_ctr_p = ctx.pinned;  // pinned: the per-step read-back does not go through a staging copy
    Counters& h_ctr = *h_ctr_p;
    h_ctr = Counters{{0, 0}, (unsigned long long)n, 0, 0, {0, 0}};
    HIP_TRY(hipMemcpyAsync(d_ctr, h_ctr_p, sizeof h_ctr, hipMemcpyHostToDevice, stream));

    Chunk cur;
    if ((rc = new_chunk(n, cur))) return rc;
    cur.count = n;
    OldSolution old_tab{};
    if (prev) {
        if ((rc = build_retrace_tables(prev, stream))) return rc;
        old_tab.nseg = (const int32_t*)prev->n_nseg.p;
        old_tab.status = (const int32_t*)prev->n_status.p;
        old_tab.first_child = (const int32_t*)prev->rt_first_child.p;
        old_tab.rec_start = (const int32_t*)prev->rt_rec_start.p;
        old_tab.rec_obj = (const int32_t*)prev->rt_rec_obj.p;
        old_tab.aux = (const double*)prev->n_aux.p;
        old_tab.rec_loc = (const int64_t*)prev->rt_rec_loc.p;
        old_tab.chunks = (const OldChunk*)prev->rt_chunks.p;
    }
    if (has_split && n > 0 && (rc = tbits_buf.alloc((size_t)n * 8))) return rc;
    if (n > 0) {
        // a retrace re-walks the stored first ray whatever r_max says (System.jl:197); root j re-walks old node j
        hipLaunchKernelGGL((init_roots_kernel<KIND>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const double*)batch->planes.p,
                           (const double*)(batch->perm.p ? batch->binned.p : batch->planes.p), (const int32_t*)batch->li.p, (const int32_t*)batch->perm.p, n, cur,
                           node_arrays(), prev ? 0x7fffffff : opts->r_max, (int32_t)batch->n_planes);
        if (prev) hipLaunchKernelGGL(iota_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (int32_t*)R->n_old.p, n);
    }
    lap("setup");
    const uint32_t blob_bytes = (uint32_t)scene->blob.size();
    // the scene tables stay in global memory: the kernels read them with scalar loads through constant-address-space pointers
    // (bmo_lane.hpp "scalar scene access"); LDS holds the per-lane columns only
    const int use_lds = 0;
    DBG("roots initialised n=%lld blob=%u use_lds=%d", (long long)n, blob_bytes, use_lds);
    if (dbg_on()) {
        HIP_TRY(hipStreamSynchronize(stream));
        HIP_TRY(hipGetLastError());
        DBG("init kernel done");
    }
    // + block_alloc scratch + per-lane columns: child cache (BMO_CC_MAX doubles) and the lane memory of tracing_step (BMO_LANE_MEM doubles)
    const size_t lds_bytes = (use_lds ? blob_bytes : 0) + 64 + (size_t)(BMO_CC_MAX + BMO_LANE_MEM) * BMO_BLOCK * 8;
    void (*kern)(StepParams) = nullptr, (*kern_inw)(StepParams) = nullptr;  // kern_inw: the Beam kernels' variant with in-loop beam splitters
    void (*kern_wide)(StepParams) = nullptr, (*kern_inw_wide)(StepParams) = nullptr;  // the 4-waves-per-SIMD builds of the two, for large launches (step_waves)
    const int ext = scene->hdr.has_asphere ? 2 : (scene->hdr.has_meniscus ? 1 : 0);  // extended-shapes level of the kernels (bmo_lane.hpp sdf_leaf)
#if defined(BMO_DEV_RAY_LDS_ONLY)  // developer build (kernel work on one variant): everything else is refused, nothing falls back
    if constexpr (KIND == BMO_BEAM_RAY) {
        if (!prev && ext == 0) {
            kern = &step_kernel<BMO_BEAM_RAY, 0, false, false>, kern_inw = &step_kernel<BMO_BEAM_RAY, 0, false, true>;
#if !defined(BMO_MIN_WAVES)
            kern_wide = &step_kernel<BMO_BEAM_RAY, 0, false, false, 4>, kern_inw_wide = &step_kernel<BMO_BEAM_RAY, 0, false, true, 4>;
#endif
        }
    }
    if (!kern) return fail(BMO_ERR_UNSUPPORTED, "developer build: only step_kernel<RAY> is compiled in");
#elif defined(BMO_DEV_GAUSS_ONLY)  // developer build: the fresh GaussianBeamlet kernel of the plain-shapes level only
    if constexpr (KIND == BMO_BEAM_GAUSSIAN) {
        if (!prev && ext == 0) kern = &step_kernel_gauss<0, false>;
    }
    if (!kern) return fail(BMO_ERR_UNSUPPORTED, "developer build: only step_kernel_gauss<0, false> is compiled in");
#else
    if constexpr (KIND == BMO_BEAM_GAUSSIAN) {
        if (prev) kern = ext == 2 ? &step_kernel_gauss<2, true> : (ext == 1 ? &step_kernel_gauss<1, true> : &step_kernel_gauss<0, true>);
        else kern = ext == 2 ? &step_kernel_gauss<2, false> : (ext == 1 ? &step_kernel_gauss<1, false> : &step_kernel_gauss<0, false>);
    } else {
        if (prev) kern = ext == 2 ? &step_kernel<KIND, 2, true, false> : (ext == 1 ? &step_kernel<KIND, 1, true, false> : &step_kernel<KIND, 0, true, false>);
        else kern = ext == 2 ? &step_kernel<KIND, 2, false, false> : (ext == 1 ? &step_kernel<KIND, 1, false, false> : &step_kernel<KIND, 0, false, false>);
        if (prev) kern_inw = ext == 2 ? &step_kernel<KIND, 2, true, true> : (ext == 1 ? &step_kernel<KIND, 1, true, true> : &step_kernel<KIND, 0, true, true>);
        else kern_inw = ext == 2 ? &step_kernel<KIND, 2, false, true> : (ext == 1 ? &step_kernel<KIND, 1, false, true> : &step_kernel<KIND, 0, false, true>);
#if !defined(BMO_MIN_WAVES)
        if constexpr (KIND == BMO_BEAM_RAY) {
            if (!prev && ext == 0) kern_wide = &step_kernel<BMO_BEAM_RAY, 0, false, false, 4>, kern_inw_wide = &step_kernel<BMO_BEAM_RAY, 0, false, true, 4>;
        }
#endif
    }
#endif
    if (lds_bytes > 48 * 1024) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        if (kern_inw) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern_inw), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        if (kern_wide) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern_wide), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        if (kern_inw_wide) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern_inw_wide), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    }

    int64_t n_nodes = n;
    double kernel_ms = 0;
    int steps = 0;
    // bounces per launch: fewer launches and host round trips; holes instead of compaction inside a launch.  BMO_FUSE=1 (BMO_FUSE_GAUSS=1)
    // restores one launch per bounce level.
    // measured on C1-C5: 8 levels per launch beat 4 by 1-4 % at 10^6 beams and by 8-10 % below 10^5; with per-wave loops 16 beat 8 by
    // another 1-3 %, and 32 beat 16 by 1-2.5 % (config C2 reaches its splitter, level 17, in the first launch: 2 launches instead of 3).
    // The GaussianBeamlet kernels need one in-place chunk more than they fuse levels (step_kernel_gauss): one level fewer.
    constexpr bool GAUSS = KIND == BMO_BEAM_GAUSSIAN;
    int fuse_max = GAUSS ? MAX_FUSE - 1 : MAX_FUSE;
    if (const char* e = getenv(GAUSS ? "BMO_FUSE_GAUSS" : "BMO_FUSE")) fuse_max = std::max(1, std::min(fuse_max, atoi(e)));
    DevBuf gstage;
    int64_t gstage_cap = 0;
    DevBuf gkeep;     // StepParams::gkeep
    DevBuf pend_buf;  // StepParams::pend
    int64_t pend_cap = 0;
    static const bool keep_kids_on = !(getenv("BMO_KEEP_KIDS") && atoi(getenv("BMO_KEEP_KIDS")) == 0);
    double keep_ratio = 1.0;  // share of the previous launch's beams that went on (the fuse-count rule of round 2: launches above BMO_INWAVE_MAX only)
    while (cur.count > 0) {
        const int64_t m = cur.count;
        static const double keep_hi = getenv("BMO_KEEP_HI") ? atof(getenv("BMO_KEEP_HI")) : 0.9, keep_lo = getenv("BMO_KEEP_LO") ? atof(getenv("BMO_KEEP_LO")) : 0.6;
        static const int fuse_mid = getenv("BMO_FUSE_MID") ? atoi(getenv("BMO_FUSE_MID")) : 2, fuse_lo = getenv("BMO_FUSE_LO") ? atoi(getenv("BMO_FUSE_LO")) : 1;
        // records per wave (StepParams::lane_shift).  BMO_THIN_WAVES=<n>: a launch with fewer than n waves spreads its records over more waves.
        // Off by default: it took 6 % off the vignetted bundle while launches were level-synchronous, nothing since the beam splitters are
        // handled in the loop, and it costs small batches dearly (100 rays through config 2: 0.84 ms against 0.55 — every wave on a CU of
        // its own pays that CU's instruction- and scalar-cache warm-up): profiles/r03_ab_inwave.txt.
        int lane_shift = 6;
        {
            static const int64_t thin_waves = getenv("BMO_THIN_WAVES") ? atoll(getenv("BMO_THIN_WAVES")) : 0;
            while (lane_shift > 0 && ((m + (1ll << (lane_shift - 1)) - 1) >> (lane_shift - 1)) <= thin_waves) lane_shift -= 1;
        }
        // Beam kernels: every launch fuses all its levels, whatever share of its beams ends, and handles its beam splitters in the loop
        // (step_kernel<.., INW>): holes cost lane time, launches cost the wait for the slowest march of every generation, and the second
        // is the dearer one — the vignetted bundle takes 2 launches and 14 ms this way, 12 launches and 23 ms with the keep-ratio rule of
        // round 2, which BMO_INWAVE_MAX (largest launch treated like this, in records) brings back for launches above it.
        static const int64_t inwave_max = getenv("BMO_INWAVE_MAX") ? atoll(getenv("BMO_INWAVE_MAX")) : INT64_MAX;
        const bool tail = m <= inwave_max;
        int n_fuse = tail ? fuse_max : keep_ratio >= keep_hi ? fuse_max : (keep_ratio >= keep_lo ? std::min(fuse_mid, fuse_max) : std::min(fuse_lo, fuse_max));
        // the in-place levels of a launch are allocated up front: at most 24 GB of them (2^24 beams: 8 levels)
        if (keep_log) n_fuse = (int)std::min<int64_t>(n_fuse, 1 + (int64_t)(((size_t)24 << 30) / ((size_t)std::max<int64_t>(m, 1) * rec_bytes)));
        const int64_t n_waves = (m + (1ll << lane_shift) - 1) >> lane_shift;
        const unsigned n_blocks = (unsigned)((n_waves + BMO_BLOCK / 64 - 1) / (BMO_BLOCK / 64));
        Chunk nxt, inner[MAX_FUSE - 1];
        // the next launch's chunk first, then the in-place levels: the levels no wave reaches go back to the arena after the launch
        // (in-loop beam splitters of the Beam kernels: room for up to m reflected children more, StepParams::inwave_cap)
        const int64_t inwave_cap = (has_split && (kern_inw || GAUSS) && n_fuse > 1 && tail) ? m : 0;
        // (kept reflected children, StepParams::pend: a lane can end its wave's loop with a kept child AND two fresh ones — one record more per lane)
        const bool keep_kids = inwave_cap > 0 && keep_kids_on;
        if ((rc = new_chunk((has_split ? 2 * m : m) + inwave_cap + (keep_kids ? m : 0), nxt))) return rc;
        if (keep_kids && !GAUSS && m > pend_cap) {  // one buffer for the whole solve (the previous launch has completed: its block goes back at once)
            PoolHold::Now at_once;
            pend_buf.release();
            pend_cap = (m + 1) & ~(int64_t)1;
            if ((rc = pend_buf.alloc((size_t)pend_cap * rec_bytes))) return rc;
        }
        uint8_t* wl = nullptr;
        if (keep_log && n_fuse > 1) {
            auto b = std::make_unique<DevBuf>();
            // one byte per wave of the GRID, not of the batch: the tail waves of the last workgroup (no record, j >= m) note their level too
            if ((rc = b->alloc((size_t)n_blocks * (BMO_BLOCK / 64)))) return rc;
            wl = static_cast<uint8_t*>(b->p);
            R->wave_last.push_back(std::move(b));
        }
        size_t top_after[MAX_FUSE], blocks_after[MAX_FUSE];  // arena state behind nxt [0] and behind every in-place level [q + 1]
        top_after[0] = top;
        blocks_after[0] = R->arena.size();
        for (int q = 0; q < (GAUSS ? n_fuse : n_fuse - 1); ++q) {
            if (!keep_log && q > (GAUSS ? 1 : 0)) {
                // nobody reads the log: a lane's in-place record is dead once it has been read back, one chunk serves all levels (a beamlet's
                // step writes its next rays while the hits of the level it reads are still being filled in: two chunks, taken in turns)
                inner[q] = inner[GAUSS ? (q & 1) : 0];
                continue;
            }
            if ((rc = new_chunk(m, inner[q]))) {
                if (rc != BMO_ERR_OOM || q == 0) return rc;
                n_fuse = GAUSS ? q : q + 1;  // no room for this many in-place levels: fuse the ones that fit (the launch before fused fewer, too)
                rc = BMO_OK;
                break;
            }
            inner[q].count = m;  // same slot numbering as cur; records of beams that ended earlier are marked node = -1
            inner[q].wl = wl;
            inner[q].level = q + 1;
            inner[q].wl_shift = lane_shift;
            top_after[q + 1] = top;
            blocks_after[q + 1] = R->arena.size();
        }
        if (has_split && (rc = grow_nodes(n_nodes + 2 * m + 2 * inwave_cap))) return rc;
        // (Gaussian) staging planes for the reflected children's rays of this launch: one buffer for the whole solve, grown when a launch has more records
        // than any before it (the previous launch has completed by then: its block goes back to the pool at once)
        if (KIND == BMO_BEAM_GAUSSIAN && m > gstage_cap) {
            PoolHold::Now at_once;
            gstage.release();
            gkeep.release();
            gstage_cap = ((m + m / 8 + 1) & ~(int64_t)1);
            if ((rc = gstage.alloc((size_t)gstage_cap * 21 * 8)) || (rc = gkeep.alloc((size_t)gstage_cap * 24 * 8))) return rc;
        }
        StepParams P;
        P.gstage = (double*)gstage.p;
        P.gstage_cap = gstage_cap;
        P.hdr = scene->hdr;
        P.blob = dblob;
        P.blob_bytes = blob_bytes;
        P.use_lds = use_lds;
        P.cur = cur;
        P.nxt = nxt;
        for (int q = 0; q < MAX_FUSE - 1; ++q) P.inner[q] = q < (GAUSS ? n_fuse : n_fuse - 1) ? inner[q] : Chunk{nullptr, nullptr, 0, 0};
        P.n_fuse = n_fuse;
        P.ctr = d_ctr;
        P.call_shards = static_cast<unsigned long long*>(shard_buf.p);
        P.nodes = node_arrays();
        P.r_max = opts->r_max;
        P.parity = steps & 1;
        P.old = old_tab;
        P.wave_last = wl;
        P.lane_shift = lane_shift;
        P.inwave_cap = inwave_cap;
        P.nodes0 = n_nodes;
        P.pend = Chunk{nullptr, nullptr, 0, 0};
        P.gkeep = (GAUSS && keep_kids) ? (double*)gkeep.p : nullptr;
        if (keep_kids && !GAUSS) {
            P.pend.d = static_cast<double*>(pend_buf.p);
            P.pend.i = reinterpret_cast<int32_t*>(static_cast<char*>(pend_buf.p) + (size_t)pend_cap * (size_t)L::ND * 8);
            P.pend.cap = pend_cap;
            P.pend.count = m;
        }
        P.tile_order = nullptr;
        P.tile_cost = nullptr;
        {
            static const bool lpt_on = !(getenv("BMO_LPT") && atoi(getenv("BMO_LPT")) == 0);
            if (lpt_on && steps == 0 && lane_shift == 6 && n_blocks >= 64) {
                if (batch->tile_n != (int64_t)n_blocks) {
                    PoolHold::Now at_once;
                    batch->cost_valid = false;
                    batch->tile_n = 0;
                    batch->tile_cost.release(), batch->tile_order.release(), batch->lpt_ids.release(), batch->lpt_keys.release(), batch->lpt_tmp.release();
                    size_t tb = 0;
                    HIP_TRY(hipcub::DeviceRadixSort::SortPairsDescending(nullptr, tb, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const int32_t*)nullptr, (int32_t*)nullptr,
                                                                         (int)n_blocks, 0, 32, stream));
                    if ((rc = batch->tile_cost.alloc((size_t)n_blocks * 4)) || (rc = batch->tile_order.alloc((size_t)n_blocks * 4)) ||
                        (rc = batch->lpt_ids.alloc((size_t)n_blocks * 4)) || (rc = batch->lpt_keys.alloc((size_t)n_blocks * 4)) || (rc = batch->lpt_tmp.alloc(std::max<size_t>(tb, 16))))
                        return rc;
                    hipLaunchKernelGGL(iota_kernel, dim3((n_blocks + 255) / 256), dim3(256), 0, stream, (int32_t*)batch->lpt_ids.p, (int64_t)n_blocks);
                    batch->tile_n = (int64_t)n_blocks;
                }
                if (!batch->cost_valid && prev && prev->tile_n == (int64_t)n_blocks && prev->tile_cost.p) {  // feedback from the solution that is retraced
                    HIP_TRY(hipMemcpyAsync(batch->tile_cost.p, prev->tile_cost.p, (size_t)n_blocks * 4, hipMemcpyDeviceToDevice, stream));
                    batch->cost_valid = true;
                }
                if (batch->cost_valid) {
                    size_t tb = batch->lpt_tmp.bytes;
                    HIP_TRY(hipcub::DeviceRadixSort::SortPairsDescending(batch->lpt_tmp.p, tb, (const uint32_t*)batch->tile_cost.p, (uint32_t*)batch->lpt_keys.p,
                                                                         (const int32_t*)batch->lpt_ids.p, (int32_t*)batch->tile_order.p, (int)n_blocks, 0, 32, stream));
                    P.tile_order = (const int32_t*)batch->tile_order.p;
                }
                P.tile_cost = (uint32_t*)batch->tile_cost.p;
                batch->cost_valid = true;  // (after this launch)
            }
        }
        {
            // without feedback (StepParams::tile_order) a launch starts from the END of its chunk: a disc source has its marginal rays there,
            // children and survivors are queued in the order their parents got there — the slow ones last (StepParams::reverse)
            static const int reverse_order = getenv("BMO_REVERSE") ? atoi(getenv("BMO_REVERSE")) : -1;
            P.reverse = reverse_order >= 0 ? reverse_order : 1;
        }
#if defined(BMO_DEV_TIMELINE)  // developer builds, BMO_TIMELINE=1: how many waves are at work over the course of every launch
        DevBuf tl_buf;
        P.tl = nullptr;
        if (getenv("BMO_TIMELINE")) {
            if ((rc = tl_buf.alloc((size_t)n_waves * 16 + 64))) return rc;
            HIP_TRY(hipMemsetAsync(tl_buf.p, 0, tl_buf.bytes, stream));
            P.tl = (unsigned long long*)tl_buf.p;
        }
#endif
        DBG("step %d launching m=%lld, %d records per wave", steps, (long long)m, 1 << lane_shift);
        // launch timing: one event pair per step out of a cached pool, read after the loop (nothing but the counter read-back
        // sits between two launches)
        while ((int)ctxp->step_ev.size() < 2 * (steps + 1)) {
            hipEvent_t e;
            HIP_TRY(hipEventCreate(&e));
            ctxp->step_ev.push_back(e);
        }
        HIP_TRY(hipEventRecord(ctxp->step_ev[2 * steps], stream));
        // (step_kernel_gauss handles P.inwave_cap itself: the GaussianBeamlet branch above assigns no kern_inw)
        // (the 4-waves-per-SIMD build for a launch that fills the device at that occupancy — 256 CUs x 16 waves —, the 3-waves one below: step_waves.
        //  Between 4 096 and ~10 000 waves the two scenes measured disagree — config 5 - 13 % / - 6 %, config 2 + 2 % / + 3 % — and the larger effect decides.)
        const char* const wide_env = getenv("BMO_WIDE_MIN_WAVES");  // (read per launch: the tests switch it between solves)
        const int64_t wide_min_waves = wide_env ? atoll(wide_env) : 4096;
        const bool wide = kern_wide && n_waves >= wide_min_waves;
        void (*const launch_kern)(StepParams) = (inwave_cap > 0 && kern_inw) ? (wide ? kern_inw_wide : kern_inw) : (wide ? kern_wide : kern);
        if (!launch_kern) return fail(BMO_ERR_INTERNAL, "no step kernel selected for this beam kind / scene level");
        hipLaunchKernelGGL(launch_kern, dim3(n_blocks), dim3(BMO_BLOCK), lds_bytes, stream, P);
        HIP_TRY(hipEventRecord(ctxp->step_ev[2 * steps + 1], stream));
        HIP_TRY(hipMemcpyAsync(h_ctr_p, d_ctr, sizeof h_ctr, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        HIP_TRY(hipGetLastError());
#if defined(BMO_DEV_TIMELINE)
        if (P.tl) {
            const size_t nw = (size_t)n_waves;
            std::vector<unsigned long long> t(2 * nw + 3);
            HIP_TRY(hipMemcpy(t.data(), P.tl, nw * 16 + 24, hipMemcpyDeviceToHost));
            {
                const double a = (double)t[2 * nw], b2 = (double)t[2 * nw + 1], c2 = (double)t[2 * nw + 2], sum = a + b2 + c2;
                // (finer timers inside the lane code — per sdf_any / normal_any / object — measure mostly their own s_memrealtime latency)
                fprintf(stderr, "[bmo] timeline step %d: wave time before / in / after tracing_step: %.1f %% / %.1f %% / %.1f %%  (%.1f us per wave in all)\n", steps,
                        100 * a / sum, 100 * b2 / sum, 100 * c2 / sum, sum / nw / 100.0);
            }
            unsigned long long t0 = ~0ull, t1 = 0;
            for (size_t w = 0; w < nw; ++w) {
                t0 = std::min(t0, t[2 * w]);
                t1 = std::max(t1, t[2 * w + 1]);
            }
            const double span = (double)(t1 - t0);
            const int NB = 40;
            std::vector<double> busy(NB, 0.0);  // wave residency integrated per bin
            for (size_t w = 0; w < nw; ++w) {
                const double a = (double)(t[2 * w] - t0) / span * NB, b2 = (double)(t[2 * w + 1] - t0) / span * NB;
                for (int q = (int)a; q < NB && q <= (int)b2; ++q) busy[q] += std::min(b2, (double)q + 1) - std::max(a, (double)q);
            }
            fprintf(stderr, "[bmo] timeline step %d: %zu waves, span %.3f ms; mean waves at work per 1/%d of the span:\n ", steps, nw, span / 1e5, NB);
            for (int q = 0; q < NB; ++q) fprintf(stderr, " %.0f", busy[q]);
            fprintf(stderr, "\n");
            {  // the waves that end last: where in the grid they sit, when they started
                std::vector<size_t> idx(nw);
                for (size_t w = 0; w < nw; ++w) idx[w] = w;
                const size_t top = std::min<size_t>(nw, 12);
                std::partial_sort(idx.begin(), idx.begin() + (long)top, idx.end(), [&](size_t a, size_t b2) { return t[2 * a + 1] > t[2 * b2 + 1]; });
                fprintf(stderr, "[bmo] timeline step %d: last waves to end (wave of the grid: start .. end in ms):", steps);
                for (size_t q = 0; q < top; ++q) fprintf(stderr, "  %zu: %.3f .. %.3f", idx[q], (double)(t[2 * idx[q]] - t0) / 1e5, (double)(t[2 * idx[q] + 1] - t0) / 1e5);
                fprintf(stderr, "\n");
                std::vector<double> dur(nw);
                for (size_t w = 0; w < nw; ++w) dur[w] = (double)(t[2 * w + 1] - t[2 * w]) / 1e5;
                std::vector<double> srt = dur;
                std::sort(srt.begin(), srt.end());
                fprintf(stderr, "[bmo] timeline step %d: wave durations ms p50 %.3f p90 %.3f p99 %.3f p99.9 %.3f max %.3f\n", steps, srt[nw / 2], srt[nw * 9 / 10], srt[nw * 99 / 100],
                        srt[std::min(nw - 1, nw * 999 / 1000)], srt[nw - 1]);
            }
        }
#endif
        const unsigned long long produced = h_ctr.next_count[steps & 1];
        DBG("step %d done next=%llu nodes=%llu deepest in-place level %llu of %d", steps, produced, h_ctr.node_count, h_ctr.max_level[steps & 1], n_fuse);
        steps += 1;
        if (h_ctr.overflow) return fail(BMO_ERR_INTERNAL, "queue overflow (internal capacity bound violated)");
        if (keep_log) {
            const int used = (int)std::min<unsigned long long>(h_ctr.max_level[(steps - 1) & 1], (unsigned long long)(n_fuse - 1));  // in-place levels reached
            R->chunks.push_back(cur);
            for (int q = 0; q < used; ++q) R->chunks.push_back(inner[q]);
            // the levels behind them were never touched: their room goes back to the arena (allocation is a bump, so everything
            // allocated after the last level in use belongs to them)
            if (used < (GAUSS ? n_fuse : n_fuse - 1)) {  // (the GaussianBeamlet kernels' extra level is always given back)
                PoolHold::Now at_once;  // (the launch has completed: nothing in flight touches these blocks)
                while (R->arena.size() > blocks_after[used]) R->arena.pop_back();
                top = top_after[used];
            }
        } else {  // the launch above has completed (counter read-back): its input and in-place levels are dead
            drop_chunk(cur);
            if (GAUSS || n_fuse > 1) drop_chunk(inner[0]);
            if (GAUSS && n_fuse > 1) drop_chunk(inner[1]);
        }
        shrink_last(nxt, (int64_t)produced);
        keep_ratio = (double)std::min<unsigned long long>(produced, (unsigned long long)m) / (double)m;
        n_nodes = (int64_t)h_ctr.node_count;
        if (opts->max_beams > 0 && n_nodes > (int64_t)opts->max_beams)
            return fail(BMO_ERR_LIMIT, "beam tree exceeds bmo_trace_opts.max_beams (" + std::to_string(n_nodes) + " beams after " + std::to_string(steps) +
                                           " launches): a splitter facing a mirror?");
        cur = nxt;
    }
    for (int q = 0; q < steps; ++q) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, ctxp->step_ev[2 * q], ctxp->step_ev[2 * q + 1]));
        kernel_ms += ms;
        DBG("step %d kernel %.3f ms", q, ms);
        if (q + 1 < steps && getenv("BMO_GAPS")) {  // diagnosis: device idle time between two step launches (host round trip)
            float gap = 0;
            (void)hipEventElapsedTime(&gap, ctxp->step_ev[2 * q + 1], ctxp->step_ev[2 * q + 2]);
            fprintf(stderr, "[bmo] step %d kernel %.3f ms, gap to the next launch %.3f ms\n", q, ms, gap);
        }
    }
    lap("steps");
    R->n_nodes = n_nodes;
    R->n_steps = steps;
    R->kernel_ms = kernel_ms;
    if (batch->cost_valid && batch->tile_n > 0 && batch->tile_cost.p && !R->tile_cost.alloc((size_t)batch->tile_n * 4)) {  // (best effort)
        if (hipMemcpyAsync(R->tile_cost.p, batch->tile_cost.p, (size_t)batch->tile_n * 4, hipMemcpyDeviceToDevice, stream) == hipSuccess) R->tile_n = batch->tile_n;
    }
    // Everything below is queued behind ONE synchronisation at the end: temporaries stay alive until then (`keep`: a block that went
    // back to the pool could be handed to a view running on another stream), counts are read back last.
    std::vector<std::unique_ptr<DevBuf>> keep;
    auto temp = [&](size_t bytes) -> DevBuf* {
        auto b = std::make_unique<DevBuf>();
        if (b->alloc(bytes)) return nullptr;
        keep.push_back(std::move(b));
        return keep.back().get();
    };
    // pinned read-back area: [0..1023] call-counter shards (64 x 16 words), [1024] segment count, [1025..] hit offsets
    static_assert(sizeof(unsigned long long) == 8, "");
    HostBuf h_tail;
    if ((rc = h_tail.alloc((1024 + 1 + 1025) * 8))) return rc;
    unsigned long long* const h_sh = static_cast<unsigned long long*>(h_tail.p);
    long long* const h_sum = reinterpret_cast<long long*>(h_sh + 1024);
    int32_t* const h_off = reinterpret_cast<int32_t*>(h_sh + 1025);
    *h_sum = 0;
    HIP_TRY(hipMemcpyAsync(h_sh, shard_buf.p, 64 * 128, hipMemcpyDeviceToHost, stream));
    if (n_nodes > 0) {  // segments = sum of the beams' segment counts (chunk counts include the holes of fused levels)
        DevBuf* d_sum = temp(8);
        if (!d_sum) return BMO_ERR_OOM;
        size_t tmp_bytes = 0;
        HIP_TRY(hipcub::DeviceReduce::Sum(nullptr, tmp_bytes, (const int32_t*)R->n_nseg.p, (long long*)d_sum->p, (int)n_nodes, stream));
        DevBuf* tmp = temp(tmp_bytes);
        if (!tmp) return BMO_ERR_OOM;
        HIP_TRY(hipcub::DeviceReduce::Sum(tmp->p, tmp_bytes, (const int32_t*)R->n_nseg.p, (long long*)d_sum->p, (int)n_nodes, stream));
        HIP_TRY(hipMemcpyAsync(h_sum, d_sum->p, 8, hipMemcpyDeviceToHost, stream));
    }

    // ---- canonical node order (bundle order x BFS order): sort by (root, depth, path)
    if ((rc = R->order.alloc((size_t)std::max<int64_t>(n_nodes, 1) * 4))) return rc;
    const unsigned nb = (unsigned)((n_nodes + 255) / 256);
    // test hooks: BMO_FORCE_SORT_ORDER / BMO_FORCE_DEEP_ORDER take the radix-sort / level-by-level path for any tree
    const bool heap_order = n_nodes > n && h_ctr.max_depth <= (unsigned long long)MAX_HEAP_LEVELS && !std::getenv("BMO_FORCE_SORT_ORDER") &&
                            !std::getenv("BMO_FORCE_DEEP_ORDER");
    if (heap_order) {
        DevBuf *bits = temp((size_t)n * 8), *cnt = temp((size_t)n * 4), *base = temp((size_t)n * 4);
        if (!bits || !cnt || !base) return BMO_ERR_OOM;
        const unsigned rb = (unsigned)((n + 255) / 256);
        if (tbits_buf.p) {  // collected while the nodes were made
            bits = &tbits_buf;
        } else {
            HIP_TRY(hipMemsetAsync(bits->p, 0, (size_t)n * 8, stream));
            hipLaunchKernelGGL(tree_bits_kernel, dim3(nb), dim3(256), 0, stream, (const int32_t*)R->n_root.p, (const unsigned long long*)R->n_key.p, n_nodes,
                               (unsigned long long*)bits->p);
        }
        hipLaunchKernelGGL(tree_count_kernel, dim3(rb), dim3(256), 0, stream, (const unsigned long long*)bits->p, n, (int32_t*)cnt->p);
        size_t tmp_bytes = 0;
        HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, (const int32_t*)cnt->p, (int32_t*)base->p, (int)n, stream));
        DevBuf* st = temp(tmp_bytes);
        if (!st) return BMO_ERR_OOM;
        HIP_TRY(hipcub::DeviceScan::ExclusiveSum(st->p, tmp_bytes, (const int32_t*)cnt->p, (int32_t*)base->p, (int)n, stream));
        hipLaunchKernelGGL(tree_rank_kernel, dim3(nb), dim3(256), 0, stream, (const int32_t*)R->n_root.p, (const unsigned long long*)R->n_key.p, n_nodes,
                           (const unsigned long long*)bits->p, (const int32_t*)base->p, (int32_t*)R->order.p);
    } else if (n_nodes > 0) {
        hipLaunchKernelGGL(iota_kernel, dim3(nb), dim3(256), 0, stream, (int32_t*)R->order.p, n_nodes);
    }
    if (n_nodes > n && !heap_order) {
        auto bits_for = [](unsigned long long v) {  // bits needed to hold values 0..v
            int b = 0;
            while (b < 64 && (v >> b)) ++b;
            return b;
        };
        const int bits_depth = bits_for(h_ctr.max_depth), bits_root = bits_for((unsigned long long)std::max<int64_t>(n - 1, 0));
        DevBuf keys_in, keys_out, vals_out, tmp;
        const bool force_deep = std::getenv("BMO_FORCE_DEEP_ORDER") != nullptr;
        if (h_ctr.max_depth <= (unsigned long long)MAX_PATH_LEVELS && !force_deep) {
            const int bits_path = (int)h_ctr.max_depth;  // one bit per level
            const int bits = bits_root + bits_depth + bits_path;
            const bool narrow = bits <= 32;
            DevBuf *k_in = temp((size_t)n_nodes * (narrow ? 4 : 8)), *k_out = temp((size_t)n_nodes * (narrow ? 4 : 8)), *v_out = temp((size_t)n_nodes * 4);
            if (!k_in || !k_out || !v_out) return BMO_ERR_OOM;
            hipLaunchKernelGGL(pack_keys_kernel, dim3(nb), dim3(256), 0, stream, (const int32_t*)R->n_root.p, (const unsigned long long*)R->n_key.p, n_nodes, bits_depth,
                               bits_path, narrow ? (uint32_t*)k_in->p : nullptr, narrow ? nullptr : (unsigned long long*)k_in->p);
            size_t tmp_bytes = 0;
            if (narrow) {
                HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, (const uint32_t*)k_in->p, (uint32_t*)k_out->p, (const int32_t*)R->order.p,
                                                           (int32_t*)v_out->p, (int)n_nodes, 0, bits, stream));
                DevBuf* st = temp(tmp_bytes);
                if (!st) return BMO_ERR_OOM;
                HIP_TRY(hipcub::DeviceRadixSort::SortPairs(st->p, tmp_bytes, (const uint32_t*)k_in->p, (uint32_t*)k_out->p, (const int32_t*)R->order.p,
                                                           (int32_t*)v_out->p, (int)n_nodes, 0, bits, stream));
            } else {
                HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, (const unsigned long long*)k_in->p, (unsigned long long*)k_out->p,
                                                           (const int32_t*)R->order.p, (int32_t*)v_out->p, (int)n_nodes, 0, bits, stream));
                DevBuf* st = temp(tmp_bytes);
                if (!st) return BMO_ERR_OOM;
                HIP_TRY(hipcub::DeviceRadixSort::SortPairs(st->p, tmp_bytes, (const unsigned long long*)k_in->p, (unsigned long long*)k_out->p,
                                                           (const int32_t*)R->order.p, (int32_t*)v_out->p, (int)n_nodes, 0, bits, stream));
            }
            std::swap(R->order.p, v_out->p);  // the sorted ids are the order; the iota goes to `keep`
            std::swap(R->order.bytes, v_out->bytes);
        } else {
            // deep trees: ranks within each tree level, level by level, then sort by (root, depth, rank)
            const int64_t md = (int64_t)h_ctr.max_depth;
            const int32_t* parent = (const int32_t*)R->n_parent.p;
            const unsigned long long* key = (const unsigned long long*)R->n_key.p;
            DevBuf depth, depth_sorted, by_depth, starts, rank, flag, scan, k64, k64_out, ids2;
            if ((rc = depth.alloc((size_t)n_nodes * 4)) || (rc = depth_sorted.alloc((size_t)n_nodes * 4)) || (rc = by_depth.alloc((size_t)n_nodes * 4)) ||
                (rc = starts.alloc((size_t)(md + 2) * 4)) || (rc = rank.alloc((size_t)n_nodes * 4)) || (rc = flag.alloc((size_t)n_nodes * 4)) ||
                (rc = scan.alloc((size_t)n_nodes * 4)) || (rc = k64.alloc((size_t)n_nodes * 8)) || (rc = k64_out.alloc((size_t)n_nodes * 8)) ||
                (rc = ids2.alloc((size_t)n_nodes * 4)) || (rc = vals_out.alloc((size_t)n_nodes * 4)) || (rc = keys_in.alloc((size_t)n_nodes * 4)) ||
                (rc = keys_out.alloc((size_t)n_nodes * 4)))
                return rc;
            hipLaunchKernelGGL(node_depth_kernel, dim3(nb), dim3(256), 0, stream, key, n_nodes, (uint32_t*)depth.p);
            size_t tb = 0, tb2 = 0, tb3 = 0, tb4 = 0;
            HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, (const uint32_t*)depth.p, (uint32_t*)depth_sorted.p, (const int32_t*)R->order.p,
                                                       (int32_t*)by_depth.p, (int)n_nodes, 0, bits_depth, stream));
            HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tb2, (const int32_t*)flag.p, (int32_t*)scan.p, (int)n_nodes, stream));
            HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tb3, (const uint32_t*)keys_in.p, (uint32_t*)keys_out.p, (const int32_t*)R->order.p,
                                                       (int32_t*)ids2.p, (int)n_nodes, 0, 32, stream));
            HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tb4, (const unsigned long long*)k64.p, (unsigned long long*)k64_out.p, (const int32_t*)ids2.p,
                                                       (int32_t*)vals_out.p, (int)n_nodes, 0, 64, stream));
            if ((rc = tmp.alloc(std::max(std::max(tb, tb2), std::max(tb3, tb4))))) return rc;
            HIP_TRY(hipcub::DeviceRadixSort::SortPairs(tmp.p, tb, (const uint32_t*)depth.p, (uint32_t*)depth_sorted.p, (const int32_t*)R->order.p,
                                                       (int32_t*)by_depth.p, (int)n_nodes, 0, bits_depth, stream));
            hipLaunchKernelGGL(level_starts_kernel, dim3(nb), dim3(256), 0, stream, (const uint32_t*)depth_sorted.p, n_nodes, md, (int32_t*)starts.p);
            std::vector<int32_t> h_start((size_t)md + 2);
            HIP_TRY(hipMemcpyAsync(h_start.data(), starts.p, (size_t)(md + 2) * 4, hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipStreamSynchronize(stream));
            const int32_t* ids = (const int32_t*)by_depth.p;
            hipLaunchKernelGGL(level_root_rank_kernel, dim3((unsigned)((h_start[1] + 255) / 256)), dim3(256), 0, stream, ids, h_start[1], (int32_t*)rank.p);
            int32_t max_level = h_start[1];
            for (int64_t d = 1; d <= md; ++d) {
                const int32_t cnt_prev = h_start[d] - h_start[d - 1], cnt = h_start[d + 1] - h_start[d];
                if (cnt <= 0 || cnt_prev <= 0) continue;
                max_level = std::max(max_level, cnt);
                const unsigned lb = (unsigned)((cnt + 255) / 256);
                HIP_TRY(hipMemsetAsync(flag.p, 0, (size_t)cnt_prev * 4, stream));
                hipLaunchKernelGGL(level_flag_kernel, dim3(lb), dim3(256), 0, stream, ids + h_start[d], cnt, parent, key, (const int32_t*)rank.p, (int32_t*)flag.p);
                size_t t = tmp.bytes;
                HIP_TRY(hipcub::DeviceScan::ExclusiveSum(tmp.p, t, (const int32_t*)flag.p, (int32_t*)scan.p, (int)cnt_prev, stream));
                hipLaunchKernelGGL(level_rank_kernel, dim3(lb), dim3(256), 0, stream, ids + h_start[d], cnt, parent, key, (const int32_t*)scan.p, (int32_t*)rank.p);
            }
            // LSD: by rank within the level, then (stable) by (root, depth)
            hipLaunchKernelGGL(gather_u32_kernel, dim3(nb), dim3(256), 0, stream, (const int32_t*)R->order.p, (const int32_t*)rank.p, n_nodes, (uint32_t*)keys_in.p);
            size_t t3 = tmp.bytes;
            HIP_TRY(hipcub::DeviceRadixSort::SortPairs(tmp.p, t3, (const uint32_t*)keys_in.p, (uint32_t*)keys_out.p, (const int32_t*)R->order.p, (int32_t*)ids2.p,
                                                       (int)n_nodes, 0, bits_for((unsigned long long)max_level), stream));
            hipLaunchKernelGGL(root_depth_key_kernel, dim3(nb), dim3(256), 0, stream, (const int32_t*)ids2.p, (const int32_t*)R->n_root.p, key, n_nodes, bits_depth,
                               (unsigned long long*)k64.p);
            size_t t4 = tmp.bytes;
            HIP_TRY(hipcub::DeviceRadixSort::SortPairs(tmp.p, t4, (const unsigned long long*)k64.p, (unsigned long long*)k64_out.p, (const int32_t*)ids2.p,
                                                       (int32_t*)vals_out.p, (int)n_nodes, 0, bits_root + bits_depth, stream));
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipStreamSynchronize(stream));  // the level tables go back to the pool
            std::swap(R->order.p, vals_out.p);
            std::swap(R->order.bytes, vals_out.bytes);
        }
    }
    lap("order");
    // ---- detector hits in reference push! order: flags -> exclusive scan -> gather
    const int nd = R->n_detectors;
    R->det_count.assign(nd, 0);
    R->det_offset.assign(nd, 0);
    const bool with_hits = nd > 0 && n_nodes > 0;
    if (with_hits) {
        const int64_t tot = (int64_t)nd * n_nodes;
        if (tot >= ((int64_t)1 << 31)) return fail(BMO_ERR_UNSUPPORTED, "detectors x beams exceeds 2^31: split the batch");
        if (nd + 1 > 1024) return fail(BMO_ERR_UNSUPPORTED, "more than 1023 detectors");
        DevBuf *flags = temp((size_t)tot * 4), *offs = temp((size_t)tot * 4), *d_off = temp((size_t)(nd + 1) * 4);
        if (!flags || !offs || !d_off) return BMO_ERR_OOM;
        hipLaunchKernelGGL(hit_flags_kernel, dim3(nb), dim3(256), 0, stream, (const int32_t*)R->order.p, (const int32_t*)R->n_hitdet.p, n_nodes, nd,
                           nsub, (int32_t*)flags->p);
        size_t tmp_bytes = 0;
        HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, (const int32_t*)flags->p, (int32_t*)offs->p, (int)tot, stream));
        DevBuf* st = temp(tmp_bytes);
        if (!st) return BMO_ERR_OOM;
        HIP_TRY(hipcub::DeviceScan::ExclusiveSum(st->p, tmp_bytes, (const int32_t*)flags->p, (int32_t*)offs->p, (int)tot, stream));
        // per-detector offsets = scan value at the start of each detector's segment; total = last offs + last flag
        hipLaunchKernelGGL(hit_offsets_kernel, dim3(1), dim3(1024), 0, stream, (const int32_t*)offs->p, (const int32_t*)flags->p, n_nodes, nd, (int32_t*)d_off->p);
        HIP_TRY(hipMemcpyAsync(h_off, d_off->p, (size_t)(nd + 1) * 4, hipMemcpyDeviceToHost, stream));
        // the hit table is sized before the counts are back on the host: a beam with a hit has ended and every splitting beam has two
        // children, so there are at most (beams + roots) / 2 of them
        const int64_t bound = (n_nodes + n) / 2 * nsub + nsub;
        if ((rc = R->det_data.alloc((size_t)std::max<int64_t>(bound, 1) * 72)) || (rc = R->det_node.alloc((size_t)std::max<int64_t>(bound, 1) * 4))) return rc;
        hipLaunchKernelGGL(hit_gather_kernel, dim3(nb), dim3(256), 0, stream, (const int32_t*)R->order.p, (const int32_t*)R->n_hitdet.p,
                           (const double*)R->n_hit.p, n_nodes, nsub, (const int32_t*)offs->p, bound, (double*)R->det_data.p, (int32_t*)R->det_node.p,
                           &d_ctr->overflow);
    }
    HIP_TRY(hipMemcpyAsync(h_ctr_p, d_ctr, sizeof h_ctr, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipEventRecord(ev_t1, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    HIP_TRY(hipGetLastError());
    lap("hits");
    if (h_ctr.overflow) return fail(BMO_ERR_INTERNAL, "hit table overflow (internal capacity bound violated)");
    {
        unsigned long long tot = 0;
        for (int q = 0; q < 64; ++q) tot += h_sh[(size_t)q * 16];
        R->calls = tot;
        R->n_records = (int64_t)*h_sum;
    }
    if (with_hits)
        for (int d = 0; d < nd; ++d) {
            R->det_offset[d] = h_off[d];
            R->det_count[d] = h_off[d + 1] - h_off[d];
        }
    float tms = 0;
    HIP_TRY(hipEventElapsedTime(&tms, ev_t0, ev_t1));
    R->total_ms = tms;
    return BMO_OK;
}

// bmo_selftest: the branch-free forms of bmo_lane.hpp against the rules they stand for, bitwise
__global__ void selftest_minmax_kernel(const double* __restrict__ v, int n, int32_t* __restrict__ bad) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n * n) return;
    const double x = v[i / n], y = v[i % n];
    auto same = [](double a, double b) { return (isnan_(a) && isnan_(b)) || __double_as_longlong(a) == __double_as_longlong(b); };
    int form = 0;
    if (!same(jmax(x, y), jmax_rule(x, y))) form = 1;
    else if (!same(jmin(x, y), jmin_rule(x, y))) form = 2;
    else {
        // Dual forms: the winner's value and partials
        const Dual X{x, {1.0, 2.0, 3.0}}, Y{y, {5.0, 6.0, 7.0}};
        const Dual mx = jmax(X, Y), mn = jmin(X, Y), mx0 = jmax(X, y), mn0 = jmin(X, y);
        // (selection, ties and unordered operands to the second argument: bmo_lane.hpp above jsqrt(DualN))
        const bool xwx = x > y, xwn = x < y;
        if (!same(mx.v, xwx ? x : y) || mx.p[0] != (xwx ? 1.0 : 5.0) || mx.p[2] != (xwx ? 3.0 : 7.0)) form = 3;
        else if (!same(mn.v, xwn ? x : y) || mn.p[0] != (xwn ? 1.0 : 5.0) || mn.p[2] != (xwn ? 3.0 : 7.0)) form = 4;
        else if (!same(mx0.v, xwx ? x : y) || mx0.p[1] != (xwx ? 2.0 : 0.0)) form = 5;
        else if (!same(mn0.v, xwn ? x : y) || mn0.p[1] != (xwn ? 2.0 : 0.0)) form = 6;
        const Dual ab = jabs(X);
        if (!form && (!same(ab.v, fabs(x)) || ab.p[0] != (sgn(x) ? -1.0 : 1.0))) form = 7;
    }
    if (form && atomicCAS(bad, -1, i) == -1) bad[1] = form;
}

// Julia Base's elementary functions (bmo_jlmath.hpp) on the device, six results per argument: sin, cos, tan, acos(clamped), atan, atan(1, x)
__host__ __device__ inline void jl_six(double x, double* o) {
    o[0] = jl::sin(x);
    o[1] = jl::cos(x);
    o[2] = jl::tan(x);
    o[3] = jl::acos(x < -1.0 ? -1.0 : (x > 1.0 ? 1.0 : x));
    o[4] = jl::atan(x);
    o[5] = jl::atan2(1.0, x);
}
__global__ void selftest_jl_kernel(const double* __restrict__ x, int n, double* __restrict__ out) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i < n) jl_six(x[i], out + (size_t)6 * i);
}

template <class T>
int dl(std::vector<T>& h, const void* d, size_t count) {
    h.resize(count);
    if (count == 0) return BMO_OK;
    if (hipMemcpy(h.data(), d, count * sizeof(T), hipMemcpyDeviceToHost) != hipSuccess) return fail(BMO_ERR_NO_DEVICE, "hipMemcpy D2H");
    return BMO_OK;
}

}  // namespace

// =================================================================== C ABI
extern "C" {

int bmo_version(void) { return BMO_ABI_VERSION; }
#if !defined(BMO_SOURCE_HASH)
#define BMO_SOURCE_HASH ""
#endif
#if !defined(BMO_FLAGS_HASH)
#define BMO_FLAGS_HASH ""
#endif
// (the markers in front let the build script read the hashes out of the file without loading the library; the second one is the hash of
//  the compiler flags, so that a change of flags rebuilds the library too)
const char* bmo_source_hash(void) {
    static const char tagged[] = "BMO_SOURCE_HASH=" BMO_SOURCE_HASH "\0BMO_FLAGS_HASH=" BMO_FLAGS_HASH;
    return tagged + 16;
}
const char* bmo_build_flags_hash(void) {
    static const char tagged[] = "BMO_BUILT_WITH=" BMO_FLAGS_HASH;
    return tagged + 15;
}
const char* bmo_last_error(void) { return g_err.c_str(); }

int bmo_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

double bmo_jl_trig(int32_t which, double x, double y) {
    switch (which) {
        case 0: return jl::sin(x);
        case 1: return jl::cos(x);
        case 2: return jl::tan(x);
        case 3: return jl::acos(x);
        case 4: return jl::atan(x);
        case 5: return jl::atan2(y, x);
    }
    return (double)NAN;
}

int bmo_selftest(int32_t device) {
    if (hipSetDevice(device) != hipSuccess) return fail(BMO_ERR_NO_DEVICE, "hipSetDevice");
    const double tiny = 4.9406564584124654e-324, big = 1.7976931348623157e308;
    std::vector<double> v{0.0, -0.0, tiny, -tiny, 2.2250738585072014e-308, -2.2250738585072014e-308, 1.0, -1.0, 1.0000000000000002, 0.5, -0.5, 3.0, -3.0,
                          big, -big, (double)INFINITY, -(double)INFINITY, (double)NAN, -(double)NAN, 1e-10, -1e-10, 1e-300, 123.456, -123.456};
    const int n = (int)v.size();
    DevBuf d_v, d_bad;
    int rc;
    if ((rc = d_v.alloc((size_t)n * 8)) || (rc = d_bad.alloc(16))) return rc;
    HIP_TRY(hipMemcpy(d_v.p, v.data(), (size_t)n * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(d_bad.p, 0xFF, 16));
    hipLaunchKernelGGL(selftest_minmax_kernel, dim3((unsigned)((n * n + 255) / 256)), dim3(256), 0, 0, (const double*)d_v.p, n, (int32_t*)d_bad.p);
    HIP_TRY(hipGetLastError());
    int32_t bad[4];
    HIP_TRY(hipMemcpy(bad, d_bad.p, 16, hipMemcpyDeviceToHost));
    if (bad[0] != -1)
        return fail(BMO_ERR_INTERNAL, "device min/max differs from the rule: form " + std::to_string(bad[1]) + " on (" + std::to_string(v[(size_t)(bad[0] / n)]) + ", " +
                                          std::to_string(v[(size_t)(bad[0] % n)]) + ")");
    // ... and the elementary functions (bmo_jlmath.hpp): the device's results against this library's host build of the same header, bit for bit
    {
        std::vector<double> xs;
        unsigned long long s = 0x243F6A8885A308D3ull;
        auto rnd = [&]() {
            s = s * 6364136223846793005ull + 1442695040888963407ull;
            return (double)(s >> 11) * 0x1p-53;
        };
        for (int q = 0; q < 4096; ++q) xs.push_back(-7.0 + 14.0 * rnd());            // the reductions by 0 .. 4 pi/2
        for (int q = 0; q < 2048; ++q) xs.push_back(-1.0 + 2.0 * rnd());             // acos's three ranges, the kernels without reduction
        for (int q = 0; q < 1024; ++q) xs.push_back((rnd() < 0.5 ? -1.0 : 1.0) * std::ldexp(1.0 + rnd(), (int)(rnd() * 80) - 40));  // atan's five ranges, tiny and large
        for (int q = 1; q <= 8; ++q)                                                 // next to multiples of pi/2: the three-constant reduction
            for (int e = -3; e <= 3; ++e) xs.push_back(std::nextafter(q * 1.5707963267948966, e < 0 ? -100.0 : 100.0) + e * 1e-9);
        for (double sp : {0.0, -0.0, 1.0, -1.0, 0.5, -0.5, 0.6744, 0.67434, 0.4375, 0.6875, 1.1875, 2.4375, 1e-9, 1e-300, 1e6, 1.5e6, (double)INFINITY, (double)NAN})
            xs.push_back(sp);
        const int m = (int)xs.size();
        DevBuf d_x, d_o;
        if ((rc = d_x.alloc((size_t)m * 8)) || (rc = d_o.alloc((size_t)m * 48))) return rc;
        HIP_TRY(hipMemcpy(d_x.p, xs.data(), (size_t)m * 8, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(selftest_jl_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, 0, (const double*)d_x.p, m, (double*)d_o.p);
        HIP_TRY(hipGetLastError());
        std::vector<double> got((size_t)m * 6);
        HIP_TRY(hipMemcpy(got.data(), d_o.p, (size_t)m * 48, hipMemcpyDeviceToHost));
        static const char* const names[6] = {"sin", "cos", "tan", "acos", "atan", "atan(1, x)"};
        for (int i = 0; i < m; ++i) {
            double want[6];
            jl_six(xs[(size_t)i], want);
            for (int f = 0; f < 6; ++f) {
                const double a = got[(size_t)6 * i + f], b = want[f];
                if (!((a != a && b != b) || std::memcmp(&a, &b, 8) == 0)) {
                    char buf[160];
                    std::snprintf(buf, sizeof buf, "device %s(%.17g) = %.17g, host build of the same code %.17g", names[f], xs[(size_t)i], a, b);
                    return fail(BMO_ERR_INTERNAL, buf);
                }
            }
        }
    }
    return BMO_OK;
}

int bmo_scene_create(const bmo_scene_desc* d, bmo_scene** out) {
    if (!d || !out) return fail(BMO_ERR_INVALID, "null argument");
    if (d->abi_version != BMO_ABI_VERSION) return fail(BMO_ERR_INVALID, "abi version mismatch");
    if (d->n_objects < 0 || d->n_shapes < 0 || d->n_lambda < 1) return fail(BMO_ERR_INVALID, "bad counts");
    for (int i = 0; i < d->n_shapes; ++i) {
        const bmo_shape& s = d->shapes[i];
        if (s.kind < 0 || s.kind >= BMO_SHAPE_KIND_COUNT) return fail(BMO_ERR_UNSUPPORTED, "unknown shape kind");
        if (s.kind == BMO_SHAPE_UNION || s.kind == BMO_SHAPE_MENISCUS) {
            if (s.child_begin < 0 || s.child_begin + s.child_count > d->n_children || s.child_count < 1)
                return fail(BMO_ERR_INVALID, "child range out of bounds");
            if (s.kind == BMO_SHAPE_MENISCUS && s.child_count != 3) return fail(BMO_ERR_INVALID, "meniscus needs 3 children");
            for (int c = 0; c < s.child_count; ++c) {
                int ch = d->children[s.child_begin + c];
                if (ch < 0 || ch >= d->n_shapes) return fail(BMO_ERR_INVALID, "child id out of bounds");
                int ck = d->shapes[ch].kind;
                if (ck == BMO_SHAPE_UNION || ck == BMO_SHAPE_MESH) return fail(BMO_ERR_INVALID, "nested union / mesh child");
                if (s.kind == BMO_SHAPE_MENISCUS && ck == BMO_SHAPE_MENISCUS) return fail(BMO_ERR_INVALID, "nested meniscus");
            }
        }
        if ((s.kind == BMO_SHAPE_ASPH_CONVEX || s.kind == BMO_SHAPE_ASPH_CONCAVE || s.kind == BMO_SHAPE_ACYL_CONVEX || s.kind == BMO_SHAPE_ACYL_CONCAVE) &&
            (s.child_begin < 0 || s.child_count < 0 || s.child_begin + s.child_count > d->n_coefs))
            return fail(BMO_ERR_INVALID, "aspheric coefficient range out of bounds");
        if (s.kind == BMO_SHAPE_MESH && (s.tri_begin < 0 || s.tri_begin + s.tri_count > d->n_tris))
            return fail(BMO_ERR_INVALID, "triangle range out of bounds");
    }
    bool has_split = false, has_asph = false;
    for (int i = 0; i < d->n_shapes; ++i)
        if (d->shapes[i].kind >= BMO_SHAPE_ASPH_CONVEX && d->shapes[i].kind <= BMO_SHAPE_ACYL_CONCAVE) has_asph = true;  // extended shapes
    for (int i = 0; i < d->n_objects; ++i) {
        const bmo_object& o = d->objects[i];
        if (o.kind < 0 || o.kind >= BMO_OBJ_KIND_COUNT) return fail(BMO_ERR_UNSUPPORTED, "unknown object kind");
        int np = (o.kind == BMO_OBJ_DOUBLET || o.kind == BMO_OBJ_PLATE_BS) ? 2 : (o.kind == BMO_OBJ_CUBE_BS ? 3 : 1);
        for (int k = 0; k < np; ++k)
            if (o.shape[k] < 0 || o.shape[k] >= d->n_shapes) return fail(BMO_ERR_INVALID, "object shape id out of bounds");
        for (int k = 0; k < 2; ++k)
            if (o.medium[k] >= d->n_media) return fail(BMO_ERR_INVALID, "medium id out of bounds");
        if ((o.kind == BMO_OBJ_REFRACTIVE || o.kind == BMO_OBJ_DOUBLET || o.kind == BMO_OBJ_PLATE_BS || o.kind == BMO_OBJ_CUBE_BS) && o.medium[0] < 0)
            return fail(BMO_ERR_INVALID, "refractive object without medium");
        if ((o.kind == BMO_OBJ_DOUBLET || o.kind == BMO_OBJ_CUBE_BS) && o.medium[1] < 0) return fail(BMO_ERR_INVALID, "missing second medium");
        if ((o.kind == BMO_OBJ_SPOTDETECTOR || o.kind == BMO_OBJ_PSFDETECTOR || o.kind == BMO_OBJ_PHOTODETECTOR) &&
            (o.detector < 0 || o.detector >= d->n_detectors))
            return fail(BMO_ERR_INVALID, "detector slot out of bounds");
        if (o.kind == BMO_OBJ_THIN_BS || o.kind == BMO_OBJ_PLATE_BS || o.kind == BMO_OBJ_CUBE_BS) has_split = true;
    }
    auto sc = std::make_unique<bmo_scene>();
    auto al = [](size_t x) { return (x + 15) & ~(size_t)15; };
    BlobHeader h{};
    h.n_objects = d->n_objects;
    h.n_shapes = d->n_shapes;
    h.n_children = d->n_children;
    h.n_tris = d->n_tris;
    h.n_media = d->n_media;
    h.n_lambda = d->n_lambda;
    h.n_detectors = d->n_detectors;
    h.march_iters = d->march_iters;
    h.eps_srf = d->eps_srf;
    h.eps_ray = d->eps_ray;
    h.eps_ins = d->eps_ins;
    h.mt_keps = d->mt_keps;
    h.mt_leps = d->mt_leps;
    h.grad_h = d->grad_h;
    h.has_splitter = has_split ? 1 : 0;
    h.has_asphere = has_asph ? 1 : 0;
    for (int i = 0; i < d->n_shapes; ++i)
        if (d->shapes[i].kind == BMO_SHAPE_MENISCUS) h.has_meniscus = 1;
    size_t off = al(sizeof(BlobHeader));
    h.off_objects = (uint32_t)off;
    off = al(off + sizeof(bmo_object) * (size_t)d->n_objects);
    h.off_shapes = (uint32_t)off;
    off = al(off + sizeof(bmo_shape) * (size_t)d->n_shapes);
    h.off_children = (uint32_t)off;
    off = al(off + 4 * (size_t)d->n_children);
    h.off_tris = (uint32_t)off;
    off = al(off + 72 * (size_t)d->n_tris);
    h.off_ntable = (uint32_t)off;
    off = al(off + 8 * (size_t)std::max(1, d->n_media) * (size_t)d->n_lambda);
    h.off_coefs = (uint32_t)off;
    off = al(off + 8 * (size_t)std::max(1, d->n_coefs));
    h.off_cands = (uint32_t)off;
    h.n_cands = fill_candidates(d->objects, d->n_objects, d->shapes, nullptr);
    off = al(off + sizeof(Cand) * (size_t)(std::max(1, h.n_cands) + 3));  // (+ 3 zero entries: the collection of tracing_step reads four per trip)
    h.total = (uint32_t)off;
    sc->blob.assign(off, 0);
    std::memcpy(sc->blob.data(), &h, sizeof h);
    if (d->n_objects) std::memcpy(sc->blob.data() + h.off_objects, d->objects, sizeof(bmo_object) * (size_t)d->n_objects);
    if (d->n_shapes) std::memcpy(sc->blob.data() + h.off_shapes, d->shapes, sizeof(bmo_shape) * (size_t)d->n_shapes);
    for (int i = 0; i < d->n_shapes; ++i) {  // unions whose children sit back to back in the shape table: no children[] look-up on the device
        bmo_shape* sh = reinterpret_cast<bmo_shape*>(sc->blob.data() + h.off_shapes) + i;
        sh->flags &= ~BMO_SHAPE_FLAG_CONSECUTIVE;
        if (sh->kind != BMO_SHAPE_UNION || sh->child_count <= 0) continue;
        bool consecutive = true;
        for (int c = 1; c < sh->child_count; ++c)
            if (d->children[sh->child_begin + c] != d->children[sh->child_begin] + c) consecutive = false;
        if (consecutive) {
            sh->flags |= BMO_SHAPE_FLAG_CONSECUTIVE;
            sh->tri_begin = d->children[sh->child_begin];
        }
    }
    if (d->n_children) std::memcpy(sc->blob.data() + h.off_children, d->children, 4 * (size_t)d->n_children);
    if (d->n_tris) std::memcpy(sc->blob.data() + h.off_tris, d->tris, 72 * (size_t)d->n_tris);
    if (d->n_media) std::memcpy(sc->blob.data() + h.off_ntable, d->n_table, 8 * (size_t)d->n_media * (size_t)d->n_lambda);
    if (d->n_coefs > 0) std::memcpy(sc->blob.data() + h.off_coefs, d->coefs, 8 * (size_t)d->n_coefs);
    fill_candidates(d->objects, d->n_objects, d->shapes, reinterpret_cast<Cand*>(sc->blob.data() + h.off_cands));  // trace_all's flat slot list
    {  // the scene's bounding sphere: centre of the box around the candidates' spheres, radius to the farthest of them
        const Cand* cd = reinterpret_cast<const Cand*>(sc->blob.data() + h.off_cands);
        double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
        int nb = 0;
        for (int i = 0; i < h.n_cands; ++i) {
            if (!(cd[i].R >= 0.0) || !std::isfinite(cd[i].R)) continue;
            const double c[3] = {cd[i].cx, cd[i].cy, cd[i].cz};
            for (int a = 0; a < 3; ++a) lo[a] = std::min(lo[a], c[a] - cd[i].R), hi[a] = std::max(hi[a], c[a] + cd[i].R);
            ++nb;
        }
        if (nb) {
            double R = 0.0;
            for (int a = 0; a < 3; ++a) sc->bound[a] = 0.5 * (lo[a] + hi[a]);
            for (int i = 0; i < h.n_cands; ++i) {
                if (!(cd[i].R >= 0.0) || !std::isfinite(cd[i].R)) continue;
                const double dx = cd[i].cx - sc->bound[0], dy = cd[i].cy - sc->bound[1], dz = cd[i].cz - sc->bound[2];
                R = std::max(R, std::sqrt(dx * dx + dy * dy + dz * dz) + cd[i].R);
            }
            sc->bound[3] = R;
        }
    }
    sc->hdr = h;
    *out = sc.release();
    return BMO_OK;
}

int bmo_scene_destroy(bmo_scene* s) {
    delete s;
    return BMO_OK;
}

int bmo_pool_release(void) {
    pool_release_all();
    return BMO_OK;
}

int bmo_batch_upload(bmo_scene* scene, const bmo_ray_batch* in, int32_t device, bmo_device_batch** out) {
    if (!scene || !in || !out) return fail(BMO_ERR_INVALID, "null argument");
    if (in->kind != BMO_BEAM_RAY && in->kind != BMO_BEAM_POLARIZED && in->kind != BMO_BEAM_GAUSSIAN) return fail(BMO_ERR_INVALID, "bad beam kind");
    const int want = in->kind == BMO_BEAM_RAY ? BMO_PLANES_RAY : (in->kind == BMO_BEAM_POLARIZED ? BMO_PLANES_POLARIZED : BMO_PLANES_GAUSSIAN);
    const bool continued = in->kind == BMO_BEAM_GAUSSIAN && in->n_planes == BMO_PLANES_GAUSSIAN_CONTINUED;
    if ((in->n_planes != want && !continued) || in->n < 0) return fail(BMO_ERR_INVALID, "bad plane count");
    if (in->n > 0x7fffffff / 4) return fail(BMO_ERR_INVALID, "batch too large for one device (shard it)");
    for (int64_t i = 0; i < in->n; ++i)
        if (in->lambda_idx[i] < 0 || in->lambda_idx[i] >= scene->hdr.n_lambda) return fail(BMO_ERR_INVALID, "lambda_idx out of range");
    int ndev = bmo_device_count();
    if (ndev <= 0) return fail(BMO_ERR_NO_DEVICE, "no HIP device: the engine has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(BMO_ERR_INVALID, "bad device ordinal");
    HIP_TRY(hipSetDevice(device));
    auto b = std::make_unique<bmo_device_batch>();
    b->device = device;
    b->kind = in->kind;
    b->n = in->n;
    b->n_planes = in->n_planes;
    int rc;
    if ((rc = b->planes.alloc((size_t)in->n * in->n_planes * 8)) || (rc = b->li.alloc((size_t)in->n * 4))) return rc;
    if (in->n) {
        HIP_TRY(hipMemcpy(b->planes.p, in->planes, (size_t)in->n * in->n_planes * 8, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(b->li.p, in->lambda_idx, (size_t)in->n * 4, hipMemcpyHostToDevice));
    }
    // Roots are binned by coherence key once, here (a stable sort: bundle order within a key).  A bundle whose rays all have one key —
    // a single disc source aimed at one train — keeps its order and costs nothing later; BMO_NO_BINNING=1 switches the step off.
    // BMO_ROOT_ORDER = chord (default: root_order_key_kernel) | mask (round 3's candidate-set keys, below) | none
    const char* order_env = getenv("BMO_ROOT_ORDER");
    std::string root_order = order_env ? order_env : (getenv("BMO_NO_BINNING") ? "none" : "auto");
    if (in->n >= 4096 && scene->hdr.n_cands > 0 && scene->bound[3] > 0.0 && (root_order == "chord" || root_order == "auto")) {
        const int64_t n = in->n;
        DevBuf chord, lim, keys, keys_out, ids, tmp;
        if ((rc = chord.alloc((size_t)n * 6 * 8)) || (rc = lim.alloc(13 * 8)) || (rc = keys.alloc((size_t)n * 8)) || (rc = keys_out.alloc((size_t)n * 8)) ||
            (rc = ids.alloc((size_t)n * 4)) || (rc = b->perm.alloc((size_t)n * 4)))
            return rc;
        const unsigned nb = (unsigned)((n + 255) / 256);
        hipLaunchKernelGGL(root_chord_kernel, dim3(nb), dim3(256), 0, nullptr, (const double*)b->planes.p, n, scene->bound[0], scene->bound[1], scene->bound[2],
                           scene->bound[3], (double*)chord.p);
        size_t tb = 0, tb2 = 0;
        HIP_TRY(hipcub::DeviceReduce::Min(nullptr, tb, (const double*)chord.p, (double*)lim.p, (int)n, nullptr));
        HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tb2, (const unsigned long long*)keys.p, (unsigned long long*)keys_out.p, (const int32_t*)ids.p,
                                                   (int32_t*)b->perm.p, (int)n, 0, 64, nullptr));
        if ((rc = tmp.alloc(std::max(tb, tb2)))) return rc;
        for (int q = 0; q < 6; ++q) {
            size_t t1 = tmp.bytes;
            HIP_TRY(hipcub::DeviceReduce::Min(tmp.p, t1, (const double*)chord.p + (size_t)q * n, (double*)lim.p + q, (int)n, nullptr));
            t1 = tmp.bytes;
            HIP_TRY(hipcub::DeviceReduce::Max(tmp.p, t1, (const double*)chord.p + (size_t)q * n, (double*)lim.p + 6 + q, (int)n, nullptr));
        }
        bool own_order = false;  // keep the bundle's own numbering (+ the candidate-set bins below)
        if (root_order == "auto") {
            double jump = 0.0;
            HIP_TRY(hipMemsetAsync((double*)lim.p + 12, 0, 8, nullptr));
            hipLaunchKernelGGL(root_ring_jump_kernel, dim3(nb), dim3(256), 0, nullptr, (const double*)chord.p, n, (const double*)lim.p, (double*)lim.p + 12);
            HIP_TRY(hipMemcpy(&jump, (const double*)lim.p + 12, 8, hipMemcpyDeviceToHost));
            own_order = jump / (double)n < 0.01;  // (a spiral: ~1e-6; positions or directions drawn at random: ~0.3)
            DBG("root order: mean ring jump %.3g -> %s", jump / (double)n, own_order ? "the bundle's own order" : "Morton order of the chords");
        }
        if (own_order) {
            b->perm.release();
            root_order = "mask";
        } else {
        root_order = "chord";
        hipLaunchKernelGGL(root_order_key_kernel, dim3(nb), dim3(256), 0, nullptr, (const double*)chord.p, n, (const double*)lim.p, (unsigned long long*)keys.p,
                           (int32_t*)ids.p);
        size_t t2 = tmp.bytes;
        HIP_TRY(hipcub::DeviceRadixSort::SortPairs(tmp.p, t2, (const unsigned long long*)keys.p, (unsigned long long*)keys_out.p, (const int32_t*)ids.p,
                                                   (int32_t*)b->perm.p, (int)n, 0, 64, nullptr));
        if ((rc = b->binned.alloc((size_t)n * in->n_planes * 8))) return rc;
        hipLaunchKernelGGL(bin_planes_kernel, dim3(nb), dim3(256), 0, nullptr, (const double*)b->planes.p, (const int32_t*)b->perm.p, n, (int)in->n_planes,
                           (double*)b->binned.p);
        HIP_TRY(hipDeviceSynchronize());  // the temporaries go back to the pool
        HIP_TRY(hipGetLastError());
        }
    }
    if (root_order == "auto") root_order = "mask";  // (no bound to take chords through)
    if (in->n >= 4096 && scene->hdr.n_cands > 0 && root_order == "mask") {
        const char* dblob = scene->device_blob(device, rc);
        if (rc) return rc;
        const int64_t n = in->n;
        DevBuf keys, keys_out, ids, tmp;
        if ((rc = keys.alloc((size_t)n * 8)) || (rc = keys_out.alloc((size_t)n * 8)) || (rc = ids.alloc((size_t)n * 4)) || (rc = b->perm.alloc((size_t)n * 4))) return rc;
        hipLaunchKernelGGL(root_key_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, dblob, scene->hdr, (const double*)b->planes.p, n,
                           (unsigned long long*)keys.p, (int32_t*)ids.p);
        const int bits = std::min(64, std::max(1, scene->hdr.n_cands));
        size_t tb = 0;
        HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, (const unsigned long long*)keys.p, (unsigned long long*)keys_out.p, (const int32_t*)ids.p,
                                                   (int32_t*)b->perm.p, (int)n, 0, bits, nullptr));
        if ((rc = tmp.alloc(tb))) return rc;
        HIP_TRY(hipcub::DeviceRadixSort::SortPairs(tmp.p, tb, (const unsigned long long*)keys.p, (unsigned long long*)keys_out.p, (const int32_t*)ids.p,
                                                   (int32_t*)b->perm.p, (int)n, 0, bits, nullptr));
        unsigned long long k0 = 0, k1 = 0;
        HIP_TRY(hipMemcpy(&k0, keys_out.p, 8, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(&k1, (const char*)keys_out.p + (size_t)(n - 1) * 8, 8, hipMemcpyDeviceToHost));
        if (k0 == k1) {
            b->perm.release();  // one key: nothing to bin
        } else {
            if ((rc = b->binned.alloc((size_t)n * in->n_planes * 8))) return rc;
            hipLaunchKernelGGL(bin_planes_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, (const double*)b->planes.p, (const int32_t*)b->perm.p, n,
                               (int)in->n_planes, (double*)b->binned.p);
            HIP_TRY(hipDeviceSynchronize());  // the sort's temporaries go back to the pool
        }
    }
    *out = b.release();
    return BMO_OK;
}

int bmo_batch_free(bmo_device_batch* b) {
    if (b) (void)hipSetDevice(b->device);
    delete b;
    return BMO_OK;
}

static int trace_or_retrace(bmo_scene* scene, bmo_device_batch* batch, const bmo_trace_opts* opts, bmo_trace_result* prev, bmo_trace_result** out) {
    if (!scene || !batch || !opts || !out) return fail(BMO_ERR_INVALID, "null argument");
    if (prev && prev->n_objects != scene->hdr.n_objects)
        return fail(BMO_ERR_INVALID, "retrace: the scene does not have the object numbering the previous solution was solved with");
    auto R = std::make_unique<bmo_trace_result>();
    R->n_objects = scene->hdr.n_objects;
    int rc;
    if (batch->kind == BMO_BEAM_RAY) rc = run_trace<BMO_BEAM_RAY>(scene, batch, opts, R.get(), prev);
    else if (batch->kind == BMO_BEAM_POLARIZED) rc = run_trace<BMO_BEAM_POLARIZED>(scene, batch, opts, R.get(), prev);
    else if (batch->kind == BMO_BEAM_GAUSSIAN) rc = run_trace<BMO_BEAM_GAUSSIAN>(scene, batch, opts, R.get(), prev);
    else return fail(BMO_ERR_UNSUPPORTED, "beam kind not built yet");
    if (rc) return rc;
    *out = R.release();
    return BMO_OK;
}

int bmo_trace_device(bmo_scene* scene, bmo_device_batch* batch, const bmo_trace_opts* opts, bmo_trace_result** out) {
    return trace_or_retrace(scene, batch, opts, nullptr, out);
}

int bmo_retrace_device(bmo_scene* scene, bmo_device_batch* batch, bmo_trace_result* prev, const bmo_trace_opts* opts, bmo_trace_result** out) {
    if (!prev) return fail(BMO_ERR_INVALID, "retrace: no previous solution");
    return trace_or_retrace(scene, batch, opts, prev, out);
}

int bmo_retrace(bmo_scene* scene, const bmo_ray_batch* in, bmo_trace_result* prev, const bmo_trace_opts* opts, bmo_trace_result** out) {
    if (!scene || !in || !opts || !out || !prev) return fail(BMO_ERR_INVALID, "null argument");
    bmo_device_batch* b = nullptr;
    int rc = bmo_batch_upload(scene, in, prev->device, &b);  // the stored solution pins the device
    if (rc) return rc;
    rc = bmo_retrace_device(scene, b, prev, opts, out);
    bmo_batch_free(b);
    return rc;
}

int bmo_trace(bmo_scene* scene, const bmo_ray_batch* in, const bmo_trace_opts* opts, bmo_trace_result** out) {
    if (!opts) return fail(BMO_ERR_INVALID, "null opts");
    bmo_device_batch* b = nullptr;
    int rc = bmo_batch_upload(scene, in, opts->device, &b);
    if (rc) return rc;
    rc = bmo_trace_device(scene, b, opts, out);
    bmo_batch_free(b);
    return rc;
}

int bmo_result_device_hits(bmo_trace_result* r, int32_t det, const double** data, int64_t* count) {
    if (!r || det < 0 || det >= r->n_detectors) return fail(BMO_ERR_INVALID, "bad detector");
    *count = r->det_count[det];
    *data = r->det_data.p ? static_cast<const double*>(r->det_data.p) + 9 * r->det_offset[det] : nullptr;
    return BMO_OK;
}

int bmo_result_copy_hits(bmo_trace_result* r, int32_t det, double* dst, int64_t max_hits) {
    if (!r || det < 0 || det >= r->n_detectors) return fail(BMO_ERR_INVALID, "bad argument");
    const int64_t n = std::min<int64_t>(max_hits, r->det_count[det]);
    if (n <= 0) return BMO_OK;  // nothing to copy: an empty destination (NULL data pointer of a 0-row tensor) is fine
    if (!dst) return fail(BMO_ERR_INVALID, "null destination");
    HIP_TRY(hipSetDevice(r->device));
    HIP_TRY(hipMemcpy(dst, static_cast<const double*>(r->det_data.p) + 9 * r->det_offset[det], (size_t)n * 72, hipMemcpyDefault));
    return BMO_OK;
}

int bmo_result_timing(bmo_trace_result* r, double* k, double* t, int32_t* n) {
    if (!r) return fail(BMO_ERR_INVALID, "null result");
    if (k) *k = r->kernel_ms;
    if (t) *t = r->total_ms;
    if (n) *n = r->n_steps;
    return BMO_OK;
}

int bmo_result_counts(bmo_trace_result* r, int64_t* calls, int64_t* records, int64_t* nodes, int64_t* hits) {
    if (!r) return fail(BMO_ERR_INVALID, "null result");
    if (calls) *calls = (int64_t)r->calls;
    if (records) *records = r->n_records;
    if (nodes) *nodes = r->n_nodes;
    if (hits) {
        *hits = 0;
        for (int d = 0; d < r->n_detectors; ++d) *hits += r->det_count[d];
    }
    return BMO_OK;
}

int bmo_result_view_select(bmo_trace_result* r, uint32_t what, bmo_trace_result_view* v) {
    if (!r || !v) return fail(BMO_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(r->device));
    const int64_t nn = r->n_nodes, nr = r->n_records;
    const int P = r->abi_planes;
    int rc;
    if (nr >= ((int64_t)1 << 31)) return fail(BMO_ERR_UNSUPPORTED, "result view: more than 2^31 segments (node_first_rec is 32-bit)");
    const unsigned nb = (unsigned)((nn + 255) / 256);
    if (!r->nodes_viewed) {
        // canonical node tables on the device, one pinned copy each
        const size_t n1 = (size_t)std::max<int64_t>(nn, 1);
        DevBuf d_root, d_parent, d_fc, d_nseg, d_status, d_aux, d_last, tmp;
        if ((rc = r->c_rank.alloc(n1 * 4)) || (rc = r->c_first_rec.alloc(n1 * 4)) || (rc = d_root.alloc(n1 * 4)) || (rc = d_parent.alloc(n1 * 4)) ||
            (rc = d_fc.alloc(n1 * 4)) || (rc = d_nseg.alloc(n1 * 4)) || (rc = d_status.alloc(n1 * 4)) || (rc = d_aux.alloc(n1 * 32)) ||
            (rc = r->h_root.alloc(n1 * 4)) || (rc = r->h_parent.alloc(n1 * 4)) || (rc = r->h_first_child.alloc(n1 * 4)) || (rc = r->h_first_rec.alloc(n1 * 4)) ||
            (rc = r->h_last_rec.alloc(n1 * 4)) || (rc = r->h_nseg.alloc(n1 * 4)) || (rc = r->h_status.alloc(n1 * 4)) || (rc = r->h_aux.alloc(n1 * 32)))
            return rc;
        if (nn > 0) {
            hipLaunchKernelGGL(rank_kernel, dim3(nb), dim3(256), 0, 0, (const int32_t*)r->order.p, nn, (int32_t*)r->c_rank.p);
            HIP_TRY(hipMemsetAsync(d_fc.p, 0xFF, (size_t)nn * 4, 0));
            hipLaunchKernelGGL(canon_nodes_kernel, dim3(nb), dim3(256), 0, 0, (const int32_t*)r->order.p, (const int32_t*)r->c_rank.p, nn,
                               r->kind == BMO_BEAM_GAUSSIAN ? 1 : 0, (const int32_t*)r->n_root.p, (const int32_t*)r->n_parent.p, (const int32_t*)r->n_nseg.p,
                               (const int32_t*)r->n_status.p, (const unsigned long long*)r->n_key.p, (const double*)r->n_lambda.p, (const double*)r->n_aux.p,
                               (int32_t*)d_root.p, (int32_t*)d_parent.p, (int32_t*)d_fc.p, (int32_t*)d_nseg.p, (int32_t*)d_status.p, (double*)d_aux.p);
            size_t tb = 0;
            HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, (const int32_t*)d_nseg.p, (int32_t*)r->c_first_rec.p, (int)nn, (hipStream_t)0));
            if ((rc = tmp.alloc(tb))) return rc;
            HIP_TRY(hipcub::DeviceScan::ExclusiveSum(tmp.p, tb, (const int32_t*)d_nseg.p, (int32_t*)r->c_first_rec.p, (int)nn, (hipStream_t)0));
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(r->h_root.p, d_root.p, (size_t)nn * 4, hipMemcpyDeviceToHost, 0));
            HIP_TRY(hipMemcpyAsync(r->h_parent.p, d_parent.p, (size_t)nn * 4, hipMemcpyDeviceToHost, 0));
            HIP_TRY(hipMemcpyAsync(r->h_first_child.p, d_fc.p, (size_t)nn * 4, hipMemcpyDeviceToHost, 0));
            HIP_TRY(hipMemcpyAsync(r->h_first_rec.p, r->c_first_rec.p, (size_t)nn * 4, hipMemcpyDeviceToHost, 0));
            HIP_TRY(hipMemcpyAsync(r->h_nseg.p, d_nseg.p, (size_t)nn * 4, hipMemcpyDeviceToHost, 0));
            HIP_TRY(hipMemcpyAsync(r->h_status.p, d_status.p, (size_t)nn * 4, hipMemcpyDeviceToHost, 0));
            HIP_TRY(hipMemcpyAsync(r->h_aux.p, d_aux.p, (size_t)nn * 32, hipMemcpyDeviceToHost, 0));
            // LAST view: record i belongs to node i (d_root is free again once its copy is queued: the stream keeps the order)
            hipLaunchKernelGGL(iota_kernel, dim3(nb), dim3(256), 0, 0, (int32_t*)d_root.p, nn);
            HIP_TRY(hipMemcpyAsync(r->h_last_rec.p, d_root.p, (size_t)nn * 4, hipMemcpyDeviceToHost, 0));
            HIP_TRY(hipStreamSynchronize(0));  // the staging buffers go back to the pool when this scope ends
        }
        r->nodes_viewed = true;
    }
    int64_t tot = 0;
    for (int d = 0; d < r->n_detectors; ++d) tot += r->det_count[d];
    if ((what & BMO_VIEW_HITS) && !r->hits_viewed) {
        if ((rc = r->h_det.alloc((size_t)tot * 72)) || (rc = r->h_det_node.alloc((size_t)tot * 4))) return rc;
        if (tot > 0) {  // queued behind the node tables, completed by the synchronisation at the end of this call
            HIP_TRY(hipMemcpyAsync(r->h_det.p, r->det_data.p, (size_t)tot * 72, hipMemcpyDeviceToHost, 0));
            HIP_TRY(hipMemcpyAsync(r->h_det_node.p, r->det_node.p, (size_t)tot * 4, hipMemcpyDeviceToHost, 0));
        }
        r->hits_viewed = true;
    }
    const int want_mode = !r->has_log ? 0 : ((what & BMO_VIEW_SEGMENTS) ? 2 : ((what & BMO_VIEW_LAST_SEGMENT) ? 1 : 0));
    if (want_mode != 0 && want_mode != r->rec_mode && !(want_mode == 1 && r->rec_mode == 2)) {
        const int64_t cols = want_mode == 2 ? nr : nn;  // records on the host
        if ((rc = r->h_rec.alloc((size_t)P * cols * 8)) || (rc = r->h_rec_obj.alloc((size_t)cols * 4)) || (rc = r->h_rec_shape.alloc((size_t)cols * 4))) return rc;
        if (cols > 0) {
            // re-order on the device, then one copy per table
            DevBuf d_base, d_rec, d_obj, d_shape;
            if ((rc = d_rec.alloc((size_t)P * cols * 8)) || (rc = d_obj.alloc((size_t)cols * 4)) || (rc = d_shape.alloc((size_t)cols * 4))) return rc;
            // every beam has a last record and every (node, k) of the log exactly one record: both forms write every column
            // (a hole in the tables would be a bug of the log, which test_selective_views / compare() would show as garbage)
            if (want_mode == 2) {
                if ((rc = d_base.alloc((size_t)nn * 4))) return rc;
                hipLaunchKernelGGL(dst_base_kernel, dim3(nb), dim3(256), 0, 0, (const int32_t*)r->order.p, (const int32_t*)r->c_first_rec.p, nn, (int32_t*)d_base.p);
            }
            for (const Chunk& c : r->chunks) {
                if (c.count <= 0) continue;
                if (want_mode == 2)
                    hipLaunchKernelGGL(order_records_kernel, dim3((unsigned)((c.count + 255) / 256)), dim3(256), 0, 0, c, P, (const int32_t*)d_base.p, nr,
                                       (double*)d_rec.p, (int32_t*)d_obj.p, (int32_t*)d_shape.p);
                else
                    hipLaunchKernelGGL(last_records_kernel, dim3((unsigned)((c.count + 255) / 256)), dim3(256), 0, 0, c, P, (const int32_t*)r->c_rank.p,
                                       (const int32_t*)r->n_nseg.p, nn, (double*)d_rec.p, (int32_t*)d_obj.p, (int32_t*)d_shape.p);
            }
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(r->h_rec.p, d_rec.p, (size_t)P * cols * 8, hipMemcpyDeviceToHost, 0));
            HIP_TRY(hipMemcpyAsync(r->h_rec_obj.p, d_obj.p, (size_t)cols * 4, hipMemcpyDeviceToHost, 0));
            HIP_TRY(hipMemcpyAsync(r->h_rec_shape.p, d_shape.p, (size_t)cols * 4, hipMemcpyDeviceToHost, 0));
            HIP_TRY(hipStreamSynchronize(0));  // the staging buffers go back to the pool when this scope ends
        }
        r->rec_mode = want_mode;
    }
    HIP_TRY(hipStreamSynchronize(0));
    // what this view shows: the whole log only when asked for it (or when nothing narrower was asked and it is there)
    const int show = (what & BMO_VIEW_SEGMENTS) ? (r->rec_mode == 2 ? 2 : 0) : ((what & BMO_VIEW_LAST_SEGMENT) ? (r->rec_mode == 1 ? 1 : 0) : 0);
    if ((what & BMO_VIEW_LAST_SEGMENT) && !(what & BMO_VIEW_SEGMENTS) && r->rec_mode == 2)
        return fail(BMO_ERR_INVALID, "result view: the whole log of this result is already on the host; ask for BMO_VIEW_SEGMENTS");
    std::memset(v, 0, sizeof *v);
    v->n_roots = r->n_roots;
    v->n_nodes = r->n_nodes;
    v->n_records = show == 2 ? nr : (show == 1 ? nn : 0);  // 0 when the log was not kept (record_segments = 0) or not asked for
    v->n_intersect_calls = (int64_t)r->calls;
    v->n_steps = r->n_steps;
    v->beam_kind = r->kind;
    v->rec_planes = r->abi_planes;
    v->n_detectors = r->n_detectors;
    v->node_root = r->h_root.as<int32_t>();
    v->node_parent = r->h_parent.as<int32_t>();
    v->node_first_child = r->h_first_child.as<int32_t>();
    v->node_first_rec = show == 1 ? r->h_last_rec.as<int32_t>() : r->h_first_rec.as<int32_t>();
    v->node_nseg = r->h_nseg.as<int32_t>();
    v->node_status = r->h_status.as<int32_t>();
    v->node_aux = r->h_aux.as<double>();
    if (show) {
        v->rec_obj = r->h_rec_obj.as<int32_t>();
        v->rec_shape = r->h_rec_shape.as<int32_t>();
        v->rec = r->h_rec.as<double>();
    }
    v->det_count = r->det_count.data();
    v->det_offset = r->det_offset.data();
    if (what & BMO_VIEW_HITS) {
        v->det_node = r->h_det_node.as<int32_t>();
        v->det_data = r->h_det.as<double>();
    }
    return BMO_OK;
}

int bmo_result_view(bmo_trace_result* r, bmo_trace_result_view* v) { return bmo_result_view_select(r, BMO_VIEW_HITS | BMO_VIEW_SEGMENTS, v); }

int bmo_result_copy_hit_columns(bmo_trace_result* r, int32_t det, int32_t n_cols, double* dst, int64_t max_hits) {
    if (!r || det < 0 || det >= r->n_detectors || n_cols < 1 || n_cols > 9) return fail(BMO_ERR_INVALID, "bad argument");
    const int64_t n = std::min<int64_t>(max_hits, r->det_count[det]);
    if (n <= 0) return BMO_OK;
    if (!dst) return fail(BMO_ERR_INVALID, "null destination");
    HIP_TRY(hipSetDevice(r->device));
    const double* src = static_cast<const double*>(r->det_data.p) + 9 * r->det_offset[det];
    hipPointerAttribute_t at{};
    const bool on_device = hipPointerGetAttributes(&at, dst) == hipSuccess && at.type == hipMemoryTypeDevice;
    (void)hipGetLastError();  // a pageable host pointer makes the query fail: not an error here
    const unsigned grid = (unsigned)((n * n_cols + 255) / 256);
    if (on_device) {
        hipLaunchKernelGGL(hit_columns_kernel, dim3(grid), dim3(256), 0, 0, src, n, (int)n_cols, dst);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(0));
        return BMO_OK;
    }
    DevBuf packed;
    int rc = packed.alloc((size_t)n * n_cols * 8);
    if (rc) return rc;
    hipLaunchKernelGGL(hit_columns_kernel, dim3(grid), dim3(256), 0, 0, src, n, (int)n_cols, (double*)packed.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(dst, packed.p, (size_t)n * n_cols * 8, hipMemcpyDeviceToHost));
    return BMO_OK;
}

int bmo_result_free(bmo_trace_result* r) {
    if (r) (void)hipSetDevice(r->device);
    delete r;
    return BMO_OK;
}

}  // extern "C"

#include "bmo_readout.inc.hpp"
