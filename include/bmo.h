/*
 * bmo.h — C ABI of the MI355X-native trace engine for BeamletOptics.jl's hot path.
 *
 * This is the drop-in boundary.  The reference has no FFI; the
 * entry points below replace, for a whole batch of beams at once,
 *
 *   solve_system!(system, bg::AbstractBeamGroup; r_max, retrace)   src/System.jl:463-468
 *   solve_system!(system, beam::AbstractBeam;   r_max, retrace)    src/System.jl:444-461
 *     -> trace_system!(system, beam::Beam)                          src/System.jl:130-154
 *     -> trace_system!(system, gauss::GaussianBeamlet)              src/System.jl:274-318
 *     -> tracing_step! / trace_one / trace_all                      src/System.jl:57-110
 *     -> intersect3d(object|shape, ray)                             src/AbstractTypes/AbstractRay.jl:118-155,
 *                                                                   src/Mesh.jl:244-267, src/SDFs/AbstractSDF.jl:166-181
 *     -> interact3d(system, object, beam, ray)                      src/OpticalComponents/ (all files)
 *
 * Everything is plain C: pointers + sizes, FP64, SI metres.  No exceptions cross
 * the boundary: functions return 0 on success or a negative bmo_status code and
 * bmo_last_error() gives the thread-local message.  Data-dependent faults that
 * are exceptions in the reference (non-unit vectors OpticUtils.jl:33-35, ...)
 * are reported per beam node in node_status so the wrapper can re-raise them.
 *
 * Threading: a bmo_scene is immutable after creation and may be shared between threads; trace / retrace calls may be issued
 * from several host threads (the reference's trace loop is single-threaded, System.jl:239, :403) — calls for the same device
 * are serialised inside the library (one trace stream per device), calls for different devices run concurrently.  A
 * bmo_trace_result may be read (view, hits, retrace source, read-outs) from one thread at a time.
 *
 * The library is libbmo_hip.so (HIP, gfx950).  There is no CPU fallback inside
 * it: every trace call needs a GPU and fails with BMO_ERR_NO_DEVICE otherwise.
 * The independent CPU restatement of the reference algorithm lives in oracle/
 * (test infrastructure) and exports bmo_cpu_trace() with the same descriptor.
 */
#ifndef BMO_H
#define BMO_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BMO_ABI_VERSION 1

/* ------------------------------------------------------------------ status */
enum bmo_status {
    BMO_OK = 0,
    BMO_ERR_INVALID = -1,     /* bad descriptor / argument                     */
    BMO_ERR_NO_DEVICE = -2,   /* no HIP device / HIP runtime failure           */
    BMO_ERR_OOM = -3,         /* device or host allocation failed              */
    BMO_ERR_UNSUPPORTED = -4, /* shape/object/beam kind not built yet          */
    BMO_ERR_INTERNAL = -5,
    BMO_ERR_LIMIT = -6        /* bmo_trace_opts.max_beams exceeded             */
};

/* per-node status bits (node_status) */
enum bmo_node_status {
    BMO_NODE_MISS = 1,        /* last segment found no intersection (System.jl:142-144)   */
    BMO_NODE_STOPPED = 2,     /* interact3d returned nothing (System.jl:147-149)          */
    BMO_NODE_RMAX = 4,        /* length(rays) reached r_max (System.jl:133)               */
    BMO_NODE_SPLIT = 8,       /* children!(beam,[t,r]) was called (ThinBeamsplitter.jl:108-115) */
    BMO_NODE_DETECTED = 16,   /* a detector recorded this beam                            */
    BMO_NODE_ERR_UNIT = 32,   /* refraction3d unit-length ArgumentError (OpticUtils.jl:33-35) */
    BMO_NODE_GAUSS_DIVERGED = 64, /* chief/waist/div did not hit same shape (System.jl:298-304) */
    BMO_NODE_BLOCKED = 128,   /* PolarizationFilter blocked ray (PolarizationFilter.jl:42) */
    BMO_NODE_ERR_ORTHO = 256, /* E0 not orthogonal to dir, ErrorException (PolarizedRays.jl:54-56) */
    BMO_NODE_RETRACE_STALE = 512 /* retrace only, a NOTE (the result is the reference's): retrace_system! acted on stale data here.
                                    (i) The stored beam had children and the re-walk ended in a `nothing` interaction before
                                    reaching the splitter: the reference sets no cleanup flag (System.jl:232-239), keeps the
                                    children and retraces each from its stored first ray — so does this library (the beam then
                                    has children without BMO_NODE_SPLIT).  (ii) A GaussianBeamlet split BEFORE the end of its
                                    stored path: the children's w0 / E0 come from gauss_parameters(gauss, length(gauss)) with
                                    the stale tail still attached (ThinBeamsplitter.jl:125; the tail is cut after the loop,
                                    System.jl:417-421) — reproduced from the previous solution's segment log.
                                    (Rounds 1-3 dropped such children / sized them at the split and asked the wrapper to fall back.) */
};

/* ------------------------------------------------------------------ shapes */
enum bmo_shape_kind {
    BMO_SHAPE_MESH = 0,        /* src/Mesh.jl:33-39;  tri_begin/tri_count, world-space vertices   */
    BMO_SHAPE_SPHERE = 1,      /* SphericalLensSDF.jl:72-89   p = {radius}                         */
    BMO_SHAPE_PLANO = 2,       /* SphericalLensSDF.jl:41-65   p = {thickness, diameter}            */
    BMO_SHAPE_CONVEX = 3,      /* SphericalLensSDF.jl:186-232 p = {radius, diameter, sag, height}  */
    BMO_SHAPE_CONCAVE = 4,     /* SphericalLensSDF.jl:131-170 p = {radius, diameter, sag}          */
    BMO_SHAPE_UNION = 5,       /* UnionSDF.jl:22-91           children                             */
    BMO_SHAPE_BOX = 6,         /* PrimitiveSDF.jl:13-46       p = {hx, hy, hz} half edge lengths   */
    BMO_SHAPE_CYLINDER = 7,    /* PrimitiveSDF.jl:53-76       p = {radius, height}                 */
    BMO_SHAPE_CUTSPHERE = 8,   /* PrimitiveSDF.jl:83-124      p = {radius, height, w}              */
    BMO_SHAPE_RING = 9,        /* PrimitiveSDF.jl:132-166     p = {inner_radius, hwidth, hthickness} */
    BMO_SHAPE_PRISM = 10,      /* PrimitiveSDF.jl:183-210     p = {hx, hy, hz}                     */
    BMO_SHAPE_MENISCUS = 11,   /* MeniscusLensSDF.jl:19-46    children = {convex, cylinder, concave} in the meniscus frame */
    BMO_SHAPE_POINT = 12,      /* test/runtests.jl:926-947 TestPointSDF: sdf = norm(p)             */
    BMO_SHAPE_ASPH_CONVEX = 13,  /* AsphericalLensSDF.jl:19-28,309-327  p = {radius, conic, diameter, max_sag}; coefs = child range */
    BMO_SHAPE_ASPH_CONCAVE = 14, /* AsphericalLensSDF.jl:85-96,329-349  p = {radius, conic, diameter, max_sag}; coefs = child range */
    BMO_SHAPE_CYL_CONVEX = 15,   /* CylindricalSDF.jl:25-85   p = {radius, diameter, height}           */
    BMO_SHAPE_CYL_CONCAVE = 16,  /* CylindricalSDF.jl:92-139  p = {radius, diameter, height}           */
    BMO_SHAPE_ACYL_CONVEX = 17,  /* AcylindricalSDF.jl:16-74   p = {radius, diameter, height, conic, max_sag}; coefs = child range */
    BMO_SHAPE_ACYL_CONCAVE = 18, /* AcylindricalSDF.jl:83-141  p = {radius, diameter, height, conic, max_sag}; coefs = child range */
    BMO_SHAPE_KIND_COUNT
};

#define BMO_SHAPE_NPARAM 8
/* The SDF is a first-order distance ESTIMATE (aspheres: |z - z(r)| / |grad|), not 1-Lipschitz: only the bounding-sphere
 * culls apply to it, not the running-t prune nor the union child skip (DESIGN.md "miss cull").  Unions inherit it. */
#define BMO_SHAPE_FLAG_INEXACT 1
/* Library-internal (set by bmo_scene_create on its own copy of the tables, ignored on input): the children of this UNION are
 * consecutive shape ids starting at tri_begin (unused by unions otherwise), so the walk over children needs no children[] read. */
#define BMO_SHAPE_FLAG_CONSECUTIVE 2

typedef struct bmo_shape {
    int32_t kind;
    int32_t child_begin;   /* UNION, MENISCUS: index into bmo_scene_desc.children;
                              ASPH_*: index into bmo_scene_desc.coefs (even aspheric coefficients A4, A6, ...) */
    int32_t child_count;
    int32_t tri_begin;     /* MESH: first triangle                                   */
    int32_t tri_count;
    int32_t flags;         /* BMO_SHAPE_FLAG_* */
    double pos[3];         /* position(shape)                                        */
    double dir[9];         /* orientation(shape), row-major 3x3                      */
    double tdir[9];        /* transposed_orientation(shape) (AbstractSDF.jl:20-27), row-major;
                              identity for SPHERE (SphericalLensSDF.jl:82-84)        */
    double p[BMO_SHAPE_NPARAM];
    /* Conservative world-space bounding sphere, used ONLY for the provably
       equivalent miss shortcut (DESIGN.md "miss cull"); radius < 0 disables it. */
    double bs_center[3];
    double bs_radius;
} bmo_shape;

/* ----------------------------------------------------------------- objects */
enum bmo_object_kind {
    BMO_OBJ_MIRROR = 0,          /* AbstractReflectiveOptic   Mirrors.jl:39-69                    */
    BMO_OBJ_REFRACTIVE = 1,      /* Lens / Prism              Lenses.jl:46-126                    */
    BMO_OBJ_DOUBLET = 2,         /* DoubletLens               DoubletLenses.jl:26-76  shape={front,back} medium={n1,n2} */
    BMO_OBJ_THIN_BS = 3,         /* ThinBeamsplitter          ThinBeamsplitter.jl:16-168          */
    BMO_OBJ_PLATE_BS = 4,        /* plate splitter            PlateBeamsplitter.jl:160-275 shape={substrate,coating}    */
    BMO_OBJ_CUBE_BS = 5,         /* cube splitter             CubeBeamsplitter.jl:63-121   shape={front,back,coating}   */
    BMO_OBJ_SPOTDETECTOR = 6,    /* Spotdetector.jl:50-61                                         */
    BMO_OBJ_PSFDETECTOR = 7,     /* PSFDetector.jl:77-89                                          */
    BMO_OBJ_INTERSECTABLE = 8,   /* Intersectable.jl:15                                           */
    BMO_OBJ_NONINTERACTABLE = 9, /* NonInteractable.jl:19-20                                      */
    BMO_OBJ_POLARIZER = 10,      /* PolarizationFilter.jl:31-48                                   */
    BMO_OBJ_PHOTODETECTOR = 11,  /* Detectors/Photodetector.jl:57-107: a GaussianBeamlet that hits it stops and is
                                    recorded in the detector slot (3 hit rows per beamlet, row 0 = {proj, 0...});
                                    the complex field is read out with bmo_photodetector_field.  Ray / PolarizedRay
                                    beams stop without a record (Photodetector.jl:57-60 warns)              */
    BMO_OBJ_KIND_COUNT
};

typedef struct bmo_object {
    int32_t kind;
    int32_t shape[3];      /* part shapes, -1 when unused (see kind)                 */
    int32_t medium[2];     /* rows of n_table for refractive parts, -1 when unused   */
    int32_t detector;      /* detector slot (0..n_detectors-1) or -1                 */
    int32_t reserved;
    double reflectance;    /* amplitude R (ThinBeamsplitter.jl:47-50)                */
    double transmittance;  /* amplitude T                                            */
    double cutoff;         /* PolarizationFilter cutoff                              */
    double jones[18];      /* GlobalJonesBasis 3x3 complex, row-major (re,im) pairs  */
} bmo_object;

/* ------------------------------------------------------------------- scene */
typedef struct bmo_scene_desc {
    int32_t abi_version;   /* BMO_ABI_VERSION */
    int32_t n_objects;     /* leaf objects in Leaves() order (System.jl:21)          */
    int32_t n_shapes;
    int32_t n_children;
    int32_t n_tris;
    int32_t n_media;
    int32_t n_lambda;
    int32_t n_detectors;
    const bmo_object* objects;
    const bmo_shape* shapes;
    const int32_t* children;  /* shape ids                                           */
    const double* tris;       /* 9 doubles per triangle: V1 V2 V3 (world space)      */
    const double* n_table;    /* [n_media][n_lambda]: n_obj(lambda) evaluated by the host
                                 (RefractiveIndexUtils.jl functors cannot cross a C ABI) */
    const double* lambdas;    /* [n_lambda] distinct wavelengths of the batch        */
    const double* coefs;      /* [n_coefs] pooled aspheric coefficients              */
    /* tracing constants; the reference's compile-time values are the defaults */
    double eps_srf;        /* 1e-9  AbstractSDF.jl:1  */
    double eps_ray;        /* 1e-10 AbstractSDF.jl:2  */
    double eps_ins;        /* 1.0   AbstractSDF.jl:3  */
    double mt_keps;        /* 1e-9  Mesh.jl:203       */
    double mt_leps;        /* 1e-9  Mesh.jl:203       */
    double grad_h;         /* 1e-8  AbstractSDF.jl:83 */
    int32_t march_iters;   /* 1000  AbstractSDF.jl:105,135 */
    int32_t n_coefs;
} bmo_scene_desc;

typedef struct bmo_scene bmo_scene; /* opaque, immutable after create, shareable */

/* ------------------------------------------------------------------- beams */
enum bmo_beam_kind {
    BMO_BEAM_RAY = 0,        /* Beam{T,Ray{T}}           Rays.jl:14-20        */
    BMO_BEAM_POLARIZED = 1,  /* Beam{T,PolarizedRay{T}}  PolarizedRays.jl:37-66 */
    BMO_BEAM_GAUSSIAN = 2    /* GaussianBeamlet          Gaussian.jl:33-42    */
};

/* Planar (structure-of-arrays) input: plane i occupies planes[i*n .. (i+1)*n).
 *   RAY        (8 planes):  px py pz dx dy dz lambda n
 *   POLARIZED  (14 planes): the 8 above, then Re(E0x) Im(E0x) Re(E0y) Im(E0y) Re(E0z) Im(E0z)
 *   GAUSSIAN   (25 planes): chief px..dz (6), waist px..dz (6), divergence px..dz (6),
 *                           lambda, n, w0, Re(E0), Im(E0), then 2 reserved (0)
 *              (31 planes): the 25 above, then lenA lenB l0 oplC oplW oplD — the lengths a SOLVED beamlet has accumulated up to the
 *                           segment this batch continues (solve_system!(...; retrace = false) on beamlets whose last ray is open,
 *                           System.jl:449-458): lenA = sum of the chief's own earlier segment lengths folded from 0
 *                           (length_rays, Beam.jl:160-169), lenB = the same fold started from l0, l0 = length(parent chief beam)
 *                           (Beam.jl:125-130; 0 for a root), oplC = optical path of the chief, parents included, oplW / oplD =
 *                           optical path of the waist / divergence beams' own earlier segments (Beam.jl:137-149)
 * dir must already be normalised by the caller exactly as the Ray constructor does
 * (Rays.jl:32-42); the engine does not renormalise first segments.               */
#define BMO_PLANES_RAY 8
#define BMO_PLANES_POLARIZED 14
#define BMO_PLANES_GAUSSIAN 25
#define BMO_PLANES_GAUSSIAN_CONTINUED 31

typedef struct bmo_ray_batch {
    int64_t n;                  /* number of root beams                              */
    int32_t kind;               /* bmo_beam_kind                                     */
    int32_t n_planes;
    const double* planes;       /* host pointer, n_planes * n doubles                */
    const int32_t* lambda_idx;  /* [n] index of each beam's wavelength in lambdas    */
} bmo_ray_batch;

typedef struct bmo_trace_opts {
    int32_t r_max;              /* solve_system! default 100 (System.jl:444)         */
    int32_t device;             /* HIP device ordinal                                */
    int32_t record_segments;    /* 1: keep the full segment log (reference behaviour).  0: keep only the beam tree, statuses,
                                   counts and detector hits - a solve then holds three bounce levels in HBM instead of all
                                   of them (8x less on config C2), for spot diagrams / PSFs of very large bundles; the view of
                                   such a result reports n_records = 0, and it cannot be retraced or read out as a
                                   Photodetector field (both need the segments): BMO_ERR_INVALID.                     */
    int32_t max_beams;          /* 0: no limit (reference behaviour).  > 0: stop with BMO_ERR_LIMIT once the solve holds more
                                   than this many beams (tree nodes).  A splitter facing a mirror multiplies beams on every
                                   pass; solve_system! recurses until memory runs out on such a system, and so does this
                                   library (BMO_ERR_OOM after filling HBM) unless a limit is set.                      */
} bmo_trace_opts;

/* ------------------------------------------------------------------ result
 * All arrays are owned by the result handle until bmo_result_free.  Nodes are
 * beam-tree nodes (one Beam / GaussianBeamlet each) in REFERENCE order: bundle
 * order, and inside one root's tree the BFS order of solve_system! (System.jl:446-458,
 * transmitted child before reflected child, Beamsplitters.jl:16-19).  Records are
 * ray segments grouped by node in that order, k = 0..nseg-1 inside a node.
 * Planes per record:
 *   RAY:       px py pz dx dy dz n | t nx ny nz                       (11)
 *   POLARIZED: the 11 above, then Re/Im E0x E0y E0z                   (17)
 *   GAUSSIAN:  chief(11) waist(11) divergence(11)                     (33)
 * A record whose ray has no intersection has obj = shape = -1, t = +Inf.          */
typedef struct bmo_trace_result_view {
    int64_t n_roots;
    int64_t n_nodes;
    int64_t n_records;
    int64_t n_intersect_calls;  /* intersect3d(object|hint shape, ray) calls the reference
                                   algorithm performs for this trace (BASELINE metric) */
    int32_t n_steps;            /* step-kernel launches (each advances every active beam by up to 32 bounces) */
    int32_t beam_kind;
    int32_t rec_planes;         /* planes per record                                 */
    int32_t n_detectors;
    const int32_t* node_root;
    const int32_t* node_parent;       /* node index or -1                            */
    const int32_t* node_first_child;  /* node index of transmitted child (+1 = reflected) or -1 */
    const int32_t* node_first_rec;    /* first record of the node                    */
    const int32_t* node_nseg;         /* length(rays(beam))                          */
    const int32_t* node_status;       /* bmo_node_status bits                        */
    const double* node_aux;           /* [n_nodes*4]: GAUSSIAN w0, Re(E0), Im(E0), lambda; else lambda,0,0,0 */
    const int32_t* rec_obj;           /* Intersection.object as leaf index           */
    const int32_t* rec_shape;         /* Intersection.shape as shape id              */
    const double* rec;                /* planar [rec_planes][n_records]              */
    /* detector hit lists in reference push! order */
    const int64_t* det_count;         /* [n_detectors]                               */
    const int64_t* det_offset;        /* [n_detectors] offset into det_data (in hits) */
    const int32_t* det_node;          /* [total hits] node that produced the hit     */
    const double* det_data;           /* [total hits][9]: Spot: x z 0..; PSF: hit(3) dir(3) opl proj k */
} bmo_trace_result_view;

typedef struct bmo_trace_result bmo_trace_result; /* opaque */

/* ---------------------------------------------------------------- functions */
int bmo_version(void);
const char* bmo_last_error(void);
/* SHA-256 (hex) of the engine sources this library was built from (csrc/bmo_engine.hip, bmo_lane.hpp, bmo_readout.inc.hpp and this
   header, in that order), recorded by the build; "" for a build that did not record one.  The host side compares it with the sources
   it finds next to the library and refuses a stale binary (there is no reference counterpart: housekeeping of the boundary). */
const char* bmo_source_hash(void);
/* First 16 hex digits of the SHA-256 of the compiler flag list the library was built with (__graft_entry__.HIP_FLAGS joined by
   blanks); "" for a build that did not record one.  Bit-level parity with the reference depends on some of those flags
   (-ffp-contract=off: Julia never contracts a*b+c), so the host side refuses a library built with other flags, like a stale one. */
const char* bmo_build_flags_hash(void);

/* Number of usable HIP devices (0 = none). */
int bmo_device_count(void);

/* Device self-test of the scalar rules of the lane code as the device compiler built them: Base.max / Base.min for Float64 (NaN if
   either operand is NaN, -0.0 < +0.0; the AbstractSDF leaf formulas) written without control flow against the rule written with compares,
   their ForwardDiff.Dual forms (selection: the winner's value and partials, ties to the second argument) and abs(::Dual), bit for bit over
   every pair of a table of special values (zeros, denormals, infinities, NaN, ordinary numbers); then the elementary functions of
   csrc/bmo_jlmath.hpp on ~7 000 arguments against the host build of the same code.  BMO_OK, or BMO_ERR_INTERNAL with the first mismatch
   in bmo_last_error(). */
int bmo_selftest(int32_t device);

/* The elementary functions of the step path as the engine evaluates them (csrc/bmo_jlmath.hpp: Julia Base's sin / cos / tan / acos / atan —
   base/special/trig.jl, rem_pio2.jl — restated; the reference calls them in OpticUtils.jl:121-131, LinearAlgebraUtils.jl:103-108,
   Gaussian.jl:326-345), host build, one value per call: which = 0 sin(x), 1 cos(x), 2 tan(x), 3 acos(x), 4 atan(x), 5 atan(y, x).  For a
   maintainer to compare with Base itself; bmo_selftest compares the device's results with these.  NaN for another `which`. */
double bmo_jl_trig(int32_t which, double x, double y);

int bmo_scene_create(const bmo_scene_desc* desc, bmo_scene** out);
int bmo_scene_destroy(bmo_scene* scene);
/* The engine keeps freed device blocks in a per-device pool for reuse by the next trace; this returns them to HIP. */
int bmo_pool_release(void);

/* One call = solve_system!(system, group; r_max) for a fresh (un-solved) batch:
 * upload, trace on the device, canonicalise, download.                            */
int bmo_trace(bmo_scene* scene, const bmo_ray_batch* in, const bmo_trace_opts* opts,
              bmo_trace_result** out);

/* Split form used by benchmarks and multi-GPU drivers: inputs stay resident in HBM. */
typedef struct bmo_device_batch bmo_device_batch; /* opaque */
int bmo_batch_upload(bmo_scene* scene, const bmo_ray_batch* in, int32_t device, bmo_device_batch** out);
int bmo_batch_free(bmo_device_batch* batch);
/* Runs the whole trace on the device (all bounce steps, child spawning, detector
 * hit compaction + ordering).  Results stay on the device inside *out; blocks until done. */
int bmo_trace_device(bmo_scene* scene, bmo_device_batch* batch, const bmo_trace_opts* opts,
                     bmo_trace_result** out);
/* Device pointers of the ordered detector hit buffers (for RCCL all-gather):
 * data = [count][9] doubles on the device the trace ran on.                        */
int bmo_result_device_hits(bmo_trace_result* res, int32_t detector, const double** data, int64_t* count);
/* Copies up to max_hits ordered hits (9 doubles each) of one detector into dst: a pointer on the device the
 * trace ran on (e.g. the data_ptr of a torch tensor used as RCCL all-gather input) or host memory (page-locked
 * memory copies at PCIe speed; pageable memory works, slower) - the Spotdetector read-out that does not need
 * the segment log.                                                                                          */
int bmo_result_copy_hits(bmo_trace_result* res, int32_t detector, double* dst, int64_t max_hits);
/* Kernel timing of the last trace: sum over step-kernel launches, HIP events on the trace stream. */
int bmo_result_timing(bmo_trace_result* res, double* step_kernel_ms, double* total_ms, int32_t* n_launches);
/* Size of the solution without downloading it: reference intersect3d calls, segments, beams (tree nodes), detector hits. */
int bmo_result_counts(bmo_trace_result* res, int64_t* n_intersect_calls, int64_t* n_records, int64_t* n_nodes, int64_t* n_hits);
/* Materialise host views (downloads + canonical ordering of the segment log): the whole solution, i.e.
 * bmo_result_view_select(res, BMO_VIEW_HITS | BMO_VIEW_SEGMENTS, view). */
int bmo_result_view(bmo_trace_result* res, bmo_trace_result_view* view);
/* Selective view: the wrapper that only needs what the reference's users usually read after solve_system! —
 * last(rays(beam)) (src/Beam.jl:79), the beam tree (src/AbstractTypes/AbstractBeam.jl:48-78) and the detectors' data
 * (Spotdetector.jl:27, PSFDetector.jl:57) — does not pay for the segment log (3.7 GB per solve of config C2 over PCIe).
 * The node tables (node_*) are always filled.  `what` is a mask of:
 *   BMO_VIEW_HITS          det_node / det_data (det_count / det_offset are always valid)
 *   BMO_VIEW_LAST_SEGMENT  rec / rec_obj / rec_shape hold ONE record per beam, its last ray segment: n_records = n_nodes and
 *                          node_first_rec[i] = i, while node_nseg[i] stays the beam's true length(rays(beam))
 *   BMO_VIEW_SEGMENTS      the whole log (wins over BMO_VIEW_LAST_SEGMENT)
 * Without a record flag n_records = 0 and rec* = NULL.  Every part is downloaded once per result and kept in page-locked
 * host memory; asking for BMO_VIEW_SEGMENTS after BMO_VIEW_LAST_SEGMENT replaces the record tables (earlier views' rec*
 * pointers die), asking for BMO_VIEW_LAST_SEGMENT alone after the whole log is BMO_ERR_INVALID (the log already has it).
 * A view may be taken from another host thread while the next solve of another result runs: its kernels and copies go to
 * the NULL stream, which does not synchronise with the library's (non-blocking) trace stream. */
#define BMO_VIEW_HITS 1u
#define BMO_VIEW_LAST_SEGMENT 2u
#define BMO_VIEW_SEGMENTS 4u
int bmo_result_view_select(bmo_trace_result* res, uint32_t what, bmo_trace_result_view* view);
/* bmo_result_copy_hits for the leading n_cols (1..9) columns only, packed [hits][n_cols]: a Spotdetector keeps x, y
 * (Spotdetector.jl:59), so the multi-GPU all-gather moves 16 B per hit instead of 72.  dst: device or host memory. */
int bmo_result_copy_hit_columns(bmo_trace_result* res, int32_t detector, int32_t n_cols, double* dst, int64_t max_hits);
int bmo_result_free(bmo_trace_result* res);

/* ------------------------------------------------------------------------------------------------
 * Retracing: the second and later solve_system!(system, beams) on already solved beams —
 * retrace_system! src/System.jl:188-255 (Beam), :326-428 (GaussianBeamlet), driven by solve_system! :444-461.
 *
 * `prev` is the result of the previous solve of the SAME beams (bmo_trace*, or an earlier bmo_retrace*); it stays valid and
 * is not modified.  `scene` is the system as it is now (elements moved by the caller): it must have the object and shape
 * numbering `prev` was solved with.  The batch supplies the root heads (first ray of every root beam; for Gaussian beamlets
 * also lambda, w0, E0) and must have prev's root count and beam kind.
 *
 * Every stored ray is re-intersected with the object of its stored intersection only (or with the shape hinted by the
 * preceding interaction), interacted, and the following stored ray is overwritten (replace!, Beam.jl:81-94); where the
 * stored path ends or breaks the beam is cut, its children are dropped and normal tracing continues without a hint.
 * Splitter children of an intact parent keep their identity and are re-walked with heads rewritten by the parent
 * (children!, AbstractBeam.jl:62-76; a Gaussian child keeps its stored w0, Gaussian.jl:154-161).
 * The result has the layout of a fresh trace.  n_intersect_calls counts the reference's intersect3d calls (one per
 * re-walked ray and sub-ray, plus those of the continued trace).                                                     */
int bmo_retrace(bmo_scene* scene, const bmo_ray_batch* in, bmo_trace_result* prev, const bmo_trace_opts* opts, bmo_trace_result** out);
int bmo_retrace_device(bmo_scene* scene, bmo_device_batch* batch, bmo_trace_result* prev, const bmo_trace_opts* opts,
                       bmo_trace_result** out);

/* ------------------------------------------------------------------------------------------------
 * Detector read-out: intensity(psf::PSFDetector) — src/OpticalComponents/Detectors/PSFDetector.jl:190-237.
 *
 *   field(i,j) = sum over hits h of  proj_h * cis(k_h * (opl_h + dot(p_ij - hit_h, dir_h))),   p_ij = origin + xs[i]*e1 + zs[j]*e2
 *   intensity  = abs2(field)                       (raw / unscaled, as the reference returns it)
 *
 * hits    : [n_hits][9] doubles (hit xyz, dir xyz, opl, proj, k) — the PSF rows of bmo_trace_result_view::det_data or the
 *           device pointer of bmo_result_device_hits (hits_on_device != 0: pointer on `device`).
 * xs, zs  : the n sample coordinates of the detector-local x and z axes (host arrays; the reference's LinRange + shift).
 * out_intensity : host, n*n doubles, element (i,j) at [i + n*j]  (Julia's column-major Matrix).
 * out_field     : optional host buffer, n*n complex values as (re, im) pairs, same order; may be NULL.
 * kernel_ms     : optional; HIP-event time of the accumulation kernels.
 * The sum over hits is evaluated in a fixed blocked order (deterministic, but not the reference's sequential order):
 * parity with the reference is to the floating-point tolerance of a re-associated sum, not bit-exact.            */
int bmo_psf_intensity(const double* hits, int64_t n_hits, int32_t hits_on_device, const double origin[3], const double e1[3],
                      const double e2[3], const double* xs, const double* zs, int32_t n, int32_t device, double* out_intensity,
                      double* out_field, double* kernel_ms);

/* ------------------------------------------------------------------------------------------------
 * Photodetector field: interact3d(::Photodetector, ::GaussianBeamlet, ray_id) —
 * src/OpticalComponents/Detectors/Photodetector.jl:69-107 with electric_field(gauss, r, z) src/Gaussian.jl:381-392,
 * gauss_parameters :298-353, point_on_beam src/Beam.jl:177-205, electric_field(r, z, ...) src/Utils/OpticUtils.jl:87-89.
 *
 * For every beamlet of `res` recorded on detector slot `detector` (reference order) and every grid point (i, j):
 *   p1 = position + xs[i]*orientation[:,1] + ys[j]*orientation[:,3]      (written out per component like the reference)
 *   l1 = dot(p1 - p0, d0);  r = |p1 - (p0 + l1*d0)|;  z = l0 + l1       (p0, d0: last chief ray; l0 = length up to it)
 *   field(i,j) += electric_field(gauss, r, z) * sqrt(proj)
 * The reference does this inside solve_system!; here the trace only records the hit and this call evaluates the sum on the
 * GPU from the segment log still resident in `res` (any segment of the beamlet may be selected by point_on_beam).
 *
 * position / orientation : the detector's pose (orientation row-major 3x3, columns = local axes) at solve time.
 * xs[nx], ys[ny]         : local sample coordinates (the reference's LinRange pd.x, pd.y).
 * field_inout            : host, nx*ny complex values as (re, im) pairs, element (i,j) at [i + nx*j]; contributions are ADDED
 *                          (the reference accumulates until empty!(pd)).
 * The sum over beamlets is evaluated in a fixed blocked order: parity with the reference is to FP64 re-association
 * tolerance, not bit-exact.                                                                                          */
int bmo_photodetector_field(bmo_trace_result* res, int32_t detector, const double position[3], const double orientation[9], const double* xs,
                            const double* ys, int32_t nx, int32_t ny, double* field_inout, double* kernel_ms);

/* gauss_parameters(gauss, z) — src/Gaussian.jl:298-353 with point_on_beam src/Beam.jl:177-205 — for ONE solved beamlet of `res`
 * (index in the result's node order) at n distances z along the beam (measured like the reference's: from the start of the root
 * beamlet, parents included): out[4*i .. 4*i+3] = w (local radius), R (wavefront curvature 1/r), psi (Gouy phase), w0 (local waist).
 * Evaluated on the GPU by the code bmo_photodetector_field runs per grid point; exists so that the reference's Gaussian
 * known-answer tests (test/runtests.jl:1811-1932) can be run against the device arithmetic itself.                         */
int bmo_gauss_parameters(bmo_trace_result* res, int64_t node, const double* zs, int32_t n, double* out);

/* Earlier segments of CONTINUED GaussianBeamlets — solve_system!(system, beams; retrace = false) on beamlets solved before traces their
 * open last rays on (src/System.jl:449-458, solve_leaf! :470-475); `res` is the solution of such a continuation (root beamlet i = the
 * i-th continued beamlet from its open ray on, bmo.h "31 planes").  gauss_parameters and the Photodetector field of a beamlet are
 * functions of ALL its rays (point_on_beam, length, optical_path_length: src/Beam.jl:125-205), so the rays in front of the open one are
 * handed over here before bmo_photodetector_field / bmo_gauss_parameters are called on `res`:
 *   prefix_start[n_roots + 1] : exclusive scan of the number of earlier segments per root beamlet (0 for a beamlet without any);
 *   prefix_segs[24][total]    : plane 8 b + q of segment s at [(8 b + q) * total + s]; b = chief, waist, divergence;
 *                               q = pos 0-2, dir 3-5, refractive index 6, length t 7;   total = prefix_start[n_roots];
 *   opl_parent[n_roots]       : optical_path_length(parent(chief beam)), 0 without a parent.
 * The tables are copied to the device and replace an earlier prefix of `res`.  n_roots must be the root count of `res`.        */
int bmo_result_set_gauss_prefix(bmo_trace_result* res, int64_t n_roots, const int32_t* prefix_start, const double* prefix_segs, const double* opl_parent);

#ifdef __cplusplus
}
#endif
#endif /* BMO_H */
