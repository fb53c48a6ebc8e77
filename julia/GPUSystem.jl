# GPUSystem.jl — the BeamletOptics.jl side of the drop-in boundary (SURVEY.md §8b): a container type that sends
# `solve_system!` to the MI355X engine (libbmo_hip.so, C ABI of include/bmo.h) through `ccall`.
#
# How to bind it: copy this file to `src/GPUSystem.jl` of BeamletOptics.jl, add `include("GPUSystem.jl")` after
# `include("System.jl")` in `src/BeamletOptics.jl` and `export GPUSystem`.  Nothing else of the package changes: the precedent for
# "swap the container to change the execution strategy" is `StaticSystem` (src/System.jl:38-45).
#
#     sys  = GPUSystem(System([lens, splitter, detector]))          # wraps any AbstractSystem
#     solve_system!(sys, CollimatedSource(...))                     # src/System.jl:463-468, on the GPU
#     solve_system!(sys, beam)                                      # src/System.jl:444-461 (a second call retraces, :188-255)
#
# What runs where: this file only flattens the scene (`Leaves` order, src/System.jl:21), packs the first rays into planes, calls the
# library and rebuilds `Beam.rays` / `Intersection` / `children` / detector data from the result tables, which arrive in the
# reference's own order (bundle order x the BFS order of solve_system!).  All tracing and read-out arithmetic is in the library.
# Anything the engine does not model (a user-defined AbstractObject / AbstractShape / beam type, an `n(λ)` that throws) makes the
# call fall back to the wrapped system — the result is then the reference's by construction.
#
# Julia is not installed in the image this engine was built in, so this file has never been executed; the same call sequence,
# struct layouts and flattening rules are exercised through ctypes by beamletoptics.jl_amd/system.py (its Python twin, function by
# function) and by every `-m gpu` test.  Struct sizes are asserted below against the values tests/test_abi.py pins, and
# tests/test_julia_binding.py checks this file against include/bmo.h (constants, field order, every ccall'ed symbol and its arity).
# The engine computes in Float64; beams of another element type are converted on the way in and come back as Float64 values.

const LIBBMO = get(ENV, "BMO_ENGINE_LIB", "libbmo_hip")

# ------------------------------------------------------------------------------------------------ C structs (include/bmo.h)
const BMO_ABI_VERSION = Int32(1)
const BMO_OK = Cint(0)
const BMO_ERR_UNSUPPORTED = Cint(-4)

# bmo_node_status bits
const NODE_MISS, NODE_STOPPED, NODE_RMAX, NODE_SPLIT, NODE_DETECTED = 1, 2, 4, 8, 16
const NODE_ERR_UNIT, NODE_GAUSS_DIVERGED, NODE_BLOCKED, NODE_ERR_ORTHO, NODE_RETRACE_STALE = 32, 64, 128, 256, 512

# bmo_shape_kind
const K_MESH, K_SPHERE, K_PLANO, K_CONVEX, K_CONCAVE, K_UNION, K_BOX, K_CYLINDER, K_CUTSPHERE, K_RING, K_PRISM, K_MENISCUS = Int32.(0:11)
const K_ASPH_CONVEX, K_ASPH_CONCAVE, K_CYL_CONVEX, K_CYL_CONCAVE, K_ACYL_CONVEX, K_ACYL_CONCAVE = Int32.(13:18)
const SHAPE_FLAG_INEXACT = Int32(1)
# bmo_object_kind
const O_MIRROR, O_REFRACTIVE, O_DOUBLET, O_THIN_BS, O_PLATE_BS, O_CUBE_BS, O_SPOT, O_PSF, O_INTERSECTABLE, O_NONINTERACTABLE, O_POLARIZER,
      O_PHOTODETECTOR = Int32.(0:11)
# bmo_beam_kind and the plane counts of bmo_ray_batch / bmo_trace_result_view
const BEAM_RAY, BEAM_POLARIZED, BEAM_GAUSSIAN = Int32(0), Int32(1), Int32(2)
const PLANES_IN = (8, 14, 25)
const VIEW_HITS, VIEW_LAST_SEGMENT, VIEW_SEGMENTS = UInt32(1), UInt32(2), UInt32(4)

struct BmoShape                                    # bmo_shape, 288 bytes
    kind::Int32; child_begin::Int32; child_count::Int32; tri_begin::Int32; tri_count::Int32; flags::Int32
    pos::NTuple{3, Float64}
    dir::NTuple{9, Float64}                        # orientation(shape), ROW-major
    tdir::NTuple{9, Float64}                       # transposed_orientation(shape), row-major
    p::NTuple{8, Float64}
    bs_center::NTuple{3, Float64}; bs_radius::Float64
end
struct BmoObject                                   # bmo_object, 200 bytes
    kind::Int32; shape::NTuple{3, Int32}; medium::NTuple{2, Int32}; detector::Int32; reserved::Int32
    reflectance::Float64; transmittance::Float64; cutoff::Float64
    jones::NTuple{18, Float64}
end
struct BmoSceneDesc                                # bmo_scene_desc
    abi_version::Int32; n_objects::Int32; n_shapes::Int32; n_children::Int32
    n_tris::Int32; n_media::Int32; n_lambda::Int32; n_detectors::Int32
    objects::Ptr{BmoObject}; shapes::Ptr{BmoShape}; children::Ptr{Int32}
    tris::Ptr{Float64}; n_table::Ptr{Float64}; lambdas::Ptr{Float64}; coefs::Ptr{Float64}
    eps_srf::Float64; eps_ray::Float64; eps_ins::Float64; mt_keps::Float64; mt_leps::Float64; grad_h::Float64
    march_iters::Int32; n_coefs::Int32
end
struct BmoRayBatch
    n::Int64; kind::Int32; n_planes::Int32; planes::Ptr{Float64}; lambda_idx::Ptr{Int32}
end
struct BmoTraceOpts
    r_max::Int32; device::Int32; record_segments::Int32; max_beams::Int32
end
struct BmoResultView                               # bmo_trace_result_view
    n_roots::Int64; n_nodes::Int64; n_records::Int64; n_intersect_calls::Int64
    n_steps::Int32; beam_kind::Int32; rec_planes::Int32; n_detectors::Int32
    node_root::Ptr{Int32}; node_parent::Ptr{Int32}; node_first_child::Ptr{Int32}; node_first_rec::Ptr{Int32}
    node_nseg::Ptr{Int32}; node_status::Ptr{Int32}; node_aux::Ptr{Float64}
    rec_obj::Ptr{Int32}; rec_shape::Ptr{Int32}; rec::Ptr{Float64}
    det_count::Ptr{Int64}; det_offset::Ptr{Int64}; det_node::Ptr{Int32}; det_data::Ptr{Float64}
end
@assert sizeof(BmoShape) == 288 && sizeof(BmoObject) == 200 && sizeof(BmoSceneDesc) == 144 && sizeof(BmoTraceOpts) == 16

# ------------------------------------------------------------------------------------------------ the container
"""
    GPUSystem(inner::AbstractSystem; device = 0, max_beams = 0, segments = :all)

An [`AbstractSystem`](@ref) whose `solve_system!` runs on the MI355X engine.  `inner` stays the source of truth for
`objects(system)` and is what unsupported cases fall back to.  `max_beams > 0` stops a solve whose beam tree outgrows it (a
splitter facing a mirror; `solve_system!` itself recurses until memory runs out).
"""
mutable struct GPUSystem{S <: AbstractSystem} <: AbstractSystem
    inner::S
    device::Int32
    max_beams::Int32
    # :all  — every ray of every beam comes back (`bmo_result_view`: what solve_system! leaves behind, 2.3 GB over PCIe for config 2);
    # :last — only last(rays(beam)) of every beam, the beam tree and the detectors' data (`bmo_result_view_select`, 0.5 GB): each rebuilt
    #         beam then holds ONE ray, its last; length(rays(beam)) is NOT the reference's — for spot diagrams / PSFs of large bundles
    segments::Symbol
    # beams (or the beam group) solved before => bmo_trace_result* kept resident in HBM so that the next solve can retrace it
    solved::IdDict{Any, Ptr{Cvoid}}
    function GPUSystem(inner::S; device::Integer = 0, max_beams::Integer = 0, segments::Symbol = :all) where {S <: AbstractSystem}
        segments in (:all, :last) || throw(ArgumentError("segments must be :all or :last"))
        sys = new{S}(inner, Int32(device), Int32(max_beams), segments, IdDict{Any, Ptr{Cvoid}}())
        finalizer(release!, sys)
        return sys
    end
end
objects(s::GPUSystem) = objects(s.inner)
refractive_index(s::GPUSystem, λ::Real) = refractive_index(s.inner, λ)

"Frees the solutions kept for retracing (device memory of the segment logs)."
function release!(sys::GPUSystem)
    for (_, h) in sys.solved
        h != C_NULL && ccall((:bmo_result_free, LIBBMO), Cint, (Ptr{Cvoid},), h)
    end
    empty!(sys.solved)
    return nothing
end

struct BmoUnsupported <: Exception
    what::String
end
struct BmoError <: Exception
    code::Int
    msg::String
end
"""
The library this process loaded, checked once (the twin of `abi._check_source_hash` in beamletoptics.jl_amd/abi.py): its ABI version has to
be the one this file binds; the hashes of the sources and of the compiler flags it was built from (`bmo_source_hash`,
`bmo_build_flags_hash`: bit-level parity with `solve_system!` needs `-ffp-contract=off` & co.) are compared with `ENV["BMO_SOURCE_HASH"]` /
`ENV["BMO_FLAGS_HASH"]` when a deployment pins them, and kept in `LIB_INFO` for bug reports.
"""
const LIB_INFO = Ref{Union{Nothing, NamedTuple{(:abi, :source_hash, :flags_hash), Tuple{Int, String, String}}}}(nothing)
function check_library()
    LIB_INFO[] === nothing || return LIB_INFO[]
    abi = Int(ccall((:bmo_version, LIBBMO), Cint, ()))
    abi == BMO_ABI_VERSION || error("GPUSystem: $(LIBBMO) speaks ABI version $abi, this file binds version $(BMO_ABI_VERSION)")
    src = unsafe_string(ccall((:bmo_source_hash, LIBBMO), Cstring, ()))
    flg = unsafe_string(ccall((:bmo_build_flags_hash, LIBBMO), Cstring, ()))
    want_src, want_flg = get(ENV, "BMO_SOURCE_HASH", ""), get(ENV, "BMO_FLAGS_HASH", "")
    isempty(want_src) || want_src == src || error("GPUSystem: $(LIBBMO) was built from other sources (source hash $src, expected $want_src)")
    isempty(want_flg) || want_flg == flg || error("GPUSystem: $(LIBBMO) was built with other compiler flags (flags hash $flg, expected $want_flg)")
    LIB_INFO[] = (abi = abi, source_hash = src, flags_hash = flg)
    return LIB_INFO[]
end

"""
    check_elementary_functions(; n = 100_000)

The engine does not call a C library for `sin` / `cos` (`fresnel_coefficients`), `acos` (`angle3d`), `tan` and `atan(y, x)` (`gauss_parameters`): it
evaluates Base's own algorithms (base/special/trig.jl), restated in csrc/bmo_jlmath.hpp, so that a traced `E0` or `w0` carries the bits `solve_system!`
produces.  The restatement was written without a Julia at hand; this is the one-minute check a maintainer runs once per Julia version: every function on
`n` random arguments against Base, `==` on the bits.  Returns the number of mismatches per function (all zero = the engine's numerics contract holds here).
"""
function check_elementary_functions(; n::Int = 100_000)
    f(which, x, y = 0.0) = ccall((:bmo_jl_trig, LIBBMO), Cdouble, (Int32, Cdouble, Cdouble), Int32(which), Float64(x), Float64(y))
    same(a, b) = (isnan(a) && isnan(b)) || reinterpret(UInt64, a) == reinterpret(UInt64, b)
    xs = 14 .* rand(n) .- 7
    us = 2 .* rand(n) .- 1
    ws = randn(n) .* exp.(80 .* rand(n) .- 40)
    return (sin = count(x -> !same(f(0, x), sin(x)), xs), cos = count(x -> !same(f(1, x), cos(x)), xs),
            tan = count(x -> !same(f(2, x), tan(x)), xs ./ 4.5), acos = count(x -> !same(f(3, x), acos(x)), us),
            atan = count(x -> !same(f(4, x), atan(x)), ws), atan2 = count(x -> !same(f(5, x, 1.0), atan(1.0, x)), abs.(ws)))
end

function check(rc::Cint)
    rc == BMO_OK && return nothing
    msg = unsafe_string(ccall((:bmo_last_error, LIBBMO), Cstring, ()))
    rc == BMO_ERR_UNSUPPORTED && throw(BmoUnsupported(msg))
    throw(BmoError(rc, msg))
end

# ------------------------------------------------------------------------------------------------ scene flattening
# row-major NTuple{9} of a 3x3 (Julia matrices are column-major)
rowmajor(M) = ntuple(i -> Float64(M[(i - 1) ÷ 3 + 1, (i - 1) % 3 + 1]), 9)
tup3(v) = (Float64(v[1]), Float64(v[2]), Float64(v[3]))
pad8(v...) = ntuple(i -> i <= length(v) ? Float64(v[i]) : 0.0, 8)

mutable struct SceneTables
    slope::Vector{Float64}                         # per shape id: K of "sdf >= dist / K" (1 for exact sdfs, NaN: no bound => never culled)
    shapes::Vector{BmoShape}
    shape_refs::Vector{Any}                        # Julia shape of every shape id (Intersection.shape is rebuilt from it)
    shape_ids::IdDict{Any, Int32}
    bounds::Vector{Tuple{NTuple{3, Float64}, Float64}}   # uninflated world bounding sphere per shape id
    children::Vector{Int32}
    tris::Vector{Float64}
    coefs::Vector{Float64}
    media::Vector{Vector{Float64}}
    media_ids::IdDict{Any, Int32}
    objects::Vector{BmoObject}
    detectors::Vector{Any}
    lambdas::Vector{Float64}
end
SceneTables(λs) = SceneTables(Float64[], BmoShape[], Any[], IdDict{Any, Int32}(), Tuple{NTuple{3, Float64}, Float64}[], Int32[], Float64[], Float64[],
                              Vector{Float64}[], IdDict{Any, Int32}(), BmoObject[], Any[], λs)

# miss-cull inflation of bounding spheres (DESIGN.md "miss cull"; beamletoptics.jl_amd/system.py _BS_REL, _BS_ABS)
const BS_REL, BS_ABS = 1e-6, 1e-6

# local-frame bounding sphere (centre, radius) of every SDF kind — the twin of `_local_bound` in beamletoptics.jl_amd/shapes.py
local_bound(s::PlanoSurfaceSDF) = ((0.0, s.thickness / 2, 0.0), hypot(s.thickness / 2, s.diameter / 2))
local_bound(s::SphereSDF) = ((0.0, 0.0, 0.0), Float64(s.radius))
local_bound(s::ConcaveSphericalSurfaceSDF) = ((0.0, -s.sag / 2, 0.0), hypot(s.sag / 2, s.diameter / 2))
local_bound(s::ConvexSphericalSurfaceSDF) = ((0.0, s.sag / 2, 0.0), hypot(s.sag / 2, s.diameter / 2))
local_bound(s::BoxSDF) = ((0.0, 0.0, 0.0), Float64(norm(s.dimensions)))
local_bound(s::CylinderSDF) = ((0.0, 0.0, 0.0), hypot(s.radius, s.height))
local_bound(s::CutSphereSDF) = ((0.0, 0.0, 0.0), Float64(s.radius))
local_bound(s::RingSDF) = ((0.0, 0.0, 0.0), hypot(s.inner_radius + s.hwidth, s.hthickness))
local_bound(s::RightAnglePrismSDF) = ((0.0, 0.0, 0.0), Float64(norm(s.dimensions)))
function local_bound(s::AbstractAsphericalSurfaceSDF)
    edge = aspheric_equation(s.diameter / 2, 1 / s.radius, s.conic_constant, s.coefficients)
    lo, hi = extrema((0.0, edge, s.max_sag[1]))
    return ((0.0, (lo + hi) / 2, 0.0), hypot((hi - lo) / 2, s.diameter / 2))
end
function local_bound(s::ConvexCylinderSDF)   # extrusion along local x of a cut disk in (y, z)
    hc = sqrt(s.radius^2 - (s.diameter / 2)^2)
    return ((0.0, 0.0, (hc + s.radius) / 2), sqrt((s.height / 2)^2 + (s.diameter / 2)^2 + ((s.radius - hc) / 2)^2))
end
function local_bound(s::ConcaveCylinderSDF)
    sg = sag(abs(s.radius), s.diameter)
    return ((0.0, sg / 2 * sign(s.radius), 0.0), sqrt((s.height / 2)^2 + (sg / 2)^2 + (s.diameter / 2)^2))
end
function local_bound(s::AbstractAcylindricalSurfaceSDF)
    edge = aspheric_equation(s.diameter / 2, 1 / s.radius, s.conic_constant, s.coefficients)
    lo, hi = extrema((0.0, edge, s.max_sag[1]))
    return ((0.0, (lo + hi) / 2, 0.0), sqrt(((hi - lo) / 2)^2 + (s.diameter / 2)^2 + (s.height / 2)^2))
end
function world_bound(s::AbstractSDF)
    c, r = local_bound(s)
    return (tup3(position(s) + orientation(s) * Point3{Float64}(c...)), Float64(r))
end
"Conservative sphere around spheres (the twin of `_enclose` in shapes.py)."
function enclose(spheres)
    c, r = collect(spheres[1][1]), spheres[1][2]
    for (c2t, r2) in spheres[2:end]
        c2 = collect(c2t)
        d = norm(c2 - c)
        d + r2 <= r && continue
        if d + r <= r2
            c, r = c2, r2
            continue
        end
        nr = (d + r + r2) / 2
        c = c + (c2 - c) * ((nr - r) / d)
        r = nr
    end
    return (tup3(c), Float64(r))
end

# kind and parameter vector of every leaf SDF (include/bmo.h enum bmo_shape_kind lists the order of p[])
leaf_record(s::SphereSDF) = (K_SPHERE, pad8(s.radius))
leaf_record(s::PlanoSurfaceSDF) = (K_PLANO, pad8(s.thickness, s.diameter))
leaf_record(s::ConvexSphericalSurfaceSDF) = (K_CONVEX, pad8(s.radius, s.diameter, s.sag, s.height))
leaf_record(s::ConcaveSphericalSurfaceSDF) = (K_CONCAVE, pad8(s.radius, s.diameter, s.sag))
leaf_record(s::BoxSDF) = (K_BOX, pad8(s.dimensions...))
leaf_record(s::CylinderSDF) = (K_CYLINDER, pad8(s.radius, s.height))
leaf_record(s::CutSphereSDF) = (K_CUTSPHERE, pad8(s.radius, s.height, s.w))
leaf_record(s::RingSDF) = (K_RING, pad8(s.inner_radius, s.hwidth, s.hthickness))
leaf_record(s::RightAnglePrismSDF) = (K_PRISM, pad8(s.dimensions...))
leaf_record(s::ConvexAsphericalSurfaceSDF) = (K_ASPH_CONVEX, pad8(s.radius, s.conic_constant, s.diameter, s.max_sag[1]))
leaf_record(s::ConcaveAsphericalSurfaceSDF) = (K_ASPH_CONCAVE, pad8(s.radius, s.conic_constant, s.diameter, s.max_sag[1]))
leaf_record(s::ConvexCylinderSDF) = (K_CYL_CONVEX, pad8(s.radius, s.diameter, s.height))
leaf_record(s::ConcaveCylinderSDF) = (K_CYL_CONCAVE, pad8(s.radius, s.diameter, s.height))
leaf_record(s::AconvexCylinderSDF) = (K_ACYL_CONVEX, pad8(s.radius, s.diameter, s.height, s.conic_constant, s.max_sag[1]))
leaf_record(s::AconcaveCylinderSDF) = (K_ACYL_CONCAVE, pad8(s.radius, s.diameter, s.height, s.conic_constant, s.max_sag[1]))
leaf_record(s::AbstractShape) = throw(BmoUnsupported("shape type $(typeof(s))"))
has_coefficients(s) = s isa AbstractAsphericalSurfaceSDF || s isa AbstractAcylindricalSurfaceSDF

"""
K >= 1 with sdf(p) >= dist(p, solid) / K outside the solid, for the first-order aspheric / acylindric distance estimates
(AsphericalLensSDF.jl:186-307): K = sqrt(1 + G^2), G = 1.05 x the largest |z'(r)| over 4097 samples of [0, d/2]; NaN when the conic
term leaves its domain inside the aperture (then the shape is never culled).  Derivation: beamletoptics.jl_amd/shapes.py slope_bound.
"""
function slope_bound(s)
    g_max = 0.0
    for r in range(0.0, s.diameter / 2; length = 4097)
        g = gradient_aspheric_equation(r, 1 / s.radius, s.conic_constant, s.coefficients)[1]
        z = aspheric_equation(r, 1 / s.radius, s.conic_constant, s.coefficients)
        (isnan(g) || isnan(z) || isinf(g)) && return NaN
        g_max = max(g_max, abs(g))
    end
    g_max *= 1.05
    return sqrt(1 + g_max^2)
end

function push_shape!(tb::SceneTables, s, rec_fields, bound, K::Float64 = 1.0)
    kind, child_begin, child_count, tri_begin, tri_count, flags, p = rec_fields
    c, r = bound
    push!(tb.slope, K)
    bs_radius = isnan(K) ? -1.0 : r + K * (r * BS_REL + BS_ABS)   # outside it sdf >= r*1e-6 + 1 um: the reference can only return `nothing`
    dir = orientation(s)                           # SphereSDF: identity (SphericalLensSDF.jl:82-84)
    tdir = s isa AbstractSDF ? transposed_orientation(s) : transpose(dir)   # the stored copy(dir') of AbstractSDF.jl:20-27
    push!(tb.shapes, BmoShape(kind, child_begin, child_count, tri_begin, tri_count, flags, tup3(position(s)), rowmajor(dir), rowmajor(tdir), p,
                              c, bs_radius))
    push!(tb.shape_refs, s)
    push!(tb.bounds, bound)
    id = Int32(length(tb.shapes) - 1)
    tb.shape_ids[s] = id
    return id
end

"Adds `s` (and, for unions and meniscus lenses, its children) to the shape table; returns its 0-based shape id."
function add_shape!(tb::SceneTables, s::AbstractShape)
    haskey(tb.shape_ids, s) && return tb.shape_ids[s]
    return _add_shape!(tb, s)
end
function _add_shape!(tb::SceneTables, m::Mesh)
    V, F = vertices(m), faces(m)
    tri_begin = Int32(length(tb.tris) ÷ 9)
    for f in axes(F, 1), k in 1:3, x in 1:3     # 9 doubles per triangle: V1 V2 V3 in world space, as intersect3d reads them (Mesh.jl:252)
        push!(tb.tris, V[F[f, k], x])
    end
    lo, hi = vec(minimum(V, dims = 1)), vec(maximum(V, dims = 1))
    c = (lo + hi) / 2
    r = maximum(norm(V[i, :] - c) for i in axes(V, 1))
    return push_shape!(tb, m, (K_MESH, Int32(0), Int32(0), tri_begin, Int32(size(F, 1)), Int32(0), pad8()), (tup3(c), Float64(r)))
end
function _add_shape!(tb::SceneTables, u::UnionSDF)
    ids = Int32[add_shape!(tb, c) for c in u.sdfs]   # children first, in tuple order (UnionSDF.jl:53-56 folds left to right)
    any(tb.shapes[i + 1].kind in (K_UNION, K_MESH) for i in ids) && throw(BmoUnsupported("nested UnionSDF"))
    child_begin = Int32(length(tb.children))
    append!(tb.children, ids)
    flags = any(tb.shapes[i + 1].flags & SHAPE_FLAG_INEXACT != 0 for i in ids) ? SHAPE_FLAG_INEXACT : Int32(0)
    bound = enclose([tb.bounds[i + 1] for i in ids])
    K = maximum(tb.slope[i + 1] for i in ids)       # min over children: the weakest bound holds (NaN propagates)
    return push_shape!(tb, u, (K_UNION, child_begin, Int32(length(ids)), Int32(0), Int32(0), flags, pad8()), bound, K)
end
function _add_shape!(tb::SceneTables, ml::MeniscusLensSDF)
    # children = {convex, cylinder, concave}; their pos / dir are expressed in the meniscus frame (MeniscusLensSDF.jl:42-46)
    ids = Int32[add_shape!(tb, ml.convex), add_shape!(tb, ml.cylinder), add_shape!(tb, ml.concave)]
    child_begin = Int32(length(tb.children))
    append!(tb.children, ids)
    flags = any(tb.shapes[i + 1].flags & SHAPE_FLAG_INEXACT != 0 for i in ids) ? SHAPE_FLAG_INEXACT : Int32(0)
    c_local, r = enclose([tb.bounds[ids[1] + 1], tb.bounds[ids[2] + 1]])   # max(min(convex, cylinder), -concave) lies inside convex U cylinder
    c = tup3(position(ml) + orientation(ml) * Point3{Float64}(c_local...))
    K = max(tb.slope[ids[1] + 1], tb.slope[ids[2] + 1])
    return push_shape!(tb, ml, (K_MENISCUS, child_begin, Int32(3), Int32(0), Int32(0), flags, pad8()), (c, r), K)
end
function _add_shape!(tb::SceneTables, s::AbstractSDF)
    kind, p = leaf_record(s)
    child_begin, child_count, flags = Int32(0), Int32(0), Int32(0)
    if has_coefficients(s)       # even aspheric coefficients live in the pooled `coefs` table; first-order distance estimate
        child_begin, child_count, flags = Int32(length(tb.coefs)), Int32(length(s.coefficients)), SHAPE_FLAG_INEXACT
        append!(tb.coefs, Float64.(s.coefficients))
    end
    K = has_coefficients(s) ? slope_bound(s) : 1.0
    return push_shape!(tb, s, (kind, child_begin, child_count, Int32(0), Int32(0), flags, p), world_bound(s), K)
end
_add_shape!(::SceneTables, s::AbstractShape) = throw(BmoUnsupported("shape type $(typeof(s))"))

"Row of the n(λ) table for a refractive index functor, evaluated on the host (functors cannot cross a C ABI)."
function medium!(tb::SceneTables, n)
    haskey(tb.media_ids, n) && return tb.media_ids[n]
    push!(tb.media, Float64[n(λ) for λ in tb.lambdas])   # a DiscreteRefractiveIndex KeyError surfaces here, before tracing
    return tb.media_ids[n] = Int32(length(tb.media) - 1)
end

const NO3 = (Int32(-1), Int32(-1), Int32(-1))
const NO_JONES = ntuple(_ -> 0.0, 18)
bmo_object(kind; shape = NO3, medium = (Int32(-1), Int32(-1)), detector = Int32(-1), R = 0.0, T = 0.0, cutoff = 0.0, jones = NO_JONES) =
    BmoObject(kind, shape, medium, detector, Int32(0), Float64(R), Float64(T), Float64(cutoff), jones)
one_shape(id) = (id, Int32(-1), Int32(-1))
function detector_slot!(tb, o)
    push!(tb.detectors, o)
    return Int32(length(tb.detectors) - 1)
end

# one bmo_object per leaf object (object ids = position in `Leaves` order)
object_record(tb, o::AbstractReflectiveOptic) = bmo_object(O_MIRROR; shape = one_shape(add_shape!(tb, shape(o))))
object_record(tb, o::Union{Lens, Prism}) =
    bmo_object(O_REFRACTIVE; shape = one_shape(add_shape!(tb, shape(o))), medium = (medium!(tb, refractive_index(o)), Int32(-1)))
object_record(tb, o::DoubletLens) =
    bmo_object(O_DOUBLET; shape = (add_shape!(tb, shape(o.front)), add_shape!(tb, shape(o.back)), Int32(-1)),
               medium = (medium!(tb, refractive_index(o.front)), medium!(tb, refractive_index(o.back))))
object_record(tb, o::ThinBeamsplitter) =
    bmo_object(O_THIN_BS; shape = one_shape(add_shape!(tb, shape(o))), R = reflectance(o), T = transmittance(o))
object_record(tb, o::AbstractPlateBeamsplitter) =   # shape = {substrate, coating} (PlateBeamsplitter.jl:44)
    bmo_object(O_PLATE_BS; shape = (add_shape!(tb, shape(substrate(o))), add_shape!(tb, shape(coating(o))), Int32(-1)),
               medium = (medium!(tb, refractive_index(substrate(o))), Int32(-1)), R = reflectance(coating(o)), T = transmittance(coating(o)))
object_record(tb, o::CubeBeamsplitter) =            # shape = {front, back, coating} (CubeBeamsplitter.jl:30)
    bmo_object(O_CUBE_BS; shape = (add_shape!(tb, shape(o.front)), add_shape!(tb, shape(o.back)), add_shape!(tb, shape(o.coating))),
               medium = (medium!(tb, refractive_index(o.front)), medium!(tb, refractive_index(o.back))),
               R = reflectance(o.coating), T = transmittance(o.coating))
object_record(tb, o::Spotdetector) = bmo_object(O_SPOT; shape = one_shape(add_shape!(tb, shape(o))), detector = detector_slot!(tb, o))
object_record(tb, o::PSFDetector) = bmo_object(O_PSF; shape = one_shape(add_shape!(tb, shape(o))), detector = detector_slot!(tb, o))
object_record(tb, o::Photodetector) = bmo_object(O_PHOTODETECTOR; shape = one_shape(add_shape!(tb, shape(o))), detector = detector_slot!(tb, o))
object_record(tb, o::IntersectableObject) = bmo_object(O_INTERSECTABLE; shape = one_shape(add_shape!(tb, shape(o))))
object_record(tb, o::NonInteractableObject) = bmo_object(O_NONINTERACTABLE; shape = one_shape(add_shape!(tb, shape(o))))
function object_record(tb, o::PolarizationFilter)
    J = o.JMat.data                                  # GlobalJonesBasis, 3x3 complex; row-major (re, im) pairs
    jones = ntuple(i -> (z = ComplexF64(J[(i - 1) ÷ 6 + 1, ((i - 1) ÷ 2) % 3 + 1]); isodd(i) ? real(z) : imag(z)), 18)
    return bmo_object(O_POLARIZER; shape = one_shape(add_shape!(tb, shape(o))), cutoff = o.cutoff, jones = jones)
end
object_record(tb, o::AbstractObject) = throw(BmoUnsupported("object type $(typeof(o))"))

"Flat tables of `leaves` at the wavelengths `λs` (sorted, distinct) + the descriptor that points into them (keep `tb` alive)."
function flatten_scene(leaves, λs::Vector{Float64})
    tb = SceneTables(λs)
    for o in leaves
        push!(tb.objects, object_record(tb, o))
    end
    ntab = isempty(tb.media) ? ones(max(1, length(λs))) : reduce(vcat, tb.media)   # [n_media][n_lambda], row-major
    return tb, ntab
end
"The descriptor over `tb`'s arrays; the counts are the true ones, empty tables get a one-element stand-in so that no pointer is NULL."
function scene_desc(tb::SceneTables, ntab::Vector{Float64})
    counts = (length(tb.children), length(tb.tris) ÷ 9, length(tb.coefs))
    isempty(tb.children) && push!(tb.children, Int32(0))
    isempty(tb.tris) && append!(tb.tris, zeros(9))
    isempty(tb.coefs) && push!(tb.coefs, 0.0)
    return BmoSceneDesc(BMO_ABI_VERSION, length(tb.objects), length(tb.shapes), counts[1], counts[2], length(tb.media), length(tb.lambdas),
                        length(tb.detectors),
                        pointer(tb.objects), pointer(tb.shapes), pointer(tb.children), pointer(tb.tris), pointer(ntab), pointer(tb.lambdas), pointer(tb.coefs),
                        1e-9, 1e-10, 1.0, 1e-9, 1e-9, 1e-8,      # AbstractSDF.jl:1-3, Mesh.jl:203, AbstractSDF.jl:83
                        1000, counts[3])                          # AbstractSDF.jl:105,135
end

# ------------------------------------------------------------------------------------------------ first rays -> planes
beam_kind(::Beam{T, Ray{T}}) where {T} = BEAM_RAY
beam_kind(::Beam{T, PolarizedRay{T}}) where {T} = BEAM_POLARIZED
beam_kind(::GaussianBeamlet) = BEAM_GAUSSIAN
beam_kind(b::AbstractBeam) = throw(BmoUnsupported("beam type $(typeof(b))"))
head_wavelength(b::Beam) = wavelength(first(rays(b)))
head_wavelength(g::GaussianBeamlet) = wavelength(g)

"""
SoA planes of the root heads (include/bmo.h bmo_ray_batch): the FIRST ray of every root beam, dir as stored (the Ray constructor
normalised it, Rays.jl:32-42).  Returns (planes [n_planes*n], lambda_idx [n]).
"""
function pack_first_rays(roots::Vector, kind::Int32, λs::Vector{Float64})
    n = length(roots)
    np = PLANES_IN[kind + 1]
    P = zeros(Float64, n * np)
    li = Vector{Int32}(undef, n)
    put!(plane, j, v) = (P[(plane - 1) * n + j] = v)
    for (j, b) in enumerate(roots)
        λ = head_wavelength(b)
        li[j] = Int32(searchsortedfirst(λs, λ) - 1)
        if kind == BEAM_GAUSSIAN
            for (base, sub) in ((0, b.chief), (6, b.waist), (12, b.divergence))
                r = first(rays(sub))
                for x in 1:3
                    put!(base + x, j, position(r)[x])
                    put!(base + 3 + x, j, direction(r)[x])
                end
            end
            put!(19, j, λ); put!(20, j, refractive_index(first(rays(b.chief)))); put!(21, j, beam_waist(b))
            put!(22, j, real(electric_field(b))); put!(23, j, imag(electric_field(b)))
        else
            r = first(rays(b))
            for x in 1:3
                put!(x, j, position(r)[x])
                put!(3 + x, j, direction(r)[x])
            end
            put!(7, j, λ); put!(8, j, refractive_index(r))
            if kind == BEAM_POLARIZED
                E = polarization(r)
                for x in 1:3
                    put!(8 + 2x - 1, j, real(E[x]))
                    put!(8 + 2x, j, imag(E[x]))
                end
            end
        end
    end
    return P, li
end

# ------------------------------------------------------------------------------------------------ result tables -> beams
struct HostView                                   # Julia arrays over the library-owned tables (valid until bmo_result_free)
    v::BmoResultView
    root::Vector{Int32}; parent::Vector{Int32}; first_rec::Vector{Int32}; nseg::Vector{Int32}; status::Vector{Int32}
    aux::Matrix{Float64}                          # [4, n_nodes]
    rec_obj::Vector{Int32}; rec_shape::Vector{Int32}
    rec::Matrix{Float64}                          # [n_records, rec_planes]: plane p is column p
    det_count::Vector{Int64}; det_offset::Vector{Int64}
    det::Matrix{Float64}                          # [9, total hits]
    det_node::Vector{Int32}                       # node (result order) that made each hit
end
function HostView(v::BmoResultView)
    nn, nr, nd = Int(v.n_nodes), Int(v.n_records), Int(v.n_detectors)
    w(p, dims...) = prod(dims) == 0 ? Array{eltype(p)}(undef, dims...) : unsafe_wrap(Array, p, dims)
    cnt = w(v.det_count, nd)
    tot = Int(sum(cnt))
    return HostView(v, w(v.node_root, nn), w(v.node_parent, nn), w(v.node_first_rec, nn), w(v.node_nseg, nn), w(v.node_status, nn),
                    w(v.node_aux, 4, nn), w(v.rec_obj, nr), w(v.rec_shape, nr), w(v.rec, nr, Int(v.rec_planes)), cnt, w(v.det_offset, nd),
                    w(v.det_data, 9, tot), w(v.det_node, tot))
end

# ray k (1-based record index r) of one (sub-)beam; `base` = first plane of that ray inside the record (0, 11, 22)
function make_ray(::Type{T}, hv::HostView, r::Int, base::Int, λ, leaves, shape_refs; E0 = nothing) where {T}
    R = hv.rec
    pos = Point3{T}(R[r, base + 1], R[r, base + 2], R[r, base + 3])
    dir = Point3{T}(R[r, base + 4], R[r, base + 5], R[r, base + 6])
    t = R[r, base + 8]
    isect = nothing
    if isfinite(t)                                 # t = +Inf <=> intersection === nothing
        o, s = hv.rec_obj[r], hv.rec_shape[r]
        isect = Intersection{T}(o >= 0 ? leaves[o + 1] : nothing, s >= 0 ? shape_refs[s + 1] : nothing, T(t),
                                Point3{T}(R[r, base + 9], R[r, base + 10], R[r, base + 11]))
    end
    n = T(R[r, base + 7])
    E0 === nothing && return Ray{T}(pos, dir, isect, T(λ), n)
    return PolarizedRay{T}(pos, dir, isect, T(λ), n, E0)   # the inner constructor re-checks E0 ⟂ dir (PolarizedRays.jl:54-56)
end

"""
Rebuilds the beam trees from the result tables.  Nodes arrive in reference order (bundle order x BFS order, parents before
children, transmitted child before reflected child), so one forward pass links everything.  Root nodes are the caller's own beam
objects (mutated in place, like `solve_system!` does); child beams are created here.
"""
function rebuild_beams!(roots::Vector, kind::Int32, hv::HostView, leaves, shape_refs; last_only::Bool = false)
    nn = length(hv.root)
    nodes = Vector{Any}(undef, nn)
    for i in 1:nn
        par = hv.parent[i]
        # (a BMO_VIEW_LAST_SEGMENT view holds ONE record per beam, node_first_rec[i] = i, while node_nseg stays the true ray count)
        f, n = Int(hv.first_rec[i]), last_only ? 1 : Int(hv.nseg[i])
        recs = (f + 1):(f + n)
        if kind == BEAM_GAUSSIAN
            w0, E0, λ = hv.aux[1, i], complex(hv.aux[2, i], hv.aux[3, i]), hv.aux[4, i]
            T = typeof(λ)
            sub(base) = Beam{T, Ray{T}}([make_ray(T, hv, r, base, λ, leaves, shape_refs) for r in recs], nothing, Vector{Beam{T, Ray{T}}}())
            if par < 0
                g = roots[hv.root[i] + 1]
                g.chief.rays, g.waist.rays, g.divergence.rays = sub(0).rays, sub(11).rays, sub(22).rays
                empty!(g.children)                  # children are re-attached below (retrace keeps their identity in the engine)
            else
                g = GaussianBeamlet(sub(0), sub(11), sub(22), T(λ), T(w0), Complex{T}(E0))
                p = nodes[par + 1]
                parent!(g, p)                       # also links chief.parent (Gaussian.jl:113-117)
                push!(children(p), g)
            end
            nodes[i] = g
        else
            λ = hv.aux[1, i]
            T = typeof(λ)
            mk(r) = kind == BEAM_POLARIZED ?
                    make_ray(T, hv, r, 0, λ, leaves, shape_refs;
                             E0 = Point3{Complex{T}}(complex(hv.rec[r, 12], hv.rec[r, 13]), complex(hv.rec[r, 14], hv.rec[r, 15]), complex(hv.rec[r, 16], hv.rec[r, 17]))) :
                    make_ray(T, hv, r, 0, λ, leaves, shape_refs)
            if par < 0
                b = roots[hv.root[i] + 1]
                b.rays = [mk(r) for r in recs]
                empty!(b.children)
            else
                p = nodes[par + 1]
                b = typeof(p)([mk(r) for r in recs], nothing, typeof(p.children)())
                parent!(b, p)
                push!(children(p), b)
            end
            nodes[i] = b
        end
    end
    return nodes
end

"Exceptions the reference throws while tracing come back as status bits: re-raise for the first offending beam."
function raise_status(hv::HostView)
    for i in eachindex(hv.status)
        s = hv.status[i]
        s & NODE_ERR_UNIT != 0 && throw(ArgumentError("dir and normal must have a unit length of 1"))             # OpticUtils.jl:33-35
        s & NODE_ERR_ORTHO != 0 && throw(ErrorException("Ray dir. and E0 must be orthogonal."))                    # PolarizedRays.jl:54-56
    end
    return nothing
end

"Appends the detector records of this solve in the reference's push! order (detectors are not reset, as in the reference)."
function push_detector_data!(tb::SceneTables, hv::HostView, res::Ptr{Cvoid}; defer = nothing, nodes = nothing, opl0 = nothing)
    # `defer` (an IdDict beam => records, with `nodes` = the beams of the result in node order): Spot / PSF records are not pushed but
    # left with the beam that made them, for a caller that merges several solves into the reference's order (trace_open_leaves!);
    # `opl0[root + 1]`: optical path a continued Ray / PolarizedRay beam had behind it, added to its PSF records
    keep!(h, rec, det) = push!(get!(() -> Any[], defer, nodes[hv.det_node[h] + 1]), (det, rec))
    for (slot0, det) in enumerate(tb.detectors)
        cnt, off = Int(hv.det_count[slot0]), Int(hv.det_offset[slot0])
        if det isa Spotdetector
            T = typeof(det.hw)
            for h in (off + 1):(off + cnt)
                rec = Point2{T}(hv.det[1, h], hv.det[2, h])
                defer === nothing ? push!(det, rec) : keep!(h, rec, det)                                            # Spotdetector.jl:50-61
            end
        elseif det isa PSFDetector
            T = eltype(vertices(shape(det)))
            for h in (off + 1):(off + cnt)
                D = view(hv.det, :, h)
                opl = opl0 === nothing ? D[7] : D[7] + opl0[hv.root[hv.det_node[h] + 1] + 1]
                rec = PSFData{T}(Point3{T}(D[1], D[2], D[3]), Point3{T}(D[4], D[5], D[6]), opl, D[8], D[9])
                defer === nothing ? push!(det, rec) : keep!(h, rec, det)                                            # PSFDetector.jl:77-89
            end
        elseif det isa Photodetector && cnt > 0
            # Photodetector.jl:69-107: field[i, j] += electric_field(gauss, r, z) * sqrt(proj), evaluated on the GPU from the segment
            # log still resident in `res`.  Julia's column-major field[i, j] is the ABI's element [i + nx*j].
            pos = collect(Float64, position(det)); ori = collect(rowmajor(orientation(det)))
            xs, ys = collect(Float64, det.x), collect(Float64, det.y)
            F = det.field isa Matrix{ComplexF64} ? det.field : ComplexF64.(det.field)
            GC.@preserve pos ori xs ys F check(ccall((:bmo_photodetector_field, LIBBMO), Cint,
                (Ptr{Cvoid}, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int32, Int32, Ptr{ComplexF64}, Ptr{Float64}),
                res, slot0 - 1, pos, ori, xs, ys, length(xs), length(ys), F, C_NULL))
            F === det.field || (det.field .= F)
        end
    end
    return nothing
end

# ------------------------------------------------------------------------------------------------ solve_system!
"""
    solve_system!(system::GPUSystem, bg::AbstractBeamGroup; r_max = 100, retrace = true)
    solve_system!(system::GPUSystem, beam::AbstractBeam;    r_max = 100, retrace = true)

src/System.jl:444-468 on the GPU.  Fresh beams are traced (`bmo_trace`); beams this system solved before are retraced
(`bmo_retrace`, src/System.jl:188-255 and :326-428) against the solution kept in HBM.  `retrace = false` on solved beams only
traces leaves on whose last ray has no intersection (src/System.jl:449-458, :470-475): `trace_open_leaves!` below, on the GPU.
"""
function solve_system!(sys::GPUSystem, bg::AbstractBeamGroup; r_max::Int = 100, retrace::Bool = true)
    return gpu_solve!(sys, bg, collect(beams(bg)); r_max, retrace)
end
function solve_system!(sys::GPUSystem, beam::AbstractBeam; r_max::Int = 100, retrace::Bool = true)
    return gpu_solve!(sys, beam, Any[beam]; r_max, retrace)
end

function gpu_solve!(sys::GPUSystem, key, roots::Vector; r_max::Int, retrace::Bool)
    isempty(roots) && return nothing
    check_library()
    prev = get(sys.solved, key, C_NULL)
    fallback() = (forget!(sys, key); solve_system!(sys.inner, key; r_max, retrace))
    if prev != C_NULL && !retrace   # solved beams, not re-walked: only the open leaves are traced on (System.jl:449-458, :470-475)
        sys.segments === :all || return fallback()   # (their open last rays are all a :last solution holds, but not the lengths in front of them)
        try
            return trace_open_leaves!(sys, key, roots; r_max)
        catch e
            e isa BmoUnsupported || rethrow()
            return fallback()
        end
    end
    local tb, ntab, kind
    try
        kind = beam_kind(first(roots))
        all(b -> beam_kind(b) == kind, roots) || throw(BmoUnsupported("mixed beam kinds in one group"))
        λs = sort(unique(Float64[head_wavelength(b) for b in roots]))
        leaves = collect(objects(sys.inner))                                     # Leaves order = object ids (System.jl:21)
        tb, ntab = flatten_scene(leaves, λs)
        planes, li = pack_first_rays(roots, kind, λs)
        scene, res, view = Ref{Ptr{Cvoid}}(C_NULL), Ref{Ptr{Cvoid}}(C_NULL), Ref{BmoResultView}()
        GC.@preserve tb ntab planes li begin
            desc = scene_desc(tb, ntab)
            check(ccall((:bmo_scene_create, LIBBMO), Cint, (Ref{BmoSceneDesc}, Ref{Ptr{Cvoid}}), desc, scene))
            try
                batch = BmoRayBatch(length(roots), kind, PLANES_IN[kind + 1], pointer(planes), pointer(li))
                opts = BmoTraceOpts(r_max, sys.device, 1, sys.max_beams)
                if prev == C_NULL   # fresh beams: retrace_system! is a no-op on them (System.jl:197-206)
                    check(ccall((:bmo_trace, LIBBMO), Cint, (Ptr{Cvoid}, Ref{BmoRayBatch}, Ref{BmoTraceOpts}, Ref{Ptr{Cvoid}}), scene[], batch, opts, res))
                else                # the batch supplies the current root heads: edits of the first ray / E0 / w0 are honoured
                    check(ccall((:bmo_retrace, LIBBMO), Cint, (Ptr{Cvoid}, Ref{BmoRayBatch}, Ptr{Cvoid}, Ref{BmoTraceOpts}, Ref{Ptr{Cvoid}}),
                                scene[], batch, prev, opts, res))
                end
            finally
                ccall((:bmo_scene_destroy, LIBBMO), Cint, (Ptr{Cvoid},), scene[])
            end
        end
        try
            if sys.segments === :all
                check(ccall((:bmo_result_view, LIBBMO), Cint, (Ptr{Cvoid}, Ref{BmoResultView}), res[], view))
            else   # beam tree, last(rays(beam)) of every beam and the detectors' data: no segment log over PCIe
                check(ccall((:bmo_result_view_select, LIBBMO), Cint, (Ptr{Cvoid}, UInt32, Ref{BmoResultView}), res[], VIEW_HITS | VIEW_LAST_SEGMENT, view))
            end
        catch
            ccall((:bmo_result_free, LIBBMO), Cint, (Ptr{Cvoid},), res[])   # the solution stays in HBM otherwise (ADVICE r02)
            rethrow()
        end
        hv = HostView(view[])
        # (NODE_RETRACE_STALE is a note since round 4: kept stale children and stale-tail splits are what retrace_system! does, DESIGN.md §6 f1)
        try
            raise_status(hv)
            rebuild_beams!(roots, kind, hv, leaves, tb.shape_refs; last_only = sys.segments === :last)
            push_detector_data!(tb, hv, res[])
        catch e
            ccall((:bmo_result_free, LIBBMO), Cint, (Ptr{Cvoid},), res[])
            forget!(sys, key)
            # beams (and possibly some detectors) have been updated already: falling back to the wrapped System now would append the
            # detector data a second time, so an "unsupported" raised this late is an error, not a fallback (ADVICE r02)
            e isa BmoUnsupported && error("GPUSystem: " * sprint(showerror, e) * " (raised after the beams were rebuilt; no CPU fallback at this point)")
            rethrow()
        end
        forget!(sys, key)
        sys.solved[key] = res[]        # stays in HBM for the next (re)trace; freed by forget! / release! / the finalizer
    catch e
        e isa BmoUnsupported || rethrow()
        return fallback()
    end
    return nothing
end
# ------------------------------------------------------------------------------------------------ retrace = false on solved beams
nrays(b::Beam) = length(rays(b))
nrays(g::GaussianBeamlet) = length(rays(g.chief))
ray_length(r) = intersection(r) === nothing ? Inf : length(intersection(r))

"""
`solve_system!(...; retrace = false)` on beams solved before (src/System.jl:449-458, solve_leaf! :470-475): nothing is re-walked; every beam
of the trees whose LAST ray has no intersection is traced on from that ray, without a hint, up to `r_max` rays in the beam; everything else
stays as it is.  The open last rays are traced as one fresh batch per distinct remaining length and spliced back (the twin of
`_trace_open_leaves` in beamletoptics.jl_amd/system.py, where every step of this is tested on the GPU).  A batch applies ONE ray limit —
what the beams it continues have left — while the reference gives every child born in the continuation the full `r_max`: children that
ran into the batch's limit are open beams themselves and are continued by the next pass.  Detector records are appended at the end, in the
order the reference's loop makes them: root by root (System.jl:463-468), breadth first (:446-458).
"""
function trace_open_leaves!(sys::GPUSystem, key, roots::Vector; r_max::Int)
    open = Any[]
    for root in roots
        queue = Any[root]
        while !isempty(queue)
            b = popfirst!(queue)
            append!(queue, children(b))
            _last_beam_intersection(b) === nothing && nrays(b) < r_max && push!(open, b)
        end
    end
    pending = IdDict{Any, Vector{Any}}()
    while !isempty(open)
        open = continue_open_beams!(sys, open, pending; r_max)
    end
    for root in roots   # one breadth-first walk per root: root 1's whole tree before root 2
        queue = Any[root]
        while !isempty(queue)
            b = popfirst!(queue)
            append!(queue, children(b))
            for (det, rec) in get(pending, b, ())
                push!(det, rec)
            end
        end
    end
    forget!(sys, key)   # the resident solution no longer describes these beams: the next solve traces them afresh
    return nothing
end

"One pass of `trace_open_leaves!`; returns the beams born in it that ran into the pass's ray limit below `r_max`."
function continue_open_beams!(sys::GPUSystem, open_beams::Vector, pending; r_max::Int)
    again = Any[]
    by_left = Dict{Int, Vector{Any}}()
    for b in open_beams
        push!(get!(() -> Any[], by_left, r_max - nrays(b) + 1), b)
    end
    for left in sort!(collect(keys(by_left)))
        group = by_left[left]
        kind = beam_kind(first(group))
        gaussian = kind == BEAM_GAUSSIAN
        # heads: fresh beams that hold the open last ray(s) only
        heads = Any[]
        for b in group
            if gaussian
                T = typeof(wavelength(b))
                sub(part) = Beam{T, Ray{T}}([last(rays(part))], nothing, Vector{Beam{T, Ray{T}}}())
                push!(heads, GaussianBeamlet(sub(b.chief), sub(b.waist), sub(b.divergence), wavelength(b), beam_waist(b), electric_field(b)))
            else
                push!(heads, typeof(b)([last(rays(b))], nothing, typeof(b.children)()))
            end
        end
        λs = sort(unique(Float64[head_wavelength(h) for h in heads]))
        leaves = collect(objects(sys.inner))
        tb, ntab = flatten_scene(leaves, λs)
        planes, li = pack_first_rays(heads, kind, λs)
        n = length(heads)
        np = PLANES_IN[kind + 1]
        if gaussian
            # the lengths the beamlets have accumulated up to their open rays (include/bmo.h "31 planes": lenA, lenB, l0, oplC, oplW, oplD),
            # folded exactly as a solve that had never stopped would have folded them (bmo_lane.hpp GaussIn)
            acc = zeros(Float64, 6 * n)
            for (i, b) in enumerate(group)
                l0 = b.parent === nothing ? 0.0 : Float64(length(b.parent))
                len_a, len_b = 0.0, l0
                opl_c = b.chief.parent === nothing ? 0.0 : Float64(optical_path_length(b.chief.parent))
                for r in rays(b.chief)[1:(end - 1)]
                    len_a += ray_length(r); len_b += ray_length(r); opl_c += ray_length(r) * refractive_index(r)
                end
                opl_w = sum(Float64[ray_length(r) * refractive_index(r) for r in rays(b.waist)[1:(end - 1)]]; init = 0.0)
                opl_d = sum(Float64[ray_length(r) * refractive_index(r) for r in rays(b.divergence)[1:(end - 1)]]; init = 0.0)
                for (q, v) in enumerate((len_a, len_b, l0, opl_c, opl_w, opl_d))
                    acc[(q - 1) * n + i] = v
                end
            end
            planes = vcat(planes, acc)
            np += 6
        end
        scene, res, view = Ref{Ptr{Cvoid}}(C_NULL), Ref{Ptr{Cvoid}}(C_NULL), Ref{BmoResultView}()
        GC.@preserve tb ntab planes li begin
            desc = scene_desc(tb, ntab)
            check(ccall((:bmo_scene_create, LIBBMO), Cint, (Ref{BmoSceneDesc}, Ref{Ptr{Cvoid}}), desc, scene))
            try
                batch = BmoRayBatch(n, kind, np, pointer(planes), pointer(li))
                opts = BmoTraceOpts(left, sys.device, 1, sys.max_beams)
                check(ccall((:bmo_trace, LIBBMO), Cint, (Ptr{Cvoid}, Ref{BmoRayBatch}, Ref{BmoTraceOpts}, Ref{Ptr{Cvoid}}), scene[], batch, opts, res))
            finally
                ccall((:bmo_scene_destroy, LIBBMO), Cint, (Ptr{Cvoid},), scene[])
            end
        end
        try
            if gaussian && any(o -> o isa Photodetector, leaves)
                # the field of a beamlet on a Photodetector is a function of ALL its rays (point_on_beam, length, optical_path_length:
                # Beam.jl:125-205); the continuation holds only those from the open ray on, the ones in front of it go along as a prefix
                starts, cols, opl_par = Int32[0], Float64[], Float64[]
                nseg = 0
                for b in group
                    for k in 1:(nrays(b) - 1)
                        for part in (b.chief, b.waist, b.divergence)
                            r = rays(part)[k]
                            append!(cols, (position(r)..., direction(r)..., refractive_index(r), ray_length(r)))
                        end
                        nseg += 1
                    end
                    push!(starts, Int32(nseg))
                    push!(opl_par, b.chief.parent === nothing ? 0.0 : Float64(optical_path_length(b.chief.parent)))
                end
                segs = nseg == 0 ? Float64[] : vec(permutedims(reshape(cols, 24, nseg)))   # [24][total]: plane-major
                GC.@preserve starts segs opl_par check(ccall((:bmo_result_set_gauss_prefix, LIBBMO), Cint,
                    (Ptr{Cvoid}, Int64, Ptr{Int32}, Ptr{Float64}, Ptr{Float64}), res[], n, starts, segs, opl_par))
            end
            check(ccall((:bmo_result_view, LIBBMO), Cint, (Ptr{Cvoid}, Ref{BmoResultView}), res[], view))
            hv = HostView(view[])
            raise_status(hv)
            nodes = rebuild_beams!(heads, kind, hv, leaves, tb.shape_refs)
            # optical path each continued Ray / PolarizedRay beam had behind it (parents included) up to the start of its open ray: what PSF
            # records of the continuation lack (a continued beamlet brings its path lengths along in the batch)
            opl0 = gaussian ? nothing : Float64[(b.parent === nothing ? 0.0 : Float64(optical_path_length(b.parent))) +
                                                sum(Float64[ray_length(r) * refractive_index(r) for r in rays(b)[1:(end - 1)]]; init = 0.0) for b in group]
            push_detector_data!(tb, hv, res[]; defer = pending, nodes = nodes, opl0 = opl0)
            for (b, h) in zip(group, heads)   # splice: the open ray (now with its intersection, if any) and what followed it
                if gaussian
                    for (part, hp) in ((b.chief, h.chief), (b.waist, h.waist), (b.divergence, h.divergence))
                        pop!(rays(part)); append!(rays(part), rays(hp))
                    end
                    for c in children(h)
                        parent!(c, b)               # also links chief.parent (Gaussian.jl:113-117)
                    end
                else
                    pop!(rays(b)); append!(rays(b), rays(h))
                    for c in children(h)
                        parent!(c, b)
                    end
                end
                b.children = children(h)
                haskey(pending, h) && append!(get!(() -> Any[], pending, b), pop!(pending, h))
            end
            if left < r_max
                for i in (length(group) + 1):length(nodes)   # beams born in this pass
                    nd = nodes[i]
                    (hv.status[i] & NODE_RMAX) != 0 && _last_beam_intersection(nd) === nothing && nrays(nd) < r_max && push!(again, nd)
                end
            end
        finally
            ccall((:bmo_result_free, LIBBMO), Cint, (Ptr{Cvoid},), res[])
        end
    end
    return again
end

function forget!(sys::GPUSystem, key)
    h = pop!(sys.solved, key, C_NULL)
    h != C_NULL && ccall((:bmo_result_free, LIBBMO), Cint, (Ptr{Cvoid},), h)
    return nothing
end

# ------------------------------------------------------------------------------------------------ detector read-out on the GPU
"""
    gpu_intensity(psf::PSFDetector; n = 100, crop_factor = 1, x0_shift = 0, z0_shift = 0, device = 0)

`intensity(psf)` (PSFDetector.jl:190-237) with the n² x hits coherent sum evaluated by `bmo_psf_intensity`; the sample grid is
computed exactly as the reference does (`calc_local_lims`, two `LinRange`s).  Returns `(xs, zs, I)` like the reference.
"""
function gpu_intensity(psf::PSFDetector{T}; n::Int = 100, crop_factor::Real = 1, center::Symbol = :centroid, x_min = Inf, x_max = Inf, z_min = Inf,
                       z_max = Inf, x0_shift::Real = 0, z0_shift::Real = 0, device::Integer = 0) where {T}
    _x_min, _x_max, _z_min, _z_max = calc_local_lims(psf; crop_factor = crop_factor, center = center)
    (x_min != Inf && x_max != Inf) && ((_x_min, _x_max) = (x_min, x_max))
    (z_min != Inf && z_max != Inf) && ((_z_min, _z_max) = (z_min, z_max))
    xs = collect(Float64, LinRange(_x_min, _x_max, n) .+ x0_shift)
    zs = collect(Float64, LinRange(_z_min, _z_max, n) .+ z0_shift)
    hits = reinterpret(Float64, psf.data)                                      # 9 doubles per PSFData: hit, dir, opl, proj, k
    R = orientation(psf)
    origin, e1, e2 = collect(Float64, position(psf)), collect(Float64, R[:, 1]), collect(Float64, R[:, 3])
    I = Matrix{Float64}(undef, n, n)
    GC.@preserve hits xs zs origin e1 e2 I check(ccall((:bmo_psf_intensity, LIBBMO), Cint,
        (Ptr{Float64}, Int64, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int32, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
        pointer(hits), length(psf.data), 0, origin, e1, e2, xs, zs, n, device, I, C_NULL, C_NULL))
    return xs, zs, I
end
