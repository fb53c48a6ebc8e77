// bmo_lane.hpp — per-lane arithmetic of the trace kernel (gfx950 HIP engine).
//
// One lane = one beam node advancing by ONE bounce: tracing_step! (System.jl:57-110) against
// the scene staged in LDS, followed by interact3d.  Everything here is FP64, written so that
// the operation order equals the reference's expressions (compiled with -ffp-contract=off);
// the dual-number type mirrors ForwardDiff's rules so normals are reproduced including the
// NaN -> central-difference fallback (AbstractSDF.jl:90-95).
//
// Code-shape rules (they matter on CDNA: no device calls, small I-cache footprint):
//   * every routine is force-inlined and every heavy routine has exactly ONE call site:
//     the SDF leaf switch is instantiated three times in the whole kernel (value evaluation,
//     central-difference stencil, dual-number gradient); sphere tracing (classification, inside
//     march and outside march of AbstractSDF.jl:102-181) is one state-machine loop around one SDF
//     evaluation; trace_one/trace_all (System.jl:57-85) is one loop around one shape intersection;
//     interact3d decodes the object kind into an action and then runs one copy of each action.
//
// What is NOT in the reference and only here:
//   * the "miss cull": a ray whose line misses a shape's (inflated) bounding sphere, or which is
//     outside that sphere and receding, can never satisfy the reference's hit test
//     (dist < eps_ray, AbstractSDF.jl:118) because every SDF here is >= the distance to the
//     bounding sphere; the reference would burn its 1000 iterations and return `nothing`.
//     The cull returns that `nothing` at once.  Hits replay the reference's iteration sequence
//     exactly.  See DESIGN.md "miss cull" for the argument and tests/test_cull.py for the check.
//
// The file also compiles with a host compiler (BMO_HD empty) for the sanitizer / lane-emulator
// build used by tests (tests/emu); that build is test-only and never shipped or dispatched to.
#pragma once
#include <math.h>
#include <stdint.h>

#include "../../include/bmo.h"
#include "bmo_jlmath.hpp"

#if defined(__HIPCC__)
#define BMO_HD __host__ __device__ __forceinline__
#define BMO_NOUNROLL _Pragma("unroll 1")
#else
#define BMO_HD inline __attribute__((always_inline))
#define BMO_NOUNROLL
#endif

#if !defined(BMO_DUAL_PASSES)
#define BMO_DUAL_PASSES 1  /* evaluations per dual-number gradient: 1 (duals of three partials), 2 (two + one), 3 (one each) — see normal_of */
#endif

namespace bmo {

#if defined(BMO_EMU_STATS)  // host-only instrumentation of the test emulator (tools/emu_stats.py)
inline long g_emu_sdf_any = 0, g_emu_sdf_leaf = 0, g_emu_normal = 0, g_emu_normal_fd = 0;
#endif

// Scalar scene access (device builds).  The scene tables (objects, shapes, children, triangles, n(lambda), candidate table) are
// read-only for the life of a kernel and sit in global memory behind CONSTANT-address-space pointers (CShape, CObject, CDouble, CInt,
// CCand below): a read whose index is the same in all lanes of the wave then compiles to a scalar load (s_load, served by the scalar
// cache) and its value lives in scalar registers — the FP64 vector instructions take it as an operand directly, no vector register and
// no LDS round trip per table entry.  BMO_UNIFORM(i) (readfirstlane) marks such an index.  The wave therefore works on ONE shape at a
// time: the slots of trace_all are wave-uniform by construction (the candidate table is walked by a scalar loop); where lanes can
// disagree — the hinted shape of trace_one, the shape that won the step, the arg-min child of a union whose normal is needed — the
// lanes are taken through a "waterfall": one pass per distinct id among the active lanes (one pass in a coherent wave).  A read with a
// per-lane index through the same pointers is an ordinary vector load (interact3d's look-up of the object that was hit).
// Host builds (emulator, sanitizers): plain pointers, BMO_UNIFORM is the identity and every waterfall has one lane and one pass.
#if defined(__HIP_DEVICE_COMPILE__)
#define BMO_UNIFORM(i) __builtin_amdgcn_readfirstlane(i)
#define BMO_KONST __attribute__((address_space(4)))
// Waterfall test "is this lane's id the one the wave works on now": the scalar id goes through an empty asm so that the optimiser does
// not learn `u == id` inside the branch and substitute the per-lane id back for the scalar one (its reads would turn into vector loads).
__device__ __forceinline__ bool bmo_same_id(int32_t u, int32_t id) {
    asm volatile("" : "+s"(u));
    return u == id;
}
#else
#define BMO_UNIFORM(i) (i)
#define BMO_KONST
inline bool bmo_same_id(int32_t u, int32_t id) { return u == id; }
#endif
// BMO_WAVE_ALL(p) / BMO_WAVE_ANY(p): p holds in every / some lane of the wave that executes this statement (host: the lane itself)
#if defined(__HIP_DEVICE_COMPILE__)
#define BMO_WAVE_ALL(p) (__all(p) != 0)
#define BMO_WAVE_ANY(p) (__any(p) != 0)
#else
#define BMO_WAVE_ALL(p) (p)
#define BMO_WAVE_ANY(p) (p)
#endif
typedef const BMO_KONST bmo_shape CShape;
typedef const BMO_KONST bmo_object CObject;
typedef const BMO_KONST double CDouble;
typedef const BMO_KONST int32_t CInt;


struct SceneView {
    CObject* objects;
    CShape* shapes;
    CInt* children;
    CDouble* tris;
    CDouble* n_table;
    CDouble* coefs;
    const BMO_KONST struct Cand* cands;  // candidate table (fill_candidates)
    int32_t n_objects, n_lambda, n_cands;
    double eps_srf, eps_ray, eps_ins, mt_keps, mt_leps, grad_h;
    int32_t march_iters;
};

struct d3 {
    double x, y, z;
};
struct Hit {
    double t;
    d3 n;
    int32_t obj, shape;  // shape < 0: no intersection
};

BMO_HD double kinf() { return __builtin_huge_val(); }
BMO_HD bool sgn(double x) { return __builtin_signbit(x); }
BMO_HD bool isnan_(double x) { return x != x; }
BMO_HD Hit no_hit() {
    Hit h;
    h.t = kinf();
    h.n = {0, 0, 0};
    h.obj = -1;
    h.shape = -1;
    return h;
}

// ------------------------------------------------------------------ scalar rules (Julia Base)
// max / min of Base for Float64: NaN if either is NaN, -0.0 < +0.0.  The rule as written:
BMO_HD double jmax_rule(double x, double y) {
    if (isnan_(x) || isnan_(y)) return x + y;
    return ((y > x) || (sgn(y) < sgn(x))) ? y : x;
}
BMO_HD double jmin_rule(double x, double y) {
    if (isnan_(x) || isnan_(y)) return x + y;
    return ((y < x) || (sgn(y) > sgn(x))) ? y : x;
}
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BMO_RULE_MINMAX)
// Device form: v_max_f64 / v_min_f64 order the zeros the same way (+0 > -0) and return the other operand for a NaN, so one NaN select
// (x + y is NaN exactly when x or y is) on top gives the rule without control flow: 5 instructions, where the rule form compiles to
// compare-and-branch sequences with execution-mask bookkeeping.  The instruction is written out because __builtin_fmax first
// canonicalises both operands (two more v_max_f64).  bmo_selftest compares this form with the rule, bit for bit on the special values,
// on the device.  -DBMO_RULE_MINMAX builds the rule form for A/B runs.  (Without the NaN select — NOT the rule — C2 would be 1.2 % faster:
// profiles/r03_ab_inwave.txt.)
BMO_HD double jmax(double x, double y) {
    double m;
    asm("v_max_f64 %0, %1, %2" : "=v"(m) : "v"(x), "v"(y));
    return __builtin_isunordered(x, y) ? x + y : m;
}
BMO_HD double jmin(double x, double y) {
    double m;
    asm("v_min_f64 %0, %1, %2" : "=v"(m) : "v"(x), "v"(y));
    return __builtin_isunordered(x, y) ? x + y : m;
}
#else
BMO_HD double jmax(double x, double y) { return jmax_rule(x, y); }
BMO_HD double jmin(double x, double y) { return jmin_rule(x, y); }
#endif
BMO_HD double jabs(double x) { return fabs(x); }
BMO_HD double jsqrt(double x) { return sqrt(x); }
BMO_HD double val(double x) { return x; }

// ------------------------------------------------------------------ ForwardDiff.Dual with N partials
// (N = 3 is the reference's gradient of three; the kernels evaluate the partials in passes of fewer — normal_of — because the register need
//  of a leaf grows with N: 27 vector registers for the value alone, 86 with three partials.  Every rule below treats the partials one by one,
//  so a pass over a subset computes exactly the numbers the full gradient holds for that subset.)
template <int N>
struct DualN {
    double v;
    double p[N];
};
using Dual = DualN<3>;
#define BMO_DN template <int N> BMO_HD
BMO_DN double val(const DualN<N>& x) { return x.v; }
BMO_DN DualN<N> operator+(const DualN<N>& x, const DualN<N>& y) {
    DualN<N> r;
    r.v = x.v + y.v;
    for (int q = 0; q < N; ++q) r.p[q] = x.p[q] + y.p[q];
    return r;
}
BMO_DN DualN<N> operator-(const DualN<N>& x, const DualN<N>& y) {
    DualN<N> r;
    r.v = x.v - y.v;
    for (int q = 0; q < N; ++q) r.p[q] = x.p[q] - y.p[q];
    return r;
}
BMO_DN DualN<N> operator+(const DualN<N>& x, double c) {
    DualN<N> r = x;
    r.v = x.v + c;
    return r;
}
BMO_DN DualN<N> operator+(double c, const DualN<N>& x) {
    DualN<N> r = x;
    r.v = c + x.v;
    return r;
}
BMO_DN DualN<N> operator-(const DualN<N>& x, double c) {
    DualN<N> r = x;
    r.v = x.v - c;
    return r;
}
BMO_DN DualN<N> operator-(double c, const DualN<N>& x) {
    DualN<N> r;
    r.v = c - x.v;
    for (int q = 0; q < N; ++q) r.p[q] = -x.p[q];
    return r;
}
BMO_DN DualN<N> operator-(const DualN<N>& x) {
    DualN<N> r;
    r.v = -x.v;
    for (int q = 0; q < N; ++q) r.p[q] = -x.p[q];
    return r;
}
BMO_DN DualN<N> operator*(const DualN<N>& x, const DualN<N>& y) {
    DualN<N> r;
    r.v = x.v * y.v;
    for (int q = 0; q < N; ++q) r.p[q] = (x.p[q] * y.v) + (y.p[q] * x.v);
    return r;
}
BMO_DN DualN<N> operator*(const DualN<N>& x, double c) {
    DualN<N> r;
    r.v = x.v * c;
    for (int q = 0; q < N; ++q) r.p[q] = x.p[q] * c;
    return r;
}
BMO_DN DualN<N> operator*(double c, const DualN<N>& x) {
    DualN<N> r;
    r.v = c * x.v;
    for (int q = 0; q < N; ++q) r.p[q] = x.p[q] * c;
    return r;
}
BMO_DN DualN<N> operator/(const DualN<N>& x, double c) {
    DualN<N> r;
    r.v = x.v / c;
    for (int q = 0; q < N; ++q) r.p[q] = x.p[q] / c;
    return r;
}
// The dual-number rules below are the ones under which EVERY reference KAT holds at its original assertion (oracle/RULE_TABLE.md, round 4):
//   sqrt: deriv * partials as ForwardDiff writes it — at 0 the derivative is Inf, a zero partial becomes 0 * Inf = NaN (rounds 1 - 3 kept zero
//         partials zero);
//   max / min: SELECTION — the winner's value and partials and nothing else (rounds 1 - 3: DiffRules' product form dvx * px + dvy * py, in
//         which a NaN partial of the LOSING operand poisons the result), ties go to the SECOND argument (`x > y ? x : y`, `x < y ? x : y`), a
//         Real operand counts as a Dual with zero partials.
// What they do together: `norm(max.(q, 0))` of a zero vector is NaN; where it sits in the winning operand of a shape's outer max (the flat
// faces of boxes and prisms: runtests.jl:2629-2630 needs the central-difference normal there) the normal falls back to the numeric gradient
// (AbstractSDF.jl:90-95), where it sits in the losing operand (every hit on a concave spherical surface: runtests.jl:2755-2761 needs the dual
// normal there) it is dropped, and at a concave apex, where the operands tie, the clean second one wins (runtests.jl:1309-1314).
BMO_DN DualN<N> jsqrt(const DualN<N>& x) {
    const double s = sqrt(x.v);
    const double d = 1.0 / (2.0 * s);
    DualN<N> r;
    r.v = s;
    for (int q = 0; q < N; ++q) r.p[q] = x.p[q] * d;
    return r;
}
BMO_DN DualN<N> jabs(const DualN<N>& x) {
    const bool neg = sgn(x.v);
    DualN<N> r;
    r.v = neg ? -x.v : x.v;
    for (int q = 0; q < N; ++q) r.p[q] = neg ? -x.p[q] : x.p[q];
    return r;
}
BMO_DN DualN<N> jmax(const DualN<N>& x, const DualN<N>& y) {
    const bool xw = x.v > y.v;
    DualN<N> r;
    r.v = xw ? x.v : y.v;
    for (int q = 0; q < N; ++q) r.p[q] = xw ? x.p[q] : y.p[q];
    return r;
}
BMO_DN DualN<N> jmin(const DualN<N>& x, const DualN<N>& y) {
    const bool xw = x.v < y.v;
    DualN<N> r;
    r.v = xw ? x.v : y.v;
    for (int q = 0; q < N; ++q) r.p[q] = xw ? x.p[q] : y.p[q];
    return r;
}
BMO_DN DualN<N> jmax(const DualN<N>& x, double y) {
    const bool xw = x.v > y;
    DualN<N> r;
    r.v = xw ? x.v : y;
    for (int q = 0; q < N; ++q) r.p[q] = xw ? x.p[q] : 0.0;
    return r;
}
BMO_DN DualN<N> jmin(const DualN<N>& x, double y) {
    const bool xw = x.v < y;
    DualN<N> r;
    r.v = xw ? x.v : y;
    for (int q = 0; q < N; ++q) r.p[q] = xw ? x.p[q] : 0.0;
    return r;
}

template <class T>
struct v3 {
    T x, y, z;
};

// Keeps a leaf's arithmetic INSIDE the branch of its kind: the kind of a shape is wave-uniform, the dispatch over it a chain of scalar
// branches, and whatever the optimiser may evaluate speculatively in front of such a branch (a norm with its square root, hoisted out of
// the branch that needs it) is evaluated for every leaf of every kind.  An empty asm on the local point pins everything that depends on it
// behind the branch it stands in (device only; no instruction is emitted).
BMO_HD void pin(double& x) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BMO_NO_PIN)
    asm volatile("" : "+v"(x));
#else
    (void)x;
#endif
}
template <int N>
BMO_HD void pin(DualN<N>& x) {
    pin(x.v);  // (the value alone: pinning the partials too costs 1 % and 20 B of scratch — measured)
}
template <class T>
BMO_HD void pin3(v3<T>& p) {
    pin(p.x);  // (pinning x alone — every square root of the leaves depends on it — measures the same)
    pin(p.y);
    pin(p.z);
}
BMO_HD double dot3(const d3& a, const d3& b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
BMO_HD double norm3(const d3& a) { return sqrt(dot3(a, a)); }
BMO_HD d3 sub3(const d3& a, const d3& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
BMO_HD d3 neg3(const d3& a) { return {-a.x, -a.y, -a.z}; }
BMO_HD d3 cross3(const d3& a, const d3& b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
BMO_HD d3 axpy3(const d3& p, double s, const d3& d) { return {p.x + s * d.x, p.y + s * d.y, p.z + s * d.z}; }
BMO_HD d3 normalize_div(const d3& a) {  // normalize(::Point3) = a ./ norm(a)
    double n = norm3(a);
    return {a.x / n, a.y / n, a.z / n};
}
BMO_HD d3 normalize_inv(const d3& a) {  // normalize(::Vector) = a * inv(norm(a))
    double inv = 1.0 / norm3(a);
    return {a.x * inv, a.y * inv, a.z * inv};
}
template <class T>
BMO_HD T norm2(const T& x, const T& y) {
    return jsqrt(x * x + y * y);
}
template <class T>
BMO_HD T norm3t(const T& x, const T& y, const T& z) {
    return jsqrt((x * x + y * y) + z * z);
}

// isapprox for reals (Base): default rtol = sqrt(eps) iff atol == 0
BMO_HD bool isapprox(double x, double y, double atol) {
    double rtol = atol > 0.0 ? 0.0 : 1.4901161193847656e-08;
    if (x == y) return true;
    if (isnan_(x) || isnan_(y) || fabs(x) == kinf() || fabs(y) == kinf()) return false;
    return fabs(x - y) <= fmax(atol, rtol * fmax(fabs(x), fabs(y)));
}

// ------------------------------------------------------------------ SDF evaluation
// _world_to_sdf (AbstractSDF.jl:35-40): rows of the stored transposed orientation, left fold
template <class T>
BMO_HD v3<T> to_local(CShape& s, const v3<T>& pt) {
    T dx = pt.x - s.pos[0], dy = pt.y - s.pos[1], dz = pt.z - s.pos[2];
    CDouble* m = s.tdir;
    return {(m[0] * dx + m[1] * dy) + m[2] * dz, (m[3] * dx + m[4] * dy) + m[5] * dz, (m[6] * dx + m[7] * dy) + m[8] * dz};
}

// min(maximum(d), 0) + norm(max.(d, 0))   [== norm(max.(d,0)) + min(max(d1,d2),0): IEEE + commutes]
template <class T>
BMO_HD T slab2(const T& dx, const T& dy) {
    T mx = jmax(dx, dy);
    T ax = jmax(dx, 0.0), ay = jmax(dy, 0.0);
    return jmin(mx, 0.0) + norm2(ax, ay);
}


// ------------------------------------------------------------------ aspheres / acylinders
// AsphericalLensSDF.jl:133-307, AcylindricalSDF.jl.  T = double (aspheres: never differentiated, :5) or Dual (acylinders
// use the default normal_fd).  Extra ForwardDiff rules: Dual/Dual, Real/Dual, literal powers, run-time integer powers, clamp.
BMO_DN DualN<N> operator/(const DualN<N>& x, const DualN<N>& y) {
    const double ia = 1.0 / y.v, fb = -(x.v / (y.v * y.v));
    DualN<N> r;
    r.v = x.v / y.v;
    for (int q = 0; q < N; ++q) r.p[q] = (x.p[q] * ia) + (y.p[q] * fb);
    return r;
}
BMO_DN DualN<N> operator/(double x, const DualN<N>& y) {
    const double divv = x / y.v, f = -(divv / y.v);
    DualN<N> r;
    r.v = divv;
    for (int q = 0; q < N; ++q) r.p[q] = y.p[q] * f;
    return r;
}
BMO_HD bool lt(double a, double b) { return a < b; }
BMO_DN bool lt(const DualN<N>& a, double b) { return a.v < b; }
BMO_DN bool lt(double a, const DualN<N>& b) { return a < b.v; }
BMO_DN bool lt(const DualN<N>& a, const DualN<N>& b) { return a.v < b.v; }
BMO_HD double lit2(double x) { return x * x; }
BMO_HD double lit3(double x) { return x * x * x; }
BMO_DN DualN<N> lit2(const DualN<N>& x) {
    const double d = 2 * x.v;
    DualN<N> r;
    r.v = x.v * x.v;
    for (int q = 0; q < N; ++q) r.p[q] = x.p[q] * d;
    return r;
}
BMO_DN DualN<N> lit3(const DualN<N>& x) {
    const double d = 3 * (x.v * x.v);
    DualN<N> r;
    r.v = x.v * x.v * x.v;
    for (int q = 0; q < N; ++q) r.p[q] = x.p[q] * d;
    return r;
}
BMO_HD double ipow(double x, int n) {  // Base.power_by_squaring (small exponents multiply out like literal_pow)
    if (n == 0) return 1.0;
    if (n == 1) return x;
    if (n == 2) return x * x;
    if (n == 3) return x * x * x;
    double r = 1.0;
    while (n > 0) {
        if (n & 1) r *= x;
        x *= x;
        n >>= 1;
    }
    return r;
}
BMO_DN DualN<N> ipow(const DualN<N>& x, int n) {  // Dual(v^n, (partials * n) * v^(n-1)); zero partials short-cut (partial by partial: 0 * n * f = 0 for the finite f of n >= 1)
    const double ev = ipow(x.v, n);
    const double f = n == 0 ? 0.0 : ipow(x.v, n - 1);
    DualN<N> r;
    r.v = ev;
    for (int q = 0; q < N; ++q) r.p[q] = (n == 0 || x.p[q] == 0) ? 0.0 : (x.p[q] * n) * f;
    return r;
}
BMO_HD double knan() { return kinf() - kinf(); }
BMO_HD double mknan(double) { return knan(); }
BMO_DN DualN<N> mknan(const DualN<N>&) {
    DualN<N> r;
    r.v = knan();
    for (int q = 0; q < N; ++q) r.p[q] = knan();
    return r;
}
BMO_HD double jsign(double x) { return x > 0 ? 1.0 : (x < 0 ? -1.0 : x); }
BMO_HD double clamp01(double x) { return x > 1.0 ? 1.0 : (x < 0.0 ? 0.0 : x); }
BMO_DN DualN<N> clamp01(const DualN<N>& x) {
    DualN<N> r = x;
    if (x.v > 1.0 || x.v < 0.0) {
        r.v = x.v > 1.0 ? 1.0 : 0.0;
        for (int q = 0; q < N; ++q) r.p[q] = 0.0;
    }
    return r;
}
// aspheric_equation :133-141 and the first component of gradient_aspheric_equation :147-156, one pass over the coefficients
template <class T>
BMO_HD void asph_eval(const T& r, double c, double k, CDouble* a, int na, T& z, T& g) {
    const T r2 = lit2(r);
    const T sa1 = 1 - (1 + k) * (c * c) * r2;
    const double Ri = 1 / c;
    const T sa2 = 1 - lit2(r) * (1 + k) / (Ri * Ri);
    T sum_a = r2 * 0.0, sum_r = r * 0.0;
    BMO_NOUNROLL
    for (int i = 1; i <= na; ++i) {
        const T ta = a[i - 1] * ipow(r2, i);
        const T tr = (2 * i * a[i - 1]) * ipow(r, 2 * (i - 1) + 1);
        sum_a = i == 1 ? ta : sum_a + ta;
        sum_r = i == 1 ? tr : sum_r + tr;
    }
    z = lt(sa1, 0.0) ? mknan(r) : c * r2 / (1 + jsqrt(sa1)) + sum_a;
    if (lt(sa2, 0.0)) {
        g = mknan(r);
    } else {
        const T sq = jsqrt(sa2);
        const T gr = 2 * r / (Ri * (sq + 1)) + lit3(r) * (1 + k) / ((Ri * Ri * Ri) * sq * lit2(sq + 1));
        g = -sum_r - gr;
    }
}
template <class T>
BMO_HD T norm_g(const T& g1) { return jsqrt(lit2(g1) + 1.0 * 1.0); }
template <class T>
BMO_HD T sd_line_segment(const T& px, const T& py, double ax, double ay, double bx, double by) {  // :164-169
    const T pax = px - ax, pay = py - ay;
    const double bax = bx - ax, bay = by - ay;
    const T h = clamp01((pax * bax + pay * bay) / (bax * bax + bay * bay));
    const T ex = pax - h * bax, ey = pay - h * bay;
    return jsqrt(ex * ex + ey * ey);
}
// convex_/concave_aspheric_surface_distance :186-307
template <class T>
BMO_HD T asph_distance(bool convex, const T& r, const T& z, double c, double k, double d, CDouble* a, int na, double max_sag) {
    const T r2 = lit2(r);
    const double r2_bound = (d / 2) * (d / 2);
    T zv, g;
    double zb, gb;
    asph_eval<T>(r, c, k, a, na, zv, g);
    asph_eval<double>(d / 2, c, k, a, na, zb, gb);
    const T e = r - jsign(val(r)) * d / 2;
    const double ngb = norm_g<double>(gb);
    if (convex) {
        if (isnan_(val(zv)) || isnan_(val(g)) || lt(r2_bound, r2)) {
            T db;
            if (lt(z, zb)) db = jsqrt(lit2(e) + lit2(z - zb));
            else if (lt(zb, z) && lt(z, 0.0)) db = jsqrt(lit2(e));
            else if (lt(0.0, z) && (jsign(c) == 1 && zb < 0)) db = jsqrt(lit2(e) + lit2(z));
            else db = jsqrt(lit2(e) + lit2(z - zb));
            return db / ngb;
        }
        const T da = jabs(z - zv) / norm_g<T>(g);
        if (jsign(c) == 1 && zb < 0) {
            const T s1 = sd_line_segment<T>(r, z, d / 2, zb, d / 2, max_sag) / ngb;
            const T s2 = sd_line_segment<T>(r, z, d / 2, max_sag, -d / 2, max_sag) / ngb;
            const T s3 = sd_line_segment<T>(r, z, -d / 2, max_sag, -d / 2, zb) / ngb;
            const T m = jmin(jmin(jmin(da, s1), s2), s3);
            return (lt(zv, z) && lt(z, max_sag)) ? -m : m;
        }
        const T sdl = sd_line_segment<T>(r, z, d / 2, zb, -d / 2, zb) / ngb;
        if (lt(jsign(c) * zv, jsign(c) * z) && lt(jsign(c) * z, jsign(c) * zb)) return -jmin(sdl, da);
        return jmin(sdl, da);
    }
    if (isnan_(val(zv)) || isnan_(val(g))) {
        T db;
        if (lt(z, 0.0)) db = jsqrt(lit2(e) + lit2(z));
        else if (lt(0.0, z) && lt(z, zb)) db = jsqrt(lit2(e));
        else db = jsqrt(lit2(e) + lit2(z - zb));
        return db / ngb;
    }
    const T da = jabs(z - zv) / norm_g<T>(g);
    if (max_sag > 0 && zb < 0) {
        const T sdl = sd_line_segment<T>(r, z, d / 2, zb, -d / 2, zb) / ngb;
        if (lt(r2_bound, r2)) return sdl;
        if (lt(zb, z) && lt(z, zv)) return -jmin(da, sdl);
        if (zb > 0 && (lt(0.0, z) && lt(z, zv))) return -jmin(da, sdl);
        return jmin(da, sdl);
    }
    const T s1 = sd_line_segment<T>(r, z, d / 2, zb, d / 2, 0.0) / ngb;
    const T s2 = sd_line_segment<T>(r, z, d / 2, 0.0, -d / 2, 0.0) / ngb;
    const T s3 = sd_line_segment<T>(r, z, -d / 2, 0.0, -d / 2, zb) / ngb;
    if (lt(r2_bound, r2)) return jmin(jmin(s1, s2), s3);
    const T m = jmin(jmin(jmin(da, s1), s2), s3);
    if (zb < 0 && (lt(zv, z) && lt(z, 0.0))) return -m;
    if (zb > 0 && (lt(0.0, z) && lt(z, zv))) return -m;
    return m;
}
// aspheres are never differentiated (normal3d = numeric_gradient, AsphericalLensSDF.jl:5): the Dual overload returns NaN,
// which sends normal_any straight to the same central-difference stencil.
BMO_HD double asph_leaf(CShape& s, CDouble* coefs, double r, double y) {
    return asph_distance<double>(s.kind == BMO_SHAPE_ASPH_CONVEX, r, y, 1 / s.p[0], s.p[1], s.p[2], coefs + s.child_begin, s.child_count, s.p[3]);
}
BMO_DN DualN<N> asph_leaf(CShape&, CDouble*, const DualN<N>& r, const DualN<N>&) { return mknan(r); }

// leaf SDFs; `pt` is in the parent's frame (world, or the meniscus frame)
// EXT ("extended shapes" level of the kernel): 0 = the spherical / primitive leaves only; 1 = + MeniscusLensSDF; 2 = + the aspheric
// and cylinder-lens leaves.  A scene runs the kernels of the lowest level that covers its shapes (less code, fewer registers: the
// meniscus fold keeps a second dual number and a second frame alive around the leaf, the aspheric leaves are the largest by far).
template <class T, int EXT>
BMO_HD T sdf_leaf(CShape& s, const int kind, const v3<T>& pt, CDouble* coefs) {
    // every table read of this leaf in one batch (the reads are LDS round trips of ~100 cycles each: issued one by one at their
    // points of use they, not the arithmetic, set the pace of the march); `kind` comes from the caller, who has read it already
    const double P0 = s.p[0], P1 = s.p[1], P2 = s.p[2], P3 = s.p[3];
    v3<T> p = to_local(s, pt);
    // leaf class of the kinds 0 .. 15, four bits each (1 plano / cylinder / ring, 2 convex / cut sphere, 3 concave, 4 box / prism, 5 sphere /
    // point): two scalar instructions in front of a dense dispatch instead of a tree of compares over the sparse kind values
    // (round 4: config 2 - 4.3 %, config 5 - 4.7 %)
    const unsigned cls = (unsigned)((0x0005041214032150ull >> (4 * (kind & 15))) & 15ull) * (kind < 16 ? 1u : 0u);
    if (cls == 1) {  // PLANO | CYLINDER | RING
        // SphericalLensSDF.jl:60-65, PrimitiveSDF.jl:71-76, :151-166
        pin3(p);
        double ra, ha, off, sub;
        if (kind == BMO_SHAPE_PLANO) {
            ra = P1 / 2;
            ha = P0 / 2;
            off = P0 / 2;
            sub = 0.0;
        } else if (kind == BMO_SHAPE_CYLINDER) {
            ra = P0;
            ha = P1;
            off = 0.0;
            sub = 0.0;
        } else {
            ra = P1;
            ha = P2;
            off = 0.0;
            sub = P0;
        }
        T r = norm2(p.x, p.z);
        if (kind == BMO_SHAPE_RING) r = r - sub;
        T h = kind == BMO_SHAPE_PLANO ? p.y - off : p.y;
        return slab2(jabs(r) - ra, jabs(h) - ha);
    }
    if (cls == 2) {  // CONVEX | CUTSPHERE  // SphericalLensSDF.jl:219-232, PrimitiveSDF.jl:112-124
        pin3(p);
        double radius = P0, w, height;
        T q1 = norm2(p.x, p.z), q2;
        if (kind == BMO_SHAPE_CONVEX) {
            w = P1 / 2;
            height = P3;
            q2 = -p.y + radius;
        } else {
            height = P1;
            w = P2;
            q2 = p.y;
        }
        double a = val(q1), b = val(q2);
        double sv = jmax((height - radius) * (a * a) + (w * w) * (height + radius - 2 * b), height * a - w * b);
        if (sv < 0) return norm2(q1, q2) - radius;
        if (a < w) return height - q2;
        return norm2(q1 - w, q2 - height);
    }
    if (cls == 3) {  // CONCAVE  // SphericalLensSDF.jl:159-170
        pin3(p);
        double radius = P0, dia = P1, sag = P2;
        T x0 = p.x + 0.0, z0 = p.z + 0.0;
        T y1 = p.y + sag / 2;
        T r = norm2(x0, z0);
        T sdf1 = slab2(jabs(r) - dia / 2, jabs(y1) - sag / 2);
        T y2 = p.y + radius;
        T sdf2 = norm3t(x0, y2, z0) - radius;
        return jmax(sdf1, -sdf2);
    }
    if (cls == 4) {  // BOX | PRISM  // PrimitiveSDF.jl:41-46, :204-210
        pin3(p);
        T qx = jabs(p.x) - P0, qy = jabs(p.y) - P1, qz = jabs(p.z) - P2;
        T box = norm3t(jmax(qx, 0.0), jmax(qy, 0.0), jmax(qz, 0.0)) + jmin(jmax(qx, jmax(qy, qz)), 0.0);
        if (kind == BMO_SHAPE_BOX) return box;
        T pln = (p.x + p.y) / 1.4142135623730951;  // sqrt(2)
        return jmax(box, pln);
    }
    if constexpr (EXT >= 2) {
        if (kind == BMO_SHAPE_ASPH_CONVEX || kind == BMO_SHAPE_ASPH_CONCAVE) {  // AsphericalLensSDF.jl:309-349 (op_revolve_z)
            T r = norm2(p.x, p.z) - 0.0;
            return asph_leaf(s, coefs, r, p.y);
        }
    }
    if constexpr (EXT >= 2) {
        if (kind == BMO_SHAPE_ACYL_CONVEX || kind == BMO_SHAPE_ACYL_CONCAVE) {  // AcylindricalSDF.jl:55-74, :122-141
            const double height = P2;
            T d2 = asph_distance<T>(kind == BMO_SHAPE_ACYL_CONVEX, p.z, p.y, 1 / P0, P3, P1, coefs + s.child_begin, s.child_count, s.p[4]);
            T w2 = jabs(p.x) - height / 2;
            return jmin(jmax(d2, w2), 0.0) + norm2(jmax(d2, 0.0), jmax(w2, 0.0));
        }
        if (kind == BMO_SHAPE_CYL_CONVEX) {  // CylindricalSDF.jl:62-85: op_extrude_x of sdf_cut_disk
            const double r = P0, dia = P1, height = P2;
            const double h = sqrt(r * r - (dia / 2) * (dia / 2));
            const double w = sqrt(r * r - h * h);
            T q1 = jabs(p.y), q2 = p.z;
            const double a = val(q1), b = val(q2);
            const double sv = jmax((h - r) * (a * a) + (w * w) * (h + r - 2 * b), h * a - w * b);
            T d2 = (sv < 0) ? norm2(q1, q2) - r : ((a < w) ? h - q2 : norm2(q1 - w, q2 - h));
            T w2 = jabs(p.x) - height / 2;
            return jmin(jmax(d2, w2), 0.0) + norm2(jmax(d2, 0.0), jmax(w2, 0.0));
        }
        if (kind == BMO_SHAPE_CYL_CONCAVE) {  // CylindricalSDF.jl:123-139
            const double radius = P0, dia = P1, height = P2;
            const double ar = fabs(radius);
            const double sg = ar - sqrt(ar * ar - 0.25 * (dia * dia));
            T x0 = p.x + 0.0, z0 = p.z + 0.0;
            T y1 = p.y + (-radius);
            T c = slab2(jabs(norm2(p.z, y1)) - ar, jabs(x0) - height / 2);
            T y2 = p.y + (-sg / 2 * jsign(radius));
            T qx = jabs(x0) - height / 2, qy = jabs(y2) - sg / 2, qz = jabs(z0) - dia / 2;
            T l = norm3t(jmax(qx, 0.0), jmax(qy, 0.0), jmax(qz, 0.0)) + jmin(jmax(qx, jmax(qy, qz)), 0.0);
            return jmax(l, -c);
        }
    }
    // LAST in the chain on purpose (round 4): at its head the optimiser evaluated this branch's norm — a square root — speculatively in front of the
    // kind dispatch, for EVERY leaf of every kind: config 2 on SURVEY 8(d)'s bundle 3.27 ms of kernel with it there, 3.06 without it
    if (cls == 5) {  // SPHERE | POINT  // SphericalLensSDF.jl:86-89, runtests.jl:943-946
        pin3(p);
        T r = norm3t(p.x, p.y, p.z);
        return kind == BMO_SHAPE_SPHERE ? r - P0 : r;
    }
    return T{} + kinf();
}

// leaf or MeniscusLensSDF (MeniscusLensSDF.jl:42-46: max(min(convex, cylinder), -concave) in the meniscus frame): one loop so the leaf
// switch is inlined once; the three leaf values are folded as they come (one running value live, not three)
template <class T, int EXT>
BMO_HD T sdf_simple(const SceneView& S, CShape& s, const v3<T>& pt) {
    const int kind0 = s.kind;  // (read through a wave-uniform reference: a scalar already)
    const bool men = EXT >= 1 && kind0 == BMO_SHAPE_MENISCUS;
    const int nleaf = men ? 3 : 1;
    T acc = T{};
    BMO_NOUNROLL
    for (int q = 0; q < nleaf; ++q) {
        CShape& leaf = men ? S.shapes[S.children[s.child_begin + q]] : s;
        const T v = sdf_leaf<T, EXT>(leaf, men ? leaf.kind : kind0, men ? to_local(s, pt) : pt, S.coefs);
        if (q == 0) acc = v;
        else if (q == 1) acc = jmin(acc, v);
        else acc = jmax(acc, -v);
    }
    return acc;
}

// Per-lane memory of the child values of ONE union along ONE march (device: a column of LDS, host emulator: a local array).
// v[c * stride] holds a lower bound of child c's sdf at the point of the previous evaluation (-inf = unknown).
constexpr int BMO_CC_MAX = 8;
struct ChildCache {
    double* v;      // v[c * stride]: lower bound of child c's sdf at the point where `acc` was last zero (-inf = unknown)
    int stride;
    int prev_best;  // arg-min child of the previous evaluation (evaluated first: it usually gives the tightest bound)
    double others_lb = -__builtin_huge_val();  // min of v[c] over the children other than prev_best (-inf: some child has no usable bound)
    double acc = 0.0;                          // upper bound of the distance the evaluation point has moved since the v[] were stored
    bool valid = false;                        // v[] belongs to the current march (false until its first evaluation has stored every child)
};
BMO_HD void child_cache_reset(ChildCache& cc) {
    cc.prev_best = 0;
    cc.others_lb = -kinf();  // the first evaluation of a march looks at every child and stores its value
    cc.acc = 0.0;
    cc.valid = false;
}

// sdf(shape, p) for any SDF shape incl. UnionSDF (UnionSDF.jl:53-56, left-fold min) together with
// the first-minimum child (argmin of UnionSDF.jl:86-91; same ordering rule as Base.min).
//
// Children that provably cannot be the minimum are not evaluated (result-preserving, DESIGN.md "union child skip"):
//  Lipschitz memory: every exact child sdf is 1-Lipschitz, so after the march moved the point by `moved` the child's value is
//  >= (its value, or lower bound, at the previous point) - moved.  (Round 1 also tested each child's bounding sphere per evaluation;
//  that test cost more than it saved — profiles/r02_ab_child_skip_variants.txt — and is gone.)
// A child whose lower bound exceeds max(bound, 0) by a margin is strictly greater than the final minimum (bound >= final minimum),
// so it changes neither the fold min nor the first-minimum index.  Evaluation order: the previous arg-min child first, then index
// order; min is exact and the arg-min rule below is written order-independently (lowest index among equal minima, -0.0 < +0.0
// like isless), so the result equals the reference's left fold.
// The usual iteration of a march — the previous arg-min child still wins and every other child is out of reach — is decided with ONE
// compare: `others_lb` is the smallest stored bound among the other children, so `others_lb - acc > bound` implies each child's own
// test; the stored values are then left alone and only `acc`, the distance moved since they were stored, grows.  Otherwise every
// child is looked at (evaluated, or its bound re-based to this point), which zeroes `acc` again.
// The integer fields of a shape's table entry, read ONCE per intersect3d (every march iteration used to re-read them from LDS, one
// dependent round trip each, because the compiler cannot hoist a load over the child-cache stores of the loop).
struct ShapeHead {
    int32_t kind, child_count, flags;
    int32_t child_begin, tri_begin;    // per lane
};
BMO_HD ShapeHead shape_head(CShape& s) {
    ShapeHead h;
    h.kind = s.kind;
    h.child_count = s.child_count;
    h.flags = s.flags;
    h.child_begin = s.child_begin;
    h.tri_begin = s.tri_begin;
    return h;
}
template <int EXT>
BMO_HD double sdf_any(const SceneView& S, const ShapeHead& H, CShape& s, const d3& p, int32_t& best_child, ChildCache& cc, double moved) {
#if defined(BMO_EMU_STATS)
    ++g_emu_sdf_any;
#endif
    const v3<double> pt{p.x, p.y, p.z};
    const bool uni = H.kind == BMO_SHAPE_UNION;
    const int nch = uni ? H.child_count : 1;
    double best = kinf();
    best_child = 0;
    // the child evaluated first: this lane's previous arg-min; where the shape tables are read wave-uniformly, the first active
    // lane's (any order gives the same result, see above; a coherent wave mostly agrees on it)
    const int first = (uni && BMO_UNIFORM(cc.prev_best) < nch) ? BMO_UNIFORM(cc.prev_best) : 0;
    const double acc = cc.acc + (moved * (1.0 + 1e-9) + 1e-12);  // |dir| is 1 only to rounding
    // children stored back to back in the shape table (the usual case; flagged by the engine when it copies the scene): child c is
    // shape first_id + c, no look-up in children[]
    const int32_t sflags = H.flags;
    const bool consecutive = uni && (sflags & BMO_SHAPE_FLAG_CONSECUTIVE);
    const int32_t cb = H.child_begin, first_id = H.tri_begin;
    const bool all_exact = !(sflags & BMO_SHAPE_FLAG_INEXACT);
    double m1 = kinf(), m2 = kinf();  // the two smallest bounds stored by this evaluation, and whose the smallest is
    int i1 = -1;
    BMO_NOUNROLL
    for (int q = 0; q < nch; ++q) {
        const int c = q == 0 ? first : (q <= first ? q - 1 : q);
        CShape& ch = uni ? S.shapes[consecutive ? first_id + c : S.children[cb + c]] : s;
        bool skip = false;
        double lb = -kinf();
        // a union is flagged INEXACT when one of its children is (scene compiler): only then the child's own flag is looked up
        const bool usable = uni && c < BMO_CC_MAX && (all_exact || !(ch.flags & BMO_SHAPE_FLAG_INEXACT));
        if (q > 0 && usable && cc.valid) {  // (the first child is always evaluated)
            const double bound = best > 0.0 ? best : 0.0;
            lb = cc.v[c * cc.stride] - acc;
            skip = lb > bound + 1e-12;
        }
        double stored = lb;
        if (!skip) {
#if defined(BMO_EMU_STATS)
            ++g_emu_sdf_leaf;
#endif
            const double v = sdf_simple<double, EXT>(S, ch, pt);
            stored = v;
            if (q == 0) {
                best = v;
                best_child = c;
                if (uni && first == cc.prev_best) {  // all the others out of reach?
                    const double bound = v > 0.0 ? v : 0.0;
                    if (cc.others_lb - acc > bound + 1e-12) {
                        cc.acc = acc;
                        return v;  // prev_best, others_lb and the stored values stay as they are
                    }
                }
            } else {
                const bool less = (v < best) || (v == best && sgn(v) && !sgn(best));
                const bool same = (v == best) && (sgn(v) == sgn(best));
                if (less || (same && c < best_child)) best_child = c;
                best = jmin(best, v);
            }
        }
        if (uni && c < BMO_CC_MAX) cc.v[c * cc.stride] = stored;  // a value, or a bound that is still valid at this point
        const double key = (usable && stored == stored) ? stored : -kinf();  // NaN: never skipped, so no bound
        if (key < m1) {
            m2 = m1;
            m1 = key;
            i1 = c;
        } else if (key < m2) {
            m2 = key;
        }
    }
    if (uni) {
        cc.prev_best = best_child;
        cc.others_lb = (i1 == best_child) ? m2 : m1;
        cc.acc = 0.0;
        cc.valid = true;
    }
    return best;
}

// normal3d(shape, p) of a non-union shape (UnionSDF -> normal of its argmin child, UnionSDF.jl:86-91: normal_any below):
// normal_fd (AbstractSDF.jl:90-95): dual-number gradient, NaN => numeric_gradient (:81-88).
template <int EXT>
BMO_HD d3 normal_of(const SceneView& S, CShape& sh, const d3& p) {
#if !defined(BMO_STUB_NORMAL)  // (developer builds: the numeric gradient alone — WRONG results, register / timing experiments only)
    {
#if defined(BMO_EMU_STATS)
        ++g_emu_normal;
#endif
        // ForwardDiff.gradient evaluates the sdf ONCE on duals of three partials.  BMO_DUAL_PASSES > 1 takes the three partials from two or
        // three evaluations on duals of fewer (DualN above: the same numbers, partial by partial; the emulator == oracle corpus is green in
        // every mode) — a leaf on duals of three needs 86 vector registers and holds every step kernel at 168 (3 waves per SIMD).  Measured in
        // round 4 (profiles/r04_ab_scheduling.txt item 8) and NOT the default: two + one at 128 registers keeps the march and the normal
        // code spill-free (176 B of scratch, all of it once per level around the interaction) and 4 waves per SIMD, and is 3.5 % SLOWER on
        // config 2 and even on config 5 — the extra evaluation costs what the fourth wave gives; one each in a loop spills 212 - 348 B.
        double g0 = 0.0, g1 = 0.0, g2 = 0.0;
#if BMO_DUAL_PASSES == 1
        {
            v3<DualN<3>> x{{p.x, {1, 0, 0}}, {p.y, {0, 1, 0}}, {p.z, {0, 0, 1}}};
            const DualN<3> y = sdf_simple<DualN<3>, EXT>(S, sh, x);
            g0 = y.p[0], g1 = y.p[1], g2 = y.p[2];
        }
#elif BMO_DUAL_PASSES == 2
        {
            v3<DualN<2>> x{{p.x, {1, 0}}, {p.y, {0, 1}}, {p.z, {0, 0}}};
            const DualN<2> y = sdf_simple<DualN<2>, EXT>(S, sh, x);
            g0 = y.p[0], g1 = y.p[1];
        }
        {
            v3<DualN<1>> x{{p.x, {0}}, {p.y, {0}}, {p.z, {1}}};
            const DualN<1> y = sdf_simple<DualN<1>, EXT>(S, sh, x);
            g2 = y.p[0];
        }
#else
        BMO_NOUNROLL
        for (int ax = 0; ax < 3; ++ax) {  // one code site, three trips: d/dx, d/dy, d/dz
            v3<DualN<1>> x{{p.x, {ax == 0 ? 1.0 : 0.0}}, {p.y, {ax == 1 ? 1.0 : 0.0}}, {p.z, {ax == 2 ? 1.0 : 0.0}}};
            const DualN<1> y = sdf_simple<DualN<1>, EXT>(S, sh, x);
            g0 = ax == 0 ? y.p[0] : g0;
            g1 = ax == 1 ? y.p[0] : g1;
            g2 = ax == 2 ? y.p[0] : g2;
        }
#endif
        d3 n = normalize_inv(d3{g0, g1, g2});
        if (!isnan_(n.x) && !isnan_(n.y) && !isnan_(n.z)) return n;
#if defined(BMO_EMU_STATS)
        ++g_emu_normal_fd;
#endif
    }
#endif
    const double e = S.grad_h;
    double g0 = 0, g1 = 0, g2 = 0;
    BMO_NOUNROLL
    for (int q = 0; q < 6; ++q) {  // +x -x +y -y +z -z ; the untouched components get +0.0 / -0.0 like Point3(eps,0,0)
        const int ax = q >> 1;
        const bool minus = q & 1;
        double ox = ax == 0 ? e : 0.0, oy = ax == 1 ? e : 0.0, oz = ax == 2 ? e : 0.0;
        v3<double> pt = minus ? v3<double>{p.x - ox, p.y - oy, p.z - oz} : v3<double>{p.x + ox, p.y + oy, p.z + oz};
#if defined(BMO_EMU_STATS)
        ++g_emu_sdf_leaf;
#endif
        double v = sdf_simple<double, EXT>(S, sh, pt);
        // f(p+h) - f(p-h): the '+' value is stored first, the '-' value subtracted from it
        if (ax == 0) g0 = minus ? g0 - v : v;
        else if (ax == 1) g1 = minus ? g1 - v : v;
        else g2 = minus ? g2 - v : v;
    }
    return normalize_div(d3{g0, g1, g2});
}
// normal3d of shape `sid` or, for a UnionSDF, of its arg-min child (UnionSDF.jl:86-91); `kind`, `child_begin`: the shape's table fields.
// The arg-min children of the lanes may differ: one pass per distinct shape among them (scalar scene access, file header).
template <int EXT>
BMO_HD d3 normal_at(const SceneView& S, int32_t sid, int32_t kind, int32_t child_begin, const d3& p, int32_t best_child) {
    int32_t tid = sid;
    if (kind == BMO_SHAPE_UNION) tid = S.children[child_begin + best_child];
    d3 n{0, 0, 0};
    for (bool todo = true; todo;) {
        const int32_t u = BMO_UNIFORM(tid);
        if (bmo_same_id(u, tid)) {
            n = normal_of<EXT>(S, S.shapes[u], p);
            todo = false;
        }
    }
    return n;
}

// ------------------------------------------------------------------ miss cull (see file header)
// Returns < 0 when the reference provably returns `nothing` for this shape and ray (the line misses the
// inflated bounding sphere, or the origin is outside it and receding); otherwise a lower bound (>= 0) of the
// ray parameter t of any hit: the hit point lies inside the sphere, so t >= the sphere entry parameter.
BMO_HD double cull_entry_raw(double cx, double cy, double cz, double R, const d3& pos, const d3& dir);
BMO_HD double cull_entry(CShape& s, const d3& pos, const d3& dir) { return cull_entry_raw(s.bs_center[0], s.bs_center[1], s.bs_center[2], s.bs_radius, pos, dir); }
BMO_HD double cull_entry_raw(double cx, double cy, double cz, double R, const d3& pos, const d3& dir) {
    if (!(R >= 0.0)) return 0.0;
    d3 oc{cx - pos.x, cy - pos.y, cz - pos.z};
    double dd = dot3(dir, dir), b = dot3(oc, dir), cc = dot3(oc, oc), R2 = R * R;
    if (!(cc > R2)) return 0.0;          // origin inside the sphere: no bound
    if (b < 0.0) return -1.0;            // outside and receding
    double disc = b * b - dd * (cc - R2);
    if (disc < 0.0) return -1.0;         // the whole line misses the sphere
    return (b - sqrt(disc)) / dd;
}
struct BoundSphere {  // bs_center / bs_radius of a shape
    double cx, cy, cz, R;
};
BMO_HD bool cull_receding(const BoundSphere& b, const d3& pos, const d3& dir) {
    if (!(b.R >= 0.0)) return false;
    d3 co{pos.x - b.cx, pos.y - b.cy, pos.z - b.cz};
    return dot3(co, co) > b.R * b.R && dot3(co, dir) > 0.0;
}

// MoellerTrumboreAlgorithm Mesh.jl:203-237
BMO_HD double moeller_trumbore(CDouble* f, const d3& pos, const d3& dir, double keps, double leps) {
    d3 V1{f[0], f[1], f[2]}, V2{f[3], f[4], f[5]}, V3{f[6], f[7], f[8]};
    d3 E1 = sub3(V2, V1), E2 = sub3(V3, V1);
    d3 Pv = cross3(dir, E2);
    double Det = dot3(E1, Pv);
    if (fabs(Det) < keps) return kinf();
    d3 Tv = sub3(pos, V1);
    double invDet = 1 / Det;
    double u = dot3(Tv, Pv) * invDet;
    if ((u < 0 - keps) || (u > 1 + keps)) return kinf();
    d3 Qv = cross3(Tv, E1);
    double v = dot3(dir, Qv) * invDet;
    if ((v < 0 - keps) || (u + v > 1 + keps)) return kinf();
    double t = dot3(E2, Qv) * invDet;
    if (t < leps) return kinf();
    return t;
}

// intersect3d(mesh, ray) Mesh.jl:244-267: nearest face with strict `<` (the first face wins ties).  Returns t (inf: none) and the face.
BMO_HD double mesh_nearest(const SceneView& S, int tri0, int ntri, const d3& pos, const d3& dir, int32_t& fid) {
    double t0 = kinf();
    fid = -1;
    BMO_NOUNROLL
    for (int i = 0; i < ntri; ++i) {
        const double t = moeller_trumbore(S.tris + 9 * (tri0 + i), pos, dir, S.mt_keps, S.mt_leps);
        if (t < t0) {
            t0 = t;
            fid = i;
        }
    }
    return t0;
}
// normal3d(mesh, fID) Mesh.jl:183-192, normalised a second time by intersect3d (Mesh.jl:265)
BMO_HD d3 mesh_normal(const SceneView& S, int tri) {
    CDouble* f = S.tris + 9 * tri;
    d3 V1{f[0], f[1], f[2]}, V2{f[3], f[4], f[5]}, V3{f[6], f[7], f[8]};
    d3 n = normalize_div(cross3(sub3(V2, V1), sub3(V3, V1)));
    return normalize_div(n);
}

// One entry of the candidate table: the parts of the leaf objects in the reference's order (objects in Leaves order, parts in the
// order intersect3d(object, ray) visits them; a plate splitter's coating before its substrate, PlateBeamsplitter.jl:160-187;
// NonInteractable objects have none, NonInteractable.jl:19), each with its shape's bounding sphere.  Built once per scene
// (fill_candidates below) so that trace_all (System.jl:57-72) is one flat loop over 48-byte records instead of a nest over
// objects and parts with table look-ups at every level.
struct Cand {
    double cx, cy, cz, R;  // bs_center / bs_radius of the part's shape
    int32_t sid, obj;
    int32_t info;          // bit 0: the part is the substrate of a plate splitter (tested after its coating); bit 1: its coating
    int32_t pad;
};
enum { CAND_PLATE_SUBSTRATE = 1, CAND_PLATE_COATING = 2 };
// number of candidates of a scene / the table itself (host side, at scene creation; `out` may be null to count)
inline int fill_candidates(const bmo_object* objects, int n_objects, const bmo_shape* shapes, Cand* out) {
    int n = 0;
    for (int o = 0; o < n_objects; ++o) {
        const bmo_object& ob = objects[o];
        int np = (ob.kind == BMO_OBJ_DOUBLET) ? 2 : (ob.kind == BMO_OBJ_CUBE_BS ? 3 : (ob.kind == BMO_OBJ_PLATE_BS ? 2 : 1));
        if (ob.kind == BMO_OBJ_NONINTERACTABLE) np = 0;
        for (int k = 0; k < np; ++k) {
            int32_t sid = ob.shape[k];
            int32_t info = 0;
            if (ob.kind == BMO_OBJ_PLATE_BS) {  // coating (shape[1]) first, then substrate (shape[0])
                sid = ob.shape[1 - k];
                info = k == 0 ? CAND_PLATE_COATING : CAND_PLATE_SUBSTRATE;
            }
            if (out) {
                const bmo_shape& sh = shapes[sid];
                out[n] = Cand{sh.bs_center[0], sh.bs_center[1], sh.bs_center[2], sh.bs_radius, sid, o, info, 0};
            }
            ++n;
        }
    }
    return n;
}

// Per-lane memory of one tracing step besides the child cache (device: columns of LDS, host: a local array), m[c * stride]:
//   0..2  end point of the march that produced the best hit so far; after the step: the normal of the winning hit
//   3..5  origin of the ray (re-read where a candidate is set up instead of being held in registers across the marches)
//   6     PLATE_BS: t of the coating's hit
// The normal of a hit is evaluated once, for the hit that wins the tracing step (see tracing_step).
constexpr int BMO_LANE_MEM = 11;  // 0..6: tracing_step (above); 7..10: the Beam step kernels' level loop (refractive index and optical path of
                                  // the lane's ray, wavelength index and wavelength of its beam: bmo_engine.hip)
struct LaneMem {
    double* m;
    int stride;
    BMO_HD d3 get3(int c) const { return d3{m[c * stride], m[(c + 1) * stride], m[(c + 2) * stride]}; }
    BMO_HD void put3(int c, const d3& v) const {
        m[c * stride] = v.x;
        m[(c + 1) * stride] = v.y;
        m[(c + 2) * stride] = v.z;
    }
};

// the `nothing` decisions of cull_entry without its square root: the line misses the (inflated) bounding sphere, or the origin is
// outside it and receding
BMO_HD bool cull_miss(double cx, double cy, double cz, double R, const d3& pos, const d3& dir) {
    const d3 oc{cx - pos.x, cy - pos.y, cz - pos.z};
    const double dd = dot3(dir, dir), b = dot3(oc, dir), cc = dot3(oc, oc), R2 = R * R;
    // (`&` / `|` on purpose: one straight line of compares instead of an execution-mask branch per early exit — the candidate collection of
    //  tracing_step runs this for every table entry and level)
    return (R >= 0.0) & (cc > R2) & ((b < 0.0) | (b * b - dd * (cc - R2) < 0.0));
}

// tracing_step! (System.jl:100-110) = trace_one (:74-85) falling back to trace_all (:57-72), with intersect3d of objects
// (SingleShape / MultiShape AbstractRay.jl:118-155, plate splitter PlateBeamsplitter.jl:160-187, NonInteractable.jl:19), of meshes
// (Mesh.jl:244-267) and of SDF shapes (AbstractSDF.jl:166-181 with _raymarch_outside :102-125 and _raymarch_inside :132-159) written as
// ONE flat nest of wave-uniform loops around one SDF evaluation site and one normal site:
//
//   pass   0 (RETR builds): the probe of retrace_system! (System.jl:208-218) — only the hinted shape, or without a hint only the object
//            of the stored intersection; a probe miss is the reference's cleanup path, after which solve_leaf! traces the same ray on
//            WITHOUT a hint (System.jl:130-133) if `fresh_allowed` (length(rays) < r_max);
//          1 trace_one / trace_all;   2 the normal of the winning hit.
//   slot   -1 the hinted SHAPE (if any), 0..NC-1 the candidate table (parts of the leaf objects in the reference's order).  Before
//          slot 0 every lane that goes on to trace_all collects the candidates it has to march in a bit mask (`cull_miss` against the
//          bounding spheres: one tight loop over the table), so that the slots themselves only visit what some lane needs.
//   round  march loop, then the normal site for the lanes that wait for one, at most twice:
//   march  per lane a state machine around one sdf evaluation per trip:
//            CLASSIFY: d = sdf(p0); d > eps_srf -> OUTSIDE; else the lane waits for the normal site -> INSIDE or `nothing`
//            INSIDE:   p += eps_ins*dir, t_in += eps_ins; sdf(p) > 0 -> OUTSIDE backwards from here (dist = that sdf)
//            OUTSIDE:  p += dist*dir', dist = sdf(p), t0 += dist, hit when dist < eps_ray
//          The reference's _raymarch_outside re-evaluates sdf at its start point; that value equals the one just computed at the same
//          point, so it is reused (bit-identical, one evaluation less).  The backward march uses p + (-dist)*dir, which is bit for bit
//          p + dist*(-dir) (IEEE negation is exact and commutes with rounding).
//
// What differs from the reference's evaluation ORDER, never from its results:
//   * A hit is folded into the running best at once (strict `<` over slots in the reference's order keeps the first of equal
//     minima, which is what the reference's two-level fold — parts, then objects — keeps; the plate splitter's coating-wins-ties rule
//     needs the coating's t while the substrate is marched: lane memory 6).
//   * normal3d is a pure function of (shape, point): the reference evaluates it for every hit, here it is evaluated for the ONE hit
//     that wins (pass 2) from the stored end point of its march and arg-min child; the on-surface start classification
//     (dot(dir, normal) <= 0 => entering, AbstractSDF.jl:171-177) evaluates it where the reference does.  Lanes that need a normal
//     wait until no lane of the wave is marching, then the dual-number code runs once for all of them — outside the march loop, and
//     the march variables are re-initialised behind it, so nothing of a march is live across the normal code.
//   * `lim`: a hit whose t exceeds the best t so far cannot win (System.jl:67), so an outside march stops once its running t0 exceeds
//     it by the 1e-6 margin; t0 only grows by positive steps until the final sub-tolerance step (DESIGN.md "nearest-hit prune").
//   * the miss cull (file header).  `calls` counts the reference's intersect3d calls, culled or not.
// Returns the winning hit; its normal is in lane memory 0..2 (X.n is filled from there).
template <int EXT, bool RETR = false>
BMO_HD Hit tracing_step(const SceneView& S, const d3& pos_in, const d3& dir0, int32_t hint_obj, int32_t hint_shape, uint32_t& calls, ChildCache& cc,
                        const LaneMem& lm, bool probe = false, int32_t probe_obj = -1, bool fresh_allowed = true, bool* probe_missed = nullptr) {
    enum { CLASSIFY = 0, INSIDE = 1, OUTSIDE = 2, FINAL = 3 };
    double x_t = kinf();
    int32_t x_shape = -1, x_obj = -1;
    int32_t x_aux = 0;          // arg-min child of the best hit (SDF) or its face (mesh)
    bool done = false;
    int32_t skip_obj = -1;      // object already tested by the probe with this very ray (pure function => same `nothing`)
    int32_t tested_shape = -1;  // shape tested in slot -1 (or by the probe) with this very ray
    int32_t coat_obj = -1;      // plate splitter whose coating this ray has hit (its t: lane memory 6)
    const int n_obj = S.n_objects, n_cand = S.n_cands;
    lm.put3(3, pos_in);
    BMO_NOUNROLL
    for (int pass = RETR ? 0 : 1; pass <= 2; ++pass) {
        bool in_pass;
        int32_t hs = -1;  // shape tested in slot -1 of this pass (-1: none)
        int o_lo = 0, o_hi = n_obj;
        if (pass == 2) {
            in_pass = x_shape >= 0;
        } else if (RETR && pass == 0) {
            in_pass = probe;
            if (in_pass) {
                calls += 1;
                hs = hint_shape;
                if (hint_shape >= 0) {
                    o_hi = 0;
                } else {
                    o_lo = probe_obj;
                    o_hi = probe_obj + 1;
                }
            }
        } else {
            in_pass = !done;
            if (RETR && probe) {
                if (in_pass) {  // the probe missed
                    if (probe_missed) *probe_missed = true;
                    if (!fresh_allowed) {
                        in_pass = false;
                    } else {
                        if (hint_shape < 0) skip_obj = probe_obj;
                        calls += (uint32_t)n_obj;
                    }
                }
            } else if (in_pass) {
                hs = hint_shape;
                calls += hs >= 0 ? 1u : (uint32_t)n_obj;
            }
        }
        if (in_pass && hs >= 0) tested_shape = hs;
        const int c_end = pass == 2 ? 0 : n_cand;
        BMO_NOUNROLL
        for (int c0 = 0; c0 == 0 || c0 < c_end; c0 += 64) {  // the candidate table in chunks of 64 (one mask); slot -1 belongs to the first
            unsigned long long mask = 0;       // candidates of this chunk the lane has to march
            unsigned long long wave_mask = 0;  // ... some lane of the wave has to march (a scalar: the slot loop walks its set bits)
            const int c1 = c_end - c0 < 64 ? c_end - c0 : 64;
            // slot -1 (first chunk only), then the set bits of wave_mask in rising order; q == -2: a later chunk starts with the collection
            BMO_NOUNROLL
            for (int q = c0 == 0 ? -1 : -2;;) {  // wave-uniform; lanes skip the slots they do not use
                const int slot = q < 0 ? -1 : c0 + q;
                bool take = false;
                int32_t sid = 0, info = 0, obj = -1;
                if (q == -1) {
                    take = in_pass && (pass == 2 || hs >= 0);
                    sid = pass == 2 ? x_shape : hs;
                } else if (q >= 0) {
                    take = ((mask >> q) & 1ull) != 0;
                    const BMO_KONST Cand& cd = S.cands[slot];
                    sid = cd.sid;
                    obj = cd.obj;
                    info = cd.info;
                }
                // one pass per distinct shape among the lanes that take this slot (a candidate slot has one shape for all of them; the
                // hinted shape and the winning shape can differ between lanes): inside, `sid` and everything read from its table entry
                // is wave-uniform (scalar scene access, file header)
                for (bool todo = take; todo;) {
                    const int32_t usid = BMO_UNIFORM(sid);
                    if (!bmo_same_id(usid, sid)) continue;
                    todo = false;
                    bool active = true;
                    CShape& s = S.shapes[usid];
                    const int32_t s_kind = s.kind, s_flags = s.flags;
                    double lim = kinf();
                    d3 pos{0, 0, 0};
                    if (pass != 2) {
                        pos = lm.get3(3);
                        if (slot < 0) {
                            if (cull_miss(s.bs_center[0], s.bs_center[1], s.bs_center[2], s.bs_radius, pos, dir0)) active = false;  // provable `nothing`
                        } else if (x_shape >= 0) {  // (the mask holds only candidates whose sphere the ray's line meets)
                            lim = x_t + 1e-6 * (1.0 + x_t);
                            const double t_lb = cull_entry(s, pos, dir0);
                            if (t_lb * (1.0 - 1e-9) - 1e-9 > lim) active = false;  // provable loser of the nearest-hit selection
                        }
                    }
                    // a hit of this part at t (end point of the march p, arg-min child / face a): the reference's selection rules
                    auto offer = [&](double t, int32_t a, const d3& p) {
                        bool accept = x_shape < 0 || t < x_t;
                        if (info & CAND_PLATE_COATING) {
                            lm.m[6 * lm.stride] = t;
                            coat_obj = obj;
                        }
                        if ((info & CAND_PLATE_SUBSTRATE) && coat_obj == obj) {
                            // this plate's coating was hit by the same ray: it wins ties (isapprox) and when strictly nearer
                            const double t_coat = lm.m[6 * lm.stride];
                            if (isapprox(t_coat, t, 0.0) || t_coat < t) accept = false;
                        }
                        if (accept) {
                            x_t = t;
                            x_shape = usid;
                            x_obj = slot < 0 ? hint_obj : obj;
                            x_aux = a;
                            lm.put3(0, p);
                            if (slot < 0) done = true;
                        }
                    };
                    if (s_kind == BMO_SHAPE_MESH) {
                        if (pass == 2) {
                            lm.put3(0, mesh_normal(S, s.tri_begin + x_aux));
                        } else if (active) {
                            int32_t fid = -1;
                            const double t = mesh_nearest(S, s.tri_begin, s.tri_count, pos, dir0, fid);
                            if (fid >= 0) offer(t, fid, d3{0, 0, 0});
                        }
                        active = false;
                    }
                    // ---- SDF shapes: sphere tracing
                    if (BMO_WAVE_ANY(active)) {
                        const ShapeHead H{s_kind, s.child_count, s_flags, s.child_begin, s.tri_begin};
                        const bool exact = !(s_flags & BMO_SHAPE_FLAG_INEXACT);  // the running-t prune needs a 1-Lipschitz SDF
                        // the lane's march state in ONE integer (phase in bits 0-1): kept in a vector register and tested with vector
                        // compares — as separate booleans, updated in different branches of a per-lane loop, it lived in scalar
                        // lane masks that cost three scalar instructions per update and merge
                        enum { ST_BACK = 4, ST_PENDING = 8, ST_ACTIVE = 16 };  // PENDING: the lane waits for the normal site
                        int st = (active ? ST_ACTIVE : 0) | (pass == 2 ? (FINAL | ST_PENDING) : CLASSIFY);
                        int32_t bc = 0;
                        if (pass == 2) {
                            pos = lm.get3(0);
                            bc = x_aux;
                        }
                        BMO_NOUNROLL
                        for (int round = 0; round < 2; ++round) {
                            double dist = 0.0, t0 = 0.0, t_in = 0.0;
                            int it = 1;  // iteration counter of the running march (inside, then outside)
                            child_cache_reset(cc);
                            // (a per-lane loop: lanes leave it as their marches end, the wave stays until the last one has left)
                            while ((st & (ST_ACTIVE | ST_PENDING)) == ST_ACTIVE) {
                                const int phase = st & 3;
                                double moved = 0.0;  // how far this trip moves the evaluation point (Lipschitz memory of the union children)
                                if (phase == INSIDE) {
                                    pos = axpy3(pos, S.eps_ins, dir0);
                                    t_in += S.eps_ins;
                                    moved = S.eps_ins;
                                } else if (phase == OUTSIDE) {
                                    pos = axpy3(pos, (st & ST_BACK) ? -dist : dist, dir0);
                                    moved = fabs(dist);
                                }
                                const double d = sdf_any<EXT>(S, H, s, pos, bc, cc, moved);
                                if (phase == OUTSIDE) {
                                    // (the receding test below only when the distance GREW with this step: outside the bounding ball and moving away
                                    //  from it an exact sdf grows with every step, and a march that is converging skips 16 instructions per trip;
                                    //  for the inexact leaves the test merely comes a step later — it proves misses, it never makes one.  Round 4:
                                    //  ragged bundle - 1.3 %, the others - 0.2 %)
                                    const bool growing = d > dist;
                                    dist = d;
                                    t0 += d;
                                    it += 1;
                                    if (d < S.eps_ray) {
                                        offer((st & ST_BACK) ? t_in - t0 : t0, bc, pos);
                                        st = 0;
                                    } else {
                                        // outside the bounding sphere and receding (the backward march runs along -dir: dot(co, -dir) =
                                        // -dot(co, dir) exactly): provable miss, skip the rest of the 1000 evaluations
                                        bool recede = false;
                                        if (growing) {
                                        const double R = s.bs_radius;
                                        const d3 co{pos.x - s.bs_center[0], pos.y - s.bs_center[1], pos.z - s.bs_center[2]};
                                        const double cd = dot3(co, dir0);
                                        recede = R >= 0.0 && dot3(co, co) > R * R && ((st & ST_BACK) ? cd < 0.0 : cd > 0.0);
                                        }
                                        if (recede || (exact && !(st & ST_BACK) && t0 > lim)  // provable loser of the nearest-hit selection
                                            || !(it <= S.march_iters))
                                            st = 0;
                                    }
                                } else if (phase == CLASSIFY) {
                                    if (d > S.eps_srf) {
                                        st = ST_ACTIVE | OUTSIDE;
                                        dist = d;
                                        t0 = d;
                                        it = 1;
                                        if (!(it <= S.march_iters) || (exact && t0 > lim)) st = 0;  // the reference returns `nothing`
                                    } else {
                                        st |= ST_PENDING;
                                    }
                                } else {  // INSIDE
                                    if (d > 0) {  // once outside, fall back to _raymarch_outside with -dir
                                        st = ST_ACTIVE | ST_BACK | OUTSIDE;
                                        dist = d;
                                        t0 = d;
                                        it = 1;
                                    } else {
                                        it += 1;
                                    }
                                    if (!(it <= S.march_iters)) st = 0;
                                }
                            }
                            // nobody is marching: every lane still active waits for a normal (or nobody is left)
                            if (!BMO_WAVE_ANY(st & ST_ACTIVE)) break;
                            if (st & ST_ACTIVE) {
                                // single normal site: the reference's normal3d, for the start classification on the surface
                                // (dot(dir, normal) <= 0 => entering, AbstractSDF.jl:171-177) and for the winning hit
                                const d3 n = normal_at<EXT>(S, usid, s_kind, H.child_begin, pos, bc);
                                if ((st & 3) == FINAL) {
                                    lm.put3(0, n);
                                    st = 0;
                                } else if (dot3(dir0, n) <= 0 && 1 <= S.march_iters) {  // entering
                                    st = ST_ACTIVE | INSIDE;
                                } else {
                                    st = 0;  // on the surface and leaving: no intersection
                                }
                            }
                        }
                    }
                }
                // trace_one missed the hinted shape: fall back to trace_all (System.jl:104-108); the retrace probe does not
                if (slot < 0 && pass == 1 && take && !done) calls += (uint32_t)n_obj;
                // ---- next slot: the lowest candidate of this chunk that some lane still has to march
                if (q < 0) {
                    if (pass == 2) break;
                    // the candidates of this chunk the lane has to march: not culled by its bounding sphere, not excluded by the retrace
                    // probe's rules, not the shape tested a moment ago with the same ray.  Four table entries per trip so that their
                    // scalar loads are in flight together.
                    const bool collect = in_pass && !done && o_lo < o_hi;
                    if (c1 > 0 && BMO_WAVE_ANY(collect)) {
                        const d3 p0 = lm.get3(3);
                        BMO_NOUNROLL
                        for (int i0 = 0; i0 < c1; i0 += 4) {
                            bool want[4];
                            const BMO_KONST Cand* cp = S.cands + (c0 + i0);  // (the table is padded: entries behind the last one are zeros)
                            for (int u = 0; u < 4; ++u) {
                                const BMO_KONST Cand& cd = cp[u];
                                const int32_t co = cd.obj, cs = cd.sid;
                                // (only the retrace probe narrows the object range or excludes an object)
                                const bool in_range = RETR ? ((co >= o_lo) & (co < o_hi) & (co != skip_obj)) : true;
                                want[u] = (i0 + u < c1) & collect & in_range & (cs != tested_shape) & !cull_miss(cd.cx, cd.cy, cd.cz, cd.R, p0, dir0);
                            }
                            for (int u = 0; u < 4; ++u) {
                                mask |= (unsigned long long)want[u] << (i0 + u);
                                if (BMO_WAVE_ANY(want[u])) wave_mask |= 1ull << (i0 + u);
                            }
                        }
                    }
                } else {
                    wave_mask &= ~(1ull << q);
                }
                if (!wave_mask) break;
                q = (int)__builtin_ctzll(wave_mask);
            }
        }
        if (RETR && pass == 0 && x_shape >= 0) done = true;  // the stored path still holds
    }
    Hit X;
    X.t = x_t;
    X.obj = x_obj;
    X.shape = x_shape;
    X.n = x_shape >= 0 ? lm.get3(0) : d3{0, 0, 0};
    return X;
}

// ------------------------------------------------------------------ interactions
BMO_HD d3 reflection3d(const d3& dir, const d3& n) {  // OpticUtils.jl:7-9
    double k = 2 * dot3(dir, n);
    return {dir.x - k * n.x, dir.y - k * n.y, dir.z - k * n.z};
}
// refraction3d OpticUtils.jl:31-45; false = unit-length ArgumentError
BMO_HD bool refraction3d(const d3& dir, const d3& normal, double n1, double n2, d3& out, bool& tir) {
    if (!isapprox(norm3(dir), 1.0, 0.0)) return false;
    if (!isapprox(norm3(normal), 1.0, 0.0)) return false;
    double n = n1 / n2;
    double ci = -dot3(normal, dir);
    double st2 = (n * n) * (1 - ci * ci);
    if (st2 > 1.0) {
        out = reflection3d(dir, normal);
        tir = true;
        return true;
    }
    double ct = sqrt(1 - st2);
    double k = n * ci - ct;
    out = {n * dir.x + k * normal.x, n * dir.y + k * normal.y, n * dir.z + k * normal.z};
    tir = false;
    return true;
}

// complex helpers (textbook formulas; same convention as the oracle)
struct cx {
    double re, im;
};
BMO_HD cx cadd(cx a, cx b) { return {a.re + b.re, a.im + b.im}; }
BMO_HD cx csub(cx a, cx b) { return {a.re - b.re, a.im - b.im}; }
BMO_HD cx cneg(cx a) { return {-a.re, -a.im}; }
BMO_HD cx cmul(cx a, cx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
BMO_HD cx rmul(double s, cx a) { return {s * a.re, s * a.im}; }
BMO_HD cx cmulr(cx a, double s) { return {a.re * s, a.im * s}; }
BMO_HD cx cdiv(cx a, cx b) {
    double den = b.re * b.re + b.im * b.im;
    return {(a.re * b.re + a.im * b.im) / den, (a.im * b.re - a.re * b.im) / den};
}
BMO_HD double cabs2(cx a) { return a.re * a.re + a.im * a.im; }

// one ray of a beam node as the kernel carries it between bounces
struct RayS {
    d3 pos, dir;
    double n;
    cx E0[3];
};

enum StepOutcome { OUT_MISS = 0, OUT_STOP = 1, OUT_CONTINUE = 2, OUT_SPLIT = 3 };

struct StepOut {
    int outcome;
    int status;              // node status bits to OR in
    int32_t hint_obj, hint_shape;
    RayS next;               // OUT_CONTINUE: new segment; OUT_SPLIT: transmitted child
    RayS refl;               // OUT_SPLIT: reflected child
    int det_slot;            // >= 0: a detector hit was produced
    double* det;             // destination of the detector record, 9 doubles (the node's hit slot: written in place, set by the caller)
};

BMO_HD double n_medium(const SceneView& S, int medium, int li) { return S.n_table[medium * S.n_lambda + li]; }

BMO_HD bool isparallel3d(const d3& a, const d3& b) { return isapprox(fabs(dot3(normalize_div(a), normalize_div(b))), 1.0, 2.220446049250313e-16); }

BMO_HD d3 perpendicular_to(const d3& in) {  // deterministic stand-in for the reference's random normal3d(input)
    double nn = norm3(in);
    d3 nw = fabs(in.x) > 0.9 * nn ? d3{0, 1, 0} : d3{1, 0, 0};
    double k = dot3(nw, in), n2 = nn * nn;
    nw = {nw.x - k * in.x / n2, nw.y - k * in.y / n2, nw.z - k * in.z / n2};
    return normalize_div(nw);
}

// res = (A * B) * v for 3x3 complex A, real B given row-major; left folds like StaticArrays
BMO_HD void mat3_apply(const cx* P, const cx* v, cx* res) {
    for (int i = 0; i < 3; ++i) res[i] = cadd(cadd(cmul(P[3 * i + 0], v[0]), cmul(P[3 * i + 1], v[1])), cmul(P[3 * i + 2], v[2]));
}

// _calculate_global_E0 PolarizedRays.jl:165-207 with J = diag(j11, j22, 1)
BMO_HD void global_E0(const d3& in, const d3& out, const d3& normal, cx j11, cx j22, const cx* E0, cx* res) {
    bool par = isparallel3d(in, out);
    d3 v = par ? normal : out;
    if (isparallel3d(in, normal)) v = perpendicular_to(in);
    d3 s = normalize_div(cross3(in, v));
    d3 p1 = cross3(in, s);
    double Oin[9] = {s.x, s.y, s.z, p1.x, p1.y, p1.z, in.x, in.y, in.z};
    double Oo[9];
    d3 mo = neg3(out);
    bool anti = norm3(sub3(in, mo)) <= 1.4901161193847656e-08 * fmax(norm3(in), norm3(mo));
    d3 c1 = p1, c2 = in;
    if (!(par && !anti)) {
        c1 = cross3(out, s);
        c2 = out;
    }
    Oo[0] = s.x; Oo[1] = c1.x; Oo[2] = c2.x;
    Oo[3] = s.y; Oo[4] = c1.y; Oo[5] = c2.y;
    Oo[6] = s.z; Oo[7] = c1.z; Oo[8] = c2.z;
    cx J[9] = {j11, {0, 0}, {0, 0}, {0, 0}, j22, {0, 0}, {0, 0}, {0, 0}, {1, 0}};
    cx A[9], P[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            A[3 * i + j] = cadd(cadd(rmul(Oo[3 * i + 0], J[0 + j]), rmul(Oo[3 * i + 1], J[3 + j])), rmul(Oo[3 * i + 2], J[6 + j]));
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            P[3 * i + j] = cadd(cadd(cmulr(A[3 * i + 0], Oin[0 + j]), cmulr(A[3 * i + 1], Oin[3 + j])), cmulr(A[3 * i + 2], Oin[6 + j]));
    mat3_apply(P, E0, res);
}
BMO_HD bool e0_orthogonal(const d3& dir, const cx* E0) {  // PolarizedRays.jl:54-56
    cx d = cadd(cadd(rmul(dir.x, E0[0]), rmul(dir.y, E0[1])), rmul(dir.z, E0[2]));
    return sqrt(cabs2(d)) <= 1e-14;
}
BMO_HD void fresnel(double theta, double n, cx& rs, cx& rp, cx& ts, cx& tp) {  // OpticUtils.jl:121-131
    double ct = jl::cos(theta), st = jl::sin(theta);  // Base's own sin / cos (bmo_jlmath.hpp): runtests.jl:157 needs their bits
    double x = n * n - st * st;
    cx r = x >= 0.0 ? cx{sqrt(x), 0.0} : cx{0.0, sqrt(-x)};
    cx c{ct, 0};
    rs = cdiv(csub(c, r), cadd(c, r));
    cx n2c{(n * n) * ct, 0};
    rp = cdiv(cadd(cx{-(n * n) * ct, 0}, r), cadd(n2c, r));
    ts = cadd(rs, cx{1, 0});
    tp = cdiv(cx{2 * n * ct, 0}, cadd(n2c, r));
}
BMO_HD double angle3d(const d3& a, const d3& b) {  // LinearAlgebraUtils.jl:103-108
    double arg = dot3(a, b) / (norm3(a) * norm3(b));
    arg = arg < -1.0 ? -1.0 : (arg > 1.0 ? 1.0 : arg);
    return jl::acos(arg);
}

BMO_HD d3 hit_point(const RayS& r, double t) { return axpy3(r.pos, t, r.dir); }

// interact3d for one Ray / PolarizedRay: the object kind is decoded into ONE action so that each
// action body (reflect / refract / split / detectors / polarizer) is inlined exactly once.
//   reflect   Mirrors.jl:39-69          refract  Lenses.jl:46-126 (+ DoubletLenses.jl:66-76 hint rewrite)
//   split     ThinBeamsplitter.jl:73-115, PlateBeamsplitter.jl:189-228, CubeBeamsplitter.jl:63-92
//   detectors Spotdetector.jl:50-61, PSFDetector.jl:77-89     polarizer PolarizationFilter.jl:31-48
// `opl_before` = optical path length of the beam (incl. parents) up to the START of this segment.
// Where the geometric part of the ray that goes on (OUT_CONTINUE: the new segment, OUT_SPLIT: the transmitted child) is put the moment
// it is known.  NextInOut leaves it in StepOut::next; the Beam step kernels hand in the lane memory instead (bmo_engine.hip NextInLaneMem):
// every branch of the interaction stores its result there at once, so that nothing of it is held in registers — or spilled — across the
// wave's decision what to do with the record.
struct NextInOut {
    StepOut& o;
    BMO_HD void put(const d3& pos, const d3& dir, double n) const {
        o.next.pos = pos;
        o.next.dir = dir;
        o.next.n = n;
    }
};
template <int KIND, class NextSink>
BMO_HD void interact(const SceneView& S, const RayS& ray, const Hit& X, int li, double lambda, double opl_before, StepOut& o,
                     const NextSink& sink, int ent_override = -1) {
    CObject& ob = S.objects[X.obj];
    o.outcome = OUT_STOP;
    o.hint_obj = o.hint_shape = -1;
    o.det_slot = -1;
    enum { A_STOP = 0, A_REFLECT, A_REFRACT, A_SPLIT, A_SPOT, A_PSF, A_POLARIZER };
    int action = A_STOP;
    int medium = -1;
    int32_t hshape = -1;      // Hint(optic) shape of the refracting part
    int32_t force_hobj = -1;  // composite objects overwrite the hint after the part interaction
    int32_t force_hshape = -1;
    int split_mode = 0;       // 0 thin, 1 plate (refract transmitted child), 2 cube (children n := n_glass)
    switch (ob.kind) {
        case BMO_OBJ_MIRROR: action = A_REFLECT; break;
        case BMO_OBJ_REFRACTIVE:
            action = A_REFRACT;
            medium = ob.medium[0];
            hshape = ob.shape[0];
            break;
        case BMO_OBJ_DOUBLET: {
            if (KIND != BMO_BEAM_RAY) break;  // no PolarizedRay method
            int part = X.shape == ob.shape[0] ? 0 : 1;
            action = A_REFRACT;
            medium = ob.medium[part];
            hshape = ob.shape[part];
            force_hobj = X.obj;
            force_hshape = ob.shape[1 - part];
            break;
        }
        case BMO_OBJ_THIN_BS: action = A_SPLIT; break;
        case BMO_OBJ_PLATE_BS:
            if (X.shape == ob.shape[0]) {
                action = A_REFRACT;
                medium = ob.medium[0];
                hshape = ob.shape[0];
                force_hobj = X.obj;
                force_hshape = ob.shape[1];
            } else if (X.shape == ob.shape[1]) {
                action = A_SPLIT;
                split_mode = 1;
            }
            break;
        case BMO_OBJ_CUBE_BS:
            if (X.shape == ob.shape[2]) {
                action = A_SPLIT;
                split_mode = 2;
            } else {
                int part = X.shape == ob.shape[0] ? 0 : 1;
                action = A_REFRACT;
                medium = ob.medium[part];
                hshape = ob.shape[part];
                force_hobj = X.obj;
                force_hshape = ob.shape[2];
            }
            break;
        case BMO_OBJ_SPOTDETECTOR: action = A_SPOT; break;
        case BMO_OBJ_PSFDETECTOR:
            if (KIND != BMO_BEAM_POLARIZED) action = A_PSF;
            break;
        case BMO_OBJ_POLARIZER:
            if (KIND == BMO_BEAM_POLARIZED) action = A_POLARIZER;
            break;
        default: break;  // Intersectable.jl:15
    }

    const d3 hp = hit_point(ray, X.t);
    if (action == A_REFLECT || action == A_REFRACT) {
        d3 nd;
        double n_out = ray.n;
        cx j11{-1, 0}, j22{1, 0};
        int32_t ho = -1, hs = -1;
        if (action == A_REFLECT) {
            nd = reflection3d(ray.dir, X.n);
        } else {
            d3 normal = X.n;
            const bool entering = dot3(ray.dir, X.n) < 0;  // isentering AbstractRay.jl:234-237
            double n1, n2;
            if (entering) {
                n1 = ray.n;
                n2 = n_medium(S, medium, li);
                ho = X.obj;
                hs = hshape;
            } else {
                n1 = n_medium(S, medium, li);
                n2 = 1.0;  // refractive_index(system, λ) AbstractSystem.jl:21
                normal = neg3(normal);
            }
            bool tir = false, do_refract = true;
            if (KIND == BMO_BEAM_POLARIZED) {
                double th = angle3d(ray.dir, neg3(normal));
                cx rs, rp, ts, tp;
                fresnel(th, n2 / n1, rs, rp, ts, tp);
                if (isapprox(cabs2(rs), 1.0, 1e-6) && isapprox(cabs2(rp), 1.0, 1e-6)) {  // is_internally_reflected
                    ho = X.obj;
                    hs = hshape;
                    n2 = n_medium(S, medium, li);
                    nd = reflection3d(ray.dir, normal);
                    j11 = cneg(rs);
                    j22 = rp;
                    do_refract = false;
                } else {
                    j11 = ts;
                    j22 = tp;
                }
            }
            if (do_refract) {
                if (!refraction3d(ray.dir, normal, n1, n2, nd, tir)) {
                    o.status |= BMO_NODE_ERR_UNIT;
                    return;
                }
                if (KIND != BMO_BEAM_POLARIZED && tir) {
                    ho = X.obj;
                    hs = hshape;
                    n2 = n_medium(S, medium, li);
                }
            }
            n_out = n2;
        }
        if (KIND == BMO_BEAM_POLARIZED) {
            global_E0(ray.dir, nd, X.n, j11, j22, ray.E0, o.next.E0);
            if (!e0_orthogonal(nd, o.next.E0)) {
                o.status |= BMO_NODE_ERR_ORTHO;
                return;
            }
        }
        sink.put(hp, nd, n_out);
        o.hint_obj = ho;
        o.hint_shape = hs;
        if (force_hobj >= 0) {
            o.hint_obj = force_hobj;
            o.hint_shape = force_hshape;
        }
        o.outcome = OUT_CONTINUE;
        return;
    }
    if (action == A_SPLIT) {
        // children via the public Ray ctor: dir normalised, n = 1 (ThinBeamsplitter.jl:73-106, Rays.jl:32-42)
        const d3 rdir = reflection3d(ray.dir, X.n);
        o.refl.pos = hp;
        o.refl.n = 1.0;
        double next_n = 1.0;
        if (KIND == BMO_BEAM_POLARIZED) {
            BMO_NOUNROLL
            for (int w = 0; w < 2; ++w) {
                const d3 od = w == 0 ? ray.dir : rdir;
                const cx a = w == 0 ? cx{ob.transmittance, 0} : cx{-ob.reflectance, 0};
                const cx b = w == 0 ? cx{ob.transmittance, 0} : cx{ob.reflectance, 0};
                cx e[3];
                global_E0(ray.dir, od, X.n, a, b, ray.E0, e);
                for (int c = 0; c < 3; ++c) {
                    if (w == 0) o.next.E0[c] = e[c];
                    else o.refl.E0[c] = e[c];
                }
            }
        }
        d3 next_dir = normalize_div(ray.dir);
        o.refl.dir = normalize_div(rdir);
        if (KIND == BMO_BEAM_POLARIZED && !(e0_orthogonal(next_dir, o.next.E0) && e0_orthogonal(o.refl.dir, o.refl.E0))) {
            o.status |= BMO_NODE_ERR_ORTHO;
            return;
        }
        if (split_mode == 1) {  // PlateBeamsplitter.jl:203-225
            const double n_opt = n_medium(S, ob.medium[0], li);
            const bool ent = dot3(ray.dir, X.n) < 0;
            // a GaussianBeamlet picks n_t / n_r from the CHIEF ray's isentering (PlateBeamsplitter.jl:245-262)
            const bool ent_n = ent_override < 0 ? ent : (ent_override != 0);
            const double nt = ent_n ? n_opt : 1.0, nr = ent_n ? 1.0 : n_opt;
            d3 nml = X.n;  // refraction3d(ray, n2) AbstractRay.jl:244-253
            if (!ent) nml = {nml.x * -1, nml.y * -1, nml.z * -1};
            d3 nd;
            bool tir;
            if (!refraction3d(ray.dir, nml, ray.n, nt, nd, tir)) {
                o.status |= BMO_NODE_ERR_UNIT;
                return;
            }
            next_n = nt;
            o.refl.n = nr;
            next_dir = normalize_div(nd);  // direction! AbstractRay.jl:83-86
        } else if (split_mode == 2) {  // CubeBeamsplitter.jl:78-84
            const double ng = n_medium(S, ob.medium[0], li);
            next_n = ng;
            o.refl.n = ng;
        }
        sink.put(hp, next_dir, next_n);
        o.outcome = OUT_SPLIT;
        return;
    }
    if (action == A_SPOT) {
        CShape& m = S.shapes[ob.shape[0]];
        d3 loc{hp.x - m.pos[0], hp.y - m.pos[1], hp.z - m.pos[2]};
        o.det[0] = dot3(loc, d3{m.dir[0], m.dir[3], m.dir[6]});
        o.det[1] = dot3(loc, d3{m.dir[2], m.dir[5], m.dir[8]});
        for (int k = 2; k < 9; ++k) o.det[k] = 0.0;
        o.det_slot = ob.detector;
        o.status |= BMO_NODE_DETECTED;
        return;
    }
    if (action == A_PSF) {
        o.det[0] = hp.x; o.det[1] = hp.y; o.det[2] = hp.z;
        o.det[3] = ray.dir.x; o.det[4] = ray.dir.y; o.det[5] = ray.dir.z;
        o.det[6] = opl_before + X.t * ray.n;
        o.det[7] = fabs(dot3(ray.dir, X.n));
        o.det[8] = 2 * 3.141592653589793 / lambda;
        o.det_slot = ob.detector;
        o.status |= BMO_NODE_DETECTED;
        return;
    }
    if (action == A_POLARIZER) {  // JonesCalculus.jl:29-45: P = Q (R J R') Q, Q = I - d d'
        CShape& m = S.shapes[ob.shape[0]];
        CDouble* R = m.dir;
        cx J[9], A[9], P[9], B[9], P2[9];
        for (int i = 0; i < 9; ++i) J[i] = {ob.jones[2 * i], ob.jones[2 * i + 1]};
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                A[3 * i + j] = cadd(cadd(rmul(R[3 * i + 0], J[0 + j]), rmul(R[3 * i + 1], J[3 + j])), rmul(R[3 * i + 2], J[6 + j]));
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                P[3 * i + j] = cadd(cadd(cmulr(A[3 * i + 0], R[3 * j + 0]), cmulr(A[3 * i + 1], R[3 * j + 1])), cmulr(A[3 * i + 2], R[3 * j + 2]));
        double d[3] = {ray.dir.x, ray.dir.y, ray.dir.z}, Q[9];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) Q[3 * i + j] = (i == j ? 1.0 : 0.0) - d[i] * d[j];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                B[3 * i + j] = cadd(cadd(rmul(Q[3 * i + 0], P[0 + j]), rmul(Q[3 * i + 1], P[3 + j])), rmul(Q[3 * i + 2], P[6 + j]));
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                P2[3 * i + j] = cadd(cadd(cmulr(B[3 * i + 0], Q[0 + j]), cmulr(B[3 * i + 1], Q[3 + j])), cmulr(B[3 * i + 2], Q[6 + j]));
        mat3_apply(P2, ray.E0, o.next.E0);
        double nrm = sqrt((cabs2(o.next.E0[0]) + cabs2(o.next.E0[1])) + cabs2(o.next.E0[2]));
        if (isapprox(nrm, ob.cutoff, 0.0)) {
            o.status |= BMO_NODE_BLOCKED;
            return;
        }
        sink.put(hp, ray.dir, ray.n);
        if (!e0_orthogonal(ray.dir, o.next.E0)) {
            o.status |= BMO_NODE_ERR_ORTHO;
            return;
        }
        o.outcome = OUT_CONTINUE;
        return;
    }
}

// ------------------------------------------------------------------ GaussianBeamlet (Gaussian.jl, System.jl:274-318)
// A GaussianBeamlet is three geometric rays (chief, waist, divergence) traced in lock-step; one lane carries
// all three.  Beam parameters are reconstructed only when a splitter spawns children (gauss_parameters).
struct GaussIn {
    RayS c, w, d;
    int32_t hint_obj, hint_shape;
    double lenA;   // sum of the chief's own segment lengths before this one, folded from 0 (length_rays, Beam.jl:160-169)
    double lenB;   // length(parent chief) + the same sum, folded in that order (point_on_beam's `temp`, Beam.jl:180-190)
    double l0;     // length(parent chief beam) (0 for roots)
    double oplC, oplW, oplD;  // optical path lengths up to this segment's start (chief incl. parents; waist/div own rays only)
    double lambda, w0;
    cx E0;
    int li;
};
// The accumulators and beam constants of a beamlet record, without its rays
struct GaussAcc {
    double lenA, lenB, oplC, oplW, oplD;
    double lambda, l0, w0;
    cx E0;
    int li;
};
// interact's sink (NextInOut) of a beamlet's ray r: straight into the record
template <class Rec>
struct GaussNextSink {
    Rec& rec;
    int r;
    BMO_HD void put(const d3& pos, const d3& dir, double n) const {
        RayS x;
        x.pos = pos;
        x.dir = dir;
        x.n = n;
        rec.put_next(r, x);
    }
};
struct GaussOut {
    int outcome, status;
    Hit Xc, Xw, Xd;                // filled by the struct-backed wrapper only
    int32_t hit_obj, hit_shape;    // object / shape the chief ray hit (-1: none)
    int32_t hint_obj, hint_shape;
    RayS nc, nw, nd;  // next segment, or transmitted child   } filled by the struct-backed wrapper (gauss_step) only: the record-backed
    RayS rc, rw, rd;  // reflected child                      } form hands the rays to Rec::put_next / put_refl as they are computed
    int det_slot, n_det;
    double* det;  // destination of the detector record(s) of this beamlet, 3 x 9 doubles (the node's hit slot): written in place
    double child_w0, child_l0;
    cx Et, Er;
    double oplC, oplW, oplD, lenA, lenB;  // accumulators for the next segment
};

// line_plane_distance3d LinearAlgebraUtils.jl:127-136 (NaN stands for `nothing`)
BMO_HD double line_plane_distance3d(const d3& pp, const d3& pn, const d3& lp, const d3& ld) {
    double denom = dot3(pn, ld);
    if (fabs(denom) > 1e-6) return dot3(sub3(pp, lp), pn) / denom;
    return kinf() - kinf();
}

// gauss_parameters(gauss, length(gauss)) Gaussian.jl:298-353, evaluated at the splitter hit (last segment);
// point_on_beam (Beam.jl:177-205) reduces to p0 = pos + (t - temp)*dir on the last chief segment.  Returns w0.
BMO_HD double gauss_w0_at(const GaussIn& g, double t_total, double temp) {
    const double b = t_total - temp;
    const d3 p0 = axpy3(g.c.pos, b, g.c.dir);
    double y_d, m_d, y_w, m_w;
    {
        double il = line_plane_distance3d(p0, g.c.dir, g.d.pos, g.d.dir);
        d3 y0{g.d.pos.x + il * g.d.dir.x - p0.x, g.d.pos.y + il * g.d.dir.y - p0.y, g.d.pos.z + il * g.d.dir.z - p0.z};
        y_d = norm3(y0);
        y0 = {y0.x / y_d, y0.y / y_d, y0.z / y_d};
        m_d = jl::tan(3.141592653589793 / 2 - angle3d(y0, g.d.dir));
    }
    {
        double il = line_plane_distance3d(p0, g.c.dir, g.w.pos, g.w.dir);
        d3 y0{g.w.pos.x + il * g.w.dir.x - p0.x, g.w.pos.y + il * g.w.dir.y - p0.y, g.w.pos.z + il * g.w.dir.z - p0.z};
        y_w = norm3(y0);
        y0 = {y0.x / y_w, y0.y / y_w, y0.z / y_w};
        m_w = jl::tan(3.141592653589793 / 2 - angle3d(y0, g.w.dir));
    }
    const double n = g.c.n;
    double H = fabs(n * (y_w * m_d - y_d * m_w));
    if (!isapprox(H, g.lambda / 3.141592653589793, 1e-6)) H = g.lambda / 3.141592653589793;
    const double F = sqrt(m_d * m_d + m_w * m_w);
    const double w = sqrt(y_d * y_d + y_w * y_w);
    double w0 = H / (n * F);
    if (isnan_(w0)) w0 = w;
    return w0;
}


// gauss_parameters(gauss, z; hint = (p0, index)) Gaussian.jl:298-353 for the chief / waist / divergence rays of segment `index`
// and the point p0 on the chief ray: w, R (curvature 1/r), psi (Gouy phase, -atan convention), w0 (local waist).
BMO_HD void gauss_parameters_at(const RayS& c, const RayS& wr, const RayS& dr, const d3& p0, double lambda, double& w, double& R, double& psi,
                                double& w0) {
    double y_d, m_d, y_w, m_w;
    {
        double il = line_plane_distance3d(p0, c.dir, dr.pos, dr.dir);
        d3 y0{dr.pos.x + il * dr.dir.x - p0.x, dr.pos.y + il * dr.dir.y - p0.y, dr.pos.z + il * dr.dir.z - p0.z};
        y_d = norm3(y0);
        y0 = {y0.x / y_d, y0.y / y_d, y0.z / y_d};
        m_d = jl::tan(3.141592653589793 / 2 - angle3d(y0, dr.dir));
    }
    {
        double il = line_plane_distance3d(p0, c.dir, wr.pos, wr.dir);
        d3 y0{wr.pos.x + il * wr.dir.x - p0.x, wr.pos.y + il * wr.dir.y - p0.y, wr.pos.z + il * wr.dir.z - p0.z};
        y_w = norm3(y0);
        y0 = {y0.x / y_w, y0.y / y_w, y0.z / y_w};
        m_w = jl::tan(3.141592653589793 / 2 - angle3d(y0, wr.dir));
    }
    const double n = c.n;
    double H = fabs(n * (y_w * m_d - y_d * m_w));
    if (!isapprox(H, lambda / 3.141592653589793, 1e-6)) H = lambda / 3.141592653589793;
    const double E_kt = y_d * m_d + y_w * m_w;
    const double F_kt = sqrt(m_d * m_d + m_w * m_w);
    w = sqrt(y_d * y_d + y_w * y_w);
    R = E_kt / (w * w);
    const double zz = E_kt / (F_kt * F_kt);
    psi = -jl::atan2(1.0, sqrt(1 / (R * zz) - 1));
    w0 = H / (n * F_kt);
    if (isnan_(R)) R = 0;
    if (isnan_(psi)) psi = 0;
    if (isnan_(w0)) w0 = w;
    if (R < 0) psi = -psi;
}

// by-value selection of one of three rays / hits with scalar selects (a reference select `r == 0 ? g.c : ...` forces the
// structs into scratch memory: the first Gaussian kernel spilled 1.8 KB per lane)
BMO_HD double sel3(int r, double a, double b, double c) { return r == 0 ? a : (r == 1 ? b : c); }
BMO_HD RayS pick_ray(int r, const RayS& a, const RayS& b, const RayS& c) {
    RayS o;
    o.pos = {sel3(r, a.pos.x, b.pos.x, c.pos.x), sel3(r, a.pos.y, b.pos.y, c.pos.y), sel3(r, a.pos.z, b.pos.z, c.pos.z)};
    o.dir = {sel3(r, a.dir.x, b.dir.x, c.dir.x), sel3(r, a.dir.y, b.dir.y, c.dir.y), sel3(r, a.dir.z, b.dir.z, c.dir.z)};
    o.n = sel3(r, a.n, b.n, c.n);
    return o;
}
BMO_HD Hit pick_hit(int r, const Hit& a, const Hit& b, const Hit& c) {
    Hit o;
    o.t = sel3(r, a.t, b.t, c.t);
    o.n = {sel3(r, a.n.x, b.n.x, c.n.x), sel3(r, a.n.y, b.n.y, c.n.y), sel3(r, a.n.z, b.n.z, c.n.z)};
    o.obj = r == 0 ? a.obj : (r == 1 ? b.obj : c.obj);
    o.shape = r == 0 ? a.shape : (r == 1 ? b.shape : c.shape);
    return o;
}

// RETR builds: with `probe` the three stored rays are first re-intersected with the hinted shape or the stored object
// (retrace_system! System.jl:360-375; all three are always evaluated), and the stored path holds only if all three still
// hit the same shape (:377-392).  Otherwise the beamlet is cut here and trace_system! goes on from these rays without a
// hint (System.jl:274-318) when `fresh_allowed`.
//
// The beamlet's record is reached through `Rec` (ray(r), put_hit(r, X), hit(r), clear_hits(), hint_obj(), hint_shape(), load()):
// a ray is fetched when its march starts and its hit is put back when the march ends, so nothing of the beamlet but a few integers
// is live across the three sphere-tracing marches.  On the GPU the backing store is the record itself (HBM / L2); held in
// registers, the 3 rays + 3 hits were spilled around every march (1.5 KB of scratch per lane, 5 x the algorithmic HBM traffic).
//
// `tail` (retrace): the stored rays still attached behind the one a re-walking beamlet is at — more() of them, chief hit length t(q)
// (+Inf: no intersection) and rays ray(q, 0 | 1 | 2) of the q-th.  A beamlet that meets a splitter BEFORE the end of its stored path has
// its children sized by gauss_parameters(gauss, length(gauss)) (ThinBeamsplitter.jl:125) while that stale tail is still part of the
// beamlet: retrace_system! deletes it only after its loop (System.jl:417-421).  length() and point_on_beam() then run over the new rays up
// to this one AND the stale ones behind it (Beam.jl:125-169, 177-205), and the waist comes from the rays point_on_beam lands on.
struct NoStaleTail {
    BMO_HD int more() const { return 0; }
    BMO_HD double t(int) const { return kinf(); }
    BMO_HD RayS ray(int, int) const { return RayS{}; }
};
template <int EXT, bool RETR, class Rec, class Tail = NoStaleTail>
BMO_HD void gauss_step_rec(const SceneView& S, Rec& rec, GaussOut& o, uint32_t& calls, ChildCache& cc, const LaneMem& lm, bool probe = false,
                           int32_t probe_obj = -1, bool fresh_allowed = true, bool* probe_missed = nullptr, const Tail& tail = Tail()) {
    o.outcome = OUT_MISS;
    o.status = 0;
    o.hint_obj = o.hint_shape = -1;
    o.det_slot = -1;
    o.n_det = 0;
    o.hit_obj = o.hit_shape = -1;
    // chief, waist, divergence in that order; stop at the first ray without intersection (System.jl:283-296).  Every path below leaves all
    // three hit records written: a marched ray writes its own, the rays behind a miss are written as `no hit` there (clearing all three up
    // front and overwriting them cost 96 B of HBM writes per beamlet and level)
    bool all_hit = true;
    bool held = RETR && probe;  // the stored path holds at this ray (set back when the probe fails)
    int32_t hint_obj = rec.hint_obj(), hint_shape = rec.hint_shape();
    int32_t sh0 = -1, sh1 = -1, sh2 = -1, ob0 = -1;
    BMO_NOUNROLL
    for (int phase = (RETR && probe) ? 0 : 1; phase < 2; ++phase) {
        const bool probing = RETR && phase == 0;
        all_hit = true;
        sh0 = sh1 = sh2 = ob0 = -1;
        BMO_NOUNROLL
        for (int r = 0; r < 3; ++r) {
            const RayS ray = rec.ray(r);
            Hit X = tracing_step<EXT, RETR>(S, ray.pos, ray.dir, hint_obj, hint_shape, calls, cc, lm, probing, probe_obj, false, nullptr);
            rec.put_hit(r, X);
            if (r == 0) {
                sh0 = X.shape;
                ob0 = X.obj;
            } else if (r == 1) {
                sh1 = X.shape;
            } else {
                sh2 = X.shape;
            }
            if (!probing && X.shape < 0) {
                all_hit = false;
                for (int q = r + 1; q < 3; ++q) rec.put_hit(q, no_hit());
                break;
            }
        }
        if (probing) {
            // the stored path holds only if all three rays still hit the same shape (System.jl:377-392)
            if (sh0 >= 0 && sh1 >= 0 && sh2 >= 0 && sh0 == sh1 && sh1 == sh2) break;
            if (probe_missed) *probe_missed = true;
            held = false;
            rec.clear_hits();
            sh0 = sh1 = sh2 = ob0 = -1;
            if (!fresh_allowed) {
                o.status = BMO_NODE_RMAX;
                return;
            }
            hint_obj = hint_shape = -1;
        }
    }
    o.hit_obj = ob0;
    o.hit_shape = sh0;
    if (!all_hit) {
        o.status = BMO_NODE_MISS;
        return;
    }
    if (!(sh0 == sh1 && sh1 == sh2)) {  // _beams_hits_same_shape Gaussian.jl:171-180
        rec.clear_hits();
        o.hit_obj = o.hit_shape = -1;
        o.status = BMO_NODE_GAUSS_DIVERGED;
        return;
    }
    // From here on the record is read piece by piece, each where it is needed (the accumulators now, a ray when its interaction comes, all
    // three once more for a splitter's waist): the whole beamlet held in registers across the three interactions is what the fused
    // 168-register kernel spilled.
    const GaussAcc a = rec.acc();
    const Hit Xc = rec.hit(0);     // t and normal of the chief's hit
    const double tw = rec.hit(1).t, td = rec.hit(2).t;
    const int32_t oid = ob0;
    CObject& ob = S.objects[oid];
    if (ob.kind == BMO_OBJ_PHOTODETECTOR) {  // Photodetector.jl:69-107: record the hit (field read-out is a separate pass), stop
        const RayS c = rec.ray(0);
        o.det_slot = ob.detector;
        for (int q = 0; q < 27; ++q) o.det[q] = 0.0;
        o.det[0] = fabs(dot3(c.dir, Xc.n));  // proj = abs(dot(d0, normal3d(ray_int)))
        o.n_det = 3;
        o.status |= BMO_NODE_DETECTED;
        o.lenA = a.lenA + Xc.t;
        o.lenB = a.lenB + Xc.t;
        o.oplC = a.oplC + Xc.t * c.n;
        o.oplW = a.oplW + tw * rec.ray(1).n;
        o.oplD = a.oplD + td * rec.ray(2).n;
        o.outcome = OUT_STOP;
        return;
    }
    const bool coating = ob.kind == BMO_OBJ_THIN_BS || (ob.kind == BMO_OBJ_PLATE_BS && sh0 == ob.shape[1]) || (ob.kind == BMO_OBJ_CUBE_BS && sh0 == ob.shape[2]);
    // every sub-beam interacts with the object found by the CHIEF ray (System.jl:306-309, Gaussian.jl:124-135)
    bool all_continue = true;
    int entering_hint = 0;
    double n_c = 0.0, n_w = 0.0, n_d = 0.0;  // refractive index along the three rays of this segment
    BMO_NOUNROLL
    for (int r = 0; r < 3; ++r) {
        const RayS ray = rec.ray(r);
        if (r == 0) {
            entering_hint = dot3(ray.dir, Xc.n) < 0 ? 1 : 0;
            n_c = ray.n;
        } else if (r == 1) {
            n_w = ray.n;
        } else {
            n_d = ray.n;
        }
        Hit X = rec.hit(r);
        X.obj = oid;
        X.shape = sh0;
        const double opl = r == 0 ? a.oplC : (r == 1 ? a.oplW : a.oplD);
        StepOut so;
        so.status = 0;
        so.det = o.det + 9 * r;
        // the next segment (or the transmitted child) of ray r goes to the record's sink inside the interaction, its reflected child right
        // behind it: held in registers until the slot allocation, the six rays of a beamlet were what the 168-register kernel spilled
        interact<BMO_BEAM_RAY>(S, ray, X, a.li, a.lambda, opl, so, GaussNextSink<Rec>{rec, r}, entering_hint);
        o.status |= so.status;
        if (so.outcome == OUT_SPLIT) rec.put_refl(r, so.refl);
        if (r == 0) {
            o.hint_obj = so.hint_obj;
            o.hint_shape = so.hint_shape;
        }
        if (so.det_slot >= 0) {
            o.det_slot = so.det_slot;
            o.n_det = r + 1;  // (the record went straight to o.det + 9 * r)
        }
        const int want = coating ? OUT_SPLIT : OUT_CONTINUE;
        if (so.outcome != want) all_continue = false;
    }
    o.oplC = a.oplC + Xc.t * n_c;
    o.oplW = a.oplW + tw * n_w;
    o.oplD = a.oplD + td * n_d;
    o.lenA = a.lenA + Xc.t;
    o.lenB = a.lenB + Xc.t;
    if (!all_continue) {
        o.outcome = OUT_STOP;
        return;
    }
    if (!coating) {
        o.outcome = OUT_CONTINUE;
        return;
    }
    // splitter: ThinBeamsplitter.jl:117-168 (+ PlateBeamsplitter.jl:230-275, CubeBeamsplitter.jl:94-121 via interact's split modes)
    const GaussIn g = rec.load();
    const double len_here = g.lenA + Xc.t;  // length_rays up to this hit (Beam.jl:160-169)
    double w0;
    const int more = (RETR && held) ? tail.more() : 0;
    if (RETR && more > 0) {
        // the stale tail is still attached (see `tail` above): length(gauss) and point_on_beam run over it too
        double l = len_here;
        BMO_NOUNROLL
        for (int q = 0; q < more; ++q) {
            const double tq = tail.t(q);
            if (!(tq < kinf())) break;  // a ray without intersection ends length_rays
            l += tq;
        }
        const double t_total = l + g.l0;
        // point_on_beam(chief, t_total) Beam.jl:177-205: rays 1..i-1 are in lenB, ray i is this one, then the stale ones; the last ray
        // of the beam is never tested (index == numEl: break)
        double temp = g.lenB + Xc.t;
        int at = -2;  // -1: this ray; q >= 0: stale ray q
        d3 p0{0, 0, 0};
        if (t_total < temp) {
            const double b = temp - t_total;
            p0 = axpy3(g.c.pos, Xc.t - b, g.c.dir);
            at = -1;
        }
        BMO_NOUNROLL
        for (int q = 0; q + 1 < more && at == -2; ++q) {
            const double len = tail.t(q);
            temp += len;
            if (t_total < temp) {
                const double b = temp - t_total;
                const RayS cq = tail.ray(q, 0);
                p0 = axpy3(cq.pos, len - b, cq.dir);
                at = q;
            }
        }
        if (at == -2) {
            at = more - 1;
            const RayS cq = tail.ray(at, 0);
            p0 = axpy3(cq.pos, t_total - temp, cq.dir);
        }
        const RayS pc = at < 0 ? g.c : tail.ray(at, 0), pw = at < 0 ? g.w : tail.ray(at, 1), pd = at < 0 ? g.d : tail.ray(at, 2);
        double w_, R_, psi_;
        gauss_parameters_at(pc, pw, pd, p0, g.lambda, w_, R_, psi_, w0);
    } else {
        const double t_total = len_here + g.l0;  // length(gauss) = length_rays + length_parent (Beam.jl:125-130)
        w0 = gauss_w0_at(g, t_total, g.lenB);
    }
    o.child_w0 = w0;
    o.child_l0 = len_here + g.l0;  // length(parent chief) as the children will see it: the tail is gone by then
    const double ratio = g.w0 / w0;
    o.Et = cmulr(rmul(ob.transmittance, g.E0), ratio);
    o.Er = cmulr(rmul(ob.reflectance, g.E0), ratio);
    const double df = dot3(g.c.dir, Xc.n);
    const cx ph = df < 0 ? cx{-1.0, 1.2246467991473532e-16} : cx{1.0, 0.0};  // exp(im*π) | exp(im*0)
    o.Er = cmul(o.Er, ph);
    o.outcome = OUT_SPLIT;
}

// struct-backed record (host emulator, tests): same code path as the device's memory-backed one
struct GaussRecLocal {
    const GaussIn& g;
    Hit X[3];
    RayS nxt[3], rfl[3];
    BMO_HD void put_next(int r, const RayS& x) { nxt[r] = x; }
    BMO_HD void put_refl(int r, const RayS& x) { rfl[r] = x; }
    BMO_HD RayS ray(int r) const { return pick_ray(r, g.c, g.w, g.d); }
    BMO_HD void put_hit(int r, const Hit& x) { X[r] = x; }
    BMO_HD Hit hit(int r) const { return X[r]; }
    BMO_HD void clear_hits() { X[0] = X[1] = X[2] = no_hit(); }
    BMO_HD int32_t hint_obj() const { return g.hint_obj; }
    BMO_HD int32_t hint_shape() const { return g.hint_shape; }
    BMO_HD GaussIn load() const { return g; }
    BMO_HD GaussAcc acc() const { return GaussAcc{g.lenA, g.lenB, g.oplC, g.oplW, g.oplD, g.lambda, g.l0, g.w0, g.E0, g.li}; }
};
template <int EXT, bool RETR = false, class Tail = NoStaleTail>
BMO_HD void gauss_step(const SceneView& S, const GaussIn& g, GaussOut& o, uint32_t& calls, ChildCache& cc, const LaneMem& lm, bool probe = false,
                       int32_t probe_obj = -1, bool fresh_allowed = true, bool* probe_missed = nullptr, const Tail& tail = Tail()) {
    GaussRecLocal rec{g, {no_hit(), no_hit(), no_hit()}, {}, {}};
    gauss_step_rec<EXT, RETR, GaussRecLocal, Tail>(S, rec, o, calls, cc, lm, probe, probe_obj, fresh_allowed, probe_missed, tail);
    o.nc = rec.nxt[0];
    o.nw = rec.nxt[1];
    o.nd = rec.nxt[2];
    o.rc = rec.rfl[0];
    o.rw = rec.rfl[1];
    o.rd = rec.rfl[2];
    o.Xc = rec.X[0];
    o.Xw = rec.X[1];
    o.Xd = rec.X[2];
}

}  // namespace bmo
