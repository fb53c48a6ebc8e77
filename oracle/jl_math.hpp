// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
//
// jl_math.hpp — scalar / vector / dual-number arithmetic restated from the Julia
// packages the reference computes with.  Every function states which Julia rule it
// follows.  The reference itself cannot run here (no julia binary) and
// the third-party packages are not vendored under /root/reference, so these rules are
// restated from the packages' published semantics:
//   * ForwardDiff.jl  (Project.toml compat "0.10, 1.0.1"; call site src/SDFs/AbstractSDF.jl:91)
//   * DiffRules.jl    (max/min/abs/sqrt derivative rules used by ForwardDiff's Dual)
//   * GeometryBasics.jl 0.5 (Point2/Point3: norm = sqrt(dot(a,a)), normalize = a ./ norm(a))
//   * StaticArrays.jl 1  (SMatrix * vector: row-wise left-fold of products; cross; dot)
//   * LinearAlgebra   (normalize(::Vector) = rmul!(a, inv(norm(a))))
// Bit-level parity with Julia for these rules is UNPINNED (no reference test pins them
// bit-exactly); parity at the reference's own tolerances is pinned by the
// known-answer tests in tests/test_oracle_kat.py.
#pragma once
#include <cmath>
#include <cstdint>
#include <limits>

// ---------------------------------------------------------------------------
// Rule-set switches for the UNPINNED third-party arithmetic (oracle/rule_table.py builds every combination and runs the transcribed
// reference KATs, the two exact ones at their original assertions, over the product; the table is oracle/RULE_TABLE.md):
//   BMO_RULE_SQRT0  0  sqrt(Dual(0, zeros)) keeps zero partials (ForwardDiff's NaN-safe partial scaling; the default of rounds 1 - 3)
//                   1  0 * Inf = NaN (default ForwardDiff: deriv * partials)           THE DEFAULT  [-DBMO_SQRT_PLAIN is the old spelling]
//   BMO_RULE_TIE    0  max / min(::Dual, ::Real) at EQUAL values and equal zero signs: the Dual operand wins and keeps its partials
//                      (DiffRules: y wins iff (y > x) | (signbit(y) < signbit(x)))                       THE DEFAULT
//                   1  the Real operand wins the tie (zero partials), as an `ifelse(x > y, x, y)` style definition would give
//   BMO_RULE_NORM0  0  norm(a) = sqrt(dot(a, a)) always (GeometryBasics 0.5 fixed_arrays.jl)             THE DEFAULT
//                   1  a vector whose component VALUES are all zero returns before any sqrt, like LinearAlgebra.generic_norm2
//                      (`iszero(maxabs) && return maxabs`): the result is the first component's abs, a Dual(0, its partials)
#if defined(BMO_SQRT_PLAIN) && !defined(BMO_RULE_SQRT0)
#define BMO_RULE_SQRT0 1
#endif
#ifndef BMO_RULE_SQRT0
#define BMO_RULE_SQRT0 1
#endif
#ifndef BMO_RULE_TIE
#define BMO_RULE_TIE 0
#endif
#ifndef BMO_RULE_NORM0
#define BMO_RULE_NORM0 0
#endif
//   BMO_RULE_SELECT 0  max / min of Duals in DiffRules' product form dvx * px + dvy * py: a NaN partial of the LOSING operand poisons the
//                      result (0 * NaN)                                                              (the default of rounds 1 - 3)
//                   1  max / min by SELECTION: the winner's value and partials and nothing else; a Real operand is promoted to a Dual with
//                      zero partials first; ties return the FIRST argument (`ifelse(isless(x, y), y, x)`)
//                   2  selection, ties return the SECOND argument (`x > y ? x : y`, `x < y ? x : y`)                  THE DEFAULT
// Round 4 found the one combination under which EVERY transcribed reference KAT holds at its original assertion, the two exact ones included
// (oracle/RULE_TABLE.md: 67 of 67): SQRT0 = 1, NORM0 = 0, SELECT = 2 with Julia Base's elementary functions.  What decides it: the prism's
// entry face needs the central-difference normal (`norm(max.(q, 0))` of a zero vector is NaN and sits in the WINNING operand of the sdf's
// outer `max`), the concave surfaces need their dual normals (the same NaN sits in the LOSING operand there and must not leak), and at the
// apex of a concave surface the two operands tie — only "the second one wins" keeps the on-axis ray's normal valid.
#ifndef BMO_RULE_SELECT
#define BMO_RULE_SELECT 2
#endif

namespace jl {

// ---------------------------------------------------------------------------
// Julia Base.max / Base.min for Float64 (base/math.jl): NaN-propagating, -0.0 < +0.0.
inline double jmax(double x, double y) {
    if (std::isnan(x) || std::isnan(y)) return std::numeric_limits<double>::quiet_NaN();
    return ((y > x) || (std::signbit(y) < std::signbit(x))) ? y : x;
}
inline double jmin(double x, double y) {
    if (std::isnan(x) || std::isnan(y)) return std::numeric_limits<double>::quiet_NaN();
    return ((y < x) || (std::signbit(y) > std::signbit(x))) ? y : x;
}
inline double jabs(double x) { return std::fabs(x); }
inline double jsqrt(double x) { return std::sqrt(x); }
inline double value(double x) { return x; }

// ---------------------------------------------------------------------------
// ForwardDiff.Dual{T,Float64,3}: value + 3 partials (gradient w.r.t. world x,y,z).
struct Dual {
    double v;
    double p[3];
};
inline double value(const Dual& a) { return a.v; }
inline Dual mkdual(double v, double p0, double p1, double p2) { return Dual{v, {p0, p1, p2}}; }

// +, - : componentwise (ForwardDiff dual.jl @define_binary_dual_op +/-)
inline Dual operator+(const Dual& a, const Dual& b) { return Dual{a.v + b.v, {a.p[0] + b.p[0], a.p[1] + b.p[1], a.p[2] + b.p[2]}}; }
inline Dual operator-(const Dual& a, const Dual& b) { return Dual{a.v - b.v, {a.p[0] - b.p[0], a.p[1] - b.p[1], a.p[2] - b.p[2]}}; }
inline Dual operator+(const Dual& a, double r) { return Dual{a.v + r, {a.p[0], a.p[1], a.p[2]}}; }
inline Dual operator+(double r, const Dual& a) { return Dual{r + a.v, {a.p[0], a.p[1], a.p[2]}}; }
inline Dual operator-(const Dual& a, double r) { return Dual{a.v - r, {a.p[0], a.p[1], a.p[2]}}; }
inline Dual operator-(double r, const Dual& a) { return Dual{r - a.v, {-a.p[0], -a.p[1], -a.p[2]}}; }
inline Dual operator-(const Dual& a) { return Dual{-a.v, {-a.p[0], -a.p[1], -a.p[2]}}; }
// * : Dual*Dual = Dual(vx*vy, _mul_partials(px, py, vy, vx)) with
//     _mul_partials(a, b, x_a, x_b) = a*x_a + b*x_b  (ForwardDiff partials.jl mul_tuples)
inline Dual operator*(const Dual& a, const Dual& b) {
    return Dual{a.v * b.v, {(a.p[0] * b.v) + (b.p[0] * a.v), (a.p[1] * b.v) + (b.p[1] * a.v), (a.p[2] * b.v) + (b.p[2] * a.v)}};
}
inline Dual operator*(const Dual& a, double r) { return Dual{a.v * r, {a.p[0] * r, a.p[1] * r, a.p[2] * r}}; }
inline Dual operator*(double r, const Dual& a) { return Dual{r * a.v, {a.p[0] * r, a.p[1] * r, a.p[2] * r}}; }
// / : Dual/Real = Dual(v/r, partials/r)
inline Dual operator/(const Dual& a, double r) { return Dual{a.v / r, {a.p[0] / r, a.p[1] / r, a.p[2] / r}}; }
// comparisons act on values
inline bool operator<(const Dual& a, const Dual& b) { return a.v < b.v; }
inline bool operator<(const Dual& a, double b) { return a.v < b; }
inline bool operator>(const Dual& a, double b) { return a.v > b; }

// sqrt: DiffRules  d/dx sqrt(x) = inv(2*sqrt(x));  Dual(val, deriv * partials).
// At x = 0 the derivative is Inf.  When ALL partials are zero the product keeps them zero instead of 0*Inf = NaN — the
// behaviour of ForwardDiff's NaN-safe partial scaling (partials.jl: `x = ifelse(!isfinite(x) && iszero(partials), one(x), x)`).
// Which variant the reference's (unpinned) dependency stack runs cannot be read off /root/reference; it is pinned by the
// reference's own known-answer test "Double Gauss lens / point source (narrow)" (test/runtests.jl:2755-2761, regression for its
// issue #11): all 1000 rays, which cross the concave surfaces within 25 um of their apexes, must land within 2e-7 m.  With
// 0*Inf = NaN every concave-surface normal (norm(max.(d, 0)) of a zero vector inside the cylinder part, SphericalLensSDF.jl:159-170)
// falls back to central differences, whose 1e-8 stencil pokes through the < 1e-8-thin wedge near the apex: the spot is 1e-3 m
// and the on-axis ray gets a NaN normal.  With the rule below the dual gradient is valid there: spot 1.3e-7 m, 1000 of 1000 rays.
//
// Where the rule can come from upstream (round 2; written from the packages' published sources as far as they can be restated
// without network access — nothing of them is vendored under /root/reference — so file names are given, line numbers are not):
//   (A) ForwardDiff src/dual.jl builds sqrt(::Dual) from DiffRules' `@define_diffrule Base.sqrt(x) = :(inv(2 * sqrt($x)))` through
//       `dual_definition_retval(::Val{T}, val, deriv, partials) = Dual{T}(val, deriv * partials)`; src/partials.jl has
//       `*(partials::Partials, x::Real) = Partials(scale_tuple(partials.values, x))` and, under `if NANSAFE_MODE_ENABLED`, the
//       variant with `x = ifelse(!isfinite(x) && iszero(partials), one(x), x)`.  NANSAFE_MODE_ENABLED is the preference
//       "nansafe_mode" (src/prefs.jl), default false in 0.10.x and, as far as recalled, in 1.0.x.  Default build: 0 * Inf = NaN.
//   (B) GeometryBasics 0.5 dropped StaticArrays and defines its own fixed-size vectors (src/fixed_arrays.jl); recalled there:
//       `LinearAlgebra.norm(a::StaticVector) = sqrt(dot(a, a))` and `LinearAlgebra.normalize(a::StaticVector) = a ./ norm(a)`.  With
//       these, norm(max.(d, 0)) of a zero vector IS sqrt(Dual(0, zeros)) and (A) decides.  If instead Point2 fell through to
//       LinearAlgebra's generic norm, `generic_norm2` returns `maxabs` as soon as `iszero(maxabs)` — a Dual(0, zeros), no sqrt, no
//       NaN — which is observationally the rule coded below (for non-zero vectors both forms give sqrt of the same left-folded sum).
//   So the reference's own tests decide between "default ForwardDiff + sqrt(dot)" (NaN) and "(A) NaN-safe or (B) generic norm"
//   (zero partials kept).  What NaN would mean, quantitatively: the dual gradient of EVERY point on a concave spherical surface
//   inside its aperture is NaN (d = (r - dia/2, |y + sag/2| - sag/2) has both components <= 0 there), so normal_fd falls back to
//   numeric_gradient with its +-1e-8 stencil.  ConcaveSphericalSurfaceSDF is a cylinder of height sag minus a ball whose top touches
//   the cylinder's cap at the apex, so the glass under the cap is y_cap - y_sphere = r^2 / (2R) thick at distance r from the axis:
//   thinner than the stencil for r < sqrt(2 R 1e-8) (24 um at R = 30 mm).  Inside that disc the +y stencil point leaves the cylinder,
//   max(sdf1, -sdf2) switches branch, and the y difference comes out as -|y| instead of -2e-8: a normal that is wrong by tens of
//   degrees.  The reference's "Double Gauss lens / point source (narrow)" test sends 1000 rays through exactly that disc of several
//   concave surfaces and requires a 2e-7 m spot; measured with this oracle: 1.3e-7 m with zero partials kept, 1e-3 m with NaN
//   (build with -DBMO_SQRT_PLAIN to see it).  That test passing upstream is incompatible with the NaN variant.
//   One reference assertion seemed to point the other way (the cube splitter rotated by pi/2 returning `direction == [0, 1, 0]` exactly,
//   runtests.jl:2629-2630, needs the exactly axis-aligned normal that only the numeric fallback yields on the prism's entry face,
//   where norm(max.(q, 0)) is again the norm of a zero vector), and rounds 1 - 3 concluded "both cannot hold under one rule for the same
//   expression".  ROUND 4: they can — the expression is the same, its PLACE is not.  In the prism the NaN term is the operand that WINS the
//   sdf's outer max(box_dist, pln_dist); in the concave surface it is the operand that LOSES max(sdf1, -sdf2).  With max / min of dual numbers
//   taken by SELECTION (the winner's partials and nothing else) instead of DiffRules' product form (0 * NaN poisons), sqrt may be the
//   default ForwardDiff one (NaN at zero) on GeometryBasics' sqrt(dot): the prism falls back to central differences, the concave surfaces
//   do not, and with ties going to the second argument the on-axis apex (where sdf1 and -sdf2 tie at 0) keeps its dual normal too.  That
//   rule set passes all 67 transcribed KATs at their original assertions (oracle/RULE_TABLE.md row 1) and is the default now; it is applied in
//   the oracle and in the engine alike (csrc/bmo_lane.hpp).  Whether upstream's ForwardDiff really selects is still not something that can be
//   READ here (no package sources); it is what the reference's tests, taken together, leave.
inline Dual jsqrt(const Dual& a) {
    double s = std::sqrt(a.v);
    double d = 1.0 / (2.0 * s);
#if BMO_RULE_SQRT0 == 0
    if (!std::isfinite(d) && a.p[0] == 0 && a.p[1] == 0 && a.p[2] == 0) return Dual{s, {a.p[0], a.p[1], a.p[2]}};
#endif
    return Dual{s, {a.p[0] * d, a.p[1] * d, a.p[2] * d}};
}
// abs: ForwardDiff dual.jl  Base.abs(d::Dual) = signbit(value(d)) ? -d : d
inline Dual jabs(const Dual& a) { return std::signbit(a.v) ? -a : a; }

// max/min.  BMO_RULE_SELECT = 0 (rounds 1 - 3) is the form one gets when ForwardDiff builds them from DiffRules' binary rules:
//   Dual(max(vx,vy), _mul_partials(px, py, dvx, dvy)),  (dvx,dvy) in {(1,0),(0,1)}
//   DiffRules:  max: y wins iff (y > x) | (signbit(y) < signbit(x));  min: (y < x) | (signbit(y) > signbit(x))
// with the multiplication by 0/1: a NaN/Inf partial of the LOSING operand contaminates the result (0*NaN = NaN).
// BMO_RULE_SELECT = 2 (the default since round 4, see above jsqrt): plain selection.
#if BMO_RULE_SELECT == 2  // selection, ties to the SECOND argument: `x > y ? x : y` / `x < y ? x : y`
inline Dual jmax(const Dual& x, const Dual& y) { return x.v > y.v ? x : y; }
inline Dual jmin(const Dual& x, const Dual& y) { return x.v < y.v ? x : y; }
inline Dual jmax(const Dual& x, double y) { return jmax(x, Dual{y, {0, 0, 0}}); }
inline Dual jmin(const Dual& x, double y) { return jmin(x, Dual{y, {0, 0, 0}}); }
#elif BMO_RULE_SELECT == 1
inline Dual jmax(const Dual& x, const Dual& y) { return (x.v < y.v || (x.v == y.v && std::signbit(x.v) > std::signbit(y.v))) ? y : x; }  // isless(x, y) ? y : x
inline Dual jmin(const Dual& x, const Dual& y) { return (y.v < x.v || (x.v == y.v && std::signbit(y.v) > std::signbit(x.v))) ? y : x; }  // isless(y, x) ? y : x
inline Dual jmax(const Dual& x, double y) { return jmax(x, Dual{y, {0, 0, 0}}); }
inline Dual jmin(const Dual& x, double y) { return jmin(x, Dual{y, {0, 0, 0}}); }
#else
inline Dual jmax(const Dual& x, const Dual& y) {
    bool ywins = (y.v > x.v) || (std::signbit(y.v) < std::signbit(x.v));
    double dx = ywins ? 0.0 : 1.0, dy = ywins ? 1.0 : 0.0;
    return Dual{jmax(x.v, y.v), {(x.p[0] * dx) + (y.p[0] * dy), (x.p[1] * dx) + (y.p[1] * dy), (x.p[2] * dx) + (y.p[2] * dy)}};
}
inline Dual jmin(const Dual& x, const Dual& y) {
    bool ywins = (y.v < x.v) || (std::signbit(y.v) > std::signbit(x.v));
    double dx = ywins ? 0.0 : 1.0, dy = ywins ? 1.0 : 0.0;
    return Dual{jmin(x.v, y.v), {(x.p[0] * dx) + (y.p[0] * dy), (x.p[1] * dx) + (y.p[1] * dy), (x.p[2] * dx) + (y.p[2] * dy)}};
}
// Dual vs Real: Dual(val, dvx * partials(x))
inline Dual jmax(const Dual& x, double y) {
    bool ywins = (y > x.v) || (std::signbit(y) < std::signbit(x.v));
#if BMO_RULE_TIE == 1
    if (y == x.v && std::signbit(y) == std::signbit(x.v)) ywins = true;
#endif
    double dx = ywins ? 0.0 : 1.0;
    return Dual{jmax(x.v, y), {x.p[0] * dx, x.p[1] * dx, x.p[2] * dx}};
}
inline Dual jmin(const Dual& x, double y) {
    bool ywins = (y < x.v) || (std::signbit(y) > std::signbit(x.v));
#if BMO_RULE_TIE == 1
    if (y == x.v && std::signbit(y) == std::signbit(x.v)) ywins = true;
#endif
    double dx = ywins ? 0.0 : 1.0;
    return Dual{jmin(x.v, y), {x.p[0] * dx, x.p[1] * dx, x.p[2] * dx}};
}
#endif

// ---- additional Dual rules used by the aspheric / acylindric SDFs (ForwardDiff dual.jl) ----------------------
// Dual / Dual: Dual(vx/vy, _div_partials(px, py, vx, vy)), _div_partials(a, b, av, bv) = _mul_partials(a, b, inv(bv), -(av/(bv*bv)))
inline Dual operator/(const Dual& x, const Dual& y) {
    double ia = 1.0 / y.v, fb = -(x.v / (y.v * y.v));
    return Dual{x.v / y.v, {(x.p[0] * ia) + (y.p[0] * fb), (x.p[1] * ia) + (y.p[1] * fb), (x.p[2] * ia) + (y.p[2] * fb)}};
}
// Real / Dual: divv = x / v; Dual(divv, -(divv / v) * partials(y))
inline Dual operator/(double x, const Dual& y) {
    double divv = x / y.v, f = -(divv / y.v);
    return Dual{divv, {y.p[0] * f, y.p[1] * f, y.p[2] * f}};
}
inline bool operator>(const Dual& a, const Dual& b) { return a.v > b.v; }
inline bool operator<(double a, const Dual& b) { return a < b.v; }
inline bool operator>(double a, const Dual& b) { return a > b.v; }
// literal_pow(^, x::Dual, Val(y)), y = 2, 3: Dual(v^y, (y * v^(y-1)) * partials)
inline double lit2(double x) { return x * x; }
inline double lit3(double x) { return x * x * x; }
inline Dual lit2(const Dual& x) {
    double d = 2 * x.v;
    return Dual{x.v * x.v, {x.p[0] * d, x.p[1] * d, x.p[2] * d}};
}
inline Dual lit3(const Dual& x) {
    double d = 3 * (x.v * x.v);
    return Dual{x.v * x.v * x.v, {x.p[0] * d, x.p[1] * d, x.p[2] * d}};
}
// x^n, n a run-time Int >= 1: power by squaring for reals; Dual: Dual(v^n, (partials * n) * v^(n-1)), zero partials short-cut
inline double ipow(double x, int n) {
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
inline Dual ipow(const Dual& x, int n) {
    double ev = ipow(x.v, n);
    if (n == 0 || (x.p[0] == 0 && x.p[1] == 0 && x.p[2] == 0)) return Dual{ev, {0, 0, 0}};
    double f = ipow(x.v, n - 1);
    return Dual{ev, {(x.p[0] * n) * f, (x.p[1] * n) * f, (x.p[2] * n) * f}};
}
inline bool jisnan(double x) { return std::isnan(x); }
inline bool jisnan(const Dual& x) { return std::isnan(x.v); }
inline double jsignv(double x) { return x > 0 ? 1.0 : (x < 0 ? -1.0 : x); }
inline double jsignv(const Dual& x) { return jsignv(x.v); }
// clamp(x, 0.0, 1.0) (Base): ifelse(x > hi, hi, ifelse(x < lo, lo, x)); constants carry zero partials
inline double clamp01(double x) { return x > 1.0 ? 1.0 : (x < 0.0 ? 0.0 : x); }
inline Dual clamp01(const Dual& x) { return x.v > 1.0 ? Dual{1.0, {0, 0, 0}} : (x.v < 0.0 ? Dual{0.0, {0, 0, 0}} : x); }
inline Dual mknan(const Dual&) { double n = std::numeric_limits<double>::quiet_NaN(); return Dual{n, {n, n, n}}; }
inline double mknan(double) { return std::numeric_limits<double>::quiet_NaN(); }

// ---------------------------------------------------------------------------
template <class T>
struct V3 {
    T x, y, z;
};
template <class T>
struct V2 {
    T x, y;
};
using D3 = V3<double>;

// elementwise ops (GeometryBasics Point broadcasting)
template <class T> inline V3<T> operator+(const V3<T>& a, const V3<T>& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <class T> inline V3<T> operator-(const V3<T>& a, const V3<T>& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <class T> inline V3<T> operator-(const V3<T>& a) { return {-a.x, -a.y, -a.z}; }
template <class T> inline V3<T> operator*(double s, const V3<T>& a) { return {s * a.x, s * a.y, s * a.z}; }
template <class T> inline V2<T> operator-(const V2<T>& a, const V2<T>& b) { return {a.x - b.x, a.y - b.y}; }

// dot: left fold a1*b1 + a2*b2 + a3*b3
template <class T> inline T dot(const V3<T>& a, const V3<T>& b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
template <class T> inline T dot(const V2<T>& a, const V2<T>& b) { return a.x * b.x + a.y * b.y; }
// norm = sqrt(dot(a,a))   (BMO_RULE_NORM0 == 1: an all-zero vector returns abs of its first component before any sqrt)
template <class T> inline T norm(const V3<T>& a) {
#if BMO_RULE_NORM0 == 1
    if (value(a.x) == 0 && value(a.y) == 0 && value(a.z) == 0) return jabs(a.x);
#endif
    return jsqrt(dot(a, a));
}
template <class T> inline T norm(const V2<T>& a) {
#if BMO_RULE_NORM0 == 1
    if (value(a.x) == 0 && value(a.y) == 0) return jabs(a.x);
#endif
    return jsqrt(dot(a, a));
}
// cross (StaticArrays / GeometryBasics): (a2*b3-a3*b2, a3*b1-a1*b3, a1*b2-a2*b1)
inline D3 cross(const D3& a, const D3& b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
// normalize(::Point3) = a ./ norm(a)  (GeometryBasics 0.5)
inline D3 normalize_pt(const D3& a) {
    double n = norm(a);
    return {a.x / n, a.y / n, a.z / n};
}
// normalize(::Vector) = a * inv(norm(a))  (LinearAlgebra.__normalize!, used on ForwardDiff.gradient's Vector result)
inline D3 normalize_vec(const D3& a) {
    double n = norm(a);
    double inv = 1.0 / n;
    return {a.x * inv, a.y * inv, a.z * inv};
}

// 3x3 row-major matrix times vector: each row left-folded (StaticArrays _mul)
template <class T>
inline V3<T> matvec(const double* m, const V3<T>& v) {
    return {(m[0] * v.x + m[1] * v.y) + m[2] * v.z, (m[3] * v.x + m[4] * v.y) + m[5] * v.z, (m[6] * v.x + m[7] * v.y) + m[8] * v.z};
}

// isapprox(x, y; atol, rtol) for reals: |x-y| <= max(atol, rtol*max(|x|,|y|));
// default rtol = sqrt(eps) only when atol == 0 (Base floatfuncs.jl)
inline bool isapprox(double x, double y, double atol = 0.0) {
    double rtol = atol > 0.0 ? 0.0 : 1.4901161193847656e-08;
    if (x == y) return true;
    if (!std::isfinite(x) || !std::isfinite(y)) return false;
    return std::fabs(x - y) <= std::fmax(atol, rtol * std::fmax(std::fabs(x), std::fabs(y)));
}

// simple complex with textbook formulas (shared convention with the device code; Julia's
// Complex division/sqrt use scaled algorithms that differ in the last ulp — unpinned)
struct Cx {
    double re, im;
};
inline Cx operator+(Cx a, Cx b) { return {a.re + b.re, a.im + b.im}; }
inline Cx operator-(Cx a, Cx b) { return {a.re - b.re, a.im - b.im}; }
inline Cx operator-(Cx a) { return {-a.re, -a.im}; }
inline Cx operator*(Cx a, Cx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
inline Cx operator*(double s, Cx a) { return {s * a.re, s * a.im}; }
inline Cx operator*(Cx a, double s) { return {a.re * s, a.im * s}; }
inline Cx operator/(Cx a, Cx b) {
    double den = b.re * b.re + b.im * b.im;
    return {(a.re * b.re + a.im * b.im) / den, (a.im * b.re - a.re * b.im) / den};
}
inline double abs2(Cx a) { return a.re * a.re + a.im * a.im; }
inline Cx csqrt_real(double x) { return x >= 0.0 ? Cx{std::sqrt(x), 0.0} : Cx{0.0, std::sqrt(-x)}; }

}  // namespace jl
