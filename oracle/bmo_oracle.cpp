// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
//
// bmo_oracle.cpp — single-threaded-per-beam CPU restatement of BeamletOptics.jl's
// solve_system!/trace loop (reference @ v0.10.2), op for op, INCLUDING the reference's
// cost behaviour (a missed SDF burns march_iters = 1000 evaluations, brute-force
// trace_all, array-of-structs beams, recursion over the beam tree).  It is the checker
// for the HIP engine and the "port" CPU baseline of bench.py; nothing in the product
// path links or calls it.
//
// Parity status: the reference cannot be executed in this environment (no julia
// toolchain), so this restatement is pinned by the reference's own
// known-answer tests (test/runtests.jl, transcribed in tests/test_oracle_kat.py), not by
// Julia-generated vectors.  Third-party arithmetic rules are restated in jl_math.hpp.
//
// Each function cites the reference file:line it follows (paths relative to
// /root/reference/src unless noted).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <deque>
#include <limits>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "../include/bmo.h"
#include "jl_math.hpp"
#include "jl_trig.hpp"

// The elementary functions of the step path: Julia Base's own (jl_trig.hpp) — or, for the rule table only (-DBMO_RULE_LIBM=1), the C library's,
// which is what rounds 1 - 3 used and what fails runtests.jl:157.
namespace trig {
#if defined(BMO_RULE_LIBM) && BMO_RULE_LIBM
inline double sin(double x) { return std::sin(x); }
inline double cos(double x) { return std::cos(x); }
inline double tan(double x) { return std::tan(x); }
inline double acos(double x) { return std::acos(x); }
inline double atan2(double y, double x) { return std::atan2(y, x); }
#else
using jlm::acos;
using jlm::atan2;
using jlm::cos;
using jlm::sin;
using jlm::tan;
#endif
}  // namespace trig

using namespace jl;

namespace {

thread_local std::string g_err;
constexpr double kInf = std::numeric_limits<double>::infinity();
constexpr int kErrOrtho = 256;  // PolarizedRays.jl:54-56 (E0 not orthogonal to dir)

// ------------------------------------------------------------------ scene
struct Scene {
    std::vector<bmo_object> objects;
    std::vector<bmo_shape> shapes;
    std::vector<int32_t> children;
    std::vector<double> tris;
    std::vector<double> n_table;
    std::vector<double> lambdas;
    std::vector<double> coefs;
    int n_lambda = 0, n_detectors = 0;
    double eps_srf, eps_ray, eps_ins, mt_keps, mt_leps, grad_h;
    int march_iters;
};

bool build_scene(const bmo_scene_desc* d, Scene& S) {
    if (!d || d->abi_version != BMO_ABI_VERSION) {
        g_err = "bad descriptor / abi version";
        return false;
    }
    S.objects.assign(d->objects, d->objects + d->n_objects);
    S.shapes.assign(d->shapes, d->shapes + d->n_shapes);
    S.children.assign(d->children, d->children + d->n_children);
    S.tris.assign(d->tris, d->tris + 9 * (size_t)d->n_tris);
    S.n_table.assign(d->n_table, d->n_table + (size_t)d->n_media * d->n_lambda);
    S.lambdas.assign(d->lambdas, d->lambdas + d->n_lambda);
    if (d->n_coefs > 0) S.coefs.assign(d->coefs, d->coefs + d->n_coefs);
    S.n_lambda = d->n_lambda;
    S.n_detectors = d->n_detectors;
    S.eps_srf = d->eps_srf;
    S.eps_ray = d->eps_ray;
    S.eps_ins = d->eps_ins;
    S.mt_keps = d->mt_keps;
    S.mt_leps = d->mt_leps;
    S.grad_h = d->grad_h;
    S.march_iters = d->march_iters;
    return true;
}

// ------------------------------------------------------------------ SDFs
// _world_to_sdf  SDFs/AbstractSDF.jl:35-40 :  T * (point - position(sdf))
template <class T>
V3<T> world_to_sdf(const bmo_shape& s, const V3<T>& pt) {
    V3<T> d{pt.x - s.pos[0], pt.y - s.pos[1], pt.z - s.pos[2]};
    return matvec<T>(s.tdir, d);
}

// min(maximum(d), zero(T)) + norm(max.(d, zero(T)))   (cylinder-type 2D box distance)
template <class T>
T cyl2d(const V2<T>& d) {
    T mx = jmax(d.x, d.y);
    V2<T> m{jmax(d.x, 0.0), jmax(d.y, 0.0)};
    return jmin(mx, 0.0) + norm(m);
}


// ---- aspheres / acylinders (AsphericalLensSDF.jl, AcylindricalSDF.jl), T = double or Dual ---------------------
const double kNaN = std::numeric_limits<double>::quiet_NaN();
// aspheric_equation AsphericalLensSDF.jl:133-141
template <class T>
T aspheric_equation(const T& r, double c, double k, const double* a, int na) {
    T r2 = lit2(r);
    T sqrt_arg = 1 - (1 + k) * (c * c) * r2;
    if (sqrt_arg < 0.0) return mknan(r);
    T sum_a = r2 * 0.0;
    for (int i = 1; i <= na; ++i) {
        T term = a[i - 1] * ipow(r2, i);
        sum_a = i == 1 ? term : sum_a + term;
    }
    return c * r2 / (1 + jsqrt(sqrt_arg)) + sum_a;
}
// gradient_aspheric_equation :147-156 ; first component (the second is 1); NaN if undefined
template <class T>
T gradient_aspheric_equation(const T& r, double c, double k, const double* a, int na) {
    double Ri = 1 / c;
    T sqrt_arg = 1 - lit2(r) * (1 + k) / (Ri * Ri);
    if (sqrt_arg < 0.0) return mknan(r);
    T sq = jsqrt(sqrt_arg);
    T gr = 2 * r / (Ri * (sq + 1)) + lit3(r) * (1 + k) / ((Ri * Ri * Ri) * sq * lit2(sq + 1));
    T sum_r = r * 0.0;
    for (int m = 1; m <= na; ++m) {
        T term = (2 * m * a[m - 1]) * ipow(r, 2 * (m - 1) + 1);
        sum_r = m == 1 ? term : sum_r + term;
    }
    return -sum_r - gr;
}
template <class T>
T norm_g(const T& g1) { return jsqrt(lit2(g1) + 1.0 * 1.0); }
double jsign(double x) { return x > 0 ? 1.0 : (x < 0 ? -1.0 : x); }
// sd_line_segment :164-169
template <class T>
T sd_line_segment(const T& px, const T& py, double ax, double ay, double bx, double by) {
    T pax = px - ax, pay = py - ay;
    double bax = bx - ax, bay = by - ay;
    T h = clamp01((pax * bax + pay * bay) / (bax * bax + bay * bay));
    T ex = pax - h * bax, ey = pay - h * bay;
    return jsqrt(ex * ex + ey * ey);
}
// convex_aspheric_surface_distance :186-241
template <class T>
T convex_aspheric_surface_distance(const T& r, const T& z, double c, double k, double d, const double* a, int na, double max_sag) {
    T r2 = lit2(r);
    double r2_bound = (d / 2) * (d / 2);
    T zv = aspheric_equation<T>(r, c, k, a, na);
    T g = gradient_aspheric_equation<T>(r, c, k, a, na);
    double zb = aspheric_equation<double>(d / 2, c, k, a, na);
    double gb = gradient_aspheric_equation<double>(d / 2, c, k, a, na);
    if (jisnan(zv) || jisnan(g) || r2 > r2_bound) {
        T e = r - jsignv(r) * d / 2;
        T db = e;
        if (z < zb) db = jsqrt(lit2(e) + lit2(z - zb));
        else if (zb < z && z < 0.0) db = jsqrt(lit2(e));
        else if (z > 0.0 && (jsign(c) == 1 && zb < 0)) db = jsqrt(lit2(e) + lit2(z));
        else db = jsqrt(lit2(e) + lit2(z - zb));
        return db / norm_g<double>(gb);
    }
    T da = jabs(z - zv) / norm_g<T>(g);
    if (jsign(c) == 1 && zb < 0) {
        double n = norm_g<double>(gb);
        T s1 = sd_line_segment<T>(r, z, d / 2, zb, d / 2, max_sag) / n;
        T s2 = sd_line_segment<T>(r, z, d / 2, max_sag, -d / 2, max_sag) / n;
        T s3 = sd_line_segment<T>(r, z, -d / 2, max_sag, -d / 2, zb) / n;
        T m = jmin(jmin(jmin(da, s1), s2), s3);
        if (zv < z && z < max_sag) return -m;
        return m;
    }
    T sdl = sd_line_segment<T>(r, z, d / 2, zb, -d / 2, zb) / norm_g<double>(gb);
    if (jsign(c) * zv < jsign(c) * z && jsign(c) * z < jsign(c) * zb) return -jmin(sdl, da);
    return jmin(sdl, da);
}
// concave_aspheric_surface_distance :243-307
template <class T>
T concave_aspheric_surface_distance(const T& r, const T& z, double c, double k, double d, const double* a, int na, double max_sag) {
    T r2 = lit2(r);
    double r2_bound = (d / 2) * (d / 2);
    T zv = aspheric_equation<T>(r, c, k, a, na);
    T g = gradient_aspheric_equation<T>(r, c, k, a, na);
    double zb = aspheric_equation<double>(d / 2, c, k, a, na);
    double gb = gradient_aspheric_equation<double>(d / 2, c, k, a, na);
    if (jisnan(zv) || jisnan(g)) {
        T e = r - jsignv(r) * d / 2;
        T db = e;
        if (z < 0.0) db = jsqrt(lit2(e) + lit2(z));
        else if (0.0 < z && z < zb) db = jsqrt(lit2(e));
        else db = jsqrt(lit2(e) + lit2(z - zb));
        return db / norm_g<double>(gb);
    }
    T da = jabs(z - zv) / norm_g<T>(g);
    if (max_sag > 0 && zb < 0) {
        T sdl = sd_line_segment<T>(r, z, d / 2, zb, -d / 2, zb) / norm_g<double>(gb);
        if (r2 > r2_bound) return sdl;
        if (zb < z && z < zv) return -jmin(da, sdl);
        if (zb > 0 && (0.0 < z && z < zv)) return -jmin(da, sdl);
        return jmin(da, sdl);
    }
    double n = norm_g<double>(gb);
    T s1 = sd_line_segment<T>(r, z, d / 2, zb, d / 2, 0.0) / n;
    T s2 = sd_line_segment<T>(r, z, d / 2, 0.0, -d / 2, 0.0) / n;
    T s3 = sd_line_segment<T>(r, z, -d / 2, 0.0, -d / 2, zb) / n;
    if (r2 > r2_bound) return jmin(jmin(s1, s2), s3);
    T m = jmin(jmin(jmin(da, s1), s2), s3);
    if (zb < 0 && (zv < z && z < 0.0)) return -m;
    if (zb > 0 && (0.0 < z && z < zv)) return -m;
    return m;
}
inline double asph_sdf(const Scene& S, const bmo_shape& s, const V3<double>& point) {  // :309-349 (op_revolve_z, AbstractSDF.jl:191-194)
    V3<double> p = world_to_sdf(s, point);
    double r = norm(V2<double>{p.x, p.z}) - 0.0;
    const double* a = S.coefs.data() + s.child_begin;
    if (s.kind == BMO_SHAPE_ASPH_CONVEX) return convex_aspheric_surface_distance<double>(r, p.y, 1 / s.p[0], s.p[1], s.p[2], a, s.child_count, s.p[3]);
    return concave_aspheric_surface_distance<double>(r, p.y, 1 / s.p[0], s.p[1], s.p[2], a, s.child_count, s.p[3]);
}
inline Dual asph_sdf(const Scene&, const bmo_shape&, const V3<Dual>&) { return Dual{kNaN, {kNaN, kNaN, kNaN}}; }  // never differentiated (:5)

// op_extrude_x AbstractSDF.jl:229-234
template <class T>
T op_extrude_x(const T& d2, const T& px, double height) {
    T w1 = d2, w2 = jabs(px) - height;
    V2<T> m{jmax(w1, 0.0), jmax(w2, 0.0)};
    return jmin(jmax(w1, w2), 0.0) + norm(m);
}

template <class T>
T sdf_shape(const Scene& S, int sid, const V3<T>& point);

// sdf evaluated on an already-local point (MeniscusLensSDF.jl:42-46 children)
template <class T>
T sdf_kind(const Scene& S, const bmo_shape& s, const V3<T>& point) {
    switch (s.kind) {
        case BMO_SHAPE_SPHERE: {  // SphericalLensSDF.jl:86-89
            V3<T> p = world_to_sdf(s, point);
            return norm(p) - s.p[0];
        }
        case BMO_SHAPE_POINT: {  // test/runtests.jl:943-946
            V3<T> p = world_to_sdf(s, point);
            return norm(p);
        }
        case BMO_SHAPE_PLANO: {  // SphericalLensSDF.jl:60-65
            V3<T> p = world_to_sdf(s, point);
            double th = s.p[0], dia = s.p[1];
            V2<T> a{norm(V2<T>{p.x, p.z}), p.y - th / 2};
            V2<T> d{jabs(a.x) - dia / 2, jabs(a.y) - th / 2};
            return cyl2d(d);
        }
        case BMO_SHAPE_CYLINDER: {  // PrimitiveSDF.jl:71-76
            V3<T> p = world_to_sdf(s, point);
            V2<T> a{norm(V2<T>{p.x, p.z}), p.y};
            V2<T> d{jabs(a.x) - s.p[0], jabs(a.y) - s.p[1]};
            return cyl2d(d);
        }
        case BMO_SHAPE_CONVEX: {  // SphericalLensSDF.jl:219-232
            V3<T> p = world_to_sdf(s, point);
            double radius = s.p[0], dia = s.p[1], height = s.p[3];
            V2<T> q{norm(V2<T>{p.x, p.z}), -p.y + radius};
            double q1 = value(q.x), q2 = value(q.y);
            double sv = jmax((height - radius) * (q1 * q1) + ((dia / 2) * (dia / 2)) * (height + radius - 2 * q2),
                             height * q1 - dia / 2 * q2);
            if (sv < 0) return norm(q) - radius;
            if (q1 < dia / 2) return height - q.y;
            return norm(V2<T>{q.x - dia / 2, q.y - height});
        }
        case BMO_SHAPE_CUTSPHERE: {  // PrimitiveSDF.jl:112-124
            V3<T> p = world_to_sdf(s, point);
            double radius = s.p[0], height = s.p[1], w = s.p[2];
            V2<T> q{norm(V2<T>{p.x, p.z}), p.y};
            double q1 = value(q.x), q2 = value(q.y);
            double sv = jmax((height - radius) * (q1 * q1) + (w * w) * (height + radius - 2 * q2), height * q1 - w * q2);
            if (sv < 0) return norm(q) - radius;
            if (q1 < w) return height - q.y;
            return norm(V2<T>{q.x - w, q.y - height});
        }
        case BMO_SHAPE_CONCAVE: {  // SphericalLensSDF.jl:159-170
            V3<T> p = world_to_sdf(s, point);
            double radius = s.p[0], dia = s.p[1], sag = s.p[2];
            V3<T> ps{p.x + 0.0, p.y + sag / 2, p.z + 0.0};
            V2<T> a{norm(V2<T>{ps.x, ps.z}), ps.y};
            V2<T> d{jabs(a.x) - dia / 2, jabs(a.y) - sag / 2};
            T sdf1 = cyl2d(d);
            V3<T> pr{p.x + 0.0, p.y + radius, p.z + 0.0};
            T sdf2 = norm(pr) - radius;
            return jmax(sdf1, -sdf2);
        }
        case BMO_SHAPE_BOX: {  // PrimitiveSDF.jl:41-46
            V3<T> p = world_to_sdf(s, point);
            V3<T> q{jabs(p.x) - s.p[0], jabs(p.y) - s.p[1], jabs(p.z) - s.p[2]};
            V3<T> m{jmax(q.x, 0.0), jmax(q.y, 0.0), jmax(q.z, 0.0)};
            return norm(m) + jmin(jmax(q.x, jmax(q.y, q.z)), 0.0);
        }
        case BMO_SHAPE_PRISM: {  // PrimitiveSDF.jl:204-210
            V3<T> p = world_to_sdf(s, point);
            V3<T> q{jabs(p.x) - s.p[0], jabs(p.y) - s.p[1], jabs(p.z) - s.p[2]};
            V3<T> m{jmax(q.x, 0.0), jmax(q.y, 0.0), jmax(q.z, 0.0)};
            T box_dist = norm(m) + jmin(jmax(q.x, jmax(q.y, q.z)), 0.0);
            T pln_dist = (p.x + p.y) / std::sqrt(2.0);
            return jmax(box_dist, pln_dist);
        }
        case BMO_SHAPE_RING: {  // PrimitiveSDF.jl:151-166 (sdf_box)
            V3<T> p = world_to_sdf(s, point);
            V2<T> pp{norm(V2<T>{p.x, p.z}) - s.p[0], p.y};
            V2<T> d{jabs(pp.x) - s.p[1], jabs(pp.y) - s.p[2]};
            V2<T> m{jmax(d.x, 0.0), jmax(d.y, 0.0)};
            return norm(m) + jmin(jmax(d.x, d.y), 0.0);
        }
        case BMO_SHAPE_ASPH_CONVEX:
        case BMO_SHAPE_ASPH_CONCAVE: return asph_sdf(S, s, point);
        case BMO_SHAPE_ACYL_CONVEX:
        case BMO_SHAPE_ACYL_CONCAVE: {  // AcylindricalSDF.jl:55-74, :122-141: op_extrude_x of the 2D aspheric profile
            V3<T> p = world_to_sdf(s, point);
            const double* a = S.coefs.data() + s.child_begin;
            double c = 1 / s.p[0], dia = s.p[1], height = s.p[2], k = s.p[3], max_sag = s.p[4];
            T d2 = s.kind == BMO_SHAPE_ACYL_CONVEX ? convex_aspheric_surface_distance<T>(p.z, p.y, c, k, dia, a, s.child_count, max_sag)
                                                   : concave_aspheric_surface_distance<T>(p.z, p.y, c, k, dia, a, s.child_count, max_sag);
            return op_extrude_x<T>(d2, p.x, height / 2);
        }
        case BMO_SHAPE_CYL_CONVEX: {  // CylindricalSDF.jl:62-85 (sdf_cut_disk :76-85)
            V3<T> p = world_to_sdf(s, point);
            double r = s.p[0], dia = s.p[1], height = s.p[2];
            double h = std::sqrt(r * r - (dia / 2) * (dia / 2));
            double w = std::sqrt(r * r - h * h);
            V2<T> q{jabs(p.y), p.z};
            double q1 = value(q.x), q2 = value(q.y);
            double sv = jmax((h - r) * (q1 * q1) + (w * w) * (h + r - 2 * q2), h * q1 - w * q2);
            T d2 = (sv < 0) ? norm(q) - r : ((q1 < w) ? h - q.y : norm(V2<T>{q.x - w, q.y - h}));
            return op_extrude_x<T>(d2, p.x, height / 2);
        }
        case BMO_SHAPE_CYL_CONCAVE: {  // CylindricalSDF.jl:123-139
            V3<T> p = world_to_sdf(s, point);
            double radius = s.p[0], dia = s.p[1], height = s.p[2];
            double ar = std::fabs(radius);
            double sg = ar - std::sqrt(ar * ar - 0.25 * (dia * dia));
            V3<T> ps{p.x + 0.0, p.y + (-radius), p.z + 0.0};
            V2<T> a{norm(V2<T>{p.z, ps.y}), ps.x};
            V2<T> d{jabs(a.x) - ar, jabs(a.y) - height / 2};
            T c = cyl2d(d);
            V3<T> pp{p.x + 0.0, p.y + (-sg / 2 * jsign(radius)), p.z + 0.0};
            V3<T> q{jabs(pp.x) - height / 2, jabs(pp.y) - sg / 2, jabs(pp.z) - dia / 2};
            V3<T> m{jmax(q.x, 0.0), jmax(q.y, 0.0), jmax(q.z, 0.0)};
            T l = norm(m) + jmin(jmax(q.x, jmax(q.y, q.z)), 0.0);
            return jmax(l, -c);
        }
        case BMO_SHAPE_UNION: {  // UnionSDF.jl:53-56  minimum(sdf(_sdf, pos) for _sdf in s.sdfs)
            T best = sdf_shape<T>(S, S.children[s.child_begin], point);
            for (int c = 1; c < s.child_count; ++c) best = jmin(best, sdf_shape<T>(S, S.children[s.child_begin + c], point));
            return best;
        }
        case BMO_SHAPE_MENISCUS: {  // MeniscusLensSDF.jl:42-46
            V3<T> p = world_to_sdf(s, point);
            T a = sdf_shape<T>(S, S.children[s.child_begin + 0], p);
            T b = sdf_shape<T>(S, S.children[s.child_begin + 1], p);
            T c = sdf_shape<T>(S, S.children[s.child_begin + 2], p);
            return jmax(jmin(a, b), -c);
        }
        default: return T{} + kInf;
    }
}

template <class T>
T sdf_shape(const Scene& S, int sid, const V3<T>& point) {
    return sdf_kind<T>(S, S.shapes[sid], point);
}

double sdf_d(const Scene& S, int sid, const D3& p) { return sdf_shape<double>(S, sid, p); }

// numeric_gradient  AbstractSDF.jl:81-88
D3 numeric_gradient(const Scene& S, int sid, const D3& pos) {
    double e = S.grad_h;
    D3 g{sdf_d(S, sid, D3{pos.x + e, pos.y + 0.0, pos.z + 0.0}) - sdf_d(S, sid, D3{pos.x - e, pos.y - 0.0, pos.z - 0.0}),
         sdf_d(S, sid, D3{pos.x + 0.0, pos.y + e, pos.z + 0.0}) - sdf_d(S, sid, D3{pos.x - 0.0, pos.y - e, pos.z - 0.0}),
         sdf_d(S, sid, D3{pos.x + 0.0, pos.y + 0.0, pos.z + e}) - sdf_d(S, sid, D3{pos.x - 0.0, pos.y - 0.0, pos.z - e})};
    return normalize_pt(g);
}

// normal_fd  AbstractSDF.jl:90-95  (ForwardDiff.gradient, NaN => numeric fallback)
D3 normal_fd(const Scene& S, int sid, const D3& p) {
    V3<Dual> x{mkdual(p.x, 1, 0, 0), mkdual(p.y, 0, 1, 0), mkdual(p.z, 0, 0, 1)};
    Dual y = sdf_shape<Dual>(S, sid, x);
    D3 n = normalize_vec(D3{y.p[0], y.p[1], y.p[2]});
    if (!std::isnan(n.x) && !std::isnan(n.y) && !std::isnan(n.z)) return n;
    return numeric_gradient(S, sid, p);
}

// normal3d(s::AbstractSDF, pos) AbstractSDF.jl:79 ; UnionSDF.jl:86-91 (argmin child, first on ties)
D3 normal3d_sdf(const Scene& S, int sid, const D3& pos) {
    const bmo_shape& s = S.shapes[sid];
    if (s.kind == BMO_SHAPE_UNION) {
        int best = 0;
        double bv = sdf_d(S, S.children[s.child_begin], pos);
        for (int c = 1; c < s.child_count; ++c) {
            double v = sdf_d(S, S.children[s.child_begin + c], pos);
            // Base._rf_findmin: replace only if isgreater(fm, fx): strict, with -0.0 < 0.0
            bool greater = (v < bv) || (v == bv && std::signbit(v) && !std::signbit(bv));
            if (greater) {
                bv = v;
                best = c;
            }
        }
        return normal3d_sdf(S, S.children[s.child_begin + best], pos);
    }
    if (s.kind == BMO_SHAPE_ASPH_CONVEX || s.kind == BMO_SHAPE_ASPH_CONCAVE) return numeric_gradient(S, sid, pos);  // AsphericalLensSDF.jl:5
    return normal_fd(S, sid, pos);
}

// ------------------------------------------------------------------ intersection
struct Isect {
    bool hit = false;
    double t = kInf;
    D3 n{0, 0, 0};
    int obj = -1, shape = -1;
};

// _raymarch_outside  AbstractSDF.jl:102-125
Isect raymarch_outside(const Scene& S, int sid, D3 pos, const D3& dir) {
    double dist = sdf_d(S, sid, pos);
    double t0 = dist;
    int i = 1;
    while (i <= S.march_iters) {
        pos = D3{pos.x + dist * dir.x, pos.y + dist * dir.y, pos.z + dist * dir.z};
        dist = sdf_d(S, sid, pos);
        t0 += dist;
        i += 1;
        if (dist < S.eps_ray) {
            Isect r;
            r.hit = true;
            r.t = t0;
            r.n = normal3d_sdf(S, sid, pos);
            r.shape = sid;
            return r;
        }
    }
    return Isect{};
}

// _raymarch_inside  AbstractSDF.jl:132-159
Isect raymarch_inside(const Scene& S, int sid, D3 pos, const D3& dir) {
    double dl = S.eps_ins;
    double t0 = 0;
    int i = 1;
    while (i <= S.march_iters) {
        pos = D3{pos.x + dl * dir.x, pos.y + dl * dir.y, pos.z + dl * dir.z};
        t0 += dl;
        double dist = sdf_d(S, sid, pos);
        if (dist > 0) {
            Isect r = raymarch_outside(S, sid, pos, -dir);
            if (!r.hit) break;
            r.t = t0 - r.t;
            return r;
        }
        i += 1;
    }
    return Isect{};
}

// intersect3d(object::AbstractSDF, ray)  AbstractSDF.jl:166-181
Isect intersect_sdf(const Scene& S, int sid, const D3& pos, const D3& dir) {
    double d = sdf_d(S, sid, pos);
    if (d > S.eps_srf) return raymarch_outside(S, sid, pos, dir);
    D3 n = normal3d_sdf(S, sid, pos);
    if (dot(dir, n) <= 0) return raymarch_inside(S, sid, pos, dir);
    return Isect{};
}

// MoellerTrumboreAlgorithm  Mesh.jl:203-237
double moeller_trumbore(const double* f, const D3& pos, const D3& dir, double keps, double leps) {
    D3 V1{f[0], f[1], f[2]}, V2{f[3], f[4], f[5]}, V3_{f[6], f[7], f[8]};
    D3 E1 = V2 - V1, E2 = V3_ - V1;
    D3 Pv = cross(dir, E2);
    double Det = dot(E1, Pv);
    if (std::fabs(Det) < keps) return kInf;
    D3 Tv = pos - V1;
    double invDet = 1 / Det;
    double u = dot(Tv, Pv) * invDet;
    if ((u < 0 - keps) || (u > 1 + keps)) return kInf;
    D3 Qv = cross(Tv, E1);
    double v = dot(dir, Qv) * invDet;
    if ((v < 0 - keps) || (u + v > 1 + keps)) return kInf;
    double t = dot(E2, Qv) * invDet;
    if (t < leps) return kInf;
    return t;
}

// intersect3d(mesh, ray)  Mesh.jl:244-267 ; normal3d(mesh, fID) Mesh.jl:183-192
Isect intersect_mesh(const Scene& S, int sid, const D3& pos, const D3& dir) {
    const bmo_shape& s = S.shapes[sid];
    int fID = -1;
    double t0 = kInf;
    for (int i = 0; i < s.tri_count; ++i) {
        double t = moeller_trumbore(&S.tris[9 * (size_t)(s.tri_begin + i)], pos, dir, S.mt_keps, S.mt_leps);
        if (t < t0) {
            t0 = t;
            fID = i;
        }
    }
    if (std::isinf(t0)) return Isect{};
    const double* f = &S.tris[9 * (size_t)(s.tri_begin + fID)];
    D3 V1{f[0], f[1], f[2]}, V2{f[3], f[4], f[5]}, V3_{f[6], f[7], f[8]};
    D3 n = normalize_pt(cross(V2 - V1, V3_ - V1));
    Isect r;
    r.hit = true;
    r.t = t0;
    r.n = normalize_pt(n);
    r.shape = sid;
    return r;
}

Isect intersect_shape(const Scene& S, int sid, const D3& pos, const D3& dir) {
    if (S.shapes[sid].kind == BMO_SHAPE_MESH) return intersect_mesh(S, sid, pos, dir);
    return intersect_sdf(S, sid, pos, dir);
}

// intersect3d(object, ray)  AbstractRay.jl:118-155 ; PlateBeamsplitter.jl:160-187 ; NonInteractable.jl:19
Isect intersect_object(const Scene& S, int oid, const D3& pos, const D3& dir) {
    const bmo_object& o = S.objects[oid];
    Isect res;
    switch (o.kind) {
        case BMO_OBJ_NONINTERACTABLE: return Isect{};
        case BMO_OBJ_DOUBLET:
        case BMO_OBJ_CUBE_BS: {  // MultiShape default: nearest part, strict <
            int nparts = o.kind == BMO_OBJ_DOUBLET ? 2 : 3;
            for (int k = 0; k < nparts; ++k) {
                Isect tmp = intersect_shape(S, o.shape[k], pos, dir);
                if (!tmp.hit) continue;
                if (!res.hit) {
                    res = tmp;
                    continue;
                }
                if (tmp.t < res.t) res = tmp;
            }
            break;
        }
        case BMO_OBJ_PLATE_BS: {
            Isect ic = intersect_shape(S, o.shape[1], pos, dir);  // coating
            Isect is = intersect_shape(S, o.shape[0], pos, dir);  // substrate
            if (!ic.hit && !is.hit) return Isect{};
            if (!is.hit) res = ic;
            else if (!ic.hit) res = is;
            else if (isapprox(ic.t, is.t)) res = ic;
            else if (ic.t < is.t) res = ic;
            else res = is;
            break;
        }
        default: res = intersect_shape(S, o.shape[0], pos, dir);
    }
    if (res.hit) res.obj = oid;
    return res;
}

// ------------------------------------------------------------------ rays & beams
struct Ray {
    D3 pos, dir;
    bool has_isect = false;
    Isect isect;
    double lambda = 0;
    int li = 0;
    double n = 1;
    Cx E0[3] = {{0, 0}, {0, 0}, {0, 0}};
};

struct Hint {
    bool set = false;
    int obj = -1, shape = -1;
};

struct Beam {
    std::vector<Ray> rays;
    Beam* parent = nullptr;
};

struct Node {  // one Beam (RAY/POLARIZED) or one GaussianBeamlet
    Beam chief, waist, div;  // RAY/POLARIZED use chief only
    Node* parent = nullptr;
    std::vector<std::unique_ptr<Node>> children;
    double lambda = 0, w0 = 0;
    Cx E0{0, 0};
    int status = 0;
};

struct DetHit {
    double d[9];
};

struct Ctx {
    const Scene* S;
    int kind;
    int r_max;
    std::vector<std::vector<DetHit>> det;  // per detector, this root's hits in push! order
    int64_t calls = 0;
};

// trace_all  System.jl:57-72
Isect trace_all(Ctx& C, const Ray& ray) {
    const Scene& S = *C.S;
    Isect result;
    for (int o = 0; o < (int)S.objects.size(); ++o) {
        C.calls += 1;
        Isect tmp = intersect_object(S, o, ray.pos, ray.dir);
        if (!tmp.hit) continue;
        if (!result.hit || tmp.t < result.t) result = tmp;
    }
    return result;
}

// tracing_step! / trace_one  System.jl:74-110
void tracing_step(Ctx& C, Ray& ray, const Hint& hint) {
    Isect X;
    if (hint.set) {
        C.calls += 1;
        X = intersect_shape(*C.S, hint.shape, ray.pos, ray.dir);
        if (!X.hit) X = trace_all(C, ray);
        else X.obj = hint.obj;
    } else {
        X = trace_all(C, ray);
    }
    ray.has_isect = X.hit;
    ray.isect = X;
}

// reflection3d OpticUtils.jl:7-9
D3 reflection3d(const D3& dir, const D3& normal) {
    double k = 2 * dot(dir, normal);
    return D3{dir.x - k * normal.x, dir.y - k * normal.y, dir.z - k * normal.z};
}

// refraction3d OpticUtils.jl:31-45 ; returns false on the unit-length ArgumentError
bool refraction3d(const D3& dir, const D3& normal, double n1, double n2, D3& out, bool& tir) {
    if (!isapprox(norm(dir), 1.0)) return false;
    if (!isapprox(norm(normal), 1.0)) return false;
    double n = n1 / n2;
    double cosi = -dot(normal, dir);
    double sint2 = (n * n) * (1 - cosi * cosi);
    if (sint2 > 1.0) {
        out = reflection3d(dir, normal);
        tir = true;
        return true;
    }
    double cost = std::sqrt(1 - sint2);
    double k = n * cosi - cost;
    out = D3{n * dir.x + k * normal.x, n * dir.y + k * normal.y, n * dir.z + k * normal.z};
    tir = false;
    return true;
}

double n_medium(const Scene& S, int medium, int li) { return S.n_table[(size_t)medium * S.n_lambda + li]; }

D3 hit_point(const Ray& r) {  // position(ray) + length(ray) * direction(ray)
    return D3{r.pos.x + r.isect.t * r.dir.x, r.pos.y + r.isect.t * r.dir.y, r.pos.z + r.isect.t * r.dir.z};
}

// ---- polarization helpers -------------------------------------------------
// isparallel3d LinearAlgebraUtils.jl:6-8
bool isparallel3d(const D3& a, const D3& b) { return isapprox(std::fabs(dot(normalize_pt(a), normalize_pt(b))), 1.0, 2.220446049250313e-16); }

// deterministic stand-in for normal3d(input) (LinearAlgebraUtils.jl:35-47 uses rand; the
// P-matrix is basis independent to rounding, SURVEY.md §7 "Random vectors")
D3 perpendicular_to(const D3& input) {
    D3 nw = std::fabs(input.x) > 0.9 * norm(input) ? D3{0, 1, 0} : D3{1, 0, 0};
    double nn = norm(input);
    double k = dot(nw, input);
    double n2 = nn * nn;
    nw = D3{nw.x - k * input.x / n2, nw.y - k * input.y / n2, nw.z - k * input.z / n2};
    return normalize_pt(nw);
}

// _calculate_global_E0  PolarizedRays.jl:165-207 ; J = diag(j11, j22, 1) (SPBasis :117-125)
void global_E0(const D3& in_dir, const D3& out_dir, const D3& normal, Cx j11, Cx j22, const Cx* E0, Cx* out) {
    bool par_io = isparallel3d(in_dir, out_dir);
    D3 v = par_io ? normal : out_dir;
    if (isparallel3d(in_dir, normal)) v = perpendicular_to(in_dir);
    D3 s = normalize_pt(cross(in_dir, v));
    D3 p1 = cross(in_dir, s);
    double Oin[9] = {s.x, s.y, s.z, p1.x, p1.y, p1.z, in_dir.x, in_dir.y, in_dir.z};
    double Oout[9];
    // in_dir ≈ -out_dir : norm(x - y) <= rtol*max(norm(x), norm(y))
    D3 mo = -out_dir;
    bool anti = norm(in_dir - mo) <= 1.4901161193847656e-08 * std::fmax(norm(in_dir), norm(mo));
    if (par_io && !anti) {
        double tmp[9] = {s.x, p1.x, in_dir.x, s.y, p1.y, in_dir.y, s.z, p1.z, in_dir.z};
        std::memcpy(Oout, tmp, sizeof tmp);
    } else {
        D3 p2 = cross(out_dir, s);
        double tmp[9] = {s.x, p2.x, out_dir.x, s.y, p2.y, out_dir.y, s.z, p2.z, out_dir.z};
        std::memcpy(Oout, tmp, sizeof tmp);
    }
    Cx J[9] = {j11, {0, 0}, {0, 0}, {0, 0}, j22, {0, 0}, {0, 0}, {0, 0}, {1, 0}};
    // P = (O_out * J) * O_in, left-fold sums
    Cx A[9], P[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) A[3 * i + j] = (Oout[3 * i + 0] * J[0 + j] + Oout[3 * i + 1] * J[3 + j]) + Oout[3 * i + 2] * J[6 + j];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) P[3 * i + j] = (A[3 * i + 0] * Oin[0 + j] + A[3 * i + 1] * Oin[3 + j]) + A[3 * i + 2] * Oin[6 + j];
    for (int i = 0; i < 3; ++i) out[i] = (P[3 * i + 0] * E0[0] + P[3 * i + 1] * E0[1]) + P[3 * i + 2] * E0[2];
}

// isorthogonal3d(dir, E0; atol=1e-14)  PolarizedRays.jl:54-56  (dot conjugates its first arg; dir is real)
bool e0_orthogonal(const D3& dir, const Cx* E0) {
    Cx d = (dir.x * E0[0] + dir.y * E0[1]) + dir.z * E0[2];
    return std::sqrt(abs2(d)) <= 1e-14;
}

// fresnel_coefficients OpticUtils.jl:121-131
void fresnel(double theta, double n, Cx& rs, Cx& rp, Cx& ts, Cx& tp) {
    double cost = trig::cos(theta);
    double st = trig::sin(theta);
    Cx n2s2 = csqrt_real(n * n - st * st);
    Cx c{cost, 0};
    rs = (c - n2s2) / (c + n2s2);
    Cx n2c{(n * n) * cost, 0};
    rp = (Cx{-(n * n) * cost, 0} + n2s2) / (n2c + n2s2);
    ts = rs + Cx{1, 0};
    tp = Cx{2 * n * cost, 0} / (n2c + n2s2);
}

// angle3d LinearAlgebraUtils.jl:103-108
double angle3d(const D3& a, const D3& b) {
    double arg = dot(a, b) / (norm(a) * norm(b));
    arg = arg < -1.0 ? -1.0 : (arg > 1.0 ? 1.0 : arg);
    return trig::acos(arg);
}

// ---- single-ray interactions ----------------------------------------------
struct Inter {
    bool some = false;  // false == `nothing`
    Hint hint;
    Ray ray;
};

// interact3d(::AbstractRefractiveOptic, Ray) Lenses.jl:46-77 ; PolarizedRay :86-126
Inter interact_refractive(Ctx& C, int oid, int medium, const Ray& ray, Node& node, int hint_shape) {
    const Scene& S = *C.S;
    Inter out;
    D3 normal = ray.isect.n;
    bool entering = dot(ray.dir, ray.isect.n) < 0;  // isentering AbstractRay.jl:234-237
    double n1, n2;
    Hint hint;
    if (entering) {
        n1 = ray.n;
        n2 = n_medium(S, medium, ray.li);
        hint = Hint{true, oid, hint_shape};
    } else {
        n1 = n_medium(S, medium, ray.li);
        n2 = 1.0;  // refractive_index(system, λ) AbstractSystem.jl:21
        normal = -normal;
    }
    Ray nr;
    nr.lambda = ray.lambda;
    nr.li = ray.li;
    if (C.kind == BMO_BEAM_POLARIZED) {
        D3 raypos = hit_point(ray);
        double th = angle3d(ray.dir, -normal);
        Cx rs, rp, ts, tp;
        fresnel(th, n2 / n1, rs, rp, ts, tp);
        D3 ndir;
        Cx j11, j22;
        // is_internally_reflected OpticUtils.jl:144-146
        if (isapprox(abs2(rs), 1.0, 1e-6) && isapprox(abs2(rp), 1.0, 1e-6)) {
            hint = Hint{true, oid, hint_shape};
            n2 = n_medium(S, medium, ray.li);
            ndir = reflection3d(ray.dir, normal);
            j11 = -rs;
            j22 = rp;
        } else {
            bool tir;
            if (!refraction3d(ray.dir, normal, n1, n2, ndir, tir)) {
                node.status |= BMO_NODE_ERR_UNIT;
                return out;
            }
            j11 = ts;
            j22 = tp;
        }
        global_E0(ray.dir, ndir, ray.isect.n, j11, j22, ray.E0, nr.E0);
        nr.pos = raypos;
        nr.dir = ndir;
        nr.n = n2;
        if (!e0_orthogonal(nr.dir, nr.E0)) {
            node.status |= kErrOrtho;
            return out;
        }
    } else {
        D3 ndir;
        bool tir;
        if (!refraction3d(ray.dir, normal, n1, n2, ndir, tir)) {
            node.status |= BMO_NODE_ERR_UNIT;
            return out;
        }
        nr.pos = hit_point(ray);
        if (tir) {
            hint = Hint{true, oid, hint_shape};
            n2 = n_medium(S, medium, ray.li);
        }
        nr.dir = ndir;
        nr.n = n2;
    }
    out.some = true;
    out.hint = hint;
    out.ray = nr;
    return out;
}

// interact3d(::AbstractReflectiveOptic, ...) Mirrors.jl:39-69
Inter interact_mirror(Ctx& C, const Ray& ray, Node& node) {
    Inter out;
    Ray nr;
    nr.lambda = ray.lambda;
    nr.li = ray.li;
    nr.pos = hit_point(ray);
    nr.dir = reflection3d(ray.dir, ray.isect.n);
    nr.n = ray.n;
    if (C.kind == BMO_BEAM_POLARIZED) {
        global_E0(ray.dir, nr.dir, ray.isect.n, Cx{-1, 0}, Cx{1, 0}, ray.E0, nr.E0);
        if (!e0_orthogonal(nr.dir, nr.E0)) {
            node.status |= kErrOrtho;
            return out;
        }
    }
    out.some = true;
    out.ray = nr;
    return out;
}

// child first rays of a thin splitter: ThinBeamsplitter.jl:73-106 (public ctor: dir normalised, n = 1)
bool bs_child_rays(Ctx& C, const bmo_object& o, const Ray& ray, Ray& tr, Ray& rf) {
    D3 pos = hit_point(ray);
    D3 rdir = reflection3d(ray.dir, ray.isect.n);
    tr = Ray{};
    rf = Ray{};
    tr.lambda = rf.lambda = ray.lambda;
    tr.li = rf.li = ray.li;
    tr.pos = rf.pos = pos;
    tr.n = rf.n = 1.0;
    bool ok = true;
    if (C.kind == BMO_BEAM_POLARIZED) {
        global_E0(ray.dir, ray.dir, ray.isect.n, Cx{o.transmittance, 0}, Cx{o.transmittance, 0}, ray.E0, tr.E0);
        global_E0(ray.dir, rdir, ray.isect.n, Cx{-o.reflectance, 0}, Cx{o.reflectance, 0}, ray.E0, rf.E0);
    }
    tr.dir = normalize_pt(ray.dir);
    rf.dir = normalize_pt(rdir);
    if (C.kind == BMO_BEAM_POLARIZED) ok = e0_orthogonal(tr.dir, tr.E0) && e0_orthogonal(rf.dir, rf.E0);
    return ok;
}

// optical_path_length(beam) Beam.jl:137-149
double opl_beam(const Beam& b) {
    double l0 = b.parent ? opl_beam(*b.parent) : 0.0;
    for (const Ray& r : b.rays) {
        if (!r.has_isect) break;
        l0 += r.isect.t * r.n;
    }
    return l0;
}
// length(beam) Beam.jl:125-169
double len_beam(const Beam& b) {
    double l0 = b.parent ? len_beam(*b.parent) : 0.0;
    double l = 0;
    for (const Ray& r : b.rays) {
        if (!r.has_isect) break;
        l += r.isect.t;
    }
    return l + l0;
}

void push_det(Ctx& C, int slot, const double* d9) {
    DetHit h;
    std::memcpy(h.d, d9, sizeof h.d);
    C.det[slot].push_back(h);
}

// refraction3d(ray, n2) AbstractRay.jl:244-253
bool refraction3d_ray(const Ray& ray, double n2, D3& out) {
    D3 nml = ray.isect.n;
    if (!(dot(ray.dir, ray.isect.n) < 0)) nml = D3{nml.x * -1, nml.y * -1, nml.z * -1};
    bool tir;
    return refraction3d(ray.dir, nml, ray.n, n2, out, tir);
}

// interact3d dispatch for Beam{T,Ray} / Beam{T,PolarizedRay}.
// Children (splitters) are appended to node.children with their first rays.
// replace!(beam, interaction, index) Beam.jl:81-94 and the per-ray part of _modify_beam_head! Beam.jl:97-113:
// position, direction, wavelength, refractive index (and polarization); the stored intersection is kept.
void replace_ray_head(Ray& old, const Ray& nw) {
    old.pos = nw.pos;
    old.dir = nw.dir;
    old.lambda = nw.lambda;
    old.li = nw.li;
    old.n = nw.n;
    for (int k = 0; k < 3; ++k) old.E0[k] = nw.E0[k];
}
// children!(beam, [t, r]) AbstractBeam.jl:62-76: no children yet -> link and append; same number of children (a retrace)
// -> _modify_beam_head! of each existing child (Beam.jl:97-113; Gaussian.jl:154-161: the three heads, lambda and E0 —
// NOT w0, which keeps the value of the previous solve).
void set_children(int kind, Node& node, std::vector<std::unique_ptr<Node>>& fresh) {
    if (node.children.empty()) {
        for (auto& ch : fresh) node.children.push_back(std::move(ch));
    } else if (node.children.size() == fresh.size()) {
        for (size_t i = 0; i < fresh.size(); ++i) {
            Node& old = *node.children[i];
            replace_ray_head(old.chief.rays.front(), fresh[i]->chief.rays.front());
            if (kind == BMO_BEAM_GAUSSIAN) {
                replace_ray_head(old.waist.rays.front(), fresh[i]->waist.rays.front());
                replace_ray_head(old.div.rays.front(), fresh[i]->div.rays.front());
                old.lambda = fresh[i]->lambda;
                old.E0 = fresh[i]->E0;
            }
        }
    }
    node.status |= BMO_NODE_SPLIT;
}

Inter interact_beam(Ctx& C, Node& node, Beam& beam, const Ray& ray) {
    const Scene& S = *C.S;
    int oid = ray.isect.obj;
    const bmo_object& o = S.objects[oid];
    auto spawn = [&](const Ray& tr, const Ray& rf) {
        std::vector<std::unique_ptr<Node>> fresh;
        for (const Ray* r : {&tr, &rf}) {
            auto ch = std::make_unique<Node>();
            ch->chief.rays.push_back(*r);
            ch->chief.parent = &beam;
            ch->parent = &node;
            ch->lambda = node.lambda;
            fresh.push_back(std::move(ch));
        }
        set_children(C.kind, node, fresh);
    };
    switch (o.kind) {
        case BMO_OBJ_MIRROR: return interact_mirror(C, ray, node);
        case BMO_OBJ_REFRACTIVE: return interact_refractive(C, oid, o.medium[0], ray, node, o.shape[0]);
        case BMO_OBJ_DOUBLET: {  // DoubletLenses.jl:66-76 (Ray only)
            if (C.kind == BMO_BEAM_POLARIZED) return Inter{};  // Gaussian sub-beams are Beam{T,Ray{T}}: the Ray method applies
            if (ray.isect.shape == o.shape[0]) {
                Inter i = interact_refractive(C, oid, o.medium[0], ray, node, o.shape[0]);
                if (i.some) i.hint = Hint{true, oid, o.shape[1]};
                return i;
            } else {
                Inter i = interact_refractive(C, oid, o.medium[1], ray, node, o.shape[1]);
                if (i.some) i.hint = Hint{true, oid, o.shape[0]};
                return i;
            }
        }
        case BMO_OBJ_THIN_BS: {  // ThinBeamsplitter.jl:108-115
            Ray tr, rf;
            if (!bs_child_rays(C, o, ray, tr, rf)) {
                node.status |= kErrOrtho;
                return Inter{};
            }
            spawn(tr, rf);
            return Inter{};
        }
        case BMO_OBJ_PLATE_BS: {  // PlateBeamsplitter.jl:189-228
            if (ray.isect.shape == o.shape[0]) {
                Inter i = interact_refractive(C, oid, o.medium[0], ray, node, o.shape[0]);
                if (i.some) i.hint = Hint{true, oid, o.shape[1]};
                return i;
            }
            if (ray.isect.shape == o.shape[1]) {
                Ray tr, rf;
                if (!bs_child_rays(C, o, ray, tr, rf)) {
                    node.status |= kErrOrtho;
                    return Inter{};
                }
                double n_optics = n_medium(S, o.medium[0], ray.li), n_system = 1.0;
                double nt, nr_;
                D3 nd;
                bool ok;
                if (dot(ray.dir, ray.isect.n) < 0) {
                    nt = n_optics;
                    nr_ = n_system;
                    ok = refraction3d_ray(ray, n_optics, nd);
                } else {
                    nt = n_system;
                    nr_ = n_optics;
                    ok = refraction3d_ray(ray, n_system, nd);
                }
                if (!ok) {
                    node.status |= BMO_NODE_ERR_UNIT;
                    return Inter{};
                }
                tr.n = nt;
                rf.n = nr_;
                tr.dir = normalize_pt(nd);  // direction! AbstractRay.jl:83-86
                spawn(tr, rf);
                return Inter{};
            }
            return Inter{};
        }
        case BMO_OBJ_CUBE_BS: {  // CubeBeamsplitter.jl:63-92
            if (ray.isect.shape == o.shape[0]) {
                Inter i = interact_refractive(C, oid, o.medium[0], ray, node, o.shape[0]);
                if (i.some) i.hint = Hint{true, oid, o.shape[2]};
                return i;
            }
            if (ray.isect.shape == o.shape[2]) {
                Ray tr, rf;
                if (!bs_child_rays(C, o, ray, tr, rf)) {
                    node.status |= kErrOrtho;
                    return Inter{};
                }
                double ng = n_medium(S, o.medium[0], ray.li);
                tr.n = ng;
                rf.n = ng;
                spawn(tr, rf);
                return Inter{};
            }
            if (ray.isect.shape == o.shape[1]) {
                Inter i = interact_refractive(C, oid, o.medium[1], ray, node, o.shape[1]);
                if (i.some) i.hint = Hint{true, oid, o.shape[2]};
                return i;
            }
            return Inter{};
        }
        case BMO_OBJ_SPOTDETECTOR: {  // Spotdetector.jl:50-61
            const bmo_shape& m = S.shapes[o.shape[0]];
            D3 hp = hit_point(ray);
            D3 loc{hp.x - m.pos[0], hp.y - m.pos[1], hp.z - m.pos[2]};
            double x = dot(loc, D3{m.dir[0], m.dir[3], m.dir[6]});
            double z = dot(loc, D3{m.dir[2], m.dir[5], m.dir[8]});
            double d9[9] = {x, z, 0, 0, 0, 0, 0, 0, 0};
            push_det(C, o.detector, d9);
            node.status |= BMO_NODE_DETECTED;
            return Inter{};
        }
        case BMO_OBJ_PSFDETECTOR: {  // PSFDetector.jl:77-89 (Beam{T,Ray{T}} only)
            if (C.kind == BMO_BEAM_POLARIZED) return Inter{};
            D3 hp = hit_point(ray);
            double l = opl_beam(beam);
            double proj = std::fabs(dot(ray.dir, ray.isect.n));
            double d9[9] = {hp.x, hp.y, hp.z, ray.dir.x, ray.dir.y, ray.dir.z, l, proj, 2 * M_PI / ray.lambda};
            push_det(C, o.detector, d9);
            node.status |= BMO_NODE_DETECTED;
            return Inter{};
        }
        case BMO_OBJ_POLARIZER: {  // PolarizationFilter.jl:31-48 ; JonesCalculus.jl:29-45
            if (C.kind != BMO_BEAM_POLARIZED) return Inter{};
            const bmo_shape& m = S.shapes[o.shape[0]];
            // P = R * J * R'
            Cx J[9], A[9], P[9];
            for (int i = 0; i < 9; ++i) J[i] = Cx{o.jones[2 * i], o.jones[2 * i + 1]};
            const double* R = m.dir;
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) A[3 * i + j] = (R[3 * i + 0] * J[0 + j] + R[3 * i + 1] * J[3 + j]) + R[3 * i + 2] * J[6 + j];
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) P[3 * i + j] = (A[3 * i + 0] * R[3 * j + 0] + A[3 * i + 1] * R[3 * j + 1]) + A[3 * i + 2] * R[3 * j + 2];
            // Q = I - in_dir * in_dir'
            double d[3] = {ray.dir.x, ray.dir.y, ray.dir.z};
            double Q[9];
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) Q[3 * i + j] = (i == j ? 1.0 : 0.0) - d[i] * d[j];
            Cx B[9], P2[9];
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) B[3 * i + j] = (Q[3 * i + 0] * P[0 + j] + Q[3 * i + 1] * P[3 + j]) + Q[3 * i + 2] * P[6 + j];
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) P2[3 * i + j] = (B[3 * i + 0] * Q[0 + j] + B[3 * i + 1] * Q[3 + j]) + B[3 * i + 2] * Q[6 + j];
            Inter out;
            Ray nr;
            nr.lambda = ray.lambda;
            nr.li = ray.li;
            for (int i = 0; i < 3; ++i) nr.E0[i] = (P2[3 * i + 0] * ray.E0[0] + P2[3 * i + 1] * ray.E0[1]) + P2[3 * i + 2] * ray.E0[2];
            double nrm = std::sqrt((abs2(nr.E0[0]) + abs2(nr.E0[1])) + abs2(nr.E0[2]));
            if (isapprox(nrm, o.cutoff)) {
                node.status |= BMO_NODE_BLOCKED;
                return Inter{};
            }
            nr.pos = hit_point(ray);
            nr.dir = ray.dir;
            nr.n = ray.n;
            if (!e0_orthogonal(nr.dir, nr.E0)) {
                node.status |= kErrOrtho;
                return Inter{};
            }
            out.some = true;
            out.ray = nr;
            return out;
        }
        default: return Inter{};  // Intersectable.jl:15, NonInteractable.jl:20
    }
}

// trace_system!(system, beam::Beam)  System.jl:130-154
void trace_beam(Ctx& C, Node& node) {
    Beam& beam = node.chief;
    bool have = false;
    Inter interaction;
    while ((int)beam.rays.size() < C.r_max) {
        Ray& ray = beam.rays.back();
        tracing_step(C, ray, have ? interaction.hint : Hint{});
        if (!ray.has_isect) {
            node.status |= BMO_NODE_MISS;
            return;
        }
        Ray rcopy = ray;  // interact may reallocate children, not rays; copy for safety
        interaction = interact_beam(C, node, beam, rcopy);
        if (!interaction.some) {
            node.status |= BMO_NODE_STOPPED;
            return;
        }
        have = true;
        beam.rays.push_back(interaction.ray);
    }
    node.status |= BMO_NODE_RMAX;
}

// ---- Gaussian beamlets -------------------------------------------------------
// point_on_beam Beam.jl:177-205
void point_on_beam(const Beam& beam, double t, D3& point, int& index) {
    double temp = beam.parent ? len_beam(*beam.parent) : 0.0;
    int numEl = (int)beam.rays.size();
    for (int idx = 1; idx <= numEl; ++idx) {
        const Ray& ray = beam.rays[idx - 1];
        if (idx == numEl) break;
        double len = ray.has_isect ? ray.isect.t : kInf;
        temp += len;
        if (t < temp) {
            double b = temp - t;
            double k = len - b;
            point = D3{ray.pos.x + k * ray.dir.x, ray.pos.y + k * ray.dir.y, ray.pos.z + k * ray.dir.z};
            index = idx;
            return;
        }
    }
    const Ray& ray = beam.rays.back();
    double b = t - temp;
    point = D3{ray.pos.x + b * ray.dir.x, ray.pos.y + b * ray.dir.y, ray.pos.z + b * ray.dir.z};
    index = numEl;
}

// line_plane_distance3d LinearAlgebraUtils.jl:127-136 ; returns NaN for `nothing`
double line_plane_distance3d(const D3& pp, const D3& pn, const D3& lp, const D3& ld) {
    double denom = dot(pn, ld);
    if (std::fabs(denom) > 1e-6) {
        double c = dot(pp - lp, pn);
        return c / denom;
    }
    return std::numeric_limits<double>::quiet_NaN();
}

// gauss_parameters Gaussian.jl:298-353 ; returns w, R, psi, w0
void gauss_parameters(const Node& g, double z, double out[4]) {
    D3 p0;
    int index;
    point_on_beam(g.chief, z, p0, index);
    const Ray& chief = g.chief.rays[index - 1];
    const Ray& div = g.div.rays[index - 1];
    const Ray& waist = g.waist.rays[index - 1];
    double il = line_plane_distance3d(p0, chief.dir, div.pos, div.dir);
    D3 y0{div.pos.x + il * div.dir.x - p0.x, div.pos.y + il * div.dir.y - p0.y, div.pos.z + il * div.dir.z - p0.z};
    double y_d = norm(y0);
    y0 = D3{y0.x / y_d, y0.y / y_d, y0.z / y_d};
    double m_d = trig::tan(M_PI / 2 - angle3d(y0, div.dir));
    il = line_plane_distance3d(p0, chief.dir, waist.pos, waist.dir);
    y0 = D3{waist.pos.x + il * waist.dir.x - p0.x, waist.pos.y + il * waist.dir.y - p0.y, waist.pos.z + il * waist.dir.z - p0.z};
    double y_w = norm(y0);
    y0 = D3{y0.x / y_w, y0.y / y_w, y0.z / y_w};
    double m_w = trig::tan(M_PI / 2 - angle3d(y0, waist.dir));
    double n = chief.n;
    double H = std::fabs(n * (y_w * m_d - y_d * m_w));
    double lam = g.lambda;
    if (!isapprox(H, lam / M_PI, 1e-6)) H = lam / M_PI;
    double E_kt = y_d * m_d + y_w * m_w;
    double F_kt = std::sqrt(m_d * m_d + m_w * m_w);
    double w = std::sqrt(y_d * y_d + y_w * y_w);
    double R = E_kt / (w * w);
    double zz = E_kt / (F_kt * F_kt);
    double psi = -trig::atan2(1.0, std::sqrt(1 / (R * zz) - 1));
    double w0 = H / (n * F_kt);
    if (std::isnan(R)) R = 0;
    if (std::isnan(psi)) psi = 0;
    if (std::isnan(w0)) w0 = w;
    if (R < 0) psi = -psi;
    out[0] = w;
    out[1] = R;
    out[2] = psi;
    out[3] = w0;
}

// _beams_hits_same_shape Gaussian.jl:171-180
bool same_shape(const Ray& c, const Ray& w, const Ray& d) {
    bool nc = !c.has_isect, nw = !w.has_isect, nd = !d.has_isect;
    if (nc || nw || nd) return nc && nw && nd;
    return c.isect.shape == w.isect.shape && w.isect.shape == d.isect.shape;
}

struct GInter {
    bool some = false;
    Inter c, w, d;
};

// interact3d(system, object, gauss, ray_id): generic Gaussian.jl:124-135, thin splitter
// ThinBeamsplitter.jl:117-168, plate PlateBeamsplitter.jl:230-275, cube CubeBeamsplitter.jl:94-121
GInter interact_gauss(Ctx& C, Node& g, int id /*1-based*/) {
    const Scene& S = *C.S;
    Ray rc = g.chief.rays[id - 1], rw = g.waist.rays[id - 1], rd = g.div.rays[id - 1];
    int oid = rc.isect.obj;
    const bmo_object& o = S.objects[oid];
    auto generic = [&](void) {
        GInter gi;
        // each sub-beam interacts with the object found by the CHIEF ray (System.jl:306-309)
        Ray w2 = rw, d2 = rd;
        w2.isect.obj = oid;
        d2.isect.obj = oid;
        gi.c = interact_beam(C, g, g.chief, rc);
        gi.w = interact_beam(C, g, g.waist, w2);
        gi.d = interact_beam(C, g, g.div, d2);
        gi.some = gi.c.some && gi.w.some && gi.d.some;
        return gi;
    };
    auto split = [&](double nt, double nr_, bool set_n, bool refract) -> bool {
        // _beamsplitter_{transmitted,reflected}_beam for the three sub beams (ThinBeamsplitter.jl:117-137)
        Ray t[3], r[3];
        const Ray* src[3] = {&rc, &rw, &rd};
        for (int k = 0; k < 3; ++k) bs_child_rays(C, o, *src[k], t[k], r[k]);
        double gp[4];
        gauss_parameters(g, len_beam(g.chief), gp);
        double w0 = gp[3];
        Cx Et = (o.transmittance * g.E0) * (g.w0 / w0);
        Cx Er = (o.reflectance * g.E0) * (g.w0 / w0);
        double df = dot(rc.dir, rc.isect.n);
        if (df < 0) Er = Er * Cx{-1.0, 1.2246467991473532e-16};  // exp(im*π)
        else Er = Er * Cx{1.0, 0.0};
        if (refract) {
            for (int k = 0; k < 3; ++k) {
                D3 nd;
                if (!refraction3d_ray(*src[k], nt, nd)) {
                    g.status |= BMO_NODE_ERR_UNIT;
                    return false;
                }
                t[k].dir = normalize_pt(nd);
            }
        }
        if (set_n)
            for (int k = 0; k < 3; ++k) {
                t[k].n = nt;
                r[k].n = nr_;
            }
        std::vector<std::unique_ptr<Node>> fresh;
        for (int which = 0; which < 2; ++which) {
            auto ch = std::make_unique<Node>();
            Ray* rr = which == 0 ? t : r;
            ch->chief.rays.push_back(rr[0]);
            ch->waist.rays.push_back(rr[1]);
            ch->div.rays.push_back(rr[2]);
            ch->chief.parent = &g.chief;  // parent! Gaussian.jl:113-117
            ch->parent = &g;
            ch->lambda = g.lambda;
            ch->w0 = w0;
            ch->E0 = which == 0 ? Et : Er;
            fresh.push_back(std::move(ch));
        }
        set_children(C.kind, g, fresh);
        return true;
    };
    switch (o.kind) {
        case BMO_OBJ_PHOTODETECTOR: {  // Photodetector.jl:69-107: the field is accumulated by photodetector_field() below; the hit is recorded
            double d9[9] = {std::fabs(dot(rc.dir, rc.isect.n)), 0, 0, 0, 0, 0, 0, 0, 0}, z9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
            push_det(C, o.detector, d9);
            push_det(C, o.detector, z9);
            push_det(C, o.detector, z9);
            g.status |= BMO_NODE_DETECTED;
            return GInter{};
        }
        case BMO_OBJ_THIN_BS: split(1.0, 1.0, false, false); return GInter{};
        case BMO_OBJ_PLATE_BS: {
            if (rc.isect.shape == o.shape[0]) {
                GInter gi = generic();
                if (gi.some) gi.c.hint = Hint{true, oid, o.shape[1]};
                return gi;
            }
            if (rc.isect.shape == o.shape[1]) {
                double n_optics = n_medium(S, o.medium[0], rc.li);
                if (dot(rc.dir, rc.isect.n) < 0) split(n_optics, 1.0, true, true);
                else split(1.0, n_optics, true, true);
            }
            return GInter{};
        }
        case BMO_OBJ_CUBE_BS: {
            if (rc.isect.shape == o.shape[2]) {
                double ng = n_medium(S, o.medium[0], rc.li);
                split(ng, ng, true, false);
                return GInter{};
            }
            GInter gi = generic();
            if (gi.some) gi.c.hint = Hint{true, oid, o.shape[2]};
            return gi;
        }
        default: return generic();
    }
}

// trace_system!(system, gauss)  System.jl:274-318
void trace_gauss(Ctx& C, Node& g) {
    bool have = false;
    Hint hint;
    int seg_counter = (int)g.chief.rays.size();
    while (seg_counter < C.r_max) {
        Hint h = have ? hint : Hint{};
        tracing_step(C, g.chief.rays.back(), h);
        if (!g.chief.rays.back().has_isect) {
            g.status |= BMO_NODE_MISS;
            return;
        }
        tracing_step(C, g.waist.rays.back(), h);
        if (!g.waist.rays.back().has_isect) {
            g.status |= BMO_NODE_MISS;
            return;
        }
        tracing_step(C, g.div.rays.back(), h);
        if (!g.div.rays.back().has_isect) {
            g.status |= BMO_NODE_MISS;
            return;
        }
        if (!same_shape(g.chief.rays[seg_counter - 1], g.waist.rays[seg_counter - 1], g.div.rays[seg_counter - 1])) {
            g.chief.rays.back().has_isect = false;
            g.waist.rays.back().has_isect = false;
            g.div.rays.back().has_isect = false;
            g.status |= BMO_NODE_GAUSS_DIVERGED;
            return;
        }
        GInter gi = interact_gauss(C, g, seg_counter);
        if (!gi.some) {
            g.status |= BMO_NODE_STOPPED;
            return;
        }
        g.chief.rays.push_back(gi.c.ray);
        g.waist.rays.push_back(gi.w.ray);
        g.div.rays.push_back(gi.d.ray);
        hint = gi.c.hint;
        have = true;
        seg_counter += 1;
    }
    g.status |= BMO_NODE_RMAX;
}

// retrace_system!(system, beam::Beam)  System.jl:188-255
void retrace_beam(Ctx& C, Node& node) {
    const Scene& S = *C.S;
    Beam& beam = node.chief;
    bool cleanup_children = false, cleanup_tail = false, reset_intersection = false;
    int cutoff = 0;
    Hint hint;
    const int n = (int)beam.rays.size();
    for (int i = 1; i <= n; ++i) {
        Ray& ray = beam.rays[i - 1];
        if (!ray.has_isect) {
            cleanup_children = cleanup_tail = reset_intersection = true;
            cutoff = i;
            break;
        }
        C.calls += 1;
        Isect X;
        if (!hint.set) {
            X = intersect_object(S, ray.isect.obj, ray.pos, ray.dir);  // intersect3d(object(_intersection), ray)
        } else {
            X = intersect_shape(S, hint.shape, ray.pos, ray.dir);  // intersect3d(shape(_hint), ray)
            if (X.hit) X.obj = hint.obj;
        }
        ray.has_isect = X.hit;
        ray.isect = X;
        if (!X.hit) {
            cleanup_children = cleanup_tail = reset_intersection = true;
            cutoff = i;
            break;
        }
        Ray rcopy = ray;
        Inter inter = interact_beam(C, node, beam, rcopy);
        hint = inter.some ? inter.hint : Hint{};  // hint(::Nothing) = nothing, AbstractSystem.jl:82
        if (!inter.some) {
            node.status |= BMO_NODE_STOPPED;
            if (!node.children.empty() && !(node.status & BMO_NODE_SPLIT)) node.status |= BMO_NODE_RETRACE_STALE;
            if (n > i) {  // nothing is a valid interaction: only the disconnected tail goes
                cleanup_tail = true;
                cutoff = i;
            }
            break;
        }
        if (i < n) {
            replace_ray_head(beam.rays[i], inter.ray);
        } else {
            cleanup_children = true;
            beam.rays.push_back(inter.ray);
            break;
        }
    }
    if (cleanup_children) node.children.clear();
    if (cleanup_tail) beam.rays.resize((size_t)cutoff);
    if (reset_intersection) beam.rays.back().has_isect = false;
}

// retrace_system!(system, gauss)  System.jl:326-428
bool retrace_gauss(Ctx& C, Node& g) {
    const Scene& S = *C.S;
    bool cleanup_children = false, cleanup_tail = false, reset_intersection = false;
    int cutoff = 0;
    Hint hint;
    const int n_c = (int)g.chief.rays.size(), n_w = (int)g.waist.rays.size(), n_d = (int)g.div.rays.size();
    if (!(n_c == n_w && n_w == n_d)) return false;  // error("Gaussian beamlet is broken")
    for (int i = 1; i <= n_c; ++i) {
        Ray& c = g.chief.rays[i - 1];
        Ray& w = g.waist.rays[i - 1];
        Ray& d = g.div.rays[i - 1];
        if (!c.has_isect) {
            cleanup_children = cleanup_tail = reset_intersection = true;
            cutoff = i;
            break;
        }
        int obj;
        C.calls += 3;
        if (!hint.set) {
            obj = c.isect.obj;
            for (Ray* r : {&c, &w, &d}) {
                Isect X = intersect_object(S, obj, r->pos, r->dir);
                r->has_isect = X.hit;
                r->isect = X;
            }
        } else {
            obj = hint.obj;
            for (Ray* r : {&c, &w, &d}) {
                Isect X = intersect_shape(S, hint.shape, r->pos, r->dir);
                r->has_isect = X.hit;
                r->isect = X;
            }
        }
        if (!same_shape(c, w, d) || !c.has_isect) {
            cleanup_children = cleanup_tail = reset_intersection = true;
            cutoff = i;
            break;
        }
        c.isect.obj = w.isect.obj = d.isect.obj = obj;
        GInter gi = interact_gauss(C, g, i);
        hint = gi.some ? gi.c.hint : Hint{};
        if (!gi.some) {
            g.status |= BMO_NODE_STOPPED;
            if (!g.children.empty() && !(g.status & BMO_NODE_SPLIT)) g.status |= BMO_NODE_RETRACE_STALE;
            // a split before the end of the stored path: the children's w0 / E0 come from gauss_parameters(gauss, length(gauss)) with the
            // stale tail still attached (ThinBeamsplitter.jl:125; the tail goes only after this loop).  Reproduced here; flagged because
            // the engine, which has no stale tail, evaluates them at the split point.
            if ((g.status & BMO_NODE_SPLIT) && n_c > i) g.status |= BMO_NODE_RETRACE_STALE;
            if (n_c > i) {
                cleanup_tail = true;
                cutoff = i;
            }
            break;
        }
        if (i < n_c) {
            replace_ray_head(g.chief.rays[i], gi.c.ray);
            replace_ray_head(g.waist.rays[i], gi.w.ray);
            replace_ray_head(g.div.rays[i], gi.d.ray);
        } else {
            cleanup_children = true;
            g.chief.rays.push_back(gi.c.ray);
            g.waist.rays.push_back(gi.w.ray);
            g.div.rays.push_back(gi.d.ray);
            break;
        }
    }
    if (cleanup_children) g.children.clear();
    if (cleanup_tail) {
        g.chief.rays.resize((size_t)cutoff);
        g.waist.rays.resize((size_t)cutoff);
        g.div.rays.resize((size_t)cutoff);
    }
    if (reset_intersection) {
        g.chief.rays.back().has_isect = false;
        g.waist.rays.back().has_isect = false;
        g.div.rays.back().has_isect = false;
    }
    return true;
}

// ------------------------------------------------------------------ result
struct Result {
    bmo_trace_result_view view{};
    std::vector<int32_t> node_root, node_parent, node_first_child, node_first_rec, node_nseg, node_status;
    std::vector<double> node_aux;
    std::vector<int32_t> rec_obj, rec_shape;
    std::vector<double> rec;
    std::vector<int64_t> det_count, det_offset;
    std::vector<int32_t> det_node;
    std::vector<double> det_data;
    std::vector<std::unique_ptr<Node>> roots;  // the solved beam trees (input of a later bmo_cpu_retrace)
    int kind = 0;
    std::vector<bmo_object> objects;           // object table the solution was solved with (photodetector read-out)
};

struct RootOut {
    std::vector<Node*> order;  // BFS order (System.jl:446-458)
    std::unique_ptr<Node> root;
    std::vector<std::vector<DetHit>> det;
    std::vector<std::vector<int>> det_node_local;  // index into order
    int64_t calls = 0;
};

int rec_planes_for(int kind) { return kind == BMO_BEAM_RAY ? 11 : (kind == BMO_BEAM_POLARIZED ? 17 : 33); }

// Root beam i of the batch; with `into` the heads of an already solved root are overwritten instead (the caller of a
// second solve_system! may have moved the source: first ray, and for a Gaussian also lambda, w0, E0).
void load_root(const bmo_ray_batch* in, int64_t i, Node& root, bool into) {
    int64_t n = in->n;
    const double* P = in->planes;
    auto pl = [&](int k) { return P[(size_t)k * n + i]; };
    int li = in->lambda_idx[i];
    auto mk = [&](int base, double lam, double nn) {
        Ray r;
        r.pos = D3{pl(base), pl(base + 1), pl(base + 2)};
        r.dir = D3{pl(base + 3), pl(base + 4), pl(base + 5)};
        r.lambda = lam;
        r.li = li;
        r.n = nn;
        return r;
    };
    auto put = [&](Beam& b, const Ray& r) {
        if (into) replace_ray_head(b.rays.front(), r);
        else b.rays.push_back(r);
    };
    if (in->kind == BMO_BEAM_GAUSSIAN) {
        double lam = pl(18), nn = pl(19);
        put(root.chief, mk(0, lam, nn));
        put(root.waist, mk(6, lam, nn));
        put(root.div, mk(12, lam, nn));
        root.lambda = lam;
        root.w0 = pl(20);
        root.E0 = Cx{pl(21), pl(22)};
    } else {
        Ray r = mk(0, pl(6), pl(7));
        if (in->kind == BMO_BEAM_POLARIZED)
            for (int k = 0; k < 3; ++k) r.E0[k] = Cx{pl(8 + 2 * k), pl(9 + 2 * k)};
        put(root.chief, r);
        root.lambda = pl(6);
    }
}

// Deep copy of a solved tree (parent links re-pointed into the copy).
std::unique_ptr<Node> clone_tree(const Node& src, Node* parent) {
    auto n = std::make_unique<Node>();
    n->chief.rays = src.chief.rays;
    n->waist.rays = src.waist.rays;
    n->div.rays = src.div.rays;
    n->parent = parent;
    n->chief.parent = parent ? &parent->chief : nullptr;
    n->lambda = src.lambda;
    n->w0 = src.w0;
    n->E0 = src.E0;
    n->status = 0;
    for (auto& ch : src.children) n->children.push_back(clone_tree(*ch, n.get()));
    return n;
}

// solve_system!(system, beam; r_max, retrace)  System.jl:444-461
void solve_tree(const Scene& S, int kind, int r_max, bool retrace, std::unique_ptr<Node> root, RootOut& out) {
    Ctx C;
    C.S = &S;
    C.kind = kind;
    C.r_max = r_max;
    C.det.resize(S.n_detectors);
    out.det_node_local.resize(S.n_detectors);
    std::deque<Node*> queue{root.get()};
    while (!queue.empty()) {
        Node* cur = queue.front();
        queue.pop_front();
        int idx = (int)out.order.size();
        out.order.push_back(cur);
        std::vector<size_t> before(S.n_detectors);
        for (int d = 0; d < S.n_detectors; ++d) before[d] = C.det[d].size();
        cur->status = 0;
        if (retrace) {  // fresh one-ray beams: a no-op (System.jl:197-206)
            if (kind == BMO_BEAM_GAUSSIAN) retrace_gauss(C, *cur);
            else retrace_beam(C, *cur);
        }
        // solve_leaf! System.jl:470-475
        if (!cur->chief.rays.back().has_isect) {
            if (kind == BMO_BEAM_GAUSSIAN) trace_gauss(C, *cur);
            else trace_beam(C, *cur);
        }
        for (int d = 0; d < S.n_detectors; ++d)
            for (size_t k = before[d]; k < C.det[d].size(); ++k) out.det_node_local[d].push_back(idx);
        for (auto& ch : cur->children) queue.push_back(ch.get());
    }
    out.det = std::move(C.det);
    out.calls = C.calls;
    out.root = std::move(root);
}

void solve_root(const Scene& S, const bmo_ray_batch* in, int r_max, int64_t i, RootOut& out) {
    auto root = std::make_unique<Node>();
    load_root(in, i, *root, false);
    solve_tree(S, in->kind, r_max, true, std::move(root), out);
}

void write_ray(std::vector<double>& rec, size_t nrec, size_t r, int base, const Ray& ray) {
    auto put = [&](int k, double v) { rec[(size_t)(base + k) * nrec + r] = v; };
    put(0, ray.pos.x);
    put(1, ray.pos.y);
    put(2, ray.pos.z);
    put(3, ray.dir.x);
    put(4, ray.dir.y);
    put(5, ray.dir.z);
    put(6, ray.n);
    put(7, ray.has_isect ? ray.isect.t : kInf);
    put(8, ray.has_isect ? ray.isect.n.x : 0.0);
    put(9, ray.has_isect ? ray.isect.n.y : 0.0);
    put(10, ray.has_isect ? ray.isect.n.z : 0.0);
}

}  // namespace

static int finalize(const Scene& S, const bmo_ray_batch* in, std::vector<RootOut>& outs, void** out_handle, bmo_trace_result_view* view);

template <class F>
static void parallel_roots(int64_t n, int n_threads, F&& body) {
    if (n_threads < 1) n_threads = 1;
    if (n_threads > n) n_threads = (int)std::max<int64_t>(1, n);
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; ++t) {
        th.emplace_back([&, t]() {
            int64_t lo = n * t / n_threads, hi = n * (t + 1) / n_threads;
            for (int64_t i = lo; i < hi; ++i) body(i);
        });
    }
    for (auto& x : th) x.join();
}

extern "C" {

const char* bmo_cpu_last_error(void) { return g_err.c_str(); }

// Same descriptor / batch / result-view contract as bmo_trace (include/bmo.h), computed by the
// reference algorithm on the CPU.  n_threads > 1 runs a plain parallel-for over root beams
// (beams are independent, System.jl:463-468); results are merged in bundle order.
int bmo_cpu_trace(const bmo_scene_desc* desc, const bmo_ray_batch* in, const bmo_trace_opts* opts, int n_threads,
                  void** out_handle, bmo_trace_result_view* view) {
    Scene S;
    if (!build_scene(desc, S)) return BMO_ERR_INVALID;
    if (!in || !opts || !out_handle || !view) {
        g_err = "null argument";
        return BMO_ERR_INVALID;
    }
    int64_t n = in->n;
    std::vector<RootOut> outs((size_t)n);
    parallel_roots(n, n_threads, [&](int64_t i) { solve_root(S, in, opts->r_max, i, outs[(size_t)i]); });
    return finalize(S, in, outs, out_handle, view);
}

// Second and later solve_system! on already solved beams (System.jl:444-461 with retrace = true): every beam of the previous
// solution is retraced (System.jl:188-255 / :326-428) against the scene `desc` (same object / shape numbering, moved
// elements), then traced on from wherever the stored path ended or broke.  `in` supplies the root heads (same count as the
// previous batch).  The previous result stays valid.
int bmo_cpu_retrace(const bmo_scene_desc* desc, const bmo_ray_batch* in, const bmo_trace_opts* opts, int n_threads, void* prev_handle,
                    void** out_handle, bmo_trace_result_view* view) {
    Scene S;
    if (!build_scene(desc, S)) return BMO_ERR_INVALID;
    if (!in || !opts || !out_handle || !view || !prev_handle) {
        g_err = "null argument";
        return BMO_ERR_INVALID;
    }
    const Result* prev = static_cast<const Result*>(prev_handle);
    int64_t n = in->n;
    if ((int64_t)prev->roots.size() != n || prev->kind != in->kind) {
        g_err = "retrace: batch does not match the previous solution (root count / beam kind)";
        return BMO_ERR_INVALID;
    }
    std::vector<RootOut> outs((size_t)n);
    parallel_roots(n, n_threads, [&](int64_t i) {
        std::unique_ptr<Node> root = clone_tree(*prev->roots[(size_t)i], nullptr);
        load_root(in, i, *root, true);
        solve_tree(S, in->kind, opts->r_max, true, std::move(root), outs[(size_t)i]);
    });
    return finalize(S, in, outs, out_handle, view);
}

}  // extern "C"

static int finalize(const Scene& S, const bmo_ray_batch* in, std::vector<RootOut>& outs, void** out_handle, bmo_trace_result_view* view) {
    const int64_t n = in->n;
    auto* R = new Result();
    R->kind = in->kind;
    R->objects = S.objects;
    int rp = rec_planes_for(in->kind);
    size_t nnodes = 0, nrec = 0;
    for (auto& o : outs) {
        nnodes += o.order.size();
        for (Node* nd : o.order) nrec += nd->chief.rays.size();
    }
    R->node_root.resize(nnodes);
    R->node_parent.resize(nnodes);
    R->node_first_child.resize(nnodes);
    R->node_first_rec.resize(nnodes);
    R->node_nseg.resize(nnodes);
    R->node_status.resize(nnodes);
    R->node_aux.assign(nnodes * 4, 0.0);
    R->rec_obj.resize(nrec);
    R->rec_shape.resize(nrec);
    R->rec.assign((size_t)rp * nrec, 0.0);
    R->det_count.assign(S.n_detectors, 0);
    R->det_offset.assign(S.n_detectors, 0);
    int64_t calls = 0;
    size_t nb = 0, rb = 0;
    std::vector<size_t> node_base(outs.size());
    for (size_t i = 0; i < outs.size(); ++i) {
        RootOut& o = outs[i];
        node_base[i] = nb;
        calls += o.calls;
        // local index lookup
        for (size_t k = 0; k < o.order.size(); ++k) {
            Node* nd = o.order[k];
            size_t id = nb + k;
            R->node_root[id] = (int32_t)i;
            R->node_parent[id] = -1;
            R->node_first_child[id] = -1;
            if (nd->parent) {
                for (size_t q = 0; q < k; ++q)
                    if (o.order[q] == nd->parent) {
                        R->node_parent[id] = (int32_t)(nb + q);
                        if (R->node_first_child[nb + q] < 0) R->node_first_child[nb + q] = (int32_t)id;
                        break;
                    }
            }
            R->node_first_rec[id] = (int32_t)rb;
            R->node_nseg[id] = (int32_t)nd->chief.rays.size();
            R->node_status[id] = nd->status;
            if (in->kind == BMO_BEAM_GAUSSIAN) {
                R->node_aux[4 * id + 0] = nd->w0;
                R->node_aux[4 * id + 1] = nd->E0.re;
                R->node_aux[4 * id + 2] = nd->E0.im;
                R->node_aux[4 * id + 3] = nd->lambda;
            } else {
                R->node_aux[4 * id + 0] = nd->lambda;
            }
            for (size_t s = 0; s < nd->chief.rays.size(); ++s) {
                const Ray& r = nd->chief.rays[s];
                R->rec_obj[rb] = r.has_isect ? r.isect.obj : -1;
                R->rec_shape[rb] = r.has_isect ? r.isect.shape : -1;
                write_ray(R->rec, nrec, rb, 0, r);
                if (in->kind == BMO_BEAM_POLARIZED)
                    for (int c = 0; c < 3; ++c) {
                        R->rec[(size_t)(11 + 2 * c) * nrec + rb] = r.E0[c].re;
                        R->rec[(size_t)(12 + 2 * c) * nrec + rb] = r.E0[c].im;
                    }
                if (in->kind == BMO_BEAM_GAUSSIAN) {
                    write_ray(R->rec, nrec, rb, 11, nd->waist.rays[s]);
                    write_ray(R->rec, nrec, rb, 22, nd->div.rays[s]);
                }
                rb += 1;
            }
        }
        nb += o.order.size();
    }
    // detector hits: bundle order x BFS order (Spotdetector.jl:27,59)
    int64_t total = 0;
    for (int d = 0; d < S.n_detectors; ++d) {
        R->det_offset[d] = total;
        for (auto& o : outs) total += (int64_t)o.det[d].size();
        R->det_count[d] = total - R->det_offset[d];
    }
    R->det_node.resize((size_t)total);
    R->det_data.resize((size_t)total * 9);
    for (int d = 0; d < S.n_detectors; ++d) {
        size_t w = (size_t)R->det_offset[d];
        for (size_t i = 0; i < outs.size(); ++i) {
            RootOut& o = outs[i];
            for (size_t k = 0; k < o.det[d].size(); ++k) {
                std::memcpy(&R->det_data[9 * w], o.det[d][k].d, 9 * sizeof(double));
                R->det_node[w] = (int32_t)(node_base[i] + o.det_node_local[d][k]);
                ++w;
            }
        }
    }
    bmo_trace_result_view& v = R->view;
    v.n_roots = n;
    v.n_nodes = (int64_t)nnodes;
    v.n_records = (int64_t)nrec;
    v.n_intersect_calls = calls;
    v.n_steps = 0;
    v.beam_kind = in->kind;
    v.rec_planes = rp;
    v.n_detectors = S.n_detectors;
    v.node_root = R->node_root.data();
    v.node_parent = R->node_parent.data();
    v.node_first_child = R->node_first_child.data();
    v.node_first_rec = R->node_first_rec.data();
    v.node_nseg = R->node_nseg.data();
    v.node_status = R->node_status.data();
    v.node_aux = R->node_aux.data();
    v.rec_obj = R->rec_obj.data();
    v.rec_shape = R->rec_shape.data();
    v.rec = R->rec.data();
    v.det_count = R->det_count.data();
    v.det_offset = R->det_offset.data();
    v.det_node = R->det_node.data();
    v.det_data = R->det_data.data();
    for (auto& o : outs) R->roots.push_back(std::move(o.root));
    *view = v;
    *out_handle = R;
    return BMO_OK;
}

extern "C" {

int bmo_cpu_result_free(void* handle) {
    delete static_cast<Result*>(handle);
    return BMO_OK;
}

// ---- unit hooks for the known-answer tests -------------------------------------
double bmo_cpu_sdf(const bmo_scene_desc* desc, int shape, const double* p) {
    Scene S;
    if (!build_scene(desc, S)) return std::numeric_limits<double>::quiet_NaN();
    return sdf_d(S, shape, D3{p[0], p[1], p[2]});
}
// jl_trig.hpp's functions one by one (tests/test_jl_trig.py): which = 0 sin, 1 cos, 2 tan, 3 acos, 4 atan, 5 atan(y, x)
double bmo_cpu_jl_trig(int which, double x, double y) {
    switch (which) {
        case 0: return jlm::sin(x);
        case 1: return jlm::cos(x);
        case 2: return jlm::tan(x);
        case 3: return jlm::acos(x);
        case 4: return jlm::atan(x);
        case 5: return jlm::atan2(y, x);
    }
    return NAN;
}
void bmo_cpu_jl_trig_n(int which, const double* x, const double* y, long long n, double* out) {
    for (long long i = 0; i < n; ++i) out[i] = bmo_cpu_jl_trig(which, x[i], y ? y[i] : 0.0);
}
int bmo_cpu_normal3d(const bmo_scene_desc* desc, int shape, const double* p, double* n_out) {
    Scene S;
    if (!build_scene(desc, S)) return BMO_ERR_INVALID;
    D3 n = normal3d_sdf(S, shape, D3{p[0], p[1], p[2]});
    n_out[0] = n.x;
    n_out[1] = n.y;
    n_out[2] = n.z;
    return BMO_OK;
}
// returns 1 on hit (t, normal filled), 0 on `nothing`
int bmo_cpu_intersect_shape(const bmo_scene_desc* desc, int shape, const double* pos, const double* dir, double* t, double* n_out) {
    Scene S;
    if (!build_scene(desc, S)) return BMO_ERR_INVALID;
    Isect r = intersect_shape(S, shape, D3{pos[0], pos[1], pos[2]}, D3{dir[0], dir[1], dir[2]});
    if (!r.hit) return 0;
    *t = r.t;
    n_out[0] = r.n.x;
    n_out[1] = r.n.y;
    n_out[2] = r.n.z;
    return 1;
}
int bmo_cpu_intersect_object(const bmo_scene_desc* desc, int object, const double* pos, const double* dir, double* t, double* n_out,
                             int* shape_out) {
    Scene S;
    if (!build_scene(desc, S)) return BMO_ERR_INVALID;
    Isect r = intersect_object(S, object, D3{pos[0], pos[1], pos[2]}, D3{dir[0], dir[1], dir[2]});
    if (!r.hit) return 0;
    *t = r.t;
    n_out[0] = r.n.x;
    n_out[1] = r.n.y;
    n_out[2] = r.n.z;
    *shape_out = r.shape;
    return 1;
}
double bmo_cpu_moeller_trumbore(const double* face9, const double* pos, const double* dir) {
    return moeller_trumbore(face9, D3{pos[0], pos[1], pos[2]}, D3{dir[0], dir[1], dir[2]}, 1e-9, 1e-9);
}
void bmo_cpu_reflection3d(const double* dir, const double* normal, double* out) {
    D3 r = reflection3d(D3{dir[0], dir[1], dir[2]}, D3{normal[0], normal[1], normal[2]});
    out[0] = r.x;
    out[1] = r.y;
    out[2] = r.z;
}
// returns -1 on ArgumentError, else TIR flag
int bmo_cpu_refraction3d(const double* dir, const double* normal, double n1, double n2, double* out) {
    D3 r;
    bool tir;
    if (!refraction3d(D3{dir[0], dir[1], dir[2]}, D3{normal[0], normal[1], normal[2]}, n1, n2, r, tir)) return -1;
    out[0] = r.x;
    out[1] = r.y;
    out[2] = r.z;
    return tir ? 1 : 0;
}
// isentering(ray) AbstractRay.jl:234-237 (false without an intersection); refraction3d(ray, n2) :244-253 — the functions interact3d calls
int bmo_cpu_isentering(const double* dir, const double* normal_or_null) {
    if (!normal_or_null) return 0;
    return dot(D3{dir[0], dir[1], dir[2]}, D3{normal_or_null[0], normal_or_null[1], normal_or_null[2]}) < 0 ? 1 : 0;
}
int bmo_cpu_refraction3d_ray(const double* dir, const double* isect_normal, double n_ray, double n2, double* out) {
    Ray r;
    r.dir = D3{dir[0], dir[1], dir[2]};
    r.has_isect = true;
    r.isect.n = D3{isect_normal[0], isect_normal[1], isect_normal[2]};
    r.n = n_ray;
    D3 o;
    if (!refraction3d_ray(r, n2, o)) return -1;
    out[0] = o.x;
    out[1] = o.y;
    out[2] = o.z;
    return 0;
}
void bmo_cpu_fresnel(double theta, double n, double* out8) {
    Cx rs, rp, ts, tp;
    fresnel(theta, n, rs, rp, ts, tp);
    double v[8] = {rs.re, rs.im, rp.re, rp.im, ts.re, ts.im, tp.re, tp.im};
    std::memcpy(out8, v, sizeof v);
}
void bmo_cpu_global_E0(const double* in_dir, const double* out_dir, const double* normal, const double* j4, const double* e0_6,
                       double* out6) {
    Cx E0[3] = {{e0_6[0], e0_6[1]}, {e0_6[2], e0_6[3]}, {e0_6[4], e0_6[5]}}, o[3];
    global_E0(D3{in_dir[0], in_dir[1], in_dir[2]}, D3{out_dir[0], out_dir[1], out_dir[2]}, D3{normal[0], normal[1], normal[2]},
              Cx{j4[0], j4[1]}, Cx{j4[2], j4[3]}, E0, o);
    for (int i = 0; i < 3; ++i) {
        out6[2 * i] = o[i].re;
        out6[2 * i + 1] = o[i].im;
    }
}

// interact3d(::Photodetector, gauss, ray_id) Photodetector.jl:69-107 for every beamlet of a solved batch that ended on the
// photodetector of slot `detector`, in solve order (bundle order x BFS order), added to field (nx*ny complex, (i,j) at [i + nx*j]).
// electric_field(gauss, r, z) Gaussian.jl:381-392; electric_field(r, z, E0, w0, w, k, psi, R) OpticUtils.jl:87-89.
// gauss_parameters(gauss, z) (Gaussian.jl:298-353) of beamlet `node` (index in solve order: bundle order x BFS order) at n values of z;
// out[4*i..] = w, R, psi, w0.  The function the Photodetector read-out and the splitter code of this oracle call.
int bmo_cpu_gauss_parameters(void* handle, long long node, const double* zs, int n, double* out) {
    const Result* R = static_cast<const Result*>(handle);
    if (!R || R->kind != BMO_BEAM_GAUSSIAN || node < 0) return BMO_ERR_INVALID;
    long long at = 0;
    for (const auto& root : R->roots) {
        std::deque<const Node*> queue{root.get()};
        while (!queue.empty()) {
            const Node* g = queue.front();
            queue.pop_front();
            for (const auto& ch : g->children) queue.push_back(ch.get());
            if (at++ != node) continue;
            for (int i = 0; i < n; ++i) gauss_parameters(*g, zs[i], out + 4 * i);
            return BMO_OK;
        }
    }
    return BMO_ERR_INVALID;
}

int bmo_cpu_photodetector_field(void* handle, int detector, const double* position, const double* orientation, const double* xs, const double* ys,
                                int nx, int ny, double* field) {
    const Result* R = static_cast<const Result*>(handle);
    if (!R || R->kind != BMO_BEAM_GAUSSIAN) return BMO_OK;
    // T = transpose(orientation): T[k,1] = orientation[1,k], T[k,3] = orientation[3,k]  (orientation row-major)
    const double ox[3] = {orientation[0], orientation[1], orientation[2]}, oy[3] = {orientation[6], orientation[7], orientation[8]};
    for (const auto& root : R->roots) {
        std::deque<const Node*> queue{root.get()};
        while (!queue.empty()) {
            const Node* g = queue.front();
            queue.pop_front();
            for (const auto& ch : g->children) queue.push_back(ch.get());
            const Ray& ray = g->chief.rays.back();
            // only beamlets the detector's interact3d ran for: a beamlet whose chief ray reaches the detector while its waist or
            // divergence ray misses it ends in trace_system! (System.jl:283-296) before any interaction
            if (!ray.has_isect || !(g->status & BMO_NODE_DETECTED)) continue;
            const bmo_object& o = R->objects[(size_t)ray.isect.obj];
            if (o.kind != BMO_OBJ_PHOTODETECTOR || o.detector != detector) continue;
            const double len_g = len_beam(g->chief);
            const double l0 = len_g - ray.isect.t;
            const D3 p0 = ray.pos, d0 = ray.dir;
            const double proj = std::fabs(dot(d0, ray.isect.n));
            const double k = 2 * M_PI / g->lambda;
            const double dl = opl_beam(g->chief) - len_g;
            const double ref_phi = dl / g->lambda * (2 * M_PI);
            for (int j = 0; j < ny; ++j) {
                const double y = ys[j];
                for (int i = 0; i < nx; ++i) {
                    const double x = xs[i];
                    const D3 p1{ox[0] * x + oy[0] * y + position[0], ox[1] * x + oy[1] * y + position[1], ox[2] * x + oy[2] * y + position[2]};
                    const double l1 = dot(p1 - p0, d0);
                    const D3 p2{p0.x + l1 * d0.x, p0.y + l1 * d0.y, p0.z + l1 * d0.z};
                    const double r = norm(p1 - p2);
                    const double z = l0 + l1;
                    double gp[4];
                    gauss_parameters(*g, z, gp);  // hint = point_on_beam(gauss, z), the default
                    const double w = gp[0], Rc = gp[1], psi = gp[2], w0 = gp[3];
                    Cx E = g->E0 * (g->w0 / w0);
                    E = E * w0;
                    E = Cx{E.re / w, E.im / w};
                    E = E * std::exp(-(r * r) / (w * w));
                    const double ph = k * z + psi + (k * (r * r) * Rc) / 2;
                    E = E * Cx{std::cos(ph), std::sin(ph)};
                    E = E * Cx{std::cos(ref_phi), std::sin(ref_phi)};
                    E = E * std::sqrt(proj);
                    field[2 * ((size_t)i + (size_t)nx * j)] += E.re;
                    field[2 * ((size_t)i + (size_t)nx * j) + 1] += E.im;
                }
            }
        }
    }
    return BMO_OK;
}

// intensity(psf::PSFDetector) PSFDetector.jl:190-237 — the coherent sum over recorded hits, evaluated per grid point in
// the order the hits were pushed (the reference's `for h in psf.data`; its @simd annotation permits re-association, the
// sequential order is the canonical one).  hits = [n_hits][9] (hit, dir, opl, proj, k); out (i,j) at [i + n*j].
void bmo_cpu_psf_intensity(const double* hits, long long n_hits, const double* origin, const double* e1, const double* e2, const double* xs,
                           const double* zs, int n, double* out_intensity, double* out_field) {
    for (int j = 0; j < n; ++j) {           // Threads.@threads for j in eachindex(zs)
        const double z = zs[j];
        for (int i = 0; i < n; ++i) {
            const double x = xs[i];
            const D3 p{(origin[0] + x * e1[0]) + z * e2[0], (origin[1] + x * e1[1]) + z * e2[1], (origin[2] + x * e1[2]) + z * e2[2]};
            double re = 0.0, im = 0.0;
            for (long long h = 0; h < n_hits; ++h) {
                const double* r = hits + 9 * h;
                const double l = ((p.x - r[0]) * r[3] + (p.y - r[1]) * r[4]) + (p.z - r[2]) * r[5];   // dot(p - position(h), direction(h))
                const double ph = r[8] * (r[6] + l);                                                // wavenumber * (opl + l)
                re += r[7] * std::cos(ph);                                                         // projection_factor * cis(...)
                im += r[7] * std::sin(ph);
            }
            out_intensity[i + (size_t)n * j] = re * re + im * im;  // abs2
            if (out_field) {
                out_field[2 * (i + (size_t)n * j)] = re;
                out_field[2 * (i + (size_t)n * j) + 1] = im;
            }
        }
    }
}

}  // extern "C"
