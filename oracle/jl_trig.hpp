// oracle/jl_trig.hpp — TEST INFRASTRUCTURE (part of the CPU oracle; only tests/, smoke() and bench.py's cpu_baseline may use it).
//
// Julia's Base.Math elementary functions the step path calls — sin, cos (OpticUtils.jl:121-131), acos (LinearAlgebraUtils.jl:103-108), tan
// (Gaussian.jl:326-336), atan(y, x) (Gaussian.jl:345) — restated from the algorithms Base ports (base/special/trig.jl and rem_pio2.jl, which
// follow FreeBSD msun / Sun fdlibm 5.3: k_sin.c, k_cos.c, k_tan.c, e_rem_pio2.c, s_atan.c, e_atan2.c, e_acos.c).  Julia's standard library
// is not part of /root/reference (third-party arithmetic, named there); the pin is the reference's own exact assertion
// runtests.jl:157 — fresnel_coefficients(atan(1.5), 1.5) has real(rp) == 0 — which holds with these functions and with no C library's
// (glibc returns the correctly rounded sin(atan(1.5)) = 0.8320502943378437; Base's port returns 0.8320502943378436).
// Every polynomial is Base's `@horner`, i.e. nested `muladd`, evaluated as a fused multiply-add (std::fma) as Julia does on x86-64 / aarch64
// machines with FMA; everything else is one rounded operation per Julia operation (the oracle is compiled with -ffp-contract=off).
// Range: |x| < 2^20 pi/2 (the Cody-Waite branches); beyond that Base switches to Payne-Hanek, not restated — NaN is returned.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <initializer_list>

namespace jlm {

inline uint64_t to_u64(double x) {
    uint64_t u;
    std::memcpy(&u, &x, sizeof u);
    return u;
}
inline double to_f64(uint64_t u) {
    double x;
    std::memcpy(&x, &u, sizeof x);
    return x;
}
inline uint32_t hi_abs(double x) { return static_cast<uint32_t>(to_u64(x) >> 32) & 0x7fffffffu; }  // poshighword
inline double zero_low(double x) { return to_f64(to_u64(x) & 0xffffffff00000000ull); }

// Base.Math.@horner(x, c0, c1, ..., cn) = muladd(x, muladd(x, ... cn ..., c1), c0)
inline double horner(double x, std::initializer_list<double> c) {
    const double* p = c.end();
    double acc = *--p;
    while (p != c.begin()) acc = std::fma(x, acc, *--p);
    return acc;
}

struct Two {  // DoubleFloat64
    double hi, lo;
};

namespace k {
constexpr double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
constexpr double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
constexpr double PIO2_1 = 1.57079632673412561417e+00, PIO2_1T = 6.07710050650619224932e-11, PIO2_2 = 6.07710050630396597660e-11,
                 PIO2_2T = 2.02226624879595063154e-21, PIO2_3 = 2.02226624871116645580e-21, PIO2_3T = 8.47842766036889956997e-32,
                 INVPIO2 = 6.36619772367581382433e-01;
constexpr double PI = 3.141592653589793, PI_LO = 1.2246467991473531772e-16, PIO2_HI = 1.57079632679489655800e+00, PIO2_LO = 6.12323399573676603587e-17,
                 PIO4 = 7.85398163397448278999e-01, PIO4_LO = 3.06161699786838301793e-17;
}  // namespace k

inline double sin_k(double y) {  // sin_kernel(::Float64)
    const double z = y * y, w = z * z;
    const double r = horner(z, {k::S2, k::S3, k::S4}) + z * w * horner(z, {k::S5, k::S6});
    const double v = z * y;
    return y + v * (k::S1 + z * r);
}
inline double sin_k(Two y) {  // sin_kernel(::DoubleFloat64)
    const double z = y.hi * y.hi, w = z * z;
    const double r = horner(z, {k::S2, k::S3, k::S4}) + z * w * horner(z, {k::S5, k::S6});
    const double v = z * y.hi;
    return y.hi - ((z * (0.5 * y.lo - v * r) - y.lo) - v * k::S1);
}
inline double cos_k(Two y) {  // cos_kernel
    const double z = y.hi * y.hi, w = z * z;
    const double r = z * horner(z, {k::C1, k::C2, k::C3}) + w * w * horner(z, {k::C4, k::C5, k::C6});
    const double hz = 0.5 * z, a = 1.0 - hz;
    return a + (((1.0 - a) - hz) + (z * r - y.hi * y.lo));
}

inline Two cw2(double x, double fn) {  // cody_waite_2c_pio2
    const double z = std::fma(-fn, k::PIO2_1, x);
    const double y1 = std::fma(-fn, k::PIO2_1T, z);
    return Two{y1, std::fma(-fn, k::PIO2_1T, z - y1)};
}
inline int cw_ext(double x, uint32_t xh, Two& y) {  // cody_waite_ext_pio2
    const double fn = std::nearbyint(x * k::INVPIO2);
    double r = std::fma(-fn, k::PIO2_1, x), w = fn * k::PIO2_1T;
    const int j = static_cast<int>(xh >> 20);
    double y1 = r - w;
    auto gap = [&](double v) { return j - static_cast<int>((hi_abs(v) >> 20) & 0x7ff); };
    if (gap(y1) > 16) {
        double t = r;
        w = fn * k::PIO2_2;
        r = t - w;
        w = std::fma(fn, k::PIO2_2T, -((t - r) - w));
        y1 = r - w;
        if (gap(y1) > 49) {
            t = r;
            w = fn * k::PIO2_3;
            r = t - w;
            w = std::fma(fn, k::PIO2_3T, -((t - r) - w));
            y1 = r - w;
        }
    }
    y = Two{y1, (r - y1) - w};
    return static_cast<int>(fn);
}
inline bool reduce(double x, int& n, Two& y) {  // rem_pio2_kernel, Cody-Waite part
    const uint32_t xh = hi_abs(x);
    const double sg = x > 0.0 ? 1.0 : -1.0;
    auto two_const = [&](int q) {
        n = x > 0.0 ? q : -q;
        y = cw2(x, sg * q);
        return true;
    };
    if (xh <= 0x400f6a7au) {
        if ((xh & 0xfffffu) == 0x921fbu) return n = cw_ext(x, xh, y), true;
        return two_const(xh <= 0x4002d97cu ? 1 : 2);
    }
    if (xh <= 0x401c463bu) {
        if (xh <= 0x4015fdbcu) {
            if (xh == 0x4012d97cu) return n = cw_ext(x, xh, y), true;
            return two_const(3);
        }
        if (xh == 0x401921fbu) return n = cw_ext(x, xh, y), true;
        return two_const(4);
    }
    if (xh < 0x413921fbu) return n = cw_ext(x, xh, y), true;
    return false;
}

inline double sin(double x) {
    const double a = std::fabs(x);
    if (a < 0.7853981633974483) return a < 1.4901161193847656e-08 ? x : sin_k(x);
    if (!std::isfinite(x)) return NAN;
    int n;
    Two y;
    if (!reduce(x, n, y)) return NAN;
    switch (n & 3) {
        case 0: return sin_k(y);
        case 1: return cos_k(y);
        case 2: return -sin_k(y);
        default: return -cos_k(y);
    }
}
inline double cos(double x) {
    const double a = std::fabs(x);
    if (a < 0.7853981633974483) return a < 1.0536712127723509e-08 ? 1.0 : cos_k(Two{x, 0.0});
    if (!std::isfinite(x)) return NAN;
    int n;
    Two y;
    if (!reduce(x, n, y)) return NAN;
    switch (n & 3) {
        case 0: return cos_k(y);
        case 1: return -sin_k(y);
        case 2: return -cos_k(y);
        default: return sin_k(y);
    }
}

inline double tan_k(Two y, int kk) {  // tan_kernel(y::DoubleFloat64, k)
    static const double T[13] = {3.33333333333334091986e-01, 1.33333333333201242699e-01, 5.39682539762260521377e-02, 2.18694882948595424599e-02,
                                 8.86323982359930005737e-03, 3.59207910759131235356e-03, 1.45620945432529025516e-03, 5.88041240820264096874e-04,
                                 2.46463134818469906812e-04, 7.81794442939557092300e-05, 7.14072491382608190305e-05, -1.85586374855275456654e-05,
                                 2.59073051863633712884e-05};
    double x = y.hi, t = y.lo;
    const bool big = std::fabs(x) >= 0.6744;
    if (big) {
        if (x < 0.0) x = -x, t = -t;
        x = (k::PIO4 - x) + (k::PIO4_LO - t);
        t = 0.0;
    }
    double z = x * x, w = z * z;
    double r = horner(w, {T[1], T[3], T[5], T[7], T[9], T[11]});
    double v = z * horner(w, {T[2], T[4], T[6], T[8], T[10], T[12]});
    double s = z * x;
    r = t + z * (s * (r + v) + t);
    r += T[0] * s;
    w = x + r;
    if (big) {
        v = static_cast<double>(kk);
        return (std::signbit(y.hi) ? -1.0 : 1.0) * (v - 2.0 * (x - (w * w / (w + v) - r)));
    }
    if (kk == 1) return w;
    z = zero_low(w);
    v = r - (z - x);
    const double a = -1.0 / w, tt = zero_low(a);
    s = 1.0 + tt * z;
    return tt + a * (s + tt * v);
}
inline double tan(double x) {
    const double a = std::fabs(x);
    if (a < 0.7853981633974483) return a < 7.450580596923828e-09 ? x : tan_k(Two{x, 0.0}, 1);
    if (!std::isfinite(x)) return NAN;
    int n;
    Two y;
    if (!reduce(x, n, y)) return NAN;
    return tan_k(y, (n & 1) ? -1 : 1);
}

inline double atan(double x) {
    static const double aT[11] = {3.33333333333329318027e-01,  -1.99999999998764832476e-01, 1.42857142725034663711e-01, -1.11111104054623557880e-01,
                                  9.09088713343650656196e-02,  -7.69187620504482999495e-02, 6.66107313738753120669e-02, -5.83357013379057348645e-02,
                                  4.97687799461593236017e-02,  -3.65315727442169155270e-02, 1.62858201153657823623e-02};
    static const double HI[4] = {4.63647609000806093515e-01, 7.85398163397448278999e-01, 9.82793723247329054082e-01, 1.57079632679489655800e+00};
    static const double LO[4] = {2.26987774529616870924e-17, 3.06161699786838301793e-17, 1.39033110312309984516e-17, 6.12323399573676603587e-17};
    if (std::isnan(x)) return x;
    const double a = std::fabs(x);
    if (a >= 0x1p66) return std::copysign(1.5707963267948966, x);
    int id = -1;
    double t = x;
    if (a < 7.0 / 16) {
        if (a < 0x1p-27) return x;
    } else if (a < 11.0 / 16) {
        id = 0, t = (2.0 * a - 1.0) / (2.0 + a);
    } else if (a < 19.0 / 16) {
        id = 1, t = (a - 1.0) / (a + 1.0);
    } else if (a < 39.0 / 16) {
        id = 2, t = (a - 1.5) / (1.0 + 1.5 * a);
    } else {
        id = 3, t = -1.0 / a;
    }
    const double z = t * t, w = z * z;
    const double p = z * horner(w, {aT[0], aT[2], aT[4], aT[6], aT[8], aT[10]});
    const double q = w * horner(w, {aT[1], aT[3], aT[5], aT[7], aT[9]});
    if (id < 0) return t - t * (p + q);
    return std::copysign(HI[id] - ((t * (p + q) - LO[id]) - t), x);
}
inline double atan2(double y, double x) {  // Base.atan(y, x)
    if (std::isnan(x) || std::isnan(y)) return NAN;
    if (x == 1.0) return jlm::atan(y);
    const int m = 2 * (std::signbit(x) ? 1 : 0) + (std::signbit(y) ? 1 : 0);
    if (y == 0.0) return m < 2 ? y : (m == 2 ? k::PI : -k::PI);
    if (x == 0.0) return std::copysign(k::PI / 2, y);
    if (std::isinf(x)) {
        if (std::isinf(y)) return m == 0 ? k::PI / 4 : m == 1 ? -k::PI / 4 : m == 2 ? 3 * k::PI / 4 : -3 * k::PI / 4;
        return m == 0 ? 0.0 : m == 1 ? -0.0 : m == 2 ? k::PI : -k::PI;
    }
    if (std::isinf(y)) return std::copysign(k::PI / 2, y);
    const int32_t e = static_cast<int32_t>(hi_abs(y) - hi_abs(x)) >> 20;
    double z;
    int mm = m;
    if (e > 60) {
        z = k::PI / 2 + 0.5 * k::PI_LO;
        mm &= 1;
    } else if (x < 0.0 && e < -60) {
        z = 0.0;
    } else {
        z = jlm::atan(std::fabs(y / x));
    }
    switch (mm) {
        case 0: return z;
        case 1: return -z;
        case 2: return k::PI - (z - k::PI_LO);
        default: return (z - k::PI_LO) - k::PI;
    }
}

inline double asin_ratio(double t) {  // arc_tRt
    const double p = t * horner(t, {1.66666666666666657415e-01, -3.25565818622400915405e-01, 2.01212532134862925881e-01, -4.00555345006794114027e-02,
                                    7.91534994289814532176e-04, 3.47933107596021167570e-05});
    const double q = horner(t, {1.0, -2.40339491173441421878e+00, 2.02094576023350569471e+00, -6.88283971605453293030e-01, 7.70381505559019352791e-02});
    return p / q;
}
inline double acos(double x) {
    if (std::isnan(x)) return x;
    const double a = std::fabs(x);
    if (a >= 1.0) return a == 1.0 ? (x > 0.0 ? 0.0 : k::PI) : NAN;
    if (a < 0.5) return a < 0x1p-57 ? k::PI / 2 : k::PIO2_HI - (x - (k::PIO2_LO - x * asin_ratio(x * x)));
    const double z = (1.0 - a) * 0.5, rr = asin_ratio(z), s = std::sqrt(z);
    if (x < 0.0) return k::PI - 2.0 * (s + (rr * s - k::PIO2_LO));
    const double df = zero_low(s), c = (z - df * df) / (s + df);
    return 2.0 * (df + (rr * s + c));
}

}  // namespace jlm
