// bmo_jlmath.hpp — the elementary functions of the step path as Julia's Base computes them.
//
// The reference calls `sin`, `cos` (fresnel_coefficients, OpticUtils.jl:121-131), `acos` (angle3d, LinearAlgebraUtils.jl:103-108), `tan`
// (gauss_parameters, Gaussian.jl:326-336) and the two-argument `atan` (Gouy phase, Gaussian.jl:345).  Julia does not call the C library for
// these: Base.Math carries its own ports of the FreeBSD msun / fdlibm routines (base/special/trig.jl, base/special/rem_pio2.jl; Julia's stdlib
// is not under /root/reference — SURVEY §8c "third-party arithmetic"), with the polynomial evaluations written as `@horner` = nested `muladd`,
// which is a fused multiply-add on every machine that has one.  They are faithful (< 1 ulp), NOT correctly rounded, so neither glibc nor ocml
// gives the reference's bits: at Brewster's angle Base's sin(atan(1.5)) is 0.8320502943378436, one unit below the correctly rounded value every
// C library here returns, and that unit is what makes the reference's own test `real(rp) ≈ 0` (runtests.jl:157; `≈ 0` means == 0) hold.
//
// This file restates the published algorithms (Sun's fdlibm 5.3 k_sin.c, k_cos.c, k_tan.c as revised in FreeBSD, e_rem_pio2.c's Cody-Waite
// branches, s_atan.c, e_atan2.c, e_acos.c) in the form Base gives them: same constants, same operation order, an explicit fma wherever Base
// writes muladd, ordinary rounded operations everywhere else (the library is compiled with -ffp-contract=off).  Device and host evaluate them
// with the same IEEE operations, so engine, emulator and oracle agree bit for bit; the oracle has its own copy (oracle/jl_trig.hpp), written
// separately.  Pins: tests/test_jl_trig.py (1 ulp against mpmath on dense samples, the Brewster zero, emulator == oracle on 10^5 arguments),
// bmo_selftest (device == host).  Arguments beyond 2^20 pi/2 would need Base's Payne-Hanek reduction, which is not restated: the step path's
// angles lie in [-pi, pi], and such an argument returns NaN here rather than something approximate.
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

namespace bmo {
namespace jl {

#if defined(__HIPCC__)
#define BMO_JL __host__ __device__ __forceinline__
#else
#define BMO_JL inline __attribute__((always_inline))
#endif

BMO_JL double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }  // Base.muladd on FMA hardware
BMO_JL uint64_t bits_of(double x) {
    uint64_t u;
    memcpy(&u, &x, 8);
    return u;
}
BMO_JL double from_bits(uint64_t u) {
    double x;
    memcpy(&x, &u, 8);
    return x;
}
BMO_JL uint32_t poshighword(double x) { return (uint32_t)(bits_of(x) >> 32) & 0x7fffffffu; }
BMO_JL double clear_low_word(double x) { return from_bits(bits_of(x) & 0xffffffff00000000ull); }

struct dd {  // Base.Math.DoubleFloat64: a value and its tail
    double hi, lo;
};

// ---- kernels on [-pi/4, pi/4] (trig.jl sin_kernel / cos_kernel; k_sin.c, k_cos.c)
BMO_JL double sin_kernel(double y) {
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double y2 = y * y, y4 = y2 * y2;
    const double r = fma_(y2, fma_(y2, S4, S3), S2) + y2 * y4 * fma_(y2, S6, S5);
    const double y3 = y2 * y;
    return y + y3 * (S1 + y2 * r);
}
BMO_JL double sin_kernel(dd y) {
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double y2 = y.hi * y.hi, y4 = y2 * y2;
    const double r = fma_(y2, fma_(y2, S4, S3), S2) + y2 * y4 * fma_(y2, S6, S5);
    const double y3 = y2 * y.hi;
    return y.hi - ((y2 * (0.5 * y.lo - y3 * r) - y.lo) - y3 * S1);
}
BMO_JL double cos_kernel(dd y) {
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double y2 = y.hi * y.hi, y4 = y2 * y2;
    const double r = y2 * fma_(y2, fma_(y2, C3, C2), C1) + y4 * y4 * fma_(y2, fma_(y2, C6, C5), C4);
    const double half_y2 = 0.5 * y2;
    const double w = 1.0 - half_y2;
    return w + (((1.0 - w) - half_y2) + (y2 * r - y.hi * y.lo));
}
BMO_JL double cos_kernel(double y) { return cos_kernel(dd{y, 0.0}); }  // (y.hi * 0.0 subtracts a zero: the same sum)

// ---- argument reduction (rem_pio2.jl rem_pio2_kernel; e_rem_pio2.c): x = n pi/2 + y, |y| <= pi/4
BMO_JL dd cody_waite_2c_pio2(double x, double fn) {
    const double pio2_1 = 1.57079632673412561417e+00, pio2_1t = 6.07710050650619224932e-11;
    const double z = fma_(-fn, pio2_1, x);
    const double y1 = fma_(-fn, pio2_1t, z);
    const double y2 = fma_(-fn, pio2_1t, z - y1);
    return dd{y1, y2};
}
BMO_JL int cody_waite_ext_pio2(double x, uint32_t xhp, dd& y) {
    const double pio2_1 = 1.57079632673412561417e+00, pio2_1t = 6.07710050650619224932e-11, pio2_2 = 6.07710050630396597660e-11,
                 pio2_2t = 2.02226624879595063154e-21, pio2_3 = 2.02226624871116645580e-21, pio2_3t = 8.47842766036889956997e-32;
    const double fn = rint(x * 6.36619772367581382433e-01);  // round(x * 2/pi), ties to even like Base.round
    double r = fma_(-fn, pio2_1, x);
    double w = fn * pio2_1t;  // 1st round, good to 85 bits
    const int j = (int)(xhp >> 20);
    double y1 = r - w;
    int i = j - (int)((poshighword(y1) >> 20) & 0x7ff);
    if (i > 16) {  // 2nd round, good to 118 bits
        double t = r;
        w = fn * pio2_2;
        r = t - w;
        w = fma_(fn, pio2_2t, -((t - r) - w));
        y1 = r - w;
        i = j - (int)((poshighword(y1) >> 20) & 0x7ff);
        if (i > 49) {  // 3rd round, 151 bits
            t = r;
            w = fn * pio2_3;
            r = t - w;
            w = fma_(fn, pio2_3t, -((t - r) - w));
            y1 = r - w;
        }
    }
    y = dd{y1, (r - y1) - w};
    return (int)fn;
}
// returns false beyond 2^20 pi/2 (Payne-Hanek range: not restated)
BMO_JL bool rem_pio2_kernel(double x, int& n, dd& y) {
    const uint32_t xhp = poshighword(x);
    if (xhp <= 0x400f6a7au) {                    // |x| ~<= 5pi/4
        if ((xhp & 0xfffffu) == 0x921fbu) {      // |x| ~= pi/2 or 2pi/2
            n = cody_waite_ext_pio2(x, xhp, y);
            return true;
        }
        if (xhp <= 0x4002d97cu) {                // |x| ~<= 3pi/4
            n = x > 0.0 ? 1 : -1;
            y = cody_waite_2c_pio2(x, x > 0.0 ? 1.0 : -1.0);
        } else {
            n = x > 0.0 ? 2 : -2;
            y = cody_waite_2c_pio2(x, x > 0.0 ? 2.0 : -2.0);
        }
        return true;
    }
    if (xhp <= 0x401c463bu) {                    // |x| ~<= 9pi/4
        if (xhp <= 0x4015fdbcu) {                // |x| ~<= 7pi/4
            if (xhp == 0x4012d97cu) {            // |x| ~= 3pi/2
                n = cody_waite_ext_pio2(x, xhp, y);
                return true;
            }
            n = x > 0.0 ? 3 : -3;
            y = cody_waite_2c_pio2(x, x > 0.0 ? 3.0 : -3.0);
        } else {
            if (xhp == 0x401921fbu) {            // |x| ~= 4pi/2
                n = cody_waite_ext_pio2(x, xhp, y);
                return true;
            }
            n = x > 0.0 ? 4 : -4;
            y = cody_waite_2c_pio2(x, x > 0.0 ? 4.0 : -4.0);
        }
        return true;
    }
    if (xhp < 0x413921fbu) {                     // |x| ~< 2^20 pi/2
        n = cody_waite_ext_pio2(x, xhp, y);
        return true;
    }
    return false;
}

BMO_JL double nan_() { return from_bits(0x7ff8000000000000ull); }

// Base.sin(::Float64)  trig.jl
BMO_JL double sin(double x) {
    const double absx = fabs(x);
    if (absx < 0.7853981633974483) {                        // T(pi)/4
        if (absx < 1.4901161193847656e-08) return x;        // sqrt(eps)
        return sin_kernel(x);
    }
    if (!(absx < __builtin_huge_val())) return nan_();      // NaN, and Base throws a DomainError for Inf
    int n;
    dd y;
    if (!rem_pio2_kernel(x, n, y)) return nan_();
    n &= 3;
    if (n == 0) return sin_kernel(y);
    if (n == 1) return cos_kernel(y);
    if (n == 2) return -sin_kernel(y);
    return -cos_kernel(y);
}
// Base.cos(::Float64)
BMO_JL double cos(double x) {
    const double absx = fabs(x);
    if (absx < 0.7853981633974483) {
        if (absx < 1.0536712127723509e-08) return 1.0;      // sqrt(eps / 2)
        return cos_kernel(x);
    }
    if (!(absx < __builtin_huge_val())) return nan_();
    int n;
    dd y;
    if (!rem_pio2_kernel(x, n, y)) return nan_();
    n &= 3;
    if (n == 0) return cos_kernel(y);
    if (n == 1) return -sin_kernel(y);
    if (n == 2) return -cos_kernel(y);
    return sin_kernel(y);
}

// trig.jl tan_kernel(y::DoubleFloat64, k); k_tan.c (FreeBSD): tan(y) for k = 1, -1 / tan(y) for k = -1
BMO_JL double tan_kernel(dd y, int k) {
    const double T0 = 3.33333333333334091986e-01, T1 = 1.33333333333201242699e-01, T2 = 5.39682539762260521377e-02, T3 = 2.18694882948595424599e-02,
                 T4 = 8.86323982359930005737e-03, T5 = 3.59207910759131235356e-03, T6 = 1.45620945432529025516e-03, T7 = 5.88041240820264096874e-04,
                 T8 = 2.46463134818469906812e-04, T9 = 7.81794442939557092300e-05, T10 = 7.14072491382608190305e-05, T11 = -1.85586374855275456654e-05,
                 T12 = 2.59073051863633712884e-05;
    const double pio4 = 7.85398163397448278999e-01, pio4lo = 3.06161699786838301793e-17;
    double yhi = y.hi, ylo = y.lo;
    const bool big = fabs(yhi) >= 0.6744;
    if (big) {
        if (yhi < 0.0) {
            yhi = -yhi;
            ylo = -ylo;
        }
        yhi = (pio4 - yhi) + (pio4lo - ylo);
        ylo = 0.0;
    }
    double z = yhi * yhi;
    double w = z * z;
    // y^5 (T1 + y^4 T3 + ... + y^20 T11)  +  y^5 y^2 (T2 + y^4 T4 + ... + y^20 T12)
    double r = fma_(w, fma_(w, fma_(w, fma_(w, fma_(w, T11, T9), T7), T5), T3), T1);
    double v = z * fma_(w, fma_(w, fma_(w, fma_(w, fma_(w, T12, T10), T8), T6), T4), T2);
    double s = z * yhi;
    r = ylo + z * (s * (r + v) + ylo);
    r += T0 * s;
    w = yhi + r;
    if (big) {
        v = (double)k;
        const double sg = y.hi < 0.0 ? -1.0 : 1.0;  // 1 - ((hx >> 30) & 2)
        return sg * (v - 2.0 * (yhi - (w * w / (w + v) - r)));
    }
    if (k == 1) return w;
    // -1 / (yhi + r) to full accuracy
    z = clear_low_word(w);
    v = r - (z - yhi);  // z + v = r + yhi
    const double a = -1.0 / w;
    const double t = clear_low_word(a);
    s = 1.0 + t * z;
    return t + a * (s + t * v);
}
// Base.tan(::Float64)
BMO_JL double tan(double x) {
    const double absx = fabs(x);
    if (absx < 0.7853981633974483) {
        if (absx < 7.450580596923828e-09) return x;  // 2^-27: x + x^3 / 3 rounds to x (and -0.0 stays -0.0)
        return tan_kernel(dd{x, 0.0}, 1);
    }
    if (!(absx < __builtin_huge_val())) return nan_();
    int n;
    dd y;
    if (!rem_pio2_kernel(x, n, y)) return nan_();
    return tan_kernel(y, (n & 1) ? -1 : 1);
}

// Base.atan(::Float64)  trig.jl; s_atan.c
BMO_JL double atan(double x) {
    const double aT0 = 3.33333333333329318027e-01, aT1 = -1.99999999998764832476e-01, aT2 = 1.42857142725034663711e-01, aT3 = -1.11111104054623557880e-01,
                 aT4 = 9.09088713343650656196e-02, aT5 = -7.69187620504482999495e-02, aT6 = 6.66107313738753120669e-02, aT7 = -5.83357013379057348645e-02,
                 aT8 = 4.97687799461593236017e-02, aT9 = -3.65315727442169155270e-02, aT10 = 1.62858201153657823623e-02;
    const double absx = fabs(x);
    if (x != x) return x;
    if (absx >= 7.378697629483821e19) return copysign(1.5707963267948966, x);  // 2^66
    double hi = 0.0, lo = 0.0, t = x;
    bool reduced = false;
    if (absx < 0.4375) {
        if (absx < 7.450580596923828e-09) return x;  // 2^-27
    } else {
        reduced = true;
        if (absx < 1.1875) {
            if (absx < 0.6875) {
                hi = 4.63647609000806093515e-01, lo = 2.26987774529616870924e-17;
                t = (2.0 * absx - 1.0) / (2.0 + absx);
            } else {
                hi = 7.85398163397448278999e-01, lo = 3.06161699786838301793e-17;
                t = (absx - 1.0) / (absx + 1.0);
            }
        } else if (absx < 2.4375) {
            hi = 9.82793723247329054082e-01, lo = 1.39033110312309984516e-17;
            t = (absx - 1.5) / (1.0 + 1.5 * absx);
        } else {
            hi = 1.57079632679489655800e+00, lo = 6.12323399573676603587e-17;
            t = -1.0 / absx;
        }
    }
    const double t2 = t * t, t4 = t2 * t2;
    const double p = t2 * fma_(t4, fma_(t4, fma_(t4, fma_(t4, fma_(t4, aT10, aT8), aT6), aT4), aT2), aT0);
    const double q = t4 * fma_(t4, fma_(t4, fma_(t4, fma_(t4, aT9, aT7), aT5), aT3), aT1);
    if (!reduced) return t - t * (p + q);
    const double z = hi - ((t * (p + q) - lo) - t);
    return copysign(z, x);
}
// Base.atan(y::Float64, x::Float64)  trig.jl; e_atan2.c
BMO_JL double atan2(double y, double x) {
    const double pi = 3.141592653589793, pi_lo = 1.2246467991473531772e-16;
    if (x != x || y != y) return nan_();
    if (x == 1.0) return atan(y);
    const int m = 2 * (int)(bits_of(x) >> 63) + (int)(bits_of(y) >> 63);
    if (y == 0.0) {
        if (m == 0 || m == 1) return y;
        return m == 2 ? pi : -pi;
    }
    if (x == 0.0) return copysign(pi / 2, y);
    const double inf = __builtin_huge_val();
    if (fabs(x) == inf) {
        if (fabs(y) == inf) {
            if (m == 0) return pi / 4;
            if (m == 1) return -pi / 4;
            if (m == 2) return 3 * pi / 4;
            return -3 * pi / 4;
        }
        if (m == 0) return 0.0;
        if (m == 1) return -0.0;
        return m == 2 ? pi : -pi;
    }
    if (fabs(y) == inf) return copysign(pi / 2, y);
    const int32_t k = ((int32_t)(poshighword(y) - poshighword(x))) >> 20;
    double z;
    int mm = m;
    if (k > 60) {  // |y / x| > 2^60
        z = pi / 2 + 0.5 * pi_lo;
        mm &= 1;
    } else if (x < 0.0 && k < -60) {
        z = 0.0;
    } else {
        z = atan(fabs(y / x));
    }
    if (mm == 0) return z;
    if (mm == 1) return -z;
    if (mm == 2) return pi - (z - pi_lo);
    return (z - pi_lo) - pi;
}

// Base.acos(::Float64)  trig.jl; e_acos.c
BMO_JL double arc_tRt(double t) {
    const double pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01, pS2 = 2.01212532134862925881e-01, pS3 = -4.00555345006794114027e-02,
                 pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05, qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00,
                 qS3 = -6.88283971605453293030e-01, qS4 = 7.70381505559019352791e-02;
    const double p = t * fma_(t, fma_(t, fma_(t, fma_(t, fma_(t, pS5, pS4), pS3), pS2), pS1), pS0);
    const double q = fma_(t, fma_(t, fma_(t, fma_(t, qS4, qS3), qS2), qS1), 1.0);
    return p / q;
}
BMO_JL double acos(double x) {
    const double pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17, pi = 3.141592653589793;
    const double absx = fabs(x);
    if (x != x) return x;
    if (absx >= 1.0) {
        if (absx == 1.0) return x > 0.0 ? 0.0 : pi;
        return nan_();  // Base throws a DomainError
    }
    if (absx < 0.5) {
        if (absx < 6.938893903907228e-18) return pi / 2;  // 2^-57
        return pio2_hi - (x - (pio2_lo - x * arc_tRt(x * x)));
    }
    const double z = (1.0 - absx) * 0.5;
    const double zRz = arc_tRt(z);
    const double s = sqrt(z);
    if (x < 0.0) return pi - 2.0 * (s + (zRz * s - pio2_lo));
    const double df = clear_low_word(s);
    const double c = (z - df * df) / (s + df);
    return 2.0 * (df + (zRz * s + c));
}

}  // namespace jl
}  // namespace bmo
