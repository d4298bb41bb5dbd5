// Arithmetic core of the covariance assembly (K1 / K2 / K8): the operations of the reference, rounded like the
// reference's, at a fraction of the instruction count of the IEEE library routines.  Plain C++ over fma / ldexp / rint so
// that the SAME text runs on the host (tests/native/check_rbfmath.cpp compares every function with the IEEE division,
// sqrt and libm exp on millions of samples) and on the device.
//
//   rbf_div(a, b, y)   = RN(a / b) for y = RN(1 / b): q0 = a y, r = a - q0 b (exact, one fma), q = q0 + r y (Markstein's
//                        correction: the correctly rounded quotient, no denormals / overflow on this path: the callers'
//                        a = d*d >= 0, b = a squared length scale).  3 full-rate operations instead of the ~10 quarter-rate
//                        ones of v_div_scale / v_rcp / v_div_fmas / v_div_fixup.
//   rbf_sqrt(s)        = RN(sqrt(s)): v_rsq_f64 seed (2^-22), one coupled Goldschmidt step (g -> sqrt, h -> 1/(2 sqrt)),
//                        two exact-residual corrections; the last one starts from an error ~2^-88, far below the
//                        2^-107 distance of a square root from a rounding boundary, so the result is the correctly rounded
//                        root.  s = 0 gives 0, denormal s is scaled.
//   rbf_exp(x), x <= 0 = exp(x) to < 1 ulp: x = k ln2/64 + r (Cody-Waite, two-part ln2/64, |r| <= ln2/128),
//                        exp(x) = 2^(k >> 6) * T[k & 63] * (1 + p(r)) with T = 2^(j/64) as a hi + lo pair (table in LDS) and
//                        p of degree 6.  ~14 full-rate operations instead of the library's ~24.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>
#ifdef __HIPCC__
#define RBF_FN __host__ __device__ __forceinline__
#else
#define RBF_FN static inline
#endif
static inline double rbf_host_rsq_seed(double a) {     /* 1/sqrt(a) cut to 22 bits: a seed of v_rsq_f64's grade (host check only) */
    double y = 1.0 / std::sqrt(a);
    uint64_t b; std::memcpy(&b, &y, 8); b &= ~((1ull << 30) - 1); std::memcpy(&y, &b, 8);
    return y;
}
#ifdef __HIP_DEVICE_COMPILE__
#define RBF_FMA(a, b, c) __builtin_fma(a, b, c)
#define RBF_RINT(a) __builtin_rint(a)
#define RBF_LDEXP(a, n) __builtin_ldexp(a, n)
#define RBF_RSQ(a) __builtin_amdgcn_rsq(a)
#else
#define RBF_FMA(a, b, c) std::fma(a, b, c)
#define RBF_RINT(a) std::nearbyint(a)
#define RBF_LDEXP(a, n) std::ldexp(a, n)
#define RBF_RSQ(a) rbf_host_rsq_seed(a)
#endif

#define RBF_INV_L 0x1.71547652b82fep+6      /* 64 / ln 2 */
#define RBF_L_HI 0x1.62e42fef00000p-7       /* ln 2 / 64, low 20 bits of the significand zero: k * L_HI is exact for |k| < 2^20 */
#define RBF_L_LO 0x1.473de6af278edp-40

// 2^(j/64), j = 0..63, as (hi, lo) pairs: hi = RN(2^(j/64)), lo = RN(2^(j/64) - hi)
#define RBF_EXP_TABLE_INIT { \
0x1.0000000000000p+0, 0x0.0p+0, \
    0x1.02c9a3e778061p+0, -0x1.19083535b085dp-56, \
    0x1.059b0d3158574p+0, 0x1.d73e2a475b465p-55, \
    0x1.0874518759bc8p+0, 0x1.186be4bb284ffp-57, \
    0x1.0b5586cf9890fp+0, 0x1.8a62e4adc610bp-54, \
    0x1.0e3ec32d3d1a2p+0, 0x1.03a1727c57b53p-59, \
    0x1.11301d0125b51p+0, -0x1.6c51039449b3ap-54, \
    0x1.1429aaea92de0p+0, -0x1.32fbf9af1369ep-54, \
    0x1.172b83c7d517bp+0, -0x1.19041b9d78a76p-55, \
    0x1.1a35beb6fcb75p+0, 0x1.e5b4c7b4968e4p-55, \
    0x1.1d4873168b9aap+0, 0x1.e016e00a2643cp-54, \
    0x1.2063b88628cd6p+0, 0x1.dc775814a8495p-55, \
    0x1.2387a6e756238p+0, 0x1.9b07eb6c70573p-54, \
    0x1.26b4565e27cddp+0, 0x1.2bd339940e9d9p-55, \
    0x1.29e9df51fdee1p+0, 0x1.612e8afad1255p-55, \
    0x1.2d285a6e4030bp+0, 0x1.0024754db41d5p-54, \
    0x1.306fe0a31b715p+0, 0x1.6f46ad23182e4p-55, \
    0x1.33c08b26416ffp+0, 0x1.32721843659a6p-54, \
    0x1.371a7373aa9cbp+0, -0x1.63aeabf42eae2p-54, \
    0x1.3a7db34e59ff7p+0, -0x1.5e436d661f5e3p-56, \
    0x1.3dea64c123422p+0, 0x1.ada0911f09ebcp-55, \
    0x1.4160a21f72e2ap+0, -0x1.ef3691c309278p-58, \
    0x1.44e086061892dp+0, 0x1.89b7a04ef80d0p-59, \
    0x1.486a2b5c13cd0p+0, 0x1.3c1a3b69062f0p-56, \
    0x1.4bfdad5362a27p+0, 0x1.d4397afec42e2p-56, \
    0x1.4f9b2769d2ca7p+0, -0x1.4b309d25957e3p-54, \
    0x1.5342b569d4f82p+0, -0x1.07abe1db13cadp-55, \
    0x1.56f4736b527dap+0, 0x1.9bb2c011d93adp-54, \
    0x1.5ab07dd485429p+0, 0x1.6324c054647adp-54, \
    0x1.5e76f15ad2148p+0, 0x1.ba6f93080e65ep-54, \
    0x1.6247eb03a5585p+0, -0x1.383c17e40b497p-54, \
    0x1.6623882552225p+0, -0x1.bb60987591c34p-54, \
    0x1.6a09e667f3bcdp+0, -0x1.bdd3413b26456p-54, \
    0x1.6dfb23c651a2fp+0, -0x1.bbe3a683c88abp-57, \
    0x1.71f75e8ec5f74p+0, -0x1.16e4786887a99p-55, \
    0x1.75feb564267c9p+0, -0x1.0245957316dd3p-54, \
    0x1.7a11473eb0187p+0, -0x1.41577ee04992fp-55, \
    0x1.7e2f336cf4e62p+0, 0x1.05d02ba15797ep-56, \
    0x1.82589994cce13p+0, -0x1.d4c1dd41532d8p-54, \
    0x1.868d99b4492edp+0, -0x1.fc6f89bd4f6bap-54, \
    0x1.8ace5422aa0dbp+0, 0x1.6e9f156864b27p-54, \
    0x1.8f1ae99157736p+0, 0x1.5cc13a2e3976cp-55, \
    0x1.93737b0cdc5e5p+0, -0x1.75fc781b57ebcp-57, \
    0x1.97d829fde4e50p+0, -0x1.d185b7c1b85d1p-54, \
    0x1.9c49182a3f090p+0, 0x1.c7c46b071f2bep-56, \
    0x1.a0c667b5de565p+0, -0x1.359495d1cd533p-54, \
    0x1.a5503b23e255dp+0, -0x1.d2f6edb8d41e1p-54, \
    0x1.a9e6b5579fdbfp+0, 0x1.0fac90ef7fd31p-54, \
    0x1.ae89f995ad3adp+0, 0x1.7a1cd345dcc81p-54, \
    0x1.b33a2b84f15fbp+0, -0x1.2805e3084d708p-57, \
    0x1.b7f76f2fb5e47p+0, -0x1.5584f7e54ac3bp-56, \
    0x1.bcc1e904bc1d2p+0, 0x1.23dd07a2d9e84p-55, \
    0x1.c199bdd85529cp+0, 0x1.11065895048ddp-55, \
    0x1.c67f12e57d14bp+0, 0x1.2884dff483cadp-54, \
    0x1.cb720dcef9069p+0, 0x1.503cbd1e949dbp-56, \
    0x1.d072d4a07897cp+0, -0x1.cbc3743797a9cp-54, \
    0x1.d5818dcfba487p+0, 0x1.2ed02d75b3707p-55, \
    0x1.da9e603db3285p+0, 0x1.c2300696db532p-54, \
    0x1.dfc97337b9b5fp+0, -0x1.1a5cd4f184b5cp-54, \
    0x1.e502ee78b3ff6p+0, 0x1.39e8980a9cc8fp-55, \
    0x1.ea4afa2a490dap+0, -0x1.e9c23179c2893p-54, \
    0x1.efa1bee615a27p+0, 0x1.dc7f486a4b6b0p-54, \
    0x1.f50765b6e4540p+0, 0x1.9d3e12dd8a18bp-54, \
    0x1.fa7c1819e90d8p+0, 0x1.74853f3a5931ep-55, \
}

RBF_FN double rbf_div(double a, double b, double y) {
#pragma clang fp contract(off)
    const double q0 = a * y;
    const double r = RBF_FMA(-q0, b, a);
    return RBF_FMA(r, y, q0);
}

RBF_FN double rbf_sqrt(double s) {
#pragma clang fp contract(off)
    // scale denormal / tiny arguments (never on the hot path: s is a squared Mahalanobis distance)
    const bool tiny = s < 0x1p-767;
    const double ss = tiny ? s * 0x1p+256 : s;
    const double y = RBF_RSQ(ss);
    double g = ss * y, h = 0.5 * y;
    const double r = RBF_FMA(-h, g, 0.5);
    g = RBF_FMA(g, r, g);
    h = RBF_FMA(h, r, h);
    double d = RBF_FMA(-g, g, ss);
    g = RBF_FMA(d, h, g);
    d = RBF_FMA(-g, g, ss);
    g = RBF_FMA(d, h, g);
    g = tiny ? g * 0x1p-128 : g;
    return s == 0. ? 0. : g;
}

// rbf_sqrt for the RBF argument: RN(sqrt(s + 2^-600)) without the zero / denormal handling.  For s >= 2^-547 the sum is
// exactly s, so this IS RN(sqrt(s)); below that the caller's exp(-dist^2 / 2) equals 1.0 for either root (|argument| <
// 2^-548); negative s still gives NaN like the reference's sqrt.
RBF_FN double rbf_sqrt_arg(double s) {
#pragma clang fp contract(off)
    const double ss = s + 0x1p-600;
    const double y = RBF_RSQ(ss);
    double g = ss * y, h = 0.5 * y;
    const double r = RBF_FMA(-h, g, 0.5);
    g = RBF_FMA(g, r, g);
    h = RBF_FMA(h, r, h);
    double d = RBF_FMA(-g, g, ss);
    g = RBF_FMA(d, h, g);
    d = RBF_FMA(-g, g, ss);
    return RBF_FMA(d, h, g);
}

// tab: 64 (hi, lo) pairs (LDS on the device)
RBF_FN double rbf_exp(double x, const double* tab) {
#pragma clang fp contract(off)
    x = x < -1000. ? -1000. : x;                      // exp underflows to 0 below -745.2; keeps k inside int range
    // k = nearest integer to x * 64 / ln 2 by the shifter constant 1.5 * 2^52: the sum's low mantissa bits ARE the integer
    // (two's complement), so neither a round-to-integer nor a convert instruction is needed
    const double kb = RBF_FMA(x, RBF_INV_L, 0x1.8p+52);
    const double k = kb - 0x1.8p+52;
    double r = RBF_FMA(-k, RBF_L_HI, x);
    r = RBF_FMA(-k, RBF_L_LO, r);
    int64_t kbits;
    __builtin_memcpy(&kbits, &kb, 8);
    const int ki = (int)(uint32_t)kbits;
    const int j = ki & 63, m = ki >> 6;
    const double th = tab[2 * j], tl = tab[2 * j + 1];
    const double r2 = r * r;
    double p = RBF_FMA(r, 0x1.6c16c16c16c17p-10, 0x1.1111111111111p-7);      // 1/720, 1/120
    p = RBF_FMA(p, r, 0x1.5555555555555p-5);                                  // 1/24
    p = RBF_FMA(p, r, 0x1.5555555555555p-3);                                  // 1/6
    p = RBF_FMA(p, r, 0.5);
    p = RBF_FMA(p, r2, r);                                                    // e^r - 1
    const double v = RBF_FMA(th, p, tl) + th;
    return RBF_LDEXP(v, m);
}
