// Host check of csrc/rbfmath.hpp against IEEE division, sqrt and libm exp (compile with -ffp-contract=off).
#include "../../cosmogp_amd/csrc/rbfmath.hpp"
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
static int64_t ulps(double a, double b) {
    int64_t ia, ib; std::memcpy(&ia, &a, 8); std::memcpy(&ib, &b, 8);
    if (ia < 0) ia = INT64_MIN - ia;
    if (ib < 0) ib = INT64_MIN - ib;
    return ia > ib ? ia - ib : ib - ia;
}
int main() {
    static const double tab[128] = RBF_EXP_TABLE_INIT;
    std::mt19937_64 rng(12345);
    std::uniform_real_distribution<double> U(0., 1.);
    long bad_div = 0, bad_sqrt = 0; int64_t worst_exp = 0; long n_exp2 = 0;
    const long N = 20000000;
    for (long i = 0; i < N; ++i) {
        // division: a = d*d over many magnitudes, b = squared length scale
        const double d = std::ldexp(U(rng) + 0.5, (int)(U(rng) * 40) - 20), b = std::ldexp(U(rng) + 0.5, (int)(U(rng) * 40) - 20);
        const double a = d * d, y = 1.0 / b;
        if (rbf_div(a, b, y) != a / b) ++bad_div;
        // sqrt
        const double s = std::ldexp(U(rng) + 0.5, (int)(U(rng) * 120) - 60);
        if (rbf_sqrt(s) != std::sqrt(s) || rbf_sqrt_arg(s) != std::sqrt(s)) ++bad_sqrt;
        // exp on [-760, 0], dense near 0 too
        const double x = (i & 1) ? -760. * U(rng) : -std::ldexp(U(rng), -(int)(U(rng) * 30));
        const int64_t u = ulps(rbf_exp(x, tab), std::exp(x));
        if (u > worst_exp) worst_exp = u;
        if (u >= 2) ++n_exp2;
    }
    const double specials[] = {0., -0., -1e-320, -1e-300, -708.3, -745.0, -745.2, -746., -800., -1e5, -1e300};
    int bad_special = 0;
    for (double x : specials) if (ulps(rbf_exp(x, tab), std::exp(x)) > 1) { ++bad_special; printf("exp(%g): %a vs %a\n", x, rbf_exp(x, tab), std::exp(x)); }
    for (double s : {0., 1e-320, 1e-200, 0x1p-548}) {          // rbf_sqrt_arg below its exact range: the RBF value is 1 either way
        const double d = rbf_sqrt_arg(s);
        if (rbf_exp(-0.5 * (d * d), tab) != 1.0 || std::exp(-0.5 * s) != 1.0) { ++bad_special; printf("sqrt_arg tiny %g\n", s); }
    }
    if (!(rbf_sqrt_arg(-1.0) != rbf_sqrt_arg(-1.0))) { ++bad_special; printf("sqrt_arg(-1) not NaN\n"); }
    if (rbf_sqrt(0.) != 0. || rbf_sqrt(1e-310) != std::sqrt(1e-310) || rbf_sqrt(4.) != 2.) { ++bad_special; printf("sqrt special\n"); }
    printf("%ld samples: division mismatches %ld, sqrt mismatches %ld, exp worst %lld ulp (%ld samples >= 2 ulp), specials bad %d\n", N, bad_div,
           bad_sqrt, (long long)worst_exp, n_exp2, bad_special);
    return (bad_div || bad_sqrt || worst_exp > 1 || bad_special) ? 1 : 0;
}
