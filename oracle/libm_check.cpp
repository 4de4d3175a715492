// libm_check.cpp -- TEST INFRASTRUCTURE: compares strelka_amd/csrc/libm_flt32.h and libm_dbl64.h (the restatement of glibc's
// powf / logf / expf / log1pf and double exp / log / log10 that the kernels use for the reference's std::pow(float,float), std::log(float), std::exp(float) and
// log1p(float) calls) with the host libm, bit for bit, on N pseudo-random arguments of the domains the path uses.
// Built and run by tests/test_libm_restatement.py.
#include "../strelka_amd/csrc/libm_dbl64.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
using namespace sk_libm;
int main(int argc, char** argv)
{
    const long n = (argc > 1) ? atol(argv[1]) : 1000000;
    unsigned long long st = 88172645463325252ull;
    long badp = 0, badl = 0, bade = 0, bad1 = 0, fall = 0, badde = 0, baddl = 0, badd10 = 0, baddp = 0;
    for (long it = 0; it < n; ++it) {
        st ^= st << 13;
        st ^= st >> 7;
        st ^= st << 17;
        // powf: error probabilities of q-scores 3..70 (and neighbouring floats), exponents in (0, 1]
        const int q = 3 + int(st % 68);
        float e = float(std::pow(10.0, -0.1 * q));
        if (it & 1) e = as_f32(as_u32(e) + uint32_t((st >> 20) % 2048));
        float v = float(double((st >> 32) % 1000000) / 1000000.0);
        if (v <= 0) v = 0.25f;
        float a = powf(e, v), b = 0;
        if (!powf_glibc(e, v, b)) { fall++; b = a; }
        badp += as_u32(a) != as_u32(b);
        // logf: 2^-25 .. 1
        const float x = as_f32(0x33000000u + uint32_t((st >> 8) % 0x0c800000u));
        a = logf(x);
        if (!logf_glibc(x, b)) { fall++; b = a; }
        badl += as_u32(a) != as_u32(b);
        // expf: (-110, 0], dense near 0
        float xe = -float(double(st % 11000000) / 100000.0);
        if (it & 1) xe = -float(double((st >> 8) % 1000000) / 1.0e8);
        a = expf(xe);
        if (!expf_glibc(xe, b)) { fall++; b = a; }
        bade += as_u32(a) != as_u32(b);
        // log1pf: ~1e-19 .. 0.01 and 0.01 .. 0.41
        const float x1 = (it & 2) ? as_f32(0x20000000u + uint32_t((st >> 16) % (0x3c23d70au - 0x20000000u)))
                                  : as_f32(0x3c23d70au + uint32_t((st >> 16) % (0x3ed413d7u - 0x3c23d70au)));
        a = log1pf(x1);
        if (!log1pf_glibc(x1, b)) { fall++; b = a; }
        bad1 += as_u32(a) != as_u32(b);
        // double exp: (-760, 0] incl. the subnormal-result range, dense near 0, tiny, and a few positive
        double xd = -double(st % 760000000) / 1.0e6;
        if ((it & 3) == 1) xd = -double((st >> 8) % 1000000) / 1.0e9;
        if ((it & 3) == 2) xd = double((st >> 8) % 1000000) / 1.0e4;
        if ((it & 7) == 7) xd = -double((st >> 8) % 1000000) * 0x1p-70;
        double da = exp(xd), db = 0;
        if (!exp_glibc(xd, db)) { fall++; db = da; }
        badde += as_u64(da) != as_u64(db);
        // double log / log10: the range close to 1, and all positive normal magnitudes
        double xl = (it & 1) ? 0.9375 + double(st % 12720000) / 1.0e8 : as_f64(0x0010000000000000ull + (st % 0x7fe0000000000000ull));
        da = log(xl);
        if (!log_glibc(xl, db)) { fall++; db = da; }
        baddl += as_u64(da) != as_u64(db);
        da = log10(xl);
        if (!log10_glibc(xl, db)) { fall++; db = da; }
        badd10 += as_u64(da) != as_u64(db);
        // double log1p: [0, 0.01) as log1p_switch uses it, and up to 0.41
        const double xp = (it & 1) ? double(st % 1000000000) / 1.0e11 : as_f64(0x3c00000000000000ull + (st % (0x3fda827a00000000ull - 0x3c00000000000000ull)));
        da = log1p(xp);
        if (!log1p_glibc(xp, db)) { fall++; db = da; }
        baddp += as_u64(da) != as_u64(db);
    }
    printf("n=%ld exp mismatches %ld log mismatches %ld log10 mismatches %ld log1p mismatches %ld\n", n, badde, baddl, badd10, baddp);
    if (badde || baddl || badd10 || baddp) return 1;
    printf("n=%ld powf mismatches %ld logf mismatches %ld expf mismatches %ld log1pf mismatches %ld fallbacks %ld\n", n, badp, badl,
           bade, bad1, fall);
    return (badp || badl || bade || bad1 || fall) ? 1 : 0;
}
