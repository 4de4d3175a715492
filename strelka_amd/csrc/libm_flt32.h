// libm_flt32.h -- glibc's single-precision powf / logf, restated for the device.
//
// The reference computes adjust_joint_eprob's `std::pow(float, float)` and get_diploid_gt_lhood's `std::log(float)` with
// the host C library.  glibc (>= 2.28; this image: 2.35) implements both in sysdeps/ieee754/flt-32/{e_powf,e_logf}.c with
// the table-driven double-precision algorithms of ARM's optimized-routines: a 16-entry log table + degree-5 (powf) or
// degree-3 (logf) polynomial, and for powf a 32-entry exp2 table + cubic.  Those few dozen double operations are
// restated below (with expf and the fdlibm log1pf the somatic strand states' float log-sum needs), so the device
// produces the SAME float, bit for bit, as the reference's libm call -- the device
// library's own pow/log are accurate but round differently now and then.  x86-64 glibc runs its FMA build of these
// routines on every CPU with FMA (ifunc); the multiply-adds below are fused accordingly.
//
// Tables: the published constants of those routines (__powf_log2_data with POWF_SCALE_BITS = 0, __exp2f_data,
// __logf_data), listed with tools/libm_tables.py.  tests/test_libm_restatement.py compiles this header for the host and
// compares it with the host libm on 10^7 arguments; sk_init() repeats a short comparison and, should the host libm ever
// be a different implementation, the kernels keep using the device library's double-precision pow/log instead.
//
// Domain: x a positive normal float, |y * log2(x)| < 126 (powf).  Anything else returns `false`: the caller falls back.
//
// PROVENANCE AND LICENCE.  Nothing here comes from /root/reference.  The algorithms and constant tables are those of ARM
// Optimized Routines (math/powf.c, logf.c, expf.c and their *_data.c; (c) Arm Limited; SPDX: MIT OR Apache-2.0 WITH LLVM-exception)
// as imported into the GNU C Library (sysdeps/ieee754/flt-32/e_powf.c, e_logf.c, e_expf.c, e_exp2f_data.c, e_logf_data.c,
// e_powf_log2_data.c; LGPL-2.1-or-later); log1pf follows glibc's s_log1pf.c, which descends from FreeBSD msun / fdlibm
// ("Copyright (C) 1993 by Sun Microsystems, Inc. ... Permission to use, copy, modify, and distribute this software is freely
// granted, provided that this notice is preserved").  This file is a restatement written for this repository (new code, the same
// operation sequence and published constants); redistribution should keep this notice and the licences named above.
#pragma once

#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#define SK_HD __host__ __device__ __forceinline__
#else
#define SK_HD inline
#endif

namespace sk_libm
{

SK_HD double fma_(const double a, const double b, const double c) { return __builtin_fma(a, b, c); }
SK_HD uint32_t as_u32(const float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}
SK_HD float as_f32(const uint32_t u)
{
    float f;
    memcpy(&f, &u, 4);
    return f;
}
SK_HD uint64_t as_u64(const double f)
{
    uint64_t u;
    memcpy(&u, &f, 8);
    return u;
}
SK_HD double as_f64(const uint64_t u)
{
    double f;
    memcpy(&f, &u, 8);
    return f;
}

/// glibc logf (e_logf.c).  Returns false outside the main path (x not a positive normal number).
SK_HD bool logf_glibc(const float x, float& out)
{
    constexpr double T[16][2] = { // { 1/c, ln(c) } for the 16 sub-intervals of [OFF, 2 OFF)
        { 0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2 }, { 0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2 },
        { 0x1.49539f0f010b0p+0, -0x1.01eae7f513a67p-2 }, { 0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3 },
        { 0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3 }, { 0x1.25e227b0b8ea0p+0, -0x1.1aa2bc79c8100p-3 },
        { 0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4 }, { 0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4 },
        { 0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5 }, { 0x1.0000000000000p+0, 0x0.0p+0 },
        { 0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5 }, { 0x1.ca4b31f026aa0p-1, 0x1.c5e53aa362eb4p-4 },
        { 0x1.b2036576afce6p-1, 0x1.526e57720db08p-3 }, { 0x1.9c2d163a1aa2dp-1, 0x1.bc2860d224770p-3 },
        { 0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2 }, { 0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2 } };
    constexpr double LN2 = 0x1.62e42fefa39efp-1;
    constexpr double A[3] = { -0x1.00ea348b88334p-2, 0x1.5575b0be00b6ap-2, -0x1.ffffef20a4123p-2 };
    const uint32_t ix = as_u32(x);
    if (ix == 0x3f800000u) {
        out = 0.f;
        return true;
    }
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) return false; // zero, subnormal, negative, inf, nan
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = int((tmp >> (23 - 4)) % 16u);
    const int k = int32_t(tmp) >> 23; // arithmetic shift
    const uint32_t iz = ix - (tmp & 0xff800000u);
    const double invc = T[i][0], logc = T[i][1];
    const double z = double(as_f32(iz));
    // log(x) = log1p(z/c - 1) + log(c) + k ln2
    const double r = fma_(z, invc, -1.0);
    const double y0 = fma_(double(k), LN2, logc);
    const double r2 = r * r;
    double y = fma_(A[1], r, A[2]);
    y = fma_(A[0], r2, y);
    y = fma_(y, r2, y0 + r);
    out = float(y);
    return true;
}

/// glibc powf (e_powf.c), main path only.  Returns false for special operands and results near over/underflow.
SK_HD bool powf_glibc(const float x, const float y, float& out)
{
    constexpr double T[16][2] = { // { 1/c, log2(c) }
        { 0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2 }, { 0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2 },
        { 0x1.49539f0f010b0p+0, -0x1.7418b0a1fb77bp-2 }, { 0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2 },
        { 0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2 }, { 0x1.25e227b0b8ea0p+0, -0x1.97c1d1b3b7af0p-3 },
        { 0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3 }, { 0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4 },
        { 0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5 }, { 0x1.0000000000000p+0, 0x0.0p+0 },
        { 0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4 }, { 0x1.ca4b31f026aa0p-1, 0x1.476a9543891bap-3 },
        { 0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3 }, { 0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2 },
        { 0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2 }, { 0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2 } };
    constexpr double A[5] = { 0x1.27616c9496e0bp-2, -0x1.71969a075c67ap-2, 0x1.ec70a6ca7baddp-2, -0x1.7154748bef6c8p-1, 0x1.71547652ab82bp+0 };
    constexpr uint64_t E[32] = { // bits of 2^(i/32), exponent field pre-adjusted
        0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
        0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
        0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
        0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
        0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
        0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
        0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
        0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull };
    constexpr double SHIFT = 0x1.8000000000000p+47; // 0x1.8p52 / 32
    constexpr double C[3] = { 0x1.c6af84b912394p-5, 0x1.ebfce50fac4f3p-3, 0x1.62e42ff0c52d6p-1 };
    const uint32_t ix = as_u32(x), iy = as_u32(y);
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) return false;  // x zero, subnormal, negative, inf, nan
    if (2u * iy - 1u >= 2u * 0x7f800000u - 1u) return false;            // y zero, inf, nan
    // log2(x) = log1p(z/c - 1)/ln2 + log2(c) + k
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = int((tmp >> (23 - 4)) % 16u);
    const uint32_t top = tmp & 0xff800000u;
    const uint32_t iz = ix - top;
    const int k = int32_t(top) >> 23; // arithmetic shift
    const double invc = T[i][0], logc = T[i][1];
    const double z = double(as_f32(iz));
    const double r = fma_(z, invc, -1.0);
    const double y0 = logc + double(k);
    const double r2 = r * r;
    double p0 = fma_(A[0], r, A[1]);
    const double p1 = fma_(A[2], r, A[3]);
    const double r4 = r2 * r2;
    double q = fma_(A[4], r, y0);
    q = fma_(p1, r2, q);
    p0 = fma_(p0, r4, q);
    const double ylogx = double(y) * p0;
    if (((as_u64(ylogx) >> 47) & 0xffffu) >= (as_u64(126.0) >> 47)) return false; // |y log2 x| >= 126
    // exp2(v) = 2^(k/32) 2^r, v = k/32 + r
    double kd = ylogx + SHIFT;
    const uint64_t ki = as_u64(kd);
    kd -= SHIFT;
    const double rr = ylogx - kd;
    uint64_t t = E[ki % 32u];
    t += ki << (52 - 5);
    const double s = as_f64(t);
    const double zz = fma_(C[0], rr, C[1]);
    const double rr2 = rr * rr;
    double v = fma_(C[2], rr, 1.0);
    v = fma_(zz, rr2, v);
    v = v * s;
    out = float(v);
    return true;
}

/// glibc expf (e_expf.c) for x <= 88: the main path plus the underflow cut-off.  Shares the exp2 table with powf.
SK_HD bool expf_glibc(const float x, float& out)
{
    constexpr uint64_t E[32] = {
        0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
        0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
        0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
        0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
        0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
        0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
        0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
        0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull };
    constexpr double INVLN2_SCALED = 0x1.71547652b82fep+5; // 32 / ln 2
    constexpr double SHIFT = 0x1.8p+52;
    constexpr double C[3] = { 0x1.c6af84b912394p-20, 0x1.ebfce50fac4f3p-13, 0x1.62e42ff0c52d6p-6 }; // poly / 32^(3,2,1)
    if (!(x <= 88.0f)) return false;      // overflow range, nan
    if (x < -0x1.9fe368p6f) {             // x < log(0x1p-150): underflows to +0 (includes -inf)
        out = 0.f;
        return true;
    }
    // x*32/ln2 = k + r, r in [-1/2, 1/2]
    const double xd = double(x);
    const double z = INVLN2_SCALED * xd;
    double kd = fma_(INVLN2_SCALED, xd, SHIFT);
    const uint64_t ki = as_u64(kd);
    kd -= SHIFT;
    const double r = fma_(INVLN2_SCALED, xd, -kd); // (z - kd with the product fused, as the FMA build does)
    (void)z;
    uint64_t t = E[ki % 32u];
    t += ki << (52 - 5);
    const double s = as_f64(t);
    const double zz = fma_(C[0], r, C[1]);
    const double r2 = r * r;
    double y = fma_(C[2], r, 1.0);
    y = fma_(zz, r2, y);
    y = y * s;
    out = float(y);
    return true;
}

/// glibc log1pf (s_log1pf.c, the fdlibm float routine) for 0 <= x < 0.41422: single-precision arithmetic, no fused ops
/// (there is no FMA build of this one).  Callers must compile with -ffp-contract=off.
SK_HD bool log1pf_glibc(const float x, float& out)
{
    const uint32_t hx = as_u32(x);
    if (hx >= 0x3ed413d7u) return false; // negative, >= 0.41422, inf, nan
    if (hx < 0x31000000u) {              // x < 2^-29
        out = (hx < 0x24800000u) ? x : x - x * x * 0.5f;
        return true;
    }
    const float Lp1 = as_f32(0x3F2AAAABu), Lp2 = as_f32(0x3ECCCCCDu), Lp3 = as_f32(0x3E924925u), Lp4 = as_f32(0x3E638E29u),
                Lp5 = as_f32(0x3E3A3325u), Lp6 = as_f32(0x3E1CD04Fu), Lp7 = as_f32(0x3E178897u);
    const float f = x;
    const float hfsq = 0.5f * f * f;
    const float s = f / (2.0f + f);
    const float z = s * s;
    const float R = z * (Lp1 + z * (Lp2 + z * (Lp3 + z * (Lp4 + z * (Lp5 + z * (Lp6 + z * Lp7))))));
    out = f - (hfsq - s * (hfsq + R));
    return true;
}

} // namespace sk_libm
