// Minimal stand-in for the Boost.Math incomplete-gamma entry points the reference uses (gamma_p_inv, gamma_p, gamma_q).
// Used only by code UPSTREAM of the hot path (the indel candidacy count cache, min_count_binom_gte_cache.cpp:72);
// regularized P(a,x) by series / continued fraction (Numerical Recipes 6.2), inverse by bisection to 1e-15 relative.
#pragma once
#include <cmath>
#include <limits>
namespace boost { namespace math {
namespace shim_detail {
inline double gamma_p_series(double a, double x) {
    double sum = 1.0 / a, del = sum, ap = a;
    for (int n = 0; n < 10000; ++n) { ap += 1; del *= x / ap; sum += del; if (std::fabs(del) < std::fabs(sum) * 1e-17) break; }
    return sum * std::exp(-x + a * std::log(x) - std::lgamma(a));
}
inline double gamma_q_cf(double a, double x) {
    const double tiny = 1e-300;
    double b = x + 1 - a, c = 1 / tiny, d = 1 / b, h = d;
    for (int i = 1; i < 10000; ++i) {
        const double an = -i * (i - a);
        b += 2; d = an * d + b; if (std::fabs(d) < tiny) d = tiny; c = b + an / c; if (std::fabs(c) < tiny) c = tiny;
        d = 1 / d; const double del = d * c; h *= del; if (std::fabs(del - 1) < 1e-17) break;
    }
    return std::exp(-x + a * std::log(x) - std::lgamma(a)) * h;
}
}
inline double gamma_p(double a, double x) {
    if (x <= 0) return 0;
    return (x < a + 1) ? shim_detail::gamma_p_series(a, x) : 1 - shim_detail::gamma_q_cf(a, x);
}
inline double gamma_q(double a, double x) { return 1 - gamma_p(a, x); }
inline double gamma_p_inv(double a, double p) {
    if (p <= 0) return 0;
    if (p >= 1) return std::numeric_limits<double>::infinity();
    double lo = 0, hi = a + 10;
    while (gamma_p(a, hi) < p) hi *= 2;
    for (int i = 0; i < 200; ++i) { const double mid = 0.5 * (lo + hi); if (gamma_p(a, mid) < p) lo = mid; else hi = mid; }
    return 0.5 * (lo + hi);
}
inline double gamma_q_inv(double a, double q) { return gamma_p_inv(a, 1 - q); }
}}
