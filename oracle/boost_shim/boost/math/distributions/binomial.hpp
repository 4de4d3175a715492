// Minimal stand-in for boost::math::binomial_distribution with cdf / complement / pdf (exact log-space summation).
// Used only by blt_util/binomial_test.cpp, upstream of the hot path.
#pragma once
#include <cmath>
#include <cassert>
#include <cfloat>
namespace boost { namespace math {
template <typename T = double> class binomial_distribution {
public:
    binomial_distribution(T n, T p) : _n(n), _p(p) {}
    T trials() const { return _n; }
    T success_fraction() const { return _p; }
private:
    T _n, _p;
};
typedef binomial_distribution<double> binomial;
template <typename D> struct shim_complement { const D& dist; double x; };
template <typename D> inline shim_complement<D> complement(const D& d, double x) { return shim_complement<D>{d, x}; }
template <typename T> inline double pdf(const binomial_distribution<T>& d, double k) {
    const double n = d.trials(), p = d.success_fraction();
    if (k < 0 || k > n) return 0;
    if (p <= 0) return k == 0 ? 1 : 0;
    if (p >= 1) return k == n ? 1 : 0;
    return std::exp(std::lgamma(n + 1) - std::lgamma(k + 1) - std::lgamma(n - k + 1) + k * std::log(p) + (n - k) * std::log1p(-p));
}
template <typename T> inline double cdf(const binomial_distribution<T>& d, double k) {
    double s = 0; const double kk = std::floor(k);
    for (double i = 0; i <= kk; ++i) s += pdf(d, i);
    return s > 1 ? 1 : s;
}
template <typename T> inline double cdf(const shim_complement<binomial_distribution<T>>& c) {
    double s = 0; const double n = c.dist.trials();
    for (double i = std::floor(c.x) + 1; i <= n; ++i) s += pdf(c.dist, i);
    return s > 1 ? 1 : s;
}
// quantile(complement(binomial(n,p), alpha)): smallest k with P(X > k) <= alpha  (Boost's default discrete-quantile
// policy rounds the complement quantile outward, i.e. up)
template <typename T> inline double quantile(const shim_complement<binomial_distribution<T>>& c) {
    const double n = c.dist.trials();
    double tail = 1.0; // P(X > k) for k = -1
    for (double k = 0; k <= n; ++k) {
        tail -= pdf(c.dist, k);
        if (tail <= c.x) return k;
    }
    return n;
}
template <typename T> inline double quantile(const binomial_distribution<T>& d, double p) {
    double s = 0; const double n = d.trials();
    for (double k = 0; k <= n; ++k) { s += pdf(d, k); if (s >= p) return k; }
    return n;
}
}}
