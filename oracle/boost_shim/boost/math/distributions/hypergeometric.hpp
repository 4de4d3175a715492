// Stand-in for boost::math::hypergeometric_distribution: pdf via lgamma.  Used only by blt_util/fisher_exact_test.cpp
// (somatic indel EVS strand-bias feature), downstream of the hot path.  TEST INFRASTRUCTURE ONLY.
#pragma once
#include <cmath>
namespace boost { namespace math {
template <typename T = double> class hypergeometric_distribution {
public:
    hypergeometric_distribution(unsigned r, unsigned n, unsigned N) : _r(r), _n(n), _N(N) {}
    unsigned defective() const { return _r; }
    unsigned sample_count() const { return _n; }
    unsigned total() const { return _N; }
private:
    unsigned _r, _n, _N;
};
typedef hypergeometric_distribution<double> hypergeometric;
namespace shim_detail {
inline double lchoose(double n, double k) { return std::lgamma(n + 1) - std::lgamma(k + 1) - std::lgamma(n - k + 1); }
}
template <typename T> inline double pdf(const hypergeometric_distribution<T>& d, unsigned k) {
    const double r(d.defective()), n(d.sample_count()), N(d.total());
    if (k > r || k > n || (n - k) > (N - r)) return 0;
    return std::exp(shim_detail::lchoose(r, k) + shim_detail::lchoose(N - r, n - k) - shim_detail::lchoose(N, n));
}
}}
