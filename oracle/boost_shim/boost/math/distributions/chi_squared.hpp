// Minimal stand-in for boost::math::chi_squared (cdf / complement / quantile) via the incomplete gamma shim.
#pragma once
#include "boost/math/special_functions/gamma.hpp"
#include "boost/math/distributions/binomial.hpp"
namespace boost { namespace math {
class chi_squared {
public:
    explicit chi_squared(double df) : _df(df) {}
    double degrees_of_freedom() const { return _df; }
private:
    double _df;
};
inline double cdf(const chi_squared& d, double x) { return gamma_p(d.degrees_of_freedom() / 2, x / 2); }
inline double cdf(const shim_complement<chi_squared>& c) { return gamma_q(c.dist.degrees_of_freedom() / 2, c.x / 2); }
inline double quantile(const chi_squared& d, double p) { return 2 * gamma_p_inv(d.degrees_of_freedom() / 2, p); }
inline double quantile(const shim_complement<chi_squared>& c) { return 2 * gamma_p_inv(c.dist.degrees_of_freedom() / 2, 1 - c.x); }
}}
