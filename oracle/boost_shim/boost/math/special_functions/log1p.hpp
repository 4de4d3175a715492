// Minimal stand-in for Boost.Math log1p (test infrastructure only; see oracle/README.md).
// Boost 1.58 forwards boost::math::log1p(double) to C99 ::log1p on glibc (BOOST_HAS_LOG1P).
#pragma once
#include <limits>
#include "boost/utility.hpp"
#include <cmath>
namespace boost { namespace math {
inline double log1p(double x) { return ::log1p(x); }
inline float log1p(float x) { return ::log1pf(x); }
inline long double log1p(long double x) { return ::log1pl(x); }
}}
