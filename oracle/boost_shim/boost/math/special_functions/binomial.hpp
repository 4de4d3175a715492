// Minimal stand-in for boost::math::binomial_coefficient<T>(n,k): exact integer value (n small), converted to T.
// Boost 1.58 computes factorial-table quotients and rounds to the nearest integer (ceil(x-0.5)); for the n<=34
// range the reference uses (starling_align_limit.cpp:48,70) that is the exact coefficient whenever it is representable.
#pragma once
#include <limits>
#include <cassert>
namespace boost { namespace math {
template <typename T> inline T binomial_coefficient(unsigned n, unsigned k) {
    if (k > n) return T(0);
    if (k > n - k) k = n - k;
    unsigned long long r = 1; // exact: each partial product is itself a binomial coefficient
    for (unsigned i = 1; i <= k; ++i) r = r * (unsigned long long)(n - k + i) / (unsigned long long)i;
    return static_cast<T>(r);
}
}}
