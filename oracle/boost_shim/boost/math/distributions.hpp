// Stand-in for boost/math/distributions.hpp: the distributions the reference's gVCF record code touches.
#pragma once
#include "boost/math/distributions/binomial.hpp"
#include "boost/math/distributions/chi_squared.hpp"
#include "boost/math/distributions/complement.hpp"
