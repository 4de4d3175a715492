#pragma once
#include "boost/math/distributions/binomial.hpp"
