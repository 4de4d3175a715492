#pragma once
#include "boost/serialization/level.hpp"
#define BOOST_SERIALIZATION_NVP(x) x
namespace boost { namespace serialization {
template <typename T> inline T& make_nvp(const char*, T& t) { return t; }
}}
