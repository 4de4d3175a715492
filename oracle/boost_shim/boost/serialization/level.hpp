// Minimal stand-in: serialization traits are declared but never exercised by the oracle.
#pragma once
#include <limits>
#include "boost/utility.hpp"
#define BOOST_CLASS_IMPLEMENTATION(T, L)
namespace boost { namespace serialization {
enum level_type { not_serializable = 0, primitive_type = 1, object_serializable = 2, object_class_info = 3 };
}}
