// boost/chrono.hpp stand-in (test infrastructure): the reference uses only nanoseconds/microseconds/duration_cast
// (L/blt_util/time_util.hh:35-41), which std::chrono provides with identical semantics.
#pragma once
#include <chrono>
namespace boost
{
namespace chrono
{
using std::chrono::duration_cast;
using std::chrono::microseconds;
using std::chrono::nanoseconds;
} // namespace chrono
} // namespace boost
