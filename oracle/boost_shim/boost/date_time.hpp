// Stand-in for boost/date_time.hpp: posix_time::second_clock::local_time() + to_simple_string(), used only to
// time-stamp exception context strings (L/common/Exceptions.cpp:41).  TEST INFRASTRUCTURE ONLY.
#pragma once
#include <ctime>
#include <string>
namespace boost { namespace posix_time {
struct ptime { std::time_t t; };
struct second_clock { static ptime local_time() { return ptime{std::time(nullptr)}; } };
inline std::string to_simple_string(const ptime& p) {
    char buf[64];
    std::tm tmv;
    localtime_r(&p.t, &tmv);
    std::strftime(buf, sizeof(buf), "%Y-%b-%d %H:%M:%S", &tmv);
    return buf;
}
}}
