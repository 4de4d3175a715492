// Minimal stand-in for the single Boost.Spirit use in the reference (blt_util/parse_util.cpp:222):
// qi::parse(first, last, double_, val) -> strtod.  Option-string parsing only; off the hot path.
#pragma once
#include <cstdlib>
#include <cstring>
#include <string>
namespace boost { namespace spirit {
struct double_type {};
static const double_type double_ = double_type();
namespace qi {
inline bool parse(const char*& first, const char* last, const double_type&, double& val) {
    const std::string tmp(first, last);
    char* end = nullptr;
    const double v = std::strtod(tmp.c_str(), &end);
    if (end == tmp.c_str()) return false;
    val = v;
    first += (end - tmp.c_str());
    return true;
}
}}}
