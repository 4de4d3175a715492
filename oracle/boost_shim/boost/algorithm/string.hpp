// Minimal stand-in for the three boost::algorithm functions used by the reference (split/is_any_of/starts_with).
#pragma once
#include <string>
#include <vector>
namespace boost {
struct shim_any_of { std::string chars; };
inline shim_any_of is_any_of(const char* c) { return shim_any_of{c}; }
template <typename Seq> inline Seq& split(Seq& out, const std::string& in, const shim_any_of& pred) {
    out.clear(); std::string cur;
    for (char ch : in) { if (pred.chars.find(ch) != std::string::npos) { out.push_back(cur); cur.clear(); } else cur.push_back(ch); }
    out.push_back(cur); return out;
}
inline bool starts_with(const std::string& s, const std::string& p) { return s.compare(0, p.size(), p) == 0; }
}
