// Minimal stand-in for boost::lexical_cast via iostreams.
#pragma once
#include <limits>
#include "boost/utility.hpp"
#include <sstream>
#include <string>
#include <stdexcept>
namespace boost {
struct bad_lexical_cast : std::bad_cast { const char* what() const noexcept override { return "bad lexical cast"; } };
template <typename Target, typename Source>
inline Target lexical_cast(const Source& s) {
    std::stringstream ss; Target t;
    if (!(ss << s) || !(ss >> t) || !(ss >> std::ws).eof()) throw bad_lexical_cast();
    return t;
}
template <> inline std::string lexical_cast<std::string, std::string>(const std::string& s) { return s; }
}
