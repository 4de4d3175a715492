// Minimal stand-in for boost::exception: a tag base class plus diagnostic_information().
#pragma once
#include <exception>
#include <string>
namespace boost {
class exception {
public:
    virtual ~exception() noexcept {}
    const char* throw_function_ = nullptr; const char* throw_file_ = nullptr; int throw_line_ = -1;
protected:
    exception() {}
};
inline std::string diagnostic_information(const exception& e) {
    std::string s;
    if (e.throw_file_) { s += e.throw_file_; s += "("; s += std::to_string(e.throw_line_); s += "): "; }
    if (e.throw_function_) { s += "Throw in function "; s += e.throw_function_; s += "\n"; }
    if (const std::exception* se = dynamic_cast<const std::exception*>(&e)) { s += "std::exception::what: "; s += se->what(); s += "\n"; }
    return s;
}
inline std::string diagnostic_information(const std::exception& e) { return std::string("std::exception::what: ") + e.what() + "\n"; }
}
