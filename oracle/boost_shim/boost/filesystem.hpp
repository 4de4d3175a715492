// Stand-in for boost/filesystem.hpp: exists() and absolute().string(), all the reference's option parsers use.
// TEST INFRASTRUCTURE ONLY.
#pragma once
#include <string>
#include <sys/stat.h>
#include <unistd.h>
#include <limits.h>
namespace boost { namespace filesystem {
class path {
public:
    path() {}
    path(const std::string& s) : _s(s) {}
    path(const char* s) : _s(s) {}
    const std::string& string() const { return _s; }
    const char* c_str() const { return _s.c_str(); }
private:
    std::string _s;
};
inline bool exists(const path& p) { struct stat st; return ::stat(p.c_str(), &st) == 0; }
inline path absolute(const path& p) {
    if (!p.string().empty() && p.string()[0] == '/') return p;
    char buf[PATH_MAX];
    if (!::getcwd(buf, sizeof(buf))) return p;
    return path(std::string(buf) + "/" + p.string());
}
}}
