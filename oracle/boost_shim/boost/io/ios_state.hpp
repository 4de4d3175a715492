// Minimal stand-in for boost::io::ios_all_saver (saves/restores flags, precision, width, fill).
#pragma once
#include <ios>
namespace boost { namespace io {
class ios_all_saver {
    std::ios& _s; std::ios::fmtflags _f; std::streamsize _p, _w; char _c;
public:
    explicit ios_all_saver(std::ios& s) : _s(s), _f(s.flags()), _p(s.precision()), _w(s.width()), _c(s.fill()) {}
    ~ios_all_saver() { restore(); }
    void restore() { _s.flags(_f); _s.precision(_p); _s.width(_w); _s.fill(_c); }
};
typedef ios_all_saver ios_flags_saver;
}}
