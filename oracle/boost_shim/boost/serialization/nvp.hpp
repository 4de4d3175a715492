// Stand-in for boost/serialization/nvp.hpp: a name/value pair the archive stand-ins (boost/archive/xml_*archive.hpp) consume.
#pragma once
#include "boost/serialization/level.hpp"
namespace boost { namespace serialization {
template <typename T> class nvp {
public:
    nvp(const char* n, T& v) : _n(n), _v(&v) {}
    const char* name() const { return _n; }
    T& value() const { return *_v; }
private:
    const char* _n;
    T* _v;
};
template <typename T> inline nvp<T> make_nvp(const char* n, T& t) { return nvp<T>(n, t); }
}}
#define BOOST_SERIALIZATION_NVP(x) boost::serialization::make_nvp(#x, x)
