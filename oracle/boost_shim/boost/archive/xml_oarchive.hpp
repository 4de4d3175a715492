// Stand-in for boost/archive/xml_oarchive.hpp: writes name/value pairs of the run-stats structs as nested XML
// elements (L/appstats/RunStats.cpp:61-66).  TEST INFRASTRUCTURE ONLY; the layout is not Boost's archive format.
#pragma once
#include "boost/serialization/nvp.hpp"
#include <ostream>
#include <type_traits>
namespace boost { namespace archive {
class xml_oarchive {
public:
    explicit xml_oarchive(std::ostream& os) : _os(os) { _os << "<?xml version=\"1.0\" encoding=\"UTF-8\" standalone=\"yes\" ?>\n<boost_serialization_shim>\n"; }
    ~xml_oarchive() { _os << "</boost_serialization_shim>\n"; }
    typedef std::true_type is_saving;
    template <typename T> xml_oarchive& operator<<(const boost::serialization::nvp<T>& p) { save(p.name(), p.value()); return *this; }
    template <typename T> xml_oarchive& operator&(const boost::serialization::nvp<T>& p) { return (*this) << p; }
private:
    template <typename T> typename std::enable_if<std::is_arithmetic<T>::value>::type save(const char* n, const T& v) {
        _os << std::string(_depth, '\t') << "<" << n << ">" << v << "</" << n << ">\n";
    }
    template <typename T> typename std::enable_if<!std::is_arithmetic<T>::value>::type save(const char* n, const T& v) {
        _os << std::string(_depth, '\t') << "<" << n << ">\n";
        ++_depth;
        const_cast<typename std::remove_const<T>::type&>(v).serialize(*this, 0u);
        --_depth;
        _os << std::string(_depth, '\t') << "</" << n << ">\n";
    }
    std::ostream& _os;
    unsigned _depth = 1;
};
}}
