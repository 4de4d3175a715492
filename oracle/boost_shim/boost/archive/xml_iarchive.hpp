// Stand-in for boost/archive/xml_iarchive.hpp: reads back what the xml_oarchive stand-in wrote (elements in
// serialisation order).  TEST INFRASTRUCTURE ONLY.
#pragma once
#include "boost/serialization/nvp.hpp"
#include <istream>
#include <sstream>
#include <string>
#include <type_traits>
namespace boost { namespace archive {
class xml_iarchive {
public:
    explicit xml_iarchive(std::istream& is) : _is(is) { std::string line; std::getline(_is, line); std::getline(_is, line); }
    typedef std::false_type is_saving;
    template <typename T> xml_iarchive& operator>>(const boost::serialization::nvp<T>& p) { load(p.value()); return *this; }
    template <typename T> xml_iarchive& operator&(const boost::serialization::nvp<T>& p) { return (*this) >> p; }
private:
    template <typename T> typename std::enable_if<std::is_arithmetic<T>::value>::type load(T& v) {
        std::string line;
        std::getline(_is, line);
        const auto a(line.find('>')), b(line.rfind("</"));
        std::istringstream iss(line.substr(a + 1, b - a - 1));
        iss >> v;
    }
    template <typename T> typename std::enable_if<!std::is_arithmetic<T>::value>::type load(T& v) {
        std::string line;
        std::getline(_is, line);
        v.serialize(*this, 0u);
        std::getline(_is, line);
    }
    std::istream& _is;
};
}}
