#pragma once
#include "boost/exception/all.hpp"
#include <type_traits>
namespace boost { namespace shim_detail {
template <typename E> inline void set_loc(E& e, const char* fn, const char* file, int line, std::true_type) { e.throw_function_ = fn; e.throw_file_ = file; e.throw_line_ = line; }
template <typename E> inline void set_loc(E&, const char*, const char*, int, std::false_type) {}
template <typename E> [[noreturn]] inline void throw_with_loc(E e, const char* fn, const char* file, int line) {
    set_loc(e, fn, file, line, std::is_base_of<boost::exception, E>()); throw e; }
}
template <typename E> [[noreturn]] inline void throw_exception(const E& e) { throw e; }
}
#define BOOST_THROW_EXCEPTION(x) ::boost::shim_detail::throw_with_loc(x, __func__, __FILE__, __LINE__)
