// Minimal stand-in for boost/program_options.hpp: only the names referenced by uninstantiated
// templates in the reference's hot-path include closure (blt_util/PrettyFloat.hh) are declared.
#pragma once
#include <limits>
#include "boost/utility.hpp"
#include <string>
#include <vector>
#include <stdexcept>
#include <typeinfo>
#include <memory>
namespace boost {
class any {
    struct holder_base { virtual ~holder_base() {} virtual holder_base* clone() const = 0; virtual const std::type_info& type() const = 0; };
    template <typename T> struct holder : holder_base {
        T held; explicit holder(const T& t) : held(t) {}
        holder_base* clone() const override { return new holder(held); }
        const std::type_info& type() const override { return typeid(T); }
    };
    std::unique_ptr<holder_base> _p;
public:
    any() {}
    template <typename T> any(const T& t) : _p(new holder<T>(t)) {}
    any(const any& o) : _p(o._p ? o._p->clone() : nullptr) {}
    any& operator=(const any& o) { _p.reset(o._p ? o._p->clone() : nullptr); return *this; }
    bool empty() const { return !_p; }
    template <typename T> friend T* any_cast(any* a);
};
template <typename T> T* any_cast(any* a) {
    if (!a || !a->_p || a->_p->type() != typeid(T)) return nullptr;
    return &static_cast<any::holder<T>*>(a->_p.get())->held;
}
namespace program_options {
struct validation_error : std::logic_error {
    enum kind_t { multiple_values_not_allowed = 30, at_least_one_value_required, invalid_bool_value, invalid_option_value, invalid_option };
    explicit validation_error(kind_t) : std::logic_error("validation_error") {}
};
}}
