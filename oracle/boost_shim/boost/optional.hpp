// Minimal stand-in for boost::optional<T> (value semantics, operator bool, *, ->, get, reset).
#pragma once
#include <limits>
#include "boost/utility.hpp"
#include <cassert>
#include <new>
#include <utility>
namespace boost {
struct none_t {};
static const none_t none = none_t();
template <typename T>
class optional {
    bool _init;
    alignas(T) unsigned char _s[sizeof(T)];
    T* p() { return reinterpret_cast<T*>(_s); }
    const T* p() const { return reinterpret_cast<const T*>(_s); }
public:
    optional() : _init(false) {}
    optional(none_t) : _init(false) {}
    optional(const T& v) : _init(true) { new (_s) T(v); }
    optional(const optional& o) : _init(o._init) { if (_init) new (_s) T(*o.p()); }
    ~optional() { reset(); }
    optional& operator=(const optional& o) { if (this != &o) { reset(); if (o._init) { new (_s) T(*o.p()); _init = true; } } return *this; }
    optional& operator=(const T& v) { reset(); new (_s) T(v); _init = true; return *this; }
    void reset() { if (_init) { p()->~T(); _init = false; } }
    void reset(const T& v) { *this = v; }
    explicit operator bool() const { return _init; }
    bool operator!() const { return !_init; }
    bool is_initialized() const { return _init; }
    T& operator*() { assert(_init); return *p(); }
    const T& operator*() const { assert(_init); return *p(); }
    T* operator->() { assert(_init); return p(); }
    const T* operator->() const { assert(_init); return p(); }
    T& get() { assert(_init); return *p(); }
    const T& get() const { assert(_init); return *p(); }
    T* get_ptr() { return _init ? p() : nullptr; }
    const T* get_ptr() const { return _init ? p() : nullptr; }
};
}
