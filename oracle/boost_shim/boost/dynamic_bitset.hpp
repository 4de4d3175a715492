// Minimal stand-in for boost::dynamic_bitset<> covering the operations used by blt_util/RangeMap.hh.
// Bit i is element i; operator<< moves bits towards higher indices (same convention as Boost).
#pragma once
#include <limits>
#include "boost/utility.hpp"
#include <vector>
#include <cassert>
#include <cstddef>
namespace boost {
template <typename Block = unsigned long>
class dynamic_bitset {
    std::vector<bool> _b;
public:
    typedef std::size_t size_type;
    dynamic_bitset() {}
    explicit dynamic_bitset(size_type n, unsigned long /*value*/ = 0) : _b(n, false) {}
    size_type size() const { return _b.size(); }
    void resize(size_type n, bool v = false) { _b.resize(n, v); }
    dynamic_bitset& reset() { _b.assign(_b.size(), false); return *this; }
    dynamic_bitset& reset(size_type i) { _b[i] = false; return *this; }
    dynamic_bitset& set() { _b.assign(_b.size(), true); return *this; }
    dynamic_bitset& set(size_type i, bool v = true) { _b[i] = v; return *this; }
    bool test(size_type i) const { return _b[i]; }
    bool operator[](size_type i) const { return _b[i]; }
    bool any() const { for (bool x : _b) if (x) return true; return false; }
    bool none() const { return !any(); }
    size_type count() const { size_type c = 0; for (bool x : _b) c += x; return c; }
    dynamic_bitset& operator&=(const dynamic_bitset& o) { for (size_type i = 0; i < _b.size(); ++i) _b[i] = _b[i] && o._b[i]; return *this; }
    dynamic_bitset& operator|=(const dynamic_bitset& o) { for (size_type i = 0; i < _b.size(); ++i) _b[i] = _b[i] || o._b[i]; return *this; }
    dynamic_bitset& operator<<=(size_type n) {
        const size_type s = _b.size();
        if (n >= s) { reset(); return *this; }
        for (size_type i = s; i-- > n;) _b[i] = _b[i - n];
        for (size_type i = 0; i < n; ++i) _b[i] = false;
        return *this;
    }
    dynamic_bitset& operator>>=(size_type n) {
        const size_type s = _b.size();
        if (n >= s) { reset(); return *this; }
        for (size_type i = 0; i + n < s; ++i) _b[i] = _b[i + n];
        for (size_type i = s - n; i < s; ++i) _b[i] = false;
        return *this;
    }
    dynamic_bitset operator<<(size_type n) const { dynamic_bitset r(*this); r <<= n; return r; }
    dynamic_bitset operator>>(size_type n) const { dynamic_bitset r(*this); r >>= n; return r; }
    dynamic_bitset operator~() const { dynamic_bitset r(*this); r._b.flip(); return r; }
};
}
