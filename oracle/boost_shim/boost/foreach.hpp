// Minimal stand-in for BOOST_FOREACH / BOOST_REVERSE_FOREACH via range-for.
#pragma once
namespace boost { namespace shim_detail {
template <typename C> struct rev_range {
    C& c;
    auto begin() -> decltype(c.rbegin()) { return c.rbegin(); }
    auto end() -> decltype(c.rend()) { return c.rend(); }
};
template <typename C> inline rev_range<C> make_rev(C& c) { return rev_range<C>{c}; }
}}
#define BOOST_FOREACH(decl, container) for (decl : container)
#define BOOST_REVERSE_FOREACH(decl, container) for (decl : ::boost::shim_detail::make_rev(container))
