// Minimal stand-in for boost/utility.hpp: only boost::noncopyable is used by the reference's hot-path TUs.
#pragma once
namespace boost {
namespace noncopyable_ {
class noncopyable {
protected:
    noncopyable() = default;
    ~noncopyable() = default;
    noncopyable(const noncopyable&) = delete;
    noncopyable& operator=(const noncopyable&) = delete;
};
}
typedef noncopyable_::noncopyable noncopyable;
}
