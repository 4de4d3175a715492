// boost/make_unique.hpp stand-in (test infrastructure)
#pragma once
#include <memory>
#include <utility>
namespace boost
{
template <typename T, typename... Args> std::unique_ptr<T> make_unique(Args&&... args)
{
    return std::unique_ptr<T>(new T(std::forward<Args>(args)...));
}
} // namespace boost
