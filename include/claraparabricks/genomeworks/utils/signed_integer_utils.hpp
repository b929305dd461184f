// signed_integer_utils.hpp -- size helpers that appear in the public cudapoa / cudaaligner signatures.
// API-compatible with the reference's utils/signed_integer_utils.hpp:30-54 (get_size, throw_on_negative).
#pragma once

#include <cassert>
#include <limits>
#include <stdexcept>
#include <type_traits>

namespace claraparabricks
{
namespace genomeworks
{

/// Container size as the signed counterpart of its size_type.
template <class Container>
auto get_size(Container const& c) -> typename std::make_signed<typename Container::size_type>::type
{
    using S = typename std::make_signed<typename Container::size_type>::type;
    assert(c.size() <= static_cast<typename Container::size_type>(std::numeric_limits<S>::max()));
    return static_cast<S>(c.size());
}

/// Container size converted to a caller-chosen integer type.
template <class Integer, class Container>
Integer get_size(Container const& c)
{
    assert(c.size() <= static_cast<typename Container::size_type>(std::numeric_limits<Integer>::max()));
    return static_cast<Integer>(c.size());
}

/// Returns x, or throws std::invalid_argument(message) when x < 0.
template <class T>
T throw_on_negative(T x, const char* message)
{
    static_assert(std::is_arithmetic<T>::value, "throw_on_negative expects an arithmetic type.");
    if (x < T(0)) throw std::invalid_argument(message);
    return x;
}

} // namespace genomeworks
} // namespace claraparabricks
