// host_common.hpp -- small shared helpers of the host library.
#pragma once
#include <string>

namespace gwhost
{
inline std::string& last_error()
{
    static thread_local std::string e;
    return e;
}
inline void set_last_error(const std::string& s) { last_error() = s; }
} // namespace gwhost
