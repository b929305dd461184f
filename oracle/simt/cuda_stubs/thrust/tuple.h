// thrust/tuple.h -- TEST INFRASTRUCTURE (oracle/simt): thrust's tuple is std::tuple here
#pragma once
#include <tuple>
#include <utility>
namespace thrust
{
using std::get;
using std::make_tuple;
using std::swap;
using std::tie;
using std::tuple;
} // namespace thrust
