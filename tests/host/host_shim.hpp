// host_shim.hpp -- TEST ONLY: lets g++ compile solorl_amd/csrc/{spatial,dynamics}.hpp for one "lane"
// so the kernel math can be checked against the oracle without a GPU (tests/test_host_harness.py).
#pragma once
#include <cmath>
#include <cstddef>
#include <type_traits>
#define SD inline
#define SNI static __attribute__((noinline))
#define SNI_SCALAR SNI
#define TEAM_SYNC() do {} while (0)
struct float4 { float x, y, z, w; };
struct double2 { double x, y; };
inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }
inline double2 make_double2(double x, double y) { return {x, y}; }
inline int __popc(int x) { return __builtin_popcount((unsigned)x); }
inline bool __any(bool x) { return x; }
inline int __shfl_xor(int v, int) { return v; }
inline int __mul24(int a, int b) { return a * b; }
using std::sqrt; using std::floor; using std::fabs; using std::atan2; using std::asin;
