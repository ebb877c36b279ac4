// spatial.hpp -- register-resident 3-D / 6-D spatial algebra for the gfx950 rollout kernels.
//
// One environment per lane: every object below lives in VGPRs of ONE lane, all indices are
// compile-time so nothing is demoted to scratch (cdna_hip_programming.md 5.4 rule 20).
// All spatial quantities of one env are expressed in ONE frame -- world-aligned axes, origin at
// the base-link origin -- so the articulated-body recursion needs no frame transforms: child
// inertias and bias forces are simply added into the parent's.
#pragma once
#ifdef SOLO_HOST_SHIM          // tests/host_harness.cpp: same source compiled by g++ for CPU debugging
#include "host_shim.hpp"
#else
#include <hip/hip_runtime.h>
#define SD __device__ __forceinline__
#define SNI __device__ __noinline__
// serial (one-lane) phases: the packed-FP32 forms need aligned register pairs, and assembling those pairs cost a
// third of the leg phase in v_mov (769 of 2 600 instructions); scalar FMAs are shorter there
#define SNI_SCALAR __device__ __noinline__ __attribute__((target("no-packed-fp32-ops")))
// Hand-off between lanes of a team through LDS INSIDE one function (one lane stores, others load): without it
// the compiler may forward the stored value on the storing path and sink the others' load into the opposite
// branch, which the SIMT lowering can run first (found with the NaN-poisoned-LDS build, tools/dev/zero_region.sh).
// Hand-offs across the non-inlined phase calls need nothing: a call orders all memory operations.
#define TEAM_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
#endif

namespace solo {

// Reciprocal / square root of the hot phases.  fp32 on the GPU: the hardware approximations (v_rcp_f32, v_sqrt_f32, v_rsq_f32: 1 ulp),
// the reciprocal with one Newton step -- an IEEE-rounded fp32 division is a 10-instruction sequence (v_div_scale x2, v_rcp, 4 FMAs,
// v_div_fmas, v_div_fixup), a rounded square root 8, and the serial phases run ~20 of them per sub-step.  fp64 (the validation
// precision) and the host build keep the exact operations.
template <typename T> SD T rcp_fast(T x) {
#ifndef SOLO_HOST_SHIM
  if constexpr (sizeof(T) == 4) { const float r = __builtin_amdgcn_rcpf(x); return __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r); }
  else
#endif
  return T(1) / x;
}
template <typename T> SD T sqrt_fast(T x) {
#ifndef SOLO_HOST_SHIM
  if constexpr (sizeof(T) == 4) return __builtin_amdgcn_sqrtf(x);
  else
#endif
  return sqrt(x);
}
template <typename T> SD T rsqrt_fast(T x) {
#ifndef SOLO_HOST_SHIM
  if constexpr (sizeof(T) == 4) { const float r = __builtin_amdgcn_rsqf(x); return r * __builtin_fmaf(__builtin_fmaf(-x * r, r, 1.0f), 0.5f, 1.0f); }
  else
#endif
  return T(1) / sqrt(x);
}

template <typename T> struct V3 { T x, y, z; };

template <typename T> SD V3<T> mk(T x, T y, T z) { return V3<T>{x, y, z}; }
template <typename T> SD V3<T> operator+(V3<T> a, V3<T> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <typename T> SD V3<T> operator-(V3<T> a, V3<T> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <typename T> SD V3<T> operator-(V3<T> a) { return {-a.x, -a.y, -a.z}; }
template <typename T> SD V3<T> operator*(V3<T> a, T s) { return {a.x * s, a.y * s, a.z * s}; }
template <typename T> SD V3<T> operator*(T s, V3<T> a) { return {a.x * s, a.y * s, a.z * s}; }
template <typename T> SD T dot(V3<T> a, V3<T> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <typename T> SD V3<T> cross(V3<T> a, V3<T> b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
template <typename T> SD V3<T> fma3(V3<T> a, T s, V3<T> acc) {  // acc + a*s
  return {acc.x + a.x * s, acc.y + a.y * s, acc.z + a.z * s};
}

// rotation matrix stored by columns (c0,c1,c2 = images of the x,y,z axes)
template <typename T> struct M3 { V3<T> c0, c1, c2; };
template <typename T> SD V3<T> mul(const M3<T>& R, V3<T> v) { return R.c0 * v.x + R.c1 * v.y + R.c2 * v.z; }

template <typename T> SD M3<T> quat_to_mat(T x, T y, T z, T w) {
  M3<T> R;
  R.c0 = {T(1) - T(2) * (y * y + z * z), T(2) * (x * y + z * w), T(2) * (x * z - y * w)};
  R.c1 = {T(2) * (x * y - z * w), T(1) - T(2) * (x * x + z * z), T(2) * (y * z + x * w)};
  R.c2 = {T(2) * (x * z + y * w), T(2) * (y * z - x * w), T(1) - T(2) * (x * x + y * y)};
  return R;
}
// R * Rot(axis, q) for axis = x (AX=0) or y (AX=1): only two columns change
template <int AX, typename T> SD M3<T> rot_axis(const M3<T>& R, T c, T s) {
  M3<T> o;
  if constexpr (AX == 0) { o.c0 = R.c0; o.c1 = R.c1 * c + R.c2 * s; o.c2 = R.c2 * c - R.c1 * s; }
  else                   { o.c1 = R.c1; o.c0 = R.c0 * c - R.c2 * s; o.c2 = R.c2 * c + R.c0 * s; }
  return o;
}

// spatial (Pluecker) vector: a = angular part (motion: omega, force: moment about the origin),
// l = linear part (motion: velocity of the body point at the origin, force: force)
template <typename T> struct SV { V3<T> a, l; };
template <typename T> SD SV<T> operator+(SV<T> p, SV<T> q) { return {p.a + q.a, p.l + q.l}; }
template <typename T> SD SV<T> operator-(SV<T> p, SV<T> q) { return {p.a - q.a, p.l - q.l}; }
template <typename T> SD SV<T> operator*(SV<T> p, T s) { return {p.a * s, p.l * s}; }
template <typename T> SD T dot(SV<T> p, SV<T> q) { return dot(p.a, q.a) + dot(p.l, q.l); }
template <typename T> SD SV<T> fma6(SV<T> p, T s, SV<T> acc) { return {fma3(p.a, s, acc.a), fma3(p.l, s, acc.l)}; }
template <typename T> SD SV<T> zero6() { return {{T(0), T(0), T(0)}, {T(0), T(0), T(0)}}; }

template <typename T> struct Sym3 { T xx, xy, xz, yy, yz, zz; };
template <typename T> SD V3<T> mul(const Sym3<T>& S, V3<T> v) {
  return {S.xx * v.x + S.xy * v.y + S.xz * v.z, S.xy * v.x + S.yy * v.y + S.yz * v.z,
          S.xz * v.x + S.yz * v.y + S.zz * v.z};
}
template <typename T> SD void add(Sym3<T>& a, const Sym3<T>& b) {
  a.xx += b.xx; a.xy += b.xy; a.xz += b.xz; a.yy += b.yy; a.yz += b.yz; a.zz += b.zz;
}
template <typename T> SD void rank1_sub(Sym3<T>& a, V3<T> u, V3<T> us) {  // a -= u * us^T (us = u*s)
  a.xx -= u.x * us.x; a.xy -= u.x * us.y; a.xz -= u.x * us.z;
  a.yy -= u.y * us.y; a.yz -= u.y * us.z; a.zz -= u.z * us.z;
}

// 6x6 symmetric (articulated-body) inertia  [[A, B], [B^T, C]]; B rows = angular index
template <typename T> struct ABI { Sym3<T> A; V3<T> B0, B1, B2; Sym3<T> C; };

template <typename T> SD SV<T> mul(const ABI<T>& I, SV<T> s) {
  SV<T> r;
  r.a = mul(I.A, s.a) + mk(dot(I.B0, s.l), dot(I.B1, s.l), dot(I.B2, s.l));
  r.l = I.B0 * s.a.x + I.B1 * s.a.y + I.B2 * s.a.z + mul(I.C, s.l);
  return r;
}
template <typename T> SD void add(ABI<T>& a, const ABI<T>& b) {
  add(a.A, b.A); a.B0 = a.B0 + b.B0; a.B1 = a.B1 + b.B1; a.B2 = a.B2 + b.B2; add(a.C, b.C);
}
// I -= U (U*s)^T
template <typename T> SD void rank1_sub(ABI<T>& I, SV<T> U, T s) {
  SV<T> Us = U * s;
  rank1_sub(I.A, U.a, Us.a);
  I.B0 = I.B0 - Us.l * U.a.x; I.B1 = I.B1 - Us.l * U.a.y; I.B2 = I.B2 - Us.l * U.a.z;
  rank1_sub(I.C, U.l, Us.l);
}

// rigid-body inertia about the common origin: mass m, first moment h = m*c, rotational inertia
// Ibar about the origin.  As a 6x6: A = Ibar, B = skew(h), C = m*1.
template <typename T> struct RBI { T m; V3<T> h; Sym3<T> I; };
template <typename T> SD SV<T> mul(const RBI<T>& I, SV<T> s) {
  return {mul(I.I, s.a) + cross(I.h, s.l), s.l * I.m - cross(I.h, s.a)};
}
template <typename T> SD void add(RBI<T>& a, const RBI<T>& b) { a.m += b.m; a.h = a.h + b.h; add(a.I, b.I); }
template <typename T> SD ABI<T> to_abi(const RBI<T>& r) {
  ABI<T> o;
  o.A = r.I;
  o.B0 = {T(0), -r.h.z, r.h.y}; o.B1 = {r.h.z, T(0), -r.h.x}; o.B2 = {-r.h.y, r.h.x, T(0)};
  o.C = {r.m, T(0), T(0), r.m, T(0), r.m};
  return o;
}
template <typename T> SD void add(ABI<T>& a, const RBI<T>& r) {
  add(a.A, r.I);
  a.B0.y -= r.h.z; a.B0.z += r.h.y; a.B1.x += r.h.z; a.B1.z -= r.h.x; a.B2.x -= r.h.y; a.B2.y += r.h.x;
  a.C.xx += r.m; a.C.yy += r.m; a.C.zz += r.m;
}

// spatial motion cross product  v x m
template <typename T> SD SV<T> crm(SV<T> v, SV<T> m) { return {cross(v.a, m.a), cross(v.a, m.l) + cross(v.l, m.a)}; }

// dense symmetric 6x6 with compile-time indexing (base inverse inertia)
template <typename T> struct Sym6 { T m[6][6]; };
template <typename T> SD SV<T> mul(const Sym6<T>& L, SV<T> f) {
  T v[6] = {f.a.x, f.a.y, f.a.z, f.l.x, f.l.y, f.l.z}, r[6];
#pragma unroll
  for (int i = 0; i < 6; i++) {
    T acc = T(0);
#pragma unroll
    for (int j = 0; j < 6; j++) acc += L.m[i][j] * v[j];
    r[i] = acc;
  }
  return {{r[0], r[1], r[2]}, {r[3], r[4], r[5]}};
}

// inverse of a symmetric positive definite 6x6 given as ABI (Cholesky, then L^-T L^-1)
template <typename T> SD Sym6<T> spd_inverse(const ABI<T>& I) {
  T a[6][6];
  a[0][0] = I.A.xx; a[1][0] = I.A.xy; a[2][0] = I.A.xz; a[1][1] = I.A.yy; a[2][1] = I.A.yz; a[2][2] = I.A.zz;
  a[3][0] = I.B0.x; a[4][0] = I.B0.y; a[5][0] = I.B0.z;
  a[3][1] = I.B1.x; a[4][1] = I.B1.y; a[5][1] = I.B1.z;
  a[3][2] = I.B2.x; a[4][2] = I.B2.y; a[5][2] = I.B2.z;
  a[3][3] = I.C.xx; a[4][3] = I.C.xy; a[5][3] = I.C.xz; a[4][4] = I.C.yy; a[5][4] = I.C.yz; a[5][5] = I.C.zz;
  T Lm[6][6], dinv[6];
#pragma unroll
  for (int j = 0; j < 6; j++) {
    T d = a[j][j];
#pragma unroll
    for (int k = 0; k < j; k++) d -= Lm[j][k] * Lm[j][k];
    T rs = T(1) / sqrt(d);
    dinv[j] = rs; Lm[j][j] = d * rs;
#pragma unroll
    for (int i = j + 1; i < 6; i++) {
      T v = a[i][j];
#pragma unroll
      for (int k = 0; k < j; k++) v -= Lm[i][k] * Lm[j][k];
      Lm[i][j] = v * rs;
    }
  }
  // X = L^-1 (lower triangular)
  T X[6][6];
#pragma unroll
  for (int j = 0; j < 6; j++) {
    X[j][j] = dinv[j];
#pragma unroll
    for (int i = j + 1; i < 6; i++) {
      T v = T(0);
#pragma unroll
      for (int k = j; k < i; k++) v -= Lm[i][k] * X[k][j];
      X[i][j] = v * dinv[i];
    }
  }
  Sym6<T> R;
#pragma unroll
  for (int i = 0; i < 6; i++)
#pragma unroll
    for (int j = 0; j <= i; j++) {
      T v = T(0);
#pragma unroll
      for (int k = i; k < 6; k++) v += X[k][i] * X[k][j];
      R.m[i][j] = v; R.m[j][i] = v;
    }
  return R;
}

// inverse of a symmetric positive definite 3x3 by cofactors (25 flops, dependency depth ~5)
template <typename T> SD Sym3<T> inverse(const Sym3<T>& S) {
  const T c00 = S.yy * S.zz - S.yz * S.yz, c01 = S.xz * S.yz - S.xy * S.zz, c02 = S.xy * S.yz - S.xz * S.yy;
  const T c11 = S.xx * S.zz - S.xz * S.xz, c12 = S.xy * S.xz - S.xx * S.yz, c22 = S.xx * S.yy - S.xy * S.xy;
  const T r = rcp_fast(S.xx * c00 + S.xy * c01 + S.xz * c02);
  return {c00 * r, c01 * r, c02 * r, c11 * r, c12 * r, c22 * r};
}

// Same result as spd_inverse by 3x3 blocks (Schur complement of the linear block C): ~170 flops with short
// dependency chains instead of the ~450 mostly dependent ones of the Cholesky route -- the base solve runs on
// one lane, where a dependent instruction issues only every ~8 cycles.
//   [[A, B], [B^T, C]]^-1 = [[S^-1, F], [F^T, C^-1 - E^T F]],   E = B C^-1,  S = A - E B^T,  F = -S^-1 E
template <typename T> SD Sym6<T> spd_inverse_block(const ABI<T>& I) {
  const Sym3<T> Ci = inverse(I.C);
  const V3<T> E0 = mul(Ci, I.B0), E1 = mul(Ci, I.B1), E2 = mul(Ci, I.B2);     // rows of E (C^-1 symmetric)
  Sym3<T> S = I.A;
  S.xx -= dot(E0, I.B0); S.xy -= dot(E0, I.B1); S.xz -= dot(E0, I.B2);
  S.yy -= dot(E1, I.B1); S.yz -= dot(E1, I.B2); S.zz -= dot(E2, I.B2);
  const Sym3<T> Si = inverse(S);
  // F = -Si E: row i of F = -(Si_i0 E0 + Si_i1 E1 + Si_i2 E2)
  const V3<T> F0 = -(E0 * Si.xx + E1 * Si.xy + E2 * Si.xz);
  const V3<T> F1 = -(E0 * Si.xy + E1 * Si.yy + E2 * Si.yz);
  const V3<T> F2 = -(E0 * Si.xz + E1 * Si.yz + E2 * Si.zz);
  Sym6<T> R;
  const T si[3][3] = {{Si.xx, Si.xy, Si.xz}, {Si.xy, Si.yy, Si.yz}, {Si.xz, Si.yz, Si.zz}};
  const T ci[3][3] = {{Ci.xx, Ci.xy, Ci.xz}, {Ci.xy, Ci.yy, Ci.yz}, {Ci.xz, Ci.yz, Ci.zz}};
  const T f[3][3] = {{F0.x, F0.y, F0.z}, {F1.x, F1.y, F1.z}, {F2.x, F2.y, F2.z}};
  const T e[3][3] = {{E0.x, E0.y, E0.z}, {E1.x, E1.y, E1.z}, {E2.x, E2.y, E2.z}};
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      R.m[i][j] = si[i][j];
      R.m[i][3 + j] = f[i][j]; R.m[3 + j][i] = f[i][j];
    }
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = i; j < 3; j++) {
      const T v = ci[i][j] - (e[0][i] * f[0][j] + e[1][i] * f[1][j] + e[2][i] * f[2][j]);
      R.m[3 + i][3 + j] = v; R.m[3 + j][3 + i] = v;
    }
  return R;
}

}  // namespace solo
