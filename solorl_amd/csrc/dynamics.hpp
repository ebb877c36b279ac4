// dynamics.hpp -- one physics sub-step of one Solo8/Solo12 env, one env per lane (gfx950).
//
// Restates, for the GPU, what the reference obtains from `p.stepSimulation()` (reference
// solo.py:264-265; Bullet btMultiBody pipeline):
//   collision (analytic primitives vs plane z=0) -> articulated-body forward dynamics with
//   Bullet's damping -> contact / joint-limit rows -> 50 PGS iterations (warm-started) ->
//   velocity clamp -> semi-implicit Euler.
// Formulation (different from oracle/solo_oracle.c on purpose):
//   * all spatial quantities in one world-aligned frame at the base origin (no X transforms);
//   * star topology exploited: a constraint row on leg L only touches the 6 base DoF and the
//     3 (2) joints of that leg:  J.dV = Jh.w + JL.yL  with  Jh = force transmitted to the base,
//     w = base delta-velocity, yL = leg delta-rates with the base held fixed;
//   * rows live in LDS ([row][16-byte chunk][lane], conflict-free b128 accesses), only the base
//     accumulator w stays in registers during the PGS sweep.
#pragma once
#include <utility>

#include "../../include/solorl_model_data.h"
#include "spatial.hpp"

namespace solo {

constexpr int MAX_CONTACTS = 8;
constexpr int MAX_LIMITS = 2;
constexpr int MAX_ROWS = MAX_LIMITS + 3 * MAX_CONTACTS;   // 26
constexpr int ROW_CORE = 20;                              // Jh6 JL3 W6 Y3 rhs dinv
constexpr int NPRIM = 20;
constexpr double LIMIT_WINDOW = 0.5;
constexpr double DISC_EPS2 = 1e-12;

template <int N, typename F, int... I>
SD void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F> SD void static_for(F&& f) { static_for_impl<N>(f, std::make_integer_sequence<int, N>{}); }

template <int ROBOT> struct Robot;
template <> struct Robot<0> {
  static constexpr int NJ = 2, NQ = 8, NL = 13;
  static constexpr const solorl_model_data& MD = SOLORL_MODEL_SOLO8;
};
template <> struct Robot<1> {
  static constexpr int NJ = 3, NQ = 12, NL = 17;
  static constexpr const solorl_model_data& MD = SOLORL_MODEL_SOLO12;
};

// physics parameters (uniform for the launch)
template <typename T> struct PhysParams {
  T dt, gravity, erp, slop, warm, damping, vmax, qlim, inv_dt;
  int iterations;
};

template <typename T, int NQ> struct PhysState {
  V3<T> pos; T qx, qy, qz, qw; V3<T> v, w;
  T q[NQ], qd[NQ];
};

template <typename T> struct JointF { SV<T> S, U, c; T Dinv, u; };

// ---------------------------------------------------------------- LDS row storage
// chunk = 16 bytes; a row's core is ROW_CORE values of T.
template <typename T> struct RowLds {
  static constexpr int PER = 16 / sizeof(T);            // values per chunk (4 float / 2 double)
  static constexpr int NCH = ROW_CORE / PER;            // 5 / 10
  using Chunk = typename std::conditional<sizeof(T) == 4, float4, double2>::type;
  Chunk* core;   // [MAX_ROWS][NCH][LANES]
  T* mu;         // [MAX_ROWS][LANES]
  T* lam;        // [MAX_ROWS][LANES]
  int* meta;     // [MAX_ROWS][LANES]   bits 0-4 parent row, 5-6 leg, 7 is_friction
  T* y;          // [12][LANES]
  int lanes, lane;

  SD void store_core(int r, const T (&v)[ROW_CORE]) const {
#pragma unroll
    for (int c = 0; c < NCH; c++) {
      Chunk ch;
      if constexpr (sizeof(T) == 4) ch = make_float4(v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]);
      else ch = make_double2(v[2 * c], v[2 * c + 1]);
      core[(r * NCH + c) * lanes + lane] = ch;
    }
  }
  SD void load_core(int r, T (&v)[ROW_CORE]) const {
#pragma unroll
    for (int c = 0; c < NCH; c++) {
      Chunk ch = core[(r * NCH + c) * lanes + lane];
      if constexpr (sizeof(T) == 4) { v[4 * c] = ch.x; v[4 * c + 1] = ch.y; v[4 * c + 2] = ch.z; v[4 * c + 3] = ch.w; }
      else { v[2 * c] = ch.x; v[2 * c + 1] = ch.y; }
    }
  }
  static constexpr size_t bytes(int lanes_) {
    return (size_t)MAX_ROWS * NCH * lanes_ * 16 + (size_t)MAX_ROWS * lanes_ * (2 * sizeof(T) + sizeof(int)) +
           (size_t)12 * lanes_ * sizeof(T);
  }
  SD void carve(unsigned char* base, int lanes_, int lane_) {
    lanes = lanes_; lane = lane_;
    core = reinterpret_cast<Chunk*>(base); base += (size_t)MAX_ROWS * NCH * lanes * 16;
    mu = reinterpret_cast<T*>(base); base += (size_t)MAX_ROWS * lanes * sizeof(T);
    lam = reinterpret_cast<T*>(base); base += (size_t)MAX_ROWS * lanes * sizeof(T);
    y = reinterpret_cast<T*>(base); base += (size_t)12 * lanes * sizeof(T);
    meta = reinterpret_cast<int*>(base);
  }
};

// ---------------------------------------------------------------- per-link rigid-body terms
// R: link axes, cw: COM relative to the common origin, v: link spatial velocity.
// Returns the link's inertia about the common origin and its bias force
//   p = v x* I v  -  damping wrench        (gyroscopic K1 + Bullet damping K3)
template <typename T>
SD void link_terms(const M3<T>& R, V3<T> cw, T m, T ix, T iy, T iz, SV<T> v, T kd, RBI<T>& I, SV<T>& p) {
  Sym3<T> Ic;
  Ic.xx = ix * R.c0.x * R.c0.x + iy * R.c1.x * R.c1.x + iz * R.c2.x * R.c2.x;
  Ic.xy = ix * R.c0.x * R.c0.y + iy * R.c1.x * R.c1.y + iz * R.c2.x * R.c2.y;
  Ic.xz = ix * R.c0.x * R.c0.z + iy * R.c1.x * R.c1.z + iz * R.c2.x * R.c2.z;
  Ic.yy = ix * R.c0.y * R.c0.y + iy * R.c1.y * R.c1.y + iz * R.c2.y * R.c2.y;
  Ic.yz = ix * R.c0.y * R.c0.z + iy * R.c1.y * R.c1.z + iz * R.c2.y * R.c2.z;
  Ic.zz = ix * R.c0.z * R.c0.z + iy * R.c1.z * R.c1.z + iz * R.c2.z * R.c2.z;
  V3<T> vc = v.l + cross(v.a, cw);
  V3<T> Icw = mul(Ic, v.a);
  T kl = kd + kd * sqrt(dot(vc, vc));
  T ka = kd + kd * sqrt(dot(v.a, v.a));
  V3<T> fc = (cross(v.a, vc) + vc * kl) * m;
  V3<T> nc = cross(v.a, Icw) + Icw * ka;
  p.a = nc + cross(cw, fc);
  p.l = fc;
  T cc = dot(cw, cw);
  I.m = m; I.h = cw * m;
  I.I.xx = Ic.xx + m * (cc - cw.x * cw.x); I.I.xy = Ic.xy - m * cw.x * cw.y; I.I.xz = Ic.xz - m * cw.x * cw.z;
  I.I.yy = Ic.yy + m * (cc - cw.y * cw.y); I.I.yz = Ic.yz - m * cw.y * cw.z;
  I.I.zz = Ic.zz + m * (cc - cw.z * cw.z);
}

template <typename T> SD V3<T> addc(V3<T> o, const M3<T>& R, double x, double y, double z) {
  // o + R*(x,y,z) with compile-time-zero components skipped
  if (x != 0.0) o = fma3(R.c0, T(x), o);
  if (y != 0.0) o = fma3(R.c1, T(y), o);
  if (z != 0.0) o = fma3(R.c2, T(z), o);
  return o;
}

// support point (relative to the common origin) of a disc with axis y of the link axes R, centre C
template <typename T> SD V3<T> disc_point(const M3<T>& R, V3<T> C, T radius) {
  T dx = -R.c0.z, dz = -R.c2.z;                       // world-down expressed in the link's x,z
  T s = radius / sqrt(dx * dx + dz * dz + T(DISC_EPS2));
  return C + R.c0 * (dx * s) + R.c2 * (dz * s);
}

template <typename T> SD void sincos_t(T x, T& s, T& c) {
  if constexpr (sizeof(T) == 4) sincosf(x, &s, &c); else sincos(x, &s, &c);
}

// ---------------------------------------------------------------- leg: FK + ABA passes 1 and 2
template <typename T, int ROBOT, int L>
SD void leg_inward(const M3<T>& R0, SV<T> v0, const T* q, const T* qd, const T* tau, T kd,
                   JointF<T> (&jf)[Robot<ROBOT>::NJ], ABI<T>& Ibase, SV<T>& pbase, V3<T>& kneeP, V3<T>& footP) {
  using RB = Robot<ROBOT>;
  constexpr int NJ = RB::NJ;
  constexpr int L0 = 1 + L * (NJ + 1);
  M3<T> Rp = R0;
  V3<T> op = mk(T(0), T(0), T(0));
  SV<T> vp = v0;
  RBI<T> Ik[NJ]; SV<T> pk[NJ]; SV<T> Sk[NJ], ck[NJ];
  static_for<NJ>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    constexpr solorl_link_data LK = RB::MD.links[L0 + k];
    constexpr int AX = LK.axis[0] != 0.0 ? 0 : 1;
    static_assert(LK.axis[2] == 0.0 && LK.jtype == 0, "revolute about x or y expected");
    V3<T> o = addc(op, Rp, LK.jorigin[0], LK.jorigin[1], LK.jorigin[2]);
    V3<T> a = AX == 0 ? Rp.c0 : Rp.c1;
    SV<T> S{a, cross(o, a)};
    T s, c; sincos_t(q[LK.dof], s, c);
    M3<T> R = rot_axis<AX>(Rp, c, s);
    SV<T> vj = S * qd[LK.dof];
    SV<T> v = vp + vj;
    Sk[k] = S; ck[k] = crm(vp, vj);
    V3<T> cw = addc(o, R, LK.com[0], LK.com[1], LK.com[2]);
    link_terms(R, cw, T(LK.mass), T(LK.inertia_box[0]), T(LK.inertia_box[1]), T(LK.inertia_box[2]), v, kd, Ik[k], pk[k]);
    if constexpr (k == NJ - 2) {  // knee disc sits on the upper leg
      constexpr solorl_prim_data PR = RB::MD.prims[12 + 2 * L];
      static_assert(PR.link == L0 + k && PR.axis == 1, "knee primitive layout");
      kneeP = disc_point(R, addc(o, R, PR.center[0], PR.center[1], PR.center[2]), T(PR.radius));
    }
    if constexpr (k == NJ - 1) {  // foot: fixed child of the last link, same axes and velocity
      constexpr solorl_link_data FT = RB::MD.links[L0 + NJ];
      static_assert(FT.jtype == 1, "foot must be a fixed joint");
      V3<T> of = addc(o, R, FT.jorigin[0], FT.jorigin[1], FT.jorigin[2]);
      V3<T> cf = addc(of, R, FT.com[0], FT.com[1], FT.com[2]);
      RBI<T> If; SV<T> pf;
      link_terms(R, cf, T(FT.mass), T(FT.inertia_box[0]), T(FT.inertia_box[1]), T(FT.inertia_box[2]), v, kd, If, pf);
      add(Ik[k], If); pk[k] = pk[k] + pf;
      constexpr solorl_prim_data PR = RB::MD.prims[13 + 2 * L];
      static_assert(PR.link == L0 + NJ && PR.axis == 1 && PR.center[0] == 0.0 && PR.center[1] == 0.0 && PR.center[2] == 0.0,
                    "foot primitive layout");
      footP = disc_point(R, of, T(PR.radius));
    }
    Rp = R; op = o; vp = v;
  });
  // pass 2: outermost joint first
  ABI<T> IA = to_abi(Ik[NJ - 1]);
  SV<T> pA = pk[NJ - 1];
  static_for<NJ>([&](auto kc) {
    constexpr int k = NJ - 1 - decltype(kc)::value;
    constexpr int dof = RB::MD.links[L0 + k].dof;
    SV<T> S = Sk[k];
    SV<T> U = mul(IA, S);
    T Dinv = T(1) / dot(S, U);
    T u = tau[dof] - dot(S, pA);
    jf[k].S = S; jf[k].U = U; jf[k].c = ck[k]; jf[k].Dinv = Dinv; jf[k].u = u;
    rank1_sub(IA, U, Dinv);
    SV<T> pa = pA + mul(IA, ck[k]) + U * (u * Dinv);
    if constexpr (k > 0) { add(IA, Ik[k - 1]); pA = pk[k - 1] + pa; }
    else { add(Ibase, IA); pbase = pbase + pa; }
  });
}

template <typename T> SD T clampv(T x, T lim) { return x > lim ? lim : (x < -lim ? -lim : x); }

// ---------------------------------------------------------------- constraint-row construction
// Builds one row for leg L: external unit wrench F applied to the link that is DEPTH joints away
// from the base (DEPTH = 0: base itself), or a unit joint torque (LIMJ >= 0, sign sg) at joint LIMJ.
template <typename T, int NJ, int DEPTH, int LIMJ>
SD void build_row(SV<T> F, T sg, const JointF<T> (&jf)[NJ], const Sym6<T>& Lam, SV<T> ub, const T* qdl /*leg rates*/,
                  T (&core)[ROW_CORE], T& denom, T& rel) {
  T t[NJ], JL[3] = {T(0), T(0), T(0)}, Y[3] = {T(0), T(0), T(0)};
  SV<T> f = F;
  static_for<NJ>([&](auto kc) {
    constexpr int k = NJ - 1 - decltype(kc)::value;
    if constexpr (LIMJ >= 0) {
      if constexpr (k > LIMJ) t[k] = T(0);
      else if constexpr (k == LIMJ) { t[k] = sg; JL[k] = sg; f = jf[k].U * (-sg * jf[k].Dinv); }
      else { t[k] = dot(jf[k].S, f); f = fma6(jf[k].U, -t[k] * jf[k].Dinv, f); }
    } else {
      if constexpr (k >= DEPTH) t[k] = T(0);
      else { t[k] = dot(jf[k].S, f); JL[k] = dot(jf[k].S, F); f = fma6(jf[k].U, -t[k] * jf[k].Dinv, f); }
    }
  });
  if constexpr (DEPTH > 0 || LIMJ >= 0) {  // leg response with the base held fixed
    SV<T> dv = zero6<T>();
    static_for<NJ>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      T yk;
      if constexpr (k == 0) yk = t[0] * jf[0].Dinv; else yk = (t[k] - dot(jf[k].U, dv)) * jf[k].Dinv;
      Y[k] = yk;
      if constexpr (k < NJ - 1) dv = fma6(jf[k].S, yk, dv);
    });
  }
  SV<T> W = mul(Lam, f);
  denom = dot(f, W);
  rel = (LIMJ >= 0) ? T(0) : dot(F, ub);
#pragma unroll
  for (int k = 0; k < NJ; k++) { denom += JL[k] * Y[k]; rel += JL[k] * qdl[k]; }
  core[0] = f.a.x; core[1] = f.a.y; core[2] = f.a.z; core[3] = f.l.x; core[4] = f.l.y; core[5] = f.l.z;
  core[6] = JL[0]; core[7] = JL[1]; core[8] = JL[2];
  core[9] = W.a.x; core[10] = W.a.y; core[11] = W.a.z; core[12] = W.l.x; core[13] = W.l.y; core[14] = W.l.z;
  core[15] = Y[0]; core[16] = Y[1]; core[17] = Y[2];
}

// the three rows (normal z, friction x, friction y) of one contact point P (relative to the base
// origin) on leg L at chain depth DEPTH; writes them to LDS and applies the warm start.
template <typename T, int NJ, int DEPTH>
SD void contact_rows(V3<T> P, T dist, T mu, T lam0, int leg, int slot_n, int slot_f, const JointF<T> (&jf)[NJ],
                     const Sym6<T>& Lam, SV<T> ub, const T* qdl, const PhysParams<T>& pp, const RowLds<T>& lds,
                     SV<T>& w, T (&yl)[3]) {
  static_for<3>([&](auto dc) {
    constexpr int d = decltype(dc)::value;   // 0: normal (z), 1: friction x, 2: friction y
    V3<T> u = d == 0 ? mk(T(0), T(0), T(1)) : (d == 1 ? mk(T(1), T(0), T(0)) : mk(T(0), T(1), T(0)));
    SV<T> F{cross(P, u), u};
    T core[ROW_CORE], denom, rel;
    build_row<T, NJ, DEPTH, -1>(F, T(0), jf, Lam, ub, qdl, core, denom, rel);
    T dinv = T(1) / denom, rhs;
    if constexpr (d == 0) {
      T pen = dist + pp.slop, pos = T(0), vel = -rel;
      if (pen > T(0)) vel -= pen * pp.inv_dt; else pos = -pen * pp.erp * pp.inv_dt;
      rhs = (pos + vel) * dinv;
    } else rhs = -rel * dinv;
    core[18] = rhs; core[19] = dinv;
    int slot = d == 0 ? slot_n : slot_f + (d - 1);
    lds.store_core(slot, core);
    lds.mu[slot * lds.lanes + lds.lane] = mu;
    lds.lam[slot * lds.lanes + lds.lane] = d == 0 ? lam0 : T(0);
    lds.meta[slot * lds.lanes + lds.lane] = (d == 0 ? 0 : (slot_n | 128)) | (leg << 5);
    if constexpr (d == 0) {  // warm start: dV += M^-1 J^T lam0
      w.a = fma3(mk(core[9], core[10], core[11]), lam0, w.a);
      w.l = fma3(mk(core[12], core[13], core[14]), lam0, w.l);
      yl[0] += core[15] * lam0; yl[1] += core[16] * lam0; yl[2] += core[17] * lam0;
    }
  });
}

// ---------------------------------------------------------------- one physics sub-step
// tau: joint torques applied during this sub-step.  lam_prev: per-primitive warm-start impulses
// (global memory, stride = nstride).  Returns the contact bit mask.
template <typename T, int ROBOT>
SD int substep(PhysState<T, Robot<ROBOT>::NQ>& st, const T* tau, const PhysParams<T>& pp, T* lam_prev, size_t nstride,
               const RowLds<T>& lds) {
  using RB = Robot<ROBOT>;
  constexpr int NJ = RB::NJ, NQ = RB::NQ;
  const T dt = pp.dt;
  M3<T> R0 = quat_to_mat(st.qx, st.qy, st.qz, st.qw);
  SV<T> v0{st.w, st.v};

  // ---- stage A: base link + the four legs (FK, velocities, inertias, ABA passes 1-2)
  JointF<T> jf[4][NJ];
  V3<T> kneeP[4], footP[4];
  ABI<T> Ibase; SV<T> pbase;
  {
    constexpr solorl_link_data B = RB::MD.links[0];
    static_assert(B.com[0] == 0.0 && B.com[1] == 0.0 && B.com[2] == 0.0, "base COM at its origin");
    RBI<T> Ib;
    link_terms(R0, mk(T(0), T(0), T(0)), T(B.mass), T(B.inertia_box[0]), T(B.inertia_box[1]), T(B.inertia_box[2]), v0,
               pp.damping, Ib, pbase);
    Ibase = to_abi(Ib);
  }
  static_for<4>([&](auto lc) {
    constexpr int L = decltype(lc)::value;
    leg_inward<T, ROBOT, L>(R0, v0, st.q, st.qd, tau, pp.damping, jf[L], Ibase, pbase, kneeP[L], footP[L]);
  });

  // ---- stage B: base acceleration (gravity via the accelerating-frame trick), pass 3, u* = u + dt*udot
  Sym6<T> Lam = spd_inverse(Ibase);
  SV<T> a0 = mul(Lam, pbase) * T(-1);
  T qds[NQ];
  static_for<4>([&](auto lc) {
    constexpr int L = decltype(lc)::value;
    SV<T> ap = a0;
    static_for<NJ>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      const JointF<T>& J = jf[L][k];
      SV<T> apc = ap + J.c;
      T qdd = (J.u - dot(J.U, apc)) * J.Dinv;
      if constexpr (k < NJ - 1) ap = fma6(J.S, qdd, apc);
      qds[L * NJ + k] = clampv(st.qd[L * NJ + k] + dt * qdd, pp.vmax);
    });
  });
  V3<T> vdot = a0.l + cross(st.w, st.v); vdot.z -= pp.gravity;
  SV<T> ub;
  ub.a = mk(clampv(st.w.x + dt * a0.a.x, pp.vmax), clampv(st.w.y + dt * a0.a.y, pp.vmax), clampv(st.w.z + dt * a0.a.z, pp.vmax));
  ub.l = mk(clampv(st.v.x + dt * vdot.x, pp.vmax), clampv(st.v.y + dt * vdot.y, pp.vmax), clampv(st.v.z + dt * vdot.z, pp.vmax));

  // ---- stage C: collision detection (start-of-step pose) and constraint rows
  T dist[NPRIM]; V3<T> PP[NPRIM];
  int mask = 0;
  static_for<12>([&](auto pc) {
    constexpr int p = decltype(pc)::value;
    constexpr solorl_prim_data PR = RB::MD.prims[p];
    static_assert(PR.link == 0 && PR.axis == -1, "base primitives are points");
    PP[p] = mul(R0, mk(T(PR.center[0]), T(PR.center[1]), T(PR.center[2])));
    dist[p] = st.pos.z + PP[p].z;
    if (dist[p] < T(PR.margin)) mask |= 1 << p;
  });
  static_for<4>([&](auto lc) {
    constexpr int L = decltype(lc)::value;
    PP[12 + 2 * L] = kneeP[L]; PP[13 + 2 * L] = footP[L];
    dist[12 + 2 * L] = st.pos.z + kneeP[L].z; dist[13 + 2 * L] = st.pos.z + footP[L].z;
    if (dist[12 + 2 * L] < T(RB::MD.prims[12 + 2 * L].margin)) mask |= 1 << (12 + 2 * L);
    if (dist[13 + 2 * L] < T(RB::MD.prims[13 + 2 * L].margin)) mask |= 1 << (13 + 2 * L);
  });
  if (__popc(mask) > MAX_CONTACTS) {   // keep the MAX_CONTACTS deepest (ties: lower primitive id)
    int keep = 0;
    static_for<NPRIM>([&](auto pc) {
      constexpr int p = decltype(pc)::value;
      int rank = 0;
      static_for<NPRIM>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        if constexpr (q != p) rank += ((mask >> q) & 1) && (dist[q] < dist[p] || (dist[q] == dist[p] && q < p));
      });
      if (((mask >> p) & 1) && rank < MAX_CONTACTS) keep |= 1 << p;
    });
    mask = keep;
  }
  const int nc = __popc(mask);

  SV<T> w = zero6<T>();
  T y[4][3];
#pragma unroll
  for (int L = 0; L < 4; L++) { y[L][0] = T(0); y[L][1] = T(0); y[L][2] = T(0); }
  int nlim = 0;
  static_for<4>([&](auto lc) {          // joint-limit rows (K5): first MAX_LIMITS in joint order
    constexpr int L = decltype(lc)::value;
    static_for<NJ>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      const T qj = st.q[L * NJ + k];
#pragma unroll
      for (int side = 0; side < 2; side++) {
        T pen = side == 0 ? qj + pp.qlim : pp.qlim - qj;
        if (pen < T(LIMIT_WINDOW) && nlim < MAX_LIMITS) {
          T sg = side == 0 ? T(1) : T(-1);
          T core[ROW_CORE], denom, rel;
          build_row<T, NJ, 0, k>(zero6<T>(), sg, jf[L], Lam, ub, &qds[L * NJ], core, denom, rel);
          T dinv = T(1) / denom, pos = T(0), vel = -rel;
          if (pen > T(0)) vel -= pen * pp.inv_dt; else pos = -pen * pp.erp * pp.inv_dt;
          core[18] = (pos + vel) * dinv; core[19] = dinv;
          lds.store_core(nlim, core);
          lds.mu[nlim * lds.lanes + lds.lane] = T(0);
          lds.lam[nlim * lds.lanes + lds.lane] = T(0);
          lds.meta[nlim * lds.lanes + lds.lane] = L << 5;
          nlim++;
        }
      }
    });
  });
  static_for<NPRIM>([&](auto pc) {
    constexpr int p = decltype(pc)::value;
    constexpr solorl_prim_data PR = RB::MD.prims[p];
    T lam0 = T(0);
    if ((mask >> p) & 1) {
      const int cidx = __popc(mask & ((1 << p) - 1));
      const int slot_n = nlim + cidx, slot_f = nlim + nc + 2 * cidx;
      lam0 = pp.warm * lam_prev[(size_t)p * nstride];
      if constexpr (p < 12) {
        T dummy[3] = {T(0), T(0), T(0)};
        contact_rows<T, NJ, 0>(PP[p], dist[p], T(PR.friction), lam0, 0, slot_n, slot_f, jf[0], Lam, ub, &qds[0], pp, lds, w, dummy);
      } else {
        constexpr int L = (p - 12) / 2;
        constexpr int DEPTH = ((p - 12) % 2 == 0) ? NJ - 1 : NJ;
        contact_rows<T, NJ, DEPTH>(PP[p], dist[p], T(PR.friction), lam0, L, slot_n, slot_f, jf[L], Lam, ub, &qds[L * NJ], pp, lds, w, y[L]);
      }
    }
  });
  const int nrows = nlim + 3 * nc;
#pragma unroll
  for (int L = 0; L < 4; L++)
#pragma unroll
    for (int k = 0; k < 3; k++) lds.y[(L * 3 + k) * lds.lanes + lds.lane] = y[L][k];

  // ---- stage D: projected Gauss-Seidel (K7): limits, normals, then friction rows
  for (int it = 0; it < pp.iterations; it++) {
    for (int r = 0; __any(r < nrows); r++) {
      if (r < nrows) {
        T c[ROW_CORE];
        lds.load_core(r, c);
        const int meta = lds.meta[r * lds.lanes + lds.lane];
        const int leg = (meta >> 5) & 3;
        T lam = lds.lam[r * lds.lanes + lds.lane];
        T* yp = lds.y + (leg * 3) * lds.lanes + lds.lane;
        T y0 = yp[0], y1 = yp[lds.lanes], y2 = yp[2 * lds.lanes];
        T lo = T(0), hi = T(1e30);
        if (meta & 128) { hi = lds.mu[r * lds.lanes + lds.lane] * lds.lam[(meta & 31) * lds.lanes + lds.lane]; lo = -hi; }
        T jdv = c[0] * w.a.x + c[1] * w.a.y + c[2] * w.a.z + c[3] * w.l.x + c[4] * w.l.y + c[5] * w.l.z +
                c[6] * y0 + c[7] * y1 + c[8] * y2;
        T sum = lam + (c[18] - jdv * c[19]);
        sum = sum < lo ? lo : (sum > hi ? hi : sum);
        T delta = sum - lam;
        lds.lam[r * lds.lanes + lds.lane] = sum;
        w.a.x += c[9] * delta; w.a.y += c[10] * delta; w.a.z += c[11] * delta;
        w.l.x += c[12] * delta; w.l.y += c[13] * delta; w.l.z += c[14] * delta;
        yp[0] = y0 + c[15] * delta; yp[lds.lanes] = y1 + c[16] * delta; yp[2 * lds.lanes] = y2 + c[17] * delta;
      }
    }
  }

  // ---- stage E: apply delta-velocities (clamp K5), cache impulses, semi-implicit Euler (K1)
  static_for<NPRIM>([&](auto pc) {
    constexpr int p = decltype(pc)::value;
    T l = T(0);
    if ((mask >> p) & 1) l = lds.lam[(nlim + __popc(mask & ((1 << p) - 1))) * lds.lanes + lds.lane];
    lam_prev[(size_t)p * nstride] = l;
  });
  st.w = mk(clampv(ub.a.x + w.a.x, pp.vmax), clampv(ub.a.y + w.a.y, pp.vmax), clampv(ub.a.z + w.a.z, pp.vmax));
  st.v = mk(clampv(ub.l.x + w.l.x, pp.vmax), clampv(ub.l.y + w.l.y, pp.vmax), clampv(ub.l.z + w.l.z, pp.vmax));
  static_for<4>([&](auto lc) {
    constexpr int L = decltype(lc)::value;
    SV<T> dv = w;
    static_for<NJ>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      const JointF<T>& J = jf[L][k];
      T z = -dot(J.U, dv) * J.Dinv;
      if constexpr (k < NJ - 1) dv = fma6(J.S, z, dv);
      T yk = lds.y[(L * 3 + k) * lds.lanes + lds.lane];
      T nv = clampv(qds[L * NJ + k] + yk + z, pp.vmax);
      st.qd[L * NJ + k] = nv;
      st.q[L * NJ + k] += dt * nv;
    });
  });
  st.pos = fma3(st.v, dt, st.pos);
  {
    T wn2 = dot(st.w, st.w), wn = sqrt(wn2), sc, s_, cw;
    sincos_t(T(0.5) * wn * dt, s_, cw);
    if (wn < T(1e-3)) sc = T(0.5) * dt - dt * dt * dt * T(1.0 / 48.0) * wn2; else sc = s_ / wn;
    T ax = st.w.x * sc, ay = st.w.y * sc, az = st.w.z * sc;
    T x = st.qx, yq = st.qy, z = st.qz, ww = st.qw;
    T nx = cw * x + ax * ww + ay * z - az * yq;
    T ny = cw * yq - ax * z + ay * ww + az * x;
    T nz = cw * z + ax * yq - ay * x + az * ww;
    T nw = cw * ww - ax * x - ay * yq - az * z;
    T inv = T(1) / sqrt(nx * nx + ny * ny + nz * nz + nw * nw);
    st.qx = nx * inv; st.qy = ny * inv; st.qz = nz * inv; st.qw = nw * inv;
  }
  return mask;
}

}  // namespace solo
