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
//     accumulator w stays in registers during the PGS sweep;
//   * register diet: the sub-step is a sequence of NON-INLINED phase functions that hand off through
//     an explicit per-lane context (private memory), so every phase -- above all the PGS sweep --
//     gets its own register allocation and nothing is spilled inside a hot loop.  Legs are
//     processed one at a time; of a finished leg only its 3x6 response matrix G
//     (dq_leg = y - G w, qdd_leg = qdd0 - G a_base) survives, its constraint rows are parked
//     half-built in LDS and completed by ONE generic loop once the base inverse inertia is known.
#pragma once
#include <utility>

#include "../../include/solorl_model_data.h"
#include "spatial.hpp"

namespace solo {

// Slots of the constraint solve.  A robot gets up to MAX_LIMITS = 4 joint-limit rows and MAX_CONTACTS = 8 contact points, but
// limit rows 3 and 4 take the place of contact points: the contact cap is 8 - max(limit rows - 2, 0), so that the row count
// stays <= 2 + 3 * 8 = 26 and the sweep keeps its 13 static slots (TeamRows).  Measured on the uncapped oracle (tests/test_oracle_caps.py,
// random policies): > 8 contact points in < 0.01 % of env-steps, > 4 joints beyond their limits never in 600 k env-steps, > 2 in
// 0.05-0.17 % -- two limit rows (rounds 1-2) changed the next joint angles of 0.2 % of env-steps by up to 0.4 rad.
constexpr int MAX_CONTACTS = 8;
constexpr int MAX_LIMITS = 4;
constexpr int LIM_SLOT_ROWS = 2;                          // limit rows in the sweep's own limit slot; the others sit at the first normal positions
constexpr int MAX_ROWS = LIM_SLOT_ROWS + 3 * MAX_CONTACTS;   // 26
SD int extra_limits(int nlt) { return nlt > LIM_SLOT_ROWS ? nlt - LIM_SLOT_ROWS : 0; }
SD int contact_cap(int nlt) { return MAX_CONTACTS - extra_limits(nlt); }
constexpr int ROW_CORE = 20;                              // Jh6 JL3 W6 Y3 rhs dinv
constexpr int NPRIM = 24;                                // 12 base points, knee + foot per leg (12 + 2 leg + i), Solo12: shoulder housing per leg (20 + leg)
// A joint-limit row exists while the joint is AT or BEYOND its limit (margin <= 0), as btMultiBodyJointLimitConstraint::createConstraintRows
// (`if (penetration > 0) continue`) [K5]; rounds 1-2 opened a speculative row 0.5 rad before the limit, which made several joints
// compete for the rows ten times as often (1.2-2.5 % of random-policy env-steps had more than two candidates, now 0.05-0.17 %).
constexpr double LIMIT_WINDOW = 0.0;
constexpr double DISC_EPS2 = 1e-12;
constexpr double DISC_RIM = 0.1;                         // sine of the tilt at which a thick disc rests fully on its face (disc_point)

// dev builds (-DSOLO_WAVE_TIMING, tools/dev/wave_hist.py): four time stamps per wavefront, kept in registers and written with
// plain stores at the very end (no atomics, nothing on the sub-steps' path): [workgroup][0] kernel start -> first sub-step,
// [1] the sub-steps, [2] last sub-step -> end, [3] whole kernel in s_memrealtime ticks (100 MHz), [4] sum over the sub-steps of
// the slots the wave swept, [5] the maximum over the sub-steps of its largest contact count
#if defined(SOLO_WAVE_TIMING) && !defined(SOLO_HOST_SHIM)
constexpr int SOLO_WT_WAVES = 65536, SOLO_WT_FIELDS = 28;    // [6..15]: the ten intervals between the kernel's stamps; [16..25]: per-phase
__device__ unsigned long long solo_wave_times[SOLO_WT_WAVES][SOLO_WT_FIELDS];                                // sums over the sub-steps (SOLO_PT)
// phase stamps of substep_team: after a full drain of the wave's memory counters (timing build only), lane 0 adds the ticks since
// the previous stamp to phase slot i of the workgroup's LDS accumulators; step_team writes them out with its own stamps
__shared__ unsigned long long solo_pt_acc[10];
#define SOLO_PT_DECL() long long pt_last_ = 0
#define SOLO_PT_BEGIN() do { __builtin_amdgcn_s_waitcnt(0); pt_last_ = clock64(); } while (0)
#define SOLO_PT(i) do { __builtin_amdgcn_s_waitcnt(0); if (threadIdx.x == 0) { const long long n_ = clock64(); solo_pt_acc[i] += (unsigned long long)(n_ - pt_last_); } pt_last_ = clock64(); } while (0)
#else
#define SOLO_PT_DECL() do {} while (0)
#define SOLO_PT_BEGIN() do {} while (0)
#define SOLO_PT(i) do {} while (0)
#endif

template <int N, typename F, int... I>
SD void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F> SD void static_for(F&& f) { static_for_impl<N>(f, std::make_integer_sequence<int, N>{}); }

template <int ROBOT> struct Robot;
template <> struct Robot<0> {
  static constexpr int NJ = 2, NQ = 8, NL = 13;
  static constexpr bool SHOULDER = false;                  // no HAA joint, no shoulder housing
  static constexpr const solorl_model_data& MD = SOLORL_MODEL_SOLO8;
};
template <> struct Robot<1> {
  static constexpr int NJ = 3, NQ = 12, NL = 17;
  static constexpr bool SHOULDER = true;                   // primitives 20..23: the shoulder housings (disc about the link's x axis)
  static constexpr const solorl_model_data& MD = SOLORL_MODEL_SOLO12;
};

// the twelve base primitives as sign patterns of prims[0] and prims[8] (phase_front_team computes them from the lane index)
constexpr bool base_points_are_sign_patterns(const solorl_model_data& md) {
  const solorl_prim_data& a = md.prims[0]; const solorl_prim_data& b = md.prims[8];
  for (int p = 0; p < 12; p++) {
    const solorl_prim_data& q = md.prims[p];
    const bool plate = p >= 8;
    const double sx = ((plate ? (p - 8) >> 1 : p >> 2) & 1) ? -1.0 : 1.0, sy = ((plate ? (p - 8) : p >> 1) & 1) ? -1.0 : 1.0;
    const double cx = (plate ? b.center[0] : a.center[0]) * sx, cy = (plate ? b.center[1] : a.center[1]) * sy;
    const double cz = plate ? b.center[2] : ((p & 1) ? -a.center[2] : a.center[2]);
    if (q.link != 0 || q.axis != -1 || q.center[0] != cx || q.center[1] != cy || q.center[2] != cz || q.margin != a.margin || q.friction != a.friction) return false;
  }
  return true;
}

// physics parameters (uniform for the launch)
template <typename T> struct PhysParams {
  T dt, gravity, erp, slop, warm, damping, vmax, qlim, inv_dt;
  int iterations;
  T tm_hw, tm_mu;              // treadmill strip (include/solorl.h treadmill_*): half width, friction factor
  T resid_thr;                 // sqrt(solver_residual_threshold): velocity-level change below which a solve stops (K7); < 0: never
  T cerp;                      // error reduction of the contact rows (solorl_config contact_erp = Bullet's m_erp2); erp above: joint-limit rows (m_erp)
  T cmargin;                   // collision margin around every primitive (solorl_config collision_margin: Bullet's 1 mm around URDF hulls): a shape
                               // swept by a sphere of that radius has its support point the margin further down -- P.z -= cmargin in this world-aligned frame
  // the switches share one word (the struct travels to every non-inlined phase function in argument VGPRs: four separate ints cost the
  // collision front its last free registers)
  int mode;
  enum { M_PIPE = 1, M_URDF = 2, M_CONE = 4, M_TREADMILL = 8 };
  SD bool pgs_pipe() const { return mode & M_PIPE; }          // team-mode sweep: software-pipelined (a wavefront per SIMD) or plain (two per SIMD), see pgs_team_variant
  SD bool urdf_inertia() const { return mode & M_URDF; }      // K2: Bullet's default box inertia from the collision AABB, or the URDF tensor (use_urdf_inertia)
  SD bool cone() const { return mode & M_CONE; }              // solorl_config friction_model: Bullet's implicit friction cone (the two friction rows of a contact solved together)
  SD bool tm_on() const { return mode & M_TREADMILL; }        // treadmill strip present
  void set_mode(bool pipe, bool urdf, bool cone_, bool treadmill) {      // (host side: make_phys)
    mode = (pipe ? M_PIPE : 0) | (urdf ? M_URDF : 0) | (cone_ ? M_CONE : 0) | (treadmill ? M_TREADMILL : 0);
  }
};
// bit 24+f of a sub-step's returned mask: foot f's contact lies on the treadmill strip (foot primitive = 13 + 2f)
SD int strip_feet_bits(int smask) {
  return (((smask >> 13) & 1) << 24) | (((smask >> 15) & 1) << 25) | (((smask >> 17) & 1) << 26) | (((smask >> 19) & 1) << 27);
}

template <typename T, int NQ> struct PhysState {
  V3<T> pos; T qx, qy, qz, qw; V3<T> v, w;
  T q[NQ], qd[NQ];
};

template <typename T> struct JointF { SV<T> S, U, c; T Dinv, u; };

// ---------------------------------------------------------------- LDS row storage
// One row = ROW_CORE values of T, stored [row][lane][ROW_CORE]: a lane's row is one 80-byte (fp32)
// record read with five ds_read_b128 at immediate offsets.  80 B = 20 dwords and 5 is coprime with
// 16, so the 16 lanes of every b128 lane group hit 64 distinct banks (conflict-free).
// The struct holds only (lanes, lane): every phase function re-derives its pointers from the
// `extern __shared__` symbol so they stay address-space-3 (ds_* instructions) across the
// non-inlined phase boundaries -- generic pointers would turn every access into a FLAT one.
#ifndef SOLO_HOST_SHIM
extern __shared__ __attribute__((aligned(16))) unsigned char solo_smem_sym[];
// The step kernels have no static LDS, so their dynamic LDS starts at offset SOLO_LDS_BASE = 0 and every phase function addresses
// it from that constant.  (Through the `extern __shared__` symbol a NON-kernel function finds the start of the dynamic LDS with a
// scalar load from a per-kernel table -- an exposed scalar-cache round trip at the top of every phase call, and, since scalar loads
// return out of order, an s_waitcnt lgkmcnt(0) that drains every LDS read in flight wherever the compiler re-loads it.)
// Checked on the HOST: solorl_step compares the loaded kernel's static LDS size (hipFuncGetAttributes) with the constant before its
// first launch (check_lds_base, solorl_hip.hip: SOLORL_ERR_HIP instead of wrong physics), and tests/test_abi.py reads the same
// number from the built code object.  Nothing is checked on the device.
#if defined(SOLO_WAVE_TIMING)
constexpr unsigned SOLO_LDS_BASE = 80;                   // behind solo_pt_acc[10] -- which only the TEAM kernel references: the timing build
                                                         // is team-mode only (check_lds_base rejects the lane-mode kernel there, by design)
#else
constexpr unsigned SOLO_LDS_BASE = 0;
#endif
typedef __attribute__((address_space(3))) unsigned char solo_lds_byte;
// (written as a non-zero LDS address minus its distance from the base: LDS offset 0 itself would be taken for the null pointer,
// whose generic counterpart is not LDS offset 0)
#define solo_smem (((unsigned char*)reinterpret_cast<solo_lds_byte*>(SOLO_LDS_BASE + 4096u)) - 4096)
#endif
#ifdef SOLO_HOST_SHIM
template <typename T> constexpr int default_lanes() { return 1; }
#else
template <typename T> constexpr int default_lanes() { return sizeof(T) == 4 ? 64 : 32; }
#endif
// LN = env columns per workgroup (compile-time, so every LDS offset below is an instruction
// immediate): 64 (fp32) / 32 (fp64) in lane mode, 4 in team mode (16 lanes per env).
template <typename T, int LN = default_lanes<T>()> struct RowLds {
  static constexpr int PER = 16 / sizeof(T);            // values per 16-byte chunk (4 float / 2 double)
  static constexpr int NCH = ROW_CORE / PER;            // 5 / 10
  static constexpr int LANES = LN;
  using Chunk = typename std::conditional<sizeof(T) == 4, float4, double2>::type;
  // Layouts (all lane-minor => conflict-free for every access width the compiler may choose;
  // the first [row][lane][20] record layout measured 60 % bank-conflict cycles, profiles/r01_pmc_v2.txt):
  //   core [MAX_ROWS][NCH][LANES] chunks     row record, 16-byte chunks
  //   aux  [MAX_ROWS][4][LANES]              mu, lam, 3*leg (as T), meta bits (as T)
  //   y    [12][LANES]                       leg delta-rates with the base held fixed
  // meta bits: 0-4 parent row, 5-6 leg, 7 is_friction, 8-9 direction (0 z, 1 x, 2 y, 3 joint)
  enum { A_MU = 0, A_LAM = 1, A_YOFF = 2, A_META = 3 };
  int lanes, lane;   // lanes kept for the launch-side size computation; device code uses LANES
#ifdef SOLO_HOST_SHIM
  unsigned char* base;
  SD unsigned char* smem() const { return base; }
#else
  SD unsigned char* smem() const { return solo_smem; }
#endif
  SD Chunk* core() const { return reinterpret_cast<Chunk*>(smem()) + lane; }                 // + (r*NCH + c)*LANES
  SD T* aux() const { return reinterpret_cast<T*>(smem() + (size_t)MAX_ROWS * LANES * ROW_CORE * sizeof(T)) + lane; }  // + (r*4 + f)*LANES
  SD T* y() const { return reinterpret_cast<T*>(smem() + (size_t)MAX_ROWS * LANES * (ROW_CORE + 4) * sizeof(T)) + lane; }  // + k*LANES
  SD T* hdr() const { return y() + 12 * LANES; }   // [8][LANES]: team-mode hand-off (nlim, nc, w[6])
  SD T& A(int r, int f) const { return aux()[(r * 4 + f) * LANES]; }

  SD void store_core(int r, const T (&v)[ROW_CORE]) const {
    Chunk* p = core() + r * (NCH * LANES);
#pragma unroll
    for (int c = 0; c < NCH; c++) {
      Chunk ch;
      if constexpr (sizeof(T) == 4) ch = make_float4(v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]);
      else ch = make_double2(v[2 * c], v[2 * c + 1]);
      p[c * LANES] = ch;
    }
  }
  SD void load_core(int r, T (&v)[ROW_CORE]) const {
    const Chunk* p = core() + r * (NCH * LANES);
#pragma unroll
    for (int c = 0; c < NCH; c++) {
      Chunk ch = p[c * LANES];
      if constexpr (sizeof(T) == 4) { v[4 * c] = ch.x; v[4 * c + 1] = ch.y; v[4 * c + 2] = ch.z; v[4 * c + 3] = ch.w; }
      else { v[2 * c] = ch.x; v[2 * c + 1] = ch.y; }
    }
  }
  static constexpr size_t bytes(int) {
    // (the 8-value team header only exists in team mode: lane mode is within 1 KB of the 160 KB LDS)
    return (size_t)MAX_ROWS * LANES * (ROW_CORE + 4) * sizeof(T) + (size_t)(LANES == 4 ? 20 : 12) * LANES * sizeof(T);
  }
};

// ---------------------------------------------------------------- per-link rigid-body terms
// R: link axes, cw: COM relative to the common origin, v: link spatial velocity.
// Returns the link's inertia about the common origin and its bias force
//   p = v x* I v  -  damping wrench        (gyroscopic K1 + Bullet damping K3)
// OD: bit 0 / 1 / 2 = the link tensor has an xy / xz / yz product (URDF inertia only; compile time, so the box rule pays nothing)
template <int OD = 0, typename T>
SD void link_terms(const M3<T>& R, V3<T> cw, T m, T ix, T iy, T iz, SV<T> v, T kd, RBI<T>& I, SV<T>& p, T ixy = T(0), T ixz = T(0),
                   T iyz = T(0)) {
  Sym3<T> Ic;
  Ic.xx = ix * R.c0.x * R.c0.x + iy * R.c1.x * R.c1.x + iz * R.c2.x * R.c2.x;
  Ic.xy = ix * R.c0.x * R.c0.y + iy * R.c1.x * R.c1.y + iz * R.c2.x * R.c2.y;
  Ic.xz = ix * R.c0.x * R.c0.z + iy * R.c1.x * R.c1.z + iz * R.c2.x * R.c2.z;
  Ic.yy = ix * R.c0.y * R.c0.y + iy * R.c1.y * R.c1.y + iz * R.c2.y * R.c2.y;
  Ic.yz = ix * R.c0.y * R.c0.z + iy * R.c1.y * R.c1.z + iz * R.c2.y * R.c2.z;
  Ic.zz = ix * R.c0.z * R.c0.z + iy * R.c1.z * R.c1.z + iz * R.c2.z * R.c2.z;
  // products of inertia: R (e_a e_b^T + e_b e_a^T) R^T = c_a c_b^T + c_b c_a^T
  auto add_od = [&](T w, const V3<T>& a, const V3<T>& b) {
    Ic.xx += T(2) * w * a.x * b.x; Ic.yy += T(2) * w * a.y * b.y; Ic.zz += T(2) * w * a.z * b.z;
    Ic.xy += w * (a.x * b.y + a.y * b.x); Ic.xz += w * (a.x * b.z + a.z * b.x); Ic.yz += w * (a.y * b.z + a.z * b.y);
  };
  if constexpr (OD & 1) add_od(ixy, R.c0, R.c1);
  if constexpr (OD & 2) add_od(ixz, R.c0, R.c2);
  if constexpr (OD & 4) add_od(iyz, R.c1, R.c2);
  V3<T> vc = v.l + cross(v.a, cw);
  V3<T> Icw = mul(Ic, v.a);
  T kl = kd + kd * sqrt_fast(dot(vc, vc));
  T ka = kd + kd * sqrt_fast(dot(v.a, v.a));
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

// support point (relative to the common origin) of a disc with axis y of the link axes R, centre C, half-thickness halfw (K6, round 3:
// a cylinder with rounded rims -- the lowest point moves from the mid-plane towards the face that points down and reaches it at a tilt
// of asin(DISC_RIM), so a link lying on its side rests half a thickness lower, as its hull does; oracle: prim_point)
template <typename T> SD V3<T> disc_point(const M3<T>& R, V3<T> C, T radius, T halfw = T(0)) {
  T dx = -R.c0.z, dz = -R.c2.z;                       // world-down expressed in the link's x,z
  T s = radius * rsqrt_fast(dx * dx + dz * dz + T(DISC_EPS2));
  V3<T> P = C + R.c0 * (dx * s) + R.c2 * (dz * s);
  T w = -R.c1.z * T(1.0 / DISC_RIM);                  // world-down along the axis, in rim units
  w = w > T(1) ? T(1) : (w < T(-1) ? T(-1) : w);
  return P + R.c1 * (halfw * w);
}
// Body of revolution about the link's y axis given by NR profile rings (ring_y[i] >= 0 from the centre plane C, ring_r[i]; model table,
// tools/compile_model.py ring_profile; the feet): the support point lies on the ring maximising r_i |d_perp| + y_i |d_axis|.  Ring 0, the
// tread, is crossed smoothly like disc_point's thickness; the others are the hull's own edges.  (oracle: prim_point)
template <typename T, int NR> SD V3<T> ring_disc_point(const M3<T>& R, V3<T> C, const T (&ry)[NR], const T (&rr)[NR]) {
  const T dx = -R.c0.z, dz = -R.c2.z, dy = -R.c1.z;   // world-down in the link's axes
  const T s2 = dx * dx + dz * dz + T(DISC_EPS2);
  const T is = rsqrt_fast(s2), s = s2 * is, a = fabs(dy);
  T aw = a * T(1.0 / DISC_RIM); aw = aw > T(1) ? T(1) : aw;
  T vb = rr[0] * s + ry[0] * a, r = rr[0], y = ry[0] * aw;
#pragma unroll
  for (int i = 1; i < NR; i++) { const T vi = rr[i] * s + ry[i] * a; const bool b = vi > vb; vb = b ? vi : vb; r = b ? rr[i] : r; y = b ? ry[i] : y; }
  const T sc = r * is;
  return C + R.c0 * (dx * sc) + R.c2 * (dz * sc) + R.c1 * (dy < T(0) ? -y : y);
}
// the same for a disc about the link's x axis (shoulder housings)
template <typename T> SD V3<T> disc_point_x(const M3<T>& R, V3<T> C, T radius) {
  T dy = -R.c1.z, dz = -R.c2.z;                       // world-down expressed in the link's y,z
  T s = radius * rsqrt_fast(dy * dy + dz * dz + T(DISC_EPS2));
  return C + R.c1 * (dy * s) + R.c2 * (dz * s);
}

// sin/cos of a bounded angle (|x| < ~1e3: joint angles are limited to +-10 rad).  fp32: 3-constant
// Cody-Waite reduction by pi/2 + minimax polynomials on [-pi/4, pi/4] (~30 VALU, <= 1 ulp), instead
// of libm's sincosf whose large-argument path costs ~200 instructions and a dozen branches per call
// (12 calls per sub-step were a third of phase_detect).  fp64 (validation build): libm.
template <typename T> SD void sincos_t(T x, T& s, T& c) {
  if constexpr (sizeof(T) == 4) {
    const float k = rintf(x * 0.636619772367581343f);
    float r = fmaf(-k, 1.5703125f, x);
    r = fmaf(-k, 4.837512969970703125e-4f, r);
    r = fmaf(-k, 7.54978995489188216e-8f, r);
    const float z = r * r;
    const float sp = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f), z * r, r);
    const float cp = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f), z * z,
                          fmaf(-0.5f, z, 1.0f));
    const int q = (int)k & 3;
    const float ss = (q & 1) ? cp : sp, cc = (q & 1) ? sp : cp;
    s = (q & 2) ? -ss : ss;
    c = ((q + 1) & 2) ? -cc : cc;
  } else {
    sincos(x, &s, &c);
  }
}

// ---------------------------------------------------------------- leg kinematics (shared)
// frames of the NJ moving links of leg L from cached sin/cos; calls f(k, R_k, o_k, a_k) per joint.
template <typename T, int ROBOT, int L, typename F>
SD void leg_frames(const M3<T>& R0, const T* sn, const T* cs, F&& f) {
  using RB = Robot<ROBOT>;
  constexpr int NJ = RB::NJ;
  constexpr int L0 = 1 + L * (NJ + 1);
  M3<T> Rp = R0;
  V3<T> op = mk(T(0), T(0), T(0));
  static_for<NJ>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    constexpr solorl_link_data LK = RB::MD.links[L0 + k];
    constexpr int AX = LK.axis[0] != 0.0 ? 0 : 1;
    static_assert(LK.axis[2] == 0.0 && LK.jtype == 0, "revolute about x or y expected");
    V3<T> o = addc(op, Rp, LK.jorigin[0], LK.jorigin[1], LK.jorigin[2]);
    V3<T> a = AX == 0 ? Rp.c0 : Rp.c1;
    M3<T> R = rot_axis<AX>(Rp, cs[LK.dof], sn[LK.dof]);
    f(kc, R, o, a);
    Rp = R; op = o;
  });
}

// knee and foot support points of leg L (relative to the base origin)
template <typename T, int ROBOT, int L>
SD void leg_prim_points(const M3<T>& R0, const T* sn, const T* cs, V3<T>& kneeP, V3<T>& footP, V3<T>& shP) {
  using RB = Robot<ROBOT>;
  constexpr int NJ = RB::NJ;
  constexpr int L0 = 1 + L * (NJ + 1);
  leg_frames<T, ROBOT, L>(R0, sn, cs, [&](auto kc, const M3<T>& R, V3<T> o, V3<T>) {
    constexpr int k = decltype(kc)::value;
    if constexpr (RB::SHOULDER && k == 0) {  // shoulder housing: disc about x on the HAA link
      constexpr solorl_prim_data PR = RB::MD.prims[20 + L];
      static_assert(PR.link == L0 && PR.axis == 0, "shoulder primitive layout");
      shP = disc_point_x(R, addc(o, R, PR.center[0], PR.center[1], PR.center[2]), T(PR.radius));
    }
    if constexpr (k == NJ - 2) {  // knee disc sits on the upper leg
      constexpr solorl_prim_data PR = RB::MD.prims[12 + 2 * L];
      static_assert(PR.link == L0 + k && PR.axis == 1, "knee primitive layout");
      kneeP = disc_point(R, addc(o, R, PR.center[0], PR.center[1], PR.center[2]), T(PR.radius), T(PR.halfw));
    }
    if constexpr (k == NJ - 1) {  // foot: fixed child of the last link
      constexpr solorl_link_data FT = RB::MD.links[L0 + NJ];
      constexpr solorl_prim_data PR = RB::MD.prims[13 + 2 * L];
      static_assert(FT.jtype == 1 && PR.link == L0 + NJ && PR.axis == 1 && PR.center[0] == 0.0 && PR.center[1] == 0.0 &&
                    PR.center[2] == 0.0, "foot primitive layout");
      static_assert(PR.nring == 4, "foot primitive: four profile rings (tools/compile_model.py)");
      const T fy[4] = {T(PR.ring_y[0]), T(PR.ring_y[1]), T(PR.ring_y[2]), T(PR.ring_y[3])}, fr[4] = {T(PR.ring_r[0]), T(PR.ring_r[1]), T(PR.ring_r[2]), T(PR.ring_r[3])};
      footP = ring_disc_point<T, 4>(R, addc(o, R, FT.jorigin[0], FT.jorigin[1], FT.jorigin[2]), fy, fr);
    }
  });
}

template <typename T> SD T clampv(T x, T lim) { return x > lim ? lim : (x < -lim ? -lim : x); }

// K2 inertia of a link, compile-time table entry LK: UI = false -> Bullet's box rule (diagonal), true -> the URDF tensor
// (inertia_urdf = ixx iyy izz ixy ixz iyz; products only where the URDF has them)
#define SOLO_LINK_TERMS_CE(UI_, LK_, R_, cw_, v_, kd_, I_, p_) do { \
    if constexpr (UI_) { \
      constexpr int od_ = ((LK_).inertia_urdf[3] != 0.0 ? 1 : 0) | ((LK_).inertia_urdf[4] != 0.0 ? 2 : 0) | ((LK_).inertia_urdf[5] != 0.0 ? 4 : 0); \
      link_terms<od_>(R_, cw_, T((LK_).mass), T((LK_).inertia_urdf[0]), T((LK_).inertia_urdf[1]), T((LK_).inertia_urdf[2]), v_, kd_, I_, p_, \
                      T((LK_).inertia_urdf[3]), T((LK_).inertia_urdf[4]), T((LK_).inertia_urdf[5])); \
    } else link_terms(R_, cw_, T((LK_).mass), T((LK_).inertia_box[0]), T((LK_).inertia_box[1]), T((LK_).inertia_box[2]), v_, kd_, I_, p_); \
  } while (0)

// what survives of a processed leg
template <typename T, int NJ> struct LegResp {
  SV<T> G[NJ];   // dq_k = y_k - G_k . w   (w = base delta-velocity);  qdd_k = qdd0_k - G_k . a_base
  T qdd0[NJ];
};

// per-lane hand-off between the phases of one sub-step (private memory)
template <typename T, int ROBOT> struct SubCtx {
  static constexpr int NJ = Robot<ROBOT>::NJ, NQ = Robot<ROBOT>::NQ;
  PhysState<T, NQ> ps;
  T tau[NQ];
  T sn[NQ], cs[NQ];
  M3<T> R0;
  V3<T> kneeP[4], footP[4], shP[4];
  T dist[NPRIM];
  int mask, nc, nlim_total, lsel;  // lsel: the joint-limit rows that are solved, bit 2j = lower limit of joint j, 2j+1 = upper
  int smask; T tmy;                // treadmill: primitives whose support point lies on the strip; the strip's centre line
  ABI<T> Ibase; SV<T> pbase;
  ABI<T> Ileg[4]; SV<T> pleg[4];   // team mode: per-leg contributions, summed by the leader
  T limpen[2 * NQ];                // team mode: the joints' margins to their limits while more than MAX_LIMITS compete
  LegResp<T, NJ> LR[4];
  SV<T> ub; T qds[NQ];
  SV<T> w; T y[4][3]; T lam_n[12];   // (lane mode: normal impulses of primitives 12..23)
  T lamp[NPRIM];                   // team mode: warm-start impulse cache, held in LDS across the sub-steps of a step
  // team mode, env logic (solorl_hip.hip step_team): the env's scalars in HBM field order (goal 2, potential, progress, goals,
  // env goals, reward-term sums 5, previous xy 2, treadmill centre line), its counters (timestep, contact mask, rng, snapshot
  // slot of a reset, done flag) and the commanded torques -- loaded, updated and stored by all 16 lanes
  T erec[14]; int irec[6]; T tau_base[NQ];
};

// Context handles: the phases take the context through a handle so that its address space survives the
// non-inlined call boundary -- private memory in lane mode (LDS is full there), LDS in team mode
// (a generic reference would turn every context access into a FLAT instruction into scratch).
template <typename T, int ROBOT> struct CtxPriv {
  SubCtx<T, ROBOT>* p;
  SD SubCtx<T, ROBOT>& get() const { return *p; }
};
#ifndef SOLO_HOST_SHIM
template <typename T, int ROBOT, size_t OFF> struct CtxLds {
  int col;
  SD SubCtx<T, ROBOT>& get() const {
    return *reinterpret_cast<SubCtx<T, ROBOT>*>(solo_smem + OFF + (size_t)col * sizeof(SubCtx<T, ROBOT>));
  }
};
#endif

// park one half-built row in LDS: f0 (force transmitted to the base), JL (raw joint torques),
// Y (leg response, base fixed), the contact point P and penetration for the finishing loop.
template <typename T, typename LDS>
SD void park_row(const LDS& lds, int slot, SV<T> f0, const T (&JL)[3], const T (&Y)[3], V3<T> P, T pen, T lam0, T mu,
                 int meta) {
  T core[ROW_CORE];
  core[0] = f0.a.x; core[1] = f0.a.y; core[2] = f0.a.z; core[3] = f0.l.x; core[4] = f0.l.y; core[5] = f0.l.z;
  core[6] = JL[0]; core[7] = JL[1]; core[8] = JL[2];
  core[9] = P.x; core[10] = P.y; core[11] = P.z; core[12] = pen; core[13] = lam0; core[14] = T(0);
  core[15] = Y[0]; core[16] = Y[1]; core[17] = Y[2]; core[18] = T(0); core[19] = T(0);
  lds.store_core(slot, core);
  using R = LDS;
  lds.A(slot, R::A_MU) = mu; lds.A(slot, R::A_LAM) = T(0);
  lds.A(slot, R::A_YOFF) = T(3 * ((meta >> 5) & 3)); lds.A(slot, R::A_META) = T(meta);
}

// ---------------------------------------------------------------- phase 1: collision detection
// start-of-step pose (K1, K6'): support points, contact mask, MAX_CONTACTS cap, row counts
// SINCOS_DONE: C.sn / C.cs were already filled (team mode: one joint per lane, substep_team)
template <typename T, int ROBOT, typename CH, bool UI, bool SINCOS_DONE = false>
SNI void phase_detect(CH ch, const PhysParams<T> pp) {
  SubCtx<T, ROBOT>& C = ch.get();
  using RB = Robot<ROBOT>;
  constexpr int NQ = RB::NQ;
  const PhysState<T, NQ>& st = C.ps;
  const M3<T> R0 = quat_to_mat(st.qx, st.qy, st.qz, st.qw);
  C.R0 = R0;
  T sn[NQ], cs[NQ];
#pragma unroll
  for (int j = 0; j < NQ; j++) {
    if constexpr (SINCOS_DONE) { sn[j] = C.sn[j]; cs[j] = C.cs[j]; }
    else { sincos_t(st.q[j], sn[j], cs[j]); C.sn[j] = sn[j]; C.cs[j] = cs[j]; }
  }
  T dist[NPRIM];
  int mask = 0, smask = 0;
  const T ty = st.pos.y - C.tmy;    // treadmill: a point P (relative to the base origin) is on the strip iff |ty + P.y| <= tm_hw
  static_for<12>([&](auto pc) {
    constexpr int p = decltype(pc)::value;
    constexpr solorl_prim_data PR = RB::MD.prims[p];
    static_assert(PR.link == 0 && PR.axis == -1, "base primitives are points");
    dist[p] = st.pos.z - pp.cmargin + R0.c0.z * T(PR.center[0]) + R0.c1.z * T(PR.center[1]) + R0.c2.z * T(PR.center[2]);
    if (dist[p] < T(PR.margin)) mask |= 1 << p;
    if (pp.tm_on() && fabs(ty + R0.c0.y * T(PR.center[0]) + R0.c1.y * T(PR.center[1]) + R0.c2.y * T(PR.center[2])) <= pp.tm_hw) smask |= 1 << p;
  });
  static_for<4>([&](auto lc) {
    constexpr int L = decltype(lc)::value;
    V3<T> kneeP, footP, shP = mk(T(0), T(0), T(0));
    leg_prim_points<T, ROBOT, L>(R0, sn, cs, kneeP, footP, shP);
    kneeP.z -= pp.cmargin; footP.z -= pp.cmargin; shP.z -= pp.cmargin;
    C.kneeP[L] = kneeP; C.footP[L] = footP; C.shP[L] = shP;
    if constexpr (RB::SHOULDER) {
      constexpr double ms_ = RB::MD.prims[20 + L].margin;
      dist[20 + L] = st.pos.z + shP.z;
      if (dist[20 + L] < T(ms_)) mask |= 1 << (20 + L);
      if (pp.tm_on() && fabs(ty + shP.y) <= pp.tm_hw) smask |= 1 << (20 + L);
    } else dist[20 + L] = T(1);
    dist[12 + 2 * L] = st.pos.z + kneeP.z; dist[13 + 2 * L] = st.pos.z + footP.z;
    constexpr double mk_ = RB::MD.prims[12 + 2 * L].margin, mf_ = RB::MD.prims[13 + 2 * L].margin;   // (constexpr: no run-time model loads)
    if (dist[12 + 2 * L] < T(mk_)) mask |= 1 << (12 + 2 * L);
    if (dist[13 + 2 * L] < T(mf_)) mask |= 1 << (13 + 2 * L);
    if (pp.tm_on() && fabs(ty + kneeP.y) <= pp.tm_hw) smask |= 1 << (12 + 2 * L);
    if (pp.tm_on() && fabs(ty + footP.y) <= pp.tm_hw) smask |= 1 << (13 + 2 * L);
  });
#pragma unroll
  for (int p = 0; p < NPRIM; p++) C.dist[p] = dist[p];
  // joint-limit rows (K5) come first: their count fixes the contact slots.  At most MAX_LIMITS are solved: the candidates
  // (joint at or beyond a limit) with the smallest margin, i.e. the most violated first (ties: lower joint, lower side).
  // Taking the first ones in joint order instead (rounds 1-2, two rows) let a third joint run radians past its limit
  // unopposed and be thrown back at erp * violation / dt once a slot freed up: robots were launched metres into the air.
  T lpen[2 * NQ];
  int lcand = 0;
#pragma unroll
  for (int j = 0; j < NQ; j++) {
    lpen[2 * j] = st.q[j] + pp.qlim; lpen[2 * j + 1] = pp.qlim - st.q[j];
    if (lpen[2 * j] <= T(LIMIT_WINDOW)) lcand |= 1 << (2 * j);
    if (lpen[2 * j + 1] <= T(LIMIT_WINDOW)) lcand |= 1 << (2 * j + 1);
  }
  int lsel = lcand;
  if (__builtin_popcount(lcand) > MAX_LIMITS) {
    lsel = 0;                          // (statically indexed: a rolled loop would put lpen[] in scratch for every lane, every sub-step)
    static_for<2 * NQ>([&](auto ac) {
      constexpr int a = decltype(ac)::value;
      int rank = 0;
      static_for<2 * NQ>([&](auto bc) {
        constexpr int b = decltype(bc)::value;
        if constexpr (b != a) rank += ((lcand >> b) & 1) && (lpen[b] < lpen[a] || (lpen[b] == lpen[a] && b < a));
      });
      if (((lcand >> a) & 1) && rank < MAX_LIMITS) lsel |= 1 << a;
    });
  }
  const int capc = contact_cap(__builtin_popcount(lsel));
  if (__builtin_popcount(mask) > capc) {   // keep the deepest (ties: lower primitive id)
    int keep = 0;
    static_for<NPRIM>([&](auto pc) {
      constexpr int p = decltype(pc)::value;
      int rank = 0;
      static_for<NPRIM>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        if constexpr (q != p) rank += ((mask >> q) & 1) && (dist[q] < dist[p] || (dist[q] == dist[p] && q < p));
      });
      if (((mask >> p) & 1) && rank < capc) keep |= 1 << p;
    });
    mask = keep;
  }
  C.mask = mask; C.nc = __builtin_popcount(mask); C.nlim_total = __builtin_popcount(lsel); C.lsel = lsel;
  C.smask = smask & mask;
  // base link terms start the articulated-inertia accumulation
  constexpr solorl_link_data B = RB::MD.links[0];
  static_assert(B.com[0] == 0.0 && B.com[1] == 0.0 && B.com[2] == 0.0, "base COM at its origin");
  RBI<T> Ib; SV<T> pb;
  const SV<T> vb{st.w, st.v};
  SOLO_LINK_TERMS_CE(UI, B, R0, mk(T(0), T(0), T(0)), vb, pp.damping, Ib, pb);
  C.Ibase = to_abi(Ib); C.pbase = pb;
}

// ---------------------------------------------------------------- phase 2 (x4): one leg
// FK + ABA passes 1-2, leg response (G, qdd0), parked limit and contact rows of this leg
template <typename T, int ROBOT, int L, typename LDS, typename CH, bool UI>
SNI void phase_leg(CH ch, const PhysParams<T> pp, const T* lam_prev, unsigned nstride, const LDS lds) {
  SubCtx<T, ROBOT>& C = ch.get();
  using RB = Robot<ROBOT>;
  constexpr int NJ = RB::NJ;
  constexpr int L0 = 1 + L * (NJ + 1);
  const T kd = pp.damping;
  const M3<T> R0 = C.R0;
  SV<T> vp{C.ps.w, C.ps.v};
  T sn[NJ], cs[NJ], qd[NJ], q[NJ], tau[NJ];
#pragma unroll
  for (int k = 0; k < NJ; k++) {
    sn[k] = C.sn[L * NJ + k]; cs[k] = C.cs[L * NJ + k]; qd[k] = C.ps.qd[L * NJ + k]; q[k] = C.ps.q[L * NJ + k];
    tau[k] = C.tau[L * NJ + k];
  }
  RBI<T> Ik[NJ]; SV<T> pk[NJ]; SV<T> Sk[NJ], ck[NJ];
  {
    M3<T> Rp = R0;
    V3<T> op = mk(T(0), T(0), T(0));
    static_for<NJ>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      constexpr solorl_link_data LK = RB::MD.links[L0 + k];
      constexpr int AX = LK.axis[0] != 0.0 ? 0 : 1;
      static_assert(LK.dof == L * NJ + k, "dof order");
      V3<T> o = addc(op, Rp, LK.jorigin[0], LK.jorigin[1], LK.jorigin[2]);
      V3<T> a = AX == 0 ? Rp.c0 : Rp.c1;
      M3<T> R = rot_axis<AX>(Rp, cs[k], sn[k]);
      SV<T> S{a, cross(o, a)};
      SV<T> vj = S * qd[k];
      SV<T> v = vp + vj;
      Sk[k] = S; ck[k] = crm(vp, vj);
      V3<T> cw = addc(o, R, LK.com[0], LK.com[1], LK.com[2]);
      SOLO_LINK_TERMS_CE(UI, LK, R, cw, v, kd, Ik[k], pk[k]);
      if constexpr (k == NJ - 1) {  // foot: fixed child of the last link, same axes and velocity
        constexpr solorl_link_data FT = RB::MD.links[L0 + NJ];
        V3<T> of = addc(o, R, FT.jorigin[0], FT.jorigin[1], FT.jorigin[2]);
        V3<T> cf = addc(of, R, FT.com[0], FT.com[1], FT.com[2]);
        RBI<T> If; SV<T> pf;
        SOLO_LINK_TERMS_CE(UI, FT, R, cf, v, kd, If, pf);
        add(Ik[k], If); pk[k] = pk[k] + pf;
      }
      Rp = R; op = o; vp = v;
    });
  }
  // pass 2: outermost joint first
  SV<T> Uk[NJ]; T Dk[NJ], uk[NJ];
  {
    ABI<T> IA = to_abi(Ik[NJ - 1]);
    SV<T> pA = pk[NJ - 1];
    static_for<NJ>([&](auto kc) {
      constexpr int k = NJ - 1 - decltype(kc)::value;
      SV<T> S = Sk[k];
      SV<T> U = mul(IA, S);
      T Dinv = T(1) / dot(S, U);
      T u = tau[k] - dot(S, pA);
      Uk[k] = U; Dk[k] = Dinv; uk[k] = u;
      rank1_sub(IA, U, Dinv);
      SV<T> pa = pA + mul(IA, ck[k]) + U * (u * Dinv);
      if constexpr (k > 0) { add(IA, Ik[k - 1]); pA = pk[k - 1] + pa; }
      else { ABI<T> Ib = C.Ibase; add(Ib, IA); C.Ibase = Ib; C.pbase = C.pbase + pa; }
    });
  }
  // leg response: qdd0 (base acceleration zero), G rows and Minv = M_LL^-1 columns from unit joint torques
  LegResp<T, NJ> LR;
  {
    SV<T> ap = zero6<T>();
    static_for<NJ>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      SV<T> apc = ap + ck[k];
      T qdd = (uk[k] - dot(Uk[k], apc)) * Dk[k];
      LR.qdd0[k] = qdd;
      if constexpr (k < NJ - 1) ap = fma6(Sk[k], qdd, apc);
    });
  }
  T Minv[3][3];   // Minv[row][col]
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) Minv[i][j] = T(0);
  static_for<NJ>([&](auto jc) {
    constexpr int j = decltype(jc)::value;   // unit torque at joint j
    T t[NJ];
    SV<T> f = Uk[j] * (-Dk[j]);
    static_for<NJ>([&](auto kc) {
      constexpr int k = NJ - 1 - decltype(kc)::value;
      if constexpr (k > j) t[k] = T(0);
      else if constexpr (k == j) t[k] = T(1);
      else { t[k] = dot(Sk[k], f); f = fma6(Uk[k], -t[k] * Dk[k], f); }
    });
    LR.G[j] = f * T(-1);
    SV<T> dv = zero6<T>();
    static_for<NJ>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      T yk;
      if constexpr (k == 0) yk = t[0] * Dk[0]; else yk = (t[k] - dot(Uk[k], dv)) * Dk[k];
      Minv[k][j] = yk;
      if constexpr (k < NJ - 1) dv = fma6(Sk[k], yk, dv);
    });
  });
  C.LR[L] = LR;
  // joint-limit rows of this leg (K5): the ones phase_detect selected, slots in joint order
  const int lsel = C.lsel;
  static_for<NJ>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
#pragma unroll
    for (int side = 0; side < 2; side++) {
      const int bit = 2 * (L * NJ + k) + side;
      if ((lsel >> bit) & 1) {
        T pen = side == 0 ? q[k] + pp.qlim : pp.qlim - q[k];
        T sg = side == 0 ? T(1) : T(-1);
        T JL[3] = {T(0), T(0), T(0)}, Y[3] = {Minv[0][k] * sg, Minv[1][k] * sg, Minv[2][k] * sg};
        JL[k] = sg;
        park_row(lds, __builtin_popcount(lsel & ((1 << bit) - 1)), LR.G[k] * (-sg), JL, Y, mk(T(0), T(0), T(0)), pen, T(0), T(0), (L << 5) | (3 << 8));
      }
    }
  });
  // contact rows of this leg's primitives: knee (chain depth NJ-1), foot (depth NJ) and, Solo12, the shoulder housing (depth 1)
  const int mask = C.mask, nlt = C.nlim_total, nc = C.nc;
  static_for<(RB::SHOULDER ? 3 : 2)>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    constexpr int p = i < 2 ? 12 + 2 * L + i : 20 + L;
    constexpr int DEPTH = i == 0 ? NJ - 1 : (i == 1 ? NJ : 1);
    constexpr solorl_prim_data PR = RB::MD.prims[p];
    if ((mask >> p) & 1) {
      const int cidx = __builtin_popcount(mask & ((1 << p) - 1));
      const int slot_n = nlt + cidx, slot_f = nlt + nc + 2 * cidx;
      const V3<T> P = i == 0 ? C.kneeP[L] : (i == 1 ? C.footP[L] : C.shP[L]);
      const T pen = C.dist[p] + pp.slop;
      const T lam0 = pp.warm * lam_prev[(unsigned)p * nstride];
      static_for<3>([&](auto dc) {
        constexpr int d = decltype(dc)::value;   // 0: normal (z), 1: friction x, 2: friction y
        V3<T> u = d == 0 ? mk(T(0), T(0), T(1)) : (d == 1 ? mk(T(1), T(0), T(0)) : mk(T(0), T(1), T(0)));
        SV<T> F{cross(P, u), u};
        T JL[3] = {T(0), T(0), T(0)}, Y[3];
        SV<T> f0 = F;
        static_for<DEPTH>([&](auto kc) {
          constexpr int k = decltype(kc)::value;
          JL[k] = dot(Sk[k], F);
          f0 = fma6(LR.G[k], -JL[k], f0);
        });
#pragma unroll
        for (int r = 0; r < 3; r++) Y[r] = Minv[r][0] * JL[0] + Minv[r][1] * JL[1] + Minv[r][2] * JL[2];
        park_row(lds, d == 0 ? slot_n : slot_f + (d - 1), f0, JL, Y, P, pen, d == 0 ? lam0 : T(0),
                 T(PR.friction) * (((C.smask >> p) & 1) ? pp.tm_mu : T(1)), (d == 0 ? 0 : (slot_n | 128)) | (L << 5) | (d << 8));
      });
    }
  });
}

#ifndef SOLO_HOST_SHIM
#ifndef SOLO_LEG_QUAD
#define SOLO_LEG_QUAD 1          // 0: one lane per leg (rounds 1-3); A/B builds
#endif
// value of sub-lane S of this lane's quad (lanes 4q .. 4q+3): one DPP move (two for a double)
template <int S, typename T> SD T quad_bcast(T x) {
  constexpr int ctrl = S | (S << 2) | (S << 4) | (S << 6);           // quad_perm:[S,S,S,S]
  if constexpr (sizeof(T) == 4) return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), ctrl, 0xF, 0xF, true));
  else {
    const long long v = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_mov_dpp((int)(v & 0xFFFFFFFFll), ctrl, 0xF, 0xF, true), hi = __builtin_amdgcn_mov_dpp((int)(v >> 32), ctrl, 0xF, 0xF, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
  }
}
// ---------------------------------------------------------------- phase 2, team mode: the four legs in parallel
// Lane L (0..3) of a team processes leg L.  The legs are mirror images, so the code is the same and
// only constants differ: they are picked per lane from the four legs' compile-time values
// (identical values fold away).  Results go to the shared (LDS) context: per-leg articulated inertia /
// bias contributions (summed by the leader), leg response, parked rows.
// The legs are mirror images (L = 0 FL, 1 FR, 2 HL, 3 HR): a per-leg constant is +-v with the sign following
// the front/hind (sx) and/or left/right (sy) reflection, so it is v times a per-lane sign instead of a lookup
// (a 4-entry table made every use a dependent global load).  Irregular constants fall back to a blend.
template <typename T> struct LegSign { T sx, sy, sxy, m1, m2, m3; };
template <typename T> SD LegSign<T> leg_sign(int L) {
  LegSign<T> g;
  g.sx = (L & 2) ? T(-1) : T(1); g.sy = (L & 1) ? T(-1) : T(1); g.sxy = g.sx * g.sy;
  g.m1 = L == 1 ? T(1) : T(0); g.m2 = L == 2 ? T(1) : T(0); g.m3 = L == 3 ? T(1) : T(0);
  return g;
}
// SEL4: the four values must be constant expressions (a constexpr local forces compile-time evaluation --
// reading the model through the `MD` reference at run time is a load chain through a global pointer)
#define SEL4(T, g, A, B, C, D) ([&]() -> T { constexpr double a_ = (A), b_ = (B), c_ = (C), d_ = (D); \
    if constexpr (a_ == b_ && b_ == c_ && c_ == d_) return T(a_); \
    else if constexpr (b_ == -a_ && c_ == a_ && d_ == -a_) return T(a_) * (g).sy; \
    else if constexpr (b_ == a_ && c_ == -a_ && d_ == -a_) return T(a_) * (g).sx; \
    else if constexpr (b_ == -a_ && c_ == -a_ && d_ == a_) return T(a_) * (g).sxy; \
    else return T(a_) + T(b_ - a_) * (g).m1 + T(c_ - a_) * (g).m2 + T(d_ - a_) * (g).m3; }())
template <typename T, int ROBOT, typename LDS, typename CH, bool UI>
SNI_SCALAR void phase_leg_rt(CH ch, const PhysParams<T>& pp, const T* lam_prev, unsigned nstride, const LDS lds, int L, int sub) {   // (pp by reference:
                                                                     // ten more live argument registers made this phase spill 101 VGPRs)
  // Round 4 (SOLO_LEG_QUAD): leg L runs on the FOUR lanes of quad L (sub = lane & 3) instead of one.  Kinematics, link terms, the
  // articulated-body pass and qdd0 are computed redundantly by the four (SIMT: the same instructions, no extra time); what used to be
  // loops over independent pieces is split across them -- the NJ columns of the leg's response (G_j, Minv[:, j]: column `sub`, then
  // exchanged with quad_perm broadcasts) and the three rows of every touching primitive (direction `sub`).  Shared results are stored
  // by sub-lane 0.  (Link terms are NOT split: selecting a link's frame, COM, velocity and constants per lane costs half of what it saves.)
  SubCtx<T, ROBOT>& C = ch.get();
  using RB = Robot<ROBOT>;
  constexpr int NJ = RB::NJ;
  constexpr int ST = NJ + 1;   // links per leg
  // every field of pp this phase uses, read HERE: pp arrives by reference (a FLAT load from the kernel's stack, ~700 cycles), and read where it is
  // used -- inside the blocks of the contact rows -- each touching primitive waited for its own round trip
  const T kd = pp.damping, pp_qlim = pp.qlim, pp_slop = pp.slop, pp_warm = pp.warm, pp_tm_mu = pp.tm_mu;
  const M3<T> R0 = C.R0;
  SV<T> vp{C.ps.w, C.ps.v};
  T sn[NJ], cs[NJ], qd[NJ], q[NJ], tau[NJ];
#pragma unroll
  for (int k = 0; k < NJ; k++) {
    sn[k] = C.sn[L * NJ + k]; cs[k] = C.cs[L * NJ + k]; qd[k] = C.ps.qd[L * NJ + k]; q[k] = C.ps.q[L * NJ + k];
    tau[k] = C.tau[L * NJ + k];
  }
  // value of link field f of joint k (or the foot, k = NJ) for this lane's leg
  const LegSign<T> lsg = leg_sign<T>(L);
#define LEGC(k, f) SEL4(T, lsg, RB::MD.links[1 + 0 * ST + (k)].f, RB::MD.links[1 + 1 * ST + (k)].f, \
                        RB::MD.links[1 + 2 * ST + (k)].f, RB::MD.links[1 + 3 * ST + (k)].f)
  RBI<T> Ik[NJ]; SV<T> pk[NJ]; SV<T> Sk[NJ], ck[NJ];
  {
    M3<T> Rp = R0;
    V3<T> op = mk(T(0), T(0), T(0));
    static_for<NJ>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      constexpr int AX = RB::MD.links[1 + k].axis[0] != 0.0 ? 0 : 1;   // same for all legs
      V3<T> o = op + mul(Rp, mk(LEGC(k, jorigin[0]), LEGC(k, jorigin[1]), LEGC(k, jorigin[2])));
      V3<T> a = AX == 0 ? Rp.c0 : Rp.c1;
      M3<T> R = rot_axis<AX>(Rp, cs[k], sn[k]);
      SV<T> S{a, cross(o, a)};
      SV<T> vj = S * qd[k];
      SV<T> v = vp + vj;
      Sk[k] = S; ck[k] = crm(vp, vj);
      V3<T> cw = o + mul(R, mk(LEGC(k, com[0]), LEGC(k, com[1]), LEGC(k, com[2])));
      // K2: box rule, or the URDF tensor with its products of inertia (their signs follow the legs' mirror symmetry)
#define LEG_OD(k_) (((RB::MD.links[1 + (k_)].inertia_urdf[3] != 0.0) ? 1 : 0) | ((RB::MD.links[1 + (k_)].inertia_urdf[4] != 0.0) ? 2 : 0) | \
                    ((RB::MD.links[1 + (k_)].inertia_urdf[5] != 0.0) ? 4 : 0))
      if constexpr (UI) link_terms<LEG_OD(k)>(R, cw, LEGC(k, mass), LEGC(k, inertia_urdf[0]), LEGC(k, inertia_urdf[1]), LEGC(k, inertia_urdf[2]), v, kd, Ik[k], pk[k],
                                              LEGC(k, inertia_urdf[3]), LEGC(k, inertia_urdf[4]), LEGC(k, inertia_urdf[5]));
      else link_terms(R, cw, LEGC(k, mass), LEGC(k, inertia_box[0]), LEGC(k, inertia_box[1]), LEGC(k, inertia_box[2]), v, kd, Ik[k], pk[k]);
      if constexpr (k == NJ - 1) {  // foot: fixed child of the last link, same axes and velocity
        V3<T> of = o + mul(R, mk(LEGC(NJ, jorigin[0]), LEGC(NJ, jorigin[1]), LEGC(NJ, jorigin[2])));
        V3<T> cf = of + mul(R, mk(LEGC(NJ, com[0]), LEGC(NJ, com[1]), LEGC(NJ, com[2])));
        RBI<T> If; SV<T> pf;
        if constexpr (UI) link_terms<LEG_OD(NJ)>(R, cf, LEGC(NJ, mass), LEGC(NJ, inertia_urdf[0]), LEGC(NJ, inertia_urdf[1]), LEGC(NJ, inertia_urdf[2]), v, kd, If, pf,
                                                 LEGC(NJ, inertia_urdf[3]), LEGC(NJ, inertia_urdf[4]), LEGC(NJ, inertia_urdf[5]));
        else link_terms(R, cf, LEGC(NJ, mass), LEGC(NJ, inertia_box[0]), LEGC(NJ, inertia_box[1]), LEGC(NJ, inertia_box[2]), v, kd, If, pf);
#undef LEG_OD
        add(Ik[k], If); pk[k] = pk[k] + pf;
      }
      Rp = R; op = o; vp = v;
    });
  }
#undef LEGC
  // pass 2: outermost joint first
  SV<T> Uk[NJ]; T Dk[NJ], uk[NJ];
  {
    ABI<T> IA = to_abi(Ik[NJ - 1]);
    SV<T> pA = pk[NJ - 1];
    static_for<NJ>([&](auto kc) {
      constexpr int k = NJ - 1 - decltype(kc)::value;
      SV<T> S = Sk[k];
      SV<T> U = mul(IA, S);
      T Dinv = rcp_fast(dot(S, U));
      T u = tau[k] - dot(S, pA);
      Uk[k] = U; Dk[k] = Dinv; uk[k] = u;
      rank1_sub(IA, U, Dinv);
      SV<T> pa = pA + mul(IA, ck[k]) + U * (u * Dinv);
      if constexpr (k > 0) { add(IA, Ik[k - 1]); pA = pk[k - 1] + pa; }
      else if (sub == 0) { C.Ileg[L] = IA; C.pleg[L] = pa; }
    });
  }
  LegResp<T, NJ> LR;
  {
    SV<T> ap = zero6<T>();
    static_for<NJ>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      SV<T> apc = ap + ck[k];
      T qdd = (uk[k] - dot(Uk[k], apc)) * Dk[k];
      LR.qdd0[k] = qdd;
      if constexpr (k < NJ - 1) ap = fma6(Sk[k], qdd, apc);
    });
  }
  T Minv[3][3];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) Minv[i][j] = T(0);
#if SOLO_LEG_QUAD
  {
    // column jj = min(sub, NJ - 1) of the response, with the column index a run-time value: the compile-time version below with its
    // `if constexpr (k > j)` cases turned into selects
    const int jj = sub < NJ - 1 ? sub : NJ - 1;
    SV<T> Uj = Uk[0]; T Dj = Dk[0];
#pragma unroll
    for (int k = 1; k < NJ; k++) {
      const bool m = jj == k;
      Uj.a.x = m ? Uk[k].a.x : Uj.a.x; Uj.a.y = m ? Uk[k].a.y : Uj.a.y; Uj.a.z = m ? Uk[k].a.z : Uj.a.z;
      Uj.l.x = m ? Uk[k].l.x : Uj.l.x; Uj.l.y = m ? Uk[k].l.y : Uj.l.y; Uj.l.z = m ? Uk[k].l.z : Uj.l.z;
      Dj = m ? Dk[k] : Dj;
    }
    T tt[NJ];
    SV<T> f = Uj * (-Dj);
    static_for<NJ>([&](auto kc) {
      constexpr int k = NJ - 1 - decltype(kc)::value;
      if constexpr (k == NJ - 1) tt[k] = jj == k ? T(1) : T(0);           // (k > jj: 0; k < jj cannot happen for the outermost joint)
      else {
        const T cand = dot(Sk[k], f);
        const bool below = k < jj;
        tt[k] = below ? cand : (k == jj ? T(1) : T(0));
        const SV<T> fn = fma6(Uk[k], -cand * Dk[k], f);
        f.a.x = below ? fn.a.x : f.a.x; f.a.y = below ? fn.a.y : f.a.y; f.a.z = below ? fn.a.z : f.a.z;
        f.l.x = below ? fn.l.x : f.l.x; f.l.y = below ? fn.l.y : f.l.y; f.l.z = below ? fn.l.z : f.l.z;
      }
    });
    const SV<T> Gm = f * T(-1);
    T col[NJ];
    SV<T> dv = zero6<T>();
    static_for<NJ>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      T yk;
      if constexpr (k == 0) yk = tt[0] * Dk[0]; else yk = (tt[k] - dot(Uk[k], dv)) * Dk[k];
      col[k] = yk;
      if constexpr (k < NJ - 1) dv = fma6(Sk[k], yk, dv);
    });
    // every sub-lane gets every column: quad_perm broadcast from sub-lane j
    static_for<NJ>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      LR.G[j].a.x = quad_bcast<j>(Gm.a.x); LR.G[j].a.y = quad_bcast<j>(Gm.a.y); LR.G[j].a.z = quad_bcast<j>(Gm.a.z);
      LR.G[j].l.x = quad_bcast<j>(Gm.l.x); LR.G[j].l.y = quad_bcast<j>(Gm.l.y); LR.G[j].l.z = quad_bcast<j>(Gm.l.z);
#pragma unroll
      for (int k = 0; k < NJ; k++) Minv[k][j] = quad_bcast<j>(col[k]);
    });
  }
  if (sub == 0) C.LR[L] = LR;
#else
  static_for<NJ>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    T t[NJ];
    SV<T> f = Uk[j] * (-Dk[j]);
    static_for<NJ>([&](auto kc) {
      constexpr int k = NJ - 1 - decltype(kc)::value;
      if constexpr (k > j) t[k] = T(0);
      else if constexpr (k == j) t[k] = T(1);
      else { t[k] = dot(Sk[k], f); f = fma6(Uk[k], -t[k] * Dk[k], f); }
    });
    LR.G[j] = f * T(-1);
    SV<T> dv = zero6<T>();
    static_for<NJ>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      T yk;
      if constexpr (k == 0) yk = t[0] * Dk[0]; else yk = (t[k] - dot(Uk[k], dv)) * Dk[k];
      Minv[k][j] = yk;
      if constexpr (k < NJ - 1) dv = fma6(Sk[k], yk, dv);
    });
  });
  C.LR[L] = LR;
#endif
  // joint-limit rows of this leg: the ones the collision phase selected (C.lsel), slots in joint order
  const int lsel = C.lsel;
  if (sub == 0 && ((lsel >> (2 * NJ * L)) & ((1 << (2 * NJ)) - 1))) {
    static_for<NJ>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
#pragma unroll
      for (int side = 0; side < 2; side++) {
        const int bit = 2 * (L * NJ + k) + side;
        if ((lsel >> bit) & 1) {
          T pen = side == 0 ? q[k] + pp_qlim : pp_qlim - q[k];
          T sg = side == 0 ? T(1) : T(-1);
          T JL[3] = {T(0), T(0), T(0)}, Y[3] = {Minv[0][k] * sg, Minv[1][k] * sg, Minv[2][k] * sg};
          JL[k] = sg;
          park_row(lds, __builtin_popcount(lsel & ((1 << bit) - 1)), LR.G[k] * (-sg), JL, Y, mk(T(0), T(0), T(0)), pen, T(0), T(0), (L << 5) | (3 << 8));
        }
      }
    });
  }
  // contact rows of this leg's primitives: knee (chain depth NJ-1), foot (depth NJ) and, Solo12, the shoulder housing (depth 1)
  const int mask = C.mask, nlt = C.nlim_total, nc = C.nc;
  static_for<(RB::SHOULDER ? 3 : 2)>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    constexpr int DEPTH = i == 0 ? NJ - 1 : (i == 1 ? NJ : 1);
    constexpr int P0 = i < 2 ? 12 + i : 20, PS = i < 2 ? 2 : 1;        // this primitive of leg L: P0 + PS * L
    const int p = P0 + PS * L;
    if ((mask >> p) & 1) {
      const int cidx = __builtin_popcount(mask & ((1 << p) - 1));
      const int slot_n = nlt + cidx, slot_f = nlt + nc + 2 * cidx;
      const V3<T> P = i == 0 ? C.kneeP[L] : (i == 1 ? C.footP[L] : C.shP[L]);
      const T pen = C.dist[p] + pp_slop;
      const T lam0 = pp_warm * C.lamp[p];
      const T fric = SEL4(T, lsg, RB::MD.prims[P0].friction, RB::MD.prims[P0 + PS].friction, RB::MD.prims[P0 + 2 * PS].friction,
                          RB::MD.prims[P0 + 3 * PS].friction) * (((C.smask >> p) & 1) ? pp_tm_mu : T(1));
#if SOLO_LEG_QUAD
      if (sub < 3) {             // direction d = sub: 0 normal (z), 1 friction x, 2 friction y -- the three rows of the contact side by side
        const int d = sub;
        const V3<T> u = mk(d == 1 ? T(1) : T(0), d == 2 ? T(1) : T(0), d == 0 ? T(1) : T(0));
        SV<T> F{cross(P, u), u};
        T JL[3] = {T(0), T(0), T(0)}, Y[3];
        SV<T> f0 = F;
        static_for<DEPTH>([&](auto kc) {
          constexpr int k = decltype(kc)::value;
          JL[k] = dot(Sk[k], F);
          f0 = fma6(LR.G[k], -JL[k], f0);
        });
#pragma unroll
        for (int r = 0; r < 3; r++) Y[r] = Minv[r][0] * JL[0] + Minv[r][1] * JL[1] + Minv[r][2] * JL[2];
        park_row(lds, d == 0 ? slot_n : slot_f + (d - 1), f0, JL, Y, P, pen, d == 0 ? lam0 : T(0), fric,
                 (d == 0 ? 0 : (slot_n | 128)) | (L << 5) | (d << 8));
      }
#else
      static_for<3>([&](auto dc) {
        constexpr int d = decltype(dc)::value;
        V3<T> u = d == 0 ? mk(T(0), T(0), T(1)) : (d == 1 ? mk(T(1), T(0), T(0)) : mk(T(0), T(1), T(0)));
        SV<T> F{cross(P, u), u};
        T JL[3] = {T(0), T(0), T(0)}, Y[3];
        SV<T> f0 = F;
        static_for<DEPTH>([&](auto kc) {
          constexpr int k = decltype(kc)::value;
          JL[k] = dot(Sk[k], F);
          f0 = fma6(LR.G[k], -JL[k], f0);
        });
#pragma unroll
        for (int r = 0; r < 3; r++) Y[r] = Minv[r][0] * JL[0] + Minv[r][1] * JL[1] + Minv[r][2] * JL[2];
        park_row(lds, d == 0 ? slot_n : slot_f + (d - 1), f0, JL, Y, P, pen, d == 0 ? lam0 : T(0), fric,
                 (d == 0 ? 0 : (slot_n | 128)) | (L << 5) | (d << 8));
      });
#endif
    }
  });
}
#endif  // !SOLO_HOST_SHIM

// ---------------------------------------------------------------- phase 3: base solve + finish rows
// base_solve: parks the base primitives' rows, inverts the base articulated inertia, computes the
// unconstrained velocities u* = u + dt*udot (C.ub, C.qds) and returns Lam and the padded leg rates.
template <typename T, int ROBOT, typename LDS, bool TEAMQ = false, typename QOUT>
SD void base_solve(SubCtx<T, ROBOT>& C, const PhysParams<T> pp, const T* lam_prev, unsigned nstride, const LDS& lds,
                   Sym6<T>& Lam, QOUT& qsl) {   // QOUT: T[4][3] padded leg rates, or (TEAMQ) SV<T> receiving a0
  using RB = Robot<ROBOT>;
  constexpr int NJ = RB::NJ;
  const T dt = pp.dt;
  const int mask = C.mask, nlt = C.nlim_total, nc = C.nc;
  const int nrows = nlt + 3 * nc;
  if (!TEAMQ && (mask & 0xFFF)) {   // base contacts: f0 = F, no leg part (team mode: parked by phase_front_team)
    const M3<T> R0 = C.R0;
    static_for<12>([&](auto pc) {
      constexpr int p = decltype(pc)::value;
      constexpr solorl_prim_data PR = RB::MD.prims[p];
      if ((mask >> p) & 1) {
        const int cidx = __builtin_popcount(mask & ((1 << p) - 1));
        const int sn_ = nlt + cidx, sf_ = nlt + nc + 2 * cidx;
        V3<T> P = mul(R0, mk(T(PR.center[0]), T(PR.center[1]), T(PR.center[2])));
        P.z -= pp.cmargin;
        const T lam0 = pp.warm * (TEAMQ ? C.lamp[p] : lam_prev[(unsigned)p * nstride]);
        const T Z[3] = {T(0), T(0), T(0)};
        static_for<3>([&](auto dc) {
          constexpr int d = decltype(dc)::value;
          V3<T> u = d == 0 ? mk(T(0), T(0), T(1)) : (d == 1 ? mk(T(1), T(0), T(0)) : mk(T(0), T(1), T(0)));
          park_row(lds, d == 0 ? sn_ : sf_ + (d - 1), SV<T>{cross(P, u), u}, Z, Z, P, C.dist[p] + pp.slop,
                   d == 0 ? lam0 : T(0), T(PR.friction) * (((C.smask >> p) & 1) ? pp.tm_mu : T(1)), (d == 0 ? 0 : (sn_ | 128)) | (d << 8));
        });
      }
    });
  }
  if (nrows == 0) {   // the branch-free PGS sweep touches row 0 of every lane: make it a null row
    const T Z[3] = {T(0), T(0), T(0)};
    park_row(lds, 0, zero6<T>(), Z, Z, mk(T(0), T(0), T(0)), T(0), T(0), T(0), 0);
  }
  // base acceleration (gravity via the accelerating-frame trick), u* = u + dt*udot
  Lam = spd_inverse_block(C.Ibase);
  const SV<T> a0 = mul(Lam, C.pbase) * T(-1);
  if constexpr (TEAMQ) qsl = a0;
  else {
    static_for<4>([&](auto lc) {
      constexpr int L = decltype(lc)::value;
#pragma unroll
      for (int k = 0; k < NJ; k++) {
        T qdd = C.LR[L].qdd0[k] - dot(C.LR[L].G[k], a0);
        C.qds[L * NJ + k] = clampv(C.ps.qd[L * NJ + k] + dt * qdd, pp.vmax);
      }
    });
#pragma unroll
    for (int L = 0; L < 4; L++)
#pragma unroll
      for (int k = 0; k < 3; k++) qsl[L][k] = k < NJ ? C.qds[L * NJ + k] : T(0);
  }
  const V3<T> bw = C.ps.w, bv = C.ps.v;
  V3<T> vdot = a0.l + cross(bw, bv); vdot.z -= pp.gravity;
  SV<T> ub;
  ub.a = mk(clampv(bw.x + dt * a0.a.x, pp.vmax), clampv(bw.y + dt * a0.a.y, pp.vmax), clampv(bw.z + dt * a0.a.z, pp.vmax));
  ub.l = mk(clampv(bv.x + dt * vdot.x, pp.vmax), clampv(bv.y + dt * vdot.y, pp.vmax), clampv(bv.z + dt * vdot.z, pp.vmax));
  C.ub = ub;
}

// finish one parked row: W = Lam f0, 1/diagonal, right-hand side (speculative / ERP, or friction)
template <typename T>
SD void finish_row(const T (&c)[ROW_CORE], int meta, const Sym6<T>& Lam, const SV<T>& ub, T q0, T q1, T q2,
                   const PhysParams<T> pp, SV<T>& W, T& rhs, T& dinv) {
  const int dir = (meta >> 8) & 3;
  const SV<T> f0{{c[0], c[1], c[2]}, {c[3], c[4], c[5]}};
  const V3<T> P = mk(c[9], c[10], c[11]);
  const V3<T> u = mk(dir == 1 ? T(1) : T(0), dir == 2 ? T(1) : T(0), dir == 0 ? T(1) : T(0));
  const T rel = dot(cross(P, u), ub.a) + dot(u, ub.l) + c[6] * q0 + c[7] * q1 + c[8] * q2;
  W = mul(Lam, f0);
  const T denom = dot(f0, W) + c[6] * c[15] + c[7] * c[16] + c[8] * c[17];
  dinv = rcp_fast(denom);
  if (meta & 128) rhs = -rel * dinv;
  else {
    const T pen = c[12];
    T pos = T(0), vel = -rel;
    if (pen > T(0)) vel -= pen * pp.inv_dt; else pos = -pen * (dir == 3 ? pp.erp : pp.cerp) * pp.inv_dt;   // joint limit: m_erp, contact: m_erp2
    rhs = (pos + vel) * dinv;
  }
}

template <typename T, int ROBOT, typename LDS, typename CH>
SNI void phase_base(CH ch, const PhysParams<T> pp, const T* lam_prev, unsigned nstride, const LDS lds) {
  SubCtx<T, ROBOT>& C = ch.get();
  constexpr int LN = LDS::LANES;
  Sym6<T> Lam; T qsl[4][3];
  base_solve<T, ROBOT, LDS>(C, pp, lam_prev, nstride, lds, Lam, qsl);
  const SV<T> ub = C.ub;
  const int nrows = C.nlim_total + 3 * C.nc;
  T* const yl = lds.y();
#pragma unroll
  for (int k = 0; k < 12; k++) yl[k * LN] = T(0);
  // finish the parked rows (W = Lam f0, diagonal, right-hand side, warm start)
  SV<T> w = zero6<T>();
  for (int r = 0; __any(r < nrows); r++) {
    if (r < nrows) {
      T c[ROW_CORE];
      lds.load_core(r, c);
      const int meta = (int)lds.A(r, LDS::A_META);
      const int leg = (meta >> 5) & 3;
      T* yp = yl + leg * 3 * LN;
      const T q0 = leg == 0 ? qsl[0][0] : leg == 1 ? qsl[1][0] : leg == 2 ? qsl[2][0] : qsl[3][0];
      const T q1 = leg == 0 ? qsl[0][1] : leg == 1 ? qsl[1][1] : leg == 2 ? qsl[2][1] : qsl[3][1];
      const T q2 = leg == 0 ? qsl[0][2] : leg == 1 ? qsl[1][2] : leg == 2 ? qsl[2][2] : qsl[3][2];
      SV<T> W; T rhs, dinv;
      finish_row(c, meta, Lam, ub, q0, q1, q2, pp, W, rhs, dinv);
      const T lam0 = c[13];
      c[9] = W.a.x; c[10] = W.a.y; c[11] = W.a.z; c[12] = W.l.x; c[13] = W.l.y; c[14] = W.l.z; c[18] = rhs; c[19] = dinv;
      lds.store_core(r, c);
      lds.A(r, LDS::A_LAM) = lam0;
      w = fma6(W, lam0, w);   // warm start: dV += M^-1 J^T lam0
      yp[0] += c[15] * lam0; yp[LN] += c[16] * lam0; yp[2 * LN] += c[17] * lam0;
    }
  }
  C.w = w;
}

// ---------------------------------------------------------------- phase 4: projected Gauss-Seidel (K7)
// Row order = LDS order: [limit rows | normals by primitive id | friction pairs by primitive id].
// Row-index driven: in sweep step r every lane updates ITS row r, so a sweep costs the wave's
// maximum row count (not the union of all primitives).  Software pipelined: row r+1's record,
// impulse, friction coefficient and leg id are fetched while row r is computed; a friction row's
// parent is found arithmetically (parent = nlim + (r - nlim - nc)/2), so there is no dependent
// LDS chain.  Accumulators: base delta-velocity w in registers, leg delta-rates y in LDS.
// CONE (solorl_config friction_model = cone): the two friction rows of a contact -- consecutive rows rfric + 2p, rfric + 2p + 1 -- are
// solved together as Bullet's resolveConeFrictionConstraintRows does [K]: the first row of a pair only computes its unclamped sum and
// parks what its update needs (Pend), the second computes its own against the SAME accumulators, the pair is projected radially onto
// the disc of radius mu * normal impulse, both deltas are applied, and the K7 residual entry is the sum of the two velocity changes.
template <typename T, int ROBOT, typename LDS, typename CH, bool CONE = false>
SNI void phase_pgs(CH ch, int iterations, T resid_thr, const LDS lds) {
  SubCtx<T, ROBOT>& C = ch.get();
  using R_ = LDS;
  using Chunk = typename R_::Chunk;
  constexpr int NCH = R_::NCH, LN = R_::LANES;
  const int nlt = C.nlim_total, nc = C.nc;
  const int nrows = nlt + 3 * nc, rfric = nlt + nc;
  SV<T> w = C.w;
  const Chunk* const corev = lds.core();   // row r, chunk c at corev[(r*NCH + c)*LN]
  T* const auxv = lds.aux();               // row r, field f at auxv[(r*4 + f)*LN]
  T* const yl = lds.y();                   // accumulator k at yl[k*LN]
  int wmax = nrows;   // wave-uniform maximum row count
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) { int t = __shfl_xor(wmax, o); wmax = t > wmax ? t : wmax; }
  const int last = nrows > 0 ? nrows - 1 : 0;   // lanes past their last row re-process it with delta forced to 0

  // The kernel's duration is the SLOWEST wave, i.e. the sequential sweep of the heaviest env, so the
  // loop is organised for LATENCY: row r+1's record, aux and leg accumulators are fetched while row
  // r is computed; if r+1 works on the same leg (or has row r as friction parent) the freshly
  // computed register values are forwarded instead of waiting for an LDS store->load round trip.
  struct Row { T c[ROW_CORE]; T mu, lam, lamp, y0, y1, y2; int yoff, par; };
  struct Pend { T W[9], dinv, lam, sum; int row; } pend;      // CONE: the first friction row of a pair, waiting for the second
  pend.dinv = T(1); pend.lam = T(0); pend.sum = T(0); pend.row = 0;
#pragma unroll
  for (int k = 0; k < 9; k++) pend.W[k] = T(0);
  bool conv = false, viol = false;
  auto fetch = [&](int rr, Row& R) {
    const Chunk* p = corev + __mul24(rr, NCH * LN);
#pragma unroll
    for (int c = 0; c < NCH; c++) {
      Chunk ch = p[c * LN];
      if constexpr (sizeof(T) == 4) { R.c[4 * c] = ch.x; R.c[4 * c + 1] = ch.y; R.c[4 * c + 2] = ch.z; R.c[4 * c + 3] = ch.w; }
      else { R.c[2 * c] = ch.x; R.c[2 * c + 1] = ch.y; }
    }
    const T* a = auxv + __mul24(rr, 4 * LN);
    R.mu = a[R_::A_MU * LN]; R.lam = a[R_::A_LAM * LN]; R.yoff = (int)a[R_::A_YOFF * LN] * LN;
    int pi = nlt + ((rr - rfric) >> 1);                  // friction parent (clamped: harmless for other rows)
    R.par = pi < 0 ? 0 : pi;
    R.lamp = auxv[__mul24(R.par, 4 * LN) + R_::A_LAM * LN];
    R.y0 = yl[R.yoff]; R.y1 = yl[R.yoff + LN]; R.y2 = yl[R.yoff + 2 * LN];
  };
  // update row r (in R); N receives row r+1, fetched before the arithmetic, patched after it
  auto step = [&](int r, Row& R, Row& N) {
    const int rn = r + 1 < nrows ? r + 1 : last;
    fetch(rn, N);
    T jdv = R.c[0] * w.a.x + R.c[1] * w.a.y + R.c[2] * w.a.z + R.c[3] * w.l.x + R.c[4] * w.l.y + R.c[5] * w.l.z;
    jdv += R.c[6] * R.y0 + R.c[7] * R.y1 + R.c[8] * R.y2;
    const bool fr = r >= rfric, live = r < nrows;
    const T hi = fr ? R.mu * R.lamp : T(1e30);
    const T lo = fr ? -hi : T(0);
    T sum = R.lam + (R.c[18] - jdv * R.c[19]);
    const int rw = live ? r : last;
    T delta, y0, y1, y2;
    const bool active = live && !conv;                    // (a converged lane's solve is over: its rows stay as they are)
    bool paired = false;
    if constexpr (CONE) paired = fr && active;
    if (paired) {
      if (((r - rfric) & 1) == 0) {                        // first row of the pair: nothing is applied yet
        pend.sum = sum; pend.lam = R.lam; pend.dinv = R.c[19]; pend.row = rw;
#pragma unroll
        for (int k = 0; k < 9; k++) pend.W[k] = R.c[9 + k];
        sum = R.lam; delta = T(0);
        y0 = R.y0; y1 = R.y1; y2 = R.y2;
      } else {
        T sa = pend.sum, sb = sum;
        const T s2 = sa * sa + sb * sb;
        if (s2 >= hi * hi) { const T sc = hi * rsqrt_fast(s2 > T(1e-30) ? s2 : T(1e-30)); sa *= sc; sb *= sc; }
        const T da = sa - pend.lam;
        sum = sb; delta = sb - R.lam;
        viol = viol || (fabs(da / pend.dinv + delta / R.c[19]) > resid_thr);      // K7: the pair's entry is the SUM of the two velocity changes
        w.a.x += pend.W[0] * da + R.c[9] * delta; w.a.y += pend.W[1] * da + R.c[10] * delta; w.a.z += pend.W[2] * da + R.c[11] * delta;
        w.l.x += pend.W[3] * da + R.c[12] * delta; w.l.y += pend.W[4] * da + R.c[13] * delta; w.l.z += pend.W[5] * da + R.c[14] * delta;
        y0 = R.y0 + pend.W[6] * da + R.c[15] * delta; y1 = R.y1 + pend.W[7] * da + R.c[16] * delta; y2 = R.y2 + pend.W[8] * da + R.c[17] * delta;
        auxv[__mul24(pend.row, 4 * LN) + R_::A_LAM * LN] = sa;
      }
    } else {
      sum = sum < lo ? lo : (sum > hi ? hi : sum);
      sum = active ? sum : R.lam;
      delta = sum - R.lam;
      const T dvel = fabs(delta) / R.c[19];                  // K7 residual: |delta impulse| / jacDiagABInv
      viol = viol || (dvel > resid_thr);
      w.a.x += R.c[9] * delta; w.a.y += R.c[10] * delta; w.a.z += R.c[11] * delta;
      w.l.x += R.c[12] * delta; w.l.y += R.c[13] * delta; w.l.z += R.c[14] * delta;
      y0 = R.y0 + R.c[15] * delta; y1 = R.y1 + R.c[16] * delta; y2 = R.y2 + R.c[17] * delta;
    }
    auxv[__mul24(rw, 4 * LN) + R_::A_LAM * LN] = sum;    // (dead lanes rewrite their last row's own value)
    yl[R.yoff] = y0; yl[R.yoff + LN] = y1; yl[R.yoff + 2 * LN] = y2;
    // forward what the prefetch could not have seen yet
    const bool samey = N.yoff == R.yoff;
    N.y0 = samey ? y0 : N.y0; N.y1 = samey ? y1 : N.y1; N.y2 = samey ? y2 : N.y2;
    N.lamp = N.par == rw ? sum : N.lamp;
    N.lam = rn == rw ? sum : N.lam;
  };
#pragma unroll 1
  for (int it = 0; it < iterations; it++) {
    Row A, B;
    fetch(0, A);
    viol = false;
#pragma unroll 1
    for (int r = 0; r < wmax; r += 2) {  // ping-pong: no register rotation
      step(r, A, B);
      step(r + 1, B, A);
    }
    conv = conv || !viol;               // K7: this lane's solve ends after the first sweep within the residual threshold
    if (!__any(!conv)) break;
  }
  C.w = w;
#pragma unroll
  for (int l = 0; l < 4; l++)
#pragma unroll
    for (int k = 0; k < 3; k++) C.y[l][k] = yl[(l * 3 + k) * LN];
#pragma unroll
  for (int i = 0; i < 12; i++) {
    const int p = 12 + i;
    C.lam_n[i] = ((C.mask >> p) & 1) ? lds.A(nlt + __builtin_popcount(C.mask & ((1 << p) - 1)), R_::A_LAM) : T(0);
  }
}

// ---------------------------------------------------------------- phase 5: apply + integrate
// delta-velocities (clamp K5), impulse cache, semi-implicit Euler (K1)
template <typename T, int ROBOT, typename LDS, typename CH>
SNI void phase_integrate(CH ch, const PhysParams<T> pp, T* lam_prev, unsigned nstride, const LDS lds) {
  SubCtx<T, ROBOT>& C = ch.get();
  using RB = Robot<ROBOT>;
  constexpr int NJ = RB::NJ;
  const T dt = pp.dt;
  const int mask = C.mask, nlt = C.nlim_total;
  static_for<NPRIM>([&](auto pc) {
    constexpr int p = decltype(pc)::value;
    T l = T(0);
    if constexpr (p < 12) {
      if ((mask >> p) & 1) l = lds.A(nlt + __builtin_popcount(mask & ((1 << p) - 1)), LDS::A_LAM);
    } else l = C.lam_n[p - 12];
    lam_prev[(unsigned)p * nstride] = l;
  });
  const SV<T> w = C.w, ub = C.ub;
  PhysState<T, RB::NQ> st = C.ps;
  st.w = mk(clampv(ub.a.x + w.a.x, pp.vmax), clampv(ub.a.y + w.a.y, pp.vmax), clampv(ub.a.z + w.a.z, pp.vmax));
  st.v = mk(clampv(ub.l.x + w.l.x, pp.vmax), clampv(ub.l.y + w.l.y, pp.vmax), clampv(ub.l.z + w.l.z, pp.vmax));
  static_for<4>([&](auto lc) {
    constexpr int L = decltype(lc)::value;
#pragma unroll
    for (int k = 0; k < NJ; k++) {
      T nv = clampv(C.qds[L * NJ + k] + C.y[L][k] - dot(C.LR[L].G[k], w), pp.vmax);
      st.qd[L * NJ + k] = nv;
      st.q[L * NJ + k] += dt * nv;
    }
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
  C.ps = st;
}

#ifndef SOLO_HOST_SHIM
// ---------------------------------------------------------------- team mode (default)
// 16 lanes (one DPP row) per env, 4 envs per wavefront.  Per sub-step (substep_team): sin/cos, collision
// points, leg rates and integration run one joint / primitive per lane, the four legs on four lanes, the
// base solve on the team leader (lane 0 of the row); rows are finished by all 16 lanes and swept by
// pgs_team_variant, where each 8-lane half holds the 18 accumulator components [w(6), y_leg0..3 (12)].

// Team-sweep rows.  The sweep works on STATIC positions, two per slot:
//   slot 0        positions 0,1           the (<= 2) joint-limit rows
//   slots 1..4    positions 2+k           normal row of contact k (k = rank of its primitive among the contacts)
//   slots 5..12   positions 10+2k, 11+2k  the two friction rows of contact k
// which keeps the solver order [limits | normals | friction pairs] and makes the parent of friction slot 5+k a
// compile-time register: the impulse of slot 1+k/2, half k&1.  A team with fewer rows than its wavefront sweeps
// null rows at the unused positions (all-zero, decided by the sweep's set-up from the team's row counts).
// The finished rows stay where they were parked (RowLds core, dense solver index r <-> position by pos_of /
// row_of), rewritten IN PLACE by phase_finish_team as the 20-value record the sweep needs:
//   [2e] J'_e  [2e+1] B~_e  for the components e = 0..5 (base) and 6..8 (the row's leg)   [18] rhs'  [19] leg id
//   (round 3: J' and B~ interleaved -- the sweep's set-up reads a lane's pair with one 64-bit LDS read)
// Friction rows are stored in units of their contact's mu (lambda~ = lambda/mu, J' = J/mu, B~ = mu B) and every row
// is pre-scaled by 1/diag, so the sweep needs neither mu nor 1/diag.  (Round 1 expanded every row into a separate
// 38-value record with explicit zeros for the other legs' components: 15.8 KB of the 34.6 KB per workgroup, which
// held the kernel at one wavefront per SIMD; now 18.4 KB -> 8 workgroups per CU, two wavefronts per SIMD.)
//   lam  [TR][4]       impulses by position (warm start in, result out)
//   bc   [4][BC]       base-solve broadcast block (Lam 36, u*_base 6, leg rates 12, a0 6)
// (J', B~) of component e interleaved at 2e, 2e+1 (e = 0..5 base, 6..8 leg): the sweep's set-up reads a lane's pair with one 64-bit LDS read
enum { E_RHS = 18, E_LEG = 19 };
constexpr int E_J(int e) { return 2 * e; }
constexpr int E_B(int e) { return 2 * e + 1; }
template <typename T, typename LDS> struct TeamRows {
  static_assert(LIM_SLOT_ROWS == 2 && MAX_LIMITS <= LIM_SLOT_ROWS + MAX_CONTACTS && MAX_CONTACTS % 2 == 0, "slot map below assumes one limit slot and paired normals");
  static constexpr int NPOS0 = 2, FPOS0 = NPOS0 + MAX_CONTACTS;              // first normal / friction position
  static constexpr int NSLOT = 1 + MAX_CONTACTS / 2 + MAX_CONTACTS;          // 13
  static constexpr int TR = 2 * NSLOT, BC = 60;
  static constexpr size_t off_lam = LDS::bytes(4);
  static constexpr size_t off_bc = off_lam + (size_t)TR * 4 * sizeof(T);
  static constexpr size_t off_ctx = (off_bc + (size_t)4 * BC * sizeof(T) + 15) & ~(size_t)15;   // 4 x SubCtx (size added by the user)
  static constexpr size_t bytes = off_ctx;
  SD static T* lam(int col) { return reinterpret_cast<T*>(solo_smem + off_lam) + col; }            // + pos*4
  SD static T* bc(int col) { return reinterpret_cast<T*>(solo_smem + off_bc) + col * BC; }         // Lam 36, ub 6, leg rates 12, a0 6
  // Limit rows beyond the limit slot's two (ne = extra_limits(nlt)) sit at the FIRST normal positions -- the solver order stays
  // [limits | normals | friction pairs] -- and shift the contacts: contact k has its normal at position NPOS0 + ne + k and its friction
  // pair at FPOS0 + 2 (ne + k); the friction positions of the first ne "contacts" stay empty (null rows).
  SD static int pos_of(int r, int nlt, int nc) {   // solver row index -> position
    const int ne = extra_limits(nlt), nl0 = nlt - ne;
    return r < nl0 ? r : (r < nlt ? NPOS0 + (r - nl0) : (r < nlt + nc ? NPOS0 + ne + (r - nlt) : FPOS0 + 2 * ne + (r - nlt - nc)));
  }
  // position -> solver row index, or -1 where the team has no row (null row)
  SD static int row_of(int pos, int nlt, int nc) {
    const int ne = extra_limits(nlt), nl0 = nlt - ne;
    const int kn = pos - NPOS0, kf = pos - FPOS0 - 2 * ne;
    // (selects, not branches: every candidate is computed -- the sweep's set-up evaluates this for 26 positions and a
    // divergent branch per position cost more than the arithmetic)
    const int rl = pos < nl0 ? pos : -1;
    const int a_lim = nl0 + kn, a_con = nlt + (kn - ne), a_fr = nlt + nc + kf;
    const int rc = (kn - ne < nc) ? a_con : -1;
    const int rn = kn < ne ? a_lim : rc;
    const int rf = ((kf >= 0) & (kf < 2 * nc)) ? a_fr : -1;
    const int rnf = pos < FPOS0 ? rn : rf;
    return pos < NPOS0 ? rl : rnf;
  }
};

// OR over the 16 lanes of a team (every lane ends with the full value)
SD int team_or16(int x) {
  x |= __builtin_amdgcn_mov_dpp(x, 0xB1, 0xF, 0xF, true);     // quad_perm [1,0,3,2]
  x |= __builtin_amdgcn_mov_dpp(x, 0x4E, 0xF, 0xF, true);     // quad_perm [2,3,0,1]
  x |= __builtin_amdgcn_mov_dpp(x, 0x141, 0xF, 0xF, true);    // row_half_mirror
  x |= __builtin_amdgcn_mov_dpp(x, 0x140, 0xF, 0xF, true);    // row_mirror
  return x;
}

// Collision detection, team mode (replaces phase_detect): the four legs' support points on lanes 0..3, the
// twelve base points on lanes 4..15, joint-limit tests one joint per lane; masks are OR-reduced over the team.
// The leader then records the counts and starts the articulated-inertia accumulation with the base link.
template <typename T, int ROBOT, typename LDS, typename CH, bool UI>
SNI_SCALAR void phase_front_team(CH ch, const PhysParams<T> pp, const LDS lds, int t, bool valid, bool lead) {
  using RB = Robot<ROBOT>;
  constexpr int NJ = RB::NJ, NQ = RB::NQ, ST = NJ + 1;
  SubCtx<T, ROBOT>& C = ch.get();
  const PhysState<T, NQ>& st = C.ps;
  const M3<T> R0 = quat_to_mat(st.qx, st.qy, st.qz, st.qw);
  const T pz = st.pos.z;
  const T ty = st.pos.y - C.tmy;    // treadmill: point P (relative to the base origin) is on the strip iff |ty + P.y| <= tm_hw
  int bits = 0, lbits = 0, sbits = 0;
  V3<T> bP = mk(T(0), T(0), T(0)); T bdist = T(0), bfric = T(0);      // base lanes: their point, distance, friction
  if (valid && t < 4) {
    const int L = t;
    const LegSign<T> lsg = leg_sign<T>(L);
#define LEGC(k, f) SEL4(T, lsg, RB::MD.links[1 + 0 * ST + (k)].f, RB::MD.links[1 + 1 * ST + (k)].f, \
                        RB::MD.links[1 + 2 * ST + (k)].f, RB::MD.links[1 + 3 * ST + (k)].f)
#define PRIMC(i, f) SEL4(T, lsg, RB::MD.prims[12 + (i)].f, RB::MD.prims[14 + (i)].f, RB::MD.prims[16 + (i)].f, RB::MD.prims[18 + (i)].f)
    M3<T> Rp = R0;
    V3<T> op = mk(T(0), T(0), T(0));
#define SHC(f) SEL4(T, lsg, RB::MD.prims[20].f, RB::MD.prims[21].f, RB::MD.prims[22].f, RB::MD.prims[23].f)
    V3<T> kneeP = op, footP = op, shP = op;
    static_for<NJ>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      constexpr int AX = RB::MD.links[1 + k].axis[0] != 0.0 ? 0 : 1;   // same for all legs
      const V3<T> o = op + mul(Rp, mk(LEGC(k, jorigin[0]), LEGC(k, jorigin[1]), LEGC(k, jorigin[2])));
      const M3<T> R = rot_axis<AX>(Rp, C.cs[L * NJ + k], C.sn[L * NJ + k]);
      if constexpr (RB::SHOULDER && k == 0)   // shoulder housing: disc about x on the HAA link (Solo12)
        shP = disc_point_x(R, o + mul(R, mk(SHC(center[0]), SHC(center[1]), SHC(center[2]))), SHC(radius));
      if constexpr (k == NJ - 2)   // knee disc sits on the upper leg
        kneeP = disc_point(R, o + mul(R, mk(PRIMC(0, center[0]), PRIMC(0, center[1]), PRIMC(0, center[2]))), PRIMC(0, radius), PRIMC(0, halfw));
      if constexpr (k == NJ - 1)   // foot: fixed child of the last link, primitive centred on its origin
      {
        static_assert(RB::MD.prims[13].nring == 4, "foot primitive: four profile rings (tools/compile_model.py)");
        const T fy[4] = {PRIMC(1, ring_y[0]), PRIMC(1, ring_y[1]), PRIMC(1, ring_y[2]), PRIMC(1, ring_y[3])};
        const T fr[4] = {PRIMC(1, ring_r[0]), PRIMC(1, ring_r[1]), PRIMC(1, ring_r[2]), PRIMC(1, ring_r[3])};
        footP = ring_disc_point<T, 4>(R, o + mul(R, mk(LEGC(NJ, jorigin[0]), LEGC(NJ, jorigin[1]), LEGC(NJ, jorigin[2]))), fy, fr);
      }
      Rp = R; op = o;
    });
    kneeP.z -= pp.cmargin; footP.z -= pp.cmargin; shP.z -= pp.cmargin;
    C.kneeP[L] = kneeP; C.footP[L] = footP;
    const T dk = pz + kneeP.z, df = pz + footP.z;
    C.dist[12 + 2 * L] = dk; C.dist[13 + 2 * L] = df;
    if (dk < PRIMC(0, margin)) bits |= 1 << (12 + 2 * L);
    if (df < PRIMC(1, margin)) bits |= 1 << (13 + 2 * L);
    if (pp.tm_on()) {
      if (fabs(ty + kneeP.y) <= pp.tm_hw) sbits |= 1 << (12 + 2 * L);
      if (fabs(ty + footP.y) <= pp.tm_hw) sbits |= 1 << (13 + 2 * L);
    }
    if constexpr (RB::SHOULDER) {
      C.shP[L] = shP;
      const T ds = pz + shP.z;
      C.dist[20 + L] = ds;
      if (ds < SHC(margin)) bits |= 1 << (20 + L);
      if (pp.tm_on() && fabs(ty + shP.y) <= pp.tm_hw) sbits |= 1 << (20 + L);
    } else C.dist[20 + L] = T(1);              // (no such primitive: never in contact, never ranked)
#undef SHC
#undef PRIMC
#undef LEGC
  } else if (valid) {
    const int p = t - 4;     // base point p: the twelve points are sign patterns of two corner points (8 box-diagonal extremes of the
    // hull, prims 0..7 = (sx, sy, sz) with p = 4 [sx < 0] + 2 [sy < 0] + [sz > 0]; 4 belly-plate corners, prims 8..11 = (sx, sy) with
    // p - 8 = 2 [sx < 0] + [sy < 0]) and share margin and friction: arithmetic on the lane index instead of a 12-way select (which the
    // compiler turned into a tree of ~40 exec-mask branches on the path of every sub-step)
    constexpr solorl_prim_data PA = RB::MD.prims[0], PB = RB::MD.prims[8];
    static_assert(base_points_are_sign_patterns(RB::MD), "base primitive table layout (tools/compile_model.py)");
    const bool plate = p >= 8;
    const T sgx = ((plate ? p >> 1 : p >> 2) & 1) ? T(-1) : T(1), sgy = ((plate ? p : p >> 1) & 1) ? T(-1) : T(1);
    const T cx = (plate ? T(PB.center[0]) : T(PA.center[0])) * sgx, cy = (plate ? T(PB.center[1]) : T(PA.center[1])) * sgy;
    const T cz = plate ? T(PB.center[2]) : ((p & 1) ? T(-PA.center[2]) : T(PA.center[2]));
    const T mg = T(PA.margin);
    bfric = T(PA.friction);
    bP = mul(R0, mk(cx, cy, cz));
    bP.z -= pp.cmargin;
    bdist = pz + bP.z;
    C.dist[p] = bdist;
    if (bdist < mg) bits |= 1 << p;
    if (pp.tm_on() && fabs(ty + bP.y) <= pp.tm_hw) { sbits |= 1 << p; bfric *= pp.tm_mu; }
  }
  if (valid && t < NQ) {    // joint limits: bit 2j = lower window, 2j+1 = upper window (same order as the row slots)
    const T q = st.q[t];
    if (q + pp.qlim <= T(LIMIT_WINDOW)) lbits |= 1 << (2 * t);
    if (pp.qlim - q <= T(LIMIT_WINDOW)) lbits |= 1 << (2 * t + 1);
  }
  int mask = team_or16(bits);
  lbits = team_or16(lbits);
  if (__builtin_popcount(lbits) > MAX_LIMITS) {
    // more joints at their limits than limit rows (MAX_LIMITS): the smallest margins, i.e. the most violated, win
    // (ties: lower joint, lower side) -- same rule as phase_detect, see there.  Lane t ranks its own joint's two sides against
    // the team's margins in LDS.
    if (valid && t < NQ) { C.limpen[2 * t] = st.q[t] + pp.qlim; C.limpen[2 * t + 1] = pp.qlim - st.q[t]; }
    TEAM_SYNC();
    int sel = 0;
    if (valid && t < NQ) {             // (left to the compiler's unrolling: pinned to a rolled loop the step was 0.7 % slower)
      for (int side = 0; side < 2; side++) {
        const int a = 2 * t + side;
        if ((lbits >> a) & 1) {
          const T pa = C.limpen[a];
          int rank = 0;
          for (int b = 0; b < 2 * NQ; b++) {
            const T pb = C.limpen[b];
            rank += (b != a && ((lbits >> b) & 1) && (pb < pa || (pb == pa && b < a))) ? 1 : 0;
          }
          if (rank < MAX_LIMITS) sel |= 1 << a;
        }
      }
    }
    lbits = team_or16(sel);
  }
  if (pp.tm_on()) sbits = team_or16(sbits);     // (uniform branch)
  const int capc = contact_cap(__builtin_popcount(lbits));   // limit rows 3 and 4 take the place of contact points (see MAX_LIMITS)
  if (__builtin_popcount(mask) > capc) {
    // more primitives touch than there are contact slots (a robot lying on the ground -- the heaviest wavefronts, which set the launch
    // time): keep the deepest, ties to the lower id.  Lane t ranks primitives t and t+16 against the team's distances in LDS
    // (the serial version on the leader was a 1 900-instruction non-inlined call that also made this phase save registers on
    // every call).
    TEAM_SYNC();                                       // C.dist: written by the primitives' own lanes above
    int keep = 0;
    for (int p = t; p < NPRIM; p += 16) {
      if ((mask >> p) & 1) {
        const T dp = C.dist[p];
        int rank = 0;
        for (int q = 0; q < NPRIM; q++) {
          const T dq = C.dist[q];
          rank += (((mask >> q) & 1) && (dq < dp || (dq == dp && q < p))) ? 1 : 0;
        }
        if (rank < capc) keep |= 1 << p;
      }
    }
    mask = team_or16(keep);
  }
  const int nl = __builtin_popcount(lbits);
  if (valid && t >= 4 && ((mask >> (t - 4)) & 1)) {    // base contacts: every touching point parks its own three rows
    const int p = t - 4, nlt = nl, nc = __builtin_popcount(mask);
    const int cidx = __builtin_popcount(mask & ((1 << p) - 1));
    const int sn_ = nlt + cidx, sf_ = nlt + nc + 2 * cidx;
    const T lam0 = pp.warm * C.lamp[p];
    const T Z[3] = {T(0), T(0), T(0)};
    static_for<3>([&](auto dc) {
      constexpr int d = decltype(dc)::value;
      const V3<T> u = d == 0 ? mk(T(0), T(0), T(1)) : (d == 1 ? mk(T(1), T(0), T(0)) : mk(T(0), T(1), T(0)));
      park_row(lds, d == 0 ? sn_ : sf_ + (d - 1), SV<T>{cross(bP, u), u}, Z, Z, bP, bdist + pp.slop, d == 0 ? lam0 : T(0), bfric,
               (d == 0 ? 0 : (sn_ | 128)) | (d << 8));
    });
  }
  if (!lead) return;
  C.R0 = R0;
  C.mask = mask; C.nc = __builtin_popcount(mask); C.nlim_total = nl; C.lsel = lbits;
  C.smask = sbits & mask;
  // base link terms start the articulated-inertia accumulation
  constexpr solorl_link_data B = RB::MD.links[0];
  static_assert(B.com[0] == 0.0 && B.com[1] == 0.0 && B.com[2] == 0.0, "base COM at its origin");
  RBI<T> Ib; SV<T> pb;
  const SV<T> vb{st.w, st.v};
  SOLO_LINK_TERMS_CE(UI, B, R0, mk(T(0), T(0), T(0)), vb, pp.damping, Ib, pb);
  C.Ibase = to_abi(Ib); C.pbase = pb;
}

// base articulated inertia / bias = base link + the four legs' contributions (phase_leg_rt): the 21 + 6 values
// are summed one (two) per lane straight in the LDS context
template <typename T, int ROBOT, typename CH>
SD void team_sum_base(CH ch, int t) {
  SubCtx<T, ROBOT>& C = ch.get();
  constexpr int NA = (int)(sizeof(ABI<T>) / sizeof(T)), NP = (int)(sizeof(SV<T>) / sizeof(T));
  static_assert(NA == 21 && NP == 6, "ABI / SV are plain arrays of T");
  T* ib = reinterpret_cast<T*>(&C.Ibase); T* pb = reinterpret_cast<T*>(&C.pbase);
  const T* il = reinterpret_cast<const T*>(&C.Ileg[0]); const T* pl = reinterpret_cast<const T*>(&C.pleg[0]);
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const int i = t + 16 * k;
    if (i < NA) ib[i] = ib[i] + ((il[i] + il[NA + i]) + (il[2 * NA + i] + il[3 * NA + i]));
    else if (i < NA + NP) { const int j = i - NA; pb[j] = pb[j] + ((pl[j] + pl[NP + j]) + (pl[2 * NP + j] + pl[3 * NP + j])); }
  }
}

// leader: base solve, then publish what the other lanes need (bc: Lam 36, u*_base 6, padded leg rates 12, -a0 6... see TeamRows)
// TEAMQ = true: the leg rates u*_leg (C.qds, bc[42..]) are left to phase_legrates_team (one joint per lane).
template <typename T, int ROBOT, typename LDS, typename CH>
SNI_SCALAR void phase_base_lead(CH ch, const PhysParams<T> pp, const T* lam_prev, unsigned nstride, const LDS lds) {
  SubCtx<T, ROBOT>& C = ch.get();
  using TRW = TeamRows<T, LDS>;
  constexpr int LN = LDS::LANES;
  // the arithmetic of base_solve<TEAMQ = true>, laid out as LOAD everything -> compute -> STORE everything: LDS pointers may alias as
  // far as the compiler knows, so a read behind a store is a separate round trip (~64 cycles each for a lone wavefront; this phase had a
  // dozen, five of them reading back the u* it had just stored)
  const int nlt = C.nlim_total, nc = C.nc;
  const ABI<T> Ib = C.Ibase;
  const SV<T> pb = C.pbase;
  const V3<T> bw = C.ps.w, bv = C.ps.v;
  const Sym6<T> Lam = spd_inverse_block(Ib);
  const SV<T> a0 = mul(Lam, pb) * T(-1);
  const T dt = pp.dt;
  V3<T> vdot = a0.l + cross(bw, bv); vdot.z -= pp.gravity;      // base acceleration (gravity via the accelerating-frame trick), u* = u + dt*udot
  SV<T> ub;
  ub.a = mk(clampv(bw.x + dt * a0.a.x, pp.vmax), clampv(bw.y + dt * a0.a.y, pp.vmax), clampv(bw.z + dt * a0.a.z, pp.vmax));
  ub.l = mk(clampv(bv.x + dt * vdot.x, pp.vmax), clampv(bv.y + dt * vdot.y, pp.vmax), clampv(bv.z + dt * vdot.z, pp.vmax));
  if (nlt + 3 * nc == 0) {   // the branch-free PGS sweep touches row 0 of every lane: make it a null row
    const T Z[3] = {T(0), T(0), T(0)};
    park_row(lds, 0, zero6<T>(), Z, Z, mk(T(0), T(0), T(0)), T(0), T(0), T(0), 0);
  }
  C.ub = ub;
  T* bc = TRW::bc(lds.lane);
#pragma unroll
  for (int i = 0; i < 6; i++)
#pragma unroll
    for (int j = 0; j < 6; j++) bc[i * 6 + j] = Lam.m[i][j];
  bc[36] = ub.a.x; bc[37] = ub.a.y; bc[38] = ub.a.z; bc[39] = ub.l.x; bc[40] = ub.l.y; bc[41] = ub.l.z;
  bc[54] = a0.a.x; bc[55] = a0.a.y; bc[56] = a0.a.z; bc[57] = a0.l.x; bc[58] = a0.l.y; bc[59] = a0.l.z;
  T* hdr = lds.hdr();
  hdr[0] = T(nlt); hdr[LN] = T(nc);
}

// unconstrained leg rates u*_leg = qd + dt (qdd0 - G.a0), clamped: joint t on lane t (padded copy for the rows)
template <typename T, int ROBOT, typename LDS, typename CH>
SD void phase_legrates_team(CH ch, const PhysParams<T> pp, const LDS lds, int t) {
  using RB = Robot<ROBOT>;
  using TRW = TeamRows<T, LDS>;
  constexpr int NJ = RB::NJ;
  T* bc = TRW::bc(lds.lane);
  if (t < RB::NQ) {
    SubCtx<T, ROBOT>& C = ch.get();
    const int L = t / NJ, k = t - L * NJ;
    const SV<T> a0{{bc[54], bc[55], bc[56]}, {bc[57], bc[58], bc[59]}};
    const T qdd = C.LR[L].qdd0[k] - dot(C.LR[L].G[k], a0);
    const T v = clampv(C.ps.qd[t] + pp.dt * qdd, pp.vmax);
    C.qds[t] = v;
    bc[42 + L * 3 + k] = v;
  } else if (NJ < 3 && t < 12) {          // Solo8: third (absent) joint of each leg
    if ((t - RB::NQ) < 4) bc[42 + (t - RB::NQ) * 3 + 2] = T(0);
  }
}

// this team's row counts and the wave-uniform sweep extent (every team of the wave sweeps the same slots)
template <typename T, typename LDS> SD void team_counts(const LDS& lds, int& nlt, int& nc, int& ncmax, int& anylim) {
  const T* hdr = lds.hdr();
  nlt = (int)hdr[0]; nc = (int)hdr[LDS::LANES];
  int m = (nc + extra_limits(nlt)) | (nlt << 8);       // normal positions in use: the contacts behind the extra limit rows
#pragma unroll
  for (int o = 32; o >= 16; o >>= 1) { int v = __shfl_xor(m, o); m = ((v & 255) > (m & 255) ? (v & 255) : (m & 255)) | ((v | m) & ~255); }
  m = __builtin_amdgcn_readfirstlane(m);          // identical in all lanes: scalar branches in the sweep
  ncmax = m & 255; anylim = m >> 8;
}

// all 16 lanes: lane t finishes rows t, t+16 IN PLACE (record layout: TeamRows) and publishes their warm-start impulses
// by position.  A lane only overwrites the rows it read itself, so no ordering between lanes is needed.
template <typename T, int ROBOT, typename LDS>
SNI_SCALAR void phase_finish_team(const PhysParams<T> pp, const LDS lds, int t) {
  using TRW = TeamRows<T, LDS>;
  const int col = lds.lane;
  int nlt, nc, ncmax, anylim;
  team_counts<T, LDS>(lds, nlt, nc, ncmax, anylim);
  const int nrows = nlt + 3 * nc, rfric = nlt + nc;
  T* const lam = TRW::lam(col);
  const T* bc = TRW::bc(col);
  // impulses of the positions this team leaves empty (the wave may sweep them as null rows)
  for (int pos = t; pos < TRW::TR; pos += 16)
    if (TRW::row_of(pos, nlt, nc) < 0) lam[pos * 4] = T(0);
  if (t >= nrows) return;
  Sym6<T> Lam;
#pragma unroll
  for (int i = 0; i < 6; i++)
#pragma unroll
    for (int j = 0; j < 6; j++) Lam.m[i][j] = bc[i * 6 + j];
  const SV<T> ub{{bc[36], bc[37], bc[38]}, {bc[39], bc[40], bc[41]}};
  // rows t and t + 16 (the second only for >= 17 rows, i.e. the heaviest teams -- the ones that set the launch time):
  // both rows' loads are issued before either row's stores so that their LDS latencies overlap instead of adding up
  const int r1 = t + 16;
  const bool two = r1 < nrows;
  T c0[ROW_CORE], c1[ROW_CORE];
  lds.load_core(t, c0);
  const int meta0 = (int)lds.A(t, LDS::A_META);
  const T mu0 = lds.A(t, LDS::A_MU);
  int meta1 = 0; T mu1 = T(1);
  if (two) { lds.load_core(r1, c1); meta1 = (int)lds.A(r1, LDS::A_META); mu1 = lds.A(r1, LDS::A_MU); }
  auto emit = [&](int r, const T (&c)[ROW_CORE], int meta, T mu) __attribute__((always_inline)) {
    const int leg = (meta >> 5) & 3;
    const T* ql = bc + 42 + leg * 3;
    SV<T> W; T rhs, dinv;
    finish_row(c, meta, Lam, ub, ql[0], ql[1], ql[2], pp, W, rhs, dinv);
    const bool fr = r >= rfric;
    const T sB = fr ? mu : T(1);
    const T isB = rcp_fast(sB);
    const T sJ = dinv * isB;
    T o[ROW_CORE];
#pragma unroll
    for (int e = 0; e < 9; e++) o[E_J(e)] = c[e] * sJ;                       // J': base 0..5, leg 6..8
    o[E_B(0)] = W.a.x * sB; o[E_B(1)] = W.a.y * sB; o[E_B(2)] = W.a.z * sB; o[E_B(3)] = W.l.x * sB; o[E_B(4)] = W.l.y * sB; o[E_B(5)] = W.l.z * sB;
    o[E_B(6)] = c[15] * sB; o[E_B(7)] = c[16] * sB; o[E_B(8)] = c[17] * sB;
    o[E_RHS] = rhs * isB;               // finish_row's rhs already carries 1/diag
    o[E_LEG] = T(leg);
    lds.store_core(r, o);
    // K7 in the row's units: |delta lambda~| / sJ = |delta impulse| / jacDiagABInv.  Friction rows of the cone model store the reciprocal
    // instead: the pair's residual entry is the SUM of its two velocity changes, |d0 / (thr sJ0) + d1 / (thr sJ1)| > 1 (pgs_team_variant)
    lds.A(r, LDS::A_LAM) = (fr && pp.cone()) ? rcp_fast(pp.resid_thr * sJ) : pp.resid_thr * sJ;
    lam[TRW::pos_of(r, nlt, nc) * 4] = c[13];      // warm-start impulse (0 for friction / limit rows)
  };
  emit(t, c0, meta0, mu0);
  if (two) emit(r1, c1, meta1, mu1);
}

// exchange between the two 8-lane halves of a 16-lane team (values are uniform within a half)
template <typename T> SD T half_swap(T x) {
  if constexpr (sizeof(T) == 4) return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x140, 0xF, 0xF, true));   // row_mirror
  else return __shfl_xor(x, 8, 16);
}

// Register-resident slot sweep.  Each 8-lane half of the team holds a full copy of the 18 accumulator
// components (3 per lane); in slot k half 0 owns position 2k and half 1 position 2k+1, and every lane keeps
// ITS row's J', B~, rhs' and impulse plus the partner row's B~ in VGPRs for the whole solve (everything is
// statically indexed): the 50 sweeps never touch memory.  Both halves reduce their own row's J'.acc at the
// same time against the pre-update accumulators; the Gauss-Seidel dependency of position 2k+1 on 2k is
// restored analytically,
//   J'_{2k+1}.(acc + B~_{2k} d0) = J'_{2k+1}.acc + c'_k d0,
// a re-association of the same sums.
//
// One instantiation per slot set of the wave (LIM: the limit slot, NNS normal slots, NFS friction slots; teams
// with fewer rows sweep null rows), so the slot sequence -- each slot's predecessor and successor, with
// wrap-around between sweeps -- is static.  A lone wavefront issues a DEPENDENT instruction only every
// ~8.5 cycles (16 after a DPP move; issue-rate microbenchmark of rounds 1-2, profiles/r02_notes.md) and Gauss-Seidel is one long dependency
// chain, so the chain is cut down to the scalar clamp: the J'.acc reduction of the NEXT slot (multiplies +
// DPP butterfly, 2/3 of a slot's latency) is started one slot early on accumulators that lack the current
// slot's update, which is added back analytically through two more precomputed couplings,
//   J'_n.(acc + B~_own d_own + B~_oth d_oth) = J'_n.acc + eo_n d_own + ex_n d_oth,
// and overlaps with the current slot's clamp chain.
template <typename T> SD T team_red8(T x) {      // sum over the 8 lanes of a half (both halves at once)
  if constexpr (sizeof(T) == 4) {
    x += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    x += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    x += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x141, 0xF, 0xF, true));   // row_half_mirror
  } else {
    x += __shfl_xor(x, 1, 16); x += __shfl_xor(x, 2, 16); x += __shfl_xor(x, 4, 16);
  }
  return x;
}

// EXIT: with the K7 residual test (solver_residual_threshold > 0); the default fixed-iteration solve carries none of it.
// PIPE: the next slot's J'.acc reduction is started one slot early and corrected with two couplings (3 more instructions per
//   slot, a 6-operation shorter dependency chain).  Measured: faster when a wavefront has its SIMD to itself (<= 4096 envs:
//   0.172 vs 0.180 ms per step), slower when two wavefronts share one and issue slots are what counts (8192 envs: 0.229 vs
//   0.223 ms) -- the launcher picks by grid size (PhysParams::pgs_pipe).  The K7 variants are always pipelined.
#ifndef SOLO_SETUP_GROUP_F32
#define SOLO_SETUP_GROUP_F32 5
#endif
#ifndef SOLO_SETUP_GROUP_F64
#define SOLO_SETUP_GROUP_F64 1
#endif
// CONE (solorl_config friction_model = cone): Bullet's implicit friction cone [K].  A friction slot already holds the two directions of
//   one contact, one per half; both halves take their unclamped sums against the pre-update accumulators (no Gauss-Seidel coupling c'
//   between the two), exchange them, and scale the pair by min(1, mu lambda_n / |pair|) -- resolveConeFrictionConstraintRows' atan2 / sin /
//   cos is this radial projection.  Its K7 residual entry is the SUM of the two velocity changes.
template <typename T, typename LDS, int LIM, int NNS, int NFS, bool EXIT, bool PIPE = true, bool CONE = false>
SNI void pgs_team_variant(int iterations_v, const LDS lds, int t) {
  // (the residual threshold of K7 reaches the sweep through the rows: phase_finish_team stores it per row, in the row's units)
  using TRW = TeamRows<T, LDS>;
  const int itv = __builtin_amdgcn_readfirstlane(iterations_v);          // function arguments arrive in VGPRs: make the sweep loop scalar
  const int iterations = itv & 0xFFFF;
  const bool warm_on = (itv >> 30) & 1;                                  // (bit 30: the rows carry warm-start impulses, phase_pgs_team)
  constexpr int LN = LDS::LANES, n = LIM + NNS + NFS;
  constexpr int S_N0 = 1, S_F0 = 1 + MAX_CONTACTS / 2;      // slot numbers of the first normal / friction slot
  static_assert(n >= 1, "empty sweeps are not instantiated");
  // sweep order i = 0..n-1 -> slot number (position pair) in the team records
  struct Ord { static constexpr int slot(int i) { return i < LIM ? 0 : (i < LIM + NNS ? S_N0 + (i - LIM) : S_F0 + (i - LIM - NNS)); } };
  const int col = lds.lane;
  const int h = t & 7, half = t >> 3;
  constexpr int SZ = (int)sizeof(T);
  constexpr int S_LAM = 4 * SZ;
  const int lam_own = (int)TRW::off_lam + col * SZ + half * S_LAM;
  // this team's row counts: position -> parked row (TeamRows::row_of), or a null row
  const int nlt = (int)lds.hdr()[0], nc = (int)lds.hdr()[LN];
  // Lane h of a half holds accumulator components h, h + 8 and (h < 2) 16 + h of [w(6), y_leg0..3(12)].  Where they
  // sit in a finished record (TeamRows): component c < 6 -> base element c; c >= 6 -> leg (c-6)/3, joint (c-6)%3,
  // present only in rows of that leg.
  const int eU = h;                                   // comp h: base 0..5 | leg 0 joints 0,1 (component index 6, 7)
  const int legU = h < 6 ? -1 : 0;                    // -1: every row
  const int eV = 6 + (h + 2) % 3, legV = (h + 2) / 3;
  const int eW = h < 2 ? 7 + h : 8, legW = h < 2 ? 3 : -2;   // -2: no row (the zero pair; the lane reads a valid component and discards it)
  // element e of parked row r sits at r * ROWB + eoff(e): the lane's six element offsets are computed once, a read is one add away from its
  // row base (written as one expression per read the compiler rebuilt the whole index arithmetic each time: 35 instructions per slot for 13 reads)
  constexpr unsigned ROWB = (unsigned)LDS::NCH * LN * 16u;
  auto eoff = [&](int e) -> unsigned { return (unsigned)(e / LDS::PER) * (LN * 16u) + (unsigned)(e % LDS::PER) * SZ + (unsigned)col * 16u; };
  using P2 = T __attribute__((ext_vector_type(2)));
  static_assert(LDS::PER % 2 == 0, "a (J', B~) pair lies inside one 16-byte chunk");
  const unsigned kU = eoff(E_J(eU)), kV = eoff(E_J(eV)), kW = eoff(E_J(eW));            // the lane's (J', B~) pairs: 8-byte aligned
  const unsigned kUB = eoff(E_B(eU)), kVB = eoff(E_B(eV)), kWB = eoff(E_B(eW));
  const unsigned kLEG = eoff(E_LEG), kRHS = eoff(E_RHS);
  auto rd = [&](unsigned rb, unsigned k) -> T { return *reinterpret_cast<const T*>(solo_smem + (rb + k)); };
  auto rd2 = [&](unsigned rb, unsigned k) -> P2 { return *reinterpret_cast<const P2*>(solo_smem + (rb + k)); };
  // all arrays below are indexed by the sweep index i.  The B~ / X~ columns of accumulator components 0 and 1 are kept as
  // explicit 2-vectors (v_pk_fma_f32 on fp32; the translation unit is built with -fno-slp-vectorize, so what is packed
  // is exactly what is written here -- left to itself the vectoriser paired unrelated multiplies of the slot's dependency
  // chain and split its fused multiply-adds).  Per slot: 3 fma + 3 DPP adds (next slot's J'.acc), 2 fma + 1 sub + clamp + sub
  // + swap + fma + clamp + sub + swap (the row pair), 2 packed + 2 scalar fma (accumulators) = 21-22 VALU instructions.
  auto fm = [](auto a, auto b, auto c) { return __builtin_elementwise_fma(a, b, c); };
  T J0[n], J1[n], J2[n], B2[n], X2[n], rh[n], cp[n], lm[n], eo[n], ex[n];
  P2 B01[n], X01[n];
  T th[EXIT ? n : 1];                            // (K7) |delta| above which a row keeps its team iterating (cone friction rows: its reciprocal)
  T thx[(EXIT && CONE && NFS > 0) ? NFS : 1];    // (K7, cone) the partner row's reciprocal threshold
  T lmo[NNS > 0 ? NNS : 1];                      // the partner half's impulse of the normal slots (friction bounds)
  P2 a01 = {T(0), T(0)}; T a2 = T(0);
  // Set-up in two passes.  Pass 1 issues EVERY LDS read of every slot -- unconditionally, from a safe row where the team has none, so that
  // there is no branch between them -- and pass 2 masks and combines them.  Written slot by slot (round 2) the reads of a slot sat behind
  // branches and the previous slot's DPP reductions: five to six separate LDS round trips per slot, ~70 per call for a 13-slot sweep --
  // 1.5 us per sub-step on exactly the wavefronts that set the launch time.
  int rown[n], roth[n];
  T lego[n], legx[n];                            // (leg ids as stored: converted in pass 2, a conversion here would wait for its read)
  T ju[n], bu[n], jv[n], bv[n], jw[n], bw[n], xu[n], xv[n], xw[n], rhs_[n];
  // In groups of SETUP_GROUP slots (all 13 at once would need ~210 registers for the raw values alone): reads of the group, a compiler
  // barrier for memory operations, a second one that re-defines the read values (an empty asm with "+v" operands -- without it the
  // scheduler hoists each slot's arithmetic, which is not a memory operation, back up to its reads and the round trips are per slot again).
  constexpr int SETUP_GROUP = sizeof(T) == 4 ? SOLO_SETUP_GROUP_F32 : SOLO_SETUP_GROUP_F64;      // (fp64: a value is a register pair -- groups of 5 / 2 / 1 slots: 1.37 / 1.36 / 1.21 ms per step)
  static_for<(n + SETUP_GROUP - 1) / SETUP_GROUP>([&](auto gc) {
    constexpr int g0 = decltype(gc)::value * SETUP_GROUP, gn = (n - g0 < SETUP_GROUP ? n - g0 : SETUP_GROUP);
    static_for<gn>([&](auto jc) {
      constexpr int i = g0 + decltype(jc)::value;
      constexpr int k = Ord::slot(i);
      constexpr bool fr = i >= LIM + NNS;     // friction slot: its two rows are the two directions of ONE contact -- consecutive rows, present together, same leg
      rown[i] = TRW::row_of(2 * k + half, nlt, nc);
      if constexpr (fr) roth[i] = rown[i] < 0 ? -1 : rown[i] + 1 - 2 * half;
      else roth[i] = TRW::row_of(2 * k + 1 - half, nlt, nc);
      const int ro = rown[i] < 0 ? 0 : rown[i], rx = roth[i] < 0 ? 0 : roth[i];        // (a safe row to read; the value is discarded)
      const unsigned bo = __umul24((unsigned)ro, ROWB), bx = __umul24((unsigned)rx, ROWB);   // record address = row base + the lane's element offset (24-bit multiply: full rate)
      lego[i] = rd(bo, kLEG);
      if constexpr (!fr) legx[i] = rd(bx, kLEG); else legx[i] = T(0);
      { const P2 u = rd2(bo, kU), v = rd2(bo, kV), w = rd2(bo, kW);       // (J', B~) of the lane's three components: one 64-bit read each
        ju[i] = u.x; bu[i] = u.y; jv[i] = v.x; bv[i] = v.y; jw[i] = w.x; bw[i] = w.y; }
      xu[i] = rd(bx, kUB); xv[i] = rd(bx, kVB); xw[i] = rd(bx, kWB);
      rhs_[i] = rd(bo, kRHS);
      if constexpr (EXIT) th[i] = lds.A(ro, LDS::A_LAM);
      lm[i] = *reinterpret_cast<const T*>(solo_smem + lam_own + 2 * k * S_LAM);
    });
#ifndef SOLO_HOST_SHIM
    asm volatile("" ::: "memory");
    static_for<gn>([&](auto jc) {
      constexpr int i = g0 + decltype(jc)::value;
      asm volatile("" : "+v"(ju[i]), "+v"(bu[i]), "+v"(jv[i]), "+v"(bv[i]), "+v"(jw[i]), "+v"(bw[i]), "+v"(xu[i]), "+v"(xv[i]), "+v"(xw[i]),
                        "+v"(rhs_[i]), "+v"(lm[i]), "+v"(lego[i]));
      (void)legx;
      if constexpr (i < LIM + NNS) asm volatile("" : "+v"(legx[i]));
      (void)th;                       // (named outside the discarded branch: a generic lambda captures it only then)
      if constexpr (EXIT) asm volatile("" : "+v"(th[i]));
    });
#endif
    static_for<gn>([&](auto jc) {
      constexpr int i = g0 + decltype(jc)::value;
      constexpr bool fr = i >= LIM + NNS;
      const int r_own = rown[i], r_oth = roth[i];
      const int lo = r_own < 0 ? -3 : (int)lego[i], lx = fr ? lo : (r_oth < 0 ? -3 : (int)legx[i]);
      const bool uo = r_own >= 0 && (legU == -1 || legU == lo), vo = legV == lo, wo = legW == lo;
      const bool ux = fr ? uo : (r_oth >= 0 && (legU == -1 || legU == lx)), vx = fr ? vo : legV == lx, wx = fr ? wo : legW == lx;
      J0[i] = uo ? ju[i] : T(0); J1[i] = vo ? jv[i] : T(0); J2[i] = wo ? jw[i] : T(0);
      B01[i] = P2{uo ? bu[i] : T(0), vo ? bv[i] : T(0)}; B2[i] = wo ? bw[i] : T(0);
      X01[i] = P2{ux ? xu[i] : T(0), vx ? xv[i] : T(0)}; X2[i] = wx ? xw[i] : T(0);
      // rhs' rides in the reduction: every lane of a half starts its partial sum at -rhs'/8 (exact scaling), so the reduced
      // value is J'.acc - rhs' and the row update needs one subtraction instead of an add and a subtract
      rh[i] = r_own >= 0 ? rhs_[i] * T(-0.125) : T(0);
      // (th of a null row: that of the safe row 0, >= 0 -- a finished row's, or 0 as parked by phase_base_lead for a team without rows -- and a
      // null row's delta is 0, never above it)
      // coupling of the slot's two rows, c' = J'_{2k+1} . B~_{2k}: half 1 reduces it from its own J' and the
      // partner's B~; half 0 takes 0 (its row does not wait for anybody)
      if constexpr (CONE && fr) {             // the cone solves a contact's two directions side by side: no coupling between them
        cp[i] = T(0);
        (void)thx;
        if constexpr (EXIT) { if (r_own < 0) th[i] = T(0); thx[i - LIM - NNS] = half_swap(th[i]); }      // (a null row: the safe row's value is not a reciprocal)
      } else {
        const T c = team_red8(J0[i] * X01[i].x + J1[i] * X01[i].y + J2[i] * X2[i]);
        cp[i] = half ? c : T(0);
      }
      if constexpr (i >= LIM && i < LIM + NNS) lmo[i - LIM] = T(0);
    });
    // warm start: acc = sum over rows of B~ * lam0   (dV = M^-1 J^T lam0).  With PyBullet's default (no multibody warm start, the engine's
    // too) every lam0 read above is zero and the whole accumulation is skipped: one uniform branch per group.
    if (warm_on) {
      static_for<gn>([&](auto jc) {
        constexpr int i = g0 + decltype(jc)::value;
        const T lx_ = half_swap(lm[i]);
        if constexpr (i >= LIM && i < LIM + NNS) lmo[i - LIM] = lx_;
        a01 = fm(X01[i], P2{lx_, lx_}, fm(B01[i], P2{lm[i], lm[i]}, a01)); a2 = fm(X2[i], lx_, fm(B2[i], lm[i], a2));
      });
    }
  });
  // couplings with the predecessor slot (wrapping around: the first slot follows the last one of the previous sweep)
  static_for<n>([&](auto ic) {
    constexpr int i = decltype(ic)::value, p = (i + n - 1) % n;
    if constexpr (PIPE) {
      eo[i] = team_red8(J0[i] * B01[p].x + J1[i] * B01[p].y + J2[i] * B2[p]);
      ex[i] = team_red8(J0[i] * X01[p].x + J1[i] * X01[p].y + J2[i] * X2[p]);
    } else { eo[i] = T(0); ex[i] = T(0); (void)p; }
  });
  auto jdot = [&](int, T j0, T j1, T j2, T r8) -> T { return team_red8(fm(j0, a01.x, fm(j1, a01.y, fm(j2, a2, r8)))); };   // J'.acc - rhs'
  T dpre = jdot(0, J0[0], J1[0], J2[0], rh[0]);
  T delp = T(0), delxp = T(0);
  // impulses and accumulators back to LDS for the leader (half 0 holds the same accumulators as half 1)
  T* const hdr = lds.hdr();
#define SOLO_PGS_WRITE_BACK() do { \
    if (h == 0) { \
      static_for<n>([&](auto ic_) { \
        constexpr int i_ = decltype(ic_)::value; \
        *reinterpret_cast<T*>(solo_smem + lam_own + 2 * Ord::slot(i_) * S_LAM) = lm[i_]; \
      }); \
    } \
    if (half == 0) { \
      if (h < 6) hdr[(2 + h) * LN] = a01.x; else lds.y()[(h - 6) * LN] = a01.x; \
      lds.y()[(h + 2) * LN] = a01.y; \
      if (h < 2) lds.y()[(10 + h) * LN] = a2; \
    } } while (0)
  // K7 early exit, per env as in the reference (one Bullet world per env): a team whose sweep changed no row by more
  // than the threshold is finished -- its result is written back right then (it keeps sweeping along with its wavefront,
  // which costs nothing and is never read) -- and the wavefront leaves the loop once all four teams are finished.
  // Bookkeeping per sweep: the lanes' flags are OR-ed over the team with four DPP steps and one compare / branch decides whether
  // any team became quiet in this sweep (round 2 split the wave's ballot into four team masks on the scalar unit: ~25 instructions
  // per sweep, a twelfth of a 6-slot sweep; now ~9).
  bool fin = false;                                                   // this lane's team is finished (uniform within the team)
#if defined(SOLO_WAVE_TIMING) && defined(SOLO_DEAD_STATS)      // (its own switch: the wave-wide test in every friction slot slows the sweep by a tenth)
  unsigned long long fr_tot = 0, fr_dead = 0;                         // friction-slot visits / those with zero bound and zero impulse in every lane
#endif
  int it_done = 0; (void)it_done;
#pragma unroll 1                                                    // (unrolled by two: no change, measured)
  for (int it = 0; it < iterations; it++) {
#if defined(SOLO_WAVE_TIMING) && defined(SOLO_SWEEP_STATS)
    it_done = it + 1;
#endif
    bool viol = false;
    static_for<n>([&](auto ic) {
      constexpr int i = decltype(ic)::value, in = (i + 1) % n;
      constexpr bool fric = i >= LIM + NNS;
      // next slot's reduction on the accumulators as they are now (without this slot's update)
      T dnext = T(0), d;
      if constexpr (PIPE) {
        dnext = jdot(in, J0[in], J1[in], J2[in], rh[in]);
        d = fm(ex[i], delxp, fm(eo[i], delp, dpre));
      } else d = jdot(i, J0[i], J1[i], J2[i], rh[i]);     // plain in-order reduction
      T hi = T(0);
      if constexpr (fric) {   // impulse of this contact's normal row: normal slot c/2, half c&1
        constexpr int c = i - LIM - NNS;
        // fp32: one DPP broadcast from the lane that owns the normal row (row_newbcast: lane 0 of the team for even contacts, lane 8 for odd
        // ones) instead of a select between the own impulse and a running copy of the partner's (one add per normal slot and four registers less)
        if constexpr (sizeof(T) == 4) hi = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(lm[LIM + (c >> 1)]), 0x150 + 8 * (c & 1), 0xF, 0xF, true));
        else hi = (half == (c & 1)) ? lm[LIM + (c >> 1)] : lmo[c >> 1];
      }
      auto clampb = [&](T x) -> T {
        if constexpr (!fric) return x > T(0) ? x : T(0);
        else if constexpr (sizeof(T) == 4) return __builtin_amdgcn_fmed3f(x, -hi, hi);
        else return x < -hi ? -hi : (x > hi ? hi : x);
      };
#if defined(SOLO_WAVE_TIMING) && defined(SOLO_DEAD_STATS)
      if constexpr (fric) { fr_tot++; if (!__any(hi != T(0) || lm[i] != T(0))) fr_dead++; }
#endif
      T sp = lm[i] - d;
      T sv;
      if constexpr (CONE && fric) {
        // both directions' unclamped sums (own: sp, partner: spx), scaled onto the disc of radius hi = lambda_n (rows are in units of mu)
        const T spx = half_swap(sp);
        const T s2 = fm(sp, sp, fm(spx, spx, T(1e-30)));                       // (+ 1e-30: a pair at rest scales by hi * 1e15 -> 1, not by 0 * inf)
        T sc;
        if constexpr (sizeof(T) == 4) {
#ifndef SOLO_HOST_SHIM
          // v_rsq_f32 (1 ulp: the scale only matters at the level of the sweep's other roundings); min(scale, 1) as med3(x, 0, 1), which
          // the compiler folds into the multiply's clamp modifier (hi >= 0)
          sc = __builtin_amdgcn_fmed3f(hi * __builtin_amdgcn_rsqf(s2), 0.0f, 1.0f);
#else
          sc = hi / sqrt(s2); sc = sc < T(1) ? sc : T(1);
#endif
        } else { sc = hi / sqrt(s2); sc = sc < T(1) ? sc : T(1); }
        sv = sp * sc;
      } else {
        // position 2k (half 0) is final after the first clamp; 2k+1 (half 1) then sees its delta through c'
        // (cp = 0 in half 0, whose second clamp therefore repeats the first)
        const T dfirst = clampb(sp) - lm[i];
        sp = fm(-cp[i], half_swap(dfirst), sp);
        sv = clampb(sp);
      }
      const T del = sv - lm[i];
      lm[i] = sv;
      const T delx = half_swap(del);
      if constexpr (EXIT) {
        if constexpr (CONE && fric) viol = viol || (fabs(fm(delx, thx[i - LIM - NNS], del * th[i])) > T(1));
        else viol = viol || (fabs(del) > th[i]);
      }
      if constexpr (sizeof(T) != 4 && i >= LIM && i < LIM + NNS) lmo[i - LIM] += delx;
      a01 = fm(X01[i], P2{delx, delx}, fm(B01[i], P2{del, del}, a01));
      a2 = fm(X2[i], delx, fm(B2[i], del, a2));
      dpre = dnext; delp = del; delxp = delx;
    });
    if constexpr (EXIT) {
      const bool quiet = team_or16(viol ? 1 : 0) == 0;                 // no lane of this team saw a change above its row's threshold
      const bool nw = quiet && !fin;
      if (__any(nw)) {                                               // (wave-uniform branch, taken at most once per team)
        if (nw) SOLO_PGS_WRITE_BACK();
        fin = fin || quiet;
        if (__all(fin)) break;
      }
    }
  }
  if constexpr (EXIT) { if (!fin) SOLO_PGS_WRITE_BACK(); }     // teams that ran all the iterations
  else SOLO_PGS_WRITE_BACK();
#if defined(SOLO_WAVE_TIMING) && defined(SOLO_SWEEP_STATS)       // sweeps the wavefront ran / solves (instead of the dead-slot counters)
  if (threadIdx.x == 0 && blockIdx.x < SOLO_WT_WAVES) { solo_wave_times[blockIdx.x][26] += (unsigned long long)it_done; solo_wave_times[blockIdx.x][27] += 1ull; }
#endif
#if defined(SOLO_WAVE_TIMING) && defined(SOLO_DEAD_STATS)
  if (threadIdx.x == 0 && blockIdx.x < SOLO_WT_WAVES) { solo_wave_times[blockIdx.x][26] += fr_tot; solo_wave_times[blockIdx.x][27] += fr_dead; }
#endif
#undef SOLO_PGS_WRITE_BACK
}

template <typename T, int ROBOT, typename LDS>
SD void phase_pgs_team(int iterations_, bool warm_on, bool early_exit, bool pipe, bool cone, const LDS lds, int t) {
  const int iterations = (iterations_ & 0xFFFF) | (warm_on ? 1 << 30 : 0);       // (one argument register: pgs_team_variant)
  constexpr int LN = LDS::LANES;
  int nlt, nc, ncmax, anylim;
  team_counts<T, LDS>(lds, nlt, nc, ncmax, anylim);
#if defined(SOLO_WAVE_TIMING)
  if (threadIdx.x == 0 && blockIdx.x < SOLO_WT_WAVES) {
    solo_wave_times[blockIdx.x][4] += (unsigned long long)(ncmax + (ncmax + 1) / 2 + (anylim ? 1 : 0));   // slots swept
    if ((unsigned long long)ncmax > solo_wave_times[blockIdx.x][5]) solo_wave_times[blockIdx.x][5] = (unsigned long long)ncmax;
  }
#endif
  // Wave priority by the length of this sweep: with two wavefronts per SIMD (more than 4096 envs) the launch still ends with
  // its slowest wavefront, and VALU issue between co-resident waves is arbitrated by priority first -- the heavy wave should run
  // as if alone and the light one fill its gaps, not the reverse.
#ifndef SOLO_NO_SETPRIO
  if (ncmax >= 7) __builtin_amdgcn_s_setprio(3);
  else if (ncmax >= 5) __builtin_amdgcn_s_setprio(2);
  else if (ncmax >= 3) __builtin_amdgcn_s_setprio(1);
  else __builtin_amdgcn_s_setprio(0);
#endif
#define SOLO_SWEEP_C(N_, F_, C_) do { if (early_exit) { if (anylim) pgs_team_variant<T, LDS, 1, N_, F_, true, true, C_>(iterations, lds, t); \
                                                        else pgs_team_variant<T, LDS, 0, N_, F_, true, true, C_>(iterations, lds, t); } \
                                      else if (pipe) { if (anylim) pgs_team_variant<T, LDS, 1, N_, F_, false, true, C_>(iterations, lds, t); \
                                                       else pgs_team_variant<T, LDS, 0, N_, F_, false, true, C_>(iterations, lds, t); } \
                                      else { if (anylim) pgs_team_variant<T, LDS, 1, N_, F_, false, false, C_>(iterations, lds, t); \
                                             else pgs_team_variant<T, LDS, 0, N_, F_, false, false, C_>(iterations, lds, t); } } while (0)
#define SOLO_SWEEP_L(N_, F_) do { if (cone) SOLO_SWEEP_C(N_, F_, true); else SOLO_SWEEP_C(N_, F_, false); } while (0)
  switch (ncmax) {       // wave-uniform
    case 0:
      if (anylim) { if (early_exit) pgs_team_variant<T, LDS, 1, 0, 0, true>(iterations, lds, t); else pgs_team_variant<T, LDS, 1, 0, 0, false>(iterations, lds, t); }
      else {             // no rows at all: the accumulators the leader reads back are zero
        const int h = t & 7;
        T* const hdr = lds.hdr();
        if ((t >> 3) == 0) {
          if (h < 6) hdr[(2 + h) * LN] = T(0); else lds.y()[(h - 6) * LN] = T(0);
          lds.y()[(h + 2) * LN] = T(0);
          if (h < 2) lds.y()[(10 + h) * LN] = T(0);
        }
      }
      break;
    case 1: SOLO_SWEEP_L(1, 1); break;
    case 2: SOLO_SWEEP_L(1, 2); break;
    case 3: SOLO_SWEEP_L(2, 3); break;
    case 4: SOLO_SWEEP_L(2, 4); break;
    case 5: SOLO_SWEEP_L(3, 5); break;
    case 6: SOLO_SWEEP_L(3, 6); break;
    case 7: SOLO_SWEEP_L(4, 7); break;
    default: SOLO_SWEEP_L(4, 8); break;
  }
#undef SOLO_SWEEP_L
#undef SOLO_SWEEP_C
}

// apply + integrate, team mode: impulse cache and joints one per lane, base pose on the leader.
// Reads the sweep's results straight from the team arrays (accumulators in hdr / y, impulses in lam).
template <typename T, int ROBOT, typename LDS, typename CH>
SNI_SCALAR void phase_integrate_team(CH ch, const PhysParams<T> pp, T* lam_prev, unsigned nstride, const LDS lds, int t, bool valid,
                              bool lead) {
  using RB = Robot<ROBOT>;
  using TRW = TeamRows<T, LDS>;
  constexpr int NJ = RB::NJ, LN = LDS::LANES;
  if (!valid) return;
  SubCtx<T, ROBOT>& C = ch.get();
  const T dt = pp.dt;
  // LOAD everything this lane needs (two LDS round trips: the context, then the impulses whose positions depend on the contact mask)
  // -> compute -> STORE: a read behind a store is a round trip of its own (LDS pointers may alias as far as the compiler knows), and this
  // phase used to make thirteen of them
  const int mask = C.mask;
  const int nlt = C.nlim_total;
  const T* hdr = lds.hdr();
  const SV<T> w{{hdr[2 * LN], hdr[3 * LN], hdr[4 * LN]}, {hdr[5 * LN], hdr[6 * LN], hdr[7 * LN]}};
  const int tj = t < RB::NQ ? t : 0;                 // (every lane reads a valid joint's operands; lanes >= NQ discard them)
  const int L = tj / NJ, k = tj - L * NJ;
  const T qds = C.qds[tj], yk = lds.y()[(L * 3 + k) * LN], qj = C.ps.q[tj];
  const SV<T> Gk = C.LR[L].G[k];
  const SV<T> ub = C.ub;
  const V3<T> pos = C.ps.pos;
  const T x = C.ps.qx, yq = C.ps.qy, z = C.ps.qz, ww = C.ps.qw;
  const T* lam = TRW::lam(lds.lane);
  const int npos0 = TRW::NPOS0 + extra_limits(nlt);
  T lv[(NPRIM + 15) / 16];
#pragma unroll
  for (int i = 0; i < (NPRIM + 15) / 16; i++) {      // warm-start cache: normal impulse of primitive p (its rank among the contacts)
    const int p = t + 16 * i;
    const int pp_ = p < NPRIM ? p : 0;
    lv[i] = lam[(npos0 + __builtin_popcount(mask & ((1 << pp_) - 1))) * 4];
    lv[i] = (p < NPRIM && ((mask >> pp_) & 1)) ? lv[i] : T(0);
  }
  // ---- compute
  const T nvj = clampv(qds + yk - dot(Gk, w), pp.vmax);
  const V3<T> nw = mk(clampv(ub.a.x + w.a.x, pp.vmax), clampv(ub.a.y + w.a.y, pp.vmax), clampv(ub.a.z + w.a.z, pp.vmax));
  const V3<T> nv = mk(clampv(ub.l.x + w.l.x, pp.vmax), clampv(ub.l.y + w.l.y, pp.vmax), clampv(ub.l.z + w.l.z, pp.vmax));
  // ---- store
#pragma unroll
  for (int i = 0; i < (NPRIM + 15) / 16; i++) { const int p = t + 16 * i; if (p < NPRIM) C.lamp[p] = lv[i]; }
  if (t < RB::NQ) { C.ps.qd[t] = nvj; C.ps.q[t] = qj + dt * nvj; }
  if (lead) {
    C.ps.w = nw; C.ps.v = nv;
    C.ps.pos = fma3(nv, dt, pos);
    T wn2 = dot(nw, nw), wn = sqrt_fast(wn2), sc, s_, cw;
    sincos_t(T(0.5) * wn * dt, s_, cw);
    if (wn < T(1e-3)) sc = T(0.5) * dt - dt * dt * dt * T(1.0 / 48.0) * wn2; else sc = s_ * rcp_fast(wn);
    T ax = nw.x * sc, ay = nw.y * sc, az = nw.z * sc;
    T nx = cw * x + ax * ww + ay * z - az * yq;
    T ny = cw * yq - ax * z + ay * ww + az * x;
    T nz = cw * z + ax * yq - ay * x + az * ww;
    T nq = cw * ww - ax * x - ay * yq - az * z;
    T inv = rsqrt_fast(nx * nx + ny * ny + nz * nz + nq * nq);
    C.ps.qx = nx * inv; C.ps.qy = ny * inv; C.ps.qz = nz * inv; C.ps.qw = nq * inv;
  }
}

template <typename T, int ROBOT, typename LDS> struct TeamCtx {
  using type = CtxLds<T, ROBOT, TeamRows<T, LDS>::off_ctx>;
  static constexpr size_t bytes = TeamRows<T, LDS>::off_ctx + 4 * sizeof(SubCtx<T, ROBOT>);
};

// the leader's context C lives in LDS (see CtxLds); `C` is only dereferenced by leader lanes
template <typename T, int ROBOT, typename LDS>
SD int substep_team(const PhysParams<T> pp, T* lam_prev, unsigned nstride, const LDS& lds, int t, bool lead, bool valid) {
  constexpr int LN = LDS::LANES;
  using TRW = TeamRows<T, LDS>;
  using CH = typename TeamCtx<T, ROBOT, LDS>::type;
  const CH ch{lds.lane};
  SOLO_PT_DECL();
  SOLO_PT_BEGIN();
  if (valid && t < Robot<ROBOT>::NQ) {   // sin/cos of joint t on lane t
    SubCtx<T, ROBOT>& C = ch.get();
    T sn, cs;
    sincos_t(C.ps.q[t], sn, cs);
    C.sn[t] = sn; C.cs[t] = cs;
  }
  SOLO_PT(0);
  const bool ui = pp.urdf_inertia();     // (uniform) K2: URDF tensors instead of the box rule
  if (ui) phase_front_team<T, ROBOT, LDS, CH, true>(ch, pp, lds, t, valid, lead);
  else phase_front_team<T, ROBOT, LDS, CH, false>(ch, pp, lds, t, valid, lead);
  SOLO_PT(1);
#if SOLO_LEG_QUAD
  if (valid) {                                                                              // four legs on four quads (phase_leg_rt)
    if (ui) phase_leg_rt<T, ROBOT, LDS, CH, true>(ch, pp, lam_prev, nstride, lds, t >> 2, t & 3);
    else phase_leg_rt<T, ROBOT, LDS, CH, false>(ch, pp, lam_prev, nstride, lds, t >> 2, t & 3);
  }
#else
  if (valid && t < 4) {                                                                     // four legs on four lanes
    if (ui) phase_leg_rt<T, ROBOT, LDS, CH, true>(ch, pp, lam_prev, nstride, lds, t, 0);
    else phase_leg_rt<T, ROBOT, LDS, CH, false>(ch, pp, lam_prev, nstride, lds, t, 0);
  }
#endif
  SOLO_PT(2);
  if (valid) team_sum_base<T, ROBOT, CH>(ch, t);
  SOLO_PT(3);
  if (lead) phase_base_lead<T, ROBOT, LDS, CH>(ch, pp, lam_prev, nstride, lds);
  else if (!valid && t == 0) {   // idle team: no rows
    lds.hdr()[0] = T(0); lds.hdr()[LN] = T(0);
  }
  SOLO_PT(4);
  if (valid) phase_legrates_team<T, ROBOT, LDS, CH>(ch, pp, lds, t);
  SOLO_PT(5);
  phase_finish_team<T, ROBOT, LDS>(pp, lds, t);
  SOLO_PT(6);
  phase_pgs_team<T, ROBOT, LDS>(pp.iterations, pp.warm != T(0), pp.resid_thr >= T(0), pp.pgs_pipe(), pp.cone(), lds, t);
  SOLO_PT(7);
  phase_integrate_team<T, ROBOT, LDS, CH>(ch, pp, lam_prev, nstride, lds, t, valid, lead);
  SOLO_PT(8);
  return lead ? (ch.get().mask | strip_feet_bits(ch.get().smask)) : 0;
}
#endif  // !SOLO_HOST_SHIM

// ---------------------------------------------------------------- one physics sub-step
// C.ps / C.tau: state and the joint torques applied during this sub-step.  lam_prev: per-primitive
// warm-start impulses (global memory, stride = nstride).  Returns the contact bit mask.
template <typename T, int ROBOT, typename LDS>
SD int substep(SubCtx<T, ROBOT>& C, const PhysParams<T> pp, T* lam_prev, unsigned nstride, const LDS& lds) {
  using CH = CtxPriv<T, ROBOT>;
  const CH ch{&C};
  if (pp.urdf_inertia()) {
    phase_detect<T, ROBOT, CH, true>(ch, pp);
    phase_leg<T, ROBOT, 0, LDS, CH, true>(ch, pp, lam_prev, nstride, lds);
    phase_leg<T, ROBOT, 1, LDS, CH, true>(ch, pp, lam_prev, nstride, lds);
    phase_leg<T, ROBOT, 2, LDS, CH, true>(ch, pp, lam_prev, nstride, lds);
    phase_leg<T, ROBOT, 3, LDS, CH, true>(ch, pp, lam_prev, nstride, lds);
  } else {
    phase_detect<T, ROBOT, CH, false>(ch, pp);
    phase_leg<T, ROBOT, 0, LDS, CH, false>(ch, pp, lam_prev, nstride, lds);
    phase_leg<T, ROBOT, 1, LDS, CH, false>(ch, pp, lam_prev, nstride, lds);
    phase_leg<T, ROBOT, 2, LDS, CH, false>(ch, pp, lam_prev, nstride, lds);
    phase_leg<T, ROBOT, 3, LDS, CH, false>(ch, pp, lam_prev, nstride, lds);
  }
  phase_base<T, ROBOT, LDS, CH>(ch, pp, lam_prev, nstride, lds);
  if (pp.cone()) phase_pgs<T, ROBOT, LDS, CH, true>(ch, pp.iterations, pp.resid_thr, lds);
  else phase_pgs<T, ROBOT, LDS, CH>(ch, pp.iterations, pp.resid_thr, lds);
  phase_integrate<T, ROBOT, LDS, CH>(ch, pp, lam_prev, nstride, lds);
  return C.mask | strip_feet_bits(C.smask);
}

}  // namespace solo
