// solorl_hip.hip -- gfx950 kernels + the C ABI of include/solorl.h.
//
// Hot path: one launch of `step_kernel` = SoloBaseEnv.step for every env
// (reference baseEnv.py:42-68) INCLUDING the worker's auto-reset (agents/ppo/envs.py:38-40):
//   A3 apply_action (solo.py:224-259, controllers/PD.py:3-10)  -> A4 simulator_step (solo.py:261-274:
//   history push, frame_skip x physics sub-step, pointgoal progress) -> A6 observation
//   (solo.py:186-222) -> A7 reward (baseEnv.py:91-157) -> A8 termination (baseEnv.py:162-180)
//   -> A1 terminal reward override / info (baseEnv.py:52-66) -> A2/A9 reset (baseEnv.py:70-82).
//
// Layout: struct-of-arrays state in HBM (field-major).  Team mode (default): 16 lanes per env, 4 envs per
// wavefront, per-env context and constraint rows in LDS, the PGS sweep register-resident (dynamics.hpp).
// Lane mode (SOLORL_TEAM=0): one env per lane, 64 envs per workgroup, rows in LDS, the rest in VGPRs.
//
// Reset is O(1): `reset = fixed pose + K zero-torque control steps` is a pure function of K
// (5..11), so the post-settle states are simulated ONCE at create time by this same kernel
// (mode = SETTLE) into a snapshot table and a terminating lane just loads snapshot[K].
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <utility>
#include <string>
#include <vector>

#include "../../include/solorl.h"
#include "dynamics.hpp"

using namespace solo;

#ifndef SOLO_TU_PART
#define SOLO_TU_PART -1      // translation-unit split, see dispatch_step below
#endif
// the thread-local error message lives in the part that holds the C ABI; the other parts (and solorl_ppo.hip) report through it
extern "C" __attribute__((visibility("hidden"))) int solorl_fail_(int code, const char* msg);

namespace {

#if SOLO_TU_PART == -1 || SOLO_TU_PART == 4
thread_local std::string g_err;
int fail(int code, const std::string& msg) { g_err = msg; return code; }
#else
int fail(int code, const std::string& msg) { return solorl_fail_(code, msg.c_str()); }
#endif
#define HIP_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(SOLORL_ERR_HIP, std::string(#x ": ") + hipGetErrorString(e_)); } while (0)

constexpr int DMAX = SOLORL_STATE_MAX_OBS;   // 42
constexpr int HSTRIDE = (DMAX + 3) & ~3;     // fields per history level in the state storage: 44 (whole 64-B sectors, see SX)
using idx_t = unsigned;   // device-side element indices: 32-bit (solorl_create checks NF*N and O*N < 2^31); 64-bit index
                          // arithmetic made every state access ~10 VALU instructions

// field-major state layout (offsets in units of N elements)
struct Layout {
  int n, D, H, NF;
  int pos, quat, v, w, q, qd, lam, hist, goal, pot, prog, goals, egoals, dr, xyprev, tmy;
};
enum { I_TIMESTEP = 0, I_MASK = 1, I_RNG = 2, I_NEEDRESET = 3, I_ENVID = 4, NI = 5 };   // I_ENVID: slot -> env id

// State storage, "AoSoA-4": slots are grouped in fours (= the four envs of a team-mode wavefront) and a group holds its
// fields as [field][4], so a group is one contiguous block of NF x 16 B.  The team's lanes (field t, t+16, ... of slot
// e & 3) then read and write whole 64-B sectors: 4 sectors per wave instruction where the field-major layout of round 1
// touched 16, and its 16-B partial writes were counted (and paid for) as full 64-B write requests -- 11.0 MB of HBM writes
// per launch against 3.2 MB of bytes actually stored.  Snapshots use the same indexing over their own slot count.
SD idx_t SX(int f, idx_t e, int nf) { return (e >> 2) * (idx_t)(nf << 2) + ((idx_t)f << 2) + (e & 3u); }

Layout make_layout(int n, int D, int H) {
  Layout L; int o = 0;
  L.n = n; L.D = D; L.H = H;
  // every block starts on a multiple of 4 fields = a 64-B sector of the group (SX), and NF is one, so groups are sector-aligned
  auto align4 = [&]() { o = (o + 3) & ~3; };
  L.pos = o; o += 3; L.quat = o; o += 4; L.v = o; o += 3; L.w = o; o += 3;
  L.q = o; o += n; L.qd = o; o += n; align4(); L.lam = o; o += NPRIM; align4(); L.hist = o; o += (H > 2 ? H : 2) * HSTRIDE; align4();      // (levels 0 and 1 always have their place; deeper ones only when asked for)
  L.goal = o; o += 2; L.pot = o++; L.prog = o++; L.goals = o++; L.egoals = o++; L.dr = o; o += 5; L.xyprev = o; o += 2; L.tmy = o++;
  align4();
  L.NF = o;
  return L;
}

struct EnvParams {
  int task, control, frame_skip, episode_length, hold_torque, disable_termination;
  int settle_min, nsettle;
  int use_treadmill; float tm_offset;         // treadmill: side redrawn at reset (simulation.py:72-74)
  int linear_ids;                             // slot e holds env e (no contact-count sorting): no slot -> env-id load on the step's critical path
  int lds_poison_on; unsigned lds_poison;     // test hook, see step_body
  unsigned seed_lo, seed_hi; long long id0;
  double kp, kd, max_torque, reward_dt;
  const double* dyn;     // device block of the handle's MUTABLE parameters ([0] = goal_radius): a by-value copy would be frozen
                         // into a captured HIP graph (solorl_increment_curriculum between replays must be seen)
};
enum { DYN_GOAL_RADIUS = 0, NDYN = 4 };

enum { MODE_STEP = 0, MODE_SETTLE = 1 };

// solorl_step_act: Policy.act on the observation a step has just written, by the wavefront that wrote it (policy_tail_team)
struct PolicyTail {
  const float *cw0, *cb0, *cw1, *cb1, *cw2, *cb2, *aw0, *ab0, *aw1, *ab1, *mw, *mb, *logstd;
  const float* noise;                       // [N][A] standard-normal draw, or null (action = mean)
  float *value, *action, *logp;             // [N], [N][A], [N]; value == null: no policy tail in this launch
};

struct Outputs {
  float* obs; float* rew; unsigned char* done;
  unsigned char *timeout, *success, *nan_reset; int* ep_len; float *ep_rew, *goals, *dr0, *dr1, *dr2, *dr3, *dr4;
  float* ep_stats;     // [SOLORL_EPSTAT_FIELDS][N] finished-episode accumulators (include/solorl.h), or null
  float* tau;          // [N][A] applied joint torques of this step (include/solorl.h applied_torque), or null
  PolicyTail pol;
};

// ---------------------------------------------------------------- device helpers
SD void philox(unsigned k0, unsigned k1, unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned (&out)[4]) {
#pragma unroll
  for (int r = 0; r < 10; r++) {
    unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
    unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1;
    unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// Bullet getEulerZYX (K10), as returned by p.getEulerFromQuaternion: roll, pitch, yaw
template <typename T> SD void euler_zyx(T x, T y, T z, T w, T& roll, T& pitch, T& yaw) {
  T sqx = x * x, sqy = y * y, sqz = z * z, squ = w * w;
  T sarg = T(-2) * (x * z - w * y);
  if (sarg <= T(-0.99999)) { pitch = T(-0.5 * M_PI); roll = T(0); yaw = T(2) * atan2(x, -y); }
  else if (sarg >= T(0.99999)) { pitch = T(0.5 * M_PI); roll = T(0); yaw = T(2) * atan2(-x, y); }
  else {
    pitch = asin(sarg);
    roll = atan2(T(2) * (y * z + w * x), squ - sqx - sqy + sqz);
    yaw = atan2(T(2) * (x * y + w * z), squ + sqx - sqy - sqz);
  }
}

template <typename T, int NQ> struct Env {
  PhysState<T, NQ> ps;
  T goal[2], pot, prog, goals, egoals, dr[5], xyprev[2], tmy;
  int timestep, mask, rng;
};

template <typename T, int NQ>
SD void load_env(Env<T, NQ>& E, const T* sf, const int* si, const Layout& L, idx_t N, idx_t e) {
  auto F = [&](int f) { return sf[SX(f, e, L.NF)]; };
  E.ps.pos = mk(F(L.pos), F(L.pos + 1), F(L.pos + 2));
  E.ps.qx = F(L.quat); E.ps.qy = F(L.quat + 1); E.ps.qz = F(L.quat + 2); E.ps.qw = F(L.quat + 3);
  E.ps.v = mk(F(L.v), F(L.v + 1), F(L.v + 2)); E.ps.w = mk(F(L.w), F(L.w + 1), F(L.w + 2));
#pragma unroll
  for (int j = 0; j < NQ; j++) { E.ps.q[j] = F(L.q + j); E.ps.qd[j] = F(L.qd + j); }
  E.goal[0] = F(L.goal); E.goal[1] = F(L.goal + 1); E.pot = F(L.pot); E.prog = F(L.prog);
  E.goals = F(L.goals); E.egoals = F(L.egoals);
#pragma unroll
  for (int k = 0; k < 5; k++) E.dr[k] = F(L.dr + k);
  E.xyprev[0] = F(L.xyprev); E.xyprev[1] = F(L.xyprev + 1); E.tmy = F(L.tmy);
  E.timestep = si[SX(I_TIMESTEP, e, NI)]; E.mask = si[SX(I_MASK, e, NI)]; E.rng = si[SX(I_RNG, e, NI)];
}
template <typename T, int NQ>
SD void store_env(const Env<T, NQ>& E, T* sf, int* si, const Layout& L, idx_t N, idx_t e) {
  auto F = [&](int f, T v) { sf[SX(f, e, L.NF)] = v; };
  F(L.pos, E.ps.pos.x); F(L.pos + 1, E.ps.pos.y); F(L.pos + 2, E.ps.pos.z);
  F(L.quat, E.ps.qx); F(L.quat + 1, E.ps.qy); F(L.quat + 2, E.ps.qz); F(L.quat + 3, E.ps.qw);
  F(L.v, E.ps.v.x); F(L.v + 1, E.ps.v.y); F(L.v + 2, E.ps.v.z);
  F(L.w, E.ps.w.x); F(L.w + 1, E.ps.w.y); F(L.w + 2, E.ps.w.z);
#pragma unroll
  for (int j = 0; j < NQ; j++) { F(L.q + j, E.ps.q[j]); F(L.qd + j, E.ps.qd[j]); }
  F(L.goal, E.goal[0]); F(L.goal + 1, E.goal[1]); F(L.pot, E.pot); F(L.prog, E.prog);
  F(L.goals, E.goals); F(L.egoals, E.egoals);
#pragma unroll
  for (int k = 0; k < 5; k++) F(L.dr + k, E.dr[k]);
  F(L.xyprev, E.xyprev[0]); F(L.xyprev + 1, E.xyprev[1]); F(L.tmy, E.tmy);
  si[SX(I_TIMESTEP, e, NI)] = E.timestep; si[SX(I_MASK, e, NI)] = E.mask; si[SX(I_RNG, e, NI)] = E.rng;
}

// SoloBase.get_current_state (solo.py:198-222): D values, compile-time indices
template <typename T, int ROBOT>
SD void current_state(const Env<T, Robot<ROBOT>::NQ>& E, int task, T (&cs)[DMAX]) {
  constexpr int NQ = Robot<ROBOT>::NQ;
  T r, p, y;
  euler_zyx(E.ps.qx, E.ps.qy, E.ps.qz, E.ps.qw, r, p, y);
  cs[0] = E.ps.pos.z;
  cs[1] = (r - T(2) * floor(r * T(0.5))) * T(0.5);   // (euler % 2*pi)/(2*pi) == (euler % 2)/2, solo.py:206
  cs[2] = (p - T(2) * floor(p * T(0.5))) * T(0.5);
  cs[3] = (y - T(2) * floor(y * T(0.5))) * T(0.5);
  cs[4] = E.ps.v.x; cs[5] = E.ps.v.y; cs[6] = E.ps.v.z;
  cs[7] = E.ps.w.x; cs[8] = E.ps.w.y; cs[9] = E.ps.w.z;
#pragma unroll
  for (int j = 0; j < NQ; j++) { cs[10 + j] = E.ps.q[j] * T(0.1); cs[10 + NQ + j] = E.ps.qd[j] * T(0.01); }   // (/10, /100: solo.py:208-209)
#pragma unroll
  for (int f = 0; f < 4; f++)    // solo.py:310-323: contacts with the plane body, which also lies under the treadmill strip -- a foot
    cs[10 + 2 * NQ + f] = ((E.mask >> (13 + 2 * f)) & 1) ? T(1) : T(0);     // over the strip (bit 24+f) is reported like any other
  if (task == SOLORL_TASK_POINTGOAL) {
    cs[14 + 2 * NQ] = E.ps.pos.x * T(0.5); cs[15 + 2 * NQ] = E.ps.pos.y * T(0.5);
    cs[16 + 2 * NQ] = E.goal[0] * T(0.5); cs[17 + 2 * NQ] = E.goal[1] * T(0.5);
  }
}

template <typename T, int NQ>
SD void sample_goal(Env<T, NQ>& E, const EnvParams& P, long long gid) {   // solo.py:325-330
  unsigned r[4];
  philox(P.seed_lo, P.seed_hi, (unsigned)gid, (unsigned)((unsigned long long)gid >> 32), (unsigned)E.rng++, 1u, r);
  T span = T(P.dyn[DYN_GOAL_RADIUS] - 1.0);
  T x = T(1) + T((double)(r[0] >> 8) * (1.0 / 16777216.0)) * span;
  T y = T(1) + T((double)(r[1] >> 8) * (1.0 / 16777216.0)) * span;
  E.goal[0] = (r[2] & 1u) ? x : -x; E.goal[1] = (r[3] & 1u) ? y : -y;
}

// SoloBaseEnv.reset (baseEnv.py:70-82) in O(1): load the pre-simulated post-settle state.
template <typename T, int ROBOT>
SD void reset_from_snapshot(Env<T, Robot<ROBOT>::NQ>& E, T* sf, const Layout& L, idx_t N, idx_t e, idx_t env, const T* snf,
                            const int* sni, int M, const EnvParams& P) {
  constexpr int NQ = Robot<ROBOT>::NQ;
  const long long gid = P.id0 + (long long)env;
  int rng = E.rng;
  T g0 = E.goal[0], g1 = E.goal[1];
  Env<T, NQ> S;
  // order of draws as the reference: treadmill side (scene.reset inside robot.reset, solo.py:166-168), goal (robot.reset),
  // then the settle count (env.reset)
  int side = 0;
  if (P.use_treadmill) {
    unsigned rt[4];
    philox(P.seed_lo, P.seed_hi, (unsigned)gid, (unsigned)((unsigned long long)gid >> 32), (unsigned)rng++, 3u, rt);
    side = (rt[0] & 1u) ? 0 : 1;               // 0: strip centred on +offset, 1: on -offset (second half of the snapshot table)
  }
  E.rng = rng;
  if (P.task == SOLORL_TASK_POINTGOAL) { sample_goal(E, P, gid); g0 = E.goal[0]; g1 = E.goal[1]; }
  rng = E.rng;
  unsigned r[4];
  philox(P.seed_lo, P.seed_hi, (unsigned)gid, (unsigned)((unsigned long long)gid >> 32), (unsigned)rng++, 2u, r);
  const int k = (int)(r[0] % (unsigned)P.nsettle) + side * P.nsettle;
  load_env(S, snf, sni, L, (idx_t)M, (idx_t)k);
  E = S;
  E.rng = rng; E.timestep = 0;
  E.goal[0] = g0; E.goal[1] = g1; E.goals = T(0); E.egoals = T(0);
#pragma unroll
  for (int q = 0; q < 5; q++) E.dr[q] = T(0);
  for (int p = 0; p < NPRIM; p++) sf[SX(L.lam + p, e, L.NF)] = snf[SX(L.lam + p, k, L.NF)];
  for (int h = 0; h < L.H; h++)
    for (int d = 0; d < L.D; d++) sf[SX(L.hist + h * HSTRIDE + d, e, L.NF)] = snf[SX(L.hist + h * HSTRIDE + d, k, L.NF)];
  if (P.task == SOLORL_TASK_POINTGOAL) {
    for (int h = 0; h < L.H; h++) {
      sf[SX(L.hist + h * HSTRIDE + L.D - 2, e, L.NF)] = g0 * T(0.5);
      sf[SX(L.hist + h * HSTRIDE + L.D - 1, e, L.NF)] = g1 * T(0.5);
    }
    T dx = E.ps.pos.x - g0, dy = E.ps.pos.y - g1, px = E.xyprev[0] - g0, py = E.xyprev[1] - g1;
    E.pot = sqrt(dx * dx + dy * dy);
    E.prog = -(E.pot - sqrt(px * px + py * py));
  }
}

template <typename T, int ROBOT>
SD void write_obs(const Env<T, Robot<ROBOT>::NQ>& E, const T* sf, const Layout& L, idx_t N, idx_t e, idx_t env, int task, float* obs) {
  T cs[DMAX];
  current_state<T, ROBOT>(E, task, cs);
  const int O = L.D * (1 + L.H);
  float* o = obs + env * (idx_t)O;
  // SoloBase.calc_state, solo.py:186-196: [s, s - hist_newest, s - hist_older].  All history loads are issued
  // before the first store: obs and the state may alias as far as the compiler knows, and a load -> wait ->
  // store chain per element cost 38 dependent memory round trips (13 us of a 210 us step).
  T hv[2][DMAX];
#pragma unroll
  for (int h = 0; h < 2; h++)
#pragma unroll
    for (int d = 0; d < DMAX; d++) hv[h][d] = (h < L.H && d < L.D) ? sf[SX(L.hist + h * HSTRIDE + d, e, L.NF)] : T(0);
#pragma unroll
  for (int d = 0; d < DMAX; d++)
    if (d < L.D) {
      o[d] = (float)cs[d];
#pragma unroll
      for (int h = 0; h < 2; h++)
        if (h < L.H) o[(h + 1) * L.D + d] = (float)(cs[d] - hv[h][d]);
      for (int h = 2; h < L.H; h++) o[(h + 1) * L.D + d] = (float)(cs[d] - sf[SX(L.hist + h * HSTRIDE + d, e, L.NF)]);    // (deeper levels: straight from HBM)
    }
}

// ---------------------------------------------------------------- lane mode (SOLORL_TEAM=0): one env per lane
// 64 envs per wavefront, the env in registers, rows in LDS.  Kept as an independent second implementation of the same
// step (GPU test: lane vs team vs sorted storage agree); the default is step_team below.
template <typename T, int ROBOT>
SD void step_body(T* __restrict__ sf, int* __restrict__ si, const T* __restrict__ snf, const int* __restrict__ sni, int M,
                  const Layout& L, int N, const EnvParams& P, const PhysParams<T>& pp, const float* __restrict__ actions,
                  const Outputs& out, int mode) {
  using RB = Robot<ROBOT>;
  constexpr int NQ = RB::NQ;
  using LDS = RowLds<T>;
  constexpr int EPB = LDS::LANES;          // envs per workgroup: compile-time (LDS immediates)
  const int col = threadIdx.x;
  idx_t e = (idx_t)blockIdx.x * EPB + col;
  if (e >= (idx_t)N) return;
  LDS lds; lds.lanes = EPB; lds.lane = col;
  Env<T, NQ> E;
  const idx_t env = (idx_t)si[SX(I_ENVID, e, NI)];   // state is kept sorted by contact count: slot e holds env `env`
  load_env(E, sf, si, L, (idx_t)N, e);

  // ---- A3 apply_action
  T tau[NQ], asq = T(0);
#pragma unroll
  for (int j = 0; j < NQ; j++) {
    T a = mode == MODE_STEP ? (T)actions[env * NQ + j] : T(0);
    asq += a * a;
    T c = a < T(-1) ? T(-1) : (a > T(1) ? T(1) : a);
    if (P.control == SOLORL_CONTROL_TORQUE) tau[j] = c * T(P.max_torque);
    else {
      T t = T(P.kp) * (c * pp.qlim - E.ps.q[j]) - T(P.kd) * E.ps.qd[j];
      tau[j] = clampv(t, T(P.max_torque));
    }
    if (mode != MODE_STEP) tau[j] = T(0);
    if (out.tau && mode == MODE_STEP) out.tau[env * NQ + j] = (float)tau[j];
  }

  // ---- A4 simulator_step: history push (pre-step state), frame_skip sub-steps
  if (L.H > 0) {
    T cs[DMAX], h0[DMAX];
    current_state<T, ROBOT>(E, P.task, cs);
    for (int lv = L.H - 1; lv >= 2; lv--)              // levels beyond the second (num_history_stack 3, 4) shift in HBM, oldest first
      for (int d = 0; d < L.D; d++) sf[SX(L.hist + lv * HSTRIDE + d, e, L.NF)] = sf[SX(L.hist + (lv - 1) * HSTRIDE + d, e, L.NF)];
#pragma unroll
    for (int d = 0; d < DMAX; d++) h0[d] = (L.H >= 2 && d < L.D) ? sf[SX(L.hist + d, e, L.NF)] : T(0);   // loads first (see write_obs)
#pragma unroll
    for (int d = 0; d < DMAX; d++)
      if (d < L.D) {
        if (L.H >= 2) sf[SX(L.hist + HSTRIDE + d, e, L.NF)] = h0[d];
        sf[SX(L.hist + d, e, L.NF)] = cs[d];
      }
  }
  E.xyprev[0] = E.ps.pos.x; E.xyprev[1] = E.ps.pos.y;
  {
    SubCtx<T, ROBOT> C;
    C.ps = E.ps; C.tmy = E.tmy;
#pragma unroll 1
    for (int ss = 0; ss < P.frame_skip; ss++) {
      const T sc = (ss == 0 || P.hold_torque) ? T(1) : T(0);   // K8: Bullet clears applied torques every step
#pragma unroll
      for (int j = 0; j < NQ; j++) C.tau[j] = tau[j] * sc;
      E.mask = substep<T, ROBOT>(C, pp, sf + SX(L.lam, e, L.NF), 4u, lds);
    }
    E.ps = C.ps;
  }
  if (P.task == SOLORL_TASK_POINTGOAL && mode == MODE_STEP) {
    T dx = E.ps.pos.x - E.goal[0], dy = E.ps.pos.y - E.goal[1];
    T np = sqrt(dx * dx + dy * dy);
    E.prog = -(np - E.pot); E.pot = np;
    if (np < T(0.5)) { E.goals += T(1); sample_goal(E, P, P.id0 + (long long)env); }
  }
  if (mode == MODE_SETTLE) { store_env(E, sf, si, L, (idx_t)N, e); return; }
  E.timestep += 1;

  // ---- A7 reward (baseEnv.py:91-157)
  const T z = E.ps.pos.z;
  T stand = z > T(0.2) ? T(0.5) : T(0), jp = T(0), balance = T(0), progress = T(0), torque = T(0);
#pragma unroll
  for (int j = 0; j < NQ; j++) jp += P.task == SOLORL_TASK_STAND ? fabs(E.ps.q[j]) : E.ps.q[j] * E.ps.q[j];
  jp = T(-0.1) * jp / T(NQ);
  if (P.task == SOLORL_TASK_WALK) {
    if (z > T(0.2)) { T vx = E.ps.v.x; progress = T(2) * (vx > T(0) ? T(1) : (vx < T(0) ? T(-1) : T(0))) * vx * vx; }
  } else if (P.task == SOLORL_TASK_POINTGOAL) {
    T r, p, y; euler_zyx(E.ps.qx, E.ps.qy, E.ps.qz, E.ps.qw, r, p, y);
    balance = T(-0.1) * (fabs(r) + fabs(p));
    if (z > T(0.2)) progress = E.prog * T(1.0 / P.reward_dt);
  }
  if (P.control == SOLORL_CONTROL_TORQUE) torque = T(-0.01) * asq;
  T reward = stand + jp + balance + progress + torque;
  E.dr[0] += stand; E.dr[1] += jp; E.dr[2] += torque; E.dr[3] += balance; E.dr[4] += progress;

  // ---- A8 termination (baseEnv.py:162-180) + NaN guard
  int done = 0, to = 0, su = 0, nanr = 0;
  {
    T chk = E.ps.pos.x + E.ps.pos.y + E.ps.pos.z + E.ps.qw + E.ps.v.x + E.ps.v.y + E.ps.v.z + E.ps.w.x + E.ps.w.y + E.ps.w.z;
#pragma unroll
    for (int j = 0; j < NQ; j++) chk += E.ps.q[j] + E.ps.qd[j];
    if (!(fabs(chk) < T(1e30))) { nanr = 1; done = 1; reward = T(0); }
  }
  if (!P.disable_termination && !nanr) {
    if (E.timestep >= P.episode_length) { done = 1; to = 1; su = P.task != SOLORL_TASK_POINTGOAL; }
    else if (z < T(0.05)) { done = 1; }
    else if (P.task == SOLORL_TASK_POINTGOAL && E.goals > E.egoals) { E.egoals = E.goals; done = 1; su = 1; }
  }
  if (done && !nanr) {   // baseEnv.py:52-60
    if (su) { if (P.task == SOLORL_TASK_POINTGOAL) reward = T(0.1) * T(P.episode_length - E.timestep); }
    else if (!to) reward = T(-10);
  }
  out.rew[env] = (float)reward;
  out.done[env] = (unsigned char)done;
  if (out.timeout) out.timeout[env] = (unsigned char)to;
  if (out.success) out.success[env] = (unsigned char)su;
  if (out.nan_reset) out.nan_reset[env] = (unsigned char)nanr;
  if (out.ep_len) out.ep_len[env] = E.timestep;
  if (out.ep_rew) out.ep_rew[env] = (float)reward;
  if (out.goals) out.goals[env] = (float)E.egoals;
  if (out.dr0) out.dr0[env] = (float)E.dr[0];
  if (out.dr1) out.dr1[env] = (float)E.dr[1];
  if (out.dr2) out.dr2[env] = (float)E.dr[2];
  if (out.dr3) out.dr3[env] = (float)E.dr[3];
  if (out.dr4) out.dr4[env] = (float)E.dr[4];
  if (done && out.ep_stats) {      // finished-episode accumulators, one column per env (agents/ppo/train.py:90-100)
    float* s = out.ep_stats + env;
    const idx_t NN = (idx_t)N;
    if (nanr) s[9 * NN] += 1.0f;
    else {
      s[0] += 1.0f; s[NN] += (float)reward; s[2 * NN] += (float)E.timestep; s[3 * NN] += (float)su;
#pragma unroll
      for (int k = 0; k < 5; k++) s[(4 + k) * NN] += (float)E.dr[k];
    }
  }

  // ---- auto-reset (agents/ppo/envs.py:39) and observation
  if (done) reset_from_snapshot<T, ROBOT>(E, sf, L, (idx_t)N, e, env, snf, sni, M, P);
  write_obs<T, ROBOT>(E, sf, L, (idx_t)N, e, env, P.task, out.obs);
  store_env(E, sf, si, L, (idx_t)N, e);
}

#ifdef SOLO_WAVE_TIMING
#define WT_STAMP(i) do { wts_[i] = clock64(); } while (0)
#else
#define WT_STAMP(i) do {} while (0)
#endif

// ---------------------------------------------------------------- team mode (default): the whole step on 16 lanes per env
// sum over the 16 lanes of a team (every lane ends with the total)
template <typename T> SD T team_sum16(T x) {
  if constexpr (sizeof(T) == 4) {
    x += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    x += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    x += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x141, 0xF, 0xF, true));   // row_half_mirror
    x += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x140, 0xF, 0xF, true));   // row_mirror
  } else {
    x += __shfl_xor(x, 1, 16); x += __shfl_xor(x, 2, 16); x += __shfl_xor(x, 4, 16); x += __shfl_xor(x, 8, 16);
  }
  return x;
}
enum { ER_GOAL = 0, ER_POT = 2, ER_PROG = 3, ER_GOALS = 4, ER_EGOALS = 5, ER_DR = 6, ER_XYPREV = 11, ER_TMY = 13, NER = 14 };   // SubCtx::erec
enum { IR_TIMESTEP = 0, IR_MASK = 1, IR_RNG = 2, IR_SNAP = 3, IR_DONE = 4 };                                                  // SubCtx::irec
template <typename T, int ROBOT> SD void env_from_lds(Env<T, Robot<ROBOT>::NQ>& E, const SubCtx<T, ROBOT>& C) {
  E.ps = C.ps;
  E.goal[0] = C.erec[ER_GOAL]; E.goal[1] = C.erec[ER_GOAL + 1]; E.pot = C.erec[ER_POT]; E.prog = C.erec[ER_PROG];
  E.goals = C.erec[ER_GOALS]; E.egoals = C.erec[ER_EGOALS];
#pragma unroll
  for (int k = 0; k < 5; k++) E.dr[k] = C.erec[ER_DR + k];
  E.xyprev[0] = C.erec[ER_XYPREV]; E.xyprev[1] = C.erec[ER_XYPREV + 1]; E.tmy = C.erec[ER_TMY];
  E.timestep = C.irec[IR_TIMESTEP]; E.mask = C.irec[IR_MASK]; E.rng = C.irec[IR_RNG];
}
template <typename T, int ROBOT> SD void scalars_to_lds(const Env<T, Robot<ROBOT>::NQ>& E, SubCtx<T, ROBOT>& C) {
  C.erec[ER_GOAL] = E.goal[0]; C.erec[ER_GOAL + 1] = E.goal[1]; C.erec[ER_POT] = E.pot; C.erec[ER_PROG] = E.prog;
  C.erec[ER_GOALS] = E.goals; C.erec[ER_EGOALS] = E.egoals;
#pragma unroll
  for (int k = 0; k < 5; k++) C.erec[ER_DR + k] = E.dr[k];
  C.erec[ER_XYPREV] = E.xyprev[0]; C.erec[ER_XYPREV + 1] = E.xyprev[1]; C.erec[ER_TMY] = E.tmy;
  C.irec[IR_TIMESTEP] = E.timestep; C.irec[IR_MASK] = E.mask; C.irec[IR_RNG] = E.rng;
}

// current_state (above) spread over the team: lane t produces elements t, t + 16, t + 32 straight from the LDS context -- one read and
// one multiply each; the three euler angles on lanes 1..3, every one of them running the leader's euler_zyx (same instructions, same
// bits) and keeping its own angle.  Replaces: the leader copying the whole context to registers (54 reads), computing the D values,
// staging them through LDS (D stores, a team sync, a read per lane).  Element map: current_state.
template <typename T, int ROBOT, int HK_>
SD void team_current_state(const SubCtx<T, ROBOT>& C, int task, int D, int t, T (&cs)[HK_]) {
  constexpr int NQ = Robot<ROBOT>::NQ;
  using PS_ = PhysState<T, NQ>;
  static_assert(offsetof(PS_, v) == 7 * sizeof(T) && offsetof(PS_, q) == 13 * sizeof(T) && sizeof(PS_) == (13 + 2 * NQ) * sizeof(T),
                "PhysState is a flat array: pos 0-2, quat 3-6, v 7-9, w 10-12, q, qd");
  const T* psv = reinterpret_cast<const T*>(&C.ps);
  const int mask = C.irec[IR_MASK];
  const T g0 = C.erec[ER_GOAL], g1 = C.erec[ER_GOAL + 1];
  T ang = T(0);
  if (t >= 1 && t <= 3) {
    T r, p, y;
    euler_zyx(C.ps.qx, C.ps.qy, C.ps.qz, C.ps.qw, r, p, y);
    const T a = t == 1 ? r : (t == 2 ? p : y);
    ang = (a - T(2) * floor(a * T(0.5))) * T(0.5);   // (euler % 2*pi)/(2*pi) == (euler % 2)/2, solo.py:206
  }
#pragma unroll
  for (int k = 0; k < HK_; k++) {
    const int d = t + 16 * k;
    const int e = d - (14 + 2 * NQ);                       // pointgoal extras: pos.x, pos.y, goal x, goal y (all * 0.5)
    const int idx = d == 0 ? 2 : (d < 10 + 2 * NQ ? d + 3 : (e >= 0 && e < 2 ? e : 0));
    const T raw = psv[idx];
    const T sc = d < 10 ? T(1) : (d < 10 + NQ ? T(0.1) : (d < 10 + 2 * NQ ? T(0.01) : T(0.5)));     // (/10, /100: solo.py:208-209)
    T v = raw * sc;
    if (d >= 1 && d <= 3) v = ang;
    const int f = d - (10 + 2 * NQ);
    if (f >= 0 && f < 4) v = ((mask >> (13 + 2 * f)) & 1) ? T(1) : T(0);     // solo.py:310-323 (see current_state)
    if (e == 2) v = g0 * T(0.5);
    if (e == 3) v = g1 * T(0.5);
    if (task != SOLORL_TASK_POINTGOAL && e >= 0) v = T(0);
    cs[k] = d < D ? v : T(0);
  }
}

// One launch = SoloBaseEnv.step for every env, 16 lanes (one DPP row) per env, 4 envs per wavefront.  Everything on the
// wavefront's critical path is spread over the team's lanes: the env's state travels HBM <-> LDS with all 16 lanes (4 loads
// per lane instead of 61 on one), joint t's action, clip and PD torque are lane t's, history / observation / impulse cache /
// snapshot reset are moved by all lanes; only the scalar env logic (euler angles, reward, termination, Philox draws) runs on
// the team leader, on values it reads from LDS.  Nothing but the six history values a lane needs for the observation deltas
// stays in registers across the sub-steps (the round-1 version kept the env there: 56 spilled VGPRs per wavefront and step,
// 11 MB of scratch write-back per launch against 5.7 MB of algorithmic traffic).
// ---------------------------------------------------------------- Policy.act in the step kernel's tail (solorl_step_act)
// agents/ppo/policy.py:33-49 on the MLP of :62-81 (two tanh MLPs obs -> 64 -> 64, critic head -> 1, actor head -> A, state-independent
// log-std) for the env whose observation this team has just produced.  Why here: at 4096 envs the launch lasts as long as its slowest
// wavefront while the MEAN wavefront is done after half that time, and the rollout's next launch -- the policy --
// can only start when the launch has drained; computed by each wavefront for its own four envs the policy rides in that idle time and
// a rollout step is ONE launch.  Hidden layers: one unit per lane for the wavefront's four envs (pol_layer); activations are exchanged
// through the (now dead) row storage in LDS, weights come from L2 as float4 rows (PyTorch layout); heads: 16 lanes per env, lanes 0..A-1
// the actor head, lane 15 the critic head.  ~1500 instructions per wavefront (~3 us) against a 15 us kernel + launch gap.  O % 4 == 0
// (one history level).
constexpr float POL_HALF_LOG_2PI = 0.91893853320467274178f;
constexpr unsigned POL_ENV_BYTES = 2496;     // per-env slice of the row storage (26 rows x 24 values x 4 B): obs <= 96, h1 128, h2 128 floats
SD float pol_tanh(float x) { const float e = __expf(2.0f * x); return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + e); }   // as solorl_ppo.hip tanh_fast
typedef float nfloat4 __attribute__((ext_vector_type(4)));          // (native vector: HIP's float4 class cannot be read through an address-space pointer)
typedef __attribute__((address_space(1))) const nfloat4 gfloat4;
SD float pol_dot4(nfloat4 w, nfloat4 x, float acc) { return fmaf(w.x, x.x, fmaf(w.y, x.y, fmaf(w.z, x.z, fmaf(w.w, x.w, acc)))); }
// A lone wavefront has nothing to hide a load's latency behind, so the weight loads are issued in explicit BATCHES (16-20 float4 per
// batch) with a compiler barrier behind them that also re-defines the loaded values -- left to itself the scheduler kept ~8 loads in
// flight and waited for them one or two at a time (19 exposed L2 round trips in layer 1 alone: the tail took 29 us).
#define POL_PIN4(x) asm volatile("" : "+v"(x))
// Hidden layers, one UNIT per lane: lane L of the wavefront computes unit L of both nets for all FOUR envs of the wavefront -- a weight row
// is loaded once per wavefront and multiplied with four observation vectors (LDS broadcast reads) instead of once per env: 70 float4
// weight loads per lane for the two layers instead of 280.
// One batch of weights: KB input chunks (float4) of lane L's row of both nets, issued together; and its multiply-adds against the four envs'
// inputs.  policy_tail_team strings five batches into a software pipeline: the NEXT batch's loads are issued before the current batch's
// arithmetic, so only the first round trip is exposed.
template <int KB> struct PolBatch { nfloat4 c[KB], a[KB]; };
template <int KB>
SD void pol_load(PolBatch<KB>& B, const float* wc_, const float* wa_, int K4, int k0, int lane) {
  gfloat4* wc = (gfloat4*)(wc_ + (size_t)lane * 4 * K4);
  gfloat4* wa = (gfloat4*)(wa_ + (size_t)lane * 4 * K4);
#pragma unroll
  for (int kk = 0; kk < KB; kk++) {
    const int k = k0 + kk < K4 ? k0 + kk : K4 - 1;          // (past the end: the last chunk again, not accumulated)
    B.c[kk] = wc[k]; B.a[kk] = wa[k];
  }
  asm volatile("" ::: "memory");                            // issued HERE (the scheduler would otherwise move them to their uses)
}
template <int KB>
SD void pol_fma(PolBatch<KB>& B, int K4, int k0, unsigned xoff, unsigned xoff_a, float (&c)[4], float (&a)[4]) {
  asm volatile("" ::: "memory");
#pragma unroll
  for (int kk = 0; kk < KB; kk++) { POL_PIN4(B.c[kk]); POL_PIN4(B.a[kk]); }
#pragma unroll
  for (int kk = 0; kk < KB; kk++)
    if (k0 + kk < K4) {
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const nfloat4 xc = reinterpret_cast<const nfloat4*>(solo_smem + e * POL_ENV_BYTES + xoff)[k0 + kk];
        const nfloat4 xa = xoff_a == xoff ? xc : reinterpret_cast<const nfloat4*>(solo_smem + e * POL_ENV_BYTES + xoff_a)[k0 + kk];
        c[e] = pol_dot4(B.c[kk], xc, c[e]); a[e] = pol_dot4(B.a[kk], xa, a[e]);
      }
    }
}
SD void pol_act_store(const float (&c)[4], const float (&a)[4], unsigned yoff, int lane) {
#pragma unroll
  for (int e = 0; e < 4; e++) {
    float* y = reinterpret_cast<float*>(solo_smem + e * POL_ENV_BYTES + yoff);
    y[lane] = pol_tanh(c[e]); y[64 + lane] = pol_tanh(a[e]);
  }
}

template <int A>
SD void policy_tail_team(const PolicyTail& P, int O, int col, int t, idx_t env, bool valid) {
  float* const buf = reinterpret_cast<float*>(solo_smem + (unsigned)col * POL_ENV_BYTES);
  float* const h2 = buf + 224;              // per env: obs at 0 (<= 96 floats), h1 at 96 [critic 64 | actor 64], h2 at 224
  const int lane = threadIdx.x & 63, O4 = O >> 2;
  static_assert(A < 15, "the critic head sits on lane 15");
  const bool head = t < A || t == 15;
  constexpr unsigned H1 = 96u * 4u, H2 = 224u * 4u;
  float c[4], a[4];
  // layer 1 (both nets read the observation) in two batches, layer 2 (critic reads h1[0..63], actor h1[64..127]) in two, then the heads' rows
  PolBatch<11> w1a, w1b;                    // (two batches of 11 chunks: observations of up to 88 values; the widest is 84)
  PolBatch<8> w2a, w2b;
  nfloat4 whb[16];
  pol_load(w1a, P.cw0, P.aw0, O4, 0, lane);
  pol_load(w1b, P.cw0, P.aw0, O4, 11, lane);
#pragma unroll
  for (int e = 0; e < 4; e++) { c[e] = P.cb0[lane]; a[e] = P.ab0[lane]; }
  pol_fma(w1a, O4, 0, 0u, 0u, c, a);
  pol_load(w2a, P.cw1, P.aw1, 16, 0, lane);
  pol_fma(w1b, O4, 11, 0u, 0u, c, a);
  pol_act_store(c, a, H1, lane);
  pol_load(w2b, P.cw1, P.aw1, 16, 8, lane);
#pragma unroll
  for (int e = 0; e < 4; e++) { c[e] = P.cb1[lane]; a[e] = P.ab1[lane]; }
  TEAM_SYNC();
  pol_fma(w2a, 16, 0, H1, H1 + 256u, c, a);
  {
    gfloat4* wh = (gfloat4*)(t == 15 ? P.cw2 : P.mw + (size_t)(head ? t : 0) * 64);
#pragma unroll
    for (int k = 0; k < 16; k++) whb[k] = wh[k];
    asm volatile("" ::: "memory");
  }
  pol_fma(w2b, 16, 8, H1, H1 + 256u, c, a);
  pol_act_store(c, a, H2, lane);
  TEAM_SYNC();
  // heads: lane t < A -> mean_t (and the action, its log-prob term), lane 15 -> value
  float out = 0.f;
  {
    asm volatile("" ::: "memory");
#pragma unroll
    for (int k = 0; k < 16; k++) POL_PIN4(whb[k]);
    const nfloat4* x = reinterpret_cast<const nfloat4*>(t == 15 ? h2 : h2 + 64);
    float o0 = t == 15 ? P.cb2[0] : P.mb[head ? t : 0], o1 = 0.f;        // (two chains: the head is 64 dependent multiply-adds otherwise)
#pragma unroll
    for (int k = 0; k < 16; k += 2) { o0 = pol_dot4(whb[k], x[k], o0); o1 = pol_dot4(whb[k + 1], x[k + 1], o1); }
    out = head ? o0 + o1 : 0.f;
  }
  float lp = 0.f, act = out;
  if (t < A) {
    const float ls = P.logstd[t];
    const float nz = P.noise ? P.noise[(size_t)env * A + t] : 0.f;
    act = out + __expf(ls) * nz;                                   // policy.py:41-45: mean + std * noise (noise null: the mean)
    lp = -0.5f * nz * nz - ls - POL_HALF_LOG_2PI;                  // log N(action; mean, std) = sum over dims, z = noise
  }
  lp = team_sum16(lp);
  if (valid) {
    if (t < A) P.action[(size_t)env * A + t] = act;
    if (t == 15) P.value[env] = out;
    if (t == 0) P.logp[env] = lp;
  }
}

template <typename T, int ROBOT>
SD void step_team(T* __restrict__ sf, int* __restrict__ si, const T* __restrict__ snf, const int* __restrict__ sni, int M,
                  const Layout& L, int N, const EnvParams& P, const PhysParams<T>& pp, const float* __restrict__ actions,
                  const Outputs& out, int mode) {
  using RB = Robot<ROBOT>;
  constexpr int NQ = RB::NQ, NPS = 13 + 2 * NQ;
  using LDS = RowLds<T, 4>;
  using CH = typename TeamCtx<T, ROBOT, LDS>::type;
  static_assert(sizeof(PhysState<T, NQ>) == NPS * sizeof(T), "PhysState is the HBM field order pos, quat, v, w, q, qd");
#ifdef SOLO_WAVE_TIMING
  long long wts_[11];
  for (int i_ = 0; i_ < 11; i_++) wts_[i_] = 0;
  const long long wr0_ = wall_clock64();
  if (threadIdx.x < 10) solo::solo_pt_acc[threadIdx.x] = 0;
  __syncthreads();
  wts_[0] = clock64();
#endif
  const int t = threadIdx.x & 15, col = threadIdx.x >> 4;
  // a workgroup touches only 16 B of each state field, so eight consecutive workgroups share every 128-B line; workgroup
  // ids go round-robin over the 8 XCDs (each with its own L2), hence give every XCD one CONTIGUOUS eighth of the env range
  // instead of every eighth workgroup (grid is a multiple of 8).
  const unsigned blk = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  idx_t e = (idx_t)blk * 4 + col;
  const bool valid = e < (idx_t)N;
  if (!valid) e = (idx_t)N - 1;           // every lane stays alive for the wave-level exchanges
  const bool lead = valid && t == 0;
  const idx_t NN = (idx_t)N;
  LDS lds; lds.lanes = 4; lds.lane = col;
  // test hook (SOLORL_POISON_LDS=<word>, read at create): pre-fill the whole dynamic LDS.  Two runs with different fill words
  // (NaN vs 0) must agree bitwise -- anything read before it is written in the launch breaks that
  // (tests/test_parity_gpu.py::test_no_lds_read_before_write; this is how a lane hand-off race was found)
  if (P.lds_poison_on) {
    using TC = TeamCtx<T, ROBOT, LDS>;
    unsigned* w = reinterpret_cast<unsigned*>(solo_smem);
    for (unsigned k = threadIdx.x; k < TC::bytes / 4; k += 64) w[k] = P.lds_poison;
    __syncthreads();
  }
  const CH ch{col};
  SubCtx<T, ROBOT>& C = ch.get();
  T* const psv = reinterpret_cast<T*>(&C.ps);
  idx_t env = 0;
  // ---- the env's state, HBM -> LDS, all lanes (the state is stored by slot: slot e holds env `env`)
  T a_t = T(0);
  if (valid) {
    env = P.linear_ids ? e : (idx_t)si[SX(I_ENVID, e, NI)];      // (the action load below depends on it: one HBM round trip less)
    for (int f = t; f < NPS; f += 16) psv[f] = sf[SX(L.pos + f, e, L.NF)];
    if (t < NER) C.erec[t] = sf[SX(L.goal + t, e, L.NF)];
    if (t < 3) C.irec[t] = si[SX(t, e, NI)];
    for (int p = t; p < NPRIM; p += 16) C.lamp[p] = sf[SX(L.lam + p, e, L.NF)];
    if (t < NQ && mode == MODE_STEP) a_t = (T)actions[env * NQ + t];
  }
  static_assert(I_TIMESTEP == IR_TIMESTEP && I_MASK == IR_MASK && I_RNG == IR_RNG, "counter order");
  TEAM_SYNC();
  WT_STAMP(1);
  // ---- A3 apply_action (solo.py:224-259, controllers/PD.py:3-10): joint t on lane t
  if (valid && t < NQ) {
    const T c = a_t < T(-1) ? T(-1) : (a_t > T(1) ? T(1) : a_t);
    T tq;
    if (P.control == SOLORL_CONTROL_TORQUE) tq = c * T(P.max_torque);
    else tq = clampv(T(P.kp) * (c * pp.qlim - C.ps.q[t]) - T(P.kd) * C.ps.qd[t], T(P.max_torque));
    C.tau_base[t] = mode == MODE_STEP ? tq : T(0);
    if (out.tau && mode == MODE_STEP) out.tau[env * NQ + t] = (float)tq;
  }
  const T asq = team_sum16(a_t * a_t);        // sum of the raw, unclipped action squares (baseEnv.py:142-144)
  WT_STAMP(2);
  // ---- A4 simulator_step: history push (pre-step state).  The leader publishes the D state values through LDS (the `bc`
  // block is free outside the sub-steps) and lane t moves elements t, t+16, t+32 -- 64-B coalesced segments -- keeping what
  // it wrote (hk0 = new newest = pre-step state, hk1 = old newest) for the observation at the end.
  constexpr int HK = (DMAX + 15) / 16;
  T hk0[HK], hk1[HK];
#pragma unroll
  for (int k = 0; k < HK; k++) { hk0[k] = T(0); hk1[k] = T(0); }
  T csk[HK];
  if (L.H > 0) team_current_state<T, ROBOT, HK>(C, P.task, L.D, t, csk);
  TEAM_SYNC();                 // (the leader's stores below: behind every lane's reads of the pre-step context)
  if (lead) {
    C.erec[ER_XYPREV] = C.ps.pos.x; C.erec[ER_XYPREV + 1] = C.ps.pos.y;
    C.tmy = C.erec[ER_TMY];
  }
  if (valid && L.H > 0) {
#pragma unroll
    for (int k = 0; k < HK; k++) {
      const int d = t + 16 * k;
      if (d < L.D) {
        const T c = csk[k];
        for (int lv = L.H - 1; lv >= 2; lv--)          // levels beyond the second (num_history_stack 3, 4) shift in HBM, oldest first; same lane, same elements
          sf[SX(L.hist + lv * HSTRIDE + d, e, L.NF)] = sf[SX(L.hist + (lv - 1) * HSTRIDE + d, e, L.NF)];
        if (L.H >= 2) { hk1[k] = sf[SX(L.hist + d, e, L.NF)]; sf[SX(L.hist + HSTRIDE + d, e, L.NF)] = hk1[k]; }
        sf[SX(L.hist + d, e, L.NF)] = c;
        hk0[k] = c;
      } else if (d < ((L.D + 3) & ~3)) {               // (pad: the level's last sector is written whole)
        sf[SX(L.hist + d, e, L.NF)] = T(0);
        for (int lv = 1; lv < L.H; lv++) sf[SX(L.hist + lv * HSTRIDE + d, e, L.NF)] = T(0);
      }
    }
  }
  WT_STAMP(3);
  // ---- frame_skip physics sub-steps
#pragma unroll 1
  for (int ss = 0; ss < P.frame_skip; ss++) {
    const T sc = (ss == 0 || P.hold_torque) ? T(1) : T(0);   // K8: Bullet clears applied torques every step
    if (valid && t < NQ) C.tau[t] = C.tau_base[t] * sc;
    const int m = substep_team<T, ROBOT>(pp, sf + SX(L.lam, e, L.NF), 4u, lds, t, lead, valid);
    if (lead) C.irec[IR_MASK] = m;
  }
  WT_STAMP(4);
  if (valid) for (int p = t; p < NPRIM; p += 16) sf[SX(L.lam + p, e, L.NF)] = C.lamp[p];
  WT_STAMP(5);
  // ---- scalar env logic on the leader (values from LDS)
  if (lead) {
    Env<T, NQ> E;
    env_from_lds<T, ROBOT>(E, C);
    int done = 0;
    if (P.task == SOLORL_TASK_POINTGOAL && mode == MODE_STEP) {     // solo.py:266-272
      T dx = E.ps.pos.x - E.goal[0], dy = E.ps.pos.y - E.goal[1];
      T np = sqrt(dx * dx + dy * dy);
      E.prog = -(np - E.pot); E.pot = np;
      if (np < T(0.5)) { E.goals += T(1); sample_goal(E, P, P.id0 + (long long)env); }
    }
    if (mode == MODE_STEP) {
      E.timestep += 1;
      // ---- A7 reward (baseEnv.py:91-157)
      const T z = E.ps.pos.z;
      T stand = z > T(0.2) ? T(0.5) : T(0), jp = T(0), balance = T(0), progress = T(0), torque = T(0);
#pragma unroll
      for (int j = 0; j < NQ; j++) jp += P.task == SOLORL_TASK_STAND ? fabs(E.ps.q[j]) : E.ps.q[j] * E.ps.q[j];
      jp = T(-0.1) * jp / T(NQ);
      if (P.task == SOLORL_TASK_WALK) {
        if (z > T(0.2)) { T vx = E.ps.v.x; progress = T(2) * (vx > T(0) ? T(1) : (vx < T(0) ? T(-1) : T(0))) * vx * vx; }
      } else if (P.task == SOLORL_TASK_POINTGOAL) {
        T r, p, y; euler_zyx(E.ps.qx, E.ps.qy, E.ps.qz, E.ps.qw, r, p, y);
        balance = T(-0.1) * (fabs(r) + fabs(p));
        if (z > T(0.2)) progress = E.prog * T(1.0 / P.reward_dt);
      }
      if (P.control == SOLORL_CONTROL_TORQUE) torque = T(-0.01) * asq;
      T reward = stand + jp + balance + progress + torque;
      E.dr[0] += stand; E.dr[1] += jp; E.dr[2] += torque; E.dr[3] += balance; E.dr[4] += progress;
      // ---- A8 termination (baseEnv.py:162-180) + NaN guard
      int to = 0, su = 0, nanr = 0;
      {
        T chk = E.ps.pos.x + E.ps.pos.y + E.ps.pos.z + E.ps.qw + E.ps.v.x + E.ps.v.y + E.ps.v.z + E.ps.w.x + E.ps.w.y + E.ps.w.z;
#pragma unroll
        for (int j = 0; j < NQ; j++) chk += E.ps.q[j] + E.ps.qd[j];
        if (!(fabs(chk) < T(1e30))) { nanr = 1; done = 1; reward = T(0); }
      }
      if (!P.disable_termination && !nanr) {
        if (E.timestep >= P.episode_length) { done = 1; to = 1; su = P.task != SOLORL_TASK_POINTGOAL; }
        else if (z < T(0.05)) { done = 1; }
        else if (P.task == SOLORL_TASK_POINTGOAL && E.goals > E.egoals) { E.egoals = E.goals; done = 1; su = 1; }
      }
      if (done && !nanr) {   // baseEnv.py:52-60
        if (su) { if (P.task == SOLORL_TASK_POINTGOAL) reward = T(0.1) * T(P.episode_length - E.timestep); }
        else if (!to) reward = T(-10);
      }
      out.rew[env] = (float)reward;
      out.done[env] = (unsigned char)done;
      if (out.timeout) out.timeout[env] = (unsigned char)to;
      if (out.success) out.success[env] = (unsigned char)su;
      if (out.nan_reset) out.nan_reset[env] = (unsigned char)nanr;
      if (out.ep_len) out.ep_len[env] = E.timestep;
      if (out.ep_rew) out.ep_rew[env] = (float)reward;
      if (out.goals) out.goals[env] = (float)E.egoals;
      if (out.dr0) out.dr0[env] = (float)E.dr[0];
      if (out.dr1) out.dr1[env] = (float)E.dr[1];
      if (out.dr2) out.dr2[env] = (float)E.dr[2];
      if (out.dr3) out.dr3[env] = (float)E.dr[3];
      if (out.dr4) out.dr4[env] = (float)E.dr[4];
      if (done && out.ep_stats) {      // finished-episode accumulators, one column per env (agents/ppo/train.py:90-100)
        float* s = out.ep_stats + env;
        if (nanr) s[9 * NN] += 1.0f;
        else {
          s[0] += 1.0f; s[NN] += (float)reward; s[2 * NN] += (float)E.timestep; s[3 * NN] += (float)su;
#pragma unroll
          for (int k = 0; k < 5; k++) s[(4 + k) * NN] += (float)E.dr[k];
        }
      }
      // ---- auto-reset (agents/ppo/envs.py:39 -> baseEnv.py:70-82), part 1: the draws, in the reference's order -- treadmill
      // side (scene.reset inside robot.reset, solo.py:166-168), goal (robot.reset), settle count (env.reset)
      if (done) {
        const long long gid = P.id0 + (long long)env;
        int side = 0;
        if (P.use_treadmill) {
          unsigned rt[4];
          philox(P.seed_lo, P.seed_hi, (unsigned)gid, (unsigned)((unsigned long long)gid >> 32), (unsigned)E.rng++, 3u, rt);
          side = (rt[0] & 1u) ? 0 : 1;               // 0: strip centred on +offset, 1: on -offset (second half of the snapshot table)
        }
        if (P.task == SOLORL_TASK_POINTGOAL) sample_goal(E, P, gid);
        unsigned r[4];
        philox(P.seed_lo, P.seed_hi, (unsigned)gid, (unsigned)((unsigned long long)gid >> 32), (unsigned)E.rng++, 2u, r);
        C.irec[IR_SNAP] = (int)(r[0] % (unsigned)P.nsettle) + side * P.nsettle;
      }
    }
    scalars_to_lds<T, ROBOT>(E, C);          // (a reset keeps the goal and the rng counter from here)
    C.irec[IR_DONE] = done;
  }
  WT_STAMP(6);
  TEAM_SYNC();
  const int done = valid ? C.irec[IR_DONE] : 0;
  if (done) {
    // ---- auto-reset, part 2 = SoloBaseEnv.reset in O(1): the pre-simulated post-settle state (snapshot slot) replaces the env's
    // state, every field moved by the team -- HBM for all of them, LDS too for what the rest of this step reads
    const idx_t k = (idx_t)C.irec[IR_SNAP], MM = (idx_t)M;
    for (int f = t; f < L.NF; f += 16) {
      const T v = snf[SX(f, k, L.NF)];
      const bool scalar = f >= L.goal && f < L.goal + NER;
      if (!scalar) sf[SX(f, e, L.NF)] = v;                     // (the scalars are stored at the end, patched)
      if (f < NPS) psv[f] = v;
      else if (scalar && f != L.goal && f != L.goal + 1) C.erec[f - L.goal] = v;     // (the goal just drawn stays)
    }
    if (t == 1) C.irec[IR_MASK] = sni[SX(I_MASK, k, NI)];
#pragma unroll
    for (int kk = 0; kk < HK; kk++) {          // this lane's history elements, for the observation deltas
      const int d = t + 16 * kk;
      if (d < L.D) {
        if (L.H >= 1) hk0[kk] = snf[SX(L.hist + d, k, L.NF)];
        if (L.H >= 2) hk1[kk] = snf[SX(L.hist + HSTRIDE + d, k, L.NF)];
      }
    }
    TEAM_SYNC();
    const T g0 = C.erec[ER_GOAL], g1 = C.erec[ER_GOAL + 1];
    if (P.task == SOLORL_TASK_POINTGOAL) {     // the history keeps the NEW goal (set before the settle steps, solo.py:170-176)
#pragma unroll
      for (int kk = 0; kk < HK; kk++) {
        const int d = t + 16 * kk;
        if (d == L.D - 2 || d == L.D - 1) {
          const T gv = (d == L.D - 2 ? g0 : g1) * T(0.5);
          if (L.H >= 1) { hk0[kk] = gv; sf[SX(L.hist + d, e, L.NF)] = gv; }
          if (L.H >= 2) { hk1[kk] = gv; sf[SX(L.hist + HSTRIDE + d, e, L.NF)] = gv; }
          for (int lv = 2; lv < L.H; lv++) sf[SX(L.hist + lv * HSTRIDE + d, e, L.NF)] = gv;
        }
      }
    }
    if (t == 0) {
      C.irec[IR_TIMESTEP] = 0;
      C.erec[ER_GOALS] = T(0); C.erec[ER_EGOALS] = T(0);
#pragma unroll
      for (int q = 0; q < 5; q++) C.erec[ER_DR + q] = T(0);
      if (P.task == SOLORL_TASK_POINTGOAL) {
        const T dx = C.ps.pos.x - g0, dy = C.ps.pos.y - g1, px = C.erec[ER_XYPREV] - g0, py = C.erec[ER_XYPREV + 1] - g1;
        const T pot = sqrt(dx * dx + dy * dy);
        C.erec[ER_POT] = pot; C.erec[ER_PROG] = -(pot - sqrt(px * px + py * py));
      }
    }
  }
  WT_STAMP(7);
  // ---- A6 observation (solo.py:186-196): [s, s - hist_newest, s - hist_older], element d on lane d mod 16
  const bool want_obs = mode == MODE_STEP;
  TEAM_SYNC();                 // (a reset's state, written by all lanes above)
  T cso[HK];
  if (want_obs) team_current_state<T, ROBOT, HK>(C, P.task, L.D, t, cso);
  WT_STAMP(8);
  bool with_policy = false;
  if constexpr (sizeof(T) == 4) with_policy = want_obs && out.pol.value != nullptr;         // (uniform: a kernel argument)
  if (valid) {
    if (want_obs) {
      float* o = out.obs + env * (idx_t)(L.D * (1 + L.H));
      float* ol = reinterpret_cast<float*>(solo_smem + (unsigned)col * POL_ENV_BYTES);       // the policy tail's copy (row storage: dead by now)
#pragma unroll
      for (int k = 0; k < HK; k++) {
        const int d = t + 16 * k;
        if (d < L.D) {
          const T c = cso[k];
          o[d] = (float)c;
          if (L.H >= 1) o[L.D + d] = (float)(c - hk0[k]);
          if (L.H >= 2) o[2 * L.D + d] = (float)(c - hk1[k]);
          for (int lv = 2; lv < L.H; lv++) {
            // levels beyond the second live in HBM only.  An env that was just reset reads its snapshot (the team's copy above was
            // written by OTHER lanes) with the new goal patched in; any other env reads what this same lane shifted at the history push
            T hv;
            if (done) {
              hv = snf[SX(L.hist + lv * HSTRIDE + d, (idx_t)C.irec[IR_SNAP], L.NF)];
              if (P.task == SOLORL_TASK_POINTGOAL && d >= L.D - 2) hv = C.erec[ER_GOAL + (d - (L.D - 2))] * T(0.5);
            } else hv = sf[SX(L.hist + lv * HSTRIDE + d, e, L.NF)];
            o[(lv + 1) * L.D + d] = (float)(c - hv);
          }
          if (with_policy) {               // (only for obs sizes the policy kernels know: at most one history level)
            ol[d] = (float)c;
            if (L.H >= 1) ol[L.D + d] = (float)(c - hk0[k]);
          }
        }
      }
    }
    // ---- the env's state, LDS -> HBM, all lanes
    for (int f = t; f < ((NPS + 3) & ~3); f += 16) sf[SX(L.pos + f, e, L.NF)] = f < NPS ? psv[f] : T(0);     // (whole sectors:
    sf[SX(L.goal + t, e, L.NF)] = t < NER ? C.erec[t] : T(0);                                                 //  the pads are written too)
    if (t < 3) si[SX(t, e, NI)] = C.irec[t];
  }
  if constexpr (sizeof(T) == 4) {
    if (with_policy) {
      TEAM_SYNC();           // the observation copy, written by all lanes above
      policy_tail_team<NQ>(out.pol, L.D * (1 + L.H), col, t, env, valid);
    }
  }
  WT_STAMP(9);
#ifdef SOLO_WAVE_TIMING
  __builtin_amdgcn_s_waitcnt(0);     // (vmcnt = lgkmcnt = 0: the stores above have left)
  if (threadIdx.x == 0 && blockIdx.x < solo::SOLO_WT_WAVES) {
    wts_[10] = clock64();
    unsigned long long* w_ = solo::solo_wave_times[blockIdx.x];
    w_[0] = (unsigned long long)(wts_[3] - wts_[0]); w_[1] = (unsigned long long)(wts_[4] - wts_[3]); w_[2] = (unsigned long long)(wts_[10] - wts_[4]);
    w_[3] = (unsigned long long)(wall_clock64() - wr0_);
    for (int i_ = 0; i_ < 10; i_++) w_[6 + i_] = (unsigned long long)(wts_[i_ + 1] > wts_[i_] && wts_[i_] ? wts_[i_ + 1] - wts_[i_] : 0);
    for (int i_ = 0; i_ < 10; i_++) w_[16 + i_] = solo::solo_pt_acc[i_];       // per-phase sums over the sub-steps (SOLO_PT)
  }
#endif
}

template <typename T, int ROBOT>
__global__ void __launch_bounds__(64)
step_kernel(T* __restrict__ sf, int* __restrict__ si, const T* __restrict__ snf, const int* __restrict__ sni, int M,
            Layout L, int N, EnvParams P, PhysParams<T> pp, const float* __restrict__ actions, Outputs out, int mode) {
  step_body<T, ROBOT>(sf, si, snf, sni, M, L, N, P, pp, actions, out, mode);
}

// Two wavefronts per SIMD: the team-mode workgroup (one wavefront, 4 envs) needs 18.4 KB of LDS, so 8 fit a CU, and the
// register budget is held to 256 (the heaviest sweep variants then keep a few callee-saved registers in scratch instead
// of AGPRs).  A lone wavefront issues a VALU instruction every 4 cycles, the SIMD-32 one every 2: above 4096 envs per GPU
// the second wavefront is what fills the issue slots (8192 envs: 0.326 -> 0.27 ms per step; 65 536: 29.5 -> 48.8 M env-steps/s).
// The fp64 instantiation (the reference's arithmetic type; validation precision) is bounded to ONE wavefront per SIMD instead: its
// workgroup needs 37 KB of LDS, so only four fit a CU anyway, and with the 512-entry register budget (256 VGPRs + 256 AGPRs) a value
// that is a register PAIR stops spilling -- measured round 4 (profiles/r04_f64_ab.txt): 1.44 -> 0.55 ms per step at 4096 envs.
#ifndef SOLO_WAVES_PER_SIMD
#define SOLO_WAVES_PER_SIMD 2
#endif
template <typename T, int ROBOT>
__global__ void __launch_bounds__(64, sizeof(T) == 8 ? 1 : SOLO_WAVES_PER_SIMD)
step_kernel_team(T* __restrict__ sf, int* __restrict__ si, const T* __restrict__ snf, const int* __restrict__ sni, int M,
                 Layout L, int N, EnvParams P, PhysParams<T> pp, const float* __restrict__ actions, Outputs out, int mode) {
  step_team<T, ROBOT>(sf, si, snf, sni, M, L, N, P, pp, actions, out, mode);
}

template <typename T, int ROBOT>
__global__ void reset_kernel(T* sf, int* si, const T* snf, const int* sni, int M, Layout L, int N, EnvParams P, float* obs) {
  const idx_t e = (idx_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (idx_t)N) return;
  Env<T, Robot<ROBOT>::NQ> E;
  load_env(E, sf, si, L, (idx_t)N, e);
  const idx_t env = (idx_t)si[SX(I_ENVID, e, NI)];
  reset_from_snapshot<T, ROBOT>(E, sf, L, (idx_t)N, e, env, snf, sni, M, P);
  if (obs) write_obs<T, ROBOT>(E, sf, L, (idx_t)N, e, env, P.task, obs);
  store_env(E, sf, si, L, (idx_t)N, e);
  si[SX(I_NEEDRESET, e, NI)] = 0;
}

template <typename T, int ROBOT>
__global__ void obs_kernel(const T* sf, const int* si, Layout L, int N, int task, float* obs) {
  const idx_t e = (idx_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (idx_t)N) return;
  Env<T, Robot<ROBOT>::NQ> E;
  load_env(E, sf, si, L, (idx_t)N, e);
  write_obs<T, ROBOT>(E, sf, L, (idx_t)N, e, (idx_t)si[SX(I_ENVID, e, NI)], task, obs);
}

// initial pose of SoloBase.robot_specific_reset (solo.py:291-296) for every env of a buffer
template <typename T>
__global__ void init_pose_kernel(T* sf, int* si, Layout L, int N, T tmy) {
  const idx_t e = (idx_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (idx_t)N) return;
  for (int f = 0; f < L.NF; f++) sf[SX(f, e, L.NF)] = T(0);
  sf[SX(L.pos + 2, e, L.NF)] = T(0.35);
  sf[SX(L.quat + 3, e, L.NF)] = T(1);
  sf[SX(L.tmy, e, L.NF)] = tmy;
  for (int k = 0; k < NI; k++) si[SX(k, e, NI)] = 0;
  si[SX(I_NEEDRESET, e, NI)] = 1;
  si[SX(I_ENVID, e, NI)] = (int)e;
}

template <typename T>
__global__ void copy_env_kernel(const T* sf, const int* si, int N, int src, T* df, int* di, int M, int dst, int NF) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f < NF) df[SX(f, (idx_t)dst, NF)] = sf[SX(f, (idx_t)src, NF)];
  if (f < NI) di[SX(f, (idx_t)dst, NI)] = si[SX(f, (idx_t)src, NI)];
}

// ---- contact-count sorting (divergence control for the PGS sweep)
// A PGS sweep costs a wave the MAXIMUM row count over its lanes.  Under a random policy the mean is
// ~4 rows but the maximum over 64 arbitrary envs ~16, so the state is kept physically sorted by the
// last contact count: a stable counting sort (9 buckets, one workgroup) yields perm[dst] = src and a
// gather kernel moves every state field (coalesced writes).  Lanes are independent, so the
// permutation never changes results -- only which envs share a wavefront.
__global__ void __launch_bounds__(256) sort_perm_kernel(const int* __restrict__ si, int N, int* __restrict__ perm) {
  __shared__ int cnt[256][9];
  __shared__ int base[9];
  const int t = threadIdx.x, nt = blockDim.x;
  const int chunk = (N + nt - 1) / nt, lo = t * chunk, hi = lo + chunk < N ? lo + chunk : N;
  int c[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = lo; i < hi; i++) { int k = __popc(si[SX(I_MASK, (idx_t)i, NI)] & 0xFFFFFF); k = k > 8 ? 8 : k; c[k]++; }
#pragma unroll
  for (int k = 0; k < 9; k++) cnt[t][k] = c[k];
  __syncthreads();
  if (t < 9) {   // exclusive scan over threads for bucket t (serial: 256 adds)
    int run = 0;
    for (int j = 0; j < nt; j++) { int v = cnt[j][t]; cnt[j][t] = run; run += v; }
    base[t] = run;
  }
  __syncthreads();
  if (t == 0) { int run = 0; for (int k = 0; k < 9; k++) { int v = base[k]; base[k] = run; run += v; } }
  __syncthreads();
  int off[9];
#pragma unroll
  for (int k = 0; k < 9; k++) off[k] = base[k] + cnt[t][k];
  for (int i = lo; i < hi; i++) {
    int k = __popc(si[SX(I_MASK, (idx_t)i, NI)] & 0xFFFFFF); k = k > 8 ? 8 : k;
    int d = 0;
#pragma unroll
    for (int q = 0; q < 9; q++) if (q == k) d = off[q]++;
    perm[d] = i;
  }
}

template <typename T>
__global__ void gather_state_kernel(const T* __restrict__ sf, const int* __restrict__ si, const int* __restrict__ perm, int N,
                                    int NF, T* __restrict__ df, int* __restrict__ di) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= N) return;
  const int src = perm[j];
  const int f = blockIdx.y;
  if (f < NF) df[SX(f, (idx_t)j, NF)] = sf[SX(f, (idx_t)src, NF)];
  else di[SX(f - NF, (idx_t)j, NI)] = si[SX(f - NF, (idx_t)src, NI)];
}

// ---- fused GAE / discounted returns (agents/ppo/storage.py:35-55): one thread per env walks the rollout
// backwards; every access is coalesced across envs.  HBM-bound: 16 B per (t, env) sample.
__global__ void returns_kernel(const float* __restrict__ rew, float* __restrict__ val, const float* __restrict__ msk,
                               const float* __restrict__ nextv, float* __restrict__ ret, int T, int N, int use_gae,
                               float gamma, float lam) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const size_t S = (size_t)N;
  if (use_gae) {
    float vnext = nextv[n];
    val[(size_t)T * S + n] = vnext;
    float gae = 0.f;
#pragma unroll 8
    for (int t = T - 1; t >= 0; t--) {
      const float m = msk[(size_t)(t + 1) * S + n], v = val[(size_t)t * S + n];
      const float delta = rew[(size_t)t * S + n] + gamma * vnext * m - v;
      gae = delta + gamma * lam * m * gae;
      ret[(size_t)t * S + n] = gae + v;
      vnext = v;
    }
  } else {
    float r = nextv[n];
    ret[(size_t)T * S + n] = r;
#pragma unroll 8
    for (int t = T - 1; t >= 0; t--) {
      r = r * gamma * msk[(size_t)(t + 1) * S + n] + rew[(size_t)t * S + n];
      ret[(size_t)t * S + n] = r;
    }
  }
}

// ---- fused clipped-surrogate PPO loss (agents/ppo/ppo.py:52-74 + the diagonal-Gaussian log-prob / entropy of
// agents/ppo/policy.py:51-58,171-173): forward AND the gradients w.r.t. the network heads in one pass over a
// mini-batch.  One thread per sample; per-block partial sums (deterministic: reduced by the caller), layout
//   part[block][0..2] = sum value-loss, sum action-loss, sample count;  part[block][3 .. 3+A) = d loss / d logstd.
// Sub-gradient conventions are torch's: minimum/maximum split ties evenly, clamp passes the gradient on its
// closed interval -- with both branches equal on a tie the result is the plain derivative.
__global__ void __launch_bounds__(256)
ppo_loss_kernel(const float* __restrict__ mean, const float* __restrict__ logstd, const float* __restrict__ values,
                const float* __restrict__ action, const float* __restrict__ old_logp, const float* __restrict__ adv,
                const float* __restrict__ vpred, const float* __restrict__ ret, int m, int A, float clip, float value_coef,
                int clipped_value, float* __restrict__ g_mean, float* __restrict__ g_values, float* __restrict__ part) {
  extern __shared__ float red[];                  // [256][3 + A] would not fit for large A: reduce column by column
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool on = i < m;
  const float inv_m = 1.0f / (float)m;
  float vl = 0.f, al = 0.f, glp = 0.f;
  float lp = 0.f;
  if (on) {
    for (int j = 0; j < A; j++) {
      const float ls = logstd[j];
      const float z = (action[(size_t)i * A + j] - mean[(size_t)i * A + j]) * __expf(-ls);
      lp += -0.5f * z * z - ls - 0.91893853320467274178f;
    }
    const float ratio = __expf(lp - old_logp[i]);
    const float a = adv[i];
    const float s1 = ratio * a;
    const float rc = fminf(fmaxf(ratio, 1.0f - clip), 1.0f + clip);
    const float s2 = rc * a;
    al = -fminf(s1, s2);
    const float inside = (ratio >= 1.0f - clip && ratio <= 1.0f + clip) ? 1.0f : 0.0f;
    // d min(s1, s2) / d logp:  s1 branch = a*ratio, s2 branch = a*ratio*inside
    const float w = s1 < s2 ? 1.0f : (s1 > s2 ? inside : 0.5f * (1.0f + inside));
    glp = -a * ratio * w * inv_m;
    const float v = values[i], r = ret[i];
    const float u = v - r;
    float gv;
    if (clipped_value) {
      const float dvp = v - vpred[i];
      const float vc = vpred[i] + fminf(fmaxf(dvp, -clip), clip);
      const float wv = vc - r;
      const float ins = (dvp >= -clip && dvp <= clip) ? 1.0f : 0.0f;
      const float uu = u * u, ww = wv * wv;
      vl = 0.5f * fmaxf(uu, ww);
      gv = uu > ww ? u : (uu < ww ? wv * ins : 0.5f * (u + wv * ins));
    } else { vl = 0.5f * u * u; gv = u; }
    g_values[i] = value_coef * gv * inv_m;
  }
  // block reductions: value loss, action loss, then d/dlogstd per action dim (also writes d/dmean)
  auto block_sum = [&](float x) -> float {
    red[threadIdx.x] = x;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
    const float r = red[0];
    __syncthreads();
    return r;
  };
  float* P = part + (size_t)blockIdx.x * (3 + A);
  const float svl = block_sum(vl), sal = block_sum(al);
  if (threadIdx.x == 0) { P[0] = svl; P[1] = sal; P[2] = (float)(min(m - (int)(blockIdx.x * blockDim.x), (int)blockDim.x)); }
  for (int j = 0; j < A; j++) {
    float gls = 0.f;
    if (on) {
      const float e = __expf(-logstd[j]);
      const float z = (action[(size_t)i * A + j] - mean[(size_t)i * A + j]) * e;
      g_mean[(size_t)i * A + j] = glp * z * e;          // d logp / d mean_j = z / sigma
      gls = glp * (z * z - 1.0f);                        // d logp / d logstd_j = z^2 - 1
    }
    const float s = block_sum(gls);
    if (threadIdx.x == 0) P[3 + j] = s;
  }
}

}  // namespace

// ================================================================== host side
struct solorl_env {
  solorl_config cfg;
  int N = 0, device = 0, n = 0, D = 0, O = 0, M = 0;
  bool f64 = false, reset_called = false;
  size_t tsize = 4;
  Layout L;
  void* sf = nullptr; int* si = nullptr; void* snf = nullptr; int* sni = nullptr;
  void* sf2 = nullptr; int* si2 = nullptr; int* perm = nullptr;   // double buffer + permutation for contact-count sorting
  bool sort = true;
  uint64_t seed = 0; int64_t id0 = 0;
  double goal_radius = 2.0;
  double* dyn = nullptr;   // device copy of the mutable parameters (EnvParams::dyn)
  int epw = 64;   // envs per wavefront (lanes per workgroup)
  bool spread = true;
  bool team = false;   // 16 lanes per env (set at create: default true)
  int lds_poison_on = 0; unsigned lds_poison = 0;   // SOLORL_POISON_LDS test hook
  int simds = 1024;          // SIMDs of the device (4 per CU): grids up to this size have a SIMD per wavefront
  int pipe_override = -1;    // SOLORL_PGS_PIPE=0/1 (dev A/B)
};

namespace {

EnvParams make_env_params(const solorl_env* h) {
  EnvParams P;
  const solorl_config& c = h->cfg;
  P.task = c.task; P.control = c.control; P.frame_skip = c.frame_skip; P.episode_length = c.episode_length;
  P.hold_torque = c.hold_torque; P.disable_termination = c.disable_termination;
  P.settle_min = c.settle_min; P.nsettle = c.settle_max - c.settle_min + 1;
  P.seed_lo = (unsigned)h->seed; P.seed_hi = (unsigned)(h->seed >> 32); P.id0 = h->id0;
  P.kp = c.kp; P.kd = c.kd; P.max_torque = c.max_torque; P.reward_dt = c.reward_dt; P.dyn = h->dyn;
  P.lds_poison_on = h->lds_poison_on; P.lds_poison = h->lds_poison;
  P.use_treadmill = c.use_treadmill; P.tm_offset = (float)c.treadmill_offset;
  P.linear_ids = h->sort ? 0 : 1;
  return P;
}
template <typename T> PhysParams<T> make_phys(const solorl_config& c) {
  PhysParams<T> p;
  p.dt = (T)c.sim_dt; p.gravity = (T)c.gravity; p.erp = (T)c.erp; p.slop = (T)c.linear_slop; p.warm = (T)c.warmstart;
  p.damping = (T)c.damping; p.vmax = (T)c.max_velocity; p.qlim = (T)c.joint_limit; p.inv_dt = (T)(1.0 / c.sim_dt);
  p.iterations = c.solver_iterations;
  p.tm_hw = (T)c.treadmill_half_width; p.tm_mu = (T)c.treadmill_friction;
  p.resid_thr = c.solver_residual_threshold > 0 ? (T)sqrt(c.solver_residual_threshold) : T(-1);
  p.cerp = (T)c.contact_erp; p.cmargin = (T)c.collision_margin;
  p.set_mode(true, c.use_urdf_inertia != 0, c.friction_model == SOLORL_FRICTION_CONE, c.use_treadmill != 0);
  return p;
}

// The phase functions address the dynamic LDS from the constant SOLO_LDS_BASE (dynamics.hpp): true exactly while a step kernel's
// static LDS has that size.  Checked against the loaded code object, once per kernel and process.
static int check_lds_base(const void* kernel, bool& checked) {
  if (checked) return 0;
  hipFuncAttributes a;
  HIP_TRY(hipFuncGetAttributes(&a, kernel));
  if (a.sharedSizeBytes != (size_t)SOLO_LDS_BASE)
    return fail(SOLORL_ERR_HIP, "step kernel's static LDS size differs from SOLO_LDS_BASE: the phase functions' dynamic-LDS base constant does not hold "
                                "(a -DSOLO_WAVE_TIMING build runs team mode only)");
  checked = true;
  return 0;
}

template <typename T, int ROBOT>
int launch_step(solorl_env* h, T* sf, int* si, int N, const float* actions, const Outputs& out, int mode, hipStream_t st) {
  static bool lds_checked[2] = {false, false};
  if (int rc = check_lds_base(h->team ? reinterpret_cast<const void*>(step_kernel_team<T, ROBOT>) : reinterpret_cast<const void*>(step_kernel<T, ROBOT>),
                              lds_checked[h->team ? 1 : 0])) return rc;
  if (h->team) {
    auto kt = step_kernel_team<T, ROBOT>;
    dim3 grid((((N + 3) / 4) + 7) & ~7), block(64);      // multiple of 8: XCD-contiguous env ranges (step_body)
    using TeamLds = RowLds<T, 4>;
    const size_t team_smem = TeamCtx<T, ROBOT, TeamLds>::bytes;
    static_assert(sizeof(T) == 8 || TeamCtx<T, ROBOT, TeamLds>::bytes <= 20480, "team-mode LDS must allow 8 workgroups per CU (two wavefronts per SIMD)");
    PhysParams<T> pp = make_phys<T>(h->cfg);
    // Pipelined sweep while every wavefront has a SIMD to itself, plain sweep above that (pgs_team_variant: 8192 envs 0.226 ->
    // 0.210 ms per step).  The two are re-associations of the same sums, so an env's last bits depend on which side of
    // h->simds workgroups its batch is; SOLORL_PGS_PIPE=1 (or 0) pins one variant for every batch size.
    bool pipe = (int)grid.x <= h->simds;
    if (h->pipe_override >= 0) pipe = h->pipe_override != 0;
    pp.mode = (pp.mode & ~PhysParams<T>::M_PIPE) | (pipe ? PhysParams<T>::M_PIPE : 0);
    hipLaunchKernelGGL(kt, grid, block, team_smem, st, sf, si, (const T*)h->snf, (const int*)h->sni, h->M, h->L, N,
                       make_env_params(h), pp, actions, out, mode);
    HIP_TRY(hipGetLastError());
    return 0;
  }
  auto kern = step_kernel<T, ROBOT>;
  const int EPB = RowLds<T>::LANES;
  // small workgroups would be packed several to a CU (sharing its LDS pipe and issue slots) while other
  // CUs idle; asking for (almost) the whole LDS forces one workgroup per CU.
  size_t smem = RowLds<T>::bytes(EPB);
  if (h->spread && (N + EPB - 1) / EPB <= 256) smem = RowLds<T>::bytes(sizeof(T) == 8 ? 32 : 64);
  static bool attr_set[8] = {false, false, false, false, false, false, false, false};
  int dev = h->device & 7;
  if (!attr_set[dev]) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)RowLds<T>::bytes(sizeof(T) == 8 ? 32 : 64)));
    attr_set[dev] = true;
  }
  dim3 grid((N + EPB - 1) / EPB), block(EPB);
  hipLaunchKernelGGL(kern, grid, block, smem, st, sf, si, (const T*)h->snf, (const int*)h->sni, h->M, h->L, N,
                     make_env_params(h), make_phys<T>(h->cfg), actions, out, mode);
  HIP_TRY(hipGetLastError());
  return 0;
}

}  // namespace

// ---- translation-unit split (build time only; solorl_amd/build.py).  The step kernels' phase functions are what takes minutes to
// compile, once per (arithmetic type, robot): this file is compiled five times in parallel -- SOLO_TU_PART = 0..3 emit ONE
// launch_step instantiation each (part = 2 * f64 + solo12) behind a hidden C symbol, part 4 the C ABI and every other kernel --
// or once with SOLO_TU_PART undefined (everything; the -DSOLO_WAVE_TIMING dev build, whose device-side counters must live in one
// translation unit).  The types below the anonymous namespace are the same source in every part.
#define SOLO_LAUNCH_ARGS solorl_env* h, void* sf, int* si, int N, const float* actions, const void* out, int mode, void* st
extern "C" {
__attribute__((visibility("hidden"))) int solorl_launch_part0(SOLO_LAUNCH_ARGS);
__attribute__((visibility("hidden"))) int solorl_launch_part1(SOLO_LAUNCH_ARGS);
__attribute__((visibility("hidden"))) int solorl_launch_part2(SOLO_LAUNCH_ARGS);
__attribute__((visibility("hidden"))) int solorl_launch_part3(SOLO_LAUNCH_ARGS);
}
#define SOLO_DEFINE_PART(K_, T_, R_) \
  extern "C" int solorl_launch_part##K_(SOLO_LAUNCH_ARGS) { \
    return launch_step<T_, R_>(h, (T_*)sf, si, N, actions, *(const Outputs*)out, mode, (hipStream_t)st); }
#if SOLO_TU_PART == -1 || SOLO_TU_PART == 0
SOLO_DEFINE_PART(0, float, 0)
#endif
#if SOLO_TU_PART == -1 || SOLO_TU_PART == 1
SOLO_DEFINE_PART(1, float, 1)
#endif
#if SOLO_TU_PART == -1 || SOLO_TU_PART == 2
SOLO_DEFINE_PART(2, double, 0)
#endif
#if SOLO_TU_PART == -1 || SOLO_TU_PART == 3
SOLO_DEFINE_PART(3, double, 1)
#endif

#if SOLO_TU_PART == -1 || SOLO_TU_PART == 4        // ---- the C ABI and everything that is not a step kernel
namespace {

int dispatch_step(solorl_env* h, void* sf, int* si, int N, const float* actions, const Outputs& out, int mode, hipStream_t st) {
  const bool s12 = h->cfg.robot == SOLORL_ROBOT_SOLO12;
  if (!h->f64) return s12 ? solorl_launch_part1(h, sf, si, N, actions, &out, mode, st) : solorl_launch_part0(h, sf, si, N, actions, &out, mode, st);
  return s12 ? solorl_launch_part3(h, sf, si, N, actions, &out, mode, st) : solorl_launch_part2(h, sf, si, N, actions, &out, mode, st);
}

template <typename T> int build_snapshots_t(solorl_env* h) {
  // one scratch env simulated from the reset pose; its state after k = settle_min..settle_max
  // zero-torque control steps is copied into snapshot slot k - settle_min.
  // With the treadmill the settle depends on which side the strip lies: a second set of snapshots (slots nsettle..) for -offset.
  T* tf = nullptr; int* ti = nullptr;
  HIP_TRY(hipMalloc(&tf, sizeof(T) * h->L.NF * 4)); HIP_TRY(hipMalloc(&ti, sizeof(int) * NI * 4));   // (one slot group)
  Outputs none; memset(&none, 0, sizeof none);
  const int nsettle = h->cfg.settle_max - h->cfg.settle_min + 1;
  for (int side = 0; side < (h->cfg.use_treadmill ? 2 : 1); side++) {
    const T tmy = h->cfg.use_treadmill ? (T)(side == 0 ? h->cfg.treadmill_offset : -h->cfg.treadmill_offset) : T(0);
    hipLaunchKernelGGL(init_pose_kernel<T>, dim3(1), dim3(64), 0, 0, tf, ti, h->L, 1, tmy);
    for (int k = 1; k <= h->cfg.settle_max; k++) {
      int rc = dispatch_step(h, tf, ti, 1, nullptr, none, MODE_SETTLE, 0);
      if (rc) return rc;
      if (k >= h->cfg.settle_min)
        hipLaunchKernelGGL(copy_env_kernel<T>, dim3((h->L.NF + 63) / 64), dim3(64), 0, 0, (const T*)tf, (const int*)ti, 1, 0,
                           (T*)h->snf, h->sni, h->M, side * nsettle + k - h->cfg.settle_min, h->L.NF);
    }
  }
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipFree(tf)); HIP_TRY(hipFree(ti));
  return 0;
}

int check_cfg(const solorl_config* c) {
  if (!c) return fail(SOLORL_ERR_INVALID, "null config");
  if (c->robot != SOLORL_ROBOT_SOLO8 && c->robot != SOLORL_ROBOT_SOLO12) return fail(SOLORL_ERR_INVALID, "robot must be SOLO8 or SOLO12");
  if (c->task < 0 || c->task > 2) return fail(SOLORL_ERR_INVALID, "task must be stand/walk/pointgoal");
  if (c->control != SOLORL_CONTROL_TORQUE && c->control != SOLORL_CONTROL_PD) return fail(SOLORL_ERR_INVALID, "control must be torque or pd");
  if (c->frame_skip < 1 || c->frame_skip > 64) return fail(SOLORL_ERR_INVALID, "frame_skip out of range");
  if (c->num_history_stack < 0 || c->num_history_stack > SOLORL_STATE_MAX_HISTORY) return fail(SOLORL_ERR_INVALID, "num_history_stack must be 0..4");
  if (c->episode_length < 1) return fail(SOLORL_ERR_INVALID, "episode_length must be >= 1");
  if (c->settle_min < 0 || c->settle_max < c->settle_min || c->settle_max > 64) return fail(SOLORL_ERR_INVALID, "bad settle range");
  if (c->settle_min < c->num_history_stack) return fail(SOLORL_ERR_INVALID, "settle_min must be >= num_history_stack");
  if (c->solver_iterations < 1 || c->solver_iterations > 1000) return fail(SOLORL_ERR_INVALID, "solver_iterations out of range");
  if (!(c->sim_dt > 0) || !(c->goal_radius > 1.0)) return fail(SOLORL_ERR_INVALID, "sim_dt must be > 0 and goal_radius > 1");
  if (c->use_treadmill && !(c->treadmill_half_width > 0 && c->treadmill_friction >= 0)) return fail(SOLORL_ERR_INVALID, "bad treadmill parameters");
  if (!(c->solver_residual_threshold >= 0)) return fail(SOLORL_ERR_INVALID, "solver_residual_threshold must be >= 0");
  if (c->friction_model != SOLORL_FRICTION_PYRAMID && c->friction_model != SOLORL_FRICTION_CONE) return fail(SOLORL_ERR_INVALID, "friction_model must be pyramid or cone");
  if (!(c->contact_erp >= 0 && c->contact_erp <= 1) || !(c->erp >= 0 && c->erp <= 1)) return fail(SOLORL_ERR_INVALID, "erp / contact_erp must lie in [0, 1]");
  if (!(c->collision_margin >= 0 && c->collision_margin <= 0.01)) return fail(SOLORL_ERR_INVALID, "collision_margin must lie in [0, 0.01] m");
  if (c->precision != SOLORL_PRECISION_F32 && c->precision != SOLORL_PRECISION_F64) return fail(SOLORL_ERR_INVALID, "bad precision");
  return 0;
}

}  // namespace

extern "C" {

const char* solorl_last_error(void) { return g_err.c_str(); }
// (the library's other translation units report through the same thread-local message; not part of the ABI)
extern "C" int solorl_fail_(int code, const char* msg) { return fail(code, msg); }

#ifdef SOLO_WAVE_TIMING
extern "C" int solorl_debug_wave_times(unsigned long long* out, int nwaves, int reset) {    // [nwaves][SOLO_WT_FIELDS]
  if (nwaves > solo::SOLO_WT_WAVES) nwaves = solo::SOLO_WT_WAVES;
  if (hipDeviceSynchronize() != hipSuccess) return -3;
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(solo::solo_wave_times), (size_t)nwaves * solo::SOLO_WT_FIELDS * sizeof(unsigned long long)) != hipSuccess) return -3;
  if (reset) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(solo::solo_wave_times)) != hipSuccess || hipMemset(p, 0, sizeof(solo::solo_wave_times)) != hipSuccess) return -3;
  }
  return 0;
}
#endif

int solorl_compute_returns(const float* rewards, float* value_preds, const float* masks, const float* next_value,
                           float* returns, int T, int N, int use_gae, float gamma, float gae_lambda, int device_id,
                           void* stream) {
  if (!rewards || !value_preds || !masks || !next_value || !returns) return fail(SOLORL_ERR_INVALID, "null array argument");
  if (T < 1 || N < 1) return fail(SOLORL_ERR_INVALID, "T and N must be >= 1");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(SOLORL_ERR_NODEVICE, "no HIP device available (no CPU fallback)");
  if (device_id < 0 || device_id >= ndev) return fail(SOLORL_ERR_NODEVICE, "device_id out of range");
  HIP_TRY(hipSetDevice(device_id));
  hipLaunchKernelGGL(returns_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, rewards, value_preds, masks,
                     next_value, returns, T, N, use_gae, gamma, gae_lambda);
  HIP_TRY(hipGetLastError());
  return 0;
}
int solorl_ppo_loss(const float* mean, const float* logstd, const float* values, const float* action, const float* old_logp,
                    const float* adv, const float* vpred, const float* ret, int m, int A, float clip, float value_coef,
                    int clipped_value, float* grad_mean, float* grad_values, float* partials, int device_id, void* stream) {
  if (!mean || !logstd || !values || !action || !old_logp || !adv || !vpred || !ret || !grad_mean || !grad_values || !partials)
    return fail(SOLORL_ERR_INVALID, "null array argument");
  if (m < 1 || A < 1) return fail(SOLORL_ERR_INVALID, "m and A must be >= 1");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(SOLORL_ERR_NODEVICE, "no HIP device available (no CPU fallback)");
  if (device_id < 0 || device_id >= ndev) return fail(SOLORL_ERR_NODEVICE, "device_id out of range");
  HIP_TRY(hipSetDevice(device_id));
  hipLaunchKernelGGL(ppo_loss_kernel, dim3((m + 255) / 256), dim3(256), 256 * sizeof(float), (hipStream_t)stream, mean, logstd, values,
                     action, old_logp, adv, vpred, ret, m, A, clip, value_coef, clipped_value, grad_mean, grad_values, partials);
  HIP_TRY(hipGetLastError());
  return 0;
}
const char* solorl_version(void) { return "solorl-hip 0.4 (gfx950)"; }
int solorl_abi_version(void) { return SOLORL_ABI_VERSION; }

int solorl_default_config(solorl_config* c, int robot, int task) {
  if (!c) return fail(SOLORL_ERR_INVALID, "null config");
  memset(c, 0, sizeof *c);
  c->robot = robot; c->task = task; c->control = SOLORL_CONTROL_TORQUE;
  c->frame_skip = 4; c->episode_length = 400; c->num_history_stack = 0;
  c->solver_iterations = 50; c->settle_min = 5; c->settle_max = 11; c->precision = SOLORL_PRECISION_F32;
  c->kp = 5.0; c->kd = 0.2; c->max_torque = 3.0; c->sim_dt = 1.0 / 240.0; c->reward_dt = 1.0 / 60.0; c->gravity = 9.81;
  c->erp = 0.2; c->linear_slop = 1e-5; c->warmstart = 0.0; c->damping = 0.04; c->max_velocity = 100.0;
  c->joint_limit = 10.0; c->goal_radius = 2.0;
  c->use_treadmill = 0; c->treadmill_offset = 0.49; c->treadmill_half_width = 0.5; c->treadmill_friction = 0.5;
  c->solver_residual_threshold = 1e-7;     // PyBullet's solverResidualThreshold (K7), see include/solorl.h
  c->collision_margin = 0.001;
  c->friction_model = SOLORL_FRICTION_CONE; c->contact_erp = 0.08;      // [K] ledger, DESIGN.md section 3 (rounds 1-3: pyramid, 0.2)
  return 0;
}

int solorl_create(const solorl_config* cfg, int num_envs, int device_id, uint64_t seed, int64_t env_id_offset, solorl_env** out) {
  if (!out) return fail(SOLORL_ERR_INVALID, "null out pointer");
  *out = nullptr;
  int rc = check_cfg(cfg);
  if (rc) return rc;
  if (num_envs < 1) return fail(SOLORL_ERR_INVALID, "num_envs must be >= 1");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(SOLORL_ERR_NODEVICE, "no HIP device available (the engine has no CPU fallback)");
  if (device_id < 0 || device_id >= ndev) return fail(SOLORL_ERR_NODEVICE, "device_id out of range");
  HIP_TRY(hipSetDevice(device_id));
  solorl_env* h = new solorl_env();
  h->cfg = *cfg; h->N = num_envs; h->device = device_id; h->seed = seed; h->id0 = env_id_offset;
  h->n = cfg->robot == SOLORL_ROBOT_SOLO12 ? 12 : 8;
  h->D = 14 + 2 * h->n + (cfg->task == SOLORL_TASK_POINTGOAL ? 4 : 0);
  h->O = h->D * (1 + cfg->num_history_stack);
  h->f64 = cfg->precision == SOLORL_PRECISION_F64; h->tsize = h->f64 ? 8 : 4;
  h->L = make_layout(h->n, h->D, cfg->num_history_stack);
  h->M = (cfg->settle_max - cfg->settle_min + 1) * (cfg->use_treadmill ? 2 : 1);
  h->goal_radius = cfg->goal_radius;
  {
    // envs per wavefront.  Measured on MI355X (profiles/r01_notes.md): narrower waves do NOT help -- the
    // kernel's duration is the sequential PGS sweep of the heaviest env, which a narrower wave
    // still has to run, while more waves contend for the LDS/TA pipes -- so waves are kept full.
    int epw = 64;
    if (const char* ev = getenv("SOLORL_ENVS_PER_WAVE")) { int v = atoi(ev); if (v >= 1 && v <= 64) epw = v; }
    if (h->f64 && epw > 32) epw = 32;
    if (const char* ev = getenv("SOLORL_SPREAD")) h->spread = atoi(ev) != 0;
    h->team = true;   // measured (tools/dev/bench_n.py): team mode wins at every batch size, 1k .. 262k envs; lane mode: SOLORL_TEAM=0
    if (const char* ev = getenv("SOLORL_TEAM")) h->team = atoi(ev) != 0;
    if (const char* ev = getenv("SOLORL_PGS_PIPE")) h->pipe_override = atoi(ev) != 0;
    { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, device_id) == hipSuccess && prop.multiProcessorCount > 0) h->simds = 4 * prop.multiProcessorCount; }
    if (const char* ev = getenv("SOLORL_POISON_LDS")) { h->lds_poison_on = 1; h->lds_poison = (unsigned)strtoul(ev, nullptr, 0); }
    h->epw = epw;
  }
  auto cleanup = [&](int code) { solorl_destroy(h); return code; };
  const size_t Np = ((size_t)num_envs + 3) & ~(size_t)3, Mp = ((size_t)h->M + 3) & ~(size_t)3;      // slot groups of 4 (SX)
  if ((unsigned long long)h->L.NF * (unsigned long long)Np >= (1ull << 31) ||
      (unsigned long long)(h->L.D * (1 + h->L.H)) * (unsigned long long)num_envs >= (1ull << 31))
    return cleanup(fail(SOLORL_ERR_INVALID, "num_envs too large for 32-bit element indices"));
  if (hipMalloc(&h->sf, h->tsize * h->L.NF * Np) != hipSuccess) return cleanup(fail(SOLORL_ERR_HIP, "hipMalloc state"));
  if (hipMalloc(&h->si, sizeof(int) * NI * Np) != hipSuccess) return cleanup(fail(SOLORL_ERR_HIP, "hipMalloc istate"));
  if (hipMalloc(&h->snf, h->tsize * h->L.NF * Mp) != hipSuccess) return cleanup(fail(SOLORL_ERR_HIP, "hipMalloc snapshot"));
  if (hipMalloc(&h->sni, sizeof(int) * NI * Mp) != hipSuccess) return cleanup(fail(SOLORL_ERR_HIP, "hipMalloc isnapshot"));
  {
    double dyn0[NDYN] = {h->goal_radius, 0.0, 0.0, 0.0};
    if (hipMalloc(&h->dyn, sizeof dyn0) != hipSuccess) return cleanup(fail(SOLORL_ERR_HIP, "hipMalloc dyn"));
    if (hipMemcpy(h->dyn, dyn0, sizeof dyn0, hipMemcpyHostToDevice) != hipSuccess) return cleanup(fail(SOLORL_ERR_HIP, "hipMemcpy dyn"));
  }
  h->sort = false;   // measured with the team-mode sweep (tools/dev/bench_sort.sh): the sort + gather launches cost more than the
                     // padding they save at every batch size (65 536 envs: 29.5 M vs 23.8 M env-steps/s); SOLORL_SORT=1 enables it
  if (const char* ev = getenv("SOLORL_SORT")) h->sort = atoi(ev) != 0 && num_envs >= 2;
  if (h->sort) {
    if (hipMalloc(&h->sf2, h->tsize * h->L.NF * Np) != hipSuccess) return cleanup(fail(SOLORL_ERR_HIP, "hipMalloc state2"));
    if (hipMalloc(&h->si2, sizeof(int) * NI * Np) != hipSuccess) return cleanup(fail(SOLORL_ERR_HIP, "hipMalloc istate2"));
    if (hipMalloc(&h->perm, sizeof(int) * (size_t)num_envs) != hipSuccess) return cleanup(fail(SOLORL_ERR_HIP, "hipMalloc perm"));
  }
  dim3 g((num_envs + 255) / 256), b(256);
  if (h->f64) hipLaunchKernelGGL(init_pose_kernel<double>, g, b, 0, 0, (double*)h->sf, h->si, h->L, num_envs, 0.0);
  else hipLaunchKernelGGL(init_pose_kernel<float>, g, b, 0, 0, (float*)h->sf, h->si, h->L, num_envs, 0.0f);
  rc = h->f64 ? build_snapshots_t<double>(h) : build_snapshots_t<float>(h);
  if (rc) return cleanup(rc);
  if (hipDeviceSynchronize() != hipSuccess) return cleanup(fail(SOLORL_ERR_HIP, "device sync after create"));
  *out = h;
  return 0;
}

int solorl_destroy(solorl_env* h) {
  if (!h) return 0;
  hipSetDevice(h->device);
  if (h->sf) hipFree(h->sf);
  if (h->si) hipFree(h->si);
  if (h->snf) hipFree(h->snf);
  if (h->sni) hipFree(h->sni);
  if (h->sf2) hipFree(h->sf2);
  if (h->si2) hipFree(h->si2);
  if (h->perm) hipFree(h->perm);
  if (h->dyn) hipFree(h->dyn);
  delete h;
  return 0;
}

int solorl_dims(const solorl_env* h, int* obs_dim, int* act_dim, int* num_envs) {
  if (!h) return fail(SOLORL_ERR_INVALID, "null handle");
  if (obs_dim) *obs_dim = h->O;
  if (act_dim) *act_dim = h->n;
  if (num_envs) *num_envs = h->N;
  return 0;
}

int solorl_get_property(const solorl_env* h, const char* name, double* value) {
  if (!h || !name || !value) return fail(SOLORL_ERR_INVALID, "null argument");
  const std::string n(name);
  if (n == "lanes_per_env") *value = h->team ? 16 : 1;
  else if (n == "sweep_variant") {
    if (h->cfg.solver_residual_threshold > 0) *value = 2;
    else if (!h->team) *value = 1;          // (lane mode: one software-pipelined row loop)
    else {
      const int grid = (((h->N + 3) / 4) + 7) & ~7;
      *value = h->pipe_override >= 0 ? h->pipe_override : (grid <= h->simds ? 1 : 0);
    }
  } else if (n == "max_contacts") *value = MAX_CONTACTS;
  else if (n == "max_limit_rows") *value = MAX_LIMITS;
  else if (n == "f64") *value = h->f64 ? 1 : 0;
  else return fail(SOLORL_ERR_INVALID, "unknown property: " + n);
  return 0;
}

int solorl_reset(solorl_env* h, float* obs_out, void* stream) {
  if (!h) return fail(SOLORL_ERR_INVALID, "null handle");
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t st = (hipStream_t)stream;
  EnvParams P = make_env_params(h);
  dim3 g((h->N + 255) / 256), b(256);
  const bool s12 = h->cfg.robot == SOLORL_ROBOT_SOLO12;
  if (h->f64) {
    if (s12) hipLaunchKernelGGL((reset_kernel<double, 1>), g, b, 0, st, (double*)h->sf, h->si, (const double*)h->snf, (const int*)h->sni, h->M, h->L, h->N, P, obs_out);
    else hipLaunchKernelGGL((reset_kernel<double, 0>), g, b, 0, st, (double*)h->sf, h->si, (const double*)h->snf, (const int*)h->sni, h->M, h->L, h->N, P, obs_out);
  } else {
    if (s12) hipLaunchKernelGGL((reset_kernel<float, 1>), g, b, 0, st, (float*)h->sf, h->si, (const float*)h->snf, (const int*)h->sni, h->M, h->L, h->N, P, obs_out);
    else hipLaunchKernelGGL((reset_kernel<float, 0>), g, b, 0, st, (float*)h->sf, h->si, (const float*)h->snf, (const int*)h->sni, h->M, h->L, h->N, P, obs_out);
  }
  HIP_TRY(hipGetLastError());
  h->reset_called = true;
  return 0;
}

int solorl_step(solorl_env* h, const float* actions, float* obs_out, float* reward_out, uint8_t* done_out,
                const solorl_info_soa* info, void* stream) {
  if (!h) return fail(SOLORL_ERR_INVALID, "null handle");
  if (!h->reset_called) return fail(SOLORL_ERR_STATE, "env.reset() must be called before step");   // baseEnv.py:43
  if (!actions || !obs_out || !reward_out || !done_out) return fail(SOLORL_ERR_INVALID, "null array argument");
  HIP_TRY(hipSetDevice(h->device));
  Outputs o; memset(&o, 0, sizeof o);
  o.obs = obs_out; o.rew = reward_out; o.done = done_out;
  if (info) {
    o.timeout = info->timeout; o.success = info->success; o.nan_reset = info->nan_reset; o.ep_len = info->episode_length;
    o.ep_rew = info->episode_reward; o.goals = info->goals_reached; o.dr0 = info->dr_stand; o.dr1 = info->dr_joint_pose;
    o.dr2 = info->dr_torque; o.dr3 = info->dr_balance; o.dr4 = info->dr_progress; o.ep_stats = info->ep_stats;
    o.tau = info->applied_torque;
  }
  if (h->sort) {   // re-sort the state by last contact count (stable), into the spare buffer
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(sort_perm_kernel, dim3(1), dim3(256), 0, st, (const int*)h->si, h->N, h->perm);
    dim3 g((h->N + 255) / 256, h->L.NF + NI), b(256);
    if (h->f64) hipLaunchKernelGGL(gather_state_kernel<double>, g, b, 0, st, (const double*)h->sf, (const int*)h->si, (const int*)h->perm, h->N, h->L.NF, (double*)h->sf2, h->si2);
    else hipLaunchKernelGGL(gather_state_kernel<float>, g, b, 0, st, (const float*)h->sf, (const int*)h->si, (const int*)h->perm, h->N, h->L.NF, (float*)h->sf2, h->si2);
    HIP_TRY(hipGetLastError());
    std::swap(h->sf, h->sf2); std::swap(h->si, h->si2);
  }
  return dispatch_step(h, h->sf, h->si, h->N, actions, o, MODE_STEP, (hipStream_t)stream);
}

int solorl_step_act(solorl_env* h, const float* actions, float* obs_out, float* reward_out, uint8_t* done_out, const solorl_info_soa* info,
                    const solorl_policy_params* p, const float* noise, float* value_out, float* action_out, float* logp_out, void* stream) {
  if (!h) return fail(SOLORL_ERR_INVALID, "null handle");
  if (!p || !value_out || !action_out || !logp_out) return fail(SOLORL_ERR_INVALID, "solorl_step_act: null policy argument");
  if (h->f64 || !h->team || h->sort) return fail(SOLORL_ERR_INVALID, "solorl_step_act needs the fp32 team-mode engine without contact-count sorting (the defaults)");
  if (p->hidden != 64 || p->obs_dim != h->O || p->act_dim != h->n || (h->O & 3) || h->O > 88)
    return fail(SOLORL_ERR_INVALID, "solorl_step_act: the policy must be the MLP of hidden size 64 on this env's observation / action sizes, obs_dim a multiple of 4 "
                                    "and at most 88 (zero or one history level)");
  const void* ptrs[] = {p->critic_w0, p->critic_b0, p->critic_w1, p->critic_b1, p->critic_w2, p->critic_b2, p->actor_w0, p->actor_b0, p->actor_w1, p->actor_b1,
                        p->mean_w, p->mean_b, p->logstd};
  for (const void* q : ptrs) if (!q) return fail(SOLORL_ERR_INVALID, "solorl_step_act: null policy parameter pointer");
  const void* vec[] = {p->critic_w0, p->critic_w1, p->critic_w2, p->actor_w0, p->actor_w1, p->mean_w};
  for (const void* q : vec) if (reinterpret_cast<uintptr_t>(q) & 15u) return fail(SOLORL_ERR_INVALID, "solorl_step_act: weight matrices must be 16-byte aligned (rows are read as float4)");
  if (!h->reset_called) return fail(SOLORL_ERR_STATE, "env.reset() must be called before step");
  if (!actions || !obs_out || !reward_out || !done_out) return fail(SOLORL_ERR_INVALID, "null array argument");
  HIP_TRY(hipSetDevice(h->device));
  Outputs o; memset(&o, 0, sizeof o);
  o.obs = obs_out; o.rew = reward_out; o.done = done_out;
  if (info) {
    o.timeout = info->timeout; o.success = info->success; o.nan_reset = info->nan_reset; o.ep_len = info->episode_length;
    o.ep_rew = info->episode_reward; o.goals = info->goals_reached; o.dr0 = info->dr_stand; o.dr1 = info->dr_joint_pose;
    o.dr2 = info->dr_torque; o.dr3 = info->dr_balance; o.dr4 = info->dr_progress; o.ep_stats = info->ep_stats;
    o.tau = info->applied_torque;
  }
  o.pol = PolicyTail{p->critic_w0, p->critic_b0, p->critic_w1, p->critic_b1, p->critic_w2, p->critic_b2, p->actor_w0, p->actor_b0, p->actor_w1, p->actor_b1,
                     p->mean_w, p->mean_b, p->logstd, noise, value_out, action_out, logp_out};
  return dispatch_step(h, h->sf, h->si, h->N, actions, o, MODE_STEP, (hipStream_t)stream);
}

int solorl_get_observation(solorl_env* h, float* obs_out, void* stream) {
  if (!h || !obs_out) return fail(SOLORL_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t st = (hipStream_t)stream;
  dim3 g((h->N + 255) / 256), b(256);
  const bool s12 = h->cfg.robot == SOLORL_ROBOT_SOLO12;
  if (h->f64) {
    if (s12) hipLaunchKernelGGL((obs_kernel<double, 1>), g, b, 0, st, (const double*)h->sf, (const int*)h->si, h->L, h->N, h->cfg.task, obs_out);
    else hipLaunchKernelGGL((obs_kernel<double, 0>), g, b, 0, st, (const double*)h->sf, (const int*)h->si, h->L, h->N, h->cfg.task, obs_out);
  } else {
    if (s12) hipLaunchKernelGGL((obs_kernel<float, 1>), g, b, 0, st, (const float*)h->sf, (const int*)h->si, h->L, h->N, h->cfg.task, obs_out);
    else hipLaunchKernelGGL((obs_kernel<float, 0>), g, b, 0, st, (const float*)h->sf, (const int*)h->si, h->L, h->N, h->cfg.task, obs_out);
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

int solorl_increment_curriculum(solorl_env* h, double value) {
  if (!h) return fail(SOLORL_ERR_INVALID, "null handle");
  if (h->cfg.task != SOLORL_TASK_POINTGOAL) return 0;                  // solo.py:332-334 (assert task == 'pointgoal')
  HIP_TRY(hipSetDevice(h->device));
  // The kernels read goal_radius from the handle's device block, so launches already captured in a HIP graph see the
  // new value at their next replay.  Host-synchronising (rare call: once per curriculum_schedule updates): every step
  // enqueued before this call samples with the old radius, every later one with the new.
  HIP_TRY(hipDeviceSynchronize());
  h->goal_radius += value;
  HIP_TRY(hipMemcpy(h->dyn + DYN_GOAL_RADIUS, &h->goal_radius, sizeof(double), hipMemcpyHostToDevice));
  return 0;
}

static int find_slot(solorl_env* h, int env, int* slot) {
  // the state is kept sorted by contact count: look the env up in the slot -> env-id map
  if (!h->sort) { *slot = env; return 0; }
  const size_t Np = ((size_t)h->N + 3) & ~(size_t)3;
  std::vector<int> all(NI * Np);
  HIP_TRY(hipMemcpy(all.data(), h->si, sizeof(int) * NI * Np, hipMemcpyDeviceToHost));
  for (int s = 0; s < h->N; s++) if (all[(size_t)(s >> 2) * (NI << 2) + ((size_t)I_ENVID << 2) + (s & 3)] == env) { *slot = s; return 0; }
  return fail(SOLORL_ERR_INVALID, "env id not found in slot map");
}

int solorl_get_state(solorl_env* h, int i, solorl_env_state* out) {
  if (!h || !out || i < 0 || i >= h->N) return fail(SOLORL_ERR_INVALID, "bad argument");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipDeviceSynchronize());
  { int rc_ = find_slot(h, i, &i); if (rc_) return rc_; }
  // slot i of the AoSoA-4 storage (SX): its fields are 4 elements apart, starting at element (i >> 2) * NF * 4 + (i & 3)
  const size_t slot0f = (size_t)(i >> 2) * ((size_t)h->L.NF << 2) + (size_t)(i & 3), slot0i = (size_t)(i >> 2) * ((size_t)NI << 2) + (size_t)(i & 3);
  std::vector<double> f(h->L.NF);
  int iv[NI];
  if (h->f64) {
    HIP_TRY(hipMemcpy2D(f.data(), sizeof(double), (const char*)h->sf + sizeof(double) * slot0f, sizeof(double) * 4, sizeof(double), h->L.NF, hipMemcpyDeviceToHost));
  } else {
    std::vector<float> ff(h->L.NF);
    HIP_TRY(hipMemcpy2D(ff.data(), sizeof(float), (const char*)h->sf + sizeof(float) * slot0f, sizeof(float) * 4, sizeof(float), h->L.NF, hipMemcpyDeviceToHost));
    for (int k = 0; k < h->L.NF; k++) f[k] = ff[k];
  }
  HIP_TRY(hipMemcpy2D(iv, sizeof(int), (const char*)h->si + sizeof(int) * slot0i, sizeof(int) * 4, sizeof(int), NI, hipMemcpyDeviceToHost));
  const Layout& L = h->L;
  memset(out, 0, sizeof *out);
  for (int k = 0; k < 3; k++) { out->pos[k] = f[L.pos + k]; out->lin_vel[k] = f[L.v + k]; out->ang_vel[k] = f[L.w + k]; }
  for (int k = 0; k < 4; k++) out->quat[k] = f[L.quat + k];
  for (int j = 0; j < h->n; j++) { out->q[j] = f[L.q + j]; out->qd[j] = f[L.qd + j]; }
  for (int p = 0; p < NPRIM; p++) out->lambda_prev[p] = f[L.lam + p];
  for (int hh = 0; hh < SOLORL_STATE_MAX_HISTORY; hh++)
    for (int d = 0; d < DMAX; d++) out->hist[hh][d] = hh < (L.H > 2 ? L.H : 2) ? f[L.hist + hh * HSTRIDE + d] : 0.0;
  out->goal[0] = f[L.goal]; out->goal[1] = f[L.goal + 1]; out->potential = f[L.pot]; out->progress = f[L.prog];
  out->goals_reached = f[L.goals]; out->env_goals_reached = f[L.egoals];
  for (int k = 0; k < 5; k++) out->dr[k] = f[L.dr + k];
  out->treadmill_y = f[L.tmy];
  out->timestep = iv[I_TIMESTEP]; out->contact_mask = iv[I_MASK]; out->rng_counter = iv[I_RNG]; out->need_reset = iv[I_NEEDRESET];
  return 0;
}

int solorl_set_state(solorl_env* h, int i, const solorl_env_state* in) {
  if (!h || !in || i < 0 || i >= h->N) return fail(SOLORL_ERR_INVALID, "bad argument");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipDeviceSynchronize());
  const int env_id = i;
  { int rc_ = find_slot(h, i, &i); if (rc_) return rc_; }
  const Layout& L = h->L;
  const size_t slot0f = (size_t)(i >> 2) * ((size_t)L.NF << 2) + (size_t)(i & 3), slot0i = (size_t)(i >> 2) * ((size_t)NI << 2) + (size_t)(i & 3);
  std::vector<double> f(L.NF, 0.0);
  for (int k = 0; k < 3; k++) { f[L.pos + k] = in->pos[k]; f[L.v + k] = in->lin_vel[k]; f[L.w + k] = in->ang_vel[k]; }
  for (int k = 0; k < 4; k++) f[L.quat + k] = in->quat[k];
  for (int j = 0; j < h->n; j++) { f[L.q + j] = in->q[j]; f[L.qd + j] = in->qd[j]; }
  for (int p = 0; p < NPRIM; p++) f[L.lam + p] = in->lambda_prev[p];
  for (int hh = 0; hh < (L.H > 2 ? L.H : 2); hh++) for (int d = 0; d < DMAX; d++) f[L.hist + hh * HSTRIDE + d] = in->hist[hh][d];
  f[L.goal] = in->goal[0]; f[L.goal + 1] = in->goal[1]; f[L.pot] = in->potential; f[L.prog] = in->progress;
  f[L.goals] = in->goals_reached; f[L.egoals] = in->env_goals_reached;
  for (int k = 0; k < 5; k++) f[L.dr + k] = in->dr[k];
  f[L.xyprev] = in->pos[0]; f[L.xyprev + 1] = in->pos[1]; f[L.tmy] = in->treadmill_y;
  int iv[NI]; iv[I_TIMESTEP] = in->timestep; iv[I_MASK] = in->contact_mask; iv[I_RNG] = in->rng_counter; iv[I_NEEDRESET] = in->need_reset;
  iv[I_ENVID] = env_id;
  if (h->f64) {
    HIP_TRY(hipMemcpy2D((char*)h->sf + sizeof(double) * slot0f, sizeof(double) * 4, f.data(), sizeof(double), sizeof(double), L.NF, hipMemcpyHostToDevice));
  } else {
    std::vector<float> ff(L.NF);
    for (int k = 0; k < L.NF; k++) ff[k] = (float)f[k];
    HIP_TRY(hipMemcpy2D((char*)h->sf + sizeof(float) * slot0f, sizeof(float) * 4, ff.data(), sizeof(float), sizeof(float), L.NF, hipMemcpyHostToDevice));
  }
  HIP_TRY(hipMemcpy2D((char*)h->si + sizeof(int) * slot0i, sizeof(int) * 4, iv, sizeof(int), sizeof(int), NI, hipMemcpyHostToDevice));
  if (!in->need_reset) h->reset_called = true;
  return 0;
}

}  // extern "C"
#endif  // SOLO_TU_PART: the C ABI part
