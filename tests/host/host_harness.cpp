// host_harness.cpp -- TEST ONLY.  One physics sub-step through the *kernel's* templated code path
// (dynamics.hpp) on the CPU, one lane, T = float or double.
#define SOLO_HOST_SHIM 1
#include <cstring>
#include <vector>
#include "../../include/solorl.h"
#include "../../solorl_amd/csrc/dynamics.hpp"
using namespace solo;

template <typename T, int ROBOT>
static void run(solorl_env_state* s, const solorl_config* c, int apply_tau) {
  using RB = Robot<ROBOT>;
  constexpr int NQ = RB::NQ;
  std::vector<unsigned char> mem(RowLds<T>::bytes(1) + 64);
  RowLds<T> lds; lds.lanes = 1; lds.lane = 0; lds.base = mem.data();
  PhysParams<T> pp;
  pp.dt = (T)c->sim_dt; pp.gravity = (T)c->gravity; pp.erp = (T)c->erp; pp.slop = (T)c->linear_slop; pp.warm = (T)c->warmstart;
  pp.damping = (T)c->damping; pp.vmax = (T)c->max_velocity; pp.qlim = (T)c->joint_limit; pp.inv_dt = (T)(1.0 / c->sim_dt);
  pp.iterations = c->solver_iterations;
  pp.resid_thr = c->solver_residual_threshold > 0 ? (T)std::sqrt(c->solver_residual_threshold) : T(-1);
  pp.cerp = (T)c->contact_erp; pp.cmargin = (T)c->collision_margin;
  pp.set_mode(true, c->use_urdf_inertia != 0, c->friction_model == SOLORL_FRICTION_CONE, c->use_treadmill != 0);
  pp.tm_hw = (T)c->treadmill_half_width; pp.tm_mu = (T)c->treadmill_friction;
  SubCtx<T, ROBOT> C;
  PhysState<T, NQ>& st = C.ps;
  C.tmy = (T)s->treadmill_y;
  st.pos = mk((T)s->pos[0], (T)s->pos[1], (T)s->pos[2]);
  st.qx = (T)s->quat[0]; st.qy = (T)s->quat[1]; st.qz = (T)s->quat[2]; st.qw = (T)s->quat[3];
  st.v = mk((T)s->lin_vel[0], (T)s->lin_vel[1], (T)s->lin_vel[2]);
  st.w = mk((T)s->ang_vel[0], (T)s->ang_vel[1], (T)s->ang_vel[2]);
  T lam[NPRIM];
  for (int j = 0; j < NQ; j++) { st.q[j] = (T)s->q[j]; st.qd[j] = (T)s->qd[j]; C.tau[j] = apply_tau ? (T)s->tau[j] : T(0); }
  for (int p = 0; p < NPRIM; p++) lam[p] = (T)s->lambda_prev[p];
  int mask = substep<T, ROBOT>(C, pp, lam, 1, lds);
  s->pos[0] = st.pos.x; s->pos[1] = st.pos.y; s->pos[2] = st.pos.z;
  s->quat[0] = st.qx; s->quat[1] = st.qy; s->quat[2] = st.qz; s->quat[3] = st.qw;
  s->lin_vel[0] = st.v.x; s->lin_vel[1] = st.v.y; s->lin_vel[2] = st.v.z;
  s->ang_vel[0] = st.w.x; s->ang_vel[1] = st.w.y; s->ang_vel[2] = st.w.z;
  for (int j = 0; j < NQ; j++) { s->q[j] = st.q[j]; s->qd[j] = st.qd[j]; if (!c->hold_torque) s->tau[j] = 0; }
  for (int p = 0; p < NPRIM; p++) s->lambda_prev[p] = lam[p];
  s->contact_mask = mask;
}

extern "C" void harness_substep(solorl_env_state* s, const solorl_config* c, int use_float) {
  if (c->robot == SOLORL_ROBOT_SOLO12) { if (use_float) run<float, 1>(s, c, 1); else run<double, 1>(s, c, 1); }
  else { if (use_float) run<float, 0>(s, c, 1); else run<double, 0>(s, c, 1); }
}
