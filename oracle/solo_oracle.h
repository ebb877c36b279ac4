/* solo_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * fp64 CPU restatement of the reference hot path (SoloBaseEnv.step/reset over PyBullet).  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it; the product path
 * (solorl_amd/, libsolorl_hip.so) never links, loads or calls it.
 *
 * PARITY UNPINNED: the physics lives in the un-vendored, un-pinned third-party `pybullet`
 * (reference README.md:4-8; call sites solo.py:66,265), which is absent here, and the reference
 * holds no tests or golden vectors for this path.  The restatement follows Bullet's
 * published btMultiBody algorithm (deviations listed in DESIGN.md)
 * and is pinned by physics known-answer tests (tests/test_oracle_physics.py) instead.
 */
#ifndef SOLO_ORACLE_H
#define SOLO_ORACLE_H
#include "../include/solorl.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_env oracle_env;

oracle_env* oracle_create(const solorl_config* cfg, int num_envs, uint64_t seed, int64_t env_id_offset);
void oracle_destroy(oracle_env* e);
void oracle_dims(const oracle_env* e, int* obs_dim, int* act_dim, int* num_envs);
void oracle_reset(oracle_env* e, double* obs_out);
/* info arrays may be NULL; layout as solorl_info_soa but double/int32 host arrays */
void oracle_step(oracle_env* e, const double* actions, double* obs_out, double* rew_out,
                 uint8_t* done_out, uint8_t* timeout, uint8_t* success, int32_t* ep_len,
                 double* ep_rew, double* goals, double* dr /* [N*5] */);
void oracle_get_observation(oracle_env* e, double* obs_out);
void oracle_increment_curriculum(oracle_env* e, double value);
void oracle_get_state(const oracle_env* e, int i, solorl_env_state* out);
void oracle_set_state(oracle_env* e, int i, const solorl_env_state* in);
void oracle_set_threads(oracle_env* e, int nthreads);

/* ---- low-level hooks for the physics known-answer tests ---- */
/* one physics sub-step of env i (tau from its state; cleared afterwards unless hold_torque) */
void oracle_substep(oracle_env* e, int i);
/* mass matrix M [nv*nv] and bias h [nv] (gravity, Coriolis/gyroscopic, damping) of env i;
 * generalized velocity order: base angular (world), base linear (world), joint rates */
void oracle_mass_matrix(const oracle_env* e, int i, double* M, double* h);
/* unconstrained accelerations udot [nv] = M^-1 (tau - h) */
void oracle_forward_dynamics(const oracle_env* e, int i, double* udot);
/* kinetic, potential energy; linear momentum[3]; angular momentum about world origin[3] */
void oracle_energy_momentum(const oracle_env* e, int i, double* out8);
/* world position of link origin/com and world point+distance of each primitive: out[np*4] */
void oracle_prim_points(const oracle_env* e, int i, double* out);
/* last solve of env i: number of rows, and per active primitive the normal impulse */
void oracle_last_lambda(const oracle_env* e, int i, double* lambda_n /* [nprims] */);
/* PGS iterations the last sub-step of env i ran (<= solver_iterations: early exit on the residual threshold, K7) */
int oracle_last_iterations(const oracle_env* e, int i);
/* the K7 residual (max over rows of the squared velocity-level change) of that solve's last sweep */
double oracle_last_residual(const oracle_env* e, int i);
/* Engine emulation for measurements (tests/test_oracle_caps.py): solve at most max_contacts contact points (the deepest) and
 * max_limits joint-limit rows (the most violated) per sub-step, as the HIP engine's slot count does; 0 = no cap (default: the
 * oracle, like Bullet, has none). */
void oracle_set_caps(oracle_env* e, int max_contacts, int max_limits);
/* 0 (default): the analytic support primitives the HIP engine uses; 1: Bullet's scheme [K6] -- the convex hull of every link's
 * collision mesh against the plane, one new point per step into a persistent <= 4-point manifold (oracle/hull_data.h).
 * Clears the manifolds.  (Env var ORACLE_CONTACT_MODEL=1 selects it at create.) */
void oracle_set_contact_model(oracle_env* e, int model);
/* [K] ledger options (tools/k_ledger.py): Bullet defaults the reference inherits that are NOT solorl_config fields,
 * switchable on the oracle so that their effect can be measured.  Names: "friction_skip_zero_normal" (0), "gyro" (1), "limit_split" (0),
 * "breaking_scale" (1), "manifold_persist" (1; 0 = contact model 1 keeps only the current support vertex of each hull).  Returns 0, or -1
 * for an unknown name. */
int oracle_set_option(oracle_env* e, const char* name, double value);
/* sub-steps that had constraint rows, by the number of PGS sweeps they ran (128 bins, summed over envs; clear != 0 zeroes the counts) */
void oracle_iteration_histogram(oracle_env* e, int64_t* out, int clear);
/* last sub-step of env i: contact points found / solved, joint-limit candidates / solved */
void oracle_last_counts(const oracle_env* e, int i, int out[4]);
/* Philox4x32-10 (for RNG parity tests) */
void oracle_philox(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                   uint32_t out[4]);
void oracle_euler_from_quat(const double q[4], double rpy[3]);

#ifdef __cplusplus
}
#endif
#endif
