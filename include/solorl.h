/* solorl.h -- C ABI of the MI355X-native Solo8/Solo12 rollout engine.
 *
 * Drop-in boundary for the reference's vec-env layer.  Each entry point
 * replaces what the reference does through N worker processes and pickled Pipe messages:
 *
 *   solorl_create          <- agents/ppo/envs.py:14-30 (make_vec_envs), :66-89 (VecEnvWrapper.__init__
 *                             forking N `simple_worker`s, each building SoloBaseEnv(config):
 *                             baseEnv.py:7-40 -> solo.py:55-67)
 *   solorl_reset           <- agents/ppo/envs.py:97-100,198-200 -> baseEnv.py:70-82 (reset)
 *   solorl_step            <- agents/ppo/envs.py:91-95,189-196 -> :36-40 (worker: step + auto-reset)
 *                             -> baseEnv.py:42-68 (step)
 *   solorl_get_observation <- agents/ppo/envs.py:102-105,206-208 -> baseEnv.py:88-89
 *   solorl_dims            <- agents/ppo/envs.py:112-114 (get_spaces) -> baseEnv.py:20-28
 *   solorl_increment_curriculum <- agents/ppo/envs.py:125-127 (pointgoal: solo.py:332-334)
 *   solorl_compute_returns <- agents/ppo/storage.py:35-55 (OPBuffer.compute_returns: GAE / discounted returns;
 *                             400 sequential tiny torch ops per update in the reference)
 *   solorl_ppo_loss        <- agents/ppo/ppo.py:52-74 (clipped surrogate + clipped value loss) with the Gaussian
 *                             log-prob of agents/ppo/policy.py:51-58,171-173: ~60 elementwise torch kernels per mini-batch
 *   solorl_policy_act      <- agents/ppo/policy.py:33-49 (Policy.act on the MLPBase of :62-81): value, sampled action, log-prob
 *   solorl_ppo_grad_stage1 <- agents/ppo/ppo.py:46-74 for one mini-batch: storage.py:57-71 gather, policy.py:51-58
 *                             evaluate_actions, the clipped losses, and autograd's back-propagation down to every layer's
 *                             pre-activation gradient (what `loss.backward()` computes before the weight-gradient GEMMs)
 *   solorl_ppo_grad_stage2 <- the weight-gradient products of that backward pass, written into the parameters' gradients
 *   solorl_ppo_clip_adam   <- agents/ppo/ppo.py:75-77 (clip_grad_norm_ + optimizer.step) on the MLP's 13 parameter tensors
 *   solorl_step_act        <- agents/ppo/train.py:84-88: the rollout's pair `actor_critic.act(obs)` + `envs.step(action)` as ONE launch
 *                             (the policy is evaluated on each observation by the wavefront that has just produced it)
 *   solorl_get_state / solorl_set_state / solorl_get_property : no reference counterpart (parity-test hooks, run records)
 *   solorl_destroy         <- agents/ppo/envs.py:129-135 (close)
 *
 * Plain pointers and sizes only; no torch types.  All array arguments are DEVICE pointers
 * (HIP, the device the handle was created on); the engine never retains them past the stream
 * work of the call.  All work is enqueued on the `stream` argument (a hipStream_t passed as
 * void*; NULL = the null stream) and no entry point synchronises the host except
 * solorl_create/solorl_destroy/solorl_get_state/solorl_set_state.
 *
 * Error convention: every function returns 0 on success or a negative code; the message is
 * available from solorl_last_error() (thread-local).  A per-env numeric failure (NaN/Inf state)
 * is NOT an error: the env is force-terminated, reset and counted in info.nan_reset.
 *
 * A handle is not thread-safe; one caller thread and one GPU per handle.
 */
#ifndef SOLORL_H
#define SOLORL_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { SOLORL_ROBOT_SOLO8 = 0, SOLORL_ROBOT_SOLO12 = 1 };
enum { SOLORL_TASK_STAND = 0, SOLORL_TASK_WALK = 1, SOLORL_TASK_POINTGOAL = 2 };
enum { SOLORL_CONTROL_TORQUE = 0, SOLORL_CONTROL_PD = 1 };
enum { SOLORL_PRECISION_F32 = 0, SOLORL_PRECISION_F64 = 1 };
enum { SOLORL_FRICTION_PYRAMID = 0, SOLORL_FRICTION_CONE = 1 };

/* Bumped whenever a struct of this header changes layout or meaning; solorl_abi_version() returns the value the library was built with
 * and a binding compares the two before its first call (a caller built against another layout would hand over short structs). */
#define SOLORL_ABI_VERSION 5

enum {
  SOLORL_OK = 0,
  SOLORL_ERR_INVALID = -1,   /* bad argument / config */
  SOLORL_ERR_NODEVICE = -2,  /* no usable HIP device */
  SOLORL_ERR_HIP = -3,       /* a HIP runtime call failed */
  SOLORL_ERR_STATE = -4      /* step() before reset() (reference baseEnv.py:43 assert) */
};

/* Task/robot/physics configuration: the YAML keys the reference reads in baseEnv.py:8-16,164
 * plus the Bullet parameters the reference leaves at PyBullet defaults. */
typedef struct solorl_config {
  int32_t robot;            /* SOLORL_ROBOT_*  (from model_urdf basename / `solo12: True`) */
  int32_t task;             /* SOLORL_TASK_*   (config 'task', baseEnv.py:11) */
  int32_t control;          /* SOLORL_CONTROL_* (config 'control', baseEnv.py:10; pd = pd/fpd/fixed_pd) */
  int32_t frame_skip;       /* config 'frame_skip' (default 4) */
  int32_t episode_length;   /* config 'episode_length' (baseEnv.py:164) */
  int32_t num_history_stack;/* config 'num_history_stack' (solo.py:48 deque(maxlen=h)): 0..4 (SOLORL_STATE_MAX_HISTORY) */
  int32_t hold_torque;      /* 0: torque acts on sub-step 1 only (Bullet clears it, K8); 1: held */
  int32_t use_urdf_inertia; /* 0: Bullet default box inertia from collision AABB (K2); 1: URDF tensor */
  int32_t solver_iterations;/* PGS iterations (PyBullet default 50, K7) */
  int32_t settle_min;       /* reset: K ~ U{settle_min .. settle_max} zero-torque control steps */
  int32_t settle_max;       /*        (baseEnv.py:79: randint(5,12) -> 5..11)                      */
  int32_t disable_termination; /* parity runs: never set done */
  int32_t precision;        /* SOLORL_PRECISION_*: arithmetic type of the HIP engine */
  int32_t use_treadmill;    /* config 'use_treadmill' (simulation.py:24-26,45-77): friction strip, see treadmill_* below */
  int32_t friction_model;   /* SOLORL_FRICTION_*: how the two friction rows of a contact are bounded by mu * normal impulse.
                             * PYRAMID: each row clamped to +-mu lambda_n on its own, Gauss-Seidel between the two (rounds 1-3).
                             * CONE: Bullet's implicit friction cone [K] -- btMultiBodyConstraintSolver::resolveConeFrictionConstraintRows, the
                             * path PyBullet takes unless setPhysicsEngineParameter(enableConeFriction=0) (the reference never calls it,
                             * simulation.py:13-35): both rows' unclamped sums against the same delta-velocities, the pair projected
                             * radially onto the disc of radius mu lambda_n.  DESIGN.md section 3 ([K] ledger) has the measured difference. */
  double kp, kd;            /* config 'gains' (configs/basic_pd.yaml:6) */
  double max_torque;        /* solo.py:53 max_joint_torque = 3 */
  double sim_dt;            /* solo.py:22 scene_timestep = 1/240 */
  double reward_dt;         /* stands in for the undefined scene.dt of baseEnv.py:137 (1/60) */
  double gravity;           /* simulation.py:19: 9.81 (pointing -z) */
  double erp;               /* contact/limit error-reduction (0.2, K7) */
  double linear_slop;       /* PyBullet solver linearSlop 1e-5 */
  double warmstart;         /* warm-start factor on cached normal impulses.  Default 0: btMultiBodyConstraintSolver starts every
                             * multibody contact row at zero impulse (its warm start is disabled in the source: "issues gaining
                             * energy") [K].  btContactSolverInfo's 0.85 (rigid bodies) was the default of rounds 1-2; with it the
                             * residual exit below destabilises a PD stance (measured) */
  double damping;           /* btMultiBody linear = angular damping 0.04 (K3) */
  double max_velocity;      /* generalized velocity clamp 100 (K5) */
  double joint_limit;       /* +-10 rad (URDF; solo.py:109) */
  double goal_radius;       /* solo.py:141 goal_radius = 2.0 */
  /* Treadmill (simulation.py:45-77): a static 50 x 2 zero heightfield = a 49 m x 1 m strip at z = 0 centred on
   * y = +-treadmill_offset (sign redrawn at every reset, simulation.py:72-74), Bullet default friction 0.5; its
   * velocity is set on a mass-0 body and has no effect.  Modelled as: a contact whose point lies on the strip
   * (|y - y_strip| <= treadmill_half_width) gets friction link_mu * treadmill_friction.  The feet sensor (solo.py:313-317
   * queries the plane body) still reports such a foot: the infinite plane lies under the strip as well. */
  double treadmill_offset;      /* 0.49 (simulation.py:49) */
  double treadmill_half_width;  /* 0.5: (numHeightfieldColumns - 1) / 2 grid units */
  double treadmill_friction;    /* 0.5: Bullet's default lateral friction of the heightfield body */
  /* PyBullet's solverResidualThreshold (1e-7, set by its physics server): the PGS loop of a sub-step
   * ends after the first iteration whose largest squared velocity-level change, max_rows (delta_impulse / jacDiagABInv)^2, is
   * <= this value (or after solver_iterations).  Default 1e-7 = the reference's setting (it never calls
   * setPhysicsEngineParameter, simulation.py:13-35).  0 = always run solver_iterations sweeps (the default of rounds 1-2).
   * Measured on the oracle: with warmstart = 0 the exit leaves a PD stance stable over 1000 control steps
   * at a mean of ~12 sweeps; with warmstart = 0.85 it does not -- the pair (0, 1e-7) is what Bullet's multibody solver runs [K]. */
  double solver_residual_threshold;
  /* Error reduction of the CONTACT rows (Bullet: btContactSolverInfo::m_erp2, read by setupMultiBodyContactConstraint); `erp` above is
   * m_erp, which the joint-limit rows use.  PyBullet's physics server sets m_erp2 = 0.08 beside m_erp = 0.2 [K]; rounds 1-3 used one
   * value (0.2) for both.  DESIGN.md section 3 ([K] ledger) has the measured difference. */
  double contact_erp;
  /* Collision margin around every collision shape [m].  Bullet gives the convex hulls of a URDF import a 1 mm margin (gUrdfDefaultCollisionMargin,
   * SURVEY.md Appendix B K6 [K]): the effective shape is the hull swept by a sphere of that radius, so every support point lies the
   * margin further down.  The analytic primitives were fitted to the bare meshes (tools/compile_model.py); rounds 1-3 ran them bare (0). */
  double collision_margin;
} solorl_config;

/* Struct-of-arrays info block (replaces the per-env dicts of baseEnv.py:62-66).  Every pointer is
 * a device array of num_envs elements, or NULL to skip that field. */
typedef struct solorl_info_soa {
  uint8_t* timeout;          /* info['timeout'] (valid where done) */
  uint8_t* success;          /* info['success'] (valid where done) */
  uint8_t* nan_reset;        /* env was force-reset because its state went non-finite */
  int32_t* episode_length;   /* info['episode_length'] = timestep */
  float*   episode_reward;   /* info['episode_reward'] = LAST step reward (baseEnv.py:65) */
  float*   goals_reached;    /* info['goals_reached'] */
  float*   dr_stand;         /* info['dr/stand_rew'] running sum */
  float*   dr_joint_pose;    /* info['dr/joint_pose_rew'] */
  float*   dr_torque;        /* info['dr/torque_rew'] */
  float*   dr_balance;       /* info['dr/roll_pitch_balance_rew'] */
  float*   dr_progress;      /* info['dr/progress_rew'] */
  /* Finished-episode accumulators, field-major [SOLORL_EPSTAT_FIELDS][num_envs], or NULL: at every env-step that ends
   * an episode the engine ADDS that episode's values to the env's own column (no atomics: deterministic); the caller
   * reduces over envs and zeroes the block whenever it logs.  Replaces iterating N info dicts per step
   * (agents/ppo/train.py:90-100).  Fields: 0 episodes finished, 1 sum info['episode_reward'], 2 sum
   * info['episode_length'], 3 sum info['success'], 4..8 sums of the info['dr/...'] terms (stand, joint_pose, torque, balance,
   * progress), 9 episodes ended by the non-finite-state guard (not counted in 0..8). */
  float*   ep_stats;
  /* [num_envs][act_dim], or NULL: the joint torques this step applied (solo.py:224-259 after the clip / the PD law of
   * controllers/PD.py:3-10) -- what the reference hands to p.setJointMotorControlArray(TORQUE_CONTROL, forces=...).  Lets a
   * test compare the engine's PD arithmetic with the reference's own PD() outputs (tests/golden/pd_golden.json). */
  float*   applied_torque;
} solorl_info_soa;
#define SOLORL_EPSTAT_FIELDS 10

/* Physical + bookkeeping state of ONE env in a fixed, precision-independent (double) layout;
 * used by solorl_get_state/solorl_set_state (HOST pointers) for parity tests. */
#define SOLORL_STATE_MAX_DOF 12
#define SOLORL_STATE_MAX_PRIMS 24
#define SOLORL_STATE_MAX_OBS 42
#define SOLORL_STATE_MAX_HISTORY 4
typedef struct solorl_env_state {
  double pos[3], quat[4] /* x y z w */, lin_vel[3], ang_vel[3];
  double q[SOLORL_STATE_MAX_DOF], qd[SOLORL_STATE_MAX_DOF], tau[SOLORL_STATE_MAX_DOF];
  double lambda_prev[SOLORL_STATE_MAX_PRIMS];  /* warm-start normal impulses per collision primitive */
  double hist[SOLORL_STATE_MAX_HISTORY][SOLORL_STATE_MAX_OBS];   /* state_history, [0] = newest */
  double goal[2], potential, progress, goals_reached, env_goals_reached;
  double dr[5];                                /* stand, joint_pose, torque, balance, progress */
  double treadmill_y;                          /* centre line of the treadmill strip (+-treadmill_offset; 0 if unused) */
  int32_t timestep, need_reset;
  int32_t contact_mask;  /* bit p (0..23): primitive p was in contact in the last sub-step; bit 24+f: foot f's contact
                          * point was on the treadmill strip (strip friction; the feet sensor reports it all the same) */
  int32_t rng_counter;
} solorl_env_state;

typedef struct solorl_env solorl_env;

/* Fills *cfg with the reference's defaults for (robot, task). */
int solorl_default_config(solorl_config* cfg, int robot, int task);

/* num_envs environments on HIP device `device_id`; `seed` + `env_id_offset` key the per-env
 * Philox streams (global env id = env_id_offset + i). */
int solorl_create(const solorl_config* cfg, int num_envs, int device_id, uint64_t seed,
                  int64_t env_id_offset, solorl_env** out);
int solorl_destroy(solorl_env* env);

int solorl_dims(const solorl_env* env, int* obs_dim, int* act_dim, int* num_envs);

int solorl_reset(solorl_env* env, float* obs_out /* [N*O] */, void* stream);
int solorl_step(solorl_env* env, const float* actions /* [N*A] */, float* obs_out /* [N*O] */,
                float* reward_out /* [N] */, uint8_t* done_out /* [N] */,
                const solorl_info_soa* info_out /* host struct of device pointers, or NULL */,
                void* stream);
int solorl_get_observation(solorl_env* env, float* obs_out, void* stream);
int solorl_increment_curriculum(solorl_env* env, double value);

/* Read-only properties of a handle (no reference counterpart): how this handle's batch is mapped and solved, so that a benchmark
 * or a scaling run can record and pin what it measured.  Names: "lanes_per_env" (16 team mode / 1 lane mode), "sweep_variant"
 * (2: PGS sweep with the K7 residual exit -- the default, one variant at every batch size; 1 / 0: fixed-iteration sweep,
 * software-pipelined / plain -- chosen by grid size when solver_residual_threshold = 0 unless SOLORL_PGS_PIPE pins it, the last bits
 * of an env then depend on which side of one wavefront per SIMD its batch is), "max_contacts", "max_limit_rows", "f64".
 * Unknown name: SOLORL_ERR_INVALID. */
int solorl_get_property(const solorl_env* env, const char* name, double* value);

int solorl_get_state(solorl_env* env, int env_index, solorl_env_state* out /* host */);
int solorl_set_state(solorl_env* env, int env_index, const solorl_env_state* in /* host */);

/* Fused return computation over a rollout of T steps x N envs (device arrays, time-major [t][n]).
 * use_gae != 0:  value_preds[T][:] = next_value;  delta_t = r_t + gamma V_{t+1} m_{t+1} - V_t;
 *                A_t = delta_t + gamma lambda m_{t+1} A_{t+1};  returns[t] = A_t + V_t      (storage.py:41-50)
 * use_gae == 0:  returns[T][:] = next_value;  returns[t] = returns[t+1] gamma m_{t+1} + r_t   (storage.py:51-55) */
int solorl_compute_returns(const float* rewards /* [T*N] */, float* value_preds /* [(T+1)*N] */,
                           const float* masks /* [(T+1)*N] */, const float* next_value /* [N] */,
                           float* returns /* [(T+1)*N] */, int T, int N, int use_gae, float gamma,
                           float gae_lambda, int device_id, void* stream);

/* Fused PPO mini-batch loss, forward + gradients w.r.t. the policy heads (device arrays, m samples, A action dims):
 *   logp_i = sum_j -0.5 z_ij^2 - logstd_j - log sqrt(2 pi),  z = (action - mean) exp(-logstd);  ratio = exp(logp - old_logp)
 *   action loss = -mean_i min(ratio adv, clamp(ratio, 1-clip, 1+clip) adv)                                  (ppo.py:54-59)
 *   value  loss = 0.5 mean_i max((v - ret)^2, (vpred + clamp(v - vpred, +-clip) - ret)^2)  [clipped_value]  (ppo.py:61-68)
 * Outputs: grad_mean [m*A] = d(action loss)/d mean, grad_values [m] = value_coef * d(value loss)/d v, and per-block
 * partial sums partials[ceil(m/256)][3 + A] = (sum value loss, sum action loss, count, d(action loss)/d logstd[A]);
 * the caller adds the blocks (deterministic) and the entropy term, which does not depend on the samples. */
int solorl_ppo_loss(const float* mean, const float* logstd, const float* values, const float* action, const float* old_logp,
                    const float* adv, const float* vpred, const float* ret, int m, int A, float clip, float value_coef,
                    int clipped_value, float* grad_mean, float* grad_values, float* partials, int device_id, void* stream);

/* Parameters of the reference's MLP actor-critic in PyTorch layout (nn.Linear weight = [out][in], row-major), device pointers:
 * base.critic.{0,2,4}, base.features.{0,2}, pi_dist.mean, pi_dist.logstd (agents/ppo/policy.py:62-81,138-148).  The kernels
 * are built for hidden = 64 and (obs_dim, act_dim) in {(38|42|76|84, 12), (30|34|60|68, 8)} -- zero or one history level; other shapes
 * return SOLORL_ERR_INVALID and the caller keeps its framework path.  Alignment: the weight matrices critic_w1/w2, actor_w1,
 * mean_w -- and, when obs_dim % 4 == 0, critic_w0, actor_w0 and every `obs` array handed to solorl_policy_act /
 * solorl_ppo_batch -- are read as float4 rows and must start on a 16-byte boundary (checked: SOLORL_ERR_INVALID otherwise). */
typedef struct solorl_policy_params {
  int obs_dim, act_dim, hidden, reserved0;
  const float *critic_w0, *critic_b0, *critic_w1, *critic_b1, *critic_w2, *critic_b2;
  const float *actor_w0, *actor_b0, *actor_w1, *actor_b1, *mean_w, *mean_b, *logstd;
} solorl_policy_params;

/* Policy.act for n observation rows: value_out[n], action_out[n][A] = mean + exp(logstd) * noise (noise[n][A]: the caller's
 * standard-normal draw; NULL = deterministic, action = mean), logp_out[n] = log N(action; mean, exp(logstd)) summed over A. */
int solorl_policy_act(const solorl_policy_params* p, const float* obs /* [n][obs_dim] */, const float* noise, int n, float* value_out,
                      float* action_out, float* logp_out, int device_id, void* stream);

/* One PPO mini-batch, everything of forward + backward except the weight-gradient GEMMs.  Sample r of the mini-batch is
 * row perm[*offset + r] of the rollout arrays (perm, offset: device memory, so a captured launch needs no host argument). */
typedef struct solorl_ppo_batch {
  const float *obs /* [n][obs_dim] */, *actions /* [n][A] */, *old_logp, *adv, *vpred, *ret /* [n] each */;
  const int64_t *perm, *offset;
  int m;                       /* mini-batch rows */
  int clipped_value;           /* ppo.py:61 use_clipped_value_loss */
  float clip, value_coef;      /* ppo.py:56 clip_param; the value-loss coefficient is folded into the critic's gradient */
} solorl_ppo_batch;
/* Outputs (m a multiple of 32), each units x m floats laid out [tile = row / 32][unit][row % 32] -- a tile of 32 rows is one contiguous
 * block (ABI 5; it was [unit][m]): xt0 = gathered observations (obs_dim units); per net (c_ critic, a_ actor) the hidden activations
 * xt1, xt2 (64 units), the pre-activation gradients g1, g2 (64 units) and the head-output gradient gh (1 / A units).
 * Weight gradients are then G X^T products over the rows:  d W0 = g1 xt0^T, d W1 = g2 xt1^T, d Whead = gh xt2^T, biases = row
 * sums of g.  partials [m / 32][3 + A]: per 32 rows (sum value loss, sum action loss, rows, sum d action-loss / d logstd_a; ABI 5:
 * one row per 32 samples, it was 64);
 * the entropy term of ppo.py:74 does not depend on the samples and is left to the caller. */
typedef struct solorl_ppo_stage1 {
  float *xt0, *c_xt1, *c_xt2, *c_g1, *c_g2, *c_gh, *a_xt1, *a_xt2, *a_g1, *a_g2, *a_gh, *partials;
} solorl_ppo_stage1;
int solorl_ppo_grad_stage1(const solorl_policy_params* p, const solorl_ppo_batch* batch, const solorl_ppo_stage1* work, int device_id,
                           void* stream);

/* Stage 2: the weight gradients d W = g x^T (+ biases) of all six layers from stage 1's arrays, and the step's bookkeeping.
 * `out` names where each gradient goes (PyTorch layouts, e.g. views of one flat bucket), plus
 *   loss_sums [3 + A]  += (sum value loss, sum action loss, rows, sum d action-loss / d logstd) of this mini-batch,
 *   logstd_sum [1]     += sum_a logstd_a            (entropy = 0.5 + log sqrt(2 pi) + mean_a logstd_a, policy.py:56)
 *   logstd             = sum d action-loss / d logstd - entropy_coef / A       (the complete gradient of ppo.py:74's loss)
 *   scratch            >= solorl_ppo_scratch_count(obs_dim, act_dim, m) floats of working memory (one solorl_ppo_grad_count block per
 *                      chunk of rows; ABI 5 cut the rows into ~170 chunks instead of ceil(m / 512)).
 * Sums are taken in a fixed order: results are reproducible run to run. */
typedef struct solorl_ppo_grads {
  float *critic_w0, *critic_b0, *critic_w1, *critic_b1, *critic_w2, *critic_b2;
  float *actor_w0, *actor_b0, *actor_w1, *actor_b1, *mean_w, *mean_b, *logstd;
  float *loss_sums, *logstd_sum, *scratch;
  float entropy_coef, reserved0;
} solorl_ppo_grads;
int solorl_ppo_grad_stage2(const solorl_policy_params* p, const solorl_ppo_stage1* work, int m, const solorl_ppo_grads* out, int device_id,
                           void* stream);
/* number of weight + bias elements of the actor-critic except logstd (one scratch chunk) */
int solorl_ppo_grad_count(int obs_dim, int act_dim);
/* floats of solorl_ppo_grads.scratch that solorl_ppo_grad_stage2 needs for a mini-batch of m rows (0 for m < 1) */
int solorl_ppo_scratch_count(int obs_dim, int act_dim, int m);

/* Gradient-norm clip + Adam step over the 13 parameter tensors in one launch: nn.utils.clip_grad_norm_(max_grad_norm)
 * (agents/ppo/ppo.py:75-76; coefficient max_norm / (norm + 1e-6) clamped to 1) followed by torch.optim.Adam's update (:32, :77:
 * bias-corrected, eps outside the square root, L2 weight_decay added to the gradient, no amsgrad).  The parameters behind `p`
 * are UPDATED IN PLACE; exp_avg / exp_avg_sq hold solorl_ppo_grad_count(obs_dim, act_dim) + act_dim floats in the order of
 * solorl_policy_params' pointers; step [1] is incremented; lr [1] is read from the device (a schedule may rewrite it between
 * calls); offset (optional) is advanced by offset_increment -- the mini-batch cursor of solorl_ppo_batch.  grad_scale folds the
 * 1 / world_size of a data-parallel gradient mean into the same launch. */
typedef struct solorl_adam_state {
  float *exp_avg, *exp_avg_sq, *step;
  const float* lr;
  int64_t* offset;
  int64_t offset_increment;
  float beta1, beta2, eps, weight_decay, max_grad_norm /* <= 0: no clipping */;
  float grad_scale;   /* every gradient element is multiplied by this before the norm and the update: 1 / world_size after a SUMMED
                       * all-reduce of the flat bucket (data-parallel mean); <= 0 is read as 1 */
} solorl_adam_state;
int solorl_ppo_clip_adam(const solorl_policy_params* p, const solorl_ppo_grads* g, const solorl_adam_state* a, int device_id, void* stream);

/* solorl_step followed by Policy.act (agents/ppo/policy.py:33-49) on the NEW observation of every env, in the same launch: value_out
 * [N], action_out [N][A] = mean + exp(logstd) * noise (noise [N][A]: the caller's standard-normal draw; NULL: action = mean),
 * logp_out [N].  What the reference's rollout loop does with two calls per step (agents/ppo/train.py:84-88) and solorl_policy_act +
 * solorl_step do with two launches.  Why one: at 4096 envs per GPU the step launch lasts as long as its slowest wavefront while the
 * mean wavefront is done in half that time; each wavefront evaluates the policy for its own four envs as soon as their observations
 * exist, so the policy costs no launch of its own.  Needs the engine's defaults (fp32, team mode, no contact-count sorting), hidden
 * = 64, obs_dim = this env's observation size, a multiple of 4 and at most 88 (zero or one history level), 16-byte aligned weight matrices; otherwise
 * SOLORL_ERR_INVALID and the caller keeps the two-call form.  Outputs equal solorl_policy_act's on the same rows up to the order
 * of the f32 sums (tested: 2e-6). */
int solorl_step_act(solorl_env* env, const float* actions, float* obs_out, float* reward_out, uint8_t* done_out,
                    const solorl_info_soa* info_out, const solorl_policy_params* p, const float* noise, float* value_out,
                    float* action_out, float* logp_out, void* stream);

const char* solorl_last_error(void);
const char* solorl_version(void);
int solorl_abi_version(void);   /* SOLORL_ABI_VERSION of the built library */

#ifdef __cplusplus
}
#endif
#endif /* SOLORL_H */
