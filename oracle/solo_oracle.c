/* solo_oracle.c -- TEST INFRASTRUCTURE ONLY (see solo_oracle.h).  "PARITY UNPINNED".
 *
 * Plain-C fp64 restatement, one env at a time, of
 *   SoloBaseEnv.step / reset              reference baseEnv.py:42-82
 *   SoloBase.apply_action / simulator_step / calc_state / reset   solo.py:166-340
 *   PD                                    controllers/PD.py:3-10
 *   worker auto-reset                     agents/ppo/envs.py:36-40
 *   p.stepSimulation()                    third-party Bullet (absent): restated from its published
 *                                         btMultiBody design, SURVEY.md Appendix B K1-K11.
 *
 * The dynamics are deliberately written in a DIFFERENT formulation from the HIP engine so that
 * agreement is a real check: here a dense joint-space method (Jacobian-projected Newton-Euler ->
 * mass matrix M and bias h, Cholesky, M^-1 J^T per constraint row, Bullet-style PGS on generalized
 * delta-velocities); the HIP engine uses an O(n) articulated-body recursion and a base-space PGS.
 */
#include "solo_oracle.h"
#include "../include/solorl_model_data.h"
#include "hull_data.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define NL_MAX SOLORL_MAX_LINKS
#define NV_MAX (6 + SOLORL_MAX_DOF)
#define NP_MAX SOLORL_MAX_PRIMS
/* No caps of its own: every joint within LIMIT_WINDOW of a limit gets its row and every contact point its three rows,
 * as in Bullet.  The HIP engine solves at most 8 contact points and 2 limit rows per robot (its PGS slot count); oracle_set_caps()
 * lets a test impose those numbers here to measure what they cost (tests/test_oracle_caps.py) -- the parity tests run uncapped. */
#define NPT_MAX (4 * NL_MAX)                       /* contact points: <= 4 per link in the manifold model, NP_MAX primitives otherwise */
#define MAX_ROWS (2 * SOLORL_MAX_DOF + 3 * NPT_MAX)
/* collision margin of URDF-imported convex hulls: solorl_config collision_margin, both contact models */
#define ITER_HIST 128
/* A joint-limit row exists while the joint is AT or BEYOND its limit (margin <= 0): btMultiBodyJointLimitConstraint::createConstraintRows
 * skips a row whose `penetration > 0` [K5].  (Rounds 1-2: a speculative row from 0.5 rad before the limit; ORACLE_LIMIT_WINDOW=0.5
 * restores it for the measurement in tests/test_oracle_caps.py.) */
static double LIMIT_WINDOW = 0.0;
#define DISC_EPS2 1e-12
#define DISC_RIM 0.1     /* sine of the tilt at which a thick disc rests fully on its face (see prim_point) */

/* ------------------------------------------------------------------ small algebra */
static void v3set(double* a, double x, double y, double z) { a[0] = x; a[1] = y; a[2] = z; }
static void v3cpy(double* a, const double* b) { a[0] = b[0]; a[1] = b[1]; a[2] = b[2]; }
static void v3add(double* r, const double* a, const double* b) { r[0] = a[0] + b[0]; r[1] = a[1] + b[1]; r[2] = a[2] + b[2]; }
static void v3sub(double* r, const double* a, const double* b) { r[0] = a[0] - b[0]; r[1] = a[1] - b[1]; r[2] = a[2] - b[2]; }
static void v3axpy(double* r, double s, const double* a) { r[0] += s * a[0]; r[1] += s * a[1]; r[2] += s * a[2]; }
static double v3dot(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static double v3norm(const double* a) { return sqrt(v3dot(a, a)); }
static void v3cross(double* r, const double* a, const double* b) {
  double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
static void m3mulv(double* r, const double* M, const double* v) { /* row-major */
  double x = M[0] * v[0] + M[1] * v[1] + M[2] * v[2];
  double y = M[3] * v[0] + M[4] * v[1] + M[5] * v[2];
  double z = M[6] * v[0] + M[7] * v[1] + M[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
static void m3tmulv(double* r, const double* M, const double* v) {
  double x = M[0] * v[0] + M[3] * v[1] + M[6] * v[2];
  double y = M[1] * v[0] + M[4] * v[1] + M[7] * v[2];
  double z = M[2] * v[0] + M[5] * v[1] + M[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
static void m3mul(double* r, const double* A, const double* B) {
  double t[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) t[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
  memcpy(r, t, sizeof t);
}
static void quat_to_mat(const double* q, double* R) { /* q = x y z w */
  double x = q[0], y = q[1], z = q[2], w = q[3];
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w);     R[2] = 2 * (x * z + y * w);
  R[3] = 2 * (x * y + z * w);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
  R[6] = 2 * (x * z - y * w);     R[7] = 2 * (y * z + x * w);     R[8] = 1 - 2 * (x * x + y * y);
}
static void axis_rot(const double* ax, double ang, double* R) { /* Rodrigues */
  double c = cos(ang), s = sin(ang), t = 1 - c, x = ax[0], y = ax[1], z = ax[2];
  R[0] = t * x * x + c;     R[1] = t * x * y - s * z; R[2] = t * x * z + s * y;
  R[3] = t * x * y + s * z; R[4] = t * y * y + c;     R[5] = t * y * z - s * x;
  R[6] = t * x * z - s * y; R[7] = t * y * z + s * x; R[8] = t * z * z + c;
}

/* K10: Bullet btQuaternion::getEulerZYX as returned by p.getEulerFromQuaternion (roll,pitch,yaw) */
void oracle_euler_from_quat(const double q[4], double rpy[3]) {
  double x = q[0], y = q[1], z = q[2], w = q[3];
  double sqx = x * x, sqy = y * y, sqz = z * z, squ = w * w;
  double sarg = -2.0 * (x * z - w * y);
  if (sarg <= -0.99999) { rpy[1] = -0.5 * M_PI; rpy[0] = 0; rpy[2] = 2 * atan2(x, -y); }
  else if (sarg >= 0.99999) { rpy[1] = 0.5 * M_PI; rpy[0] = 0; rpy[2] = 2 * atan2(-x, y); }
  else {
    rpy[1] = asin(sarg);
    rpy[0] = atan2(2 * (y * z + w * x), squ - sqx - sqy + sqz);
    rpy[2] = atan2(2 * (x * y + w * z), squ + sqx - sqy - sqz);
  }
}

/* ------------------------------------------------------------------ Philox4x32-10 */
void oracle_philox(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t out[4]) {
  for (int r = 0; r < 10; r++) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
static double u01(uint32_t x) { return (double)(x >> 8) * (1.0 / 16777216.0); }

/* ------------------------------------------------------------------ model / env */
struct oracle_env {
  const solorl_model_data* md;
  solorl_config cfg;
  int N, n, nv, nl, np, D, O, nthreads;
  uint64_t seed; int64_t id0;
  double goal_radius;
  double inertia[NL_MAX][6]; /* link inertia about COM in link axes: xx yy zz xy xz yz */
  solorl_env_state* st;
  double (*last_lambda)[NP_MAX];
  double* last_resid;            /* K7 residual (max squared velocity-level change) of the last sweep of each env's last sub-step */
  int* last_iterations;          /* PGS iterations the last sub-step of each env ran (early exit, K7) */
  int* last_counts;              /* [N][4]: last sub-step's contact points before / after the cap, limit candidates before / after */
  int cap_contacts, cap_limits;  /* 0 = none (default); oracle_set_caps */
  /* contact model: 0 = the analytic primitives the HIP engine uses (one stateless support point per primitive);
   * 1 = Bullet's own scheme [K6]: per link the convex hull of its collision mesh against the plane, ONE new point per step (the
   * support vertex) into a persistent manifold of <= 4 points, refreshed / dropped by the breaking threshold */
  int contact_model;
  unsigned manifold_links;      /* model 1: bit l = link l collides through its hull manifold; links not in the mask keep their primitives (ablation hook) */
  const oracle_hull* hulls;
  struct manifold_t* man;        /* [N][NL_MAX], model 1 */
  /* [K] ledger options (oracle_set_option; tools/k_ledger.py): each restates one Bullet default the reference inherits, so that what it
   * would cost to have it wrong can be MEASURED.  Defaults = what the HIP engine runs (solorl_config carries the ones that became fields). */
  int opt_friction_skip;         /* pyramid only: a friction row is not solved while its contact's normal impulse is 0 (Bullet's `if (totalImpulse > 0)`) */
  int opt_gyro;                  /* 1 (default): gyroscopic term w x I w (btMultiBody::m_useGyroTerm) */
  int opt_limit_split;           /* 1: a joint more than 0.04 rad beyond its limit loses the positional term (btMultiBodyJointLimitConstraint with
                                  * m_splitImpulse on and no split-impulse pass for multibodies) and uses erp2 */
  int opt_foot_points;           /* primitives: 1 (default) one support point per foot; 2: BOTH edges of the 4 mm tread (y = +-ring_y[0]) as separate points --
                                  * the cheapest stand-in for the multi-point patch Bullet's manifold keeps under a foot (K6 experiment) */
  int opt_manifold_persist;      /* manifold model: 1 (default) Bullet's persistent <= 4-point cache; 0: only this step's support vertex (ablation: geometry vs caching) */
  double opt_break_scale;        /* scale of every contact-breaking threshold (gContactBreakingThreshold 0.02) */
  int64_t (*iter_hist)[ITER_HIST]; /* [N][ITER_HIST]: sub-steps that had rows, by the number of PGS sweeps they ran */
};
typedef struct manifold_t { int n; double local[4][3], worldB[4][3], dist[4], lam[4]; } manifold_t;
static const solorl_prim_data* prim_of(const oracle_env* E, int p) { return &E->md->prims[p]; }

typedef struct {
  double R[NL_MAX][9], o[NL_MAX][3], c[NL_MAX][3], a[NL_MAX][3];
  double w[NL_MAX][3], vo[NL_MAX][3], vc[NL_MAX][3];
  double alb[NL_MAX][3], aob[NL_MAX][3], acb[NL_MAX][3];
  double Iw[NL_MAX][9];
} kin_t;

static void kinematics(const oracle_env* E, const solorl_env_state* s, kin_t* K) {
  const solorl_model_data* md = E->md;
  quat_to_mat(s->quat, K->R[0]);
  v3cpy(K->o[0], s->pos);
  v3cpy(K->w[0], s->ang_vel); v3cpy(K->vo[0], s->lin_vel);
  v3set(K->alb[0], 0, 0, 0); v3set(K->aob[0], 0, 0, 0); v3set(K->a[0], 0, 0, 0);
  for (int i = 0; i < E->nl; i++) {
    const solorl_link_data* L = &md->links[i];
    if (i > 0) {
      int p = L->parent;
      double r[3], t[3], t2[3];
      m3mulv(r, K->R[p], L->jorigin);             /* o_i - o_p in world */
      v3add(K->o[i], K->o[p], r);
      m3mulv(K->a[i], K->R[p], L->axis);          /* world joint axis (zero for fixed) */
      double qi = 0, qdi = 0;
      if (L->jtype == 0) { qi = s->q[L->dof]; qdi = s->qd[L->dof]; }
      if (L->jtype == 0) { double Rj[9]; axis_rot(L->axis, qi, Rj); m3mul(K->R[i], K->R[p], Rj); }
      else memcpy(K->R[i], K->R[p], sizeof K->R[i]);
      /* velocities */
      v3cpy(K->w[i], K->w[p]); v3axpy(K->w[i], qdi, K->a[i]);
      v3cross(t, K->w[p], r); v3add(K->vo[i], K->vo[p], t);
      /* bias accelerations (all generalized accelerations zero) */
      v3cross(t, K->w[p], K->a[i]);
      v3cpy(K->alb[i], K->alb[p]); v3axpy(K->alb[i], qdi, t);
      v3cross(t, K->alb[p], r); v3cross(t2, K->w[p], r); v3cross(t2, K->w[p], t2);
      v3add(K->aob[i], K->aob[p], t); v3add(K->aob[i], K->aob[i], t2);
    }
    double rc[3], t[3], t2[3];
    m3mulv(rc, K->R[i], L->com);
    v3add(K->c[i], K->o[i], rc);
    v3cross(t, K->w[i], rc); v3add(K->vc[i], K->vo[i], t);
    v3cross(t, K->alb[i], rc); v3cross(t2, K->w[i], rc); v3cross(t2, K->w[i], t2);
    v3add(K->acb[i], K->aob[i], t); v3add(K->acb[i], K->acb[i], t2);
    /* world inertia about COM: R I R^T */
    const double* I6 = E->inertia[i];
    double Il[9] = {I6[0], I6[3], I6[4], I6[3], I6[1], I6[5], I6[4], I6[5], I6[2]}, Rt[9], T[9];
    for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) Rt[3 * a + b] = K->R[i][3 * b + a];
    m3mul(T, K->R[i], Il); m3mul(K->Iw[i], T, Rt);
  }
}

/* generalized-force row of a unit force u applied at world point P on link l (= Jacobian row) */
static void point_jacobian_row(const oracle_env* E, const solorl_env_state* s, const kin_t* K, int l,
                               const double* P, const double* u, double* row) {
  double r[3], t[3];
  for (int k = 0; k < E->nv; k++) row[k] = 0;
  v3sub(r, P, s->pos); v3cross(t, r, u);
  row[0] = t[0]; row[1] = t[1]; row[2] = t[2];
  row[3] = u[0]; row[4] = u[1]; row[5] = u[2];
  for (int j = l; j > 0; j = E->md->links[j].parent) {
    const solorl_link_data* L = &E->md->links[j];
    if (L->jtype != 0) continue;
    v3sub(r, P, K->o[j]); v3cross(t, r, u);
    row[6 + L->dof] = v3dot(K->a[j], t);
  }
}

static void mass_and_bias(const oracle_env* E, const solorl_env_state* s, const kin_t* K, double* M, double* h) {
  int nv = E->nv;
  const double kd = E->cfg.damping;
  memset(M, 0, sizeof(double) * nv * nv);
  memset(h, 0, sizeof(double) * nv);
  for (int i = 0; i < E->nl; i++) {
    const solorl_link_data* L = &E->md->links[i];
    double Jv[3][NV_MAX], Jw[3][NV_MAX];
    memset(Jv, 0, sizeof Jv); memset(Jw, 0, sizeof Jw);
    double r[3], t[3], e[3];
    v3sub(r, K->c[i], s->pos);
    for (int k = 0; k < 3; k++) {
      v3set(e, k == 0, k == 1, k == 2);
      v3cross(t, e, r);
      Jv[0][k] = t[0]; Jv[1][k] = t[1]; Jv[2][k] = t[2];
      Jv[k][3 + k] = 1; Jw[k][k] = 1;
    }
    for (int j = i; j > 0; j = E->md->links[j].parent) {
      const solorl_link_data* Lj = &E->md->links[j];
      if (Lj->jtype != 0) continue;
      v3sub(r, K->c[i], K->o[j]); v3cross(t, K->a[j], r);
      int col = 6 + Lj->dof;
      Jv[0][col] = t[0]; Jv[1][col] = t[1]; Jv[2][col] = t[2];
      Jw[0][col] = K->a[j][0]; Jw[1][col] = K->a[j][1]; Jw[2][col] = K->a[j][2];
    }
    for (int a = 0; a < nv; a++)
      for (int b = 0; b < nv; b++) {
        double acc = 0;
        for (int k = 0; k < 3; k++) acc += L->mass * Jv[k][a] * Jv[k][b];
        for (int k = 0; k < 3; k++)
          for (int m = 0; m < 3; m++) acc += Jw[k][a] * K->Iw[i][3 * k + m] * Jw[m][b];
        M[a * nv + b] += acc;
      }
    /* link wrench terms: m*a_c^bias - F,   I*alpha^bias + w x I w - T
     * F = m g + damping (K3: -m v (k + k|v|)),  T = damping (-I w (k + k|w|)) */
    double F[3], T[3], Iw_w[3], lin[3], ang[3];
    double vn = v3norm(K->vc[i]), wn = v3norm(K->w[i]);
    m3mulv(Iw_w, K->Iw[i], K->w[i]);
    for (int k = 0; k < 3; k++) {
      F[k] = -L->mass * K->vc[i][k] * (kd + kd * vn);
      T[k] = -Iw_w[k] * (kd + kd * wn);
    }
    F[2] += -L->mass * E->cfg.gravity;
    m3mulv(ang, K->Iw[i], K->alb[i]);
    v3cross(t, K->w[i], Iw_w);                       /* gyroscopic term (K1) */
    if (!E->opt_gyro) v3set(t, 0, 0, 0);
    for (int k = 0; k < 3; k++) { lin[k] = L->mass * K->acb[i][k] - F[k]; ang[k] += t[k] - T[k]; }
    for (int a = 0; a < nv; a++)
      for (int k = 0; k < 3; k++) h[a] += Jv[k][a] * lin[k] + Jw[k][a] * ang[k];
  }
}

static int cholesky(double* A, int n) { /* in place, lower */
  for (int j = 0; j < n; j++) {
    double d = A[j * n + j];
    for (int k = 0; k < j; k++) d -= A[j * n + k] * A[j * n + k];
    if (!(d > 0)) return -1;
    d = sqrt(d); A[j * n + j] = d;
    for (int i = j + 1; i < n; i++) {
      double v = A[i * n + j];
      for (int k = 0; k < j; k++) v -= A[i * n + k] * A[j * n + k];
      A[i * n + j] = v / d;
    }
  }
  return 0;
}
static void chol_solve(const double* Lm, int n, const double* b, double* x) {
  double y[NV_MAX];
  for (int i = 0; i < n; i++) { double v = b[i]; for (int k = 0; k < i; k++) v -= Lm[i * n + k] * y[k]; y[i] = v / Lm[i * n + i]; }
  for (int i = n - 1; i >= 0; i--) { double v = y[i]; for (int k = i + 1; k < n; k++) v -= Lm[k * n + i] * x[k]; x[i] = v / Lm[i * n + i]; }
}

/* world support point of primitive p towards the ground, and its height above z=0 */
static double prim_point(const oracle_env* E, const kin_t* K, int p, double* P) {
  const solorl_prim_data* pr = prim_of(E, p);
  int l = pr->link;
  double loc[3]; v3cpy(loc, pr->center);
  if (pr->axis >= 0) {
    double up[3] = {0, 0, 1}, d[3];
    m3tmulv(d, K->R[l], up);
    for (int k = 0; k < 3; k++) d[k] = -d[k];           /* world down in link axes */
    const double da = d[pr->axis];                       /* world down along the disc axis: 0 upright, +-1 lying on a face */
    d[pr->axis] = 0;                                     /* project off the disc axis */
    double inv = pr->radius / sqrt(v3dot(d, d) + DISC_EPS2);
    v3axpy(loc, inv, d);
    /* K6, round 3: the disc has a thickness (a cylinder with rounded rims): its lowest point moves from the mid-plane towards the face
     * that points down, reaching the face at a tilt of asin(DISC_RIM) -- so a link lying on its side rests half a thickness lower
     * than its mid-plane, as its hull does, while an upright leg keeps ONE smooth contact point (a sharp rim would make the point
     * jump by the full thickness whenever the roll changes sign) */
    if (pr->nring > 1) {
      /* round 4 (K6): a body of revolution given by its profile rings (the feet: a 4 mm tread of radius 16 mm chamfered to 11.7 mm at
       * the faces, tools/compile_model.py ring_profile): the support point lies on the ring maximising r_i |d_perp| + y_i |d_axis|;
       * ring 0, the tread, is crossed smoothly as above, the others are the hull's own edges */
      const double s = sqrt(v3dot(d, d) + DISC_EPS2), a = fabs(da);
      int best = 0; double vb = pr->ring_r[0] * s + pr->ring_y[0] * a;
      for (int i = 1; i < pr->nring; i++) { const double vi = pr->ring_r[i] * s + pr->ring_y[i] * a; if (vi > vb) { vb = vi; best = i; } }
      v3cpy(loc, pr->center);
      v3axpy(loc, pr->ring_r[best] / s, d);
      double w = a / DISC_RIM; w = w > 1 ? 1 : w;
      loc[pr->axis] += (da < 0 ? -1.0 : 1.0) * (best == 0 ? pr->ring_y[0] * w : pr->ring_y[best]);
    } else if (pr->halfw > 0) { double w = da / DISC_RIM; w = w > 1 ? 1 : (w < -1 ? -1 : w); loc[pr->axis] += pr->halfw * w; }
  }
  m3mulv(P, K->R[l], loc); v3add(P, P, K->o[l]);
  return P[2];
}

/* one contact point of a sub-step: link, world point, signed distance to the plane, friction, warm-start impulse, id (primitive
 * index, or 4 * link + manifold slot) */
typedef struct { int link, id, prim; double P[3], dist, friction, lam0; } cpoint_t;   /* prim: 1 = id is a primitive index */

static int collide_primitives(const oracle_env* E, const solorl_env_state* s, const kin_t* K, cpoint_t* cp, unsigned skip_links) {
  int n = 0;
  for (int p = 0; p < E->np; p++) {
    double P[3];
    if ((skip_links >> prim_of(E, p)->link) & 1u) continue;
    /* a collision margin m around a shape (its Minkowski sum with a sphere) lowers the support point towards the plane by m */
    const solorl_prim_data* pr = prim_of(E, p);
    if (E->opt_foot_points == 4 && pr->nring > 1 && pr->axis >= 0) {
      /* K6 cross-check: ONE point through the primitive path, but the support vertex of the link's ACTUAL hull (the vertex set the manifold
       * model searches) -- separates the hull's geometry from everything else collide_manifolds does */
      const int l = pr->link;
      const oracle_hull* H = &E->hulls[l];
      const double* R = K->R[l];
      int best = 0; double zb = 1e30;
      for (int i = 0; i < H->n; i++) { const double z = R[6] * H->v[i][0] + R[7] * H->v[i][1] + R[8] * H->v[i][2]; if (z < zb) { zb = z; best = i; } }
      double loc[3] = {H->v[best][0], H->v[best][1], H->v[best][2]};
      m3mulv(P, R, loc); v3add(P, P, K->o[l]);
      const double d4 = P[2] - E->cfg.collision_margin;
      P[2] = d4;
      if (d4 < pr->margin * E->opt_break_scale) {
        cpoint_t* c = &cp[n++];
        c->link = l; c->id = p; c->prim = 1; v3cpy(c->P, P); c->dist = d4; c->friction = pr->friction; c->lam0 = 0;
      }
      continue;
    }
    if (E->opt_foot_points >= 2 && pr->nring > 1 && pr->axis >= 0) {
      /* K6 experiment: the two edges of the tread, each its own contact point (sharp edges: no smooth crossing of the tread) */
      const int l = pr->link;
      double up[3] = {0, 0, 1}, dd[3];
      m3tmulv(dd, K->R[l], up);
      for (int k = 0; k < 3; k++) dd[k] = -dd[k];
      dd[pr->axis] = 0;
      const double inv = pr->ring_r[0] / sqrt(v3dot(dd, dd) + DISC_EPS2);
      for (int side = -1; side <= 1; side += 2) {
        double loc[3]; v3cpy(loc, pr->center);
        v3axpy(loc, inv, dd);
        loc[pr->axis] += side * pr->ring_y[0];
        if (E->opt_foot_points == 3) {
          /* ONE point: the support VERTEX of a 53-gon tread with two sharp edges (what the foot hull's own tread is: 53 vertices 6.79 degrees
           * apart on each edge ring, one of them at 180 degrees) -- the contact point jumps by 1.9 mm along the tread and by 4 mm across it */
          const int a0 = (pr->axis + 1) % 3, a1 = (pr->axis + 2) % 3;
          const double da = -K->R[l][6 + pr->axis];            /* world down along the wheel axis */
          if ((da >= 0 ? 1 : -1) != side) continue;
          const double step = 2.0 * M_PI / 53.0, th = atan2(dd[a1], dd[a0]);
          const double thq = M_PI + step * floor((th - M_PI) / step + 0.5);
          v3cpy(loc, pr->center);
          loc[a0] += pr->ring_r[0] * cos(thq); loc[a1] += pr->ring_r[0] * sin(thq);
          loc[pr->axis] += side * pr->ring_y[0];
        }
        m3mulv(P, K->R[l], loc); v3add(P, P, K->o[l]);
        const double d2 = P[2] - E->cfg.collision_margin;
        P[2] = d2;
        if (d2 < pr->margin * E->opt_break_scale) {
          cpoint_t* c = &cp[n++];
          c->link = l; c->id = p; c->prim = 1; v3cpy(c->P, P); c->dist = d2; c->friction = pr->friction; c->lam0 = 0;
        }
      }
      continue;
    }
    const double d = prim_point(E, K, p, P) - E->cfg.collision_margin;
    P[2] = d;
    if (d < prim_of(E, p)->margin * E->opt_break_scale) {
      cpoint_t* c = &cp[n++];
      c->link = prim_of(E, p)->link; c->id = p; c->prim = 1; v3cpy(c->P, P); c->dist = d; c->friction = prim_of(E, p)->friction; c->lam0 = s->lambda_prev[p];
    }
  }
  return n;
}

/* Bullet's convex-vs-plane path [K6], per link: btConvexPlaneCollisionAlgorithm::processCollision adds ONE point -- the hull's
 * support vertex towards the plane (margin included) when closer than the manifold's breaking threshold -- through
 * btManifoldResult::addContactPoint (nearest cached point within the threshold is replaced, else appended, a full cache of 4
 * replaces the entry that keeps the deepest point and the largest area: btPersistentManifold::sortCachedPoints), then
 * refreshContactPoints() recomputes every cached point's distance from its link-local point and drops those whose distance or
 * tangential drift exceeds the threshold.  Threshold = the link's relative breaking threshold (model table `margin`). */
static int collide_manifolds(oracle_env* E, int ei, const kin_t* K, cpoint_t* cp) {
  int n = 0;
  for (int l = 0; l < E->nl; l++) {
    if (!((E->manifold_links >> l) & 1u)) continue;
    manifold_t* m = &E->man[(size_t)ei * NL_MAX + l];
    if (!E->opt_manifold_persist) m->n = 0;
    const oracle_hull* H = &E->hulls[l];
    double thr = 0, fric = 0.5;
    int have = 0;
    for (int p = 0; p < E->np; p++) if (prim_of(E, p)->link == l) { thr = prim_of(E, p)->margin * E->opt_break_scale; fric = prim_of(E, p)->friction; have = 1; break; }
    if (!have) {   /* links without a primitive (lower legs): threshold from the hull's own bounding sphere about the COM */
      double lo[3] = {1e9, 1e9, 1e9}, hi[3] = {-1e9, -1e9, -1e9}, c[3], hx[3];
      for (int i = 0; i < H->n; i++) for (int k = 0; k < 3; k++) { if (H->v[i][k] < lo[k]) lo[k] = H->v[i][k]; if (H->v[i][k] > hi[k]) hi[k] = H->v[i][k]; }
      for (int k = 0; k < 3; k++) { c[k] = 0.5 * (lo[k] + hi[k]) - E->md->links[l].com[k]; hx[k] = 0.5 * (hi[k] - lo[k]) + E->cfg.collision_margin; }
      thr = 0.02 * (v3norm(c) + v3norm(hx)) * E->opt_break_scale; fric = 1.0;   /* URDF <contact> friction of the lower legs is 1.0 */
    }
    /* support vertex: lowest hull vertex in the world (plane normal +z) */
    const double* R = K->R[l];
    int best = 0; double zb = 1e30;
    for (int i = 0; i < H->n; i++) {
      const double z = R[6] * H->v[i][0] + R[7] * H->v[i][1] + R[8] * H->v[i][2];
      if (z < zb) { zb = z; best = i; }
    }
    const double dist = zb + K->o[l][2] - E->cfg.collision_margin;
    if (dist < thr) {
      /* localA = link-frame coordinates of (vertex - margin * normal): what btManifoldResult stores as m_localPointA */
      double down[3] = {0, 0, -E->cfg.collision_margin}, dl[3], loc[3], wA[3];
      m3tmulv(dl, R, down);
      for (int k = 0; k < 3; k++) loc[k] = H->v[best][k] + dl[k];
      m3mulv(wA, R, loc); v3add(wA, wA, K->o[l]);
      int slot = -1; double shortest = thr * thr;
      for (int i = 0; i < m->n; i++) {            /* getCacheEntry */
        double d[3]; v3sub(d, m->local[i], loc);
        const double dd = v3dot(d, d);
        if (dd < shortest) { shortest = dd; slot = i; }
      }
      int fresh = 0;
      if (slot < 0) {
        fresh = 1;
        if (m->n < 4) slot = m->n++;
        else {                                    /* sortCachedPoints: keep the deepest, maximise the area (gContactCalcArea3Points) */
          int deep = -1; double maxpen = dist;
          for (int i = 0; i < 4; i++) if (m->dist[i] < maxpen) { deep = i; maxpen = m->dist[i]; }
          double res[4] = {0, 0, 0, 0}, a[3], b[3], c[3];
          if (deep != 0) { v3sub(a, loc, m->local[1]); v3sub(b, m->local[3], m->local[2]); v3cross(c, a, b); res[0] = v3dot(c, c); }
          if (deep != 1) { v3sub(a, loc, m->local[0]); v3sub(b, m->local[3], m->local[2]); v3cross(c, a, b); res[1] = v3dot(c, c); }
          if (deep != 2) { v3sub(a, loc, m->local[0]); v3sub(b, m->local[3], m->local[1]); v3cross(c, a, b); res[2] = v3dot(c, c); }
          if (deep != 3) { v3sub(a, loc, m->local[0]); v3sub(b, m->local[2], m->local[1]); v3cross(c, a, b); res[3] = v3dot(c, c); }
          slot = 0;
          for (int i = 1; i < 4; i++) if (res[i] > res[slot]) slot = i;
        }
      }
      v3cpy(m->local[slot], loc);
      m->worldB[slot][0] = wA[0]; m->worldB[slot][1] = wA[1]; m->worldB[slot][2] = 0;   /* the vertex projected on the plane */
      m->dist[slot] = dist;
      if (fresh) m->lam[slot] = 0;                /* replaceContactPoint keeps the cached impulse, a new point starts at 0 */
    }
    for (int i = m->n - 1; i >= 0; i--) {         /* refreshContactPoints */
      double wA[3];
      m3mulv(wA, R, m->local[i]); v3add(wA, wA, K->o[l]);
      m->dist[i] = wA[2];
      const double dx = m->worldB[i][0] - wA[0], dy = m->worldB[i][1] - wA[1];
      if (m->dist[i] > thr || dx * dx + dy * dy > thr * thr) {   /* removeContactPoint: the last entry takes its place */
        const int last = --m->n;
        if (i != last) { v3cpy(m->local[i], m->local[last]); v3cpy(m->worldB[i], m->worldB[last]); m->dist[i] = m->dist[last]; m->lam[i] = m->lam[last]; }
      }
    }
    for (int i = 0; i < m->n; i++) {
      cpoint_t* c = &cp[n++];
      c->link = l; c->id = 4 * l + i; c->prim = 0; c->dist = m->dist[i]; c->friction = fric; c->lam0 = m->lam[i];
      m3mulv(c->P, R, m->local[i]); v3add(c->P, c->P, K->o[l]);
    }
  }
  return n;
}

typedef struct { double J[NV_MAX], B[NV_MAX], rhs, dinv, lam, mu; int parent; /* -1: lo=0,hi=inf */ } row_t;

static void clampv(double* v, double lim) { if (*v > lim) *v = lim; if (*v < -lim) *v = -lim; }

static void substep(oracle_env* E, int ei) {
  solorl_env_state* s = &E->st[ei];
  const solorl_config* C = &E->cfg;
  const int nv = E->nv, n = E->n;
  const double dt = C->sim_dt;
  kin_t K;
  kinematics(E, s, &K);

  /* K5: every candidate (joint at or beyond a limit) gets its row, in joint order.  With oracle_set_caps (engine emulation)
   * only the cap_limits candidates with the SMALLEST margin, i.e. the most violated first (ties: lower joint, lower side) -- the
   * engine's rule since round 2 (the first two in joint order let a third joint run radians past its limit and be thrown back). */
  row_t rows[MAX_ROWS]; int nr = 0;
  int lim_sel[2 * SOLORL_MAX_DOF]; memset(lim_sel, 0, sizeof lim_sel);
  {
    double pens[2 * SOLORL_MAX_DOF]; int cand[2 * SOLORL_MAX_DOF], ncand = 0, nsel = 0;
    for (int j = 0; j < n; j++)
      for (int side = 0; side < 2; side++) {
        pens[2 * j + side] = side == 0 ? s->q[j] + C->joint_limit : C->joint_limit - s->q[j];
        if (pens[2 * j + side] <= LIMIT_WINDOW) cand[ncand++] = 2 * j + side;
      }
    for (int a = 0; a < ncand; a++) {
      int rank = 0;
      for (int b2 = 0; b2 < ncand; b2++)
        rank += pens[cand[b2]] < pens[cand[a]] || (pens[cand[b2]] == pens[cand[a]] && cand[b2] < cand[a]);
      lim_sel[cand[a]] = E->cap_limits <= 0 || rank < E->cap_limits;
      nsel += lim_sel[cand[a]];
    }
    E->last_counts[4 * ei + 2] = ncand; E->last_counts[4 * ei + 3] = nsel;
  }
  const int nsel_limits = E->last_counts[4 * ei + 3];
  /* --- collision detection at the start-of-step pose (K1, K6): the list of contact points */
  cpoint_t cp[NPT_MAX]; int nact = 0;
  if (E->contact_model == 0) nact = collide_primitives(E, s, &K, cp, 0u);
  else { nact = collide_manifolds(E, ei, &K, cp); nact += collide_primitives(E, s, &K, cp + nact, E->manifold_links); }
  E->last_counts[4 * ei + 0] = nact;
  /* engine emulation (oracle_set_caps): the engine's limit rows 3 and 4 take the place of contact points (dynamics.hpp MAX_LIMITS) */
  const int capc = E->cap_contacts - (nsel_limits > 2 ? nsel_limits - 2 : 0);
  if (E->cap_contacts > 0 && nact > capc) { /* keep the deepest (ties: lower id) */
    int keep[NPT_MAX];
    for (int a = 0; a < nact; a++) {
      int rank = 0;
      for (int b2 = 0; b2 < nact; b2++)
        if (cp[b2].dist < cp[a].dist || (cp[b2].dist == cp[a].dist && cp[b2].id < cp[a].id)) rank++;
      keep[a] = rank < capc;
    }
    int m = 0;
    for (int a = 0; a < nact; a++) if (keep[a]) cp[m++] = cp[a];
    nact = m;
  }
  E->last_counts[4 * ei + 1] = nact;

  /* --- unconstrained forward dynamics: u* = u + dt M^-1 (tau - h)  (K1) */
  double M[NV_MAX * NV_MAX], h[NV_MAX], rhsv[NV_MAX], udot[NV_MAX], u[NV_MAX];
  mass_and_bias(E, s, &K, M, h);
  cholesky(M, nv);
  for (int k = 0; k < 6; k++) rhsv[k] = -h[k];
  for (int j = 0; j < n; j++) rhsv[6 + j] = s->tau[j] - h[6 + j];
  chol_solve(M, nv, rhsv, udot);
  for (int k = 0; k < 3; k++) { u[k] = s->ang_vel[k]; u[3 + k] = s->lin_vel[k]; }
  for (int j = 0; j < n; j++) u[6 + j] = s->qd[j];
  for (int k = 0; k < nv; k++) { u[k] += dt * udot[k]; clampv(&u[k], C->max_velocity); } /* K5 */

  /* --- constraint rows: joint limits (selected above), contact normals, friction (K7) */
  for (int j = 0; j < n; j++) {
    for (int side = 0; side < 2; side++) {
      if (!lim_sel[2 * j + side]) continue;
      double pen = side == 0 ? s->q[j] + C->joint_limit : C->joint_limit - s->q[j];
      row_t* r = &rows[nr++];
      memset(r, 0, sizeof *r);
      r->J[6 + j] = side == 0 ? 1.0 : -1.0; r->parent = -1; r->mu = 0;
      r->rhs = pen; /* stash penetration; finished below */
    }
  }
  int normal_row[NPT_MAX];
  for (int a = 0; a < nact; a++) {
    double nrm[3] = {0, 0, 1};
    row_t* r = &rows[nr]; memset(r, 0, sizeof *r);
    point_jacobian_row(E, s, &K, cp[a].link, cp[a].P, nrm, r->J);
    r->parent = -1; r->rhs = cp[a].dist + C->linear_slop; normal_row[a] = nr++;
  }
  int nnormal_end = nr;
  /* treadmill (reference simulation.py:45-77; model: include/solorl.h treadmill_*): a contact whose point lies on
   * the strip takes the strip's friction (Bullet combines by product); the feet sensor still reports it (plane contact) */
  int on_strip[NPT_MAX];
  for (int a = 0; a < nact; a++) on_strip[a] = C->use_treadmill && fabs(cp[a].P[1] - s->treadmill_y) <= C->treadmill_half_width;
  for (int a = 0; a < nact; a++) {
    for (int d = 0; d < 2; d++) {
      /* friction directions: world x then y (model 0, the engine's order); model 1: btPlaneSpace1 of the normal (0,0,1) gives
       * (0,-1,0) then (1,0,0) [K] */
      double t[3] = {d == 0, d == 1, 0};
      if (E->contact_model == 1) { t[0] = d == 1; t[1] = d == 0 ? -1 : 0; }
      row_t* r = &rows[nr++]; memset(r, 0, sizeof *r);
      point_jacobian_row(E, s, &K, cp[a].link, cp[a].P, t, r->J);
      r->parent = normal_row[a]; r->mu = cp[a].friction * (on_strip[a] ? C->treadmill_friction : 1.0);
    }
  }
  double dV[NV_MAX]; memset(dV, 0, sizeof dV);
  for (int i = 0; i < nr; i++) {
    row_t* r = &rows[i];
    chol_solve(M, nv, r->J, r->B);
    double denom = 0, rel = 0;
    for (int k = 0; k < nv; k++) { denom += r->J[k] * r->B[k]; rel += r->J[k] * u[k]; }
    r->dinv = 1.0 / denom;
    if (i < nnormal_end) { /* limit or normal row: speculative / ERP right-hand side */
      /* Bullet keeps two error-reduction parameters: btContactSolverInfo::m_erp (non-contact rows: the joint limits, 0.2) and m_erp2
       * (contact rows; btMultiBodyConstraintSolver::setupMultiBodyContactConstraint reads m_erp2) -- solorl_config erp / contact_erp [K] */
      double pen = r->rhs, pos = 0, vel = -rel, erp = i < nsel_limits ? C->erp : C->contact_erp;
      /* [K] option: btMultiBodyJointLimitConstraint::createConstraintRows takes m_erp and the combined right-hand side only while
       * penetration > m_splitImpulsePenetrationThreshold (-0.04); deeper, the positional part goes to m_rhsPenetration, which no
       * multibody pass ever solves */
      if (i < nsel_limits && E->opt_limit_split && pen <= -0.04) erp = 0;
      if (pen > 0) vel -= pen / dt; else pos = -pen * erp / dt;
      r->rhs = (pos + vel) * r->dinv;
    } else r->rhs = -rel * r->dinv;
    r->lam = 0;
  }
  for (int a = 0; a < nact; a++) { /* warm start cached normal impulses */
    row_t* r = &rows[normal_row[a]];
    r->lam = C->warmstart * cp[a].lam0;
    for (int k = 0; k < nv; k++) dV[k] += r->B[k] * r->lam;
  }
  for (int it = 0; it < C->solver_iterations; it++) {
    double resid = 0;   /* K7: btMultiBodyConstraintSolver::solveSingleIteration's leastSquaresResidual = max (deltaImpulse / jacDiagABInv)^2 */
    for (int i = 0; i < nr; i++) {
      row_t* r = &rows[i];
      if (r->parent >= 0 && C->friction_model == SOLORL_FRICTION_CONE) {
        /* Bullet's implicit friction cone [K]: btMultiBodyConstraintSolver::solveSingleIteration, with SOLVER_USE_2_FRICTION_DIRECTIONS (set by
         * btMultiBodyDynamicsWorld's constructor) and without SOLVER_DISABLE_IMPLICIT_CONE_FRICTION (PyBullet's enableConeFriction default),
         * hands the two friction rows of a contact to resolveConeFrictionConstraintRows: both unclamped sums are taken against the SAME
         * delta-velocities (no Gauss-Seidel step between the two directions), the pair is projected radially onto the disc of radius
         * mu * normal impulse (atan2 / sin / cos in the source: |limit sin|, |limit cos| are the projected components), both deltas are
         * applied, and the residual entry is the SUM of the two velocity-level changes. */
        row_t* r2 = &rows[i + 1];
        double ja = 0, jb = 0;
        for (int k = 0; k < nv; k++) { ja += r->J[k] * dV[k]; jb += r2->J[k] * dV[k]; }
        double sa = r->lam + (r->rhs - ja * r->dinv), sb = r2->lam + (r2->rhs - jb * r2->dinv);
        const double lim = r->mu * rows[r->parent].lam;
        if (sa * sa + sb * sb >= lim * lim) {
          const double ang = atan2(sa, sb), ca = fabs(lim * sin(ang)), cb = fabs(lim * cos(ang));
          if (sa < -ca) sa = -ca; else if (sa > ca) sa = ca;
          if (sb < -cb) sb = -cb; else if (sb > cb) sb = cb;
        }
        const double da = sa - r->lam, db = sb - r2->lam;
        r->lam = sa; r2->lam = sb;
        for (int k = 0; k < nv; k++) dV[k] += r->B[k] * da + r2->B[k] * db;
        { double dvel = da / r->dinv + db / r2->dinv; if (dvel * dvel > resid) resid = dvel * dvel; }
        i++;
        continue;
      }
      /* [K] option: the pyramid branch of solveSingleIteration solves a friction row only `if (totalImpulse > 0)` */
      if (r->parent >= 0 && E->opt_friction_skip && !(rows[r->parent].lam > 0)) continue;
      double jdv = 0;
      for (int k = 0; k < nv; k++) jdv += r->J[k] * dV[k];
      double delta = r->rhs - jdv * r->dinv, sum = r->lam + delta, lo = 0, hi = 1e30;
      if (r->parent >= 0) { hi = r->mu * rows[r->parent].lam; lo = -hi; }
      if (sum < lo) sum = lo; if (sum > hi) sum = hi;
      delta = sum - r->lam; r->lam = sum;
      for (int k = 0; k < nv; k++) dV[k] += r->B[k] * delta;
      { double dvel = delta / r->dinv; if (dvel * dvel > resid) resid = dvel * dvel; }
    }
    E->last_iterations[ei] = it + 1; E->last_resid[ei] = resid;
    /* solveGroupCacheFriendlyIterations: stop once the residual is within solverResidualThreshold (PyBullet default 1e-7) */
    if (C->solver_residual_threshold > 0 && resid <= C->solver_residual_threshold) break;
  }
  if (nr == 0) E->last_iterations[ei] = 0;
  else { int b = E->last_iterations[ei]; E->iter_hist[ei][b < ITER_HIST ? b : ITER_HIST - 1]++; }
  for (int k = 0; k < nv; k++) { u[k] += dV[k]; clampv(&u[k], C->max_velocity); }
  for (int p = 0; p < NP_MAX; p++) { s->lambda_prev[p] = 0; E->last_lambda[ei][p] = 0; }
  s->contact_mask = 0;
  for (int a = 0; a < nact; a++) {
    const double lam = rows[normal_row[a]].lam;
    if (cp[a].prim) {
      const int p = cp[a].id;
      s->lambda_prev[p] = lam; E->last_lambda[ei][p] = lam;
      s->contact_mask |= 1 << p;
      for (int f = 0; f < 4; f++) if (on_strip[a] && E->md->foot_prim[f] == p) s->contact_mask |= 1 << (24 + f);
    } else {
      /* manifold model: the impulse goes back into its manifold point; mask bit of a link's representative primitive (the feet
       * sensor, solo.py:310-323, asks for ANY contact point of the foot link) */
      E->man[(size_t)ei * NL_MAX + cp[a].link].lam[cp[a].id & 3] = lam;
      for (int p = 0; p < E->np; p++)
        if (prim_of(E, p)->link == cp[a].link) {
          s->contact_mask |= 1 << p;
          for (int f = 0; f < 4; f++) if (on_strip[a] && E->md->foot_prim[f] == p) s->contact_mask |= 1 << (24 + f);
          break;
        }
    }
  }

  /* --- semi-implicit Euler position update with the new velocities (K1) */
  for (int k = 0; k < 3; k++) { s->ang_vel[k] = u[k]; s->lin_vel[k] = u[3 + k]; s->pos[k] += dt * u[3 + k]; }
  for (int j = 0; j < n; j++) { s->qd[j] = u[6 + j]; s->q[j] += dt * u[6 + j]; }
  {
    double wn = v3norm(s->ang_vel), sc, cw;
    if (wn < 1e-3) sc = 0.5 * dt - dt * dt * dt * (1.0 / 48.0) * wn * wn; else sc = sin(0.5 * wn * dt) / wn;
    cw = cos(0.5 * wn * dt);
    double ax = s->ang_vel[0] * sc, ay = s->ang_vel[1] * sc, az = s->ang_vel[2] * sc;
    double x = s->quat[0], y = s->quat[1], z = s->quat[2], w = s->quat[3];
    double nx = cw * x + ax * w + ay * z - az * y;   /* dq (x) q */
    double ny = cw * y - ax * z + ay * w + az * x;
    double nz = cw * z + ax * y - ay * x + az * w;
    double nw = cw * w - ax * x - ay * y - az * z;
    double inv = 1.0 / sqrt(nx * nx + ny * ny + nz * nz + nw * nw);
    s->quat[0] = nx * inv; s->quat[1] = ny * inv; s->quat[2] = nz * inv; s->quat[3] = nw * inv;
  }
  if (!C->hold_torque) for (int j = 0; j < n; j++) s->tau[j] = 0;   /* K8: cleared after every stepSimulation */
}

/* ------------------------------------------------------------------ env logic */
/* SoloBase.get_feet_ground_contact, solo.py:310-323: contact points between the foot link and `ground_id`, the PLANE body.
 * The infinite plane also lies under the zero-height treadmill strip (simulation.py:23 loads it before the strip, :59-64),
 * so a foot over the strip still has its plane contact and IS reported; bit 24+f of the mask only records that the
 * foot's contact point lies on the strip (friction, see substep) -- it does not mask the sensor (round 2 did). */
static int foot_contact(const oracle_env* E, const solorl_env_state* s, int f) {
  return (s->contact_mask >> E->md->foot_prim[f]) & 1;
}

/* SoloBase.get_current_state, reference solo.py:198-222 */
static void current_state(const oracle_env* E, const solorl_env_state* s, double* out) {
  int k = 0, n = E->n;
  double rpy[3];
  out[k++] = s->pos[2];
  oracle_euler_from_quat(s->quat, rpy);
  for (int i = 0; i < 3; i++) { double e = rpy[i]; e = e - 2.0 * floor(e / 2.0); out[k++] = e / 2.0; } /* (e % 2*pi)/(2*pi), solo.py:206 */
  for (int i = 0; i < 3; i++) out[k++] = s->lin_vel[i];
  for (int i = 0; i < 3; i++) out[k++] = s->ang_vel[i];
  for (int j = 0; j < n; j++) out[k++] = s->q[j] / E->cfg.joint_limit;
  for (int j = 0; j < n; j++) out[k++] = s->qd[j] / 100.0;           /* solo.py:110 joint_vel_limit */
  for (int f = 0; f < 4; f++) out[k++] = foot_contact(E, s, f) ? 1.0 : 0.0;
  if (E->cfg.task == SOLORL_TASK_POINTGOAL) {
    out[k++] = s->pos[0] / 2; out[k++] = s->pos[1] / 2; out[k++] = s->goal[0] / 2; out[k++] = s->goal[1] / 2;
  }
}
/* SoloBase.calc_state, solo.py:186-196 */
static void calc_state(const oracle_env* E, const solorl_env_state* s, double* obs) {
  current_state(E, s, obs);
  for (int hI = 0; hI < E->cfg.num_history_stack; hI++)
    for (int k = 0; k < E->D; k++) obs[(hI + 1) * E->D + k] = obs[k] - s->hist[hI][k];
}
static void push_history(const oracle_env* E, solorl_env_state* s) {
  int H = E->cfg.num_history_stack;
  if (H == 0) return;
  double cur[SOLORL_STATE_MAX_OBS];
  current_state(E, s, cur);
  for (int hI = H - 1; hI > 0; hI--) memcpy(s->hist[hI], s->hist[hI - 1], sizeof(double) * E->D);
  memcpy(s->hist[0], cur, sizeof(double) * E->D);
}
static void sample_goal(const oracle_env* E, int gi, solorl_env_state* s) { /* solo.py:325-330 */
  uint32_t r[4];
  uint64_t gid = (uint64_t)(E->id0 + gi);
  oracle_philox((uint32_t)E->seed, (uint32_t)(E->seed >> 32), (uint32_t)gid, (uint32_t)(gid >> 32),
                (uint32_t)s->rng_counter++, 1u, r);
  double x = 1.0 + u01(r[0]) * (E->goal_radius - 1.0), y = 1.0 + u01(r[1]) * (E->goal_radius - 1.0);
  s->goal[0] = (r[2] & 1u) ? x : -x; s->goal[1] = (r[3] & 1u) ? y : -y;
}
static double potential(const solorl_env_state* s) {
  double dx = s->pos[0] - s->goal[0], dy = s->pos[1] - s->goal[1];
  return sqrt(dx * dx + dy * dy);
}
/* SoloBase.simulator_step, solo.py:261-274 */
static void simulator_step(oracle_env* E, int i) {
  solorl_env_state* s = &E->st[i];
  push_history(E, s);
  for (int k = 0; k < E->cfg.frame_skip; k++) substep(E, i);
  for (int j = 0; j < E->n; j++) s->tau[j] = 0;   /* hold_torque=1 holds it for this control step only */
  if (E->cfg.task == SOLORL_TASK_POINTGOAL) {
    double old = s->potential;
    s->potential = potential(s);
    s->progress = -(s->potential - old);
    if (s->potential < 0.5) { s->goals_reached += 1; sample_goal(E, i, s); }
  }
}
/* SoloBaseEnv.reset, baseEnv.py:70-82 (+ SoloBase.reset solo.py:166-181, :291-296) */
static void env_reset(oracle_env* E, int i) {
  solorl_env_state* s = &E->st[i];
  int rc = s->rng_counter;
  double g0 = s->goal[0], g1 = s->goal[1];
  memset(s, 0, sizeof *s);
  memset(&E->man[(size_t)i * NL_MAX], 0, sizeof(manifold_t) * NL_MAX);
  s->rng_counter = rc; s->goal[0] = g0; s->goal[1] = g1;
  s->pos[2] = 0.35; s->quat[3] = 1.0;                  /* solo.py:52,292-293 */
  if (E->cfg.use_treadmill) {                          /* scene.reset -> Treadmill.reset, simulation.py:37-41,72-74: new side */
    uint32_t rt[4];
    uint64_t gidt = (uint64_t)(E->id0 + i);
    oracle_philox((uint32_t)E->seed, (uint32_t)(E->seed >> 32), (uint32_t)gidt, (uint32_t)(gidt >> 32),
                  (uint32_t)s->rng_counter++, 3u, rt);
    s->treadmill_y = (rt[0] & 1u) ? E->cfg.treadmill_offset : -E->cfg.treadmill_offset;
  }
  for (int hI = 0; hI < E->cfg.num_history_stack; hI++) current_state(E, s, s->hist[hI]);
  if (E->cfg.task == SOLORL_TASK_POINTGOAL) {
    sample_goal(E, i, s);   /* after the history fill, as solo.py:170-174: history keeps the old goal */
    s->goals_reached = 0; s->potential = potential(s); s->progress = 0;
  }
  uint32_t r[4];
  uint64_t gid = (uint64_t)(E->id0 + i);
  oracle_philox((uint32_t)E->seed, (uint32_t)(E->seed >> 32), (uint32_t)gid, (uint32_t)(gid >> 32),
                (uint32_t)s->rng_counter++, 2u, r);
  int K = E->cfg.settle_min + (int)(r[0] % (uint32_t)(E->cfg.settle_max - E->cfg.settle_min + 1));
  for (int k = 0; k < K; k++) simulator_step(E, i);
}

static void apply_action(const oracle_env* E, solorl_env_state* s, const double* a) { /* solo.py:224-259 */
  for (int j = 0; j < E->n; j++) {
    double c = a[j] < -1 ? -1 : (a[j] > 1 ? 1 : a[j]);
    if (E->cfg.control == SOLORL_CONTROL_TORQUE) s->tau[j] = c * E->cfg.max_torque;
    else {
      double qref = c * E->cfg.joint_limit;
      double t = E->cfg.kp * (qref - s->q[j]) - E->cfg.kd * s->qd[j];   /* controllers/PD.py:5 */
      if (t > E->cfg.max_torque) t = E->cfg.max_torque; if (t < -E->cfg.max_torque) t = -E->cfg.max_torque;
      s->tau[j] = t;
    }
  }
}

static void env_step(oracle_env* E, int i, const double* a, double* obs, double* rew, uint8_t* done,
                     uint8_t* timeout, uint8_t* success, int32_t* ep_len, double* ep_rew, double* goals, double* dr) {
  solorl_env_state* s = &E->st[i];
  const solorl_config* C = &E->cfg;
  int n = E->n;
  apply_action(E, s, a);
  simulator_step(E, i);
  s->timestep += 1;
  /* reward, baseEnv.py:91-157 */
  double z = s->pos[2];
  double stand = (z > 0.2) ? 0.5 : 0.0, jp = 0, balance = 0, progress = 0, torque = 0;
  for (int j = 0; j < n; j++) jp += (C->task == SOLORL_TASK_STAND) ? fabs(s->q[j]) : s->q[j] * s->q[j];
  jp = -0.1 * jp / n;
  if (C->task == SOLORL_TASK_WALK) {
    if (z > 0.2) { double vx = s->lin_vel[0]; progress = 2.0 * ((vx > 0) - (vx < 0)) * vx * vx; }
  } else if (C->task == SOLORL_TASK_POINTGOAL) {
    double rpy[3]; oracle_euler_from_quat(s->quat, rpy);
    balance = -0.1 * (fabs(rpy[0]) + fabs(rpy[1]));
    if (z > 0.2) progress = s->progress * (1.0 / C->reward_dt);
  }
  if (C->control == SOLORL_CONTROL_TORQUE) { double tp = 0; for (int j = 0; j < n; j++) tp += a[j] * a[j]; torque = -0.01 * tp; }
  double reward = stand + jp + balance + progress + torque;
  s->dr[0] += stand; s->dr[1] += jp; s->dr[2] += torque; s->dr[3] += balance; s->dr[4] += progress;
  /* termination, baseEnv.py:162-180 */
  int d = 0, to = 0, su = 0;
  if (!C->disable_termination) {
    if (s->timestep >= C->episode_length) { d = 1; to = 1; su = C->task != SOLORL_TASK_POINTGOAL; }
    else if (z < 0.05) { d = 1; }
    else if (C->task == SOLORL_TASK_POINTGOAL && s->goals_reached > s->env_goals_reached) {
      s->env_goals_reached = s->goals_reached; d = 1; su = 1;
    }
  }
  if (d) { /* baseEnv.py:52-60 */
    if (su) { if (C->task == SOLORL_TASK_POINTGOAL) reward = 0.1 * (C->episode_length - s->timestep); }
    else if (!to) reward = -10.0;
  }
  if (rew) rew[i] = reward;
  if (done) done[i] = (uint8_t)d;
  if (timeout) timeout[i] = (uint8_t)to;
  if (success) success[i] = (uint8_t)su;
  if (ep_len) ep_len[i] = s->timestep;
  if (ep_rew) ep_rew[i] = reward;
  if (goals) goals[i] = s->env_goals_reached;
  if (dr) for (int k = 0; k < 5; k++) dr[5 * i + k] = s->dr[k];
  if (d) env_reset(E, i);                               /* agents/ppo/envs.py:39 */
  calc_state(E, s, obs + (size_t)i * E->O);
}

/* ------------------------------------------------------------------ public API */
oracle_env* oracle_create(const solorl_config* cfg, int num_envs, uint64_t seed, int64_t env_id_offset) {
  oracle_env* E = (oracle_env*)calloc(1, sizeof *E);
  E->cfg = *cfg;
  E->md = cfg->robot == SOLORL_ROBOT_SOLO12 ? &SOLORL_MODEL_SOLO12 : &SOLORL_MODEL_SOLO8;
  E->N = num_envs; E->n = E->md->ndof; E->nv = 6 + E->n; E->nl = E->md->nlinks; E->np = E->md->nprims;
  E->D = 14 + 2 * E->n + (cfg->task == SOLORL_TASK_POINTGOAL ? 4 : 0);
  E->O = E->D * (1 + cfg->num_history_stack);
  E->seed = seed; E->id0 = env_id_offset; E->goal_radius = cfg->goal_radius; E->nthreads = 1;
  for (int i = 0; i < E->nl; i++) {
    const solorl_link_data* L = &E->md->links[i];
    if (cfg->use_urdf_inertia) memcpy(E->inertia[i], L->inertia_urdf, sizeof(double) * 6);
    else { E->inertia[i][0] = L->inertia_box[0]; E->inertia[i][1] = L->inertia_box[1]; E->inertia[i][2] = L->inertia_box[2]; }
  }
  E->st = (solorl_env_state*)calloc((size_t)num_envs, sizeof *E->st);
  E->last_lambda = calloc((size_t)num_envs, sizeof *E->last_lambda);
  E->last_iterations = calloc((size_t)num_envs, sizeof *E->last_iterations);
  E->last_resid = calloc((size_t)num_envs, sizeof *E->last_resid);
  E->last_counts = calloc((size_t)num_envs * 4, sizeof *E->last_counts);
  E->man = calloc((size_t)num_envs * NL_MAX, sizeof *E->man);
  E->iter_hist = calloc((size_t)num_envs, sizeof *E->iter_hist);
  E->opt_friction_skip = 0; E->opt_gyro = 1; E->opt_limit_split = 0; E->opt_break_scale = 1; E->opt_manifold_persist = 1; E->opt_foot_points = 1;
  E->hulls = cfg->robot == SOLORL_ROBOT_SOLO12 ? ORACLE_HULLS_SOLO12 : ORACLE_HULLS_SOLO8;
  E->contact_model = 0; E->cap_contacts = 0; E->cap_limits = 0; E->manifold_links = ~0u;
  if (getenv("ORACLE_MANIFOLD_LINKS")) E->manifold_links = (unsigned)strtoul(getenv("ORACLE_MANIFOLD_LINKS"), NULL, 0);
  if (getenv("ORACLE_LIMIT_WINDOW")) LIMIT_WINDOW = atof(getenv("ORACLE_LIMIT_WINDOW"));
  if (getenv("ORACLE_CONTACT_MODEL")) E->contact_model = atoi(getenv("ORACLE_CONTACT_MODEL")) != 0;
  /* K6 measurement hook (tests/test_oracle_k6.py): ORACLE_NO_SHOULDERS=1 drops the Solo12 shoulder-housing discs
   * (primitives 20..23) again, i.e. the round-1 primitive set, so that what they change stays measurable */
  if (getenv("ORACLE_NO_SHOULDERS") && atoi(getenv("ORACLE_NO_SHOULDERS")) && E->np > 20) E->np = 20;
  for (int i = 0; i < num_envs; i++) { E->st[i].quat[3] = 1; E->st[i].pos[2] = 0.35; E->st[i].need_reset = 1; }
  return E;
}
void oracle_destroy(oracle_env* E) { if (!E) return; free(E->st); free(E->last_lambda); free(E->last_iterations); free(E->last_resid); free(E->last_counts); free(E->man); free(E->iter_hist); free(E); }
int oracle_last_iterations(const oracle_env* E, int i) { return E->last_iterations[i]; }
double oracle_last_residual(const oracle_env* E, int i) { return E->last_resid[i]; }
int oracle_set_option(oracle_env* E, const char* name, double v) {
  if (!strcmp(name, "friction_skip_zero_normal")) E->opt_friction_skip = v != 0;
  else if (!strcmp(name, "gyro")) E->opt_gyro = v != 0;
  else if (!strcmp(name, "limit_split")) E->opt_limit_split = v != 0;
  else if (!strcmp(name, "breaking_scale")) E->opt_break_scale = v;
  else if (!strcmp(name, "manifold_persist")) E->opt_manifold_persist = v != 0;
  else if (!strcmp(name, "foot_points")) E->opt_foot_points = v >= 4 ? 4 : (v >= 3 ? 3 : (v >= 2 ? 2 : 1));   /* 4: the support vertex of the link's actual hull, through the primitive path */   /* 3: the support vertex of the polygonal tread (one point) */
  else return -1;
  return 0;
}
void oracle_iteration_histogram(oracle_env* E, int64_t* out /* [ITER_HIST = 128] */, int clear) {
  memset(out, 0, sizeof(int64_t) * ITER_HIST);
  for (int i = 0; i < E->N; i++) for (int b = 0; b < ITER_HIST; b++) out[b] += E->iter_hist[i][b];
  if (clear) memset(E->iter_hist, 0, sizeof(*E->iter_hist) * (size_t)E->N);
}
void oracle_set_caps(oracle_env* E, int max_contacts, int max_limits) { E->cap_contacts = max_contacts; E->cap_limits = max_limits; }
void oracle_set_contact_model(oracle_env* E, int model) { E->contact_model = model != 0; memset(E->man, 0, sizeof(manifold_t) * (size_t)E->N * NL_MAX); }
void oracle_last_counts(const oracle_env* E, int i, int out[4]) { memcpy(out, &E->last_counts[4 * i], sizeof(int) * 4); }
void oracle_dims(const oracle_env* E, int* o, int* a, int* n) { if (o) *o = E->O; if (a) *a = E->n; if (n) *n = E->N; }
void oracle_set_threads(oracle_env* E, int t) { E->nthreads = t < 1 ? 1 : t; }
/* OpenMP over envs (independent units): dynamic chunks -- an env-step costs 2-10 x more with contacts than in free flight -- of up to 16
 * envs (whole cache lines of the int32 / double output arrays), smaller when there are few envs per thread (64 envs on 16 threads) */
static int omp_chunk(const oracle_env* E) { int c = E->N / (4 * E->nthreads); return c < 1 ? 1 : (c > 16 ? 16 : c); }
void oracle_reset(oracle_env* E, double* obs) {
  const int chunk = omp_chunk(E);
#pragma omp parallel for schedule(dynamic, chunk) num_threads(E->nthreads)
  for (int i = 0; i < E->N; i++) { env_reset(E, i); E->st[i].need_reset = 0; calc_state(E, &E->st[i], obs + (size_t)i * E->O); }
}
void oracle_step(oracle_env* E, const double* actions, double* obs, double* rew, uint8_t* done, uint8_t* timeout,
                 uint8_t* success, int32_t* ep_len, double* ep_rew, double* goals, double* dr) {
  const int chunk = omp_chunk(E);
#pragma omp parallel for schedule(dynamic, chunk) num_threads(E->nthreads)
  for (int i = 0; i < E->N; i++)
    env_step(E, i, actions + (size_t)i * E->n, obs, rew, done, timeout, success, ep_len, ep_rew, goals, dr);
}
void oracle_get_observation(oracle_env* E, double* obs) {
  for (int i = 0; i < E->N; i++) calc_state(E, &E->st[i], obs + (size_t)i * E->O);
}
void oracle_increment_curriculum(oracle_env* E, double v) { if (E->cfg.task == SOLORL_TASK_POINTGOAL) E->goal_radius += v; }
void oracle_get_state(const oracle_env* E, int i, solorl_env_state* out) { *out = E->st[i]; }
void oracle_set_state(oracle_env* E, int i, const solorl_env_state* in) { E->st[i] = *in; }
void oracle_substep(oracle_env* E, int i) { substep(E, i); }
void oracle_mass_matrix(const oracle_env* E, int i, double* M, double* h) {
  kin_t K; kinematics(E, &E->st[i], &K); mass_and_bias(E, &E->st[i], &K, M, h);
}
void oracle_forward_dynamics(const oracle_env* E, int i, double* udot) {
  double M[NV_MAX * NV_MAX], h[NV_MAX], r[NV_MAX];
  oracle_mass_matrix(E, i, M, h);
  cholesky(M, E->nv);
  for (int k = 0; k < 6; k++) r[k] = -h[k];
  for (int j = 0; j < E->n; j++) r[6 + j] = E->st[i].tau[j] - h[6 + j];
  chol_solve(M, E->nv, r, udot);
}
void oracle_energy_momentum(const oracle_env* E, int i, double* out) {
  kin_t K; kinematics(E, &E->st[i], &K);
  double T = 0, V = 0, p[3] = {0, 0, 0}, Lm[3] = {0, 0, 0};
  for (int l = 0; l < E->nl; l++) {
    double m = E->md->links[l].mass, Iw[3], t[3];
    m3mulv(Iw, K.Iw[l], K.w[l]);
    T += 0.5 * m * v3dot(K.vc[l], K.vc[l]) + 0.5 * v3dot(K.w[l], Iw);
    V += m * E->cfg.gravity * K.c[l][2];
    v3axpy(p, m, K.vc[l]);
    v3cross(t, K.c[l], K.vc[l]); v3axpy(Lm, m, t); v3add(Lm, Lm, Iw);
  }
  out[0] = T; out[1] = V; memcpy(out + 2, p, sizeof p); memcpy(out + 5, Lm, sizeof Lm);
}
void oracle_prim_points(const oracle_env* E, int i, double* out) {
  kin_t K; kinematics(E, &E->st[i], &K);
  for (int p = 0; p < E->np; p++) { double P[3]; out[4 * p + 3] = prim_point(E, &K, p, P); memcpy(out + 4 * p, P, sizeof P); }
}
void oracle_last_lambda(const oracle_env* E, int i, double* lam) { memcpy(lam, E->last_lambda[i], sizeof(double) * E->np); }
