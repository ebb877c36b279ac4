// PPO-side kernels for the reference's MLP actor-critic (agents/ppo/policy.py:62-81 MLPBase: two separate tanh MLPs
// obs -> 64 -> 64, critic head 64 -> 1, actor head 64 -> A with a state-independent log-std, :138-148), gfx950.
//
// Why they exist: at 4096 envs a policy forward is 16 launches and a PPO mini-batch step ~100 launches of a few
// microseconds of work each; replayed from a HIP graph they still cost ~5 us apiece, which made the rollout 0.27 ms per
// step around a 0.17 ms env kernel and a mini-batch step 0.5 ms around ~50 us of arithmetic (profiles/r02_notes.md).
//
//   policy_act_mfma_kernel       value, sampled action and its log-prob for N observation rows      (policy.py:33-49 act)
//   ppo_grad_stage1_mfma_kernel  one mini-batch: gather, both MLPs forward, the clipped PPO losses (ppo.py:52-74) and the
//                                back-propagated pre-activation gradients of every layer, written [unit][row]
//   ppo_grad_stage2_mfma_kernel  the weight gradients G X^T of the six layers over row chunks
//   ppo_grad_stage3_kernel       their fixed-order sum into the parameters' gradients, log-std gradient, loss bookkeeping
//
// The three product kernels run on the matrix cores (exact-f32 MFMA), see "MFMA formulation" below.  Two VALU formulations
// (lane = row; weights through scalar loads, then through LDS broadcast reads) were built and measured first:
// profiles/r02_notes.md.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

#include "../../include/solorl.h"

extern "C" int solorl_fail_(int code, const char* msg);

namespace {

constexpr int H = 64;        // hidden units (the reference's default --hidden-size)
constexpr float HALF_LOG_2PI = 0.91893853320467274178f;

#define NET_ARGS const float* __restrict__ w0, const float* __restrict__ b0, const float* __restrict__ w1, const float* __restrict__ b1, \
                 const float* __restrict__ wh, const float* __restrict__ bh
#define CRITIC_ARGS const float* __restrict__ cw0, const float* __restrict__ cb0, const float* __restrict__ cw1, const float* __restrict__ cb1, \
                    const float* __restrict__ cwh, const float* __restrict__ cbh
#define ACTOR_ARGS const float* __restrict__ aw0, const float* __restrict__ ab0, const float* __restrict__ aw1, const float* __restrict__ ab1, \
                   const float* __restrict__ awh, const float* __restrict__ abh, const float* __restrict__ logstd
#define CRITIC_PASS P.critic_w0, P.critic_b0, P.critic_w1, P.critic_b1, P.critic_w2, P.critic_b2
#define ACTOR_PASS P.actor_w0, P.actor_b0, P.actor_w1, P.actor_b1, P.mean_w, P.mean_b, P.logstd

// tanh(x) = 1 - 2 / (1 + e^(2x)) on the hardware exp2 / reciprocal: 6 instructions instead of libm's ~40, absolute error
// <= 2e-7 over the whole line (saturates cleanly: e^(2x) = inf -> 1, 0 -> -1) -- the level of the f32 sums that feed it.  The
// relative error near 0 is larger (1e-4 at |x| = 1e-3), which an activation bounded by 1 does not care about.
__device__ __forceinline__ float tanh_fast(float x) {
  const float e = __expf(2.0f * x);
  return 1.0f - 2.0f * __frcp_rn(1.0f + e);
}

__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) x += __shfl_xor(x, s, 64);
  return x;
}

// ------------------------------------------------------------------------------------------------ stage 2 / 3 bookkeeping
// Stage 2 (MFMA, below) leaves per-chunk partial products in scratch; stage 3 adds them in a fixed order (reproducible, no
// atomics) straight into the parameters' gradients.  X carries a row of ones, so column K of a product is the bias gradient.
struct LayerDesc { const float* g; const float* x; float* wgrad; float* bgrad; int U, K1, off, tile0; };   // K1 = inputs + 1
struct Stage2Args { LayerDesc L[6]; int m, nchunks, total, ntiles; float* scratch; };

// stage 3: add the chunks (fixed order) into the parameters' gradients; the last block finishes the log-std gradient and the
// running loss sums from stage 1's per-wavefront partials
struct Stage3Args { Stage2Args S; const float* partials; int nwaves, A; const float* logstd; float* logstd_grad; float* loss_sums;
                    float* logstd_sum; float entropy_coef; };
__global__ void __launch_bounds__(256) ppo_grad_stage3_kernel(const Stage3Args T) {
  const Stage2Args& S = T.S;
  if (blockIdx.x == gridDim.x - 1) {          // 3 + A column sums of partials [nwaves][3 + A], one wavefront each column group
    const int nc = 3 + T.A, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int c = wv; c < nc; c += 4) {
      float a = 0.f;
      for (int w = lane; w < T.nwaves; w += 64) a += T.partials[(size_t)w * nc + c];
      a = wave_sum(a);
      if (lane == 0) {
        T.loss_sums[c] += a;
        if (c >= 3) T.logstd_grad[c - 3] = a - T.entropy_coef / (float)T.A;      // entropy = mean over batch and dims of logstd + const
      }
    }
    if (threadIdx.x == 0) { float e = 0.f; for (int a = 0; a < T.A; ++a) e += T.logstd[a]; *T.logstd_sum += e; }
    return;
  }
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= S.total) return;
  int li = 0;
#pragma unroll
  for (int i = 1; i < 6; ++i) if (e >= S.L[i].off) li = i;
  const LayerDesc& L = S.L[li];
  float a4[4] = {0.f, 0.f, 0.f, 0.f};                       // four interleaved sums: the loads of a group are independent
  int c = 0;
  for (; c + 4 <= S.nchunks; c += 4) {
#pragma unroll
    for (int i = 0; i < 4; ++i) a4[i] += S.scratch[(size_t)(c + i) * S.total + e];
  }
  for (; c < S.nchunks; ++c) a4[0] += S.scratch[(size_t)c * S.total + e];
  const float a = (a4[0] + a4[1]) + (a4[2] + a4[3]);
  const int q = e - L.off, u = q / L.K1, k = q % L.K1;
  if (k == L.K1 - 1) L.bgrad[u] = a; else L.wgrad[u * (L.K1 - 1) + k] = a;
}

// ================================================================================================ MFMA formulation
// act and stage 1 on the matrix cores (v_mfma_f32_32x32x2_f32: exact f32, D[32x32] += A[32x2] B[2x32], 64 cycles).
// Orientation: M = units, N = rows, i.e. every product is W . X^T.  Lane l = (c = l & 31, h = l >> 5):
//   A operand  lane holds A[i = c][k = h]      -> a WEIGHT element W[unit 32 mt + c][input ...]
//   B operand  lane holds B[k = h][j = c]      -> an ACTIVATION element of row c of the 32-row tile
//   D          lane holds D[i = 8 (v >> 2) + 4 h + (v & 3)][j = c] in register v = 0..15
// so a lane's 16 results belong to ITS row c and to the units {8 q + 4 h + t}.  A step of the NEXT layer contracts over two
// units, one from each lane half; taking them as (8 q + t) from half 0 and (8 q + 4 + t) from half 1 makes the next layer's B
// operand for step (q, t) exactly register v = 4 q' + t of the previous D (q' = q mod 4, tile mt = q / 4): activations never
// leave their registers between layers, forward or backward, and there is no LDS in these kernels.  The matching A element
// is W[unit 32 mt + c][input 8 q + 4 h + t]: a lane reads 16 consecutive bytes of one weight row per q -- all weights of a net
// (forward and transposed, 256 VGPRs) are loaded once per wavefront and stay in registers while it walks over its row tiles.
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

template <int O, int NOUT, bool BACKWARD> struct NetRegs {
  static constexpr int QO = (O + 7) / 8, QA = (NOUT + 7) / 8;
  float a0[2][QO][4], a1[2][8][4], ah[8][4];
  float aht[BACKWARD ? 2 : 1][QA][4], a1t[BACKWARD ? 2 : 1][BACKWARD ? 8 : 1][4];
  __device__ __forceinline__ void load(NET_ARGS, int c, int h) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
      for (int q = 0; q < QO; ++q) {
        const int i0 = 8 * q + 4 * h;
        const float* wr = w0 + (size_t)(32 * mt + c) * O + i0;
        if constexpr (O % 4 == 0) {                             // a 4-input chunk is wholly inside or outside the row
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (i0 < O) v = *reinterpret_cast<const float4*>(wr);
          a0[mt][q][0] = v.x; a0[mt][q][1] = v.y; a0[mt][q][2] = v.z; a0[mt][q][3] = v.w;
        } else {
#pragma unroll
          for (int t = 0; t < 4; ++t) a0[mt][q][t] = i0 + t < O ? wr[t] : 0.f;
        }
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(w1 + (32 * mt + c) * H + 8 * q + 4 * h);
        a1[mt][q][0] = v.x; a1[mt][q][1] = v.y; a1[mt][q][2] = v.z; a1[mt][q][3] = v.w;
      }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c < NOUT) v = *reinterpret_cast<const float4*>(wh + c * H + 8 * q + 4 * h);
      ah[q][0] = v.x; ah[q][1] = v.y; ah[q][2] = v.z; ah[q][3] = v.w;
    }
    if constexpr (BACKWARD) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
        for (int q = 0; q < QA; ++q)
#pragma unroll
          for (int t = 0; t < 4; ++t) { const int a = 8 * q + 4 * h + t; aht[mt][q][t] = a < NOUT ? wh[a * H + 32 * mt + c] : 0.f; }
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
          for (int t = 0; t < 4; ++t) a1t[mt][q][t] = w1[(8 * q + 4 * h + t) * H + 32 * mt + c];
      }
    }
  }
};

// the unit a D register belongs to
__device__ __forceinline__ constexpr int unit_of(int mt, int v, int h) { return 32 * mt + 8 * (v >> 2) + 4 * h + (v & 3); }

// layers 1 and 2 and the head for one 32-row tile; x = the lane's row, inputs 8 q + 4 h .. + 3 per q.  After the call h1, h2 hold
// tanh activations (D layout), dh the head outputs (unit = unit_of(0, v, h), valid below NOUT).
template <int O, int NOUT, bool BW, typename S1, typename S2>
__device__ __forceinline__ void mfma_forward(const NetRegs<O, NOUT, BW>& R, const float* __restrict__ b0, const float* __restrict__ b1,
                                             const float* __restrict__ bh, const float* __restrict__ xrow, int h, f32x16 (&h1)[2],
                                             f32x16 (&h2)[2], f32x16& dh, S1&& store_x, S2&& store_h) {
  constexpr int QO = NetRegs<O, NOUT, BW>::QO;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int v = 0; v < 16; ++v) { h1[mt][v] = b0[unit_of(mt, v, h)]; h2[mt][v] = b1[unit_of(mt, v, h)]; }
#pragma unroll
  for (int v = 0; v < 16; ++v) { const int a = unit_of(0, v, h); dh[v] = a < NOUT ? bh[a] : 0.f; }
#pragma unroll
  for (int q = 0; q < QO; ++q) {
    const int i0 = 8 * q + 4 * h;
    float x[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (O % 4 == 0) {
      if (i0 < O) { const float4 xv = *reinterpret_cast<const float4*>(xrow + i0); x[0] = xv.x; x[1] = xv.y; x[2] = xv.z; x[3] = xv.w; }
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t) if (i0 + t < O) x[t] = xrow[i0 + t];
    }
    if (i0 < O) store_x(i0, x);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      h1[0] = MFMA32(R.a0[0][q][t], x[t], h1[0]);
      h1[1] = MFMA32(R.a0[1][q][t], x[t], h1[1]);
    }
  }
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int v = 0; v < 16; ++v) h1[mt][v] = tanh_fast(h1[mt][v]);
  store_h(1, h1);
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      h2[0] = MFMA32(R.a1[0][4 * mt + (v >> 2)][v & 3], h1[mt][v], h2[0]);
      h2[1] = MFMA32(R.a1[1][4 * mt + (v >> 2)][v & 3], h1[mt][v], h2[1]);
    }
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int v = 0; v < 16; ++v) h2[mt][v] = tanh_fast(h2[mt][v]);
  store_h(2, h2);
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int v = 0; v < 16; ++v) dh = MFMA32(R.ah[4 * mt + (v >> 2)][v & 3], h2[mt][v], dh);
}

// ---------------------------------------------------------------- act on MFMA: grid (ceil(n / 32), 2), one wavefront per 32 rows
template <int O, int A, bool ACTOR>
__device__ __forceinline__ void act_net_mfma(NET_ARGS, const float* __restrict__ logstd, const float* __restrict__ obs,
                                             const float* __restrict__ noise, int n, float* value_out, float* action_out, float* logp_out) {
  constexpr int NOUT = ACTOR ? A : 1;
  const int l = threadIdx.x, c = l & 31, h = l >> 5, row = blockIdx.x * 32 + c;
  const bool on = row < n;
  const int r = on ? row : n - 1;
  NetRegs<O, NOUT, false> R;
  R.load(w0, b0, w1, b1, wh, bh, c, h);
  f32x16 h1[2], h2[2], dh;
  mfma_forward<O, NOUT, false>(R, b0, b1, bh, obs + (size_t)r * O, h, h1, h2, dh, [](int, const float (&)[4]) {}, [](int, f32x16 (&)[2]) {});
  if constexpr (!ACTOR) {
    if (on && h == 0) value_out[row] = dh[0];
  } else {
    float lp = 0.f;
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int a = unit_of(0, v, h);
      if (a < A && (8 * (v >> 2) + (v & 3)) < A) {              // (second test: compile-time bound on v; first: this half's unit)
        const float mean = dh[v], ls = logstd[a];
        const float act = noise ? fmaf(expf(ls), noise[(size_t)r * A + a], mean) : mean;       // policy.py:40-43
        const float z = (act - mean) * expf(-ls);                                              // ModNormal.log_probs, policy.py:171-173
        lp += -0.5f * z * z - ls - HALF_LOG_2PI;
        if (on) action_out[(size_t)row * A + a] = act;
      }
    }
    lp += __shfl_xor(lp, 32, 64);                                // the row's other action dims live in the other lane half
    if (on && h == 0) logp_out[row] = lp;
  }
}

template <int O, int A>
__global__ void __launch_bounds__(64)
policy_act_mfma_kernel(CRITIC_ARGS, ACTOR_ARGS, const float* __restrict__ obs, const float* __restrict__ noise, int n, float* value_out,
                       float* action_out, float* logp_out) {
  if (blockIdx.y == 0) act_net_mfma<O, A, false>(cw0, cb0, cw1, cb1, cwh, cbh, nullptr, obs, noise, n, value_out, action_out, logp_out);
  else act_net_mfma<O, A, true>(aw0, ab0, aw1, ab1, awh, abh, logstd, obs, noise, n, value_out, action_out, logp_out);
}

// ---------------------------------------------------------------- stage 1 on MFMA: grid (ceil(m / 64), 2), one wavefront walks two
// 32-row tiles (= rows 64 b .. 64 b + 63, so partials keep one row per 64 samples) with its net's weights resident
template <int O, int A, bool ACTOR>
__device__ __forceinline__ void grad_net_mfma(NET_ARGS, const float* __restrict__ logstd, const solorl_ppo_batch& B, float* xt0, float* xt1,
                                              float* xt2, float* g1, float* g2, float* gh, float* partials) {
  constexpr int NOUT = ACTOR ? A : 1;
  using NR = NetRegs<O, NOUT, true>;
  const int l = threadIdx.x, c = l & 31, h = l >> 5, m = B.m;
  NR R;
  R.load(w0, b0, w1, b1, wh, bh, c, h);
  const float inv_m = 1.0f / (float)m;
  float loss_acc = 0.f, gls_acc[ACTOR ? 16 : 1] = {};
  int rows_here = 0;
#pragma unroll 1
  for (int tile = 0; tile < 2; ++tile) {
    const int row = blockIdx.x * 64 + tile * 32 + c;
    if (blockIdx.x * 64 + tile * 32 >= m) break;
    const bool on = row < m;
    const int r = on ? row : m - 1;
    rows_here += min(m - (blockIdx.x * 64 + tile * 32), 32);
    const long long s = B.perm[*B.offset + r];
    f32x16 h1[2], h2[2], dh;
    auto store_units = [&](float* arr, f32x16 (&a)[2]) {
      if (on) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int v = 0; v < 16; ++v) arr[(size_t)unit_of(mt, v, h) * m + row] = a[mt][v];
      }
    };
    mfma_forward<O, NOUT, true>(R, b0, b1, bh, B.obs + (size_t)s * O, h, h1, h2, dh,
        [&](int i0, const float (&x)[4]) {
          if (!ACTOR && on) {
#pragma unroll
            for (int t = 0; t < 4; ++t) if (i0 + t < O) xt0[(size_t)(i0 + t) * m + row] = x[t];
          }
        },
        [&](int layer, f32x16 (&a)[2]) { store_units(layer == 1 ? xt1 : xt2, a); });
    // head gradient in the head's D layout: register v <-> unit 8 (v >> 2) + 4 h + (v & 3)
    f32x16 gout;
#pragma unroll
    for (int v = 0; v < 16; ++v) gout[v] = 0.f;
    if constexpr (!ACTOR) {
      const float v_ = dh[0], ret = B.ret[s], vp = B.vpred[s];
      const float u = v_ - ret;
      float vl, gv;
      if (B.clipped_value) {                                       // ppo.py:61-66
        const float dvp = v_ - vp;
        const float wv_ = vp + fminf(fmaxf(dvp, -B.clip), B.clip) - ret;
        const float ins = (dvp >= -B.clip && dvp <= B.clip) ? 1.0f : 0.0f;
        const float uu = u * u, ww = wv_ * wv_;
        vl = 0.5f * fmaxf(uu, ww);
        gv = uu > ww ? u : (uu < ww ? wv_ * ins : 0.5f * (u + wv_ * ins));
      } else { vl = 0.5f * u * u; gv = u; }                        // ppo.py:67-68
      const bool mine = on && h == 0;                              // unit 0 lives in lane half 0
      gout[0] = mine ? B.value_coef * gv * inv_m : 0.f;
      if (mine) { gh[row] = gout[0]; loss_acc += vl; }
    } else {
      float z[16], e[16], lp = 0.f;
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int a = unit_of(0, v, h);
        z[v] = 0.f; e[v] = 0.f;
        if ((8 * (v >> 2) + (v & 3)) < A && a < A) {
          const float ls = logstd[a];
          e[v] = expf(-ls);
          z[v] = (B.actions[(size_t)s * A + a] - dh[v]) * e[v];
          lp += -0.5f * z[v] * z[v] - ls - HALF_LOG_2PI;
        }
      }
      lp += __shfl_xor(lp, 32, 64);
      const float ratio = expf(lp - B.old_logp[s]), adv = B.adv[s];                                  // ppo.py:54-59
      const float s1 = ratio * adv, s2 = fminf(fmaxf(ratio, 1.0f - B.clip), 1.0f + B.clip) * adv;
      const float inside = (ratio >= 1.0f - B.clip && ratio <= 1.0f + B.clip) ? 1.0f : 0.0f;
      const float wsel = s1 < s2 ? 1.0f : (s1 > s2 ? inside : 0.5f * (1.0f + inside));
      const float glp = on ? -adv * ratio * wsel * inv_m : 0.f;
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int a = unit_of(0, v, h);
        if ((8 * (v >> 2) + (v & 3)) < A && a < A) {
          gout[v] = glp * z[v] * e[v];                               // d logp / d mean_a = z / sigma
          if (on) gh[(size_t)a * m + row] = gout[v];
          gls_acc[v] += glp * (z[v] * z[v] - 1.0f);                  // d logp / d logstd_a = z^2 - 1
        }
      }
      if (on && h == 0) loss_acc += -fminf(s1, s2);
    }
    // back through the head, layer 2, layer 1: the same register-to-operand identity as forward
    f32x16 d2[2], d1[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int v = 0; v < 16; ++v) { d2[mt][v] = 0.f; d1[mt][v] = 0.f; }
#pragma unroll
    for (int v = 0; v < 4 * NR::QA; ++v) {
      d2[0] = MFMA32(R.aht[0][v >> 2][v & 3], gout[v], d2[0]);
      d2[1] = MFMA32(R.aht[1][v >> 2][v & 3], gout[v], d2[1]);
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int v = 0; v < 16; ++v) d2[mt][v] *= 1.0f - h2[mt][v] * h2[mt][v];
    store_units(g2, d2);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        d1[0] = MFMA32(R.a1t[0][4 * mt + (v >> 2)][v & 3], d2[mt][v], d1[0]);
        d1[1] = MFMA32(R.a1t[1][4 * mt + (v >> 2)][v & 3], d2[mt][v], d1[1]);
      }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int v = 0; v < 16; ++v) d1[mt][v] *= 1.0f - h1[mt][v] * h1[mt][v];
    store_units(g1, d1);
  }
  // this wavefront's row of partials: sums over its 64 rows
  float* P_ = partials + (size_t)blockIdx.x * (3 + A);
  const float tot = wave_sum(loss_acc);
  if constexpr (!ACTOR) {
    if (l == 0) P_[0] = tot;
  } else {
    if (l == 0) { P_[1] = tot; P_[2] = (float)rows_here; }
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      if ((8 * (v >> 2) + (v & 3)) < A) {
        float sg = gls_acc[v];                                      // sum over the 32 lanes of this half (the unit differs per half)
#pragma unroll
        for (int d = 16; d > 0; d >>= 1) sg += __shfl_xor(sg, d, 64);
        const int a = unit_of(0, v, h);
        if (c == 0 && a < A) P_[3 + a] = sg;
      }
    }
  }
}

template <int O, int A>
__global__ void __launch_bounds__(64)
ppo_grad_stage1_mfma_kernel(CRITIC_ARGS, ACTOR_ARGS, const solorl_ppo_batch B, const solorl_ppo_stage1 W) {
  if (blockIdx.y == 0) grad_net_mfma<O, A, false>(cw0, cb0, cw1, cb1, cwh, cbh, nullptr, B, W.xt0, W.c_xt1, W.c_xt2, W.c_g1, W.c_g2, W.c_gh, W.partials);
  else grad_net_mfma<O, A, true>(aw0, ab0, aw1, ab1, awh, abh, logstd, B, W.xt0, W.a_xt1, W.a_xt2, W.a_g1, W.a_g2, W.a_gh, W.partials);
}

// ---------------------------------------------------------------- stage 2 on MFMA: d W = G . X^T with the ROW index as K
// One wavefront per (layer, 32-unit tile of G, 32-unit tile of X, chunk of MCHUNK rows).  Lane (c, h) reads 16 consecutive bytes
// = rows r .. r + 3 of ITS unit's row of G (A operand) and of X (B operand), half h taking rows r0 + 4 h ..: four MFMA steps per
// pair of loads, each step contracting one row from either half.  D register v of lane (c, h) is
// d W[unit 32 mt + 8 (v >> 2) + 4 h + (v & 3)][input 32 nt + c]: stores are contiguous along the input index.
constexpr int MCHUNK = 512;
__global__ void __launch_bounds__(64) ppo_grad_stage2_mfma_kernel(const Stage2Args S) {
  int li = 0;
#pragma unroll
  for (int i = 1; i < 6; ++i) if ((int)blockIdx.x >= S.L[i].tile0) li = i;
  const LayerDesc& L = S.L[li];
  const int ntn = (L.K1 + 31) / 32, ntm = (L.U + 31) / 32;
  const int t = blockIdx.x - L.tile0, chunk = t / (ntm * ntn), tt = t % (ntm * ntn), mt = tt / ntn, nt = tt % ntn;
  const int l = threadIdx.x, c = l & 31, h = l >> 5, m = S.m;
  const int ua = 32 * mt + c, ub = 32 * nt + c;
  const bool va = ua < L.U, vb = ub < L.K1;
  const float* __restrict__ ga = L.g + (size_t)(va ? ua : 0) * m;
  const float* __restrict__ xb = L.x + (size_t)(vb ? ub : 0) * m;
  const int r0 = chunk * MCHUNK, r1 = min(r0 + MCHUNK, m);
  f32x16 acc;
#pragma unroll
  for (int v = 0; v < 16; ++v) acc[v] = 0.f;
  // (m is a multiple of 64: whole float4s, both halves always in range and on the same trip count.)  The loads of the next two
  // steps are in flight while a step's four MFMAs run: a lone wavefront has nobody else to hide the L2 latency behind
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  auto lda = [&](int r) { return (va && r < r1) ? *reinterpret_cast<const float4*>(ga + r) : zero4; };
  auto ldb = [&](int r) { return (vb && r < r1) ? *reinterpret_cast<const float4*>(xb + r) : zero4; };
  int r = r0 + 4 * h;
  float4 a0 = lda(r), b0 = ldb(r), a1 = lda(r + 8), b1 = ldb(r + 8);
#pragma unroll 1
  for (; r < r1; r += 16) {
    const float4 a2 = lda(r + 16), b2 = ldb(r + 16), a3 = lda(r + 24), b3 = ldb(r + 24);
    acc = MFMA32(a0.x, b0.x, acc); acc = MFMA32(a0.y, b0.y, acc); acc = MFMA32(a0.z, b0.z, acc); acc = MFMA32(a0.w, b0.w, acc);
    acc = MFMA32(a1.x, b1.x, acc); acc = MFMA32(a1.y, b1.y, acc); acc = MFMA32(a1.z, b1.z, acc); acc = MFMA32(a1.w, b1.w, acc);
    a0 = a2; b0 = b2; a1 = a3; b1 = b3;
  }
  float* out = S.scratch + (size_t)chunk * S.total + L.off;
#pragma unroll
  for (int v = 0; v < 16; ++v) {
    const int u = 32 * mt + 8 * (v >> 2) + 4 * h + (v & 3);
    if (u < L.U && vb) out[u * L.K1 + ub] = acc[v];
  }
}

// ---------------------------------------------------------------- norm clip + Adam over all 13 parameter tensors, one workgroup
// nn.utils.clip_grad_norm_ (agents/ppo/ppo.py:75-76) and torch.optim.Adam's update (:32, :77; no amsgrad) for ~20 k parameters are
// nine small launches in PyTorch (norm, three scalar ops, scale, the multi-tensor Adam, the mini-batch cursor ...); here one
// 1024-thread workgroup reads every gradient twice.
struct AdamSeg { float* p; const float* g; int n, off; };
struct AdamArgs { AdamSeg seg[13]; int total; float* m; float* v; float* step; const float* lr; float b1, b2, eps, wd, max_norm, gscale;
                  long long* offset; long long inc; };
__global__ void __launch_bounds__(1024) ppo_clip_adam_kernel(const AdamArgs K) {
  // (a variant with one flat element per (thread, slot) and all loads of a pass issued together was slower, 31 vs 22 us: finding
  // an element's tensor means indexing the argument struct dynamically, which the compiler serves from scratch)
  __shared__ float red[16];
  __shared__ float coef_s;
  const int tid = threadIdx.x;
  float ss = 0.f;
#pragma unroll 1
  for (int s = 0; s < 13; ++s)
    for (int i = tid; i < K.seg[s].n; i += 1024) { const float g = K.seg[s].g[i] * K.gscale; ss = fmaf(g, g, ss); }
  ss = wave_sum(ss);
  if ((tid & 63) == 0) red[tid >> 6] = ss;
  __syncthreads();
  if (tid == 0) {
    float t = 0.f;
    for (int w = 0; w < 16; ++w) t += red[w];
    coef_s = K.max_norm > 0.f ? fminf(1.0f, K.max_norm / (sqrtf(t) + 1e-6f)) : 1.0f;      // clip_coef clamped to 1
  }
  __syncthreads();
  const float coef = coef_s, step = *K.step + 1.0f, lr = *K.lr;
  const float bc1 = 1.0f - powf(K.b1, step), bc2 = 1.0f - powf(K.b2, step);
  const float step_size = lr / bc1, inv_sqrt_bc2 = 1.0f / sqrtf(bc2);
#pragma unroll 1
  for (int s = 0; s < 13; ++s) {
    const AdamSeg& S = K.seg[s];
    for (int i = tid; i < S.n; i += 1024) {
      float p = S.p[i], g = S.g[i] * K.gscale * coef;        // gscale: 1 / world after a SUMMED gradient all-reduce (the mean's scale, fused)
      if (K.wd != 0.f) g = fmaf(K.wd, p, g);
      const int j = S.off + i;
      const float m = K.m[j] + (1.0f - K.b1) * (g - K.m[j]);                 // exp_avg.lerp_(grad, 1 - beta1)
      const float v = K.b2 * K.v[j] + (1.0f - K.b2) * g * g;
      K.m[j] = m; K.v[j] = v;
      S.p[i] = p - step_size * (m / (sqrtf(v) * inv_sqrt_bc2 + K.eps));
    }
  }
  __syncthreads();                                                            // everyone has read *K.step
  if (tid == 0) { *K.step = step; if (K.offset) *K.offset += K.inc; }
}

// observation widths of zero or one history level (14 + 2 n (+ 4 pointGoal), x 1 or x 2), action widths n = 8 / 12
template <typename F> int dispatch_dims(int O, int A, F&& f) {
#define SOLO_DIMS(O_, A_) if (O == O_ && A == A_) return f(std::integral_constant<int, O_>(), std::integral_constant<int, A_>());
  SOLO_DIMS(76, 12) SOLO_DIMS(84, 12) SOLO_DIMS(38, 12) SOLO_DIMS(42, 12)
  SOLO_DIMS(60, 8) SOLO_DIMS(68, 8) SOLO_DIMS(30, 8) SOLO_DIMS(34, 8)
#undef SOLO_DIMS
  return solorl_fail_(SOLORL_ERR_INVALID, "policy kernels are built for the observation / action sizes of zero or one history level "
                                          "(38, 42, 76 or 84 x 12; 30, 34, 60 or 68 x 8); use the PyTorch path for other shapes");
}

int check_policy(const solorl_policy_params* p, int device_id) {
  if (!p) return solorl_fail_(SOLORL_ERR_INVALID, "null policy parameters");
  if (p->hidden != H) return solorl_fail_(SOLORL_ERR_INVALID, "policy kernels are built for hidden size 64");
  const void* ptrs[] = {p->critic_w0, p->critic_b0, p->critic_w1, p->critic_b1, p->critic_w2, p->critic_b2, p->actor_w0, p->actor_b0,
                        p->actor_w1, p->actor_b1, p->mean_w, p->mean_b, p->logstd};
  for (const void* q : ptrs) if (!q) return solorl_fail_(SOLORL_ERR_INVALID, "null policy parameter pointer");
  // the kernels read weight rows as float4 (16 consecutive bytes per lane): every parameter tensor must start on a 16-byte
  // boundary (torch allocations do; an offset view passed by another C-ABI caller would fault on the GPU instead)
  const void* vec[] = {p->critic_w1, p->actor_w1, p->critic_w2, p->mean_w, p->obs_dim % 4 == 0 ? p->critic_w0 : nullptr, p->obs_dim % 4 == 0 ? p->actor_w0 : nullptr};
  for (const void* q : vec) if (reinterpret_cast<uintptr_t>(q) & 15u) return solorl_fail_(SOLORL_ERR_INVALID, "policy weight matrices must be 16-byte aligned (rows are read as float4)");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return solorl_fail_(SOLORL_ERR_NODEVICE, "no HIP device available (no CPU fallback)");
  if (device_id < 0 || device_id >= ndev) return solorl_fail_(SOLORL_ERR_NODEVICE, "device_id out of range");
  if (hipSetDevice(device_id) != hipSuccess) return solorl_fail_(SOLORL_ERR_HIP, "hipSetDevice");
  return 0;
}

}  // namespace

extern "C" {

int solorl_policy_act(const solorl_policy_params* p, const float* obs, const float* noise, int n, float* value_out, float* action_out,
                      float* logp_out, int device_id, void* stream) {
  if (int rc = check_policy(p, device_id)) return rc;
  if (!obs || !value_out || !action_out || !logp_out || n < 1) return solorl_fail_(SOLORL_ERR_INVALID, "solorl_policy_act: null array or n < 1");
  if (p->obs_dim % 4 == 0 && (reinterpret_cast<uintptr_t>(obs) & 15u)) return solorl_fail_(SOLORL_ERR_INVALID, "solorl_policy_act: obs must be 16-byte aligned (rows are read as float4)");
  const solorl_policy_params P = *p;
  return dispatch_dims(P.obs_dim, P.act_dim, [&](auto oc, auto ac) {
    constexpr int O = decltype(oc)::value, A = decltype(ac)::value;
    hipLaunchKernelGGL((policy_act_mfma_kernel<O, A>), dim3((n + 31) / 32, 2), dim3(64), 0, (hipStream_t)stream, CRITIC_PASS, ACTOR_PASS, obs, noise,
                       n, value_out, action_out, logp_out);
    return hipGetLastError() == hipSuccess ? 0 : solorl_fail_(SOLORL_ERR_HIP, "policy_act_mfma_kernel launch");
  });
}

int solorl_ppo_grad_stage1(const solorl_policy_params* p, const solorl_ppo_batch* batch, const solorl_ppo_stage1* work, int device_id,
                           void* stream) {
  if (int rc = check_policy(p, device_id)) return rc;
  if (!batch || !work || batch->m < 1) return solorl_fail_(SOLORL_ERR_INVALID, "solorl_ppo_grad_stage1: null argument or m < 1");
  const void* ptrs[] = {batch->obs, batch->actions, batch->old_logp, batch->adv, batch->vpred, batch->ret, batch->perm, batch->offset,
                        work->xt0, work->c_xt1, work->c_xt2, work->c_g1, work->c_g2, work->c_gh, work->a_xt1, work->a_xt2, work->a_g1,
                        work->a_g2, work->a_gh, work->partials};
  for (const void* q : ptrs) if (!q) return solorl_fail_(SOLORL_ERR_INVALID, "solorl_ppo_grad_stage1: null array");
  if (p->obs_dim % 4 == 0 && (reinterpret_cast<uintptr_t>(batch->obs) & 15u)) return solorl_fail_(SOLORL_ERR_INVALID, "solorl_ppo_grad_stage1: obs must be 16-byte aligned (rows are read as float4)");
  const solorl_policy_params P = *p;
  const solorl_ppo_batch B = *batch;
  const solorl_ppo_stage1 W = *work;
  return dispatch_dims(P.obs_dim, P.act_dim, [&](auto oc, auto ac) {
    constexpr int O = decltype(oc)::value, A = decltype(ac)::value;
    hipLaunchKernelGGL((ppo_grad_stage1_mfma_kernel<O, A>), dim3((B.m + 63) / 64, 2), dim3(64), 0, (hipStream_t)stream, CRITIC_PASS, ACTOR_PASS, B, W);
    return hipGetLastError() == hipSuccess ? 0 : solorl_fail_(SOLORL_ERR_HIP, "ppo_grad_stage1_mfma_kernel launch");
  });
}

int solorl_ppo_grad_count(int obs_dim, int act_dim) { return 2 * H * (obs_dim + 1) + 2 * H * (H + 1) + (H + 1) + act_dim * (H + 1); }

int solorl_ppo_grad_stage2(const solorl_policy_params* p, const solorl_ppo_stage1* work, int m, const solorl_ppo_grads* out, int device_id,
                           void* stream) {
  if (int rc = check_policy(p, device_id)) return rc;
  if (!work || !out || m < 1) return solorl_fail_(SOLORL_ERR_INVALID, "solorl_ppo_grad_stage2: null argument or m < 1");
  const void* ptrs[] = {out->critic_w0, out->critic_b0, out->critic_w1, out->critic_b1, out->critic_w2, out->critic_b2, out->actor_w0,
                        out->actor_b0, out->actor_w1, out->actor_b1, out->mean_w, out->mean_b, out->logstd, out->loss_sums, out->logstd_sum,
                        out->scratch, work->xt0, work->partials};
  for (const void* q : ptrs) if (!q) return solorl_fail_(SOLORL_ERR_INVALID, "solorl_ppo_grad_stage2: null array");
  const int O = p->obs_dim, A = p->act_dim;
  Stage3Args T;
  Stage2Args& S = T.S;
  const LayerDesc L[6] = {
      {work->c_g1, work->xt0, out->critic_w0, out->critic_b0, H, O + 1, 0, 0}, {work->c_g2, work->c_xt1, out->critic_w1, out->critic_b1, H, H + 1, 0, 0},
      {work->c_gh, work->c_xt2, out->critic_w2, out->critic_b2, 1, H + 1, 0, 0}, {work->a_g1, work->xt0, out->actor_w0, out->actor_b0, H, O + 1, 0, 0},
      {work->a_g2, work->a_xt1, out->actor_w1, out->actor_b1, H, H + 1, 0, 0}, {work->a_gh, work->a_xt2, out->mean_w, out->mean_b, A, H + 1, 0, 0}};
  if (m % 64 != 0) return solorl_fail_(SOLORL_ERR_INVALID, "solorl_ppo_grad_stage2: m must be a multiple of 64 rows");
  S.m = m; S.nchunks = (m + MCHUNK - 1) / MCHUNK; S.scratch = out->scratch;
  int off = 0, tile = 0;
  for (int i = 0; i < 6; ++i) {
    S.L[i] = L[i]; S.L[i].off = off; S.L[i].tile0 = tile;
    off += L[i].U * L[i].K1;
    tile += ((L[i].U + 31) / 32) * ((L[i].K1 + 31) / 32) * S.nchunks;
  }
  S.total = off; S.ntiles = tile;
  T.partials = work->partials; T.nwaves = (m + 63) / 64; T.A = A; T.logstd = p->logstd; T.logstd_grad = out->logstd;
  T.loss_sums = out->loss_sums; T.logstd_sum = out->logstd_sum; T.entropy_coef = out->entropy_coef;
  hipLaunchKernelGGL(ppo_grad_stage2_mfma_kernel, dim3(S.ntiles), dim3(64), 0, (hipStream_t)stream, S);
  hipLaunchKernelGGL(ppo_grad_stage3_kernel, dim3((S.total + 255) / 256 + 1), dim3(256), 0, (hipStream_t)stream, T);
  return hipGetLastError() == hipSuccess ? 0 : solorl_fail_(SOLORL_ERR_HIP, "ppo_grad_stage2/3 launch");
}

int solorl_ppo_clip_adam(const solorl_policy_params* p, const solorl_ppo_grads* g, const solorl_adam_state* a, int device_id, void* stream) {
  if (int rc = check_policy(p, device_id)) return rc;
  if (!g || !a || !a->exp_avg || !a->exp_avg_sq || !a->step || !a->lr) return solorl_fail_(SOLORL_ERR_INVALID, "solorl_ppo_clip_adam: null argument");
  const int O = p->obs_dim, A = p->act_dim;
  const float* ps[13] = {p->critic_w0, p->critic_b0, p->critic_w1, p->critic_b1, p->critic_w2, p->critic_b2, p->actor_w0, p->actor_b0,
                         p->actor_w1, p->actor_b1, p->mean_w, p->mean_b, p->logstd};
  const float* gs[13] = {g->critic_w0, g->critic_b0, g->critic_w1, g->critic_b1, g->critic_w2, g->critic_b2, g->actor_w0, g->actor_b0,
                         g->actor_w1, g->actor_b1, g->mean_w, g->mean_b, g->logstd};
  const int ns[13] = {H * O, H, H * H, H, H, 1, H * O, H, H * H, H, A * H, A, A};
  AdamArgs K;
  int off = 0;
  for (int i = 0; i < 13; ++i) {
    if (!gs[i]) return solorl_fail_(SOLORL_ERR_INVALID, "solorl_ppo_clip_adam: null gradient pointer");
    K.seg[i] = {const_cast<float*>(ps[i]), gs[i], ns[i], off};
    off += ns[i];
  }
  K.total = off; K.m = a->exp_avg; K.v = a->exp_avg_sq; K.step = a->step; K.lr = a->lr;
  K.b1 = a->beta1; K.b2 = a->beta2; K.eps = a->eps; K.wd = a->weight_decay; K.max_norm = a->max_grad_norm;
  K.gscale = a->grad_scale > 0.f ? a->grad_scale : 1.0f;
  K.offset = reinterpret_cast<long long*>(a->offset); K.inc = a->offset_increment;
  hipLaunchKernelGGL(ppo_clip_adam_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, K);
  return hipGetLastError() == hipSuccess ? 0 : solorl_fail_(SOLORL_ERR_HIP, "ppo_clip_adam_kernel launch");
}

}  // extern "C"
