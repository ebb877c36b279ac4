// PPO-side kernels for the reference's MLP actor-critic (agents/ppo/policy.py:62-81 MLPBase: two separate tanh MLPs
// obs -> 64 -> 64, critic head 64 -> 1, actor head 64 -> A with a state-independent log-std, :138-148), gfx950.
//
// Why they exist: at 4096 envs a policy forward is 16 launches and a PPO mini-batch step ~100 launches of a few
// microseconds of work each; replayed from a HIP graph they still cost ~5 us apiece, which made the rollout 0.27 ms per
// step around a 0.17 ms env kernel and a mini-batch step 0.5 ms around ~50 us of arithmetic (profiles/r02_notes.md).
//
//   policy_act_kernel        value, sampled action and its log-prob for N observation rows          (policy.py:33-49 act)
//   ppo_grad_stage1_kernel   one mini-batch: gather, both MLPs forward, the clipped PPO losses (ppo.py:52-74) and the
//                            back-propagated pre-activation gradients of every layer, written [unit][row] so that the
//                            weight gradients are plain (row-sliced) GEMMs G^T X for the caller
//
// Mapping: one lane = one row (sample).  The weights are then wave-uniform.  A first version fetched them with scalar loads
// (every multiply-add a single v_fma with an SGPR operand) and was latency-bound on exactly those loads: ~100 SGPRs hold one
// 16-dword chunk per accumulator plus one in flight, 1024 wavefronts stream 39 KB of weights each through the small scalar
// cache, and a mini-batch took 141 us (21 cycles per multiply-add).  Now a workgroup stages its net's weights in LDS once
// (39 KB) and every lane reads them back as broadcast ds_read_b128 -- four weights per read, no bank conflicts, VGPR operands
// the compiler can prefetch as deep as it likes.  A layer is a run-time loop over output units (four at a time: independent
// accumulation chains) with the input vector in registers.
#include <hip/hip_runtime.h>

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

// one net's parameters in LDS, PyTorch layout ([out][in] row-major), every block 16-byte aligned
template <int O, int NOUT> struct NetLds {
  static constexpr int NHP = (NOUT + 3) & ~3;
  static constexpr int W0 = 0, W1 = W0 + H * O, WH = W1 + H * H, B0 = WH + NHP * H, B1 = B0 + H, BH = B1 + H, FLOATS = BH + NHP;
  float* p;
  __device__ __forceinline__ const float* w0() const { return p + W0; }
  __device__ __forceinline__ const float* w1() const { return p + W1; }
  __device__ __forceinline__ const float* wh() const { return p + WH; }
  __device__ __forceinline__ const float* b0() const { return p + B0; }
  __device__ __forceinline__ const float* b1() const { return p + B1; }
  __device__ __forceinline__ const float* bh() const { return p + BH; }
  // cooperative copy by the whole workgroup; the caller synchronises
  __device__ __forceinline__ void stage(NET_ARGS) const {
    auto copy = [&](float* dst, const float* __restrict__ src, int n) {
      for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
    };
    copy(p + W0, w0, H * O); copy(p + W1, w1, H * H); copy(p + WH, wh, NOUT * H);
    copy(p + B0, b0, H); copy(p + B1, b1, H); copy(p + BH, bh, NOUT);
  }
};

__device__ __forceinline__ float4 lds4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// Order fence for the software-pipelined loops below: the accumulators pass through an empty volatile asm with a memory
// clobber, so the multiply-adds of the group before it cannot sink below it and the reads after it cannot rise above it.
// (A scheduling-barrier builtin alone does not do this: it pins memory operations, but instruction selection is free to place
// the arithmetic anywhere, and it put every read of a layer ahead of the first multiply-add -- 1.1 KB of spills per lane.)
template <int N> __device__ __forceinline__ void fence(float (&a)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) asm volatile("" : "+v"(a[i]) : : "memory");
}

// out[j] = act(b[j] + sum_i w[j * NIN + i] * in[i]),  j0 <= j < j1 (a multiple of four outputs), w and b in LDS.
// The input loop is fully unrolled (the input vector is a register array); left alone the scheduler hoists all 4 x NIN / 4
// broadcast reads of an iteration to its top and spills, so the reads go in groups of 8 inputs x 4 outputs (32 VGPRs) separated
// by scheduling barriers, two groups ahead of the multiply-adds that consume them.
template <int NIN, bool TANH, typename Store>
__device__ __forceinline__ void dense(const float (&in)[NIN], const float* w, const float* b, int j0, int j1, Store&& store) {
  static_assert(NIN % 4 == 0, "input width in float4");
  constexpr int NQ = NIN / 4, GQ = 2, NG = (NQ + GQ - 1) / GQ, LA = 2;     // float4 per row, per group; groups; lookahead
#pragma unroll 1
  for (int j = j0; j < j1; j += 4) {
    const float* wj = w + j * NIN;
    const float4 bv = lds4(b + j);
    float acc[4] = {bv.x, bv.y, bv.z, bv.w};
    float4 wb[NQ][4];
    auto fetch = [&](int g) {
#pragma unroll
      for (int q = g * GQ; q < (g + 1) * GQ && q < NQ; ++q)
#pragma unroll
        for (int o = 0; o < 4; ++o) wb[q][o] = lds4(wj + o * NIN + 4 * q);
    };
#pragma unroll
    for (int g = 0; g < LA && g < NG; ++g) fetch(g);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (g + LA < NG) fetch(g + LA);
#pragma unroll
      for (int q = g * GQ; q < (g + 1) * GQ && q < NQ; ++q)
#pragma unroll
        for (int o = 0; o < 4; ++o) {
          acc[o] = fmaf(wb[q][o].x, in[4 * q], acc[o]); acc[o] = fmaf(wb[q][o].y, in[4 * q + 1], acc[o]);
          acc[o] = fmaf(wb[q][o].z, in[4 * q + 2], acc[o]); acc[o] = fmaf(wb[q][o].w, in[4 * q + 3], acc[o]);
        }
      fence(acc);
    }
#pragma unroll
    for (int o = 0; o < 4; ++o) store(j + o, TANH ? tanhf(acc[o]) : acc[o]);
  }
}

// the same for a head of NOUT outputs, fully unrolled so that out[] is a register array; one output's 16 reads per group
template <int NOUT> __device__ __forceinline__ void dense_head(const float (&in)[H], const float* w, const float* b, float (&out)[NOUT]) {
  float4 wb[NOUT][H / 4];
  auto fetch = [&](int j) {
#pragma unroll
    for (int q = 0; q < H / 4; ++q) wb[j][q] = lds4(w + j * H + 4 * q);
  };
  fetch(0);
#pragma unroll
  for (int j = 0; j < NOUT; ++j) {
    if (j + 1 < NOUT) fetch(j + 1);
    float a[2] = {b[j], 0.f};
#pragma unroll
    for (int q = 0; q < H / 4; ++q) {
      a[0] = fmaf(wb[j][q].x, in[4 * q], a[0]); a[1] = fmaf(wb[j][q].y, in[4 * q + 1], a[1]);
      a[0] = fmaf(wb[j][q].z, in[4 * q + 2], a[0]); a[1] = fmaf(wb[j][q].w, in[4 * q + 3], a[1]);
    }
    fence(a);
    out[j] = a[0] + a[1];
  }
}

// out[k] = sum_j w[j * H + k] * g[j],  k < H     (back-propagation through a layer with H inputs and NJ outputs), w in LDS;
// eight k at a time, reads grouped (4 j x 8 k = 32 VGPRs) and issued two groups ahead as in dense()
template <int NJ, typename Store>
__device__ __forceinline__ void dense_t(const float (&g)[NJ], const float* w, Store&& store) {
  constexpr int KC = 8, GJ = NJ >= 4 ? 4 : NJ, NG = (NJ + GJ - 1) / GJ, LA = 2;
#pragma unroll 1
  for (int k = 0; k < H; k += KC) {
    float acc[KC];
#pragma unroll
    for (int o = 0; o < KC; ++o) acc[o] = 0.f;
    float4 wb[NJ][KC / 4];
    auto fetch = [&](int gi) {
#pragma unroll
      for (int j = gi * GJ; j < (gi + 1) * GJ && j < NJ; ++j)
#pragma unroll
        for (int q = 0; q < KC / 4; ++q) wb[j][q] = lds4(w + j * H + k + 4 * q);
    };
#pragma unroll
    for (int gi = 0; gi < LA && gi < NG; ++gi) fetch(gi);
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
      if (gi + LA < NG) fetch(gi + LA);
#pragma unroll
      for (int j = gi * GJ; j < (gi + 1) * GJ && j < NJ; ++j)
#pragma unroll
        for (int q = 0; q < KC / 4; ++q) {
          acc[4 * q] = fmaf(wb[j][q].x, g[j], acc[4 * q]); acc[4 * q + 1] = fmaf(wb[j][q].y, g[j], acc[4 * q + 1]);
          acc[4 * q + 2] = fmaf(wb[j][q].z, g[j], acc[4 * q + 2]); acc[4 * q + 3] = fmaf(wb[j][q].w, g[j], acc[4 * q + 3]);
        }
      fence(acc);
    }
#pragma unroll
    for (int o = 0; o < KC; ++o) store(k + o, acc[o]);
  }
}

template <int O> __device__ __forceinline__ void load_row(const float* __restrict__ p, float (&x)[O]) {
  static_assert(O % 4 == 0, "observation rows are read as float4");
#pragma unroll
  for (int i = 0; i < O; i += 4) {
    const float4 v = *reinterpret_cast<const float4*>(p + i);
    x[i] = v.x; x[i + 1] = v.y; x[i + 2] = v.z; x[i + 3] = v.w;
  }
}

__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) x += __shfl_xor(x, s, 64);
  return x;
}

// ------------------------------------------------------------------------------------------------ act
// grid (ceil(n / 64), 2): blockIdx.y = 0 critic (value), 1 actor (action, log-prob).  256 threads = four wavefronts working on
// the SAME 64 rows: each computes a quarter of a layer's units and they meet in LDS ([unit][row]) -- at 4096 envs the rollout
// waits for this kernel, so its latency counts, not its throughput.
template <int O, int A, bool ACTOR>
__device__ __forceinline__ void act_net(NET_ARGS, const float* __restrict__ logstd, const float* __restrict__ obs,
                                        const float* __restrict__ noise, int n, float* value_out, float* action_out, float* logp_out,
                                        float* smem) {
  constexpr int NOUT = ACTOR ? A : 1;
  using NL = NetLds<O, NOUT>;
  const NL net{smem};
  float* hb = smem + ((NL::FLOATS + 3) & ~3);            // [H][64]
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, row = blockIdx.x * 64 + lane;
  const bool on = row < n;
  const int r = on ? row : n - 1;
  net.stage(w0, b0, w1, b1, wh, bh);
  float x[O];
  load_row<O>(obs + (size_t)r * O, x);
  __syncthreads();
  auto to_lds = [&](int j, float v) { hb[j * 64 + lane] = v; };
  float h[H];
  dense<O, true>(x, net.w0(), net.b0(), 16 * wv, 16 * wv + 16, to_lds);
  __syncthreads();
#pragma unroll
  for (int j = 0; j < H; ++j) h[j] = hb[j * 64 + lane];
  __syncthreads();
  dense<H, true>(h, net.w1(), net.b1(), 16 * wv, 16 * wv + 16, to_lds);
  __syncthreads();
  if (wv != 0) return;                                   // the heads are small: one wavefront finishes
#pragma unroll
  for (int j = 0; j < H; ++j) h[j] = hb[j * 64 + lane];
  float out[NOUT];
  dense_head<NOUT>(h, net.wh(), net.bh(), out);
  if constexpr (!ACTOR) {
    if (on) value_out[row] = out[0];
  } else {
    float lp = 0.f;
#pragma unroll
    for (int a = 0; a < A; ++a) {
      const float mean = out[a], ls = logstd[a];
      const float act = noise ? fmaf(expf(ls), noise[(size_t)r * A + a], mean) : mean;       // policy.py:40-43
      const float z = (act - mean) * expf(-ls);                                              // ModNormal.log_probs, policy.py:171-173
      lp += -0.5f * z * z - ls - HALF_LOG_2PI;
      if (on) action_out[(size_t)row * A + a] = act;
    }
    if (on) logp_out[row] = lp;
  }
}

template <int O, int A> constexpr size_t act_smem_bytes() { return (size_t)(((NetLds<O, A>::FLOATS + 3) & ~3) + H * 64) * sizeof(float); }

template <int O, int A>
__global__ void __launch_bounds__(256)
policy_act_kernel(CRITIC_ARGS, ACTOR_ARGS, const float* __restrict__ obs, const float* __restrict__ noise, int n, float* value_out,
                  float* action_out, float* logp_out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if (blockIdx.y == 0) act_net<O, A, false>(cw0, cb0, cw1, cb1, cwh, cbh, nullptr, obs, noise, n, value_out, action_out, logp_out, smem);
  else act_net<O, A, true>(aw0, ab0, aw1, ab1, awh, abh, logstd, obs, noise, n, value_out, action_out, logp_out, smem);
}

// ------------------------------------------------------------------------------------------------ mini-batch gradients, stage 1
// grid (ceil(m / 256), 2): blockIdx.y = 0 critic, 1 actor; 256 threads = 256 rows sharing one LDS copy of the net.
// Row r of the mini-batch is sample perm[*offset + r].
// Outputs, all [unit][m] (coalesced for lane = row; the caller's GEMMs read them as K-major operands):
//   xt0 [O][m]        gathered observations                         (written by the critic blocks)
//   xt1, xt2 [H][m]   hidden activations of the net
//   g1, g2 [H][m]     d loss / d pre-activation of hidden layer 1, 2
//   gh [NOUT][m]      d loss / d head output (critic: value_coef * d value-loss / d v; actor: d action-loss / d mean)
//   partials [ceil(m/64)][3 + A] per wavefront: critic [0] = sum value loss; actor [1] = sum action loss, [2] = rows,
//                             [3 + a] = sum d action-loss / d logstd_a          (the caller adds them and the entropy term)
// Loss arithmetic and sub-gradient conventions: ppo_loss_kernel in solorl_hip.hip (agents/ppo/ppo.py:52-74).
template <int O, int A, bool ACTOR>
__device__ __forceinline__ void grad_net(NET_ARGS, const float* __restrict__ logstd, const solorl_ppo_batch& B, float* xt0, float* xt1,
                                         float* xt2, float* g1, float* g2, float* gh, float* partials, float* smem) {
  constexpr int NOUT = ACTOR ? A : 1;
  using NL = NetLds<O, NOUT>;
  const NL net{smem};
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, row = blockIdx.x * 256 + threadIdx.x, m = B.m;
  float* hb = smem + ((NetLds<O, A>::FLOATS + 3) & ~3) + wv * (H * 64);     // this wavefront's [unit][lane] bounce buffer
  const bool on = row < m;
  const int r = on ? row : m - 1;                       // (rows past the end recompute the last row and store nothing)
  net.stage(w0, b0, w1, b1, wh, bh);
  const long long s = B.perm[*B.offset + r];
  const float inv_m = 1.0f / (float)m;
  float x[O];
  load_row<O>(B.obs + (size_t)s * O, x);
  if (!ACTOR && on) {
#pragma unroll
    for (int i = 0; i < O; ++i) xt0[(size_t)i * m + row] = x[i];
  }
  __syncthreads();
  // a layer's outputs are produced unit by unit in a run-time loop, but the next layer wants them as a register vector:
  // they bounce through LDS (and leave for the weight-gradient GEMMs on the way)
  float h[H];
  dense<O, true>(x, net.w0(), net.b0(), 0, H, [&](int j, float v) { hb[j * 64 + lane] = v; if (on) xt1[(size_t)j * m + row] = v; });
#pragma unroll
  for (int j = 0; j < H; ++j) h[j] = hb[j * 64 + lane];
  dense<H, true>(h, net.w1(), net.b1(), 0, H, [&](int j, float v) { hb[j * 64 + lane] = v; if (on) xt2[(size_t)j * m + row] = v; });
#pragma unroll
  for (int j = 0; j < H; ++j) h[j] = hb[j * 64 + lane];                      // h = h2 from here on
  float* P_ = partials + (size_t)(row >> 6) * (3 + A);
  auto raw = [&](int k, float v) { hb[k * 64 + lane] = v; };
  float out[NOUT];
  dense_head<NOUT>(h, net.wh(), net.bh(), out);
  if constexpr (!ACTOR) {
    const float v = out[0];
    const float ret = B.ret[s], vp = B.vpred[s];
    const float u = v - ret;
    float vl, gv;
    if (B.clipped_value) {                                       // ppo.py:61-66
      const float dvp = v - vp;
      const float wv_ = vp + fminf(fmaxf(dvp, -B.clip), B.clip) - ret;
      const float ins = (dvp >= -B.clip && dvp <= B.clip) ? 1.0f : 0.0f;
      const float uu = u * u, ww = wv_ * wv_;
      vl = 0.5f * fmaxf(uu, ww);
      gv = uu > ww ? u : (uu < ww ? wv_ * ins : 0.5f * (u + wv_ * ins));
    } else { vl = 0.5f * u * u; gv = u; }                        // ppo.py:67-68
    const float gout[1] = {on ? B.value_coef * gv * inv_m : 0.f};
    if (on) gh[row] = gout[0];
    const float svl = wave_sum(on ? vl : 0.f);
    if (lane == 0 && on) P_[0] = svl;
    dense_t<1>(gout, net.wh(), raw);
  } else {
    float z[A], e[A], lp = 0.f;
#pragma unroll
    for (int a = 0; a < A; ++a) {
      const float ls = logstd[a];
      e[a] = expf(-ls);
      z[a] = (B.actions[(size_t)s * A + a] - out[a]) * e[a];
      lp += -0.5f * z[a] * z[a] - ls - HALF_LOG_2PI;
    }
    const float ratio = expf(lp - B.old_logp[s]), adv = B.adv[s];                                  // ppo.py:54-59
    const float s1 = ratio * adv, s2 = fminf(fmaxf(ratio, 1.0f - B.clip), 1.0f + B.clip) * adv;
    const float inside = (ratio >= 1.0f - B.clip && ratio <= 1.0f + B.clip) ? 1.0f : 0.0f;
    const float wsel = s1 < s2 ? 1.0f : (s1 > s2 ? inside : 0.5f * (1.0f + inside));
    const float glp = on ? -adv * ratio * wsel * inv_m : 0.f;
    float gout[A];
#pragma unroll
    for (int a = 0; a < A; ++a) {
      gout[a] = glp * z[a] * e[a];                                 // d logp / d mean_a = z / sigma
      if (on) gh[(size_t)a * m + row] = gout[a];
      const float sg = wave_sum(on ? glp * (z[a] * z[a] - 1.0f) : 0.f);       // d logp / d logstd_a = z^2 - 1
      if (lane == 0 && on) P_[3 + a] = sg;
    }
    const float sal = wave_sum(on ? -fminf(s1, s2) : 0.f);
    if (lane == 0 && on) { P_[1] = sal; P_[2] = (float)min(m - (row & ~63), 64); }
    dense_t<A>(gout, net.wh(), raw);
  }
  // d loss / d pre-activation of layer 2 = (W_head^T g_head) * (1 - h2^2)
  float d2[H];
#pragma unroll
  for (int k = 0; k < H; ++k) {
    d2[k] = hb[k * 64 + lane] * (1.0f - h[k] * h[k]);
    if (on) g2[(size_t)k * m + row] = d2[k];
  }
  // ... of layer 1 = (W1^T d2) * (1 - h1^2); h1 comes back from xt1 (this lane's own stores), requested before the products
#pragma unroll
  for (int k = 0; k < H; ++k) h[k] = xt1[(size_t)k * m + r];
  dense_t<H>(d2, net.w1(), raw);
#pragma unroll
  for (int k = 0; k < H; ++k)
    if (on) g1[(size_t)k * m + row] = hb[k * 64 + lane] * (1.0f - h[k] * h[k]);
}

template <int O, int A> constexpr size_t grad_smem_bytes() { return (size_t)(((NetLds<O, A>::FLOATS + 3) & ~3) + 4 * H * 64) * sizeof(float); }

template <int O, int A>
__global__ void __launch_bounds__(256)
ppo_grad_stage1_kernel(CRITIC_ARGS, ACTOR_ARGS, const solorl_ppo_batch B, const solorl_ppo_stage1 W) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if (blockIdx.y == 0) grad_net<O, A, false>(cw0, cb0, cw1, cb1, cwh, cbh, nullptr, B, W.xt0, W.c_xt1, W.c_xt2, W.c_g1, W.c_g2, W.c_gh, W.partials, smem);
  else grad_net<O, A, true>(aw0, ab0, aw1, ab1, awh, abh, logstd, B, W.xt0, W.a_xt1, W.a_xt2, W.a_g1, W.a_g2, W.a_gh, W.partials, smem);
}

// ------------------------------------------------------------------------------------------------ mini-batch gradients, stage 2
// Weight gradients d W[u][k] = sum_r G[u][r] X[k][r] of the six layers from stage 1's [unit][row] arrays (X carries a row of
// ones, so column K of the product is the bias gradient).  One wavefront per tile of 8 x 8 outputs and a chunk of CHUNK rows:
// lane = row again -- sixteen coalesced loads feed 64 multiply-adds per row step -- then a halving butterfly leaves lane l with the
// chunk's total of output l.  Chunk partials go to scratch and are added by stage 3 in a fixed order (deterministic, no atomics).
constexpr int TB = 8, CHUNK = 4096;
struct LayerDesc { const float* g; const float* x; float* wgrad; float* bgrad; int U, K1, off, tile0; };   // K1 = inputs + 1
struct Stage2Args { LayerDesc L[6]; int m, nchunks, total, ntiles; float* scratch; };

__global__ void __launch_bounds__(256) ppo_grad_stage2_kernel(const Stage2Args S) {
  // a workgroup = four wavefronts on 2 x 2 neighbouring tiles of the same row chunk: the G rows and X rows each of them streams
  // are also streamed by one neighbour at the same time, i.e. they hit in the CU's L1 (the kernel is bound by L2 traffic)
  int li = 0;
#pragma unroll
  for (int i = 1; i < 6; ++i) if ((int)blockIdx.x >= S.L[i].tile0) li = i;
  const LayerDesc& L = S.L[li];
  const int kb = (L.K1 + TB - 1) / TB, ub = (L.U + TB - 1) / TB, kb2 = (kb + 1) / 2, ub2 = (ub + 1) / 2;
  const int t = blockIdx.x - L.tile0, wv = threadIdx.x >> 6;
  const int chunk = t / (ub2 * kb2), tt = t % (ub2 * kb2), ut = 2 * (tt / kb2) + (wv >> 1), kt = 2 * (tt % kb2) + (wv & 1);
  if (ut >= ub || kt >= kb) return;
  const int u0 = ut * TB, k0 = kt * TB;
  const int lane = threadIdx.x & 63, m = S.m;
  const int r0 = chunk * CHUNK, r1 = min(r0 + CHUNK, m);
  float acc[TB * TB];
#pragma unroll
  for (int i = 0; i < TB * TB; ++i) acc[i] = 0.f;
  // rows of the tile that exist (a tile may hang over the edge of the layer): clamp the index, zero the factor
  const float* gp[TB]; const float* xp[TB]; float gm[TB], xm[TB];
#pragma unroll
  for (int i = 0; i < TB; ++i) {
    gm[i] = u0 + i < L.U ? 1.f : 0.f; gp[i] = L.g + (size_t)min(u0 + i, L.U - 1) * m;
    xm[i] = k0 + i < L.K1 ? 1.f : 0.f; xp[i] = L.x + (size_t)min(k0 + i, L.K1 - 1) * m;
  }
#pragma unroll 2
  for (int r = r0 + lane; r < r1; r += 64) {
    float g[TB], x[TB];
#pragma unroll
    for (int i = 0; i < TB; ++i) { g[i] = gp[i][r] * gm[i]; x[i] = xp[i][r] * xm[i]; }
#pragma unroll
    for (int i = 0; i < TB; ++i)
#pragma unroll
      for (int j = 0; j < TB; ++j) acc[i * TB + j] = fmaf(g[i], x[j], acc[i * TB + j]);
  }
  // butterfly: after the step with distance d a lane keeps the half of its values whose index has bit d equal to its own
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const bool up = (lane & d) != 0;
#pragma unroll
    for (int i = 0; i < d; ++i) {
      const float keep = up ? acc[i + d] : acc[i], send = up ? acc[i] : acc[i + d];
      acc[i] = keep + __shfl_xor(send, d, 64);
    }
  }
  const int i = lane / TB, j = lane % TB;                   // lane l now holds output l = i * 8 + j of the tile
  if (u0 + i < L.U && k0 + j < L.K1) S.scratch[(size_t)chunk * S.total + L.off + (u0 + i) * L.K1 + (k0 + j)] = acc[0];
}

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
  float a = 0.f;
  for (int c = 0; c < S.nchunks; ++c) a += S.scratch[(size_t)c * S.total + e];
  const int q = e - L.off, u = q / L.K1, k = q % L.K1;
  if (k == L.K1 - 1) L.bgrad[u] = a; else L.wgrad[u * (L.K1 - 1) + k] = a;
}

template <typename F> int dispatch_dims(int O, int A, F&& f) {
  if (O == 76 && A == 12) return f(std::integral_constant<int, 76>(), std::integral_constant<int, 12>());
  if (O == 84 && A == 12) return f(std::integral_constant<int, 84>(), std::integral_constant<int, 12>());
  if (O == 60 && A == 8) return f(std::integral_constant<int, 60>(), std::integral_constant<int, 8>());
  if (O == 68 && A == 8) return f(std::integral_constant<int, 68>(), std::integral_constant<int, 8>());
  return solorl_fail_(SOLORL_ERR_INVALID, "policy kernels are built for the observation / action sizes of one history level "
                                          "(76 or 84 x 12, 60 or 68 x 8); use the PyTorch path for other shapes");
}

int check_policy(const solorl_policy_params* p, int device_id) {
  if (!p) return solorl_fail_(SOLORL_ERR_INVALID, "null policy parameters");
  if (p->hidden != H) return solorl_fail_(SOLORL_ERR_INVALID, "policy kernels are built for hidden size 64");
  const void* ptrs[] = {p->critic_w0, p->critic_b0, p->critic_w1, p->critic_b1, p->critic_w2, p->critic_b2, p->actor_w0, p->actor_b0,
                        p->actor_w1, p->actor_b1, p->mean_w, p->mean_b, p->logstd};
  for (const void* q : ptrs) if (!q) return solorl_fail_(SOLORL_ERR_INVALID, "null policy parameter pointer");
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
  const solorl_policy_params P = *p;
  return dispatch_dims(P.obs_dim, P.act_dim, [&](auto oc, auto ac) {
    constexpr int O = decltype(oc)::value, A = decltype(ac)::value;
    constexpr size_t smem = act_smem_bytes<O, A>();
    hipLaunchKernelGGL((policy_act_kernel<O, A>), dim3((n + 63) / 64, 2), dim3(256), smem, (hipStream_t)stream, CRITIC_PASS, ACTOR_PASS, obs, noise,
                       n, value_out, action_out, logp_out);
    return hipGetLastError() == hipSuccess ? 0 : solorl_fail_(SOLORL_ERR_HIP, "policy_act_kernel launch");
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
  const solorl_policy_params P = *p;
  const solorl_ppo_batch B = *batch;
  const solorl_ppo_stage1 W = *work;
  return dispatch_dims(P.obs_dim, P.act_dim, [&](auto oc, auto ac) {
    constexpr int O = decltype(oc)::value, A = decltype(ac)::value;
    constexpr size_t smem = grad_smem_bytes<O, A>();                 // 39 KB of weights + 4 x 16 KB: above the 64 KB default
    static bool raised[16] = {};                                     // per device: once is enough (and keeps captures free of it)
    if (device_id < 16 && !raised[device_id]) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&ppo_grad_stage1_kernel<O, A>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)smem) != hipSuccess)
        return solorl_fail_(SOLORL_ERR_HIP, "hipFuncSetAttribute(ppo_grad_stage1_kernel)");
      raised[device_id] = true;
    }
    hipLaunchKernelGGL((ppo_grad_stage1_kernel<O, A>), dim3((B.m + 255) / 256, 2), dim3(256), smem, (hipStream_t)stream, CRITIC_PASS, ACTOR_PASS, B, W);
    return hipGetLastError() == hipSuccess ? 0 : solorl_fail_(SOLORL_ERR_HIP, "ppo_grad_stage1_kernel launch");
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
  S.m = m; S.nchunks = (m + CHUNK - 1) / CHUNK; S.scratch = out->scratch;
  int off = 0, tile = 0;
  for (int i = 0; i < 6; ++i) {
    S.L[i] = L[i]; S.L[i].off = off; S.L[i].tile0 = tile;
    off += L[i].U * L[i].K1;
    tile += (((L[i].U + TB - 1) / TB + 1) / 2) * (((L[i].K1 + TB - 1) / TB + 1) / 2) * S.nchunks;       // workgroups of 2 x 2 tiles
  }
  S.total = off; S.ntiles = tile;
  T.partials = work->partials; T.nwaves = (m + 63) / 64; T.A = A; T.logstd = p->logstd; T.logstd_grad = out->logstd;
  T.loss_sums = out->loss_sums; T.logstd_sum = out->logstd_sum; T.entropy_coef = out->entropy_coef;
  hipLaunchKernelGGL(ppo_grad_stage2_kernel, dim3(S.ntiles), dim3(256), 0, (hipStream_t)stream, S);
  hipLaunchKernelGGL(ppo_grad_stage3_kernel, dim3((S.total + 255) / 256 + 1), dim3(256), 0, (hipStream_t)stream, T);
  return hipGetLastError() == hipSuccess ? 0 : solorl_fail_(SOLORL_ERR_HIP, "ppo_grad_stage2/3 launch");
}

}  // extern "C"
