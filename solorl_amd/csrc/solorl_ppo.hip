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
#include <utility>

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

// tanh(x) = 1 - 2 / (1 + e^(2x)) on the hardware exp2 / reciprocal (v_exp_f32, v_rcp_f32: 1 ulp each): 5 instructions instead of libm's
// ~40, absolute error <= 3e-7 over the whole line (saturates cleanly: e^(2x) = inf -> 1, 0 -> -1) -- the level of the f32 sums that
// feed it.  The relative error near 0 is larger (1e-4 at |x| = 1e-3), which an activation bounded by 1 does not care about.
// (__frcp_rn is the correctly rounded reciprocal: a 10-instruction division sequence, 64 times per net and tile -- a fifth of stage 1's
// vector instructions.)
__device__ __forceinline__ float tanh_fast(float x) {
  const float e = __expf(2.0f * x);
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + e);
}

template <typename F, int... I> __device__ __forceinline__ void unrolled_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F> __device__ __forceinline__ void unrolled(F&& f) { unrolled_impl(f, std::make_integer_sequence<int, N>{}); }

__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) x += __shfl_xor(x, s, 64);
  return x;
}

// ------------------------------------------------------------------------------------------------ stage 2 / 3 bookkeeping
// Stage 2 (MFMA, below) leaves per-chunk partial products in scratch; stage 3 adds them in a fixed order (reproducible, no
// atomics) straight into the parameters' gradients.  Column K of a product is the bias gradient (the row sum of G).
// K1 = inputs + 1; off: the layer's first element in the flat gradient order; the layer's rows are cut into nchunks chunks of mchunk rows
// (wavefronts wave0 .. wave0 + nchunks - 1 of stage 2), chunk c's partial product sits at scratch[sbase + c U K1 ..]
struct LayerDesc { const float* g; const float* x; float* wgrad; float* bgrad; int U, K1, off, nchunks, mchunk, wave0, sbase, reserved; };
struct Stage2Args { LayerDesc L[6]; int m, total, nwaves, reserved; float* scratch; };

// stage 3: add the chunks (fixed order) into the parameters' gradients; 3 + A more workgroups finish the log-std gradient and the
// running loss sums from stage 1's per-wavefront partials, one column each (a single workgroup walking the columns one after the
// other was the kernel's critical path: 8 us at 512 rows of partials, 18 us at 1024)
struct Stage3Args { Stage2Args S; const float* partials; int nwaves, A; const float* logstd; float* logstd_grad; float* loss_sums;
                    float* logstd_sum; float entropy_coef; };
__global__ void __launch_bounds__(256) ppo_grad_stage3_kernel(const Stage3Args T) {
  const Stage2Args& S = T.S;
  const int nel = (S.total + 15) / 16;        // element workgroups; then one workgroup per column of partials [nwaves][3 + A]
  if ((int)blockIdx.x >= nel) {
    __shared__ float red[4];
    const int nc = 3 + T.A, c = blockIdx.x - nel;
    float a = 0.f;
    for (int w0 = 0; w0 < T.nwaves; w0 += 2048) {         // (eight independent loads per thread and trip)
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) { const int w = w0 + threadIdx.x + 256 * i; v[i] = T.partials[(size_t)min(w, T.nwaves - 1) * nc + c]; if (w >= T.nwaves) v[i] = 0.f; }
      a += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
    a = wave_sum(a);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
      a = (red[0] + red[1]) + (red[2] + red[3]);
      T.loss_sums[c] += a;
      if (c >= 3) T.logstd_grad[c - 3] = a - T.entropy_coef / (float)T.A;      // entropy = mean over batch and dims of logstd + const
      if (c == 0) { float e = 0.f; for (int k = 0; k < T.A; ++k) e += T.logstd[k]; *T.logstd_sum += e; }
    }
    return;
  }
  // 16 elements x 16 chunk groups per workgroup: thread (el, grp) adds chunks grp, grp + 16, ... of its element (all loads independent:
  // one memory round trip for up to 256 chunks), the groups' sums are added in a fixed order through LDS
  __shared__ float part[16][17];
  const int el = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const bool on = blockIdx.x * 16 + el < S.total;
  const int e = on ? blockIdx.x * 16 + el : S.total - 1;
  LayerDesc L = S.L[0];                  // (selects: a dynamic index into the argument struct would be served from scratch)
#pragma unroll
  for (int i = 1; i < 6; ++i) if (e >= S.L[i].off) L = S.L[i];
  const float* __restrict__ src = S.scratch + L.sbase + (e - L.off);
  const size_t stride = (size_t)L.U * L.K1;
  float a16[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {           // (unconditional loads from a clamped chunk, then a select: a branch per load serialises them)
    const int c = grp + 16 * i;
    a16[i] = src[(size_t)min(c, L.nchunks - 1) * stride];
  }
  asm volatile("" ::: "memory");
#pragma unroll
  for (int i = 0; i < 16; ++i) { asm volatile("" : "+v"(a16[i])); if (grp + 16 * i >= L.nchunks) a16[i] = 0.f; }     // (pinned: the loads stay unconditional)
  float rem = 0.f;
  if (on) for (int c = grp + 256; c < L.nchunks; c += 16) rem += src[(size_t)c * stride];      // (more than 256 chunks: not with stage2_plan's ~1024 wavefronts)
#pragma unroll
  for (int d = 8; d > 0; d >>= 1)
#pragma unroll
    for (int i = 0; i < d; ++i) a16[i] += a16[i + d];
  part[grp][el] = a16[0] + rem;
  __syncthreads();
  if (grp != 0 || !on) return;
  float a = 0.f;
#pragma unroll
  for (int g = 0; g < 16; ++g) a += part[g][el];
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
// leave their registers between layers, forward or backward.  The matching A element is W[unit 32 mt + c][input 8 q + 4 h + t]: a
// lane reads 16 consecutive bytes of one weight row per q.  The act kernel keeps a net's forward weights in registers (NetRegs, 170
// VGPRs, one wavefront per 32 rows); stage 1 needs the transposed weights too and keeps the per-lane operand image in LDS instead
// (NetLds: one copy per workgroup of four wavefronts, ds_read_b128 per four MFMAs).
typedef float f32x16 __attribute__((ext_vector_type(16)));
#ifdef SOLO_PPO_NOMFMA      // dev experiment (tools/dev/build_ppo_variant.py): what the kernels cost without their matrix products
#define MFMA32(a, b, c) (c)
#else
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#endif

template <int O, int NOUT, bool BACKWARD> struct NetRegs {
  static constexpr int QO = (O + 7) / 8, QA = (NOUT + 7) / 8;
  float a0[2][QO][4], a1[2][8][4], ah[8][4];
  float aht[BACKWARD ? 2 : 1][QA][4], a1t[BACKWARD ? 2 : 1][BACKWARD ? 8 : 1][4];
  // operand access, four consecutive contraction steps at a time (the interface NetLds shares)
  __device__ __forceinline__ float4 A0(int mt, int q) const { return make_float4(a0[mt][q][0], a0[mt][q][1], a0[mt][q][2], a0[mt][q][3]); }
  __device__ __forceinline__ float4 A1(int mt, int q) const { return make_float4(a1[mt][q][0], a1[mt][q][1], a1[mt][q][2], a1[mt][q][3]); }
  __device__ __forceinline__ float4 AH(int q) const { return make_float4(ah[q][0], ah[q][1], ah[q][2], ah[q][3]); }
  __device__ __forceinline__ float4 AHT(int mt, int q) const { return make_float4(aht[mt][q][0], aht[mt][q][1], aht[mt][q][2], aht[mt][q][3]); }
  __device__ __forceinline__ float4 A1T(int mt, int q) const { return make_float4(a1t[mt][q][0], a1t[mt][q][1], a1t[mt][q][2], a1t[mt][q][3]); }
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

// The same operands held ONCE per workgroup in LDS, as the per-lane register image of NetRegs: entry e = one (array, tile, q) of four
// consecutive contraction steps, stored [e][lane] as float4 -- a wavefront's ds_read_b128 of an entry is 1 KB of consecutive bytes
// (conflict-free), feeds four MFMAs (256 cycles), and nothing stays in registers between uses.  Stage 1 held the net in 250 registers
// per lane, spilled its activations to AGPRs and scratch around them (256 + 256 registers, 160 B of scratch) and re-read the 64 KB
// image per wavefront; with the image in LDS four wavefronts share one copy and two workgroups fit a CU.
template <int O, int NOUT> struct NetLds {
  static constexpr int QO = (O + 7) / 8, QA = (NOUT + 7) / 8;
  static constexpr int E_A0 = 0, E_A1 = E_A0 + 2 * QO, E_AH = E_A1 + 16, E_AHT = E_AH + 8, E_A1T = E_AHT + 2 * QA, NE = E_A1T + 16;
  static constexpr int BYTES = NE * 64 * 16;
  const float4* img;        // + lane
  __device__ __forceinline__ float4 A0(int mt, int q) const { return img[(E_A0 + mt * QO + q) * 64]; }
  __device__ __forceinline__ float4 A1(int mt, int q) const { return img[(E_A1 + mt * 8 + q) * 64]; }
  __device__ __forceinline__ float4 AH(int q) const { return img[(E_AH + q) * 64]; }
  __device__ __forceinline__ float4 AHT(int mt, int q) const { return img[(E_AHT + mt * QA + q) * 64]; }
  __device__ __forceinline__ float4 A1T(int mt, int q) const { return img[(E_A1T + mt * 8 + q) * 64]; }
  // entry e of the image for lane (c, h), from the PyTorch-layout parameters (what NetRegs::load puts into registers)
  static __device__ __forceinline__ float4 entry(int e, NET_ARGS, int c, int h) {
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (e < E_A1) {
      const int mt = e / QO, q = e % QO, i0 = 8 * q + 4 * h;
      const float* wr = w0 + (size_t)(32 * mt + c) * O + i0;
      if constexpr (O % 4 == 0) { if (i0 < O) return *reinterpret_cast<const float4*>(wr); }
      else {
#pragma unroll
        for (int t = 0; t < 4; ++t) if (i0 + t < O) v[t] = wr[t];
      }
    } else if (e < E_AH) {
      const int mt = (e - E_A1) / 8, q = (e - E_A1) % 8;
      return *reinterpret_cast<const float4*>(w1 + (32 * mt + c) * H + 8 * q + 4 * h);
    } else if (e < E_AHT) {
      if (c < NOUT) return *reinterpret_cast<const float4*>(wh + c * H + 8 * (e - E_AH) + 4 * h);
    } else if (e < E_A1T) {
      const int mt = (e - E_AHT) / QA, q = (e - E_AHT) % QA;
#pragma unroll
      for (int t = 0; t < 4; ++t) { const int a = 8 * q + 4 * h + t; if (a < NOUT) v[t] = wh[a * H + 32 * mt + c]; }
    } else {
      const int mt = (e - E_A1T) / 8, q = (e - E_A1T) % 8;
#pragma unroll
      for (int t = 0; t < 4; ++t) v[t] = w1[(8 * q + 4 * h + t) * H + 32 * mt + c];
    }
    return make_float4(v[0], v[1], v[2], v[3]);
  }
};

// the unit a D register belongs to
__device__ __forceinline__ constexpr int unit_of(int mt, int v, int h) { return 32 * mt + 8 * (v >> 2) + 4 * h + (v & 3); }

// layers 1 and 2 and the head for one 32-row tile; X = the lane's row (loaded by the caller: its latency is the caller's to hide).  After the call h1, h2 hold
// tanh activations (D layout), dh the head outputs (unit = unit_of(0, v, h), valid below NOUT).
__device__ __forceinline__ float comp4(const float4& f, int t) { return t == 0 ? f.x : (t == 1 ? f.y : (t == 2 ? f.z : f.w)); }   // (t: unrolled, compile-time)

// the lane's share of an observation row: inputs 8 q + 4 h .. + 3 for every q (zero past O)
template <int O> struct ObsRow {
  static constexpr int QO = (O + 7) / 8;
  float x[QO][4];
  __device__ __forceinline__ void load(const float* __restrict__ xrow, int h) {
#pragma unroll
    for (int q = 0; q < QO; ++q) {
      const int i0 = 8 * q + 4 * h;
#pragma unroll
      for (int t = 0; t < 4; ++t) x[q][t] = 0.f;
      if constexpr (O % 4 == 0) {
        if (i0 < O) { const float4 xv = *reinterpret_cast<const float4*>(xrow + i0); x[q][0] = xv.x; x[q][1] = xv.y; x[q][2] = xv.z; x[q][3] = xv.w; }
      } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) if (i0 + t < O) x[q][t] = xrow[i0 + t];
      }
    }
  }
};

template <int O, int NOUT, typename WT, typename S1, typename S2>
__device__ __forceinline__ void mfma_forward(const WT& R, const float* __restrict__ b0, const float* __restrict__ b1,
                                             const float* __restrict__ bh, const ObsRow<O>& X, int h, f32x16 (&h1)[2],
                                             f32x16 (&h2)[2], f32x16& dh, S1&& store_x, S2&& store_h) {
  constexpr int QO = WT::QO;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int v = 0; v < 16; ++v) { h1[mt][v] = b0[unit_of(mt, v, h)]; h2[mt][v] = b1[unit_of(mt, v, h)]; }
#pragma unroll
  for (int v = 0; v < 16; ++v) { const int a = unit_of(0, v, h); dh[v] = a < NOUT ? bh[a] : 0.f; }
#pragma unroll
  for (int q = 0; q < QO; ++q) {
    const int i0 = 8 * q + 4 * h;
    const float (&x)[4] = X.x[q];
    if (i0 < O) store_x(i0, x);
    const float4 wa = R.A0(0, q), wb = R.A0(1, q);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      h1[0] = MFMA32(comp4(wa, t), x[t], h1[0]);
      h1[1] = MFMA32(comp4(wb, t), x[t], h1[1]);
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
    for (int qq = 0; qq < 4; ++qq) {
      const float4 wa = R.A1(0, 4 * mt + qq), wb = R.A1(1, 4 * mt + qq);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        h2[0] = MFMA32(comp4(wa, t), h1[mt][4 * qq + t], h2[0]);
        h2[1] = MFMA32(comp4(wb, t), h1[mt][4 * qq + t], h2[1]);
      }
    }
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int v = 0; v < 16; ++v) h2[mt][v] = tanh_fast(h2[mt][v]);
  store_h(2, h2);
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
      const float4 wa = R.AH(4 * mt + qq);
#pragma unroll
      for (int t = 0; t < 4; ++t) dh = MFMA32(comp4(wa, t), h2[mt][4 * qq + t], dh);
    }
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
  ObsRow<O> X;
  X.load(obs + (size_t)r * O, h);
  mfma_forward<O, NOUT>(R, b0, b1, bh, X, h, h1, h2, dh, [](int, const float (&)[4]) {}, [](int, f32x16 (&)[2]) {});
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

// ---------------------------------------------------------------- stage 1 on MFMA: grid (ceil(m / 128), 2), workgroups of four wavefronts,
// one net per workgroup with its operand image in LDS (NetLds), one 32-row tile per wavefront (= one row of partials per 32 samples)
template <int O, int A, bool ACTOR>
__device__ __forceinline__ void grad_net_mfma(NET_ARGS, const float* __restrict__ logstd, const solorl_ppo_batch& B, float* xt0, float* xt1,
                                              float* xt2, float* g1, float* g2, float* gh, float* partials, float4* image) {
  constexpr int NOUT = ACTOR ? A : 1;
  using NR = NetLds<O, NOUT>;
  const int l = threadIdx.x & 63, wave = threadIdx.x >> 6, c = l & 31, h = l >> 5, m = B.m;
  // the tile's row: its index through the permutation, then its observation and its scalars -- issued BEFORE the image is filled, so that
  // the two dependent memory round trips run under the fill instead of after it
  const int tile = blockIdx.x * 4 + wave, tile0 = tile * 32;
  const int row = tile0 + c;
  const bool on = row < m;
  const int r = on ? row : m - 1;
#ifdef SOLO_S1_NOSTORE               // dev experiment (tools/dev/build_ppo_variant.py): the [unit][row] stores compiled but never executed
  const bool st = on && m < 0;
#else
  const bool st = on;
#endif
  const long long s = B.perm[*B.offset + r];
  ObsRow<O> X;
  X.load(B.obs + (size_t)s * O, h);
  const float sc_ret = B.ret[s], sc_vp = B.vpred[s], sc_adv = B.adv[s], sc_olp = B.old_logp[s];
  float sc_act[ACTOR ? 16 : 1] = {};           // the row's actions, in the head's D layout
  if constexpr (ACTOR) {
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int a = unit_of(0, v, h);
      if ((8 * (v >> 2) + (v & 3)) < A) sc_act[v] = B.actions[(size_t)s * A + min(a, A - 1)];
    }
  }
  // fill: entry e by wavefront e mod 4, all loads of a wavefront in flight together
  {
    float4 v[(NR::NE + 3) / 4];
#pragma unroll
#ifdef SOLO_S1_NOFILL          // dev experiment: the image without its global loads
    for (int i = 0; i < (NR::NE + 3) / 4; ++i) v[i] = make_float4(0.01f * c, 0.02f, -0.01f * h, 0.03f);
#else
    for (int i = 0; i < (NR::NE + 3) / 4; ++i) { const int e = 4 * i + wave; if (e < NR::NE) v[i] = NR::entry(e, w0, b0, w1, b1, wh, bh, c, h); }
#endif
#pragma unroll
    for (int i = 0; i < (NR::NE + 3) / 4; ++i) { const int e = 4 * i + wave; if (e < NR::NE) image[e * 64 + l] = v[i]; }
  }
  __syncthreads();
  const NR R{image + l};
  if (tile0 >= m) return;
  const float inv_m = 1.0f / (float)m;
  float loss_acc = 0.f, gls_acc[ACTOR ? 16 : 1] = {};
  const int rows_here = min(m - tile0, 32);
  {
    f32x16 h1[2], h2[2], dh;
    auto store_units = [&](float* arr, f32x16 (&a)[2]) {
      if (st) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int v = 0; v < 16; ++v) arr[(unsigned)((tile * H + unit_of(mt, v, h)) * 32 + c)] = a[mt][v];     // [tile][unit][32 rows] (32-bit element offsets: checked by the host)
      }
    };
    mfma_forward<O, NOUT>(R, b0, b1, bh, X, h, h1, h2, dh,
        [&](int i0, const float (&x)[4]) {
          if (!ACTOR && st) {
#pragma unroll
            for (int t = 0; t < 4; ++t) if (i0 + t < O) xt0[(unsigned)((tile * O + i0 + t) * 32 + c)] = x[t];
          }
        },
        [&](int layer, f32x16 (&a)[2]) { store_units(layer == 1 ? xt1 : xt2, a); });
    // head gradient in the head's D layout: register v <-> unit 8 (v >> 2) + 4 h + (v & 3)
    f32x16 gout;
#pragma unroll
    for (int v = 0; v < 16; ++v) gout[v] = 0.f;
    if constexpr (!ACTOR) {
      const float v_ = dh[0], ret = sc_ret, vp = sc_vp;
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
      if (mine) { if (st) gh[row] = gout[0]; loss_acc += vl; }
    } else {
      float z[16], e[16], lp = 0.f;
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int a = unit_of(0, v, h);
        z[v] = 0.f; e[v] = 0.f;
        if ((8 * (v >> 2) + (v & 3)) < A && a < A) {
          const float ls = logstd[a];
          e[v] = expf(-ls);
          z[v] = (sc_act[v] - dh[v]) * e[v];
          lp += -0.5f * z[v] * z[v] - ls - HALF_LOG_2PI;
        }
      }
      lp += __shfl_xor(lp, 32, 64);
      const float ratio = expf(lp - sc_olp), adv = sc_adv;                                  // ppo.py:54-59
      const float s1 = ratio * adv, s2 = fminf(fmaxf(ratio, 1.0f - B.clip), 1.0f + B.clip) * adv;
      const float inside = (ratio >= 1.0f - B.clip && ratio <= 1.0f + B.clip) ? 1.0f : 0.0f;
      const float wsel = s1 < s2 ? 1.0f : (s1 > s2 ? inside : 0.5f * (1.0f + inside));
      const float glp = on ? -adv * ratio * wsel * inv_m : 0.f;
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int a = unit_of(0, v, h);
        if ((8 * (v >> 2) + (v & 3)) < A && a < A) {
          gout[v] = glp * z[v] * e[v];                               // d logp / d mean_a = z / sigma
          if (st) gh[(unsigned)((tile * A + a) * 32 + c)] = gout[v];
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
    for (int q = 0; q < NR::QA; ++q) {
      const float4 wa = R.AHT(0, q), wb = R.AHT(1, q);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        d2[0] = MFMA32(comp4(wa, t), gout[4 * q + t], d2[0]);
        d2[1] = MFMA32(comp4(wb, t), gout[4 * q + t], d2[1]);
      }
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int v = 0; v < 16; ++v) d2[mt][v] *= 1.0f - h2[mt][v] * h2[mt][v];
    store_units(g2, d2);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        const float4 wa = R.A1T(0, 4 * mt + qq), wb = R.A1T(1, 4 * mt + qq);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          d1[0] = MFMA32(comp4(wa, t), d2[mt][4 * qq + t], d1[0]);
          d1[1] = MFMA32(comp4(wb, t), d2[mt][4 * qq + t], d1[1]);
        }
      }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int v = 0; v < 16; ++v) d1[mt][v] *= 1.0f - h1[mt][v] * h1[mt][v];
    store_units(g1, d1);
  }
  // this wavefront's row of partials: sums over its 32 rows
  float* P_ = partials + (size_t)(blockIdx.x * 4 + wave) * (3 + A);
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

template <int O, int A> constexpr int stage1_lds_bytes() { return NetLds<O, A>::BYTES > NetLds<O, 1>::BYTES ? NetLds<O, A>::BYTES : NetLds<O, 1>::BYTES; }
template <int O, int A>
__global__ void __launch_bounds__(256, 2)
ppo_grad_stage1_mfma_kernel(CRITIC_ARGS, ACTOR_ARGS, const solorl_ppo_batch B, const solorl_ppo_stage1 W) {
  extern __shared__ float4 stage1_image[];         // stage1_lds_bytes<O, A>() of dynamic LDS: 64-68 KB, two workgroups per CU
  if (blockIdx.y == 0) grad_net_mfma<O, A, false>(cw0, cb0, cw1, cb1, cwh, cbh, nullptr, B, W.xt0, W.c_xt1, W.c_xt2, W.c_g1, W.c_g2, W.c_gh, W.partials, stage1_image);
  else grad_net_mfma<O, A, true>(aw0, ab0, aw1, ab1, awh, abh, logstd, B, W.xt0, W.a_xt1, W.a_xt2, W.a_g1, W.a_g2, W.a_gh, W.partials, stage1_image);
}

// ---------------------------------------------------------------- stage 2 on MFMA: d W = G . X^T with the ROW index as K
// One wavefront per (layer, chunk of rows) computes ALL 32 x 32 tiles of the layer's product (2 x 3 for a hidden layer: 64 units x
// 64 / 76 inputs) from ONE pass over the chunk's rows of G and X: every element of stage 1's arrays is read once.  (Round 3 ran one
// wavefront per tile: G was read three times and X twice, 200 MB per mini-batch step at 4.6 TB/s = the kernel's 43 us.)
// Lane (c, h) reads rows of ITS unit's row of G (A operand, per unit tile) and of X (B operand, per input tile), each MFMA step
// contracts one row from either lane half (stage2_tiles has the row assignment).  D register v of lane (c, h) of
// tile (mt, nt) is d W[unit 32 mt + 8 (v >> 2) + 4 h + (v & 3)][input 32 nt + c]: stores are contiguous along the input index.
// The rows are cut into chunks, < 1024 wavefronts in all (stage2_plan); stage 3 adds the chunks' partial products in a fixed order.
// One wavefront per SIMD at most (stage 2 holds 250-350 registers: occupancy 1): with 1026 wavefronts on 1024 SIMDs two of them waited
// for a SIMD to come free and the kernel took two wavefront times (30 us) instead of one.
constexpr int STAGE2_WAVES = 960;

template <int NTM, int NTN>
__device__ __forceinline__ void stage2_tiles(const LayerDesc& L, int m, int r0, int r1, float* __restrict__ out) {
  const int l = threadIdx.x, c = l & 31, h = l >> 5;
  const float* ga[NTM];
  const float* xb[NTN];
  bool va[NTM], vb[NTN];
#pragma unroll
  for (int mt = 0; mt < NTM; ++mt) { va[mt] = 32 * mt + c < L.U; ga[mt] = L.g + (va[mt] ? 32 * mt + c : 0) * 32 + 4 * h; }
#pragma unroll
  for (int nt = 0; nt < NTN; ++nt) { vb[nt] = 32 * nt + c < L.K1 - 1; xb[nt] = L.x + (vb[nt] ? 32 * nt + c : 0) * 32 + 4 * h; }
  const unsigned gstep = (unsigned)L.U * 32u, xstep = (unsigned)(L.K1 - 1) * 32u;       // elements per tile of G / X
  f32x16 acc[NTM][NTN];
#pragma unroll
  for (int mt = 0; mt < NTM; ++mt)
#pragma unroll
    for (int nt = 0; nt < NTN; ++nt)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[mt][nt][v] = 0.f;
  float bsum[NTM];                          // bias gradient = the row sum of G: the lane adds up ITS unit's operands (column K1 - 1 of the product, without a tile for it)
#pragma unroll
  for (int mt = 0; mt < NTM; ++mt) bsum[mt] = 0.f;
  // Stage 1 writes its arrays [tile of 32 rows][unit][32 rows]: a wavefront's tile is one contiguous block per array (8 KB for 64 units),
  // and so is what a stage-2 wavefront reads.  An iteration contracts HALF a tile (16 rows): lane (c, h) reads rows 4 h .. 4 h + 3 and
  // 8 + 4 h .. 8 + 4 h + 3 of its unit (two float4); MFMA step s pairs a row of lane half 0 with one of half 1 -- any pairing is a valid
  // order of the sum.  Loads are UNCONDITIONAL: units past U / K1 read unit 0 and their results are never stored; the prefetch past the
  // chunk's end re-reads its last rows.
  // Measured: with the MFMAs taken out the kernel takes 22 of its 27 us, whatever the layout ([unit][m] arrays, 64 consecutive bytes per
  // lane, this blocked one): it moves 80 MB at ~3.7 TB/s and that is its bound; what the variants differ in is how finely the pipeline
  // below is cut (whole-tile iterations, one ahead: 31 us).
  constexpr int NJ = 2;
  struct Ops { float4 a[NTM][NJ], b[NTN][NJ]; };
  auto load = [&](int r) {          // r: half-tile index (16 rows)
    Ops o;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
#pragma unroll
      for (int mt = 0; mt < NTM; ++mt) o.a[mt][j] = *reinterpret_cast<const float4*>(ga[mt] + (unsigned)(r >> 1) * gstep + 16 * (r & 1) + 8 * j);
#pragma unroll
      for (int nt = 0; nt < NTN; ++nt) o.b[nt][j] = *reinterpret_cast<const float4*>(xb[nt] + (unsigned)(r >> 1) * xstep + 16 * (r & 1) + 8 * j);
    }
    return o;
  };
  // "pin": a compiler barrier that redefines the operands about to be used -- the loads written above it are ISSUED before it, the MFMAs
  // that consume the operands come after it.  (Left alone the compiler sinks the prefetch in front of its uses: a full memory round
  // trip per iteration with nothing in flight.)
  auto pin4 = [](float4& f) { asm volatile("" : "+v"(f.x), "+v"(f.y), "+v"(f.z), "+v"(f.w) :: "memory"); };
  auto pin = [&](Ops& o) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
#pragma unroll
      for (int mt = 0; mt < NTM; ++mt) pin4(o.a[mt][j]);
#pragma unroll
      for (int nt = 0; nt < NTN; ++nt) pin4(o.b[nt][j]);
    }
  };
  auto products = [&](const Ops& o) {
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int mt = 0; mt < NTM; ++mt) {
        bsum[mt] += (o.a[mt][j].x + o.a[mt][j].y) + (o.a[mt][j].z + o.a[mt][j].w);
#pragma unroll
        for (int nt = 0; nt < NTN; ++nt) {
          acc[mt][nt] = MFMA32(o.a[mt][j].x, o.b[nt][j].x, acc[mt][nt]); acc[mt][nt] = MFMA32(o.a[mt][j].y, o.b[nt][j].y, acc[mt][nt]);
          acc[mt][nt] = MFMA32(o.a[mt][j].z, o.b[nt][j].z, acc[mt][nt]); acc[mt][nt] = MFMA32(o.a[mt][j].w, o.b[nt][j].w, acc[mt][nt]);
        }
      }
  };
  int r = r0 / 16;                                                // half tiles r0 / 16 .. r1 / 16 - 1
  const int rlast = r1 / 16 - 1, nit = (r1 - r0) / 16;
  // A ring of DEPTH + 1 operand sets (static indices: registers, no copies): the loads run DEPTH iterations ahead of their use.  An
  // iteration's MFMAs take 0.43 us (a head: two tiles) to 1.28 us (six tiles) against 2-3 us of loaded-memory latency, so the depth goes
  // with the tile count: 2 for six tiles, 3 for four, 6 for two.
  constexpr int DEPTH = 12 / (NTM * NTN) < 2 ? 2 : (12 / (NTM * NTN) > 6 ? 6 : 12 / (NTM * NTN)), NS = DEPTH + 1;
  Ops o[NS];
  unrolled<DEPTH>([&](auto kc) { constexpr int k = decltype(kc)::value; o[k] = load(min(r + k, rlast)); });
  int it = 0;
#pragma unroll 1
  for (; it + NS <= nit; it += NS, r += NS) {
    unrolled<NS>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      o[(k + DEPTH) % NS] = load(min(r + k + DEPTH, rlast));
      pin(o[k]); products(o[k]);
    });
  }
  unrolled<DEPTH>([&](auto kc) { constexpr int k = decltype(kc)::value; if (it + k < nit) { pin(o[k]); products(o[k]); } });
#pragma unroll
  for (int mt = 0; mt < NTM; ++mt)
#pragma unroll
    for (int nt = 0; nt < NTN; ++nt)
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int u = 32 * mt + 8 * (v >> 2) + 4 * h + (v & 3);
        if (u < L.U && vb[nt]) out[u * L.K1 + 32 * nt + c] = acc[mt][nt][v];
      }
#pragma unroll
  for (int mt = 0; mt < NTM; ++mt) {
    const float b = bsum[mt] + __shfl_xor(bsum[mt], 32, 64);          // the unit's other rows were summed by the other lane half
    if (h == 0 && va[mt]) out[(32 * mt + c) * L.K1 + L.K1 - 1] = b;
  }
}

__global__ void __launch_bounds__(64) ppo_grad_stage2_mfma_kernel(const Stage2Args S) {
  LayerDesc L = S.L[0];                  // (wave-uniform selects: a dynamic index into the argument struct would be served from scratch)
#pragma unroll
  for (int i = 1; i < 6; ++i) if ((int)blockIdx.x >= S.L[i].wave0) L = S.L[i];
  const int chunk = blockIdx.x - L.wave0;
  const int ntn = (L.K1 - 1 + 31) / 32, ntm = (L.U + 31) / 32;          // tiles over the K1 - 1 inputs; the bias column is a row sum
  const int r0 = chunk * L.mchunk, r1 = min(r0 + L.mchunk, S.m);
  float* out = S.scratch + L.sbase + (size_t)chunk * L.U * L.K1;
  if (ntm == 2) {
    if (ntn == 3) stage2_tiles<2, 3>(L, S.m, r0, r1, out); else if (ntn == 2) stage2_tiles<2, 2>(L, S.m, r0, r1, out); else stage2_tiles<2, 1>(L, S.m, r0, r1, out);
  } else {
    if (ntn == 3) stage2_tiles<1, 3>(L, S.m, r0, r1, out); else if (ntn == 2) stage2_tiles<1, 2>(L, S.m, r0, r1, out); else stage2_tiles<1, 1>(L, S.m, r0, r1, out);
  }
}

// The chunking of a mini-batch of m rows: every layer gets wavefronts in proportion to its tile count (6 : 4 : 2 for obs -> 64 -> 64 ->
// head at 76 inputs), at most STAGE2_WAVES in all, so that all wavefronts carry the same number of MFMAs.  Chunks are multiples of 32 rows.
struct Stage2Plan { int nchunks[6], mchunk[6], wave0[6], sbase[6], off[6], nwaves, total, scratch; };
inline Stage2Plan stage2_plan(int O, int A, int m) {
  const int U[6] = {H, H, 1, H, H, A}, K1[6] = {O + 1, H + 1, H + 1, O + 1, H + 1, H + 1};
  Stage2Plan P{};
  int tiles[6], all = 0;
  for (int i = 0; i < 6; ++i) { tiles[i] = ((U[i] + 31) / 32) * ((K1[i] - 1 + 31) / 32); all += tiles[i]; }
  for (int i = 0; i < 6; ++i) {
    const int want = STAGE2_WAVES * tiles[i] / all > 0 ? STAGE2_WAVES * tiles[i] / all : 1;       // wavefronts for this layer (rounded DOWN, chunks rounded up)
    const int mc = (((m + want - 1) / want + 31) / 32) * 32;
    P.mchunk[i] = mc; P.nchunks[i] = (m + mc - 1) / mc;
    P.wave0[i] = P.nwaves; P.sbase[i] = P.scratch; P.off[i] = P.total;
    P.nwaves += P.nchunks[i]; P.scratch += P.nchunks[i] * U[i] * K1[i]; P.total += U[i] * K1[i];
  }
  return P;
}

// ---------------------------------------------------------------- norm clip + Adam over all 13 parameter tensors, one workgroup
// nn.utils.clip_grad_norm_ (agents/ppo/ppo.py:75-76) and torch.optim.Adam's update (:32, :77; no amsgrad) for ~20 k parameters are
// nine small launches in PyTorch (norm, three scalar ops, scale, the multi-tensor Adam, the mini-batch cursor ...); here one
// 1024-thread workgroup reads every gradient twice.
struct AdamSeg { float* p; const float* g; int n, off; };
struct AdamArgs { AdamSeg seg[13]; int total; float* m; float* v; float* step; const float* lr; float b1, b2, eps, wd, max_norm, gscale;
                  long long* offset; long long inc; };


// The tensor sizes are compile-time (dispatch_dims), so the walk over the 13 tensors is unrolled with STATIC indices into the argument
// struct (round 3's attempt indexed it dynamically and the compiler served it from scratch: 31 us; the tensor-by-tensor loops it
// fell back to waited for one L2 round trip per tensor and pass: 22 us).  A thread owns element tid + 1024 j of tensor s for every
// (s, j) -- 27-29 slots: every gradient is loaded ONCE, all loads of a phase are in flight together, the gradients stay in
// registers between the norm and the update.
template <int O, int A> struct AdamDims {
  static constexpr int n(int s) { constexpr int N[13] = {H * O, H, H * H, H, H, 1, H * O, H, H * H, H, A * H, A, A}; return N[s]; }
  static constexpr int cnt(int s) { return (n(s) + 1023) / 1024; }
  static constexpr int base(int s) { int b = 0; for (int i = 0; i < s; ++i) b += cnt(i); return b; }
  static constexpr int SLOTS = base(13);
  static constexpr int seg(int k) { int s = 0; while (base(s + 1) <= k) ++s; return s; }     // slot k = element tid + 1024 j(k) of tensor seg(k)
  static constexpr int j(int k) { return k - base(seg(k)); }
  static constexpr int GROUP = 3, NGROUPS = (SLOTS + GROUP - 1) / GROUP, AHEAD = 2;     // update phase: groups of three slots, two groups in flight ahead of the
                                                                                        // one being updated (128 registers at 1024 threads; ONE group ahead: a memory round trip per group, 12 us)
};
template <int O, int A>
__global__ void __launch_bounds__(1024) ppo_clip_adam_kernel(const AdamArgs K) {
  using D = AdamDims<O, A>;
  __shared__ float red[16];
  __shared__ float coef_s, step_size_s, inv_sqrt_bc2_s;
  const unsigned tid = threadIdx.x;        // (unsigned element offsets: scalar base + 32-bit vector offset addressing, no 64-bit address registers)
  float g[D::SLOTS], ss = 0.f;
  unrolled<D::SLOTS>([&](auto kc) {
    constexpr int k = decltype(kc)::value, s = D::seg(k), jj = D::j(k), nn = D::n(s);      // (constexpr variables: forced constant evaluation)
    const unsigned i = tid + 1024u * jj;
    g[k] = K.seg[s].g[min(i, (unsigned)(nn - 1))];          // (clamped and UNCONDITIONAL: all loads are issued here ...
  });
  asm volatile("" ::: "memory");
  unrolled<D::SLOTS>([&](auto kc) {
    constexpr int k = decltype(kc)::value, s = D::seg(k), jj = D::j(k), nn = D::n(s);
    asm volatile("" : "+v"(g[k]));            //  ... and pinned here: left alone the compiler sinks each load into the branch of its select
    g[k] = tid + 1024u * jj < nn ? g[k] * K.gscale : 0.f;     //  and waits for it there, one memory round trip per partly filled slot: 12 us)
  });                                         // gscale: 1 / world after a SUMMED gradient all-reduce (the mean's scale, fused)
  struct PMV { float p[D::GROUP], m[D::GROUP], v[D::GROUP]; };
  auto load_group = [&](auto gc) {
    constexpr int k0 = decltype(gc)::value * D::GROUP;
    PMV r;
    unrolled<D::GROUP>([&](auto qc) {
      constexpr int q = decltype(qc)::value, k = k0 + q;
      if constexpr (k < D::SLOTS) {
        constexpr int s = D::seg(k), jj = D::j(k), nn = D::n(s);
        const unsigned i = tid + 1024u * jj;
        const unsigned ic = min(i, (unsigned)(nn - 1));
        r.p[q] = K.seg[s].p[ic]; r.m[q] = K.m[(unsigned)K.seg[s].off + ic]; r.v[q] = K.v[(unsigned)K.seg[s].off + ic];
      }
    });
    return r;
  };
  PMV ring[D::AHEAD + 1];
  unrolled<D::AHEAD>([&](auto gc) { if constexpr (decltype(gc)::value < D::NGROUPS) ring[decltype(gc)::value] = load_group(gc); });   // (do not depend on the norm: in flight across the reduction)
#pragma unroll
  for (int k = 0; k < D::SLOTS; ++k) ss = fmaf(g[k], g[k], ss);
  ss = wave_sum(ss);
  if ((tid & 63) == 0) red[tid >> 6] = ss;
  const float step = *K.step + 1.0f, lr = *K.lr;
  __syncthreads();
  if (tid == 0) {
    float t = 0.f;
    for (int w = 0; w < 16; ++w) t += red[w];
    coef_s = K.max_norm > 0.f ? fminf(1.0f, K.max_norm / (sqrtf(t) + 1e-6f)) : 1.0f;      // clip_coef clamped to 1
    // (one thread: two powf are ~300 instructions, and the whole launch is one CU's instruction issue: 16 wavefronts on 4 SIMDs)
    const float bc1 = 1.0f - powf(K.b1, step), bc2 = 1.0f - powf(K.b2, step);
    step_size_s = lr / bc1; inv_sqrt_bc2_s = 1.0f / sqrtf(bc2);
  }
  __syncthreads();
  const float coef = coef_s, step_size = step_size_s, inv_sqrt_bc2 = inv_sqrt_bc2_s;
  unrolled<D::NGROUPS>([&](auto gc) {
    constexpr int gi = decltype(gc)::value, k0 = gi * D::GROUP;
    if constexpr (gi + D::AHEAD < D::NGROUPS) ring[(gi + D::AHEAD) % (D::AHEAD + 1)] = load_group(std::integral_constant<int, gi + D::AHEAD>{});
    PMV& cur = ring[gi % (D::AHEAD + 1)];
    asm volatile("" ::: "memory");
#pragma unroll
    for (int q = 0; q < D::GROUP; ++q) asm volatile("" : "+v"(cur.p[q]), "+v"(cur.m[q]), "+v"(cur.v[q]));       // (as above: the loads stay where they were issued)
    unrolled<D::GROUP>([&](auto qc) {
      constexpr int q = decltype(qc)::value, k = k0 + q;
      if constexpr (k < D::SLOTS) {
        constexpr int s = D::seg(k), jj = D::j(k), nn = D::n(s);
        const unsigned i = tid + 1024u * jj;
        if (i < nn) {
          float gg = g[k] * coef;
          if (K.wd != 0.f) gg = fmaf(K.wd, cur.p[q], gg);
          const float mn = cur.m[q] + (1.0f - K.b1) * (gg - cur.m[q]);                 // exp_avg.lerp_(grad, 1 - beta1)
          const float vn = K.b2 * cur.v[q] + (1.0f - K.b2) * gg * gg;
          K.m[(unsigned)K.seg[s].off + i] = mn; K.v[(unsigned)K.seg[s].off + i] = vn;
          // v_sqrt_f32 / v_rcp_f32 (1 ulp each) instead of the correctly rounded sqrt and division (~25 instructions per element)
          K.seg[s].p[i] = cur.p[q] - step_size * (mn * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(vn) * inv_sqrt_bc2 + K.eps));
        }
      }
    });
  });
  // (every thread read *K.step before the first barrier)
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
  if (batch->m % 32 != 0) return solorl_fail_(SOLORL_ERR_INVALID, "solorl_ppo_grad_stage1: m must be a multiple of 32 rows (the work arrays are written in tiles of 32)");
  if ((long long)batch->m * (p->obs_dim > H ? p->obs_dim : H) >= (1LL << 31))
    return solorl_fail_(SOLORL_ERR_INVALID, "solorl_ppo_grad_stage1: mini-batch too large (the [unit][row] arrays are indexed with 32 bits)");
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
    constexpr int LDS = stage1_lds_bytes<O, A>();
    static bool lds_ok[64] = {};             // (more than the 64 KB a kernel gets by default: raised once per instantiation and device)
    if (!lds_ok[device_id & 63]) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&ppo_grad_stage1_mfma_kernel<O, A>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
        return solorl_fail_(SOLORL_ERR_HIP, "ppo_grad_stage1_mfma_kernel: cannot raise the dynamic LDS limit");
      lds_ok[device_id & 63] = true;
    }
    hipLaunchKernelGGL((ppo_grad_stage1_mfma_kernel<O, A>), dim3((B.m + 127) / 128, 2), dim3(256), LDS, (hipStream_t)stream, CRITIC_PASS, ACTOR_PASS, B, W);
    return hipGetLastError() == hipSuccess ? 0 : solorl_fail_(SOLORL_ERR_HIP, "ppo_grad_stage1_mfma_kernel launch");
  });
}

int solorl_ppo_grad_count(int obs_dim, int act_dim) { return 2 * H * (obs_dim + 1) + 2 * H * (H + 1) + (H + 1) + act_dim * (H + 1); }

int solorl_ppo_scratch_count(int obs_dim, int act_dim, int m) { return m < 1 ? 0 : stage2_plan(obs_dim, act_dim, m).scratch; }

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
  const Stage2Plan plan = stage2_plan(O, A, m);
  S.m = m; S.scratch = out->scratch; S.total = plan.total; S.nwaves = plan.nwaves; S.reserved = 0;
  for (int i = 0; i < 6; ++i) {
    S.L[i] = L[i]; S.L[i].off = plan.off[i]; S.L[i].nchunks = plan.nchunks[i]; S.L[i].mchunk = plan.mchunk[i]; S.L[i].wave0 = plan.wave0[i];
    S.L[i].sbase = plan.sbase[i];
  }
  T.partials = work->partials; T.nwaves = (m + 31) / 32; T.A = A; T.logstd = p->logstd; T.logstd_grad = out->logstd;
  T.loss_sums = out->loss_sums; T.logstd_sum = out->logstd_sum; T.entropy_coef = out->entropy_coef;
  hipLaunchKernelGGL(ppo_grad_stage2_mfma_kernel, dim3(S.nwaves), dim3(64), 0, (hipStream_t)stream, S);
  hipLaunchKernelGGL(ppo_grad_stage3_kernel, dim3((S.total + 15) / 16 + 3 + A), dim3(256), 0, (hipStream_t)stream, T);
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
  return dispatch_dims(O, A, [&](auto oc, auto ac) {
    hipLaunchKernelGGL((ppo_clip_adam_kernel<decltype(oc)::value, decltype(ac)::value>), dim3(1), dim3(1024), 0, (hipStream_t)stream, K);
    return hipGetLastError() == hipSuccess ? 0 : solorl_fail_(SOLORL_ERR_HIP, "ppo_clip_adam_kernel launch");
  });
}

}  // extern "C"
