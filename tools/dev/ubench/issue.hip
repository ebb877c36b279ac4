// Lone-wave issue-cost microbenchmark (dev tool): cycles per instruction for dependent chains of the
// instruction kinds the team PGS sweep is made of.  One wave per workgroup, WGS workgroups.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
template <int KIND> __global__ void __launch_bounds__(64) bench(float* out, long long* cyc, int iters) {
  float a = threadIdx.x * 0.001f, b = 1.0001f, c = 0.5f, d = a + 1.f;
  float2 p = make_float2(a, b), q = make_float2(b, c);
  long long t0 = clock64();
  for (int i = 0; i < iters; i++) {
    if (KIND == 0) { REP64(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));) }
    if (KIND == 1) { REP64(asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p) : "v"(q));) }
    if (KIND == 2) { REP64(asm volatile("s_nop 1\n v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a));) }
    if (KIND == 3) { REP64(asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3" : "+v"(a), "+v"(d) : "v"(b), "v"(c));) }   // 2 independent chains
    if (KIND == 4) { REP64(asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));) }
    if (KIND == 5) { REP64(asm volatile("v_fma_f32 %0, %0, %1, %2\n s_nop 0" : "+v"(a) : "v"(b), "v"(c));) }
    if (KIND == 6) { REP64(asm volatile("s_cmp_eq_u32 %1, 12345\n s_cbranch_scc1 1f\n v_fma_f32 %0, %0, %2, %3\n 1:" : "+v"(a) : "s"(iters), "v"(b), "v"(c) : "scc");) }   // not-taken branch
    if (KIND == 7) { REP64(asm volatile("s_cmp_lg_u32 %1, 12345\n s_cbranch_scc1 1f\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %0, %0, %2, %3\n 1:\n v_fma_f32 %0, %0, %2, %3" : "+v"(a) : "s"(iters), "v"(b), "v"(c) : "scc");) }   // taken branch over 2 instr
    if (KIND == 8) { REP64(asm volatile("v_mov_b32_dpp %1, %0 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fma_f32 %0, %1, %2, %0" : "+v"(a), "+v"(d) : "v"(b));) }
    if (KIND == 9) { REP64(asm volatile("v_pk_mul_f32 %0, %0, %2\n v_add_f32 %1, %1, %1" : "+v"(p), "+v"(a) : "v"(q) );) }
  }
  long long t1 = clock64();
  out[blockIdx.x * 64 + threadIdx.x] = a + p.x + p.y + d;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main(int argc, char** argv) {
  int wgs = argc > 1 ? atoi(argv[1]) : 1024, iters = 200;
  float* out; long long* cyc; hipMalloc(&out, wgs * 64 * 4); hipMalloc(&cyc, wgs * 8);
  long long* h = (long long*)malloc(wgs * 8);
  const char* names[] = {"dep v_fma_f32", "dep v_pk_fma_f32", "dep s_nop1+v_add_dpp", "2 indep v_fma chains (per instr)", "dep v_med3", "dep v_fma + s_nop 0 (per pair)",
                         "not-taken sbranch + fma (per triple)", "taken sbranch skipping 2 + fma", "mov_dpp + dep fma (per pair)", "dep pk_mul + indep add (per pair)"};
  for (int kind = 0; kind < 10; kind++) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; rep++) {
      hipEventRecord(e0);
      switch (kind) {
#define C(K) case K: hipLaunchKernelGGL(bench<K>, dim3(wgs), dim3(64), 0, 0, out, cyc, iters); break;
        C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9)
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, cyc, wgs * 8, hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < wgs; i++) s += h[i];
    double per = s / wgs / (64.0 * iters);
    fflush(stdout); printf("%-40s clock64 %.2f per unit; wall %.3f ms -> %.2f ns per unit\n", names[kind], per, ms, ms * 1e6 / (64.0 * iters));
  }
  fflush(stdout);
  return 0;
}
