// Diagnostic: latency/throughput of v_mfma_f32_32x32x1_2b_f32 chains on gfx950 (one wave per SIMD unless noted).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v32f __attribute__((ext_vector_type(32)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x1f32((a), (b), (c), 0, 0, 0)
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
template <int MODE> __global__ void __launch_bounds__(256) k(float *out, unsigned long long *cyc, int iters) {
  const int lane = threadIdx.x & 63;
  v32f c = {}, w = {};
  v16f c16 = {};
  v4f c4 = {};
  float a = 1.0f + lane * 1e-3f, b = 0.5f;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (MODE == 0) c = MFMA(a, b, c);                                   // dependent accumulate chain
      if (MODE == 1) { c = MFMA(a, b, c); a = c[r & 3] * 1e-20f + 1.0f; }  // result -> 1 VALU -> operand
      if (MODE == 2) { c = MFMA(a, b, c); w = MFMA(a, b, w); a = c[r & 3] * 1e-20f + 1.0f; }  // + independent second MFMA
      if (MODE == 3) { c16 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c16, 0, 0, 0); a = c16[r & 3] * 1e-20f + 1.0f; }  // 1 block, K=2
      if (MODE == 4) { c4 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c4, 0, 0, 0); a = c4[r & 3] * 1e-20f + 1.0f; }      // 16 blocks of 4x4
      if (MODE == 5) { c16 = __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, c16, 0, 0, 0); a = c16[r & 3] * 1e-20f + 1.0f; }  // 4 blocks of 16x16
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = c[0] + w[1] + c16[2] + c4[3] + a;
  if (threadIdx.x == 0) *cyc = t1 - t0;
}
int main() {
  float *out; unsigned long long *cyc, h;
  hipMalloc(&out, 1024); hipMalloc(&cyc, 8);
  const char *names[] = {"32x32x1 2b dependent accumulate", "32x32x1 2b -> VALU -> operand", "same + independent 2nd mfma", "32x32x2 1b -> VALU -> operand", "4x4x1 16b -> VALU -> operand", "16x16x1 4b -> VALU -> operand"};
  for (int mode = 0; mode < 6; ++mode) {
    for (int waves = 1; waves <= 4; waves += 3) {
      const int iters = 100;
      switch (mode) {
        case 0: k<0><<<1, 64 * waves>>>(out, cyc, iters); break;
        case 1: k<1><<<1, 64 * waves>>>(out, cyc, iters); break;
        case 2: k<2><<<1, 64 * waves>>>(out, cyc, iters); break;
        case 3: k<3><<<1, 64 * waves>>>(out, cyc, iters); break;
        case 4: k<4><<<1, 64 * waves>>>(out, cyc, iters); break;
        case 5: k<5><<<1, 64 * waves>>>(out, cyc, iters); break;
      }
      hipDeviceSynchronize();
      hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
      printf("%-36s waves/block=%d: %.1f cycles per step\n", names[mode], waves, (double)h / (iters * 16));
    }
  }
  return 0;
}
