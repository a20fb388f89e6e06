// Issue cost of packed fp32 (v_pk_fma_f32) against scalar v_fma_f32 on one wave: 8 independent accumulators, 256 instructions
// per timed block, s_memtime around it.  hipcc --offload-arch=gfx950 -O3 tools/probes/pk_rate.hip -o /tmp/pk_rate && /tmp/pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2v __attribute__((ext_vector_type(2)));
__global__ void probe(float *out, unsigned long long *cyc, float s) {
  float a[8];
  f2v p[8];
  for (int k = 0; k < 8; ++k) { a[k] = threadIdx.x + k; p[k] = f2v{(float)threadIdx.x, (float)k}; }
  const f2v s2 = {s, s * 0.5f};
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < 64; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int k = 0; k < 8; ++k) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[k]) : "v"(s));
  }
  __builtin_amdgcn_s_waitcnt(0);
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < 64; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int k = 0; k < 8; ++k) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[k]) : "v"(s2));
  }
  __builtin_amdgcn_s_waitcnt(0);
  unsigned long long t2 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < 64; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int k = 0; k < 8; ++k) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[k]) : "v"(s2));
  }
  __builtin_amdgcn_s_waitcnt(0);
  unsigned long long t3 = __builtin_amdgcn_s_memtime();
  float acc = 0;
  for (int k = 0; k < 8; ++k) acc += a[k] + p[k].x + p[k].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; cyc[2] = t3 - t2; }
}
int main() {
  float *out; unsigned long long *cyc, h[3];
  hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 64);
  for (int waves = 1; waves <= 2; ++waves) {
    probe<<<1, 64 * 4 * waves>>>(out, cyc, 1.0001f);  // 4 waves = one per SIMD; 8 = two per SIMD
    hipDeviceSynchronize();
    hipMemcpy(h, cyc, 24, hipMemcpyDeviceToHost);
    // s_memtime counts at 100 MHz on this part; report the ratio and raw ticks
    printf("%d wave(s)/SIMD: 2048 instr each: v_fma_f32 %llu ticks, v_pk_fma_f32 %llu ticks (x%.2f), v_pk_add_f32 %llu ticks (x%.2f)\n", waves, h[0], h[1], (double)h[1] / h[0], h[2], (double)h[2] / h[0]);
  }
  return 0;
}
