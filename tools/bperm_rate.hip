#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ float bperm_f(int addr, float x) { return __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(x))); }
template <int MODE> __global__ void __launch_bounds__(256) k(float *out, unsigned long long *cyc, int iters) {
  const int lane = threadIdx.x & 63;
  float a[18];
#pragma unroll
  for (int i = 0; i < 18; ++i) a[i] = 1.0f + 0.001f * (lane + i);
  float f = 0.001f * lane;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {  // 18 independent bpermute+fmac per round, rounds dependent through a[] (like one GJ pivot without the pivot chain)
#pragma unroll
      for (int i = 0; i < 18; ++i) a[i] = __builtin_fmaf(-f, bperm_f(4 * ((lane & 32) + (it & 31)), a[i]), a[i]);
    } else if (MODE == 1) {  // same, plus the factor depends on a broadcast of a[0] (pivot chain)
      const float piv = bperm_f(4 * ((lane & 32) + (it & 31)), a[0]);
      f = a[1] * __builtin_amdgcn_rcpf(fmaxf(piv, 1e-15f));
#pragma unroll
      for (int i = 0; i < 18; ++i) a[i] = __builtin_fmaf(-f, bperm_f(4 * ((lane & 32) + (it & 31)), a[i]), a[i]);
    } else if (MODE == 3) {  // ds_swizzle broadcast of a fixed lane within each 32-lane half (no address register)
#pragma unroll
      for (int i = 0; i < 18; ++i) a[i] = __builtin_fmaf(-f, __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(a[i]), (7 << 5))), a[i]);
    } else if (MODE == 4) {  // v_readlane x2 + select
#pragma unroll
      for (int i = 0; i < 18; ++i) { const float lo = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a[i]), 7)), hi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a[i]), 39)); a[i] = __builtin_fmaf(-f, lane < 32 ? lo : hi, a[i]); }
    } else if (MODE == 2) {  // 18 independent LDS-free fmacs (VALU only)
#pragma unroll
      for (int i = 0; i < 18; ++i) a[i] = __builtin_fmaf(-f, a[(i + 1) % 18], a[i]);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
#pragma unroll
  for (int i = 0; i < 18; ++i) s += a[i];
  out[threadIdx.x + blockIdx.x * blockDim.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main() {
  float *out; unsigned long long *cyc, h;
  hipMalloc(&out, 4 * 256 * 4096); hipMalloc(&cyc, 8);
  const char *names[] = {"18 independent bpermute+fmac per round", "same with pivot chain (bperm->rcp->mul)", "18 fmacs", "18 ds_swizzle broadcast+fmac per round", "18 (2 readlane+select)+fmac per round"};
  for (int mode = 0; mode < 5; ++mode)
    for (int blocks : {1, 512}) {
      const int iters = 1000;
      if (mode == 0) k<0><<<blocks, 256>>>(out, cyc, iters);
      if (mode == 1) k<1><<<blocks, 256>>>(out, cyc, iters);
      if (mode == 2) k<2><<<blocks, 256>>>(out, cyc, iters);
      if (mode == 3) k<3><<<blocks, 256>>>(out, cyc, iters);
      if (mode == 4) k<4><<<blocks, 256>>>(out, cyc, iters);
      hipDeviceSynchronize();
      hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
      printf("%-44s blocks=%4d: %.1f cycles per round, %.2f per op\n", names[mode], blocks, (double)h / iters, (double)h / iters / 18);
    }
  return 0;
}
