// Diagnostic microbenchmarks for the primitives the rollout kernel leans on (not part of the product).
// Build: hipcc -O3 --offload-arch=gfx950 tools/microbench.hip -o gpurun_out/microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define DEV __device__ __forceinline__
DEV float readlane_f(float x, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), lane)); }
template <int CTRL> DEV float dpp_f(float x) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true)); }
DEV float wave_sum(float x) {
  x += dpp_f<0xB1>(x); x += dpp_f<0x4E>(x); x += dpp_f<0x141>(x); x += dpp_f<0x140>(x);
  return (readlane_f(x, 0) + readlane_f(x, 16)) + (readlane_f(x, 32) + readlane_f(x, 48));
}
extern __shared__ float lds[];

template <int MODE> __global__ void __launch_bounds__(64) bench(float *out, unsigned long long *cyc, int iters) {
  const int lane = threadIdx.x;
  float a[20];
#pragma unroll
  for (int k = 0; k < 20; ++k) a[k] = 1.0f + 0.01f * (lane * 20 + k);
  for (int i = lane; i < 4096; i += 64) lds[i] = 0.5f + 1e-3f * i;
  __syncthreads();
  float acc = lane;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {  // dependent fma chain x100
#pragma unroll
      for (int k = 0; k < 100; ++k) acc = __builtin_fmaf(acc, 1.0001f, 0.5f);
    } else if (MODE == 1) {  // 100 independent (readlane, fma) pairs, like one Cholesky column sweep
#pragma unroll
      for (int r = 0; r < 5; ++r)
#pragma unroll
        for (int k = 0; k < 20; ++k) a[k] = __builtin_fmaf(-acc, readlane_f(acc, k + r), a[k]);
      acc += a[3];
    } else if (MODE == 2) {  // 10 wave_sums, dependent
#pragma unroll
      for (int k = 0; k < 10; ++k) acc = wave_sum(acc) * 0.01f + lane;
    } else if (MODE == 3) {  // 100 dependent LDS reads (pointer chase through values)
#pragma unroll
      for (int k = 0; k < 100; ++k) acc = lds[(int)acc & 4095] + lane;
    } else if (MODE == 4) {  // 100 independent LDS reads + fma
      float s = 0;
#pragma unroll
      for (int k = 0; k < 100; ++k) s += lds[lane * 23 + k];
      acc += s;
    } else if (MODE == 5) {  // 100 ds_bpermute dependent
#pragma unroll
      for (int k = 0; k < 100; ++k) acc = __shfl(acc, (lane + k) & 63) + 1.0f;
    } else if (MODE == 7) {  // 100 independent ds_bpermute + fma (pivot-row broadcast through the crossbar)
#pragma unroll
      for (int r = 0; r < 5; ++r)
#pragma unroll
        for (int k = 0; k < 20; ++k) a[k] = __builtin_fmaf(-acc, __int_as_float(__builtin_amdgcn_ds_bpermute(((lane & 32) + k + r) * 4, __float_as_int(a[k]))), a[k]);
      acc += a[3];
    } else if (MODE == 8) {  // 100 independent (2 readlane + select) + fma
#pragma unroll
      for (int r = 0; r < 5; ++r)
#pragma unroll
        for (int k = 0; k < 20; ++k) { float lo = readlane_f(a[k], k + r), hi = readlane_f(a[k], k + r + 32); a[k] = __builtin_fmaf(-acc, (lane & 32) ? hi : lo, a[k]); }
      acc += a[3];
    } else if (MODE == 9) {  // 100 independent ds_swizzle + add
#pragma unroll
      for (int r = 0; r < 5; ++r)
#pragma unroll
        for (int k = 0; k < 20; ++k) a[k] += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(a[k]), 0x401F));
      acc += a[3];
    } else if (MODE == 6) {  // 20 x (readlane pivot -> rsq -> mul) dependent chain, the Cholesky pivot path
#pragma unroll
      for (int k = 0; k < 20; ++k) {
        float p = fmaxf(readlane_f(a[k], k), 1e-15f);
        float inv = __builtin_amdgcn_rsqf(p);
        inv = inv * __builtin_fmaf(-0.5f * p * inv, inv, 1.5f);
        a[k] *= inv;
        if (k < 19) a[k + 1] = __builtin_fmaf(-a[k], readlane_f(a[k], k + 1), a[k + 1]);
      }
      acc += a[19];
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 64 + lane] = acc + a[0] + a[7];
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE> void run(const char *name, int per_iter, int blocks_per_cu) {
  const int nblk = 256 * blocks_per_cu, iters = 200;
  float *out; unsigned long long *cyc;
  hipMalloc(&out, nblk * 64 * sizeof(float)); hipMalloc(&cyc, nblk * sizeof(unsigned long long));
  hipLaunchKernelGGL(bench<MODE>, dim3(nblk), dim3(64), 16384, 0, out, cyc, iters);
  hipLaunchKernelGGL(bench<MODE>, dim3(nblk), dim3(64), 16384, 0, out, cyc, iters);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(nblk);
  hipMemcpy(h.data(), cyc, nblk * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  double s = 0; for (auto v : h) s += v;
  printf("%-44s waves/CU=%2d  %.1f cycles per op (%d ops/iter)\n", name, blocks_per_cu, s / nblk / iters / per_iter, per_iter);
  hipFree(out); hipFree(cyc);
}
int main() {
  for (int w : {1, 8, 16}) {
    run<0>("dependent v_fma chain", 100, w);
    run<1>("independent readlane+fma pairs", 100, w);
    run<2>("wave_sum (4 dpp + 4 readlane + 3 add)", 10, w);
    run<3>("dependent LDS read", 100, w);
    run<4>("independent LDS read + add", 100, w);
    run<5>("dependent ds_bpermute (__shfl)", 100, w);
    run<6>("chol pivot chain (readlane,rsq,NR,mul,update)", 20, w);
    run<7>("independent ds_bpermute + fma", 100, w);
    run<8>("independent 2x readlane + select + fma", 100, w);
    run<9>("independent ds_swizzle + add", 100, w);
  }
  return 0;
}
