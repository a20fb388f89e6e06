// Diagnostic: per-wave issue cost of LDS read widths on gfx950, instruction choice pinned with inline asm.
// Each trip issues 8 independent reads, waits once, and folds the results; cycles per read instruction are reported for one
// wave per SIMD (1 block) and two waves per SIMD on every CU (512 blocks).
#include <hip/hip_runtime.h>
#include <cstdio>
extern __shared__ float lds[];
template <int MODE> __global__ void __launch_bounds__(256) k(float *out, unsigned long long *cyc, int iters, int stride_bytes) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 16000; i += blockDim.x) lds[i] = 1.0f + i * 1e-4f;
  __syncthreads();
  const unsigned addr = wave * 16000 + lane * stride_bytes;  // byte address
  float acc = 0.f;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    float r0, r1, r2, r3, r4, r5, r6, r7;
    if (MODE == 0) {
      asm volatile("ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:4\n ds_read_b32 %2, %8 offset:8\n ds_read_b32 %3, %8 offset:12\n"
                   "ds_read_b32 %4, %8 offset:16\n ds_read_b32 %5, %8 offset:20\n ds_read_b32 %6, %8 offset:24\n ds_read_b32 %7, %8 offset:28\n s_waitcnt lgkmcnt(0)"
                   : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7) : "v"(addr) : "memory");
      acc += ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    } else if (MODE == 1) {
      typedef float v2 __attribute__((ext_vector_type(2)));
      v2 a, b, c, d, e, f, g, h;
      asm volatile("ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:8\n ds_read_b64 %2, %8 offset:16\n ds_read_b64 %3, %8 offset:24\n"
                   "ds_read_b64 %4, %8 offset:32\n ds_read_b64 %5, %8 offset:40\n ds_read_b64 %6, %8 offset:48\n ds_read_b64 %7, %8 offset:56\n s_waitcnt lgkmcnt(0)"
                   : "=v"(a), "=v"(b), "=v"(c), "=v"(d), "=v"(e), "=v"(f), "=v"(g), "=v"(h) : "v"(addr) : "memory");
      acc += (a.x + b.y) + (c.x + d.y) + (e.x + f.y) + (g.x + h.y);
    } else if (MODE == 2) {
      typedef float v4 __attribute__((ext_vector_type(4)));
      v4 a, b, c, d, e, f, g, h;
      asm volatile("ds_read_b128 %0, %8\n ds_read_b128 %1, %8 offset:16\n ds_read_b128 %2, %8 offset:32\n ds_read_b128 %3, %8 offset:48\n"
                   "ds_read_b128 %4, %8 offset:64\n ds_read_b128 %5, %8 offset:80\n ds_read_b128 %6, %8 offset:96\n ds_read_b128 %7, %8 offset:112\n s_waitcnt lgkmcnt(0)"
                   : "=v"(a), "=v"(b), "=v"(c), "=v"(d), "=v"(e), "=v"(f), "=v"(g), "=v"(h) : "v"(addr) : "memory");
      acc += (a.x + b.y) + (c.z + d.w) + (e.x + f.y) + (g.z + h.w);
    } else if (MODE == 3) {
      typedef float v2 __attribute__((ext_vector_type(2)));
      v2 a, b, c, d, e, f, g, h;
      asm volatile("ds_read2_b32 %0, %8 offset0:0 offset1:3\n ds_read2_b32 %1, %8 offset0:1 offset1:5\n ds_read2_b32 %2, %8 offset0:2 offset1:7\n ds_read2_b32 %3, %8 offset0:4 offset1:9\n"
                   "ds_read2_b32 %4, %8 offset0:6 offset1:11\n ds_read2_b32 %5, %8 offset0:8 offset1:13\n ds_read2_b32 %6, %8 offset0:10 offset1:15\n ds_read2_b32 %7, %8 offset0:12 offset1:17\n s_waitcnt lgkmcnt(0)"
                   : "=v"(a), "=v"(b), "=v"(c), "=v"(d), "=v"(e), "=v"(f), "=v"(g), "=v"(h) : "v"(addr) : "memory");
      acc += (a.x + b.y) + (c.x + d.y) + (e.x + f.y) + (g.x + h.y);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x + blockIdx.x * blockDim.x] = acc;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int MODE> double run(int blocks, int stride_floats) {
  static float *out = nullptr; static unsigned long long *cyc = nullptr;
  if (!out) { hipMalloc(&out, 4 * 256 * 1024); hipMalloc(&cyc, 8); }
  hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  const int iters = 4000;
  k<MODE><<<blocks, 256, 65000>>>(out, cyc, iters, 4 * stride_floats);
  hipDeviceSynchronize();
  unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  return (double)h / iters / 8;
}
int main() {
  for (int st : {1, 17, 29, 2, 4}) printf("ds_read_b32   lane stride %3d floats: %6.2f cycles/instr (1 block)  %6.2f (512 blocks)\n", st, run<0>(1, st), run<0>(512, st));
  for (int st : {17, 29}) printf("ds_read2_b32  lane stride %3d floats: %6.2f cycles/instr (1 block)  %6.2f (512 blocks)\n", st, run<3>(1, st), run<3>(512, st));
  for (int st : {2, 6, 18, 34, 4}) printf("ds_read_b64   lane stride %3d floats: %6.2f cycles/instr (1 block)  %6.2f (512 blocks)\n", st, run<1>(1, st), run<1>(512, st));
  for (int st : {4, 12, 20, 36, 8}) printf("ds_read_b128  lane stride %3d floats: %6.2f cycles/instr (1 block)  %6.2f (512 blocks)\n", st, run<2>(1, st), run<2>(512, st));
  return 0;
}
