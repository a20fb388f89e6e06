// A stand-in for a collective's kernels on a second stream: `nblocks` workgroups of 256 threads that hold about `VG` vector
// registers each and spin for `cycles` of s_memtime.  Used by tools/probes/coresidency.py to see whether such workgroups can be
// resident next to amx_rollout_kernel's (which fill every CU's LDS and 88 % of the VGPR file).
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -no-hip-rt tools/probes/fake_comm.hip -o tools/probes/libfake_comm.so
#include <hip/hip_runtime.h>
template <int VG> __global__ void __launch_bounds__(256) spin(float *sink, unsigned long long cycles) {
  float r[VG];
#pragma unroll
  for (int i = 0; i < VG; ++i) r[i] = threadIdx.x * 0.001f + i;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < cycles) {
#pragma unroll
    for (int i = 0; i < VG; ++i) r[i] = r[i] * 1.0001f + r[(i + 1) % VG] * 1e-6f;
    __builtin_amdgcn_s_sleep(8);
  }
  float acc = 0;
#pragma unroll
  for (int i = 0; i < VG; ++i) acc += r[i];
  if (acc == 12345.678f) sink[0] = acc;
}
extern "C" int fake_comm(void *stream, int nblocks, unsigned long long cycles, int vg, float *sink) {
  if (vg <= 16) hipLaunchKernelGGL(spin<4>, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, sink, cycles);
  else if (vg <= 32) hipLaunchKernelGGL(spin<24>, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, sink, cycles);
  else if (vg <= 64) hipLaunchKernelGGL(spin<56>, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, sink, cycles);
  else hipLaunchKernelGGL(spin<112>, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, sink, cycles);
  return (int)hipGetLastError();
}
