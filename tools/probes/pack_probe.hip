// Probe (GPU box): the packed group sums of amx_kernels.hip against plain gsum, for 16-, 32- and 64-lane groups.
//   hipcc -O3 --offload-arch=gfx950 -I ambersim_amd/csrc -o /tmp/pack_probe tools/probes/pack_probe.hip && /tmp/pack_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include "amx_specs_gen.h"
#include "amx_kernels.hip"

template <int G> __global__ void probe(const float *in, float *out) {
  const int l = threadIdx.x;
  float v[6];
  for (int k = 0; k < 6; ++k) v[k] = in[k * 64 + l];
  float u, w;
  gsum_pack6<G>(v[0], v[1], v[2], v[3], v[4], v[5], u, w);
  const float r2 = gsum_pack2<G>(v[0], v[1]);
  const float r4 = gsum_pack4<G>(v[2], v[3], v[4], v[5]);
  out[0 * 64 + l] = quad_value<0>(u); out[1 * 64 + l] = quad_value<1>(u); out[2 * 64 + l] = quad_value<3>(u); out[3 * 64 + l] = quad_value<2>(u);
  out[4 * 64 + l] = quad_value<0>(w); out[5 * 64 + l] = quad_value<1>(w);
  out[6 * 64 + l] = quad_value<0>(r2); out[7 * 64 + l] = quad_value<1>(r2);
  for (int k = 0; k < 6; ++k) out[(8 + k) * 64 + l] = gsum<G>(v[k]);
  out[14 * 64 + l] = quad_value<0>(r4); out[15 * 64 + l] = quad_value<1>(r4); out[16 * 64 + l] = quad_value<3>(r4); out[17 * 64 + l] = quad_value<2>(r4);
}
template <int G> int run() {
  float h[6 * 64], o[18 * 64], *di, *dout;
  for (int i = 0; i < 6 * 64; ++i) h[i] = (float)((i * 7919) % 101) - 50.0f;
  hipMalloc(&di, sizeof(h)); hipMalloc(&dout, sizeof(o));
  hipMemcpy(di, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe<G>, dim3(1), dim3(64), 0, 0, di, dout);
  hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    for (int k = 0; k < 6; ++k) if (o[k * 64 + l] != o[(8 + k) * 64 + l]) { if (bad < 8) printf("G=%d lane %d value %d: packed %g plain %g\n", G, l, k, o[k * 64 + l], o[(8 + k) * 64 + l]); ++bad; }
    for (int k = 0; k < 4; ++k) if (o[(14 + k) * 64 + l] != o[(10 + k) * 64 + l]) { if (bad < 8) printf("G=%d lane %d pack4 value %d: packed %g plain %g\n", G, l, k, o[(14 + k) * 64 + l], o[(10 + k) * 64 + l]); ++bad; }
    for (int k = 0; k < 2; ++k) if (o[(6 + k) * 64 + l] != o[(8 + k) * 64 + l]) { if (bad < 8) printf("G=%d lane %d pack2 value %d: packed %g plain %g\n", G, l, k, o[(6 + k) * 64 + l], o[(8 + k) * 64 + l]); ++bad; }
  }
  printf("G=%d: %d mismatches (integer-valued inputs: sums are exact in any order)\n", G, bad);
  return bad;
}
int main() { int b = run<16>() + run<32>() + run<64>(); return b != 0; }
