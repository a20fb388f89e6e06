// Diagnostic (not part of the product): checks the v_mfma_f32_32x32x1_2b_f32 operand/result layout and the
// v_permlane32_swap semantics on gfx950, then validates and times an LDL^T solve built on rank-1 MFMA updates
// against the crossbar Gauss-Jordan solve the kernels use.
// Build+run on the GPU box: hipcc -O3 --offload-arch=gfx950 tools/mfma_probe.hip -o gpurun_out/mfma_probe && gpurun_out/mfma_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define DEV __device__ __forceinline__
typedef float v32f __attribute__((ext_vector_type(32)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x1f32((a), (b), (c), 0, 0, 0)
#define MINVAL 1e-15f
extern __shared__ float lds[];
DEV float readlane_f(float x, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), l)); }
DEV float bperm_f(int addr, float x) { return __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(x))); }

__global__ void layout_probe(float *out, unsigned *sw) {
  const int l = threadIdx.x;
  v32f c = {};
  c = MFMA((float)(1 + l), 1000.0f * (1 + l), c);
  for (int r = 0; r < 32; ++r) out[l * 32 + r] = c[r];
  auto p = __builtin_amdgcn_permlane32_swap((unsigned)l, (unsigned)(100 + l), false, false);
  sw[2 * l] = p[0]; sw[2 * l + 1] = p[1];
}

// ------------------------------------------------------------ MFMA LDL^T solve
template <int I> DEV float row_vec(const v32f &acc) {  // lane (b,k) <- acc_b[I][k]
  constexpr int v = 4 * (I / 8) + (I % 4), h = (I / 4) & 1;
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[v]), __float_as_uint(acc[16 + v]), false, false);
  return __uint_as_float(h == 0 ? r[0] : r[1]);
}
template <int NB> DEV float bcast_block(float x, int J, int lane) {
  const float lo = readlane_f(x, J);
  if (NB == 1) return lo;
  const float hi = readlane_f(x, 32 + J);
  return lane < 32 ? lo : hi;
}
template <int N, int NB, int J> struct Elim {
  static DEV void run(v32f &C, v32f &W, float &bv, float &sd, int lane) {
    const int k = lane & 31;
    const float u = row_vec<J>(C), w = row_vec<J>(W);
    const float piv = bcast_block<NB>(u, J, lane);
    const float s = __builtin_amdgcn_rcpf(fmaxf(piv, MINVAL));
    float a = (k > J) ? -u * s : 0.0f;
    if (NB == 1) a = lane < 32 ? a : 0.0f;
    C = MFMA(a, u, C);
    W = MFMA(a, w, W);
    const float bJ = bcast_block<NB>(bv, J, lane);
    bv = __builtin_fmaf(a, bJ, bv);
    sd = (k == J) ? s : sd;
    Elim<N, NB, J + 1>::run(C, W, bv, sd, lane);
  }
};
template <int N, int NB> struct Elim<N, NB, N> { static DEV void run(v32f &, v32f &, float &, float &, int) {} };

// M: full symmetric [n][ldm] per environment at lds[M_off(b)]; rhs/x strided vectors of the lane's own environment
template <int N, int NB>
__device__ __noinline__ void spd_solve_mfma(int M_off0, int M_off1, int ldm, int n, int rhs_off, int x_off, int vstride, int lane) {
  const int h = lane >> 5, j = lane & 31, jc = j < n ? j : 0;
  v32f C, W;
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int Mb = (b == 0 ? M_off0 : M_off1) + jc;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 8 * q + 4 * h + r;
        float val = 0.0f;
        if (b < NB && 8 * q < N) {
          const float m = lds[Mb + (i < n ? i : 0) * ldm];
          val = (i < n && j < n) ? m : (i == j ? 1.0f : 0.0f);
        }
        C[16 * b + 4 * q + r] = val;
        W[16 * b + 4 * q + r] = (i == j) ? 1.0f : 0.0f;
      }
  }
  float bv = lds[rhs_off + jc * vstride];
  bv = (j < n && (NB == 2 || lane < 32)) ? bv : 0.0f;
  float sd = 0.0f;
  Elim<N, NB, 0>::run(C, W, bv, sd, lane);
  const float v = bv * sd;  // D^-1 L^-1 b, lane layout
  float P[2] = {0.0f, 0.0f};
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (8 * q + r < N) {
          const float vi = bperm_f(4 * (32 * b + 8 * q + r) + 16 * h, v);
          P[b] = __builtin_fmaf(W[16 * b + 4 * q + r], vi, P[b]);
        }
  auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(P[0]), __float_as_uint(P[1]), false, false);
  const float x = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
  if (j < n && (NB == 2 || lane < 32)) lds[x_off + j * vstride] = x;
}

// ------------------------------------------------------------ reference: crossbar Gauss-Jordan (as in amx_kernels.hip)
template <int N, int J> struct GJ {
  static DEV void run(float (&a)[N], float &b, int gl, int gb4) {
    float u[N];
    const float piv = bperm_f(gb4 + 4 * J, a[J]);
    const float bj = bperm_f(gb4 + 4 * J, b);
#pragma unroll
    for (int k = J + 1; k < N; ++k) u[k] = bperm_f(gb4 + 4 * J, a[k]);
    float f = a[J] * __builtin_amdgcn_rcpf(fmaxf(piv, MINVAL));
    f = gl == J ? 0.0f : f;
    b = __builtin_fmaf(-f, bj, b);
#pragma unroll
    for (int k = J + 1; k < N; ++k) a[k] = __builtin_fmaf(-f, u[k], a[k]);
    GJ<N, J + 1>::run(a, b, gl, gb4);
  }
};
template <int N> struct GJ<N, N> { static DEV void run(float (&)[N], float &, int, int) {} };
template <int N>
__device__ __noinline__ void spd_solve_gj(int M_off, int ldm, int n, int rhs_off, int x_off, int vstride, int gl, int gb4) {
  float a[N];
  const int li = gl < n ? gl : 0;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const float v = lds[M_off + li * ldm + (k < n ? k : 0)];
    a[k] = (gl < n && k < n) ? v : (k == gl ? 1.0f : 0.0f);
  }
  float b = lds[rhs_off + li * vstride];
  b = gl < n ? b : 0.0f;
  GJ<N, 0>::run(a, b, gl, gb4);
  float d = 1.0f;
#pragma unroll
  for (int k = 0; k < N; ++k) d = (k == gl) ? a[k] : d;
  if (gl < n) lds[x_off + gl * vstride] = b / d;
}


// ------------------------------------------------------------ hybrid: part of each pivot row through the crossbar, part through
// v_readlane + exec-masked v_fmac with the SGPR as operand (two 32-lane groups per wave)
#ifndef XNUM
#define XNUM 1
#define XDEN 2
#endif
template <int NC> DEV void masked_fmac(float *a, const float *lo, const float *hi, float nf) {
  static_assert(NC >= 1 && NC <= 4, "chunk");
  unsigned long long save;
  float a0 = a[0], a1 = NC > 1 ? a[1] : 0.f, a2 = NC > 2 ? a[2] : 0.f, a3 = NC > 3 ? a[3] : 0.f;
  const float l0 = lo[0], l1 = NC > 1 ? lo[1] : 0.f, l2 = NC > 2 ? lo[2] : 0.f, l3 = NC > 3 ? lo[3] : 0.f;
  const float h0 = hi[0], h1 = NC > 1 ? hi[1] : 0.f, h2 = NC > 2 ? hi[2] : 0.f, h3 = NC > 3 ? hi[3] : 0.f;
  asm volatile(
      "s_nop 2\n\t"
      "s_mov_b64 %[sv], exec\n\t"
      "s_mov_b64 exec, 0xffffffff\n\t"
      "v_fmac_f32 %[a0], %[l0], %[nf]\n\t"
      "v_fmac_f32 %[a1], %[l1], %[nf]\n\t"
      "v_fmac_f32 %[a2], %[l2], %[nf]\n\t"
      "v_fmac_f32 %[a3], %[l3], %[nf]\n\t"
      "s_not_b64 exec, exec\n\t"
      "v_fmac_f32 %[a0], %[h0], %[nf]\n\t"
      "v_fmac_f32 %[a1], %[h1], %[nf]\n\t"
      "v_fmac_f32 %[a2], %[h2], %[nf]\n\t"
      "v_fmac_f32 %[a3], %[h3], %[nf]\n\t"
      "s_mov_b64 exec, %[sv]"
      : [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3), [sv] "=&s"(save)
      : [l0] "s"(l0), [l1] "s"(l1), [l2] "s"(l2), [l3] "s"(l3), [h0] "s"(h0), [h1] "s"(h1), [h2] "s"(h2), [h3] "s"(h3), [nf] "v"(nf)
      : "scc");
  a[0] = a0; if (NC > 1) a[1] = a1; if (NC > 2) a[2] = a2; if (NC > 3) a[3] = a3;
}
template <int N, int J> struct GJH {
  static DEV void run(float (&a)[N], float &b, int gl, int gb4) {
    constexpr int nrl = ((N - J - 1) * (XDEN - XNUM)) / XDEN;  // columns J+1 .. J+nrl by readlane, the rest by crossbar
    constexpr int ksplit = J + 1 + nrl;
    float u[N];
    const float piv = bperm_f(gb4 + 4 * J, a[J]);
    const float bj = bperm_f(gb4 + 4 * J, b);
#pragma unroll
    for (int k = ksplit; k < N; ++k) u[k] = bperm_f(gb4 + 4 * J, a[k]);
    float f = a[J] * __builtin_amdgcn_rcpf(fmaxf(piv, MINVAL));
    f = gl == J ? 0.0f : f;
    const float nf = -f;
    b = __builtin_fmaf(nf, bj, b);
#pragma unroll
    for (int k0 = J + 1; k0 < ksplit; k0 += 4) {
      float lo[4], hi[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { const int k = k0 + i < ksplit ? k0 + i : ksplit - 1; lo[i] = readlane_f(a[k], J); hi[i] = readlane_f(a[k], J + 32); }
      constexpr int dummy = 0;
      const int nc = ksplit - k0;
      if (nc >= 4) masked_fmac<4>(&a[k0], lo, hi, nf);
      else if (nc == 3) masked_fmac<3>(&a[k0], lo, hi, nf);
      else if (nc == 2) masked_fmac<2>(&a[k0], lo, hi, nf);
      else masked_fmac<1>(&a[k0], lo, hi, nf);
    }
#pragma unroll
    for (int k = ksplit; k < N; ++k) a[k] = __builtin_fmaf(nf, u[k], a[k]);
    GJH<N, J + 1>::run(a, b, gl, gb4);
  }
};
template <int N> struct GJH<N, N> { static DEV void run(float (&)[N], float &, int, int) {} };
template <int N>
__device__ __noinline__ void spd_solve_gjh(int M_off, int ldm, int n, int rhs_off, int x_off, int vstride, int gl, int gb4) {
  float a[N];
  const int li = gl < n ? gl : 0;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const float v = lds[M_off + li * ldm + (k < n ? k : 0)];
    a[k] = (gl < n && k < n) ? v : (k == gl ? 1.0f : 0.0f);
  }
  float b = lds[rhs_off + li * vstride];
  b = gl < n ? b : 0.0f;
  GJH<N, 0>::run(a, b, gl, gb4);
  float d = 1.0f;
#pragma unroll
  for (int k = 0; k < N; ++k) d = (k == gl) ? a[k] : d;
  if (gl < n) lds[x_off + gl * vstride] = b / d;
}

// each wave: 2 environments (32 lanes each); per env: M [n][ldm], rhs[n], x[n]
template <int N, int MODE>
__global__ void __launch_bounds__(256, 2) solve_bench(const float *Mg, const float *bg, float *xg, int n, int reps, unsigned long long *cyc) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, ldm = n | 1;
  const int envf = n * ldm + 2 * n + 3;  // floats per env in LDS
  const int env0 = (wave * 2) * envf, env1 = env0 + envf, mine = lane < 32 ? env0 : env1;
  const int genv = (blockIdx.x * 4 + wave) * 2 + (lane >> 5), gl = lane & 31;
  for (int e = gl; e < n * n; e += 32) lds[mine + (e / n) * ldm + e % n] = Mg[(size_t)genv * n * n + e];
  if (gl < n) lds[mine + n * ldm + gl] = bg[(size_t)genv * n + gl];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < reps; ++it) {
    if (MODE == 0) spd_solve_mfma<N, 2>(env0, env1, ldm, n, mine + n * ldm, mine + n * ldm + n, 1, lane);
    else if (MODE == 1) spd_solve_gj<N>(mine, ldm, n, mine + n * ldm, mine + n * ldm + n, 1, gl, 4 * (lane & 32));
    else spd_solve_gjh<N>(mine, ldm, n, mine + n * ldm, mine + n * ldm + n, 1, gl, 4 * (lane & 32));
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (gl < n) xg[(size_t)genv * n + gl] = lds[mine + n * ldm + n + gl];
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int N, int MODE> void run_solve(const char *name, int n, int nblocks, int reps) {
  const int nenv = nblocks * 8;
  std::vector<float> M((size_t)nenv * n * n), b((size_t)nenv * n), x((size_t)nenv * n);
  srand(1);
  for (int e = 0; e < nenv; ++e) {
    std::vector<double> A(n * n);
    for (auto &v : A) v = (rand() / (double)RAND_MAX) - 0.5;
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j) {
        double s = 0;
        for (int k = 0; k < n; ++k) s += A[i * n + k] * A[j * n + k];
        M[(size_t)e * n * n + i * n + j] = (float)(s + (i == j ? 0.5 : 0.0));
      }
    for (int i = 0; i < n; ++i) b[(size_t)e * n + i] = (float)((rand() / (double)RAND_MAX) - 0.5);
  }
  float *dM, *db, *dx; unsigned long long *dc;
  CK(hipMalloc(&dM, M.size() * 4)); CK(hipMalloc(&db, b.size() * 4)); CK(hipMalloc(&dx, x.size() * 4)); CK(hipMalloc(&dc, 8));
  CK(hipMemcpy(dM, M.data(), M.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice));
  const int ldm = n | 1, lds_bytes = 8 * (n * ldm + 2 * n + 3) * 4;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  solve_bench<N, MODE><<<nblocks, 256, lds_bytes>>>(dM, db, dx, n, 1, dc);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(x.data(), dx, x.size() * 4, hipMemcpyDeviceToHost));
  double worst = 0;
  for (int e = 0; e < nenv; ++e) {  // residual check in double
    double rn = 0, bn = 0;
    for (int i = 0; i < n; ++i) {
      double s = 0;
      for (int j = 0; j < n; ++j) s += (double)M[(size_t)e * n * n + i * n + j] * x[(size_t)e * n + j];
      rn += (s - b[(size_t)e * n + i]) * (s - b[(size_t)e * n + i]); bn += (double)b[(size_t)e * n + i] * b[(size_t)e * n + i];
    }
    worst = fmax(worst, sqrt(rn / bn));
  }
  CK(hipEventRecord(e0));
  solve_bench<N, MODE><<<nblocks, 256, lds_bytes>>>(dM, db, dx, n, reps, dc);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long c; CK(hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost));
  printf("%-28s n=%2d N=%2d blocks=%4d: worst relative residual %.3e   %.1f cycles/solve (wave 0)   %.3f ms for %d reps\n", name, n, N, nblocks, worst,
         (double)c / reps, ms, reps);
  hipFree(dM); hipFree(db); hipFree(dx); hipFree(dc);
}

int main() {
  float *out; unsigned *sw;
  CK(hipMalloc(&out, 64 * 32 * 4)); CK(hipMalloc(&sw, 128 * 4));
  layout_probe<<<1, 64>>>(out, sw);
  CK(hipDeviceSynchronize());
  std::vector<float> h(64 * 32); std::vector<unsigned> hs(128);
  CK(hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hs.data(), sw, 128 * 4, hipMemcpyDeviceToHost));
  // decode: value = (1 + la) * 1000 * (1 + lb): la = A lane (block*32 + i), lb = B lane (block*32 + j)
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 32; ++r) {
      const int b = r / 16, q = (r % 16) / 4, rr = r % 4, hh = l / 32, j = l % 32, i = 8 * q + 4 * hh + rr;
      const float expect = (float)(1 + 32 * b + i) * 1000.0f * (1 + 32 * b + j);
      if (h[l * 32 + r] != expect) { if (bad < 8) printf("layout mismatch lane %d reg %d: got %.0f expected %.0f\n", l, r, h[l * 32 + r], expect); bad++; }
    }
  printf("mfma 32x32x1 2-block layout: %s (reg 16b+4q+r of lane (h,j) = D_b[8q+4h+r][j])\n", bad ? "MISMATCH" : "as assumed");
  printf("permlane32_swap(vdst=l, vsrc=100+l): lane0 -> (%u,%u) lane31 -> (%u,%u) lane32 -> (%u,%u) lane63 -> (%u,%u)\n", hs[0], hs[1], hs[62], hs[63], hs[64], hs[65],
         hs[126], hs[127]);
  run_solve<18, 1>("crossbar Gauss-Jordan", 18, 1, 200);
  run_solve<18, 2>("hybrid crossbar+readlane GJ", 18, 1, 200);
  run_solve<18, 0>("mfma LDL^T", 18, 1, 200);
  run_solve<18, 1>("crossbar Gauss-Jordan", 18, 512, 200);
  run_solve<18, 2>("hybrid crossbar+readlane GJ", 18, 512, 200);
  run_solve<18, 0>("mfma LDL^T", 18, 512, 200);
  run_solve<30, 1>("crossbar Gauss-Jordan", 30, 512, 200);
  run_solve<30, 2>("hybrid crossbar+readlane GJ", 30, 512, 200);
  run_solve<30, 0>("mfma LDL^T", 30, 512, 200);
  return 0;
}
