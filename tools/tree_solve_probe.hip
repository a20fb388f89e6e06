// Diagnostic (not part of the product): level-scheduled tree-sparse LDL^T solve (lane = dof, registers = the dof's
// ancestor-chain entries, pivot rows published through LDS) against the dense crossbar Gauss-Jordan, on the quadruped's
// dof tree (6 base dofs in a chain + 4 legs of 3).  hipcc -O3 --offload-arch=gfx950 tools/tree_solve_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define DEV __device__ __forceinline__
#define MINVAL 1e-15f
extern __shared__ float lds[];
#define WSYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
#define CI(o) (__float_as_int(lds[(o)]))
DEV float bperm_f(int addr, float x) { return __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(x))); }

// ---- dense reference (as in amx_kernels.hip)
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
template <int J> DEV float swz_f(float x) { return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(x), (J << 5))); }
template <int N, int J> struct GJS {
  static DEV void run(float (&a)[N], float &b, int gl) {
    float u[N];
    const float piv = swz_f<J>(a[J]);
    const float bj = swz_f<J>(b);
#pragma unroll
    for (int k = J + 1; k < N; ++k) u[k] = swz_f<J>(a[k]);
    float f = a[J] * __builtin_amdgcn_rcpf(fmaxf(piv, MINVAL));
    f = gl == J ? 0.0f : f;
    b = __builtin_fmaf(-f, bj, b);
#pragma unroll
    for (int k = J + 1; k < N; ++k) a[k] = __builtin_fmaf(-f, u[k], a[k]);
    GJS<N, J + 1>::run(a, b, gl);
  }
};
template <int N> struct GJS<N, N> { static DEV void run(float (&)[N], float &, int) {} };
template <int N> __device__ __noinline__ void solve_gjs(int A_off, int ld, int n, int rhs_off, int x_off, int gl) {
  float a[N];
  const int li = gl < n ? gl : 0;
  const float4 *rowp = reinterpret_cast<const float4 *>(lds + A_off + li * ld);
#pragma unroll
  for (int c = 0; c < (N + 3) / 4; ++c) {
    const float4 v = rowp[c];
    const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) { const int k = 4 * c + q; if (k < N) a[k] = (gl < n && k < n) ? e[q] : (k == gl ? 1.0f : 0.0f); }
  }
  float b = gl < n ? lds[rhs_off + li] : 0.0f;
  GJS<N, 0>::run(a, b, gl);
  float d = 1.0f;
#pragma unroll
  for (int k = 0; k < N; ++k) d = (k == gl) ? a[k] : d;
  if (gl < n) lds[x_off + gl] = b / d;
  WSYNC();
}
template <int N> __device__ __noinline__ void solve_gj(int A_off, int ld, int n, int rhs_off, int x_off, int gl, int gb4) {
  float a[N];
  const int li = gl < n ? gl : 0;
  const float4 *rowp = reinterpret_cast<const float4 *>(lds + A_off + li * ld);
#pragma unroll
  for (int c = 0; c < (N + 3) / 4; ++c) {
    const float4 v = rowp[c];
    const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) { const int k = 4 * c + q; if (k < N) a[k] = (gl < n && k < n) ? e[q] : (k == gl ? 1.0f : 0.0f); }
  }
  float b = gl < n ? lds[rhs_off + li] : 0.0f;
  GJ<N, 0>::run(a, b, gl, gb4);
  float d = 1.0f;
#pragma unroll
  for (int k = 0; k < N; ++k) d = (k == gl) ? a[k] : d;
  if (gl < n) lds[x_off + gl] = b / d;
  WSYNC();
}

// ---- tree-sparse LDL^T.  Constant tables at lds[tab]: depth[32], anc[32][DM1] (dof index per chain slot), nlevel,
// per level: R, then T[level][r][32] (descendant at that level or -1).  DM1 = max depth + 1 (chain length).
struct TreeTab { int depth, anc, lvl_R, lvl_T, maxdepth, Rmax; };
template <int DM1>  // chain slots held in registers
__device__ __noinline__ void solve_tree(int A_off, int ld, int n, int rhs_off, int x_off, int S_off, int X_off, TreeTab t, int gl) {
  constexpr int SL = ((DM1 + 2 + 3) / 4) * 4;  // published row: chain entries, then y and 1/diag in the last two slots
  const bool on = gl < n;
  const int i = on ? gl : 0;
  const int Di = CI(t.depth + i);
  float a[DM1];
#pragma unroll
  for (int d = 0; d < DM1; ++d) {
    const int col = CI(t.anc + i * DM1 + d);
    const float v = lds[A_off + i * ld + (d <= Di ? col : i)];
    a[d] = d <= Di ? v : 0.0f;
  }
  float y = on ? lds[rhs_off + i] : 0.0f;
  float4 *Srow = reinterpret_cast<float4 *>(lds + S_off + i * SL);
  // ---- factor + forward substitution, leaves first
  for (int lv = t.maxdepth; lv >= 1; --lv) {
    if (on && Di == lv) {
      float diag = 1.0f;
#pragma unroll
      for (int d = 0; d < DM1; ++d) diag = d == lv ? a[d] : diag;
      const float dinv = __builtin_amdgcn_rcpf(fmaxf(diag, MINVAL));
#pragma unroll
      for (int c = 0; c < SL / 4; ++c) {
        float4 v;
        v.x = 4 * c < DM1 ? a[4 * c < DM1 ? 4 * c : 0] : 0.0f;
        v.y = 4 * c + 1 < DM1 ? a[4 * c + 1 < DM1 ? 4 * c + 1 : 0] : 0.0f;
        v.z = 4 * c + 2 < DM1 ? a[4 * c + 2 < DM1 ? 4 * c + 2 : 0] : (4 * c + 2 == SL - 2 ? y : 0.0f);
        v.w = 4 * c + 3 < DM1 ? a[4 * c + 3 < DM1 ? 4 * c + 3 : 0] : (4 * c + 3 == SL - 1 ? dinv : 0.0f);
        Srow[c] = v;
      }
    }
    WSYNC();
    const int R = CI(t.lvl_R + lv);
    for (int r = 0; r < R; ++r) {
      const int k = CI(t.lvl_T + (lv * t.Rmax + r) * 32 + i);
      if (on && Di < lv && k >= 0) {
        const float4 *P = reinterpret_cast<const float4 *>(lds + S_off + k * SL);
        float s[SL];
#pragma unroll
        for (int c = 0; c < SL / 4; ++c) { const float4 v = P[c]; s[4 * c] = v.x; s[4 * c + 1] = v.y; s[4 * c + 2] = v.z; s[4 * c + 3] = v.w; }
        const float mine = lds[S_off + k * SL + Di];  // H[k][i]
        const float f = mine * s[SL - 1];
#pragma unroll
        for (int d = 0; d < DM1; ++d) a[d] = __builtin_fmaf(-f, d <= Di ? s[d] : 0.0f, a[d]);
        y = __builtin_fmaf(-f, s[SL - 2], y);
      }
    }
    WSYNC();  // the scratch rows are rewritten by the next level's pivots only after every reader is done (in-order LDS per wave)
  }
  // ---- diagonal + backward substitution, root first: x_k = y_k / d_k - sum_d (a_k[d] / d_k) x_anc(d)
  float diag = 1.0f;
#pragma unroll
  for (int d = 0; d < DM1; ++d) diag = d == Di ? a[d] : diag;
  const float dinv = __builtin_amdgcn_rcpf(fmaxf(diag, MINVAL));
  float acc = y;
  for (int lv = 0; lv <= t.maxdepth; ++lv) {
    if (on && Di == lv) lds[X_off + i] = acc * dinv;
    WSYNC();
    if (on && Di > lv) {
      float al = 0.0f;
#pragma unroll
      for (int d = 0; d < DM1; ++d) al = d == lv ? a[d] : al;
      acc = __builtin_fmaf(-al, lds[X_off + CI(t.anc + i * DM1 + lv)], acc);
    }
  }
  if (on) lds[x_off + i] = lds[X_off + i];
  WSYNC();
}

template <int MODE>
__global__ void __launch_bounds__(256, 2) bench(const float *Hg, const float *bg, float *xg, const int *tabg, int ntab, TreeTab t0, int n, int reps, unsigned long long *cyc) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, ld = 20;
  for (int k = threadIdx.x; k < ntab; k += blockDim.x) lds[k] = __int_as_float(tabg[k]);
  __syncthreads();
  const int envf = n * ld + 2 * n + 32 * 16 + 32 + 8;
  const int base = ntab + (wave * 2 + (lane >> 5)) * envf, gl = lane & 31;
  const int genv = (blockIdx.x * 4 + wave) * 2 + (lane >> 5);
  for (int e = gl; e < n * n; e += 32) lds[base + (e / n) * ld + e % n] = Hg[(size_t)genv * n * n + e];
  if (gl < n) lds[base + n * ld + gl] = bg[(size_t)genv * n + gl];
  WSYNC();
  TreeTab t = t0;
  const unsigned long long c0 = __builtin_readcyclecounter();
  for (int it = 0; it < reps; ++it) {
    if (MODE == 0) solve_tree<9>(base, ld, n, base + n * ld, base + n * ld + n, base + n * ld + 2 * n, base + n * ld + 2 * n + 32 * 16, t, gl);
    else if (MODE == 1) solve_gj<18>(base, ld, n, base + n * ld, base + n * ld + n, gl, 4 * (lane & 32));
    else solve_gjs<18>(base, ld, n, base + n * ld, base + n * ld + n, gl);
  }
  const unsigned long long c1 = __builtin_readcyclecounter();
  if (gl < n) xg[(size_t)genv * n + gl] = lds[base + n * ld + n + gl];
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = c1 - c0;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
int main() {
  const int n = 18, DM1 = 9;
  std::vector<int> parent(n), depth(n);
  for (int i = 0; i < 6; ++i) parent[i] = i - 1;
  for (int c = 0; c < 4; ++c) for (int tt = 0; tt < 3; ++tt) parent[6 + 3 * c + tt] = tt == 0 ? 5 : 6 + 3 * c + tt - 1;
  for (int i = 0; i < n; ++i) depth[i] = parent[i] < 0 ? 0 : depth[parent[i]] + 1;
  int maxdepth = 0; for (int d : depth) maxdepth = std::max(maxdepth, d);
  std::vector<std::vector<int>> anc(n);
  for (int i = 0; i < n; ++i) { std::vector<int> ch; for (int j = i; j >= 0; j = parent[j]) ch.insert(ch.begin(), j); anc[i] = ch; }
  // tables
  std::vector<int> tab;
  TreeTab t; t.maxdepth = maxdepth;
  t.depth = (int)tab.size(); for (int i = 0; i < 32; ++i) tab.push_back(i < n ? depth[i] : 0);
  t.anc = (int)tab.size(); for (int i = 0; i < 32; ++i) for (int d = 0; d < DM1; ++d) tab.push_back(i < n && d < (int)anc[i].size() ? anc[i][d] : 0);
  std::vector<std::vector<std::vector<int>>> desc(maxdepth + 1, std::vector<std::vector<int>>(32));
  int Rmax = 1;
  for (int k = 0; k < n; ++k) for (int d = 0; d < depth[k]; ++d) desc[depth[k]][anc[k][d]].push_back(k);
  for (auto &L : desc) for (auto &v : L) Rmax = std::max(Rmax, (int)v.size());
  t.Rmax = Rmax;
  t.lvl_R = (int)tab.size(); for (int lv = 0; lv <= maxdepth; ++lv) { int R = 0; for (auto &v : desc[lv]) R = std::max(R, (int)v.size()); tab.push_back(R); }
  t.lvl_T = (int)tab.size();
  for (int lv = 0; lv <= maxdepth; ++lv) for (int r = 0; r < Rmax; ++r) for (int i = 0; i < 32; ++i) tab.push_back(r < (int)desc[lv][i].size() ? desc[lv][i][r] : -1);
  while (tab.size() % 4) tab.push_back(0);
  printf("nv %d maxdepth %d Rmax %d table %zu ints\n", n, maxdepth, Rmax, tab.size());
  for (int blocks : {1, 512}) {
    const int nenv = blocks * 8;
    std::vector<float> H((size_t)nenv * n * n), b((size_t)nenv * n), x((size_t)nenv * n);
    srand(2);
    for (int e = 0; e < nenv; ++e) {
      std::vector<double> L(n * n, 0.0), D(n);
      for (int k = 0; k < n; ++k) { L[k * n + k] = 1; D[k] = 0.5 + rand() / (double)RAND_MAX; for (int d = 0; d < depth[k]; ++d) L[k * n + anc[k][d]] = 0.6 * (rand() / (double)RAND_MAX - 0.5); }
      for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double s = 0; for (int k = 0; k < n; ++k) s += L[k * n + i] * D[k] * L[k * n + j]; H[(size_t)e * n * n + i * n + j] = (float)s; }
      for (int i = 0; i < n; ++i) b[(size_t)e * n + i] = (float)(rand() / (double)RAND_MAX - 0.5);
    }
    float *dH, *db, *dx; int *dt; unsigned long long *dc;
    CK(hipMalloc(&dH, H.size() * 4)); CK(hipMalloc(&db, b.size() * 4)); CK(hipMalloc(&dx, x.size() * 4)); CK(hipMalloc(&dt, tab.size() * 4)); CK(hipMalloc(&dc, 8));
    CK(hipMemcpy(dH, H.data(), H.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dt, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
    const int envf = n * 20 + 2 * n + 32 * 16 + 32 + 8, lds_bytes = ((int)tab.size() + 8 * envf) * 4;
    for (int mode = 0; mode < 3; ++mode) {
      const int reps = 200;
      for (int pass = 0; pass < 2; ++pass) {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        if (mode == 0) bench<0><<<blocks, 256, lds_bytes>>>(dH, db, dx, dt, (int)tab.size(), t, n, pass ? reps : 1, dc);
        else if (mode == 1) bench<1><<<blocks, 256, lds_bytes>>>(dH, db, dx, dt, (int)tab.size(), t, n, pass ? reps : 1, dc);
        else bench<2><<<blocks, 256, lds_bytes>>>(dH, db, dx, dt, (int)tab.size(), t, n, pass ? reps : 1, dc);
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        if (pass == 0) {
          CK(hipMemcpy(x.data(), dx, x.size() * 4, hipMemcpyDeviceToHost));
          double worst = 0;
          for (int e = 0; e < nenv; ++e) { double rn = 0, bn = 0; for (int i = 0; i < n; ++i) { double s = 0; for (int j = 0; j < n; ++j) s += (double)H[(size_t)e * n * n + i * n + j] * x[(size_t)e * n + j]; rn += (s - b[(size_t)e * n + i]) * (s - b[(size_t)e * n + i]); bn += (double)b[(size_t)e * n + i] * b[(size_t)e * n + i]; } worst = fmax(worst, sqrt(rn / bn)); }
          printf("%-22s blocks=%4d worst relative residual %.3e", mode == 0 ? "tree-sparse LDL^T" : (mode == 1 ? "bpermute Gauss-Jordan" : "swizzle Gauss-Jordan"), blocks, worst);
        } else {
          float ms; CK(hipEventElapsedTime(&ms, e0, e1));
          unsigned long long c; CK(hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost));
          printf("   %.1f cycles/solve (wave 0)   %.3f ms for %d reps\n", (double)c / reps, ms, reps);
        }
      }
    }
  }
  return 0;
}
