// amx_kernels.hip — the hot path: batched forward dynamics + constraint solve + integration
// for gfx950 (MI355X): a 16/32/64-lane group of a wavefront per environment, per-environment state in LDS.
//
// Replaces mjx.forward / mjx.step under MjxEnv.pipeline_step (reference ambersim/rl/base.py:81-96)
// and shoot (reference ambersim/trajopt/shooting.py:22-48).  Stage names follow SURVEY.md §8(a)
// A9.1-A9.14.  Layout rules:
//   * a group of G lanes (G = 16, 32 or 64, the smallest that holds the model's bodies / dofs / rows)
//     owns one environment, so a wavefront advances 64/G environments with one instruction stream;
//     solver loops run to the slowest group of the wave with per-group predicates (a finished group's
//     state is frozen, recomputing its derived vectors is idempotent);
//   * a workgroup holds AMX_WAVES_PER_BLOCK independent waves that share ONE LDS copy of the model
//     constants (staged once, one s_barrier) and never synchronise again — phases inside a wave are
//     ordered by wavefront-scope fences only (LDS operations of one wave execute in order);
//   * static lane roles inside a group: lane b owns body b, lane i owns dof i, lane r owns constraint row r, so
//     per-row solver state (D, aref, Jaref, Jv) lives in registers and role constants are read from
//     LDS records with odd strides (bank-conflict free), never through dependent HBM loads;
//   * per-body / per-dof / per-row data are odd-stride records in LDS; M and H are full symmetric with 16-byte rows
//     (ds_read_b128), the contact Jacobian is chain-compressed;
//   * tree up-scans (composite inertia, force totals) are sums over contiguous record ranges - bodies are numbered
//     depth-first - and the roots are group sums; down-scans walk each body's ancestor chain in registers (deep trees) or
//     run level by level (shallow ones, and any model whose numbering is not depth-first);
//   * the Newton Hessian M + J'DJ is accumulated on the matrix cores (v_mfma_f32_32x32x1_2b: the two blocks are the two
//     environments of a G = 32 wave) and handed to the solver through v_permlane32_swap; a spare lane of the product also
//     delivers J'f (COL31);
//   * the small dense solves (M, M+J'DJ, M+dt*damping) are one-shot Gauss-Jordan eliminations with one matrix row per
//     lane in registers; in the specialised kernels pivot rows are read from the pivot COLUMN by symmetry through DPP row
//     broadcasts (sym_pivot: no LDS instruction), in the generic ones they travel by v_readlane (G = 64) or through the
//     LDS crossbar (ds_swizzle, G < 64);
//   * hot loops are straight-line: loads are unconditional (clamped to a valid address) and conditions are selects - a
//     conditional load becomes a divergent branch with its own s_waitcnt - except where a branch lets a lane group skip
//     real arithmetic (the kernel is VALU-bound);
//   * group reductions use DPP row operations (+ one ds_swizzle / bpermute step across rows), never LDS memory;
//   * HBM traffic of a rollout is the control stream in and the state rows out, both coalesced.
#include <hip/hip_runtime.h>
#include <type_traits>
#include "amx_dev.h"

#define DEV __device__ __forceinline__
// The specialised kernels inline their eliminations: an out-of-line call makes the callee save and restore the caller's live
// vector registers through scratch memory (callee-saved v40..v255), which cost the biped 24 % (r02 A/B).  The generic
// kernels keep their size-dispatched solves out of line (fifteen instantiations per call site would not fit the I-cache).
#define AMX_STATIC_SOLVE_ATTR __device__ __forceinline__
#ifndef AMX_CHAIN_WALK_MIN_LEVELS
#define AMX_CHAIN_WALK_MIN_LEVELS 5
#endif
#ifndef AMX_LEVEL_SCANS
#define AMX_LEVEL_SCANS 0  // 1: poses and velocities travel down the tree level by level as before (A/B switch)
#endif
#ifndef AMX_MFMA_AUG
#define AMX_MFMA_AUG 1  // 0: the fused Newton solve adds M's rows after the matrix cores and takes J'f from column 31 (r02 / r03a), A/B switch
#endif
#ifndef AMX_MFMA_AUG_MIN_NV
#define AMX_MFMA_AUG_MIN_NV 24  // measured: biped (30 dofs) +0.5 %, C4 +1 %; quadruped (18 dofs: three of the tile's eight chunks are zero fill) -1.2 %
#endif
#ifndef AMX_GRADCOL_64
#define AMX_GRADCOL_64 1  // 0: static 64-lane kernels gather J'f per dof as before (A/B switch)
#endif
#ifndef AMX_NO_GRADCOL
#define AMX_NO_GRADCOL 0  // 1: J'f of the contact rows always by the per-dof gather (A/B switch)
#endif
#ifndef AMX_NO_SYM_ELIM
#define AMX_NO_SYM_ELIM 0  // 1: the static eliminations broadcast pivot rows through the crossbar as before (A/B switch)
#endif
#ifndef AMX_SQRT_D
#define AMX_SQRT_D 1  // 0: the matrix cores' operands are (D J, J) as before r05 (A/B switch).  1: both operands are sqrt(D) J - every product
#endif                //    (sqrt(D) J_i)(sqrt(D) J_k) is commutative, so J'DJ comes out bitwise symmetric; with the symmetric updates of sym_pivot
                      //    (AMX_SYM_EXACT) the column the elimination reads in place of the pivot row IS that row, bit for bit.
#ifndef AMX_SYM_EXACT
#define AMX_SYM_EXACT 1  // 0: pivot updates as (c_i / pivot) c_K (r02 - r04: symmetric only up to rounding), A/B switch
#endif
#define MINVAL 1e-15f

enum { JFREE = 0, JBALL = 1, JSLIDE = 2, JHINGE = 3 };
enum { DIS_CONSTRAINT = 1, DIS_EQUALITY = 2, DIS_FRICTIONLOSS = 4, DIS_LIMIT = 8, DIS_CONTACT = 16, DIS_PASSIVE = 32,
       DIS_GRAVITY = 64, DIS_CLAMPCTRL = 128, DIS_WARMSTART = 256, DIS_ACTUATION = 1024,
       DIS_REFSAFE = 2048, DIS_EULERDAMP = 16384 };

extern __shared__ float lds[];
#define CF(off) (lds[(off)])
#define CI(off) (__float_as_int(lds[(off)]))
// Orders this wave's LDS traffic between phases: compiler-level only, the hardware keeps one wave's
// LDS operations in issue order.
#define WSYNC()                                              \
  do {                                                       \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   \
    __builtin_amdgcn_wave_barrier();                         \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   \
  } while (0)

// Model view.  Every integer of the launch plan (sizes, options, LDS offsets - the AMX_MODEL_INTS list of amx_dev.h) is read
// through MV<S>.  S = DynSpec answers from the by-value descriptor: any model the C ABI accepts.  A generated spec
// (amx_specs_gen.h, one per shipped model) answers with compile-time constants: loops unroll with static trip counts, LDS
// offsets fold into instruction immediates, the tree sparsity of M and H is static in the eliminations, and the descriptor
// never occupies scalar registers.  Floats (dt, gravity, tolerances) stay run-time values in both.
struct DynSpec {
  static constexpr bool is_static = false;
#define X(f) static constexpr int f = 0;
  AMX_MODEL_INTS(X)
#undef X
  static constexpr bool tree_H = false;
  static constexpr int dof_parent[32] = {};
  static constexpr int p2_col[32] = {};
};
template <class S> struct MV {
  const DevModel &d;
#define X(f) DEV int f() const { return S::is_static ? S::f : d.f; }
  AMX_MODEL_INTS(X)
#undef X
#define X(f) DEV float f() const { return d.f; }
  AMX_MODEL_FLOATS(X)
#undef X
  DEV const float *image() const { return d.image; }
};

struct f3 { float x, y, z; };
struct f4 { float w, x, y, z; };
DEV f3 mk3(float x, float y, float z) { f3 r = {x, y, z}; return r; }
DEV f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
DEV f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
DEV f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
DEV float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
DEV f3 cross(f3 a, f3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
DEV f4 qmul(f4 a, f4 b) {
  f4 r;
  r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  r.y = a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x;
  r.z = a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w;
  return r;
}
struct m33 { float a[9]; };
DEV m33 q2m(f4 q) {
  m33 r;
  float w = q.w, x = q.x, y = q.y, z = q.z;
  r.a[0] = w * w + x * x - y * y - z * z; r.a[1] = 2 * (x * y - w * z); r.a[2] = 2 * (x * z + w * y);
  r.a[3] = 2 * (x * y + w * z); r.a[4] = w * w - x * x + y * y - z * z; r.a[5] = 2 * (y * z - w * x);
  r.a[6] = 2 * (x * z - w * y); r.a[7] = 2 * (y * z + w * x); r.a[8] = w * w - x * x - y * y + z * z;
  return r;
}
DEV f3 mulv(const m33 &M, f3 v) {
  return mk3(M.a[0] * v.x + M.a[1] * v.y + M.a[2] * v.z, M.a[3] * v.x + M.a[4] * v.y + M.a[5] * v.z,
             M.a[6] * v.x + M.a[7] * v.y + M.a[8] * v.z);
}
DEV f3 mulTv(const m33 &M, f3 v) {
  return mk3(M.a[0] * v.x + M.a[3] * v.y + M.a[6] * v.z, M.a[1] * v.x + M.a[4] * v.y + M.a[7] * v.z,
             M.a[2] * v.x + M.a[5] * v.y + M.a[8] * v.z);
}
DEV f3 qrot(f4 q, f3 v) { return mulv(q2m(q), v); }
// The same rotation for ONE vector per quaternion (r05 experiment): v + 2 u x (u x v + w v), u = (x, y, z) - 18 multiply-adds where the matrix
// route costs 21 to form R and 9 per vector; at the levels of the pose composition, a body's centre of mass, a geom's offset.  MEASURED AND
// REJECTED: -165 static vector instructions, +0.6 % throughput - and noisier poses (the form cancels v against 2 u x (u x v) for large rotations):
// the hard-impact worst case went 9.9e-6 -> 1.3e-5 and one fuzz draw 8e-6 -> 1.3e-5, over the tolerance.  Off; kept as an A/B switch.
#ifndef AMX_QROT1
#define AMX_QROT1 0  // 1: single-vector rotations without the 3 x 3 matrix (A/B switch)
#endif
DEV f3 qrot1(f4 q, f3 v) {
#if AMX_QROT1
  const f3 u = mk3(q.x, q.y, q.z);
  const f3 t = mk3(__builtin_fmaf(q.w, v.x, u.y * v.z - u.z * v.y), __builtin_fmaf(q.w, v.y, u.z * v.x - u.x * v.z), __builtin_fmaf(q.w, v.z, u.x * v.y - u.y * v.x));
  const f3 c = cross(u, t);
  return mk3(__builtin_fmaf(2.0f, c.x, v.x), __builtin_fmaf(2.0f, c.y, v.y), __builtin_fmaf(2.0f, c.z, v.z));
#else
  return qrot(q, v);
#endif
}
// x / y without the IEEE expansion (two v_div_scale, v_div_fmas, v_div_fixup around the same v_rcp_f32): reciprocal, one
// residual correction.  Within 1 ulp of the correctly rounded quotient for operands whose reciprocal neither overflows nor
// goes denormal - used where the operands are sums of squares, clamped pivots and step lengths (AMX_IEEE_DIV=1: plain x / y).
#ifndef AMX_IEEE_DIV
#define AMX_IEEE_DIV 0
#endif
DEV float fdiv(float x, float y) {
#if AMX_IEEE_DIV
  return x / y;
#else
  const float r = __builtin_amdgcn_rcpf(y);
  const float q = x * r;
  return __builtin_fmaf(__builtin_fmaf(-y, q, x), r, q);
#endif
}
// x / y for the line search's Newton steps on the derivative (alpha - d0 / d1): reciprocal and one multiply, 1.5 ulp.  The
// quotient only places the next trial point; every decision is taken on derivatives evaluated AT the trial points, so the last
// bit of a trial point's position is immaterial (AMX_LS_FASTDIV=0: the corrected quotient above)
#ifndef AMX_LS_END_FROM_DERIVS
#define AMX_LS_END_FROM_DERIVS 1  // 0: the bracket ends' quadratic coefficients are summed again at the end of the line search (A/B switch)
#endif
#ifndef AMX_LS_FASTDIV
#define AMX_LS_FASTDIV 1
#endif
DEV float fdiv_ls(float x, float y) {
#if AMX_LS_FASTDIV
  return x * __builtin_amdgcn_rcpf(y);
#else
  return fdiv(x, y);
#endif
}
// Unit vector where only the direction is used (the contact frame's axes): one v_rsq_f32 (1 ulp) instead of the correctly rounded
// square root (a ten-instruction expansion) and a corrected reciprocal.  normalize3 below keeps both: its length is a contact distance.
#ifndef AMX_FAST_NORM
#define AMX_FAST_NORM 1  // 0: correctly rounded square root and corrected reciprocal in the normalisations (A/B switch)
#endif
DEV float inv_norm(float n2) {
#if AMX_FAST_NORM
  return __builtin_amdgcn_rsqf(n2);
#else
  return fdiv(1.0f, sqrtf(n2));
#endif
}
// (the joint quaternions keep the correctly rounded form: a body's orientation places its contact points, and one ulp of a unit
// quaternion is 6e-8 m per metre of lever arm against penetrations of 1e-4 .. 1e-3 m - measured on the sphere-box draw of
// tests/test_convex.py: 4.9e-6 -> 1.05e-5 after one step with v_rsq_f32 here)
DEV f4 qnormalize(f4 q) {
  float n = sqrtf(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
  if (n > 0) { float s = fdiv(1.0f, n); q.w *= s; q.x *= s; q.y *= s; q.z *= s; }
  return q;
}
DEV f3 unit3(f3 v) {
  const float n2 = dot(v, v);
  const float s = n2 > 0 ? inv_norm(n2) : 1.0f;
  return mk3(v.x * s, v.y * s, v.z * s);
}
DEV f3 normalize3(f3 v, float *len) {
  float n = sqrtf(dot(v, v));
  *len = n;
  const float s = fdiv(1.0f, n > 0 ? n : 1.0f);
  return n > 0 ? mk3(v.x * s, v.y * s, v.z * s) : v;
}
// sin and cos of a joint-sized angle (|x| up to a few thousand radians): nearest multiple of pi/2 removed with a two-term
// Cody-Waite reduction (exact products through fma), then the classical single-precision minimax polynomials on
// [-pi/4, pi/4] (Cephes sinf / cosf coefficients, < 1 ulp there) and the quadrant fix-up.  libm's sincosf carries a
// large-argument path with a private table (the only scratch memory the kernels had) and costs several hundred instructions.
DEV void sincos_joint(float x, float *s, float *c) {
  const float k = rintf(x * 0.636619772367581343f);  // x * 2/pi
  float r = __builtin_fmaf(-k, 1.57079625129699707031f, x);
  r = __builtin_fmaf(-k, 7.54978941586159635335e-8f, r);
  const float r2 = r * r;
  const float sp = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(-1.9515295891e-4f, r2, 8.3321608736e-3f), r2, -1.6666654611e-1f) * r2, r, r);
  const float cp = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(2.443315711809948e-5f, r2, -1.388731625493765e-3f), r2, 4.166664568298827e-2f) * r2, r2,
                                  __builtin_fmaf(-0.5f, r2, 1.0f));
  const int q = (int)k & 3;
  const float ss = (q & 1) ? cp : sp, cc = (q & 1) ? sp : cp;
  *s = (q & 2) ? -ss : ss;
  *c = ((q + 1) & 2) ? -cc : cc;
}
DEV f4 axisangle(f3 axis, float ang) {
  float s, c;
  sincos_joint(0.5f * ang, &s, &c);
  f4 r = {c, axis.x * s, axis.y * s, axis.z * s};
  return r;
}
DEV void make_frame(f3 n, float *fr) {  // rows a, b, a x b
  f3 a = unit3(n);
  f3 b = (a.y > -0.5f && a.y < 0.5f) ? mk3(0, 1, 0) : mk3(0, 0, 1);
  b = b - a * dot(a, b);
  b = unit3(b);
  f3 c = cross(a, b);
  fr[0] = a.x; fr[1] = a.y; fr[2] = a.z; fr[3] = b.x; fr[4] = b.y; fr[5] = b.z; fr[6] = c.x; fr[7] = c.y; fr[8] = c.z;
}

// ---- packed fp32 (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32: two operations per instruction at the scalar instructions'
// issue cost).  Operands are aligned register pairs; values that arrive in pairs anyway (ds_read2_b32, ds_read_b64/b128) cost
// nothing to pack.  The auto-vectoriser is off for these kernels (it packs everything it can and pays more in v_mov than it
// saves: 7.6 k -> 8.2 k vector instructions, measured), so the packed forms are written out where the data layout suits them.
typedef float f2v __attribute__((ext_vector_type(2)));
#ifndef AMX_PK_DOT
#define AMX_PK_DOT 1  // 0: scalar multiply-adds in the dot products (A/B switch)
#endif
#ifndef AMX_PK_SV
#define AMX_PK_SV 1  // 0: scalar spatial-vector sums (A/B switch)
#endif
DEV f2v pk_fma(f2v a, f2v b, f2v c) { return __builtin_elementwise_fma(a, b, c); }

// ---- spatial algebra on [rot(3); lin(3)] vectors and 10-float com-frame inertias
struct sv { float v[6]; };
DEV void sv_acc(sv &a, const sv &b) {  // a += b
#if AMX_PK_SV
#pragma unroll
  for (int e = 0; e < 6; e += 2) { const f2v r = f2v{a.v[e], a.v[e + 1]} + f2v{b.v[e], b.v[e + 1]}; a.v[e] = r.x; a.v[e + 1] = r.y; }
#else
#pragma unroll
  for (int e = 0; e < 6; ++e) a.v[e] += b.v[e];
#endif
}
DEV float sv_dot(const sv &a, const sv &b) {
#if AMX_PK_SV
  f2v p = f2v{a.v[0], a.v[1]} * f2v{b.v[0], b.v[1]};
  p = pk_fma(f2v{a.v[2], a.v[3]}, f2v{b.v[2], b.v[3]}, p);
  p = pk_fma(f2v{a.v[4], a.v[5]}, f2v{b.v[4], b.v[5]}, p);
  return p.x + p.y;
#else
  return (a.v[0] * b.v[0] + a.v[1] * b.v[1]) + (a.v[2] * b.v[2] + a.v[3] * b.v[3]) + (a.v[4] * b.v[4] + a.v[5] * b.v[5]);
#endif
}
template <int N> DEV void pk_acc(float (&r)[N], const float (&x)[N]) {  // r += x, N even
#if AMX_PK_SV
#pragma unroll
  for (int e = 0; e + 1 < N; e += 2) { const f2v t = f2v{r[e], r[e + 1]} + f2v{x[e], x[e + 1]}; r[e] = t.x; r[e + 1] = t.y; }
  if (N & 1) r[N - 1] += x[N - 1];
#else
#pragma unroll
  for (int e = 0; e < N; ++e) r[e] += x[e];
#endif
}
DEV void sv_axpy(sv &a, const sv &c, float q) {  // a += c * q
#if AMX_PK_SV
#pragma unroll
  for (int e = 0; e < 6; e += 2) { const f2v r = pk_fma(f2v{c.v[e], c.v[e + 1]}, f2v{q, q}, f2v{a.v[e], a.v[e + 1]}); a.v[e] = r.x; a.v[e + 1] = r.y; }
#else
#pragma unroll
  for (int e = 0; e < 6; ++e) a.v[e] += c.v[e] * q;
#endif
}
DEV sv inert_mul(const float *i, const sv &v) {
  sv r;
  r.v[0] = i[0] * v.v[0] + i[3] * v.v[1] + i[4] * v.v[2] - i[8] * v.v[4] + i[7] * v.v[5];
  r.v[1] = i[3] * v.v[0] + i[1] * v.v[1] + i[5] * v.v[2] + i[8] * v.v[3] - i[6] * v.v[5];
  r.v[2] = i[4] * v.v[0] + i[5] * v.v[1] + i[2] * v.v[2] - i[7] * v.v[3] + i[6] * v.v[4];
  r.v[3] = i[8] * v.v[1] - i[7] * v.v[2] + i[9] * v.v[3];
  r.v[4] = i[6] * v.v[2] - i[8] * v.v[0] + i[9] * v.v[4];
  r.v[5] = i[7] * v.v[0] - i[6] * v.v[1] + i[9] * v.v[5];
  return r;
}
DEV sv motion_cross(const sv &vel, const sv &v) {
  f3 w = mk3(vel.v[0], vel.v[1], vel.v[2]), l = mk3(vel.v[3], vel.v[4], vel.v[5]);
  f3 a = mk3(v.v[0], v.v[1], v.v[2]), b = mk3(v.v[3], v.v[4], v.v[5]);
  f3 r0 = cross(w, a), r1 = cross(w, b) + cross(l, a);
  sv r = {{r0.x, r0.y, r0.z, r1.x, r1.y, r1.z}};
  return r;
}
DEV sv motion_cross_force(const sv &vel, const sv &f) {
  f3 w = mk3(vel.v[0], vel.v[1], vel.v[2]), l = mk3(vel.v[3], vel.v[4], vel.v[5]);
  f3 a = mk3(f.v[0], f.v[1], f.v[2]), b = mk3(f.v[3], f.v[4], f.v[5]);
  f3 r0 = cross(w, a) + cross(l, b), r1 = cross(w, b);
  sv r = {{r0.x, r0.y, r0.z, r1.x, r1.y, r1.z}};
  return r;
}

// ---- wave-level primitives (wave64)
DEV float readlane_f(float x, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), lane)); }
template <int CTRL> DEV float dpp_f(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true));
}
// Sum over the 64 lanes, result uniform: quad_perm xor1, xor2, row_half_mirror, row_mirror inside each
// 16-lane row (DPP, no LDS), then the four row totals are combined through v_readlane.
DEV float wave_sum(float x) {
  x += dpp_f<0xB1>(x);   // quad_perm [1,0,3,2]
  x += dpp_f<0x4E>(x);   // quad_perm [2,3,0,1]
  x += dpp_f<0x141>(x);  // row_half_mirror
  x += dpp_f<0x140>(x);  // row_mirror
  return (readlane_f(x, 0) + readlane_f(x, 16)) + (readlane_f(x, 32) + readlane_f(x, 48));
}

// ---- group-level primitives: a group is G consecutive lanes (G = 16, 32, 64)
template <int G> DEV float gsum(float x) {  // every lane gets the sum over its group
  x += dpp_f<0xB1>(x);
  x += dpp_f<0x4E>(x);
  x += dpp_f<0x141>(x);
  x += dpp_f<0x140>(x);  // 16-lane row total in every lane of the row
  // (Measured alternative, r02: rows exchanged on the VALU with v_permlane16_swap / v_permlane32_swap instead - 1 % slower on
  // both models: at 2 waves/SIMD the VALU is 55-65 % busy and the crossbar is not the scarcer unit, profiles/r02_ab_*.)
  if (G >= 32) x += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(x), 0x401F));  // lane ^ 16 (crossbar, no LDS memory)
  if (G >= 64) x += __shfl_xor(x, 32);
  return x;
}
// ---- packed group sums (r03).  A group sum costs five cross-lane additions whatever it sums, and the line search needs six
// at a time (two per trial point, three trial points per bracketing iteration).  Packed butterfly: DPP bank masks enable the
// four QUADS (lanes 4q..4q+3) of a 16-lane row separately, so the first stage (row_half_mirror: quad 0 <-> 1, 2 <-> 3) adds
// value A into quads 0 and 3 and value B into quads 1 and 2 of ONE register, the second stage (row_mirror: quad 0 <-> 3,
// 1 <-> 2) folds two such registers into one - quad q of a row then carries one of four values - and the stages inside the quad
// (xor 1, xor 2) and the row exchange run once per register instead of once per value: 15 vector instructions + 2 crossbar
// operations for six sums instead of 30 + 6.  On return every lane holds the group total of the value its quad selects;
// quad_value<q> reads quad q's (a row broadcast, folded into the consuming add).
// The DPP additions are inline assembly (the compiler's DPP combiner does not form bank-masked additions); the two wait states
// a DPP read needs after a vector write of its source are spelled out: a leading s_nop 1 for the inputs, and inside the block
// every source was written at least two instructions earlier (s_nop 0 where a chain has only one independent neighbour).
template <int Q> DEV float quad_value(float x) { return dpp_f<0x150 + 4 * Q>(x); }  // row_newbcast: lane 4Q of the lane's row
template <int G> DEV float rows_total(float x) {  // the cross-row steps of gsum
  if (G >= 32) x += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(x), 0x401F));
  if (G >= 64) x += __shfl_xor(x, 32);
  return x;
}
// u: quads 0, 1, 2, 3 -> sums of a0, b0, b1, a1;  w: quads 0, 1 -> sums of a2, b2 (quads 3, 2 mirror them)
#define AMX_PACK6_ASM(TAIL)                                                                          \
  asm("s_nop 1\n\t"                                                                                    \
      "v_add_f32_dpp %[ra], %[a0], %[a0] row_half_mirror row_mask:0xf bank_mask:0x9\n\t"               \
      "v_add_f32_dpp %[rb], %[a1], %[a1] row_half_mirror row_mask:0xf bank_mask:0x9\n\t"               \
      "v_add_f32_dpp %[rc], %[a2], %[a2] row_half_mirror row_mask:0xf bank_mask:0x9\n\t"               \
      "v_add_f32_dpp %[ra], %[b0], %[b0] row_half_mirror row_mask:0xf bank_mask:0x6\n\t"               \
      "v_add_f32_dpp %[rb], %[b1], %[b1] row_half_mirror row_mask:0xf bank_mask:0x6\n\t"               \
      "v_add_f32_dpp %[rc], %[b2], %[b2] row_half_mirror row_mask:0xf bank_mask:0x6\n\t"               \
      "v_add_f32_dpp %[u], %[ra], %[ra] row_mirror row_mask:0xf bank_mask:0x3\n\t"                     \
      "v_add_f32_dpp %[u], %[rb], %[rb] row_mirror row_mask:0xf bank_mask:0xc\n\t"                     \
      "v_add_f32_dpp %[w], %[rc], %[rc] row_mirror row_mask:0xf bank_mask:0xf\n\t"                     \
      "s_nop 0\n\t"                                                                                    \
      "v_add_f32_dpp %[u], %[u], %[u] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"              \
      "v_add_f32_dpp %[w], %[w], %[w] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"              \
      "s_nop 0\n\t"                                                                                    \
      "v_add_f32_dpp %[u], %[u], %[u] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"              \
      "v_add_f32_dpp %[w], %[w], %[w] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" TAIL               \
      : [ra] "=&v"(ra), [rb] "=&v"(rb), [rc] "=&v"(rc), [u] "=&v"(u), [w] "=&v"(w)                      \
      : [a0] "v"(a0), [b0] "v"(b0), [a1] "v"(a1), [b1] "v"(b1), [a2] "v"(a2), [b2] "v"(b2))
#define AMX_PACK2_ASM(TAIL)                                                                          \
  asm("s_nop 1\n\t"                                                                                    \
      "v_add_f32_dpp %[r], %[a], %[a] row_half_mirror row_mask:0xf bank_mask:0x9\n\t"                  \
      "v_add_f32_dpp %[r], %[b], %[b] row_half_mirror row_mask:0xf bank_mask:0x6\n\t"                  \
      "s_nop 1\n\t"                                                                                    \
      "v_add_f32_dpp %[r], %[r], %[r] row_mirror row_mask:0xf bank_mask:0xf\n\t"                       \
      "s_nop 1\n\t"                                                                                    \
      "v_add_f32_dpp %[r], %[r], %[r] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"              \
      "s_nop 1\n\t"                                                                                    \
      "v_add_f32_dpp %[r], %[r], %[r] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" TAIL               \
      : [r] "=&v"(r) : [a] "v"(a), [b] "v"(b))
// 16-lane groups have no row exchange behind the block: its results go straight to the consumers' DPP reads (quad_value), which
// the compiler cannot pad for - it does not know the block's last instructions wrote them - so the block ends with the two
// wait states itself; likewise in front of the lane swaps of the 32-lane row exchange
#ifndef AMX_TREE_PACK
#define AMX_TREE_PACK 7  // bits: 1 COM, 2 composite inertia, 4 force totals; 0: the tree phases' root sums (COM, composite inertia, force totals) as plain group sums (A/B switch)
#endif
#ifndef AMX_PACK_SWIZZLE
#define AMX_PACK_SWIZZLE 0  // 1: the packed sums exchange their rows through the LDS crossbar like gsum (A/B switch)
#endif
template <int G> DEV void gsum_pack6(float a0, float b0, float a1, float b1, float a2, float b2, float &u, float &w) {
  float ra, rb, rc;
  if constexpr (G == 16 || (G == 32 && !AMX_PACK_SWIZZLE)) AMX_PACK6_ASM("\n\ts_nop 1");  // the consumer is a DPP read / a lane swap: see above
  else AMX_PACK6_ASM("");
  if constexpr (G == 32 && !AMX_PACK_SWIZZLE) {
    // Row exchange on the vector unit: the line search is a chain of dependent latencies (r03 phase profile: 1250 cycles per
    // bracketing iteration for 100 vector instructions), and a crossbar round trip is the longest link in it.  One swap packs
    // the two registers by row - (u.row0, w.row0) and (u.row1, w.row1) - one add gives u's total in row 0 and w's in row 1,
    // a second swap hands both totals to both rows.
    auto s1 = __builtin_amdgcn_permlane16_swap(__float_as_uint(u), __float_as_uint(w), false, false);
    const float t = __uint_as_float(s1[0]) + __uint_as_float(s1[1]);
    auto s2 = __builtin_amdgcn_permlane16_swap(__float_as_uint(t), __float_as_uint(t), false, false);
    u = __uint_as_float(s2[0]);
    w = __uint_as_float(s2[1]);
  } else {
    u = rows_total<G>(u);
    w = rows_total<G>(w);
  }
}
#define AMX_PACK4_ASM(TAIL)                                                                          \
  asm("s_nop 1\n\t"                                                                                    \
      "v_add_f32_dpp %[ra], %[a0], %[a0] row_half_mirror row_mask:0xf bank_mask:0x9\n\t"               \
      "v_add_f32_dpp %[rb], %[a1], %[a1] row_half_mirror row_mask:0xf bank_mask:0x9\n\t"               \
      "v_add_f32_dpp %[ra], %[b0], %[b0] row_half_mirror row_mask:0xf bank_mask:0x6\n\t"               \
      "v_add_f32_dpp %[rb], %[b1], %[b1] row_half_mirror row_mask:0xf bank_mask:0x6\n\t"               \
      "s_nop 0\n\t"                                                                                    \
      "v_add_f32_dpp %[u], %[ra], %[ra] row_mirror row_mask:0xf bank_mask:0x3\n\t"                     \
      "v_add_f32_dpp %[u], %[rb], %[rb] row_mirror row_mask:0xf bank_mask:0xc\n\t"                     \
      "s_nop 1\n\t"                                                                                    \
      "v_add_f32_dpp %[u], %[u], %[u] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"              \
      "s_nop 1\n\t"                                                                                    \
      "v_add_f32_dpp %[u], %[u], %[u] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" TAIL               \
      : [ra] "=&v"(ra), [rb] "=&v"(rb), [u] "=&v"(u)                                                    \
      : [a0] "v"(a0), [b0] "v"(b0), [a1] "v"(a1), [b1] "v"(b1))
// u: quads 0, 1, 2, 3 -> sums of a0, b0, b1, a1
template <int G> DEV float gsum_pack4(float a0, float b0, float a1, float b1) {
  float ra, rb, u;
  if constexpr (G == 16 || (G == 32 && !AMX_PACK_SWIZZLE)) AMX_PACK4_ASM("\n\ts_nop 1");
  else AMX_PACK4_ASM("");
  if constexpr (G == 32 && !AMX_PACK_SWIZZLE) {
    auto s1 = __builtin_amdgcn_permlane16_swap(__float_as_uint(u), __float_as_uint(u), false, false);
    return __uint_as_float(s1[0]) + __uint_as_float(s1[1]);
  }
  return rows_total<G>(u);
}
// r: quads 0, 1 -> sums of a, b (quads 3, 2 mirror them)
template <int G> DEV float gsum_pack2(float a, float b) {
  float r;
  if constexpr (G == 16 || (G == 32 && !AMX_PACK_SWIZZLE)) AMX_PACK2_ASM("\n\ts_nop 1");
  else AMX_PACK2_ASM("");
  if constexpr (G == 32 && !AMX_PACK_SWIZZLE) {
    auto s1 = __builtin_amdgcn_permlane16_swap(__float_as_uint(r), __float_as_uint(r), false, false);  // (row 0, row 0), (row 1, row 1)
    return __uint_as_float(s1[0]) + __uint_as_float(s1[1]);
  }
  return rows_total<G>(r);
}
// value of lane `j` of my group (j compile-time or wave-uniform); gb4 = 4 * (first lane of my group)
template <int G> DEV float gbcast(float x, int j, int gb4) {
  if (G == 64) return readlane_f(x, j);
  // G < 64: through the LDS crossbar.  (Measured alternative for G = 32, two v_readlane + a select on the VALU: 12 % slower.)
  return __int_as_float(__builtin_amdgcn_ds_bpermute(gb4 + 4 * j, __float_as_int(x)));
}
// value of lane J (compile-time) of my group.  Narrow groups use ds_swizzle in bit-mask mode - lane' = (lane & and) | or
// inside each 32-lane half, an immediate pattern and no address register: a wave issues one per ~13 cycles against ~24
// for ds_bpermute (tools/bperm_rate.hip), which is what bounds the eliminations.
template <int G, int J> DEV float gbcast_s(float x) {
  if (G == 64) return readlane_f(x, J);
  constexpr int pattern = G == 32 ? (J << 5) : (0x10 | (J << 5));  // [4:0] and-mask, [9:5] or-mask, [14:10] xor-mask
  return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(x), pattern));
}
DEV bool wave_any(bool p) { return __ballot(p) != 0ull; }

// Diagnostic build only (-DAMX_PROFILE, libamx_prof.so): per-phase cycle shares via s_memtime.  Never in the product build.
#ifdef AMX_PROFILE
#define PROF_N 32
// The counters live in HBM (one row per environment, fire-and-forget atomic adds from the group's lane 0): the slab and the
// launch plan are those of the product build, so the specialised kernels are the ones profiled.
#define PROF_DECL unsigned prof_t; float *prof_row;
#define PROF(id) do { unsigned t_ = (unsigned)__builtin_amdgcn_s_memtime(); if (gl == 0 && prof_row) unsafeAtomicAdd(prof_row + (id), (float)(t_ - prof_t)); prof_t = t_; } while (0)
// inside the free solve routines: the caller's clock and counter row travel through a context pointer (last argument)
struct ProfCtx { unsigned *t; float *row; };
#define PROF_CTX(pc, id) do { if (pc) { unsigned t_ = (unsigned)__builtin_amdgcn_s_memtime(); if (gl == 0 && (pc)->row) unsafeAtomicAdd((pc)->row + (id), (float)(t_ - *(pc)->t)); *(pc)->t = t_; } } while (0)
#else
#define PROF_DECL
#define PROF(id)
struct ProfCtx {};
#define PROF_CTX(pc, id)
#endif

// ---- register-row dense Cholesky: lane i holds row i of the matrix at lds[A_off] (lower part).
// A is full symmetric [n][ld]; the factor is packed lower-triangular (row i at TRI(i)).  Solves A x = rhs.  L is written to lds[L_off] (may equal A_off: each lane overwrites only
// its own row after reading it); the two substitutions re-read it masked (strict lower rows, then
// strict lower columns), so every lane runs the same 3-instruction update per pivot.
// `factor == false` reuses the factor already stored at L_off (CG preconditioner).
// Pivot reciprocals use v_rsq_f32 + one Newton step (<= 1 ulp) instead of the IEEE sqrt/divide expansions.
// (see chol_solve_rows below)
#define TRI(i) (((i) * ((i) + 1)) >> 1)  // offset of row i in packed lower-triangular storage

template <int N, int G>
__device__ __noinline__ void chol_solve_rows(int A_off, int ld, int n, int L_off, int rhs_off, int x_off, int vstride, bool factor, int gl, int gb4) {
  float a[N];
  const int li = gl < n ? gl : 0, row = TRI(li);
  if (factor) {
#pragma unroll
    for (int k = 0; k < N; ++k) {
      const float v = lds[A_off + li * ld + (k <= li ? k : li)];  // A: full symmetric [n][ld]; L: packed lower-triangular
      a[k] = (gl < n && k <= gl) ? v : (k == gl ? 1.0f : 0.0f);
    }
#pragma unroll
    for (int j = 0; j < N; ++j) {
      const float ajj = fmaxf(gbcast<G>(a[j], j, gb4), MINVAL);
      float inv = __builtin_amdgcn_rsqf(ajj);
      inv = inv * __builtin_fmaf(-0.5f * ajj * inv, inv, 1.5f);
      const float lij = a[j] * inv;
      a[j] = lij;
#pragma unroll
      for (int k = j + 1; k < N; ++k) a[k] = __builtin_fmaf(-lij, gbcast<G>(lij, k, gb4), a[k]);
    }
    if (gl < n) {
#pragma unroll
      for (int k = 0; k < N; ++k)
        if (k <= gl) lds[L_off + row + k] = a[k];
    }
    WSYNC();
  }
  const float dinv = 1.0f / lds[L_off + row + li];
#pragma unroll
  for (int k = 0; k < N; ++k) {  // strict lower part of row `gl`
    const float v = lds[L_off + row + (k < li ? k : li)];
    a[k] = (gl < n && k < gl) ? v : 0.0f;
  }
  float b = lds[rhs_off + li * vstride];
  b = gl < n ? b : 0.0f;
#pragma unroll
  for (int j = 0; j < N; ++j) {  // forward: L y = rhs; lane j's b is final when pivot j is reached
    const float yj = gbcast<G>(b * dinv, j, gb4);
    b = __builtin_fmaf(-a[j], yj, b);
  }
  b *= dinv;
#pragma unroll
  for (int k = 0; k < N; ++k) {  // strict lower part of column `gl`: L[k][gl], k > gl
    const float v = lds[L_off + TRI(k < n ? k : 0) + (k < n && k > li ? li : 0)];
    a[k] = (gl < n && k > gl && k < n) ? v : 0.0f;
  }
#pragma unroll
  for (int j = N - 1; j >= 0; --j) {  // backward: L' x = y
    const float xj = gbcast<G>(b * dinv, j, gb4);
    b = __builtin_fmaf(-a[j], xj, b);
  }
  if (gl < n) lds[x_off + gl * vstride] = b * dinv;
  WSYNC();
}

// ---- one-shot SPD solve by Gauss-Jordan elimination, one matrix row per lane in registers.
// Used where the factor is not reused (qacc_smooth, the Newton direction, the Euler damping solve).
// Pivot j: every lane fetches row j's remaining entries through one crossbar hop (they are also column j
// by symmetry of the Schur complement only for rows below, so the pivot ROW is what is broadcast), scales
// its own column-j entry by 1/pivot and updates its row and right-hand side.  No substitutions, no LDS
// traffic besides the initial row load, and one cross-lane latency per pivot instead of two plus 2n
// serial substitution steps for Cholesky.
// Pivot-row broadcast: v_readlane for one group per wave (G == 64), the LDS crossbar for narrower groups (one instruction
// serves every group of the wave) - as ds_swizzle with an immediate pattern, since the pivot lane is a compile-time
// constant of the unrolled elimination.  The crossbar issue rate of a wave is what bounds an elimination; sending part of
// each row through v_readlane + exec-masked v_fmac instead was measured three times in this round and always lost in the
// kernel (profiles/r01_mfma_probe.txt).
template <int N, int G, int J> DEV void gj_pivot(float (&a)[N], float &b, int gl, int gb4) {
  float u[N];
  const float piv = gbcast_s<G, J>(a[J]);
  const float bj = gbcast_s<G, J>(b);
#pragma unroll
  for (int k = J + 1; k < N; ++k) u[k] = gbcast_s<G, J>(a[k]);
  float f = a[J] * __builtin_amdgcn_rcpf(fmaxf(piv, MINVAL));
  f = gl == J ? 0.0f : f;  // the pivot row itself is left alone
  b = __builtin_fmaf(-f, bj, b);
#pragma unroll
  for (int k = J + 1; k < N; ++k) a[k] = __builtin_fmaf(-f, u[k], a[k]);
}
template <int N, int G, int J> struct GJ {
  static DEV void run(float (&a)[N], float &b, int gl, int gb4) {
    gj_pivot<N, G, J>(a, b, gl, gb4);
    GJ<N, G, J + 1>::run(a, b, gl, gb4);
  }
};
template <int N, int G> struct GJ<N, G, N> { static DEV void run(float (&)[N], float &, int, int) {} };

typedef float v32f __attribute__((ext_vector_type(32)));
#ifndef AMX_MFMA_ALL_ROWS
#define AMX_MFMA_ALL_ROWS 0  // 1: every pyramid row of an active contact goes through the matrix cores (A/B switch)
#endif

// H = M + J' D J on the matrix cores, for one wave (G = 32: the wave's two environments are the two 32x32 blocks of
// v_mfma_f32_32x32x1_2b_f32; G = 64: block 1 idles).  Lane (b, j) feeds block b with column j of one contact-Jacobian row
// per instruction: A = D_r J_r[j], B = J_r[j], a rank-1 update of H_b.  The accumulator holds H_b[8q+4h+r][j] in register
// 16b+4q+r of lane (h, j).  H is symmetric, so column j is row j: one v_permlane32_swap per register pair hands lane
// (b, j) the entries of row j of its own environment as ready-made 16-byte chunks (columns 8q..8q+3 and 8q+4..8q+7),
// which are added to the chunk of M and stored as row j of the full matrix H.  Out of line: the 32 accumulator
// registers stay out of the caller's and the solver's register budgets.
//   crows: active contact rows (bit 4c+r), OR-ed over the environments of the wave; an inactive row has D = 0 anyway,
//          the mask only skips contacts that are inactive everywhere.
typedef float v16f __attribute__((ext_vector_type(16)));
// The matrix-core part: on return lane (b, j) holds row j of its environment's J'DJ as sixteen-byte chunks,
// lo[q][r] = (J'DJ)[j][8q+r], hi[q][r] = (J'DJ)[j][8q+4+r].  crows, ldc, slot_base are wave-uniform scalars.
// COL31 (nv <= 31): lane 31 of each block (64-lane groups: of each K slice) feeds A = field 1 of the row record (the row's force f_r) instead of
// D_r J_r[31] = 0, so row 31 of the product is sum_r f_r J_r[i] = the contact rows' part of J'f = qfrc_constraint.  Lane i
// holds column i of the accumulator: the entry reaches it as hi[3][3] with its Hessian row, for four more broadcast loads per
// contact.
// AUGN >= 0 (32-lane groups): the accumulator starts from *cinit instead of zero, and every row's Jacobian is extended by one
// entry at index AUGN holding the row's residual Jaref (field 2 of its record), so that row / column AUGN of the product collect
// sum_r D_r J_r[k] Jaref_r = -(J' f)[k] next to whatever cinit put there (see newton_solve_fused)
// JFR (planner: jfr, r05): the Jacobian block holds three rows per contact - normal, mu * tangent 1, mu * tangent 2 - and a pyramid row's entry
// is their sum or difference, formed here: three loads and four additions per contact instead of four loads
template <int G, bool COL31 = false, int AUGN = -1, bool JFR = false>
DEV void jtdj_rows_mfma(unsigned long long crows, int row_base, int J_base, int ldc, int slot_base, int gl, unsigned (&lo)[4][4], unsigned (&hi)[4][4],
                        const v32f *cinit = nullptr, unsigned long long crows2 = 0ull) {
  const int j = gl & 31, h = gl >> 5;
  // one bit per contact that is active in some environment of the wave (wave-uniform scalar loop over the set bits)
  unsigned long long cm = (crows | (crows >> 1) | (crows >> 2) | (crows >> 3)) & 0x1111111111111111ull;
  if (G == 32) {
    v32f C = {};
    if constexpr (AUGN >= 0) C = *cinit;
    // the first active contact starts the accumulator from the instruction's constant 0 operand: clearing 32 registers
    // with v_mov costs as much VALU time as two more contacts' operand loads
#ifndef AMX_MFMA_PREFETCH
#define AMX_MFMA_PREFETCH 1  // 0: every contact's operands are loaded right in front of its own matrix instructions (r03), A/B switch
#endif
#ifdef AMX_EXPERIMENT_DOUBLE_MFMA
    v32f C2 = {};
#endif
    // one contact's operands: D and J[j] of its four pyramid rows (+ the row's force / residual for COL31 / AUGN)
    struct Ops { float d[4], jv[4], g[4]; bool has; unsigned rbits; };
    auto load = [&]() {
      Ops o;
      const int c = __builtin_ctzll(cm) >> 2;
      cm &= cm - 1ull;
      const int slot = CI(slot_base + 32 * c + j);
      o.has = slot >= 0;
      const int Jr = J_base + (JFR ? 3 : 4) * c * ldc + (o.has ? slot : 0), Dr = row_base + 4 * c * ROW_E;
      if constexpr (JFR) { o.jv[0] = lds[Jr]; o.jv[1] = lds[Jr + ldc]; o.jv[2] = lds[Jr + 2 * ldc]; o.jv[3] = 0.0f; }  // normal, mu t1, mu t2: combined at issue
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        o.d[r] = lds[Dr + r * ROW_E];
        if constexpr (!JFR) o.jv[r] = lds[Jr + r * ldc];
        o.g[r] = 0.0f;
        if (COL31) o.g[r] = AMX_SQRT_D ? -lds[Dr + r * ROW_E + 2] : lds[Dr + r * ROW_E + 1];  // sqrt form: the force f = -D Jaref as (-Jaref sqrt D) sqrt D
        if (AUGN >= 0) o.g[r] = lds[Dr + r * ROW_E + 2];
      }
      // a pyramid row that is inactive in both environments has D = 0 and f = 0 everywhere: its instruction (64 cycles of the
      // matrix pipe, on the dependent chain) is skipped on a scalar test of its bit - except the one that starts the accumulator
      o.rbits = (unsigned)(crows >> (4 * c)) & 15u;
      return o;
    };
    auto issue = [&](const Ops &o, auto first) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float jr_ = JFR ? ((r & 1) ? o.jv[0] - o.jv[1 + (r >> 1)] : o.jv[0] + o.jv[1 + (r >> 1)]) : o.jv[r];
        const float v0 = o.has ? jr_ : 0.0f;
#if AMX_SQRT_D
        // o.d is sqrt(D) (constraint_force_and_grad publishes it): A = B = sqrt(D) J - the same bits on both sides of every product
        const float v = o.d[r] * ((AUGN >= 0 && j == AUGN) ? o.g[r] : ((COL31 && j == 31) ? 0.0f : v0));
        const float aop = (COL31 && j == 31) ? o.d[r] * o.g[r] : v;
#else
        const float v = (AUGN >= 0 && j == AUGN) ? o.g[r] : v0;
        const float aop = (COL31 && j == 31) ? o.g[r] : o.d[r] * v;
#endif
        if (AUGN < 0 && decltype(first)::value && r == 0) { const v32f Z = {}; C = __builtin_amdgcn_mfma_f32_32x32x1f32(aop, v, Z, 0, 0, 0); }
        else if (AMX_MFMA_ALL_ROWS || ((o.rbits >> r) & 1u)) {
          C = __builtin_amdgcn_mfma_f32_32x32x1f32(aop, v, C, 0, 0, 0);
#ifdef AMX_EXPERIMENT_DOUBLE_MFMA  // timing experiment (same results): every row's instruction issued twice, the copy into a dead accumulator
          C2 = __builtin_amdgcn_mfma_f32_32x32x1f32(aop, v, C2, 0, 0, 0);
#endif
        }
      }
    };
    if (cm != 0ull) {
#if AMX_MFMA_PREFETCH
      // software pipeline over the active contacts (r04): the loads of contact k + 1 are issued in front of the matrix instructions
      // of contact k - with every contact of a standing robot active, a contact was an LDS round trip (~200 cycles) followed by
      // its four 64-cycle instructions, eight times per Hessian (phase profile: 3.8 k cycles per Hessian for 2.0 k of matrix pipe)
      Ops cur = load();
      bool more = cm != 0ull;
      if (more) {
        Ops nxt = load();
        issue(cur, std::true_type{});
        cur = nxt;
        while (cm != 0ull) {
          nxt = load();
          issue(cur, std::false_type{});
          cur = nxt;
        }
        issue(cur, std::false_type{});
      } else {
        issue(cur, std::true_type{});
      }
#else
      { const Ops o = load(); issue(o, std::true_type{}); }
      while (cm != 0ull) { const Ops o = load(); issue(o, std::false_type{}); }
#endif
    }
#ifdef AMX_EXPERIMENT_DOUBLE_MFMA
    asm volatile("" :: "v"(C2));
#endif
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        // new vdst: own lower half + the other register's lower half in the upper lanes = rows 8q+r of the lane's block;
        // new vsrc: rows 8q+4+r
        auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(C[4 * q + r]), __float_as_uint(C[16 + 4 * q + r]), false, false);
        lo[q][r] = sw[0]; hi[q][r] = sw[1];
      }
  } else {
    // one environment per wave: the single-block K = 2 instruction takes two Jacobian rows at a time (lanes 32-63 feed the
    // second), two instructions per contact
    static_assert(!JFR || G == 32, "three-row Jacobian blocks: 32-lane groups");
    v16f C = {};
    // (two contact-row slots per lane, r04: rows 64 .. 127 = contacts 16 .. 31 arrive as a second mask)
    unsigned long long cm2 = (crows2 | (crows2 >> 1) | (crows2 >> 2) | (crows2 >> 3)) & 0x1111111111111111ull;
    while ((cm | cm2) != 0ull) {
      const bool second = cm == 0ull;
      const int c = second ? 16 + (__builtin_ctzll(cm2) >> 2) : (__builtin_ctzll(cm) >> 2);
      if (second) cm2 &= cm2 - 1ull; else cm &= cm - 1ull;
      const int slot = CI(slot_base + 32 * c + j);
      const bool has = slot >= 0;
      const int Jr = J_base + (4 * c + h) * ldc + (has ? slot : 0), Dr = row_base + (4 * c + h) * ROW_E;
      float d[2], jv[2], g[2];
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        d[s2] = lds[Dr + 2 * s2 * ROW_E]; jv[s2] = lds[Jr + 2 * s2 * ldc];
        g[s2] = 0.0f;
        if (COL31) g[s2] = AMX_SQRT_D ? -lds[Dr + 2 * s2 * ROW_E + 2] : lds[Dr + 2 * s2 * ROW_E + 1];
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        // COL31 (r04, as in the two-block form): lane 31 of either K slice feeds the row's force as its A entry - row 31 of the product is J'f
#if AMX_SQRT_D
        const float v = d[s2] * (has ? jv[s2] : 0.0f);
        const float aop = (COL31 && j == 31) ? d[s2] * g[s2] : v;
#else
        const float v = has ? jv[s2] : 0.0f;
        const float aop = (COL31 && j == 31) ? g[s2] : d[s2] * v;
#endif
        C = __builtin_amdgcn_mfma_f32_32x32x2f32(aop, v, C, 0, 0, 0);
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        // lanes 0-31 keep rows 8q+r and receive rows 8q+4+r from lanes 32-63
        auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(C[4 * q + r]), 0u, false, false);
        lo[q][r] = sw[0]; hi[q][r] = sw[1];
      }
  }
}

template <int G>
__device__ __noinline__ void build_H_mfma(int M_off, int H_off, int ld_, int n_, unsigned long long crows_, int ncon, int row_base, int J_base, int ldc_,
                                          int slot_base_, int gl, unsigned long long crows2_ = 0ull) {
  // arguments of an out-of-line function arrive in vector registers; these are wave-uniform, and saying so keeps the
  // loops and the chunk guards below on the scalar unit
  const int ld = __builtin_amdgcn_readfirstlane(ld_), n = __builtin_amdgcn_readfirstlane(n_), ldc = __builtin_amdgcn_readfirstlane(ldc_);
  const int slot_base = __builtin_amdgcn_readfirstlane(slot_base_);
  const unsigned long long crows = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(crows_ >> 32)) << 32) |
                                   (unsigned)__builtin_amdgcn_readfirstlane((int)crows_);
  const unsigned long long crows2 = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(crows2_ >> 32)) << 32) |
                                    (unsigned)__builtin_amdgcn_readfirstlane((int)crows2_);  // second contact-row slot of a 64-lane group (r05)
  const int j = gl & 31;
  unsigned lo[4][4], hi[4][4];  // after the swaps: lo[q][r] = H[j][8q+r], hi[q][r] = H[j][8q+4+r] of the lane's environment
  const bool writer = (G == 32 || gl < 32) && j < n;
  const int jr = (writer ? j : 0) * ld;
  const float4 *mrow = reinterpret_cast<const float4 *>(lds + M_off + jr);
  float4 *hrow = reinterpret_cast<float4 *>(lds + H_off + jr);
  const int nch = ld >> 2;  // 16-byte chunks per row
  jtdj_rows_mfma<G>(crows, row_base, J_base, ldc, slot_base, gl, lo, hi, nullptr, crows2);
  // H row = M row + J'DJ row, two halves of four chunks: the four loads of a half are in flight together
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    if (4 * half >= nch) break;  // wave-uniform
    float4 mc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { const int c = 4 * half + k; mc[k] = mrow[c < nch ? c : 0]; }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = 4 * half + k, q = c >> 1;
      const unsigned *src = (c & 1) ? hi[q] : lo[q];
      float4 v = mc[k];
      v.x += __uint_as_float(src[0]); v.y += __uint_as_float(src[1]); v.z += __uint_as_float(src[2]); v.w += __uint_as_float(src[3]);
      if (writer && c < nch) hrow[c] = v;
    }
  }
  WSYNC();
}

// (defined behind the symmetric eliminations below: the generic solves use them since r05)
template <int N, int G> constexpr bool dense_sym_ok();
template <int N, int G> DEV float dense_sym_solve(float (&a)[N], float b, int gl);
template <int N, int G>
__device__ __noinline__ void spd_solve_rows(int A_off, int ld, int n, int rhs_off, int x_off, int vstride, int gl, int gb4, int rot) {
  float a[N];
  const int li = gl < n ? gl : 0, nch = ld >> 2;
  // row li of the full symmetric matrix: 16-byte aligned, ld >= ceil4(n) floats long, so whole b128 chunks are in bounds;
  // chunk c sits at position (c + rot) mod nch (rot = 0 unless the rows are stored rotated)
  const float4 *rowp = reinterpret_cast<const float4 *>(lds + A_off + li * ld);
#pragma unroll
  for (int c = 0; c < (N + 3) / 4; ++c) {
    float4 v = {0.0f, 0.0f, 0.0f, 0.0f};
    int pc = c + rot;
    pc -= pc >= nch ? nch : 0;
    if (4 * c < ld) v = rowp[4 * c < ld ? pc : 0];
    const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k = 4 * c + q;
      if (k < N) a[k] = (gl < n && k < n) ? e[q] : (k == gl ? 1.0f : 0.0f);
    }
  }
  float b = lds[rhs_off + li * vstride];
  b = gl < n ? b : 0.0f;
  if constexpr (dense_sym_ok<N, G>()) {
    const float x = dense_sym_solve<N, G>(a, b, gl);
    if (gl < n) lds[x_off + gl * vstride] = x;
    WSYNC();
    return;
  }
  GJ<N, G, 0>::run(a, b, gl, gb4);
  // the matrix is now diagonal: x_i = b_i / a_ii.  a[gl] is a runtime register index, so select it.
  float d = 1.0f;
#pragma unroll
  for (int k = 0; k < N; ++k) d = (k == gl) ? a[k] : d;
  if (gl < n) lds[x_off + gl * vstride] = b / d;
  WSYNC();
}

// The same solve for a compile-time spec: N = nv exactly, offsets are immediates, and - TREE - the elimination follows the
// dof tree.  M[i][j] != 0 only when j is an ancestor of i (or i of j); H = M + J'DJ keeps that pattern when every constraint
// row couples dofs of one ancestor chain (Spec::tree_H).  Pivots then run from the last dof to the first (leaves before
// their ancestors): when pivot J is reached, row J holds nothing but its diagonal and its ancestors' columns, so only those
// depth(J) entries travel through the crossbar and only they are updated - no fill-in, ever (MuJoCo's own L'DL order).
// Quadruped: 135 broadcasts per solve instead of 207; biped: 291 instead of 495; eliminations of different limbs are
// independent chains the scheduler interleaves.
template <class S, int G, int J, int K> struct GJAnc {  // K walks J's ancestor chain
  static DEV void run(float (&a)[S::nv], float f) {
    a[K] = __builtin_fmaf(-f, gbcast_s<G, J>(a[K]), a[K]);
    GJAnc<S, G, J, S::dof_parent[K]>::run(a, f);
  }
};
template <class S, int G, int J> struct GJAnc<S, G, J, -1> { static DEV void run(float (&)[S::nv], float) {} };
template <class S, int G, int J> struct GJTree {
  static DEV void run(float (&a)[S::nv], float &b, int gl) {
    // (Measured alternative, r02: every lane scales its row by the reciprocal of its own entry J before the hop, so the pivot
    // itself never travels - one crossbar operation less per pivot, one multiply more per entry: -1.7 % quadruped, +0.4 % biped.)
    const float piv = gbcast_s<G, J>(a[J]);
    const float bj = gbcast_s<G, J>(b);
    float f = a[J] * __builtin_amdgcn_rcpf(fmaxf(piv, MINVAL));
    f = gl == J ? 0.0f : f;  // the pivot row itself is left alone
    b = __builtin_fmaf(-f, bj, b);
    GJAnc<S, G, J, S::dof_parent[J]>::run(a, f);
    GJTree<S, G, J - 1>::run(a, b, gl);
  }
};
template <class S, int G> struct GJTree<S, G, -1> { static DEV void run(float (&)[S::nv], float &, int) {} };


// ---- the static eliminations without the crossbar (r02b): groups of 16 or 32 lanes, nv < G.
// The Schur complement of the dofs not yet eliminated is symmetric, so the pivot ROW J is also what register a[J] (column J)
// holds across the lanes: entry (J, K) sits in lane K.  One register therefore carries everything pivot J needs - the
// pivot itself (lane J), the lane's own multiplier (its own entry) and every pivot-row entry (lane K) - and a DPP
// row_newbcast modifier on the multiply-add fetches lane K of a 16-lane row for free.  Groups of 32 lanes first make both
// rows of a group hold the wanted source row: one v_permlane16_swap of the register with itself returns (row 0, row 0) and
// (row 1, row 1).  Per pivot: two copies + one swap + max + rcp + multiply + select + one v_fmac_f32_dpp per updated
// column, no LDS instruction and a dependent chain of VALU latencies instead of two crossbar round trips (~60 cycles per
// pivot instead of ~450 measured).  The right-hand side rides along as column N of an (N+1) x (N+1) symmetric matrix whose
// row N (lane N, which the caller loads with b through a staging vector) mirrors it, so b_J is lane N of the same
// register (BCOL); 16-lane groups broadcast b directly instead.
// The multiply-adds are inline assembly because the compiler's DPP combiner does not fold a DPP move into v_fma/v_fmac
// (it does into v_rcp / v_max).  The hazard recogniser does not look into inline assembly, so the 2 wait states a DPP read
// needs after a VALU write of the same register are guaranteed by construction: every asm multiply-add consumes `nf`, which
// depends on the compiler-emitted v_max_f32_dpp of the same source registers (the compiler pads that one), and 16-lane
// groups pass the register through an explicit `s_nop 1` first.  tools/check_dpp_hazard.py re-checks the emitted code.
struct RowPair { float r0, r1; };
template <int G> DEV RowPair row_pair(float c) {
  if constexpr (G == 16) {
    return RowPair{c, c};  // the first DPP reader (sym_pivot's v_max_f32_dpp) carries the wait states itself
  } else {
    static_assert(G == 32 || G == 64, "row_pair: 16-, 32- or 64-lane groups (the matrix rows of a 64-lane group sit in its first two DPP rows)");
    const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(c), __float_as_uint(c), false, false);
    return RowPair{__uint_as_float(sw[0]), __uint_as_float(sw[1])};
  }
}
template <int K> DEV float bcast_row(const RowPair &p) { return K < 16 ? dpp_f<0x150 + (K & 15)>(p.r0) : dpp_f<0x150 + (K & 15)>(p.r1); }
// acc += (lane K of my group of p) * nf
template <int K> DEV void fmac_bcast(float &acc, const RowPair &p, float nf) {
  asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(K < 16 ? p.r0 : p.r1), "v"(nf), "n"(K & 15));
}
#ifndef AMX_TRUE_ROW_ELIM
#define AMX_TRUE_ROW_ELIM 0  // 1: the pivot row's entries come from the pivot's own lane (through the crossbar), not from the symmetric column's lanes.
#endif                       // The reduced matrix is symmetric only up to rounding and the substitution multiplies that asymmetry by the multipliers: with the
                             // solver unconverged on a hard impact (every foot corner at once, Newton 3 / 5) the product form sits 1e-5 .. 6e-5 from fp64 after one
                             // step, this form 3e-6 .. 7e-6 like the CPU oracle (profiles/r04_standing_impact.txt) - at 0.85 of the throughput.  Per-model builds
                             // take it with AMX_ACCURATE_ELIM=1 (ambersim_amd/_spec.py).
#if AMX_TRUE_ROW_ELIM
#undef AMX_SYM_EXACT
#define AMX_SYM_EXACT 0  // the true-row form multiplies the pivot row by c_i / pivot
template <class S, int G, int K, int J> struct SymAncT {
  static DEV void run(float (&a)[S::nv + 1], float nf) {
    a[K] = __builtin_fmaf(nf, gbcast_s<G, J>(a[K]), a[K]);
    SymAncT<S, G, S::dof_parent[K], J>::run(a, nf);
  }
};
template <class S, int G, int J> struct SymAncT<S, G, -1, J> { static DEV void run(float (&)[S::nv + 1], float) {} };
#endif
template <class S, int G, int K, int LO = 0> struct SymAnc {  // K walks the pivot's ancestor chain (columns below LO are somebody else's: see ElimDefer)
  static DEV void run(float (&a)[S::nv + 1], const RowPair &p, float nf) {
    if constexpr (K >= LO) {  // (a dof's ancestors have lower indices: the first one below LO ends the walk)
      fmac_bcast<K>(a[K], p, nf);
      SymAnc<S, G, S::dof_parent[K], LO>::run(a, p, nf);
    }
  }
};
template <class S, int G, int LO> struct SymAnc<S, G, -1, LO> { static DEV void run(float (&)[S::nv + 1], const RowPair &, float) {} };
template <class S, int G, int K> struct SymRest {  // dense order: every column after the pivot
  static DEV void run(float (&a)[S::nv + 1], const RowPair &p, float nf) {
    if constexpr (K < S::nv) {
      fmac_bcast<K>(a[K], p, nf);
      SymRest<S, G, K + 1>::run(a, p, nf);
    }
  }
};
#ifndef AMX_PIVOT_VCMP
#define AMX_PIVOT_VCMP 0  // 1: the pivot-lane test as v_cmp per pivot (r02), A/B switch
#endif
#ifndef AMX_PIVOT_RCP_BUILTIN
#define AMX_PIVOT_RCP_BUILTIN 1  // 0: the pivot's reciprocal as assembly with a fixed s_nop 1 in front (A/B switch)
#endif
#ifndef AMX_PIVOT_CLAMP
#define AMX_PIVOT_CLAMP 0  // 1: every pivot clamped to MINVAL before its reciprocal (r02/r03a: one v_max_f32_dpp per pivot), A/B switch
#endif
#ifndef AMX_PIVOT_ASM_MAX
#define AMX_PIVOT_ASM_MAX 1  // 0: the pivot broadcast as builtins (the compiler does not fold the DPP move into v_max here: +1 VALU per pivot, measured)
#endif
// lanes of the wave that are lane J of their group, as the 64-bit constant a v_cndmask takes from a scalar register pair:
// the pivot-lane test of a pivot costs two s_mov of literals (scalar unit, rematerialised at will) instead of a v_cmp
template <int G, int J> constexpr unsigned long long lane_mask() {
  return (G == 16 ? 0x0001000100010001ull : (G == 32 ? 0x0000000100000001ull : 1ull)) << J;
}
DEV float sel_zero(unsigned long long mask, float x) {  // masked lanes 0, the others x
  float o;
  asm("v_cndmask_b32_e64 %0, %1, 0, %2" : "=v"(o) : "v"(x), "s"(mask));
  return o;
}
DEV float sel_mask(unsigned long long mask, float yes, float no) {  // masked lanes `yes`, the others `no`
  float o;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(o) : "v"(no), "v"(yes), "s"(mask));
  return o;
}
typedef float v16acc __attribute__((ext_vector_type(16)));
// NDEF > 0 (r05, tree eliminations of 32- and 64-lane groups): this pivot's updates of the columns 0 .. NDEF - 1 - the root chain every limb
// hangs off: six of a floating base - are not made here.  They are rank-1 terms s_k(J) * (-s_i(J)) of a 16 x 16 product per (environment,
// 16-lane row): one v_mfma_f32_16x16x1_4b_f32 per pivot takes A = the scaled column as row 0's lanes hold it (ps.r0: every 16-lane row has a
// copy, so register k of every block stands for dof k) and B = the lane's own multiplier, and accumulates ALL the deferred pivots' terms on
// the matrix cores; sym_defer_collect adds the sums to a[0 .. NDEF - 1] before the first pivot that reads one of those columns.  Same
// products, same bits on both sides of the diagonal (s_k * -s_i = s_i * -s_k), summed in pivot order: the reduced matrix stays bitwise
// symmetric.  Biped: 144 of an elimination's 254 multiply-adds leave the vector unit for 24 matrix instructions and 20 lane swaps / adds.
template <class S, int G, int J, bool TREE, bool BCOL, int NDEF = 0, bool FIRST = false>
DEV void sym_pivot(float (&a)[S::nv + 1], int gl, float &diag, unsigned long long pm0, unsigned long long &pm, v16acc *cdef = nullptr, unsigned long long rowm = 0ull) {
  constexpr int N = S::nv;
  // the right-hand-side column's multiply-add rides in the pivot header's assembly block (see there)
  constexpr bool rhs_in_block = AMX_SYM_EXACT && !AMX_PIVOT_CLAMP && !AMX_PIVOT_VCMP && !AMX_TRUE_ROW_ELIM && G != 16 && BCOL && N >= 16 && N < 32;
  const float c = a[J];
  const RowPair p = row_pair<G>(c);
  float piv;
#if !AMX_PIVOT_CLAMP
  // 1 / (lane J of p) in one instruction (v_rcp_f32 takes the DPP modifier itself); the leading s_nop 1 is the 2 wait states
  // after the swap / the last writer.  The pivots of a positive definite matrix need no clamp (the reference's Cholesky has
  // none either); the solution's final division clamps the stored diagonal once per elimination instead of once per pivot.
  float rinv;
#if AMX_PIVOT_RCP_BUILTIN
  // compiler-visible (the DPP combiner does fold the broadcast into v_rcp_f32_dpp): the hazard recogniser pads only the wait states
  // the schedule leaves open between the lane swap and this read - the assembly form carried a fixed s_nop 1 per pivot, 8 cycles
  // on every pivot's dependent chain (biped +1 %, C4 +1.7 %).  The assembly multiply-adds below still depend on it through nf.
  rinv = __builtin_amdgcn_rcpf(bcast_row<J>(p));
#ifdef AMX_EXPERIMENT_PIVOT_REFINE  // accuracy experiment: one Newton step on the pivot's reciprocal (three more vector instructions per pivot)
  { const float pv = bcast_row<J>(p); rinv = rinv * __builtin_fmaf(-pv, rinv, 2.0f); }
#endif
#else
  asm("s_nop 1\n\tv_rcp_f32_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(rinv) : "v"(J < 16 ? p.r0 : p.r1), "n"(J & 15));
#endif
  piv = c;  // on the pivot lane, the only one whose `diag` takes it
#elif AMX_PIVOT_ASM_MAX
  // max(lane J of p, MINVAL) in one instruction; the leading s_nop 1 is the 2 wait states after the swap / the last writer
  asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(piv) : "v"(J < 16 ? p.r0 : p.r1), "v"(MINVAL), "n"(J & 15));
#else
  // max(lane J of p, MINVAL): the compiler folds the DPP move into v_max_f32_dpp and pads only the wait states the distance to
  // the swap leaves open (the assembly form carried a fixed s_nop 1).  Every assembly multiply-add below depends on it through
  // nf and reads a register written by the same swap, so their two wait states hold by construction as before.
  piv = fmaxf(bcast_row<J>(p), MINVAL);
#endif
#if !AMX_PIVOT_CLAMP
  float nf = -c * rinv;
#if AMX_SYM_EXACT
  // Bitwise-symmetric updates (r05).  The column is scaled by 1 / sqrt(pivot) once (both row copies) and an update is the product of two
  // scaled entries: s_i s_K on lane i, s_K s_i on lane K - the same bits, so the reduced matrix stays exactly as symmetric as it started
  // (M is stored symmetric, J'DJ is formed from sqrt(D) J on both operands: AMX_SQRT_D) and the column's lanes hold the pivot ROW bit for
  // bit: the elimination IS the true-row elimination (profiles/r04_standing_impact.txt: 5.7e-5 -> 5e-6 on the hard impact) without a
  // crossbar operation.  Two multiplies more per pivot than (c_i / pivot) c_K.
#ifdef AMX_EXPERIMENT_RSQ_REFINE  // accuracy experiment: one Newton step on 1 / sqrt(pivot)
  float rsq = __builtin_amdgcn_rsqf(bcast_row<J>(p));
  { const float pv = bcast_row<J>(p); rsq = rsq * __builtin_fmaf(-0.5f * pv * rsq, rsq, 1.5f); }
#else
  const float rsq = __builtin_amdgcn_rsqf(bcast_row<J>(p));
#endif
  if constexpr (!rhs_in_block) nf = -c * rsq;  // compiler-visible: it pads the wait state a transcendental's result needs before its first use (the
                                               // block of 32- / 64-lane groups opens with a scalar instruction instead and never reads c)
  RowPair ps;
#define AMX_SYM_P ps
#define AMX_SYM_HEADER_DONE 1
  // One assembly block: the scaled row copies, the two pivot-lane selects and - 32- / 64-lane groups - the right-hand-side column's
  // multiply-add, in this order.  Readers of the outputs need two wait states after these writes (the DPP reads of ps.r0 / ps.r1 by
  // the multiply-adds, the matrix instruction's reads of ps.r0 and nf): ps.r1 has four instructions behind it inside the block, ps.r0
  // three, nf two - no s_nop.  The block reads rsq at least one instruction after the v_rsq_f32 (it consumes nf).  volatile: the
  // headers keep their order.
  {
    float nfo, dgo;
    if constexpr (G == 16) {
      asm volatile("s_lshl_b64 %3, %5, %6\n\tv_mul_f32 %0, %7, %8\n\tv_cndmask_b32_e64 %1, %9, 0, %3\n\tv_cndmask_b32_e64 %2, %10, %11, %3"
                   : "=&v"(ps.r0), "=&v"(nfo), "=&v"(dgo), "=s"(pm) : "3"(pm), "s"(pm0), "n"(J), "v"(c), "v"(rsq), "v"(nf), "v"(diag), "v"(piv) : "scc");
      ps.r1 = ps.r0;
    } else if constexpr (rhs_in_block) {
      // (r05b) the lane's own multiplier is minus its own scaled entry - ps.r0 on the lanes of a group's first 16-lane row, ps.r1 on the
      // second's (rowm: a select with negated sources) - instead of a product with the unswapped column: nothing reads a[J] after the lane
      // swap, so the swap works in place on it and one of its two copies goes (one vector instruction less per pivot)
      float nfr;
      asm volatile("s_lshl_b64 %4, %8, %9\n\tv_mul_f32 %1, %11, %12\n\tv_mul_f32 %0, %10, %12\n\tv_cndmask_b32_e64 %6, -%0, -%1, %15\n\t"
                   "v_cndmask_b32_e64 %2, %6, 0, %4\n\tv_cndmask_b32_e64 %3, %13, %14, %4\n\t"
                   "v_fmac_f32_dpp %5, %1, %2 row_newbcast:%16 row_mask:0xf bank_mask:0xf bound_ctrl:1"
                   : "=&v"(ps.r0), "=&v"(ps.r1), "=&v"(nfo), "=&v"(dgo), "=s"(pm), "+v"(a[N]), "=&v"(nfr)
                   : "4"(pm), "s"(pm0), "n"(J), "v"(p.r0), "v"(p.r1), "v"(rsq), "v"(diag), "v"(J < 16 ? p.r0 : p.r1), "s"(rowm), "n"(N & 15) : "scc");
    } else {
      asm volatile("s_lshl_b64 %4, %6, %7\n\tv_mul_f32 %0, %8, %10\n\tv_mul_f32 %1, %9, %10\n\tv_cndmask_b32_e64 %2, %11, 0, %4\n\tv_cndmask_b32_e64 %3, %12, %13, %4"
                   : "=&v"(ps.r0), "=&v"(ps.r1), "=&v"(nfo), "=&v"(dgo), "=s"(pm)
                   : "4"(pm), "s"(pm0), "n"(J), "v"(p.r0), "v"(p.r1), "v"(rsq), "v"(nf), "v"(diag), "v"(piv) : "scc");
    }
    nf = nfo; diag = dgo;
  }
  if constexpr (NDEF > 0) {
    static_assert(G >= 32 && TREE && BCOL && J >= NDEF, "deferred columns: tree eliminations of 32- / 64-lane groups");
    if constexpr (FIRST) { const v16acc Z = {}; *cdef = __builtin_amdgcn_mfma_f32_16x16x1f32(ps.r0, nf, Z, 0, 0, 0); }
    else *cdef = __builtin_amdgcn_mfma_f32_16x16x1f32(ps.r0, nf, *cdef, 0, 0, 0);
  }
#else
#define AMX_SYM_P p
#endif
#else
  float nf = -c * __builtin_amdgcn_rcpf(piv);
#endif
#if AMX_PIVOT_VCMP && !defined(AMX_SYM_HEADER_DONE)
  nf = gl == J ? 0.0f : nf;  // the pivot row itself is left alone
  diag = gl == J ? piv : diag;  // ... and its diagonal is final: kept for the last division (same test, no pass over the row at the end)
#else
  // the lanes that are lane J of their group: pm0 (lane 0 of every group, opaque to the optimiser) shifted by J on the scalar
  // unit - a pure function of pm0, so the pivots of different limbs still interleave freely
  // volatile: the pivot headers keep their program order.  Left free, the scheduler interleaves every limb's chain for
  // instruction-level parallelism and pays for it in registers: the biped kernel spilled 22 vector and 75 scalar registers
  // and ran 3 % slower than with the pivots in the order written (r03 A/B: profiles/r03_ab_pivot_order.txt)
  // The mask lives in ONE scalar register pair for the whole elimination (the output is tied to the previous pivot's mask): the
  // selects of pivot k must read it before pivot k+1 overwrites it, which pins every pivot's header behind its predecessor's.
#ifndef AMX_SYM_HEADER_DONE
  asm volatile("s_lshl_b64 %0, %2, %3" : "=s"(pm) : "0"(pm), "s"(pm0), "n"(J) : "scc");
  nf = sel_zero(pm, nf);           // the pivot row itself is left alone
  diag = sel_mask(pm, piv, diag);  // ... and its diagonal is final: kept for the last division
#endif
#endif
#if AMX_TRUE_ROW_ELIM
  if constexpr (TREE) {
    a[N] = __builtin_fmaf(nf, gbcast_s<G, J>(a[N]), a[N]);
    SymAncT<S, G, S::dof_parent[J], J>::run(a, nf);
    return;
  }
#endif
  if constexpr (BCOL) {
    if constexpr (!rhs_in_block) fmac_bcast<N>(a[N], AMX_SYM_P, nf);
  } else {
    const RowPair pb = row_pair<G>(a[N]);
#if AMX_SYM_EXACT && !AMX_PIVOT_CLAMP
    a[N] = __builtin_fmaf(nf * rsq, bcast_row<J>(pb), a[N]);  // (nf carries one factor 1 / sqrt(pivot); the right-hand side needs no symmetry)
#else
    a[N] = __builtin_fmaf(nf, bcast_row<J>(pb), a[N]);
#endif
  }
  if constexpr (TREE) SymAnc<S, G, S::dof_parent[J], NDEF>::run(a, AMX_SYM_P, nf);
  else SymRest<S, G, J + 1>::run(a, AMX_SYM_P, nf);
#undef AMX_SYM_P
#undef AMX_SYM_HEADER_DONE
}
#ifndef AMX_ELIM_ORDER
#define AMX_ELIM_ORDER 1  // 0: pivots from the last dof to the first; 1: by rounds of leaves (A/B switch)
#endif
// Order of the tree elimination: any order that pivots a dof after all its descendants is fill-free.  Round by round, the
// current leaves of the tree (highest index first): consecutive pivots then belong to different limbs and their short
// dependent headers (swap, broadcast, reciprocal) overlap in program order, without asking the scheduler to reorder.
template <class S> struct ElimOrder {
  int seq[32];
  constexpr ElimOrder() : seq{} {
    constexpr int N = S::nv;
    bool done[32] = {};
    int n = 0;
    if (AMX_ELIM_ORDER == 0) { for (int j = N - 1; j >= 0; --j) seq[n++] = j; }
    while (n < N) {
      bool leaf[32] = {};
      for (int j = 0; j < N; ++j) leaf[j] = !done[j];
      for (int j = 0; j < N; ++j) if (!done[j] && S::dof_parent[j] >= 0) leaf[S::dof_parent[j]] = false;
      for (int j = N - 1; j >= 0; --j) if (leaf[j]) { seq[n++] = j; done[j] = true; }
    }
  }
};
#ifndef AMX_ELIM_DEFER
#define AMX_ELIM_DEFER 0  // 0: every pivot updates all of its ancestors' columns on the vector unit (r04), A/B switch
#endif
// Which columns a tree elimination defers to the matrix cores (sym_pivot, NDEF) and for how many leading pivots.  Columns 0 .. ndef - 1 are
// the chain at the root of the dof tree (dof k's parent is k - 1, at most 8: a floating base has 6); the pivots that defer are the leading
// ones of the order whose index is >= ndef - a dof's ancestors have lower indices, so none of them reads a deferred column.  Worth it when
// the multiply-adds saved outnumber the collection's swaps and adds several times.
template <class S, int G, bool BCOL> struct ElimDefer {
  int ndef, npre;
  constexpr ElimDefer() : ndef(0), npre(0) {
    constexpr ElimOrder<S> ord{};
    constexpr int N = S::nv;
    if (AMX_ELIM_DEFER && AMX_SYM_EXACT && !AMX_PIVOT_CLAMP && !AMX_PIVOT_VCMP && !AMX_TRUE_ROW_ELIM && G >= 32 && BCOL && N >= 16 && N < 32) {
      int c = 1;
      while (c < 8 && c < N && S::dof_parent[c] == c - 1) ++c;
      int pre = 0, saved = 0;
      while (pre < N && ord.seq[pre] >= c) {
        for (int k = S::dof_parent[ord.seq[pre]]; k >= 0; k = S::dof_parent[k]) saved += k < c ? 1 : 0;
        ++pre;
      }
      if (saved >= 4 * (2 * c + 4)) { ndef = c; npre = pre; }
    }
  }
};
// adds the deferred columns' sums to a[0 .. NDEF - 1].  Register 4 b + r of lane (g, j) of the accumulator holds entry 4 g + r of column j of
// block b = the sum for dof 4 g + r (the A operand's index) of the row lane j of block b holds (the B operand's): the same 4 x 4 transpose of
// 16-lane row pieces as newton_solve_pair2's, but only the first NDEF entries are wanted.
template <int NDEF, int NV> DEV void sym_defer_collect(float (&a)[NV + 1], const v16acc &C) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (r >= NDEF) break;
    auto s01 = __builtin_amdgcn_permlane16_swap(__float_as_uint(C[r]), __float_as_uint(C[4 + r]), false, false);
    auto s23 = __builtin_amdgcn_permlane16_swap(__float_as_uint(C[8 + r]), __float_as_uint(C[12 + r]), false, false);
    auto t02 = __builtin_amdgcn_permlane32_swap(s01[0], s23[0], false, false);
    a[r] += __uint_as_float(t02[0]);
    if (4 + r < NDEF) {
      auto t13 = __builtin_amdgcn_permlane32_swap(s01[1], s23[1], false, false);
      a[4 + r] += __uint_as_float(t13[0]);
    }
  }
}
template <class S, int G, int I, bool BCOL> struct SymTree {  // leaves before their ancestors
  static DEV void run(float (&a)[S::nv + 1], int gl, float &diag, unsigned long long pm0, unsigned long long &pm, v16acc &cdef, unsigned long long rowm) {
    if constexpr (I < S::nv) {
      constexpr ElimOrder<S> ord{};
      constexpr ElimDefer<S, G, BCOL> df{};
      if constexpr (I < df.npre) sym_pivot<S, G, ord.seq[I], true, BCOL, df.ndef, I == 0>(a, gl, diag, pm0, pm, &cdef, rowm);
      else {
        if constexpr (df.npre > 0 && I == df.npre) sym_defer_collect<df.ndef, S::nv>(a, cdef);
        sym_pivot<S, G, ord.seq[I], true, BCOL>(a, gl, diag, pm0, pm, nullptr, rowm);
      }
      SymTree<S, G, I + 1, BCOL>::run(a, gl, diag, pm0, pm, cdef, rowm);
    }
  }
};
template <class S, int G, int J, bool BCOL> struct SymDense {
  static DEV void run(float (&a)[S::nv + 1], int gl, float &diag, unsigned long long pm0, unsigned long long &pm, unsigned long long rowm) {
    if constexpr (J < S::nv) {
      sym_pivot<S, G, J, false, BCOL>(a, gl, diag, pm0, pm, nullptr, rowm);
      SymDense<S, G, J + 1, BCOL>::run(a, gl, diag, pm0, pm, rowm);
    }
  }
};
// entry point
template <class S, int G, bool TREE, bool BCOL> DEV void sym_eliminate(float (&a)[S::nv + 1], int gl, float &diag) {
  unsigned long long pm0 = lane_mask<G, 0>(), pm = 0;
  asm volatile("" : "+s"(pm0), "+s"(pm));  // opaque: thirty constant masks would be hoisted out of the step loop and spilled
  unsigned long long rowm = 0xFFFF0000FFFF0000ull;  // the lanes of every second 16-lane row
  asm volatile("" : "+s"(rowm));
  if constexpr (TREE) { v16acc cdef; SymTree<S, G, 0, BCOL>::run(a, gl, diag, pm0, pm, cdef, rowm); }
  else SymDense<S, G, 0, BCOL>::run(a, gl, diag, pm0, pm, rowm);
}
// (64-lane groups, r04: one environment per wave with its matrix rows in lanes 0..nv-1 < 32 - the same two DPP rows as a 32-lane
// group; lanes 32..63 hold all-zero rows and compute on them, their results are never stored)
#ifndef AMX_SYM_ELIM_64
#define AMX_SYM_ELIM_64 1  // 0: static 64-lane kernels keep the v_readlane eliminations (A/B switch)
#endif
template <class S, int G> constexpr bool sym_elim_ok() { return S::is_static && (G <= 32 ? S::nv < G : (AMX_SYM_ELIM_64 && S::nv < 32)) && !AMX_NO_SYM_ELIM; }

// ---- the generic kernels' eliminations without the crossbar (r05).  The run-time descriptor fixes neither the size nor the dof tree, so the
// generic solves dispatch on the padded size N (spd_solve, newton_fused_dyn) and broadcast every pivot-row entry through the LDS crossbar
// (gj_pivot: one ds_swizzle per updated column, ~450 cycles per pivot).  The matrices they are handed are bitwise symmetric like the static
// kernels' (M is stored symmetric, J'DJ comes from sqrt(D) J on both operands, the scattered rows add the same value to both mirror entries in
// the same order), so the symmetric DPP elimination applies as it stands: DENSE order (every column after the pivot: no tree at compile time),
// right-hand side broadcast on its own (no mirror row).  Padding rows are identity rows, lanes past N hold zero rows.
// MEASURED, OFF (r05): headline model on the generic kernels 54.3 -> 59.9 M env-steps/s (the eliminations are not what holds the generic kernels
// back most: run-time loop bounds, 421 scalar spills, no fused Newton forms), and the 1,050-model sweep moved by noise - but one convex draw of the
// test suite (seed 9) goes 8e-6 -> 1.6e-5, over its bar.  The generic kernels keep their true-row crossbar eliminations.
#ifndef AMX_GENERIC_SYM
#define AMX_GENERIC_SYM 0  // 1: dense symmetric DPP eliminations in the generic kernels (A/B switch)
#endif
template <int NN> struct DenseN {
  static constexpr int nv = NN;
  static constexpr int dof_parent[32] = {};
};
template <int N, int G> constexpr bool dense_sym_ok() {
  return AMX_GENERIC_SYM && AMX_SYM_EXACT && !AMX_PIVOT_CLAMP && !AMX_PIVOT_VCMP && !AMX_TRUE_ROW_ELIM && !AMX_NO_SYM_ELIM && (G <= 32 ? N < G : N < 32);
}
// solves in place: on return b is the lane's solution component (lanes without a row: garbage, never stored)
template <int N, int G> DEV float dense_sym_solve(float (&a)[N], float b, int gl) {
  float aa[N + 1];
#pragma unroll
  for (int k = 0; k < N; ++k) aa[k] = a[k];
  aa[N] = b;
  float d = 1.0f;
  int gle = gl;
  asm volatile("" : "+v"(gle));
  sym_eliminate<DenseN<N>, G, false, false>(aa, gle, d);
  return fdiv(aa[N], fmaxf(d, MINVAL));
}

// Row `gl` of M from the compact storage (planner: mcomp; r05) into a[0 .. N-1] - the same numbers a whole row would have given.
//   * entries on the leading chain, j < mnb0: M[j][gl] down the whole rows' column (one four-byte load each, one address for all of them);
//   * the others as 16-byte chunks from the row's own storage, whose pseudo-base `mdiag - gl` makes chunk c sit at + 4 c; a chunk
//     outside the row's stored range (bit c of mwin clear) comes from the constants' zero row.
// brow: the lane that mirrors the right-hand side as a row takes its chunks from the staged vector at bt_off and its leading entries from
// column N of the whole rows, where the caller has put them; a lane without a row reads the zero row and the always-zero column ldm - 1.
template <class S, int N>
DEV void m_row_compact(float (&a)[N + 1], int M_off, int gl, int mdiag, unsigned mwin, bool owner, bool brow, int bt_off) {
  constexpr int ld = S::ldm, nb0 = S::mnb0, bs = S::mbs;
  static_assert(!S::mcomp || ld >= N + 2, "compact M: two spare columns");
  const int colb = M_off + (gl <= N ? gl : ld - 1);
#pragma unroll
  for (int j = 0; j < nb0; ++j) a[j] = lds[colb + j * bs];
  // (zeros: every lane its own sixteen bytes of the constants' zero block - forty lanes reading ONE zero chunk were served one after the other:
  //  LDS bank-conflict cycles 45 % of the LDS pipe's time on the first build of this form)
  const int zb = S::c_zero + 4 * (gl & 31);
  // (mdiag - gl is a multiple of four floats - the planner's row offsets are; said so, or the chunk loads are split into four-byte halves)
  const int rb = brow ? bt_off : (owner ? M_off + 4 * ((mdiag - gl) >> 2) : zb);
  const unsigned win = brow ? ~0u : mwin;
#pragma unroll
  for (int c = nb0 / 4; c < (N + 3) / 4; ++c) {
    const int ad = ((win >> c) & 1u) ? rb : zb;
    const float4 v = *reinterpret_cast<const float4 *>(__builtin_assume_aligned(lds + ad + 4 * c, 16));
    const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k = 4 * c + q;
      if (k >= nb0 && k < N) a[k] = e[q];
    }
  }
}

template <class S, int G, bool TREE>
AMX_STATIC_SOLVE_ATTR void spd_solve_static(int A_off, int rhs_off, int x_off, int gl, int rot, int bt_off, int mdiag = 0, unsigned mwin = 0u) {
  constexpr int N = S::nv, ld = S::ldm, nch = ld / 4;
  if constexpr (sym_elim_ok<S, G>()) {
    // bt_off: an idle vector of ld floats (the M*v staging vector) through which b reaches lane N as a row; < 0: none, b is
    // broadcast on its own (always so for 16-lane groups, where that costs nothing)
    constexpr bool BCOL = G >= 32;
    // the pivot-lane tests (gl == J) of one elimination are recomputed here (one v_cmp each): left loop-invariant, the thirty
    // masks were hoisted out of the step loop, held in scalar registers across it and spilled (two v_readlane per use)
    int gle = gl;
    asm volatile("" : "+v"(gle));
    float a[N + 1];
    const int li = gl < N ? gl : 0;
    float b = lds[rhs_off + li * DOF_E];
    b = gl < N ? b : 0.0f;
    const bool brow = BCOL && gl == N && bt_off >= 0;
    if (BCOL && bt_off >= 0) {
      if (gl < ld) lds[bt_off + gl] = b;
      if (S::mcomp && gl < S::mnb0) lds[A_off + gl * S::mbs + N] = b;  // the mirrored row's leading entries: column N of the whole rows
      WSYNC();
    }
    if constexpr (S::mcomp) {
      static_assert(BCOL, "compact M: 32- and 64-lane groups");
      m_row_compact<S, N>(a, A_off, gl, mdiag, mwin, gl < N, brow, bt_off);
    } else {
    // lanes without a matrix row load the constants' all-zero row: no select per entry
    const float4 *rowp = reinterpret_cast<const float4 *>(lds + (brow ? bt_off : (gl < N ? A_off + li * ld : S::c_zero)));
    const int rot_l = gl < N ? rot : 0;
#pragma unroll
    for (int c = 0; c < (N + 3) / 4; ++c) {
      int pc = c + (S::mrot ? rot_l : 0);
      pc -= pc >= nch ? nch : 0;
      const float4 v = rowp[pc];
      const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int k = 4 * c + q;
        if (k < N) a[k] = e[q];
      }
    }
    }
    a[N] = b;
    float d = 1.0f;
    if (BCOL && bt_off >= 0) sym_eliminate<S, G, TREE, true>(a, gle, d);
    else sym_eliminate<S, G, TREE, false>(a, gle, d);
    if (gl < N) lds[x_off + gl * DOF_E] = fdiv(a[N], AMX_PIVOT_CLAMP ? d : fmaxf(d, MINVAL));
    WSYNC();
    return;
  }
  float a[N];
  const int li = gl < N ? gl : 0;
  const float4 *rowp = reinterpret_cast<const float4 *>(lds + A_off + li * ld);
#pragma unroll
  for (int c = 0; c < (N + 3) / 4; ++c) {
    int pc = c + (S::mrot ? rot : 0);  // chunk c of the row sits at position (c + rot) mod nch
    pc -= pc >= nch ? nch : 0;
    const float4 v = rowp[pc];
    const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k = 4 * c + q;
      if (k < N) a[k] = gl < N ? e[q] : (k == gl ? 1.0f : 0.0f);
    }
  }
  float b = lds[rhs_off + li * DOF_E];
  b = gl < N ? b : 0.0f;
  if (TREE) GJTree<S, G, N - 1>::run(a, b, gl);
  else GJ<N, G, 0>::run(a, b, gl, 0);
  float d = 1.0f;
#pragma unroll
  for (int k = 0; k < N; ++k) d = (k == gl) ? a[k] : d;
  if (gl < N) lds[x_off + gl * DOF_E] = fdiv(b, d);
  WSYNC();
}

// Newton direction without an H buffer (planner: fusedH - 32- and 64-lane groups, no equality rows): the matrix cores' J'DJ rows
// arrive in registers (jtdj_rows_mfma), the lane adds its row of M (stored chunk-rotated when MV::mrot) and the diagonal
// term of the limit row on its dof (dlim, fetched by the caller through the crossbar), and the elimination starts from
// those registers: H never touches LDS.  S static: N = nv, tree-sparse elimination when TREE; S = DynSpec: N is the padded size.
// GRADCOL (static, 32-lane groups, nv <= 31; see jtdj_rows_mfma COL31): the caller has not formed qfrc_constraint and the
// gradient; they come out of the matrix cores' column 31 here - fcon = column + flim (the lane's limit-row force),
// grad = Ma - qfrc_smooth - fcon - and are stored to fcon_off / rhs_off before the elimination uses grad as its right-hand side.
// With GRADCOL the stopping test of the NEXT solver iteration can run here, before the elimination: imp_done is the group's
// improvement test (already known to the caller), the gradient is in the registers.  When no group of the wave would iterate
// again the elimination (two thirds of this routine) is skipped and the routine returns true; the caller stops iterating.
template <class S, int N, int G, bool TREE, bool GRADCOL = false>
AMX_STATIC_SOLVE_ATTR bool newton_solve_fused(int M_off, int ld_, int n_, unsigned long long crows_, int row_base, int J_base, int ldc_, int slot_base_,
                                                int rhs_off, int x_off, float dlim, int gl, int gb4, int rot, int bt_off, float flim = 0.0f, float ma_v = 0.0f,
                                                float fs_v = 0.0f, int fcon_off = 0, bool store_ok = true, bool early_out = false, bool imp_done = false,
                                                float scale = 1.0f, float tol = 0.0f, ProfCtx *pc = nullptr, bool *group_done = nullptr, unsigned long long crows2_ = 0ull,
                                                int mdiag = 0, unsigned mwin = 0u) {
  if constexpr (sym_elim_ok<S, G>() && G >= 32) {
    // crossbar-free elimination (see sym_pivot): b reaches lane N as a row through the staging vector at bt_off
    constexpr int NS = S::nv, lds_ = S::ldm, nchs = lds_ / 4;
    int gle = gl;  // see spd_solve_static
    asm volatile("" : "+v"(gle));
    const unsigned long long crows_u = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(crows_ >> 32)) << 32) |
                                       (unsigned)__builtin_amdgcn_readfirstlane((int)crows_);
    const unsigned long long crows2_u = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(crows2_ >> 32)) << 32) |
                                        (unsigned)__builtin_amdgcn_readfirstlane((int)crows2_);
    unsigned lo[4][4], hi[4][4];
    const bool owner = gl < NS, brow = gl == NS;
    const int li = owner ? gl : 0;
    // the limit row's diagonal term goes into the stored diagonal of M for the duration of the row loads (one LDS
    // read-modify-write, hidden behind the matrix-core loop) instead of a compare + select + add per register of the row
    int dpc = (gl >> 2) + (S::mrot ? rot : 0);
    dpc -= dpc >= nchs ? nchs : 0;
    const int dg = S::mcomp ? M_off + mdiag : M_off + li * lds_ + 4 * dpc + (gl & 3);
    float d0 = 0.0f;
    if (owner) { d0 = lds[dg]; lds[dg] = d0 + dlim; }
    float b;
    if constexpr (GRADCOL && G == 32 && AMX_MFMA_AUG && (S::ldm > S::nv) && (S::nv >= AMX_MFMA_AUG_MIN_NV) && !S::mcomp) {
      // r03: the rows of M, the gradient and the mirrored right-hand-side row all come out of the matrix cores' accumulator.
      //  * The accumulator starts from M instead of zero.  In the 2-block instruction's layout lane l of the wave holds, for block
      //    0 (the wave's first environment) and block 1 (its second), rows 8q + 4 (l / 32) + r of column l % 32; after the
      //    half-wave swaps each lane keeps its own environment's whole column = its row of the symmetric matrix.  So every lane
      //    loads the chunks 2q + (l / 32) of row l % 32 of BOTH environments' M (four 16-byte loads from its own slab, four from
      //    its wave mate's) into the accumulator's registers: the same eight loads as before, no additions afterwards.
      //  * Column NS of the rows of M (padding: ldm > nv) is set to ma - fs - flim of the row's dof, and row NS is that vector as
      //    a row (staged through bt_off): the symmetric (NS+1) x (NS+1) matrix the elimination works on, before the contacts.
      //  * Every contact row's Jacobian gets one more entry at index NS: its residual Jaref.  Row / column NS of J'DJ is then
      //    sum_r D_r J_r[k] Jaref_r = -(J'f)[k]: entry NS of a dof's row becomes the gradient ma - fs - flim - J'f, which is the
      //    right-hand side, and lane NS receives the same values as a row - no staging after the matrix cores, no column 31.
      const int hw = (gb4 >> 7) & 1;  // wave half = which environment of the wave is mine
      const float pre = owner ? ma_v - fs_v - flim : 0.0f;
      if (gl < lds_) lds[bt_off + gl] = pre;
      int apc = (NS >> 2) + (S::mrot ? rot : 0);
      apc -= apc >= nchs ? nchs : 0;
      if (owner) lds[M_off + li * lds_ + 4 * apc + (NS & 3)] = pre;  // (padding column: nothing to restore)
      const int dA = hw ? -S::env_floats : 0, dB = hw ? 0 : S::env_floats;  // from my slab to the first / second environment's
      const int src = brow ? bt_off : (owner ? M_off + li * lds_ : S::c_zero), dsl = (brow | owner) ? 1 : 0;
      const float4 *rowA = reinterpret_cast<const float4 *>(lds + src + dsl * dA), *rowB = reinterpret_cast<const float4 *>(lds + src + dsl * dB);
      const int rot_c = owner ? rot : 0;
      v32f C0;
      PROF_CTX(pc, 24);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = 2 * q + hw;  // chunk of the row this lane's accumulator registers 4q .. 4q+3 (16 + 4q ..) hold
        int pcc = c + (S::mrot ? rot_c : 0);
        pcc -= pcc >= nchs ? nchs : 0;
        const bool in = c < nchs;  // a tile column past the row's end
        const float4 vA = rowA[in ? pcc : 0], vB = rowB[in ? pcc : 0];
        C0[4 * q + 0] = in ? vA.x : 0.0f; C0[4 * q + 1] = in ? vA.y : 0.0f; C0[4 * q + 2] = in ? vA.z : 0.0f; C0[4 * q + 3] = in ? vA.w : 0.0f;
        C0[16 + 4 * q + 0] = in ? vB.x : 0.0f; C0[16 + 4 * q + 1] = in ? vB.y : 0.0f; C0[16 + 4 * q + 2] = in ? vB.z : 0.0f; C0[16 + 4 * q + 3] = in ? vB.w : 0.0f;
      }
      jtdj_rows_mfma<G, false, NS, S::jfr != 0>(crows_u, row_base, J_base, S::ldc, S::c_slot, gl, lo, hi, &C0);
      PROF_CTX(pc, 25);
      if (owner) lds[dg] = d0;  // M's own diagonal back (the loads above are done: one wave's LDS operations execute in order)
      float a[NS + 1];
#pragma unroll
      for (int c = 0; c < (NS + 4) / 4; ++c) {
        const unsigned *srcr = (c & 1) ? hi[c >> 1] : lo[c >> 1];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int k = 4 * c + q;
          if (k <= NS) a[k] = __uint_as_float(srcr[q]);
        }
      }
      b = owner ? a[NS] : 0.0f;
      const float fcon = ma_v - fs_v - b;  // = flim + J'f of the contact rows
      if (owner && store_ok) { lds[fcon_off + gl * DOF_E] = fcon; lds[rhs_off + gl * DOF_E] = b; }
      if (early_out) {
        const float gn = gsum<G>(b * b);
        const float ts_ = scale * tol;
        const bool done = imp_done | (gn < ts_ * ts_);
        if (group_done) *group_done = done;
        if (!wave_any(store_ok & !done)) { WSYNC(); return true; }
      }
      a[NS] = b;
      float d = 1.0f;
      PROF_CTX(pc, 26);
      sym_eliminate<S, G, TREE, true>(a, gle, d);
      PROF_CTX(pc, 27);
      if (owner) lds[x_off + gl * DOF_E] = fdiv(a[NS], AMX_PIVOT_CLAMP ? d : fmaxf(d, MINVAL));
      WSYNC();
      return false;
    } else
    if constexpr (GRADCOL) {
      static_assert(NS <= 31, "column 31 must be spare");
      const float ma = ma_v, fs = fs_v;  // M*qacc and qfrc_smooth of the lane's dof, from the caller's registers
      PROF_CTX(pc, 24);
      jtdj_rows_mfma<G, true, -1, S::jfr != 0>(crows_u, row_base, J_base, S::ldc, S::c_slot, gl, lo, hi, nullptr, crows2_u);
      PROF_CTX(pc, 25);
      const float fcon = __uint_as_float(hi[3][3]) + flim;
      b = ma - fs - fcon;
      if (owner && store_ok) { lds[fcon_off + gl * DOF_E] = fcon; lds[rhs_off + gl * DOF_E] = b; }
      if (early_out) {
        const float gn = gsum<G>(owner ? b * b : 0.0f);
        const float ts_ = scale * tol;  // |grad| / scale < tol, squared: no square root, no division
        const bool done = imp_done | (gn < ts_ * ts_);  // `|`, `&`: no exec-mask branches (see linesearch)
        if (group_done) *group_done = done;  // the caller's next iteration starts with this very test: it need not re-read the gradient
        if (!wave_any(store_ok & !done)) {  // store_ok is the group's `alive`
          if (owner) lds[dg] = d0;
          WSYNC();
          return true;
        }
      }
    } else {
      jtdj_rows_mfma<G, false, -1, S::jfr != 0>(crows_u, row_base, J_base, S::ldc, S::c_slot, gl, lo, hi, nullptr, crows2_u);
      b = lds[rhs_off + li * DOF_E];
    }
    b = owner ? b : 0.0f;
    if (gl < lds_) lds[bt_off + gl] = b;
    if (S::mcomp && gl < S::mnb0) lds[M_off + gl * S::mbs + NS] = b;  // the mirrored row's leading entries (m_row_compact)
    WSYNC();
    // lanes without a matrix row load the constants' all-zero row, and their J'DJ registers are zero (no slot): the row is a
    // plain sum.  The diagonal term of the lane's limit row went into M's diagonal in LDS before the loads (restored below).
    float a[NS + 1];
    if constexpr (S::mcomp) {
      m_row_compact<S, NS>(a, M_off, gl, mdiag, mwin, owner, brow, bt_off);
#pragma unroll
      for (int k = 0; k < NS; ++k) {
        const unsigned *src = ((k >> 2) & 1) ? hi[k >> 3] : lo[k >> 3];
        a[k] += __uint_as_float(src[k & 3]);
      }
    } else {
    const float4 *rowp = reinterpret_cast<const float4 *>(lds + (brow ? bt_off : (owner ? M_off + li * lds_ : S::c_zero)));
    const int rot_l = owner ? rot : 0;
#pragma unroll
    for (int c = 0; c < (NS + 3) / 4; ++c) {
      int pc = c + (S::mrot ? rot_l : 0);
      pc -= pc >= nchs ? nchs : 0;
      const float4 v = rowp[pc];
      const unsigned *src = (c & 1) ? hi[c >> 1] : lo[c >> 1];
      const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int k = 4 * c + q;
        if (k < NS) a[k] = e[q] + __uint_as_float(src[q]);
      }
    }
    }
    if (owner) lds[dg] = d0;  // M's own diagonal back (after the loads above: one wave's LDS operations execute in order)
    a[NS] = b;
    float d = 1.0f;
    PROF_CTX(pc, 26);
    sym_eliminate<S, G, TREE, true>(a, gle, d);
    PROF_CTX(pc, 27);
    if (owner) lds[x_off + gl * DOF_E] = fdiv(a[NS], AMX_PIVOT_CLAMP ? d : fmaxf(d, MINVAL));
    WSYNC();
    return false;
  }
  const int ld = S::is_static ? S::ldm : __builtin_amdgcn_readfirstlane(ld_), n = S::is_static ? S::nv : __builtin_amdgcn_readfirstlane(n_);
  const int ldc = S::is_static ? S::ldc : __builtin_amdgcn_readfirstlane(ldc_), slot_base = S::is_static ? S::c_slot : __builtin_amdgcn_readfirstlane(slot_base_);
  const unsigned long long crows = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(crows_ >> 32)) << 32) |
                                   (unsigned)__builtin_amdgcn_readfirstlane((int)crows_);
  unsigned lo[4][4], hi[4][4];
  jtdj_rows_mfma<G>(crows, row_base, J_base, ldc, slot_base, gl, lo, hi);
  const bool owner = gl < n;  // row gl of the lane's environment (64-lane groups: lanes 32..63 own padding rows)
  const int li = owner ? gl : 0, nch = ld >> 2;
  const float4 *rowp = reinterpret_cast<const float4 *>(lds + M_off + li * ld);
  float a[N];
#pragma unroll
  for (int c = 0; c < (N + 3) / 4; ++c) {
    float4 v = {0.0f, 0.0f, 0.0f, 0.0f};
    int pc = c + rot;
    pc -= pc >= nch ? nch : 0;
    if (4 * c < ld) v = rowp[4 * c < ld ? pc : 0];
    const unsigned *src = (c & 1) ? hi[c >> 1] : lo[c >> 1];
    const float e[4] = {v.x + __uint_as_float(src[0]), v.y + __uint_as_float(src[1]), v.z + __uint_as_float(src[2]), v.w + __uint_as_float(src[3])};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k = 4 * c + q;
      if (k < N) a[k] = (owner && k < n) ? e[q] + (k == gl ? dlim : 0.0f) : (k == gl ? 1.0f : 0.0f);
    }
  }
  float b = lds[rhs_off + li * DOF_E];
  b = owner ? b : 0.0f;
  if constexpr (!S::is_static && dense_sym_ok<N, G>()) {
    const float x = dense_sym_solve<N, G>(a, b, gl);
    if (owner) lds[x_off + gl * DOF_E] = x;
    WSYNC();
    return false;
  }
  if constexpr (S::is_static && TREE) GJTree<S, G, N - 1>::run(a, b, gl);
  else GJ<N, G, 0>::run(a, b, gl, gb4);
  float d = 1.0f;
#pragma unroll
  for (int k = 0; k < N; ++k) d = (k == gl) ? a[k] : d;
  if (owner) lds[x_off + gl * DOF_E] = fdiv(b, d);
  WSYNC();
  return false;
}
// ---- Newton direction of a two-pair model (planner: mfma2 - a biped's two feet; r04).  Same contract as the GRADCOL form of
// newton_solve_fused; what differs is how J'DJ and J'f reach the elimination's registers.
// The 32 x 32 two-block instruction spends 64 cycles of the matrix pipe per pyramid row on a product of which 12 x 12 entries per
// environment are non-zero (a foot's rows touch the base and one leg), and the pipe's time is not hidden: with every contact of a
// standing robot active, issuing each instruction twice cost 10.9 % of the step (profiles/r04_ab_mfma.txt).  Here the product is
// formed chain-compressed: v_mfma_f32_16x16x1_4b_f32 has four 16 x 16 blocks = (environment, pair); lane 16 f + k of an environment
// feeds block (env, f) - as B operand the Jacobian entry of the dof whose ROW it will hold, as A operand the entry of the dof
// register k stands for (amx_plan.h: c_p2, Spec::p2_col; one A lane carries the row's residual, which makes column `nv` the gradient).
// One instruction takes a row of EACH pair in both environments: 16 instructions of 32 cycles per Hessian instead of 32 of 64.
// The accumulator gives lane (g, j) entries 4 g .. 4 g + 3 of column j of every block; four row exchanges per register quadruple
// transpose that into "lane (b, j) holds column j = row j of ITS block", 16 entries.  The rows then go home: the planner has put a
// pair's rows either in their dofs' own lanes (added in place) or in the other 16-lane row at (position + shift) - one
// v_permlane16_swap and a DPP row shift on the add - and every other lane the adds' bank masks enable holds an all-zero row.
// The mirrored right-hand-side row of lane nv is staged through LDS (the 32 x 32 form got it from the product's symmetry).
#ifndef AMX_MFMA_PAIR2
#define AMX_MFMA_PAIR2 1  // 0: two-pair models keep the 32 x 32 two-block product (A/B switch)
#endif
template <int CTRL_SHL, int ROWS, int BANKS> DEV void p2_add(float &acc, float src) {
  // acc += src read through DPP (row_shl:CTRL_SHL, or in place), on the lanes of the 16-lane rows ROWS (mask) and banks BANKS only
  if constexpr (CTRL_SHL == 0)
    asm("v_add_f32_dpp %0, %1, %0 quad_perm:[0,1,2,3] row_mask:%2 bank_mask:%3" : "+v"(acc) : "v"(src), "n"(ROWS), "n"(BANKS));
  else
    asm("v_add_f32_dpp %0, %1, %0 row_shl:%4 row_mask:%2 bank_mask:%3 bound_ctrl:1" : "+v"(acc) : "v"(src), "n"(ROWS), "n"(BANKS), "n"(CTRL_SHL));
}
// Two wait states between the lane swaps that wrote r[0..15] and the DPP reads (inline assembly: the hazard recogniser cannot pad them)
// that follow: every register passes through this statement, so the compiler has to keep all their writers in front of it and all
// their readers behind it.
DEV void p2_fence(float (&r)[16]) {
  asm volatile("s_nop 1" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]), "+v"(r[9]), "+v"(r[10]),
               "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15]));
}
// the same for the first N registers of an eight-entry array (the registers in use: an unused one named in the statement would have to be
// materialised - cleared - just to pass through it)
template <int N> DEV void p2_fence_n(float (&r)[8]) {
  static_assert(N >= 0 && N <= 8, "p2_fence_n");
  if constexpr (N == 1) asm volatile("s_nop 1" : "+v"(r[0]));
  if constexpr (N == 2) asm volatile("s_nop 1" : "+v"(r[0]), "+v"(r[1]));
  if constexpr (N == 3) asm volatile("s_nop 1" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]));
  if constexpr (N == 4) asm volatile("s_nop 1" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]));
  if constexpr (N == 5) asm volatile("s_nop 1" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]));
  if constexpr (N == 6) asm volatile("s_nop 1" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]));
  if constexpr (N == 7) asm volatile("s_nop 1" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]));
  if constexpr (N == 8) asm volatile("s_nop 1" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]));
}
template <class S, bool TREE>
AMX_STATIC_SOLVE_ATTR bool newton_solve_pair2(int M_off, unsigned long long crows_, int row_base, int J_base, int rhs_off, int x_off, float dlim, int gl, int gb4, int rot,
                                                int bt_off, float flim, float ma_v, float fs_v, int fcon_off, bool store_ok, bool early_out, bool imp_done, float scale,
                                                float tol, ProfCtx *pc, bool *group_done) {
  constexpr int G = 32, NS = S::nv, lds_ = S::ldm, nchs = lds_ / 4;
  static_assert(S::mfma2 && NS <= 30 && lds_ >= NS + 1, "two-pair form: planner's mfma2");
  int gle = gl;  // see spd_solve_static
  asm volatile("" : "+v"(gle));
  const unsigned crows_u = (unsigned)__builtin_amdgcn_readfirstlane((int)crows_);  // 32 contact rows: bits 0..15 pair 0, 16..31 pair 1
  const bool owner = gl < NS, brow = gl == NS;
  const int li = owner ? gl : 0;
  // the limit row's diagonal term rides in M's stored diagonal for the duration of the row loads (as in newton_solve_fused)
  int dpc = (gl >> 2) + (S::mrot ? rot : 0);
  dpc -= dpc >= nchs ? nchs : 0;
  const int dg = M_off + li * lds_ + 4 * dpc + (gl & 3);
  float d0 = 0.0f;
  if (owner) { d0 = lds[dg]; lds[dg] = d0 + dlim; }
  PROF_CTX(pc, 24);
  // the lane's row of M (lanes without one: the constants' all-zero row); entry NS starts as ma - fs - flim of the lane's dof
  float a[NS + 1];
  {
    const float4 *rowp = reinterpret_cast<const float4 *>(lds + (owner ? M_off + li * lds_ : S::c_zero));
    const int rot_l = owner ? rot : 0;
#pragma unroll
    for (int c = 0; c < (NS + 3) / 4; ++c) {
      int pcc = c + (S::mrot ? rot_l : 0);
      pcc -= pcc >= nchs ? nchs : 0;
      const float4 v = rowp[pcc];
      const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) if (4 * c + q < NS) a[4 * c + q] = e[q];
    }
  }
  a[NS] = owner ? ma_v - fs_v - flim : 0.0f;
  // ---- the product.  This lane: pair f = bit 4 of the lane, operand slots from the constants
  const int f = (gl >> 4) & 1;
  const int pk = CI(S::c_p2 + gl);
  const int sA = pk & 255, sB = (pk >> 8) & 255;
  const bool hasA = sA < 0xFE, augA = sA == 0xFE, hasB = sB < 0xFE;
  const int JB = J_base + 16 * f * S::ldc + (hasB ? sB : 0), Dr = row_base + 16 * f * ROW_E;
  // the A operand's address: entry sA of the row's Jacobian, or - the lane that carries the residual - field 2 of the row record
  const int JA = augA ? Dr + 2 : J_base + 16 * f * S::ldc + (hasA ? sA : 0), strideA = augA ? ROW_E : S::ldc;
  const unsigned tm = (crows_u | (crows_u >> 16)) & 0xFFFFu;  // row t of either pair active in some environment of the wave
  v16acc C;
  // All sixteen rows, straight-line, in two batches of eight: the operand loads of a batch are in flight together and the next
  // batch's behind this batch's instructions.  (As a loop over the active rows every row was an LDS round trip in front of its
  // 32-cycle instruction: 3.2 k cycles per Hessian for 0.5 k of matrix pipe.)  An inactive row has D = 0 in its record: its
  // instruction adds zeros, so no row is tested - a batch is skipped when none of its rows is active anywhere in the wave.
  struct Ops { float d, ja, jb; };
  auto load8 = [&](int t0, Ops (&o)[8]) {
#pragma unroll
    for (int k = 0; k < 8; ++k) { const int t = t0 + k; o[k] = Ops{lds[Dr + t * ROW_E], lds[JA + t * strideA], lds[JB + t * S::ldc]}; }
  };
  // (first: the batch that starts the accumulator takes the instruction's constant 0 operand for its first row - clearing sixteen registers
  //  with v_mov first cost as much vector time as the batch's own operand multiplies; r05)
  auto issue8 = [&](const Ops (&o)[8], auto first) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float av = o[k].d * ((hasA | augA) ? o[k].ja : 0.0f), bv = AMX_SQRT_D ? o[k].d * (hasB ? o[k].jb : 0.0f) : (hasB ? o[k].jb : 0.0f);
      if (decltype(first)::value && k == 0) { const v16acc Z = {}; C = __builtin_amdgcn_mfma_f32_16x16x1f32(av, bv, Z, 0, 0, 0); }
      else C = __builtin_amdgcn_mfma_f32_16x16x1f32(av, bv, C, 0, 0, 0);
    }
  };
  if (owner) lds[dg] = d0;  // M's own diagonal back (the row loads above are done: one wave's LDS operations execute in order)
  // No row active anywhere in the wave: nothing to add to the rows of M (one scalar branch around the whole product; r05 - merged with an
  // all-zero accumulator the compiler cleared sixteen registers in front of the branch on every Hessian)
  if (tm != 0u) {
    Ops o0[8], o1[8];
    load8(0, o0);
    load8(8, o1);
    if (tm & 0x00FFu) {
      issue8(o0, std::true_type{});
      if (tm & 0xFF00u) issue8(o1, std::false_type{});
    } else {
      issue8(o1, std::true_type{});
    }
  PROF_CTX(pc, 25);
  // ---- transpose: register 4 b + r of lane (g, j) holds entry 4 g + r of column j of block b; wanted: lane (b, j) holds entries
  // 4 g' + r, g' = 0..3, of column j of ITS block.  Per r a 4 x 4 transpose of 16-lane row pieces among four registers: two
  // v_permlane16_swap (rows 0 <-> 1, 2 <-> 3 between register pairs) and two v_permlane32_swap (rows {2, 3} <-> {0, 1}).
  float R[16];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    auto s01 = __builtin_amdgcn_permlane16_swap(__float_as_uint(C[r]), __float_as_uint(C[4 + r]), false, false);
    auto s23 = __builtin_amdgcn_permlane16_swap(__float_as_uint(C[8 + r]), __float_as_uint(C[12 + r]), false, false);
    auto t02 = __builtin_amdgcn_permlane32_swap(s01[0], s23[0], false, false);
    auto t13 = __builtin_amdgcn_permlane32_swap(s01[1], s23[1], false, false);
    R[r] = __uint_as_float(t02[0]); R[4 + r] = __uint_as_float(t13[0]); R[8 + r] = __uint_as_float(t02[1]); R[12 + r] = __uint_as_float(t13[1]);
  }
  // ---- the rows go home.  Lanes of the rows 0 / 2 of the wave (an environment's first 16 lanes) hold pair 0's block, rows 1 / 3 pair 1's.
  constexpr int BH0 = S::p2_banks & 15, BH1 = (S::p2_banks >> 4) & 15, BC0 = (S::p2_banks >> 8) & 15, BC1 = (S::p2_banks >> 12) & 15;
  p2_fence(R);
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int c0 = S::p2_col[k], c1 = S::p2_col[16 + k];
    if (c0 >= 0) p2_add<0, 0x5, BH0>(a[c0], R[k]);   // pair 0's rows in their own lanes
    if (c1 >= 0) p2_add<0, 0xA, BH1>(a[c1], R[k]);   // pair 1's
  }
  // The other half of the rows sits one 16-lane row away: pair 1's rows of dofs that live in rows 0 / 2 and pair 0's rows of dofs
  // that live in rows 1 / 3.  Registers are exchanged in PAIRS (k, k'), two v_permlane16_swap per pair and no copies (the home adds
  // above were their last readers): the first swap leaves k's pair-1 entries in rows 0 / 2 of one register and k''s pair-0
  // entries in rows 1 / 3 of the other; swapping the two results once more gives the other two combinations.
  constexpr int NK = [] { int n = 0; for (int k = 0; k < 16; ++k) n += (S::p2_col[k] >= 0 || S::p2_col[16 + k] >= 0) ? 1 : 0; return n; }();
  static_assert(NK >= 1 && NK <= 16, "two-pair form: columns");
  // (the planner fills the registers from 0 upward: the used ones are 0 .. NK - 1)
  // Two rounds, so that every swap works in place: all first swaps, their four adds per pair (the last readers of the results),
  // then all second swaps on the same registers, and the remaining adds.
  // (P, Q: one register per PAIR of columns, entries k / 2; r05: sixteen-entry arrays passed whole through the fences had their unused half
  //  cleared with v_mov on every Hessian)
  constexpr int NP = (NK + 1) / 2;
  float P[8], Q[8];
#pragma unroll
  for (int k = 0; k < NK; k += 2) {
    const int k2 = k + 1 < NK ? k + 1 : k;  // an odd register count: the last one pairs with itself (both results come out of one swap)
    auto s1 = __builtin_amdgcn_permlane16_swap(__float_as_uint(R[k]), __float_as_uint(R[k2]), false, false);  // (k.r0, k'.r0), (k.r1, k'.r1)
    P[k / 2] = __uint_as_float(s1[0]); Q[k / 2] = __uint_as_float(s1[1]);
  }
  p2_fence_n<NP>(P);
  p2_fence_n<NP>(Q);
#pragma unroll
  for (int k = 0; k < NK; k += 2) {
    const int k2 = k + 1 < NK ? k + 1 : k;
    // rows 0 / 2 read k.r1 = pair 1's entries of register k; rows 1 / 3 read k'.r0 = pair 0's entries of register k'
    if (S::p2_col[16 + k] >= 0) p2_add<S::p2_shl1, 0x5, BC0>(a[S::p2_col[16 + k] >= 0 ? S::p2_col[16 + k] : 0], Q[k / 2]);
    if (S::p2_col[k2] >= 0) p2_add<S::p2_shl0, 0xA, BC1>(a[S::p2_col[k2] >= 0 ? S::p2_col[k2] : 0], P[k / 2]);
  }
  constexpr int NP2 = NK / 2;  // pairs with two distinct registers: the second round of swaps
#pragma unroll
  for (int k = 0; k + 1 < NK; k += 2) {
    auto s2 = __builtin_amdgcn_permlane16_swap(__float_as_uint(Q[k / 2]), __float_as_uint(P[k / 2]), false, false);  // (k.r1, k.r0), (k'.r1, k'.r0)
    P[k / 2] = __uint_as_float(s2[0]); Q[k / 2] = __uint_as_float(s2[1]);
  }
  p2_fence_n<NP2>(P);
  p2_fence_n<NP2>(Q);
#pragma unroll
  for (int k = 0; k + 1 < NK; k += 2) {
    // rows 1 / 3 read k.r0 = pair 0's entries of register k; rows 0 / 2 read k'.r1 = pair 1's entries of register k'
    if (S::p2_col[k] >= 0) p2_add<S::p2_shl0, 0xA, BC1>(a[S::p2_col[k] >= 0 ? S::p2_col[k] : 0], P[k / 2]);
    if (S::p2_col[16 + k + 1] >= 0) p2_add<S::p2_shl1, 0x5, BC0>(a[S::p2_col[16 + k + 1] >= 0 ? S::p2_col[16 + k + 1] : 0], Q[k / 2]);
  }
  }
  PROF_CTX(pc, 26);
  const float b = owner ? a[NS] : 0.0f;  // the gradient ma - fs - flim - J'f
  const float fcon = ma_v - fs_v - b;    // = flim + J'f of the contact rows
  if (owner && store_ok) { lds[fcon_off + gl * DOF_E] = fcon; lds[rhs_off + gl * DOF_E] = b; }
  // the mirrored right-hand-side row: lane NS holds b as a row - staged through LDS, the store in front of the stopping test so that
  // its round trip runs under that test's group sum
  if (gl < lds_) lds[bt_off + gl] = b;
  if (early_out) {
    const float gn = gsum<G>(b * b);
    const float ts_ = scale * tol;
    const bool done = imp_done | (gn < ts_ * ts_);
    if (group_done) *group_done = done;
    if (!wave_any(store_ok & !done)) { WSYNC(); return true; }
  }
  WSYNC();
  if (brow) {
    const float4 *rowp = reinterpret_cast<const float4 *>(lds + bt_off);
#pragma unroll
    for (int c = 0; c < (NS + 3) / 4; ++c) {
      const float4 v = rowp[c];
      const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) if (4 * c + q < NS) a[4 * c + q] = e[q];
    }
  }
  a[NS] = b;
  float d = 1.0f;
  sym_eliminate<S, G, TREE, true>(a, gle, d);
  PROF_CTX(pc, 27);
  // the direction and, in the adjacent field of the dof record, the search vector -direction: the caller's copy pass (a load, a
  // store and a wave sync between the solve and the line search) is not needed
  // (Env::D_SEARCH == Env::D_MGRAD + 1: asserted there)
  if (owner) { const float xv = fdiv(a[NS], AMX_PIVOT_CLAMP ? d : fmaxf(d, MINVAL)); lds[x_off + gl * DOF_E] = xv; lds[x_off + gl * DOF_E + 1] = -xv; }
  WSYNC();
  return false;
}

template <int G>
__device__ __noinline__ void newton_fused_dyn(int nchol, int M_off, int ld, int n, unsigned long long crows, int row_base, int J_base, int ldc, int slot_base,
                                              int rhs_off, int x_off, float dlim, int gl, int gb4, int rot) {
#define AMX_NF_CASE(NN) else if (nchol <= NN) newton_solve_fused<DynSpec, NN, G, false>(M_off, ld, n, crows, row_base, J_base, ldc, slot_base, rhs_off, x_off, dlim, gl, gb4, rot, 0);
  if (nchol <= 4) newton_solve_fused<DynSpec, 4, G, false>(M_off, ld, n, crows, row_base, J_base, ldc, slot_base, rhs_off, x_off, dlim, gl, gb4, rot, 0);
  AMX_NF_CASE(6) AMX_NF_CASE(8) AMX_NF_CASE(10) AMX_NF_CASE(12) AMX_NF_CASE(14) AMX_NF_CASE(16) AMX_NF_CASE(18)
  AMX_NF_CASE(20) AMX_NF_CASE(22) AMX_NF_CASE(24) AMX_NF_CASE(26) AMX_NF_CASE(28) AMX_NF_CASE(30) AMX_NF_CASE(32)
#undef AMX_NF_CASE
}

template <int G>
__device__ __noinline__ void spd_solve(int nchol, int A_off, int ld, int n, int rhs_off, int x_off, int gl, int gb4, int rot) {
  const int vstride = DOF_E;
  // one instantiation per even size (a pivot costs a crossbar round trip, so padding rows are not free);
  // a group of G lanes holds at most G rows
#define AMX_SPD_CASE(NN) else if (nchol <= NN && NN <= G) spd_solve_rows<(NN <= G ? NN : 4), G>(A_off, ld, n, rhs_off, x_off, vstride, gl, gb4, rot);
  if (nchol <= 4) spd_solve_rows<4, G>(A_off, ld, n, rhs_off, x_off, vstride, gl, gb4, rot);
  AMX_SPD_CASE(6) AMX_SPD_CASE(8) AMX_SPD_CASE(10) AMX_SPD_CASE(12) AMX_SPD_CASE(14) AMX_SPD_CASE(16) AMX_SPD_CASE(18)
  AMX_SPD_CASE(20) AMX_SPD_CASE(22) AMX_SPD_CASE(24) AMX_SPD_CASE(26) AMX_SPD_CASE(28) AMX_SPD_CASE(30) AMX_SPD_CASE(32)
#undef AMX_SPD_CASE
}

// one out-of-line dispatcher so the call sites stay small (CG preconditioner only)
template <int G>
__device__ __noinline__ void chol_solve(int nchol, int A_off, int ld, int n, int L_off, int rhs_off, int x_off, bool factor, int gl, int gb4) {
  const int vstride = DOF_E;
  if (nchol <= 4) chol_solve_rows<4, G>(A_off, ld, n, L_off, rhs_off, x_off, vstride, factor, gl, gb4);
  else if (nchol <= 8) chol_solve_rows<8, G>(A_off, ld, n, L_off, rhs_off, x_off, vstride, factor, gl, gb4);
  else if (nchol <= 12) chol_solve_rows<12, G>(A_off, ld, n, L_off, rhs_off, x_off, vstride, factor, gl, gb4);
  else if (nchol <= 16) chol_solve_rows<(G >= 16 ? 16 : 12), G>(A_off, ld, n, L_off, rhs_off, x_off, vstride, factor, gl, gb4);
  else if (nchol <= 20) chol_solve_rows<(G >= 32 ? 20 : 12), G>(A_off, ld, n, L_off, rhs_off, x_off, vstride, factor, gl, gb4);
  else if (nchol <= 24) chol_solve_rows<(G >= 32 ? 24 : 12), G>(A_off, ld, n, L_off, rhs_off, x_off, vstride, factor, gl, gb4);
  else if (nchol <= 28) chol_solve_rows<(G >= 32 ? 28 : 12), G>(A_off, ld, n, L_off, rhs_off, x_off, vstride, factor, gl, gb4);
  else chol_solve_rows<(G >= 32 ? 32 : 12), G>(A_off, ld, n, L_off, rhs_off, x_off, vstride, factor, gl, gb4);
}

// impedance of one constraint row [A9.6]; k and b are folded on the host, rec points at {k,b,dmin,dmax,1/width,mid,power,ca,cb}
DEV float row_impedance(int rec, float pos) {
  const float dmin = CF(rec + 2), dmax = CF(rec + 3), inv_width = CF(rec + 4), mid = CF(rec + 5), power = CF(rec + 6);
  const float x = fabsf(pos) * inv_width;
  float ia, ib;
  if (power == 2.0f) { ia = CF(rec + 7) * (x * x); ib = 1.0f - CF(rec + 8) * ((1 - x) * (1 - x)); }
  else { ia = CF(rec + 7) * powf(x, power); ib = 1.0f - CF(rec + 8) * powf(1 - x, power); }
  float v = dmin + (x < mid ? ia : ib) * (dmax - dmin);
  v = fminf(fmaxf(v, dmin), dmax);
  return x > 1.0f ? dmax : v;
}


// Item loops (one lane per item, items t = gl, gl+G, ...): `fetch(t)` only loads and computes, `commit(t, v)` stores.
// Two items are fetched before either commits, so the LDS round trips of consecutive items overlap instead of
// serialising behind each item's store (the compiler cannot hoist a load above a may-alias LDS store itself).
// Out-of-range items fetch item n-1 again (valid addresses, no exec-mask branch) and skip the commit.
template <int G, class F, class C> DEV void item_loop2(int n, int gl, F fetch, C commit) {
  for (int t0 = 0; t0 < n; t0 += 2 * G) {
    const int ta = t0 + gl, tb = ta + G;
    const auto va = fetch(ta < n ? ta : n - 1);
    const auto vb = fetch(tb < n ? tb : n - 1);
    if (ta < n) commit(ta, va);
    if (tb < n) commit(tb, vb);
  }
}
// lds[dst + e] = lds[src + e], e < n; dst, src and n are multiples of 4 floats: 16-byte moves, four in flight per lane
template <int G> DEV void lds_copy(int dst, int src, int n, int gl) {
  const float4 *s = reinterpret_cast<const float4 *>(lds + src);
  float4 *d = reinterpret_cast<float4 *>(lds + dst);
  const int n4 = n >> 2;
  for (int e0 = 0; e0 < n4; e0 += 4 * G) {
    float4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { const int e = e0 + k * G + gl; v[k] = s[e < n4 ? e : n4 - 1]; }
#pragma unroll
    for (int k = 0; k < 4; ++k) { const int e = e0 + k * G + gl; if (e < n4) d[e] = v[k]; }
  }
}

// max over each 32-lane unit (a G = 32 group, or half of a G = 64 group)
template <int U> DEV float umax(float x) {  // max over the U-lane unit (16: one DPP row; 32: two rows joined through the crossbar)
  x = fmaxf(x, dpp_f<0xB1>(x)); x = fmaxf(x, dpp_f<0x4E>(x)); x = fmaxf(x, dpp_f<0x141>(x)); x = fmaxf(x, dpp_f<0x140>(x));
  if (U == 32) x = fmaxf(x, __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(x), 0x401F)));
  return x;
}
// plane - convex hull with one lane per hull vertex (MJX's 4-point manifold: deepest-1mm mask, a = first masked, b =
// farthest from a, c = farthest from line ab, d = farthest from the other two edges).  Every selection is a max over the
// unit, a ballot of the lanes that hold it and the lowest such lane - the same vertex the serial `>` scan keeps - and
// the winner's coordinates come back through the crossbar.  Used by 64-lane groups, two pairs per pass (the code also
// handles one pair per pass on a 32-lane unit).  Out of line: models without such pairs do not pay for its registers.
//   geom_off / con_off / frame_off: LDS offsets of the lane's environment's geom poses, contacts and contact frames
template <int G, int U>  // U lanes per pair: 32, or 16 when every hull has at most 16 vertices (two pairs per 32-lane pass)
DEV void plane_convex_parallel_body(int pcx_first, int pcx_num, int c_pair, int c_vert, int geom_off, int con_off, int frame_off, int gl, int gb4) {
  const int unit = gl / U, v = gl % U, per_pass = G / U;
  const int lane0 = (gb4 >> 2) + U * unit;  // first wave lane of my unit
#define GEO(f, g) lds[geom_off + (g) * GEO_E + (f)]
  for (int p0 = 0; p0 < pcx_num; p0 += per_pass) {
    const bool pair_ok = p0 + unit < pcx_num;
    const int pr = pcx_first + (pair_ok ? p0 + unit : 0), rec = c_pair + pr * PAIR_S;
    const int g1 = CI(rec), g2 = CI(rec + 1), c0 = CI(rec + 3), nvt = CI(rec + 6), va = c_vert + 3 * CI(rec + 5);
    const f3 pos1 = mk3(GEO(0, g1), GEO(1, g1), GEO(2, g1)), pos2 = mk3(GEO(0, g2), GEO(1, g2), GEO(2, g2));
    m33 R2;
#pragma unroll
    for (int k = 0; k < 9; ++k) R2.a[k] = GEO(3 + k, g2);
    const f3 nw = mk3(GEO(3 + 2, g1), GEO(3 + 5, g1), GEO(3 + 8, g1));
    const f3 pl = mulTv(R2, pos1 - pos2), n = mulTv(R2, nw);
    const bool on = pair_ok && v < nvt;
    const int vi = on ? v : 0;
    const f3 vt = mk3(CF(va + 3 * vi), CF(va + 3 * vi + 1), CF(va + 3 * vi + 2));
    const float sup = dot(pl - vt, n);
    const float smax = umax<U>(on ? sup : -1e30f);
    const float thr = fmaxf(0.0f, smax - 1e-3f);
    const float mask = sup > thr ? 0.0f : -1e6f;
    // lowest lane of my unit whose score equals the unit's maximum; returns that vertex
    auto pick = [&](float sc, int *idx) {
      const float s = on ? sc : -1e30f;
      const float best = umax<U>(s);
      const unsigned long long bal = __ballot(s == best && on);
      const unsigned mine = (unsigned)(bal >> (lane0 & 63)) & (U == 32 ? 0xFFFFFFFFu : 0xFFFFu);  // my unit's bits start at its first lane
      const int w = mine ? __builtin_ctz(mine) : 0;
      *idx = w;
      const int addr = 4 * (lane0 + w);
      return mk3(__int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(vt.x))), __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(vt.y))),
                 __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(vt.z))));
    };
    int idx[4];
    const f3 a = pick(mask, &idx[0]);
    const f3 da = a - vt;
    const f3 bb = pick(dot(da, da) + mask, &idx[1]);
    const f3 ab = cross(n, a - bb);
    const f3 cc = pick(fabsf(dot(a - vt, ab)) + mask, &idx[2]);
    const f3 ac = cross(n, a - cc), bc = cross(n, bb - cc);
    // d: the serial scan runs over the bc scores first, then the ac scores continue with the same running maximum
    const float s1 = fabsf(dot(bb - vt, bc)) + mask, s2 = fabsf(dot(a - vt, ac)) + mask;
    const float m1 = umax<U>(on ? s1 : -1e30f), m2 = umax<U>(on ? s2 : -1e30f);
    const f3 dd = pick(m2 > m1 ? s2 : s1, &idx[3]);
    const f3 sel[4] = {a, bb, cc, dd};
    if (pair_ok && v < 4) {  // lanes 0..3 of the unit write the four contacts
      f3 me = sel[0];
      int my = idx[0];
      bool unique = true;
#pragma unroll
      for (int k = 1; k < 4; ++k) { if (v == k) { me = sel[k]; my = idx[k]; } }
#pragma unroll
      for (int j = 0; j < 3; ++j) unique = unique && !(j < v && idx[j] == my);
      const float d = unique ? -dot(pl - me, n) : 1.0f;
      const f3 cp = pos2 + mulv(R2, me) - nw * (0.5f * d);
      const int co = con_off + (c0 + v) * CON_E;
      lds[co] = d; lds[co + 1] = cp.x; lds[co + 2] = cp.y; lds[co + 3] = cp.z;
    }
    if (pair_ok && v == 0) {
      float fr[9];
      make_frame(nw, fr);
#pragma unroll
      for (int k = 0; k < 9; ++k) lds[frame_off + pr * 9 + k] = fr[k];
    }
  }
#undef GEO
  WSYNC();
}
// out-of-line entry for the generic kernels (the specialised ones inline the body: see AMX_STATIC_SOLVE_ATTR)
template <int G, int U>
__device__ __noinline__ void plane_convex_parallel(int pcx_first, int pcx_num, int c_pair, int c_vert, int geom_off, int con_off, int frame_off, int gl, int gb4) {
  plane_convex_parallel_body<G, U>(pcx_first, pcx_num, c_pair, c_vert, geom_off, con_off, frame_off, gl, gb4);
}

// ---- group max / min with the value in every lane (same network as gsum)
template <int G> DEV float gmaxf(float x) {
  x = fmaxf(x, dpp_f<0xB1>(x)); x = fmaxf(x, dpp_f<0x4E>(x)); x = fmaxf(x, dpp_f<0x141>(x)); x = fmaxf(x, dpp_f<0x140>(x));
  if (G >= 32) x = fmaxf(x, __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(x), 0x401F)));
  if (G >= 64) x = fmaxf(x, __shfl_xor(x, 32));
  return x;
}
template <int G> DEV float gminf(float x) { return -gmaxf<G>(-x); }
// bits of a wave ballot that belong to my group, shifted down to bit 0
template <int G> DEV unsigned long long gballot(bool p, int gb4) {
  const unsigned long long b = __ballot(p) >> ((gb4 >> 2) & 63);
  return G == 64 ? b : (b & ((1ull << (G & 63)) - 1ull));
}
template <int G> DEV f3 gfetch3(f3 v, int lane_in_group, int gb4) {  // v of lane `lane_in_group` of my group (per-lane source: crossbar)
  const int a = gb4 + 4 * lane_in_group;
  return mk3(__int_as_float(__builtin_amdgcn_ds_bpermute(a, __float_as_int(v.x))), __int_as_float(__builtin_amdgcn_ds_bpermute(a, __float_as_int(v.y))),
             __int_as_float(__builtin_amdgcn_ds_bpermute(a, __float_as_int(v.z))));
}

// A9.5c convex - convex (box / mesh hulls): separating-axis test over the face normals of both hulls and the cross products
// of their edge pairs (an edge pair counts only if both edges are supporting edges along its axis), axis of least penetration,
// then either the face contact - reference face, incident face, Sutherland-Hodgman clipping against the reference face's side
// planes, 4-point manifold like plane_convex's - or the edge-edge contact.  Same decisions and tie rules as the oracle
// (oracle/amx_oracle.c, convex_convex).  The whole lane group works on one pair at a time: lanes over hull vertices for the
// world transform, over axes for the SAT (the argmin is a group reduction, ties to the lowest axis index), over faces for the
// reference / incident choice, over polygon edges for the clipping (ballot prefix counts compact the output in order) and
// over polygon points for the manifold picks.  Scratch (world vertices, two clip polygons) lives in the slab (e_cx).
template <int G>
__device__ __noinline__ void convex_convex_pairs(int cx_first, int ncx, int cx_maxv, int c_pair, int geom_off, int con_off, int frame_off, int scr, int gl, int gb4) {
#define GEO(f, g) lds[geom_off + (g) * GEO_E + (f)]
#define HV(base, i) mk3(lds[(base) + 3 * (i)], lds[(base) + 3 * (i) + 1], lds[(base) + 3 * (i) + 2])
  const int VA = scr, VB = scr + 3 * cx_maxv, POLY = scr + 6 * cx_maxv;
  for (int pp = 0; pp < ncx; ++pp) {
    const int pr = cx_first + pp, rec = c_pair + pr * PAIR_S;
    const int g1 = CI(rec), g2 = CI(rec + 1), c0 = CI(rec + 3), hA = CI(rec + 25), hB = CI(rec + 26);
    const int nvA = CI(hA), vA = CI(hA + 1), nfA = CI(hA + 2), nA_ = CI(hA + 3), pA = CI(hA + 4), neA = CI(hA + 5), eA = CI(hA + 6);
    const int nvB = CI(hB), vB = CI(hB + 1), nfB = CI(hB + 2), nB_ = CI(hB + 3), pB = CI(hB + 4), neB = CI(hB + 5), eB = CI(hB + 6);
    const f3 pos1 = mk3(GEO(0, g1), GEO(1, g1), GEO(2, g1)), pos2 = mk3(GEO(0, g2), GEO(1, g2), GEO(2, g2));
    m33 R1, R2;
#pragma unroll
    for (int k = 0; k < 9; ++k) { R1.a[k] = GEO(3 + k, g1); R2.a[k] = GEO(3 + k, g2); }
    // 1. world vertices
    for (int v = gl; v < nvA; v += G) { const f3 w = pos1 + mulv(R1, HV(vA, v)); lds[VA + 3 * v] = w.x; lds[VA + 3 * v + 1] = w.y; lds[VA + 3 * v + 2] = w.z; }
    for (int v = gl; v < nvB; v += G) { const f3 w = pos2 + mulv(R2, HV(vB, v)); lds[VB + 3 * v] = w.x; lds[VB + 3 * v + 1] = w.y; lds[VB + 3 * v + 2] = w.z; }
    WSYNC();
    // 2. separating axes, lane-strided.  Selection rule (as in the oracle): the lowest axis index whose penetration is within
    // a tolerance of the least one - exact ties (parallel faces) must not be left to rounding.  Pass 1: the least penetration
    // (group min); pass 2: every lane's lowest index within tolerance, then the group's lowest.
    const int nface = nfA + nfB, naxes = nface + neA * neB;
    struct Axis { bool ok; float pen, sg; f3 n; };
    auto eval_axis = [&](int t) {
      const bool valid = t < naxes;
      const int tc = valid ? t : 0;
      f3 n;
      bool ok = valid;
      int ea = 0, eb = 0;
      if (tc < nfA) n = mulv(R1, HV(nA_, tc));
      else if (tc < nface) n = mulv(R2, HV(nB_, tc - nfA));
      else {
        ea = (tc - nface) / neB; eb = (tc - nface) - ea * neB;
        float l;
        const f3 da = normalize3(HV(VA, CI(eA + 2 * ea + 1)) - HV(VA, CI(eA + 2 * ea)), &l);
        const f3 db = normalize3(HV(VB, CI(eB + 2 * eb + 1)) - HV(VB, CI(eB + 2 * eb)), &l);
        n = cross(da, db);
        const float c2 = dot(n, n);
        ok = ok && !(c2 < 1e-6f);
        n = n * (1.0f / sqrtf(fmaxf(c2, 1e-30f)));
      }
      float loA = 1e30f, hiA = -1e30f, loB = 1e30f, hiB = -1e30f;
      for (int i = 0; i < nvA; ++i) { const float sdot = dot(HV(VA, i), n); loA = fminf(loA, sdot); hiA = fmaxf(hiA, sdot); }
      for (int i = 0; i < nvB; ++i) { const float sdot = dot(HV(VB, i), n); loB = fminf(loB, sdot); hiB = fmaxf(hiB, sdot); }
      const float d1 = hiA - loB, d2 = hiB - loA;
      const float pen = d1 < d2 ? d1 : d2, sg = d1 > d2 ? -1.0f : 1.0f;
      if (tc >= nface) {
        const float sa = dot(HV(VA, CI(eA + 2 * ea)), n), sb = dot(HV(VB, CI(eB + 2 * eb)), n);
        const float supA = sg > 0 ? hiA : loA, supB = sg > 0 ? loB : hiB;
        ok = ok && !(fabsf(sa - supA) > 1e-5f || fabsf(sb - supB) > 1e-5f);
      }
      return Axis{ok, pen, sg, n};
    };
#define CX_TOL(x) (1e-4f * fabsf(x) + 1e-7f)
    float mybest = 1e30f;
    for (int t0 = 0; t0 < naxes; t0 += G) {
      const Axis ax = eval_axis(t0 + gl);
      if (ax.ok && ax.pen < mybest) mybest = ax.pen;
    }
    const float best = gminf<G>(mybest);
    const float lim = best + CX_TOL(best);
    float mysg = 1.0f;
    int myt = 0x7fffff;
    f3 myn = mk3(0, 0, 1);
    for (int t0 = 0; t0 < naxes; t0 += G) {
      const Axis ax = eval_axis(t0 + gl);
      if (ax.ok && ax.pen <= lim && myt == 0x7fffff) { myt = t0 + gl; mysg = ax.sg; myn = ax.n; }
    }
    const float tbest = gminf<G>((float)myt);  // axis indices are exact in fp32
    const unsigned long long own = gballot<G>((float)myt == tbest, gb4);
    const int owner = own ? __builtin_ctzll(own) : 0;
    const int bt = (int)tbest;
    const f3 bn = gfetch3<G>(myn, owner, gb4);
    const float bsg = __int_as_float(__builtin_amdgcn_ds_bpermute(gb4 + 4 * owner, __float_as_int(mysg)));
    const f3 N = bn * bsg;  // from geom1 to geom2
    float cd[4] = {1.0f, 1.0f, 1.0f, 1.0f};
    const f3 mid = (pos1 + pos2) * 0.5f;
    f3 cpos[4] = {mid, mid, mid, mid};
    f3 fn_ = N;
    if (bt >= nface) {
      // 3a. edge - edge: closest points of the two segments (computed by every lane of the group alike)
      const int ea = (bt - nface) / neB, eb = (bt - nface) - ea * neB;
      const f3 a0 = HV(VA, CI(eA + 2 * ea)), a1 = HV(VA, CI(eA + 2 * ea + 1)), b0 = HV(VB, CI(eB + 2 * eb)), b1 = HV(VB, CI(eB + 2 * eb + 1));
      const f3 u = a1 - a0, v = b1 - b0, r = a0 - b0;
      const float a = dot(u, u), e = dot(v, v), f = dot(v, r), c = dot(u, r), b = dot(u, v), den = a * e - b * b;
      float s_ = den > 1e-12f * a * e ? fminf(fmaxf((b * f - c * e) / den, 0.0f), 1.0f) : 0.0f;
      float t_ = (b * s_ + f) / e;
      if (t_ < 0) { t_ = 0; s_ = fminf(fmaxf(-c / a, 0.0f), 1.0f); }
      else if (t_ > 1) { t_ = 1; s_ = fminf(fmaxf((b - c) / a, 0.0f), 1.0f); }
      const f3 pa = a0 + u * s_, pb = b0 + v * t_;
      cd[0] = dot(pb - pa, N);
      cpos[0] = (pa + pb) * 0.5f;
    } else {
      // 3b. face contact.  Reference face: best aligned with the axis among both hulls (hull 1 along +N, hull 2 along -N)
      float sA = -1e30f, sB = -1e30f;
      int fA = 0x7fffff, fB = 0x7fffff;
      for (int f = gl; f < nfA; f += G) { const float sdot = dot(mulv(R1, HV(nA_, f)), N); if (sdot > sA) { sA = sdot; fA = f; } }
      for (int f = gl; f < nfB; f += G) { const float sdot = -dot(mulv(R2, HV(nB_, f)), N); if (sdot > sB) { sB = sdot; fB = f; } }
      const float da = gmaxf<G>(sA), db = gmaxf<G>(sB);
      const int fa = (int)gminf<G>(sA == da ? (float)fA : 1e9f), fb = (int)gminf<G>(sB == db ? (float)fB : 1e9f);
      const bool refA = da >= db - 1e-6f;
      const int rf = refA ? fa : fb;
      const f3 nr = refA ? mulv(R1, HV(nA_, rf)) : mulv(R2, HV(nB_, rf));
      // incident face on the other hull: most anti-parallel to the reference normal
      const int nfI = refA ? nfB : nfA, nI_ = refA ? nB_ : nA_;
      float sI = 1e30f;
      int fI = 0x7fffff;
      for (int f = gl; f < nfI; f += G) { const float sdot = dot(refA ? mulv(R2, HV(nI_, f)) : mulv(R1, HV(nI_, f)), nr); if (sdot < sI) { sI = sdot; fI = f; } }
      const float dmin = gminf<G>(sI);
      const int fi = (int)gminf<G>(sI == dmin ? (float)fI : 1e9f);
      const int polyI = (refA ? pB : pA) + 9 * fi, polyR = (refA ? pA : pB) + 9 * rf, VI = refA ? VB : VA, VR = refA ? VA : VB;
      int np_ = CI(polyI);
      const int nr_ = CI(polyR);
      int cur = POLY, oth = POLY + 3 * CX_MAXP;
      if (gl < np_) { const f3 w = HV(VI, CI(polyI + 1 + gl)); lds[cur + 3 * gl] = w.x; lds[cur + 3 * gl + 1] = w.y; lds[cur + 3 * gl + 2] = w.z; }
      WSYNC();
      for (int k = 0; k < 8; ++k) {  // Sutherland-Hodgman against the side plane through reference edge k (lane = polygon edge)
        const bool on = k < nr_ && np_ > 0;
        if (!wave_any(on)) break;
        const f3 r0 = HV(VR, CI(polyR + 1 + (on ? k : 0))), r1 = HV(VR, CI(polyR + 1 + (on && k + 1 < nr_ ? k + 1 : 0)));
        const f3 mk = cross(r1 - r0, nr);  // outward of the reference polygon
        const bool mine = on && gl < np_;
        const int i0 = mine ? gl : 0, i1 = (mine && gl + 1 < np_) ? gl + 1 : 0;
        const f3 pc = HV(cur, i0), pn = HV(cur, i1);
        const float sc = dot(pc - r0, mk), sn = dot(pn - r0, mk);
        const bool emitA = mine && sc <= 0, emitB = mine && ((sc <= 0) != (sn <= 0));
        const unsigned long long bA = gballot<G>(emitA, gb4), bB = gballot<G>(emitB, gb4);
        const unsigned long long below = (1ull << gl) - 1ull;
        int off = __popcll(bA & below) + __popcll(bB & below);
        if (emitA && off < CX_MAXP) { lds[oth + 3 * off] = pc.x; lds[oth + 3 * off + 1] = pc.y; lds[oth + 3 * off + 2] = pc.z; }
        off += emitA ? 1 : 0;
        if (emitB && off < CX_MAXP) {
          const float tt = sc / (sc - sn);
          const f3 q = pc + (pn - pc) * tt;
          lds[oth + 3 * off] = q.x; lds[oth + 3 * off + 1] = q.y; lds[oth + 3 * off + 2] = q.z;
        }
        if (on) {
          const int cnt = __popcll(bA) + __popcll(bB);
          np_ = cnt < CX_MAXP ? cnt : CX_MAXP;
          const int tmp = cur; cur = oth; oth = tmp;
        }
        WSYNC();
      }
      if (np_ > 0) {
        const bool on = gl < np_;
        const f3 me = HV(cur, on ? gl : 0);
        const f3 r0 = HV(VR, CI(polyR + 1));
        const float depth = dot(r0 - me, nr);
        const float dmax = gmaxf<G>(on ? depth : -1e30f);
        const float mask = (dmax >= 0 ? depth >= 0 : depth == dmax) ? 0.0f : -1e6f;
        // lowest lane of my group whose score equals the group's maximum; returns that point and its depth
        auto pick = [&](float sc_, int *idx, float *dep) {
          const float s2 = on ? sc_ : -1e30f;
          const float bst = gmaxf<G>(s2);
          const unsigned long long bal = gballot<G>(on && s2 >= bst - CX_TOL(bst), gb4);  // lowest index within tolerance of the maximum
          const int w = bal ? __builtin_ctzll(bal) : 0;
          *idx = w;
          *dep = __int_as_float(__builtin_amdgcn_ds_bpermute(gb4 + 4 * w, __float_as_int(depth)));
          return gfetch3<G>(me, w, gb4);
        };
        int idx[4];
        float dep[4];
        const f3 a = pick(mask, &idx[0], &dep[0]);
        const f3 dq = a - me;
        const f3 bb = pick(dot(dq, dq) + mask, &idx[1], &dep[1]);
        const f3 ab = cross(nr, a - bb);
        const f3 cc = pick(fabsf(dot(a - me, ab)) + mask, &idx[2], &dep[2]);
        const f3 ac = cross(nr, a - cc), bc = cross(nr, bb - cc);
        const float s1 = fabsf(dot(bb - me, bc)) + mask, s2 = fabsf(dot(a - me, ac)) + mask;
        const float m1 = gmaxf<G>(on ? s1 : -1e30f), m2 = gmaxf<G>(on ? s2 : -1e30f);
        const f3 dd = pick(m2 > m1 + CX_TOL(m1) ? s2 : s1, &idx[3], &dep[3]);  // farthest from edge bc, or - clearly farther - from edge ac
        const f3 sel[4] = {a, bb, cc, dd};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          bool unique = true;
#pragma unroll
          for (int j = 0; j < 4; ++j) unique = unique && !(j < k && idx[j] == idx[k]);
          if (unique) { cd[k] = -dep[k]; cpos[k] = sel[k] + nr * (0.5f * dep[k]); }
        }
#if defined(AMX_CX_DEBUG) && AMX_CX_DEBUG >= 2
        for (int k = 0; k < 4; ++k) {
          if (AMX_CX_DEBUG == 4) cd[k] = (float)idx[k];
          else cd[k] = __int_as_float(__builtin_amdgcn_ds_bpermute(gb4 + 4 * (k + (AMX_CX_DEBUG == 3 ? 4 : 0)), __float_as_int(on ? depth : -9.0f)));
        }
#endif
      }
      fn_ = refA ? nr : nr * -1.0f;
#if defined(AMX_CX_DEBUG) && AMX_CX_DEBUG == 1
      cd[1] = (float)((refA ? 1000 : 0) + 10 * rf + 0); cd[2] = (float)(100 * fi + np_);
#endif
    }
#if defined(AMX_CX_DEBUG) && AMX_CX_DEBUG == 1
    cd[0] = (float)bt; cd[3] = best;
#endif
    if (gl < 4) {
      float dsel = cd[0];
      f3 psel = cpos[0];
#pragma unroll
      for (int k = 1; k < 4; ++k) if (gl == k) { dsel = cd[k]; psel = cpos[k]; }
      const int co = con_off + (c0 + gl) * CON_E;
      lds[co] = dsel; lds[co + 1] = psel.x; lds[co + 2] = psel.y; lds[co + 3] = psel.z;
    }
    if (gl == 0) {
      float fr[9];
      make_frame(fn_, fr);
#pragma unroll
      for (int k = 0; k < 9; ++k) lds[frame_off + pr * 9 + k] = fr[k];
    }
    WSYNC();
  }
#undef GEO
#undef HV
#undef CX_TOL
}

// =====================================================================================================
// One environment's physics, run by a group of G lanes.  `eb` is the group's slab in LDS (per-lane value,
// equal inside a group); `gl` is the lane's index inside its group; constants live at the front of LDS.
// =====================================================================================================
template <int G, class S>
struct Env {
  const MV<S> m;
  const int eb, gl, gb4;  // slab base, lane-in-group, 4 * first lane of the group (crossbar address base)
  // bases of the body and dof records, opaque to the compiler: with `eb + e_body` folded into one constant every pair of fields
  // got its own `v_add_u32` (ds_read2_b32 reaches 1 KB from its base); relative to these the field offsets fit the instruction
  int ebB, ebD;
  // static role constants of this lane
  int b_parent, b_level, b_root, b_jntadr, b_jntnum, b_childadr, b_childnum, b_subend;  // body `lane` (b_subend: one past its last descendant)
  int d_body, d_jnt, d_kind, d_axis, d_root, d_jadr, d_limrow, d_frirow;      // dof `lane` (d_jadr: first dof of its joint; d_limrow / d_frirow: lane of its limit / friction-loss row)
  int m_rot, m_diag;  // row `lane` of M: chunk rotation of the row (0 unless MV::mrot), float offset of its diagonal entry inside M
  unsigned m_win = 0u;  // compact M (MV::mcomp): bit c set when the row stores chunk c (m_row_compact)
  // Constraint rows: a lane owns TWO row slots, so that a group of G lanes carries up to 2 G rows -
  //   slot A: equality / limit row `gl` (gl < nA = neq + nlim): one or two Jacobian entries, kept in registers;
  //   slot B: contact pyramid row `gl` (gl < nB = 4 ncon): its Jacobian row lives in LDS, chain-compressed.
  // (One slot per lane made the 56 rows of the biped - 24 limits, 32 contact rows - need a 64-lane group for a 30-dof model.)
  // Row r of the reference's row table is A row r, or B row r - nA.
  float aD, aAref, aJaref, aJv;  // slot A solver state
  // friction-loss rows (slot A, between the equality and the limit rows; models with nfri > 0 only): the force of such a row is
  // bounded by +-aFl, so its cost is quadratic while |Jaref| < aFl / aD and linear beyond (aFl = 0 on every other row)
  float aFl;
  float cD, cAref, cJaref, cJv;  // slot B solver state
  float cSd = 0.0f, c2Sd = 0.0f;  // sqrt(cD), sqrt(c2D): the matrix cores' operand scale (AMX_SQRT_D)
  // second contact-row slot (r04; per-model kernels with S::nbs == 2 only): row gl + G of the contact rows, same fields
  static constexpr int NBS = S::is_static ? (S::nbs == 2 ? 2 : 1) : 1;
  float c2D, c2Aref, c2Jaref, c2Jv;
  bool c2act;
  int vch_base2;
  DEV bool validB2() const { return NBS == 2 && gl + G < nB(); }
  // contact row `gl`: J*v reads v from the pair's gathered copy (stage_chain) - vch_base: where that copy starts;
  // staging item `gl` (slot t of pair p, gl = p*ldc + t): vch_dof = its dof, -1 past the list's end
  int vch_base, vch_dof;
  float rJ1, rJ2;      // equality / limit row `gl`: Jacobian entries on dofs rd1, rd2
  int rd1, rd2;
  // this lane's dof: M*qacc of the current iterate, qfrc_smooth and qacc_smooth, mirrored in registers (0 on lanes without a dof).  The
  // LDS fields stay current - other routines and the outputs read them - but the solver's own per-dof arithmetic no longer waits for them.
  float r_ma, r_fs, r_as;
  bool aact, cact;  // the slot's row is active at the current iterate (set by constraint_force_and_grad)
  bool cEq;         // the lane's slot B row belongs to a connect / weld equality (a pseudo-pair, [npc, npair - nbl)): active on both sides
  bool cBl;         // ... to a ball joint's limit (pseudo-pair [npair - nbl, npair)): single-sided like a contact row, one row per group
  PROF_DECL
#ifndef AMX_PHASE_PRIO
#define AMX_PHASE_PRIO 1  // 0: issue priority by progress only (r03: the lagging wave of a SIMD's pair at 3, the other at 0), A/B switch
#endif
  // Issue priority by phase, on top of the pair's fair sharing (amx_rollout_kernel): the solve - issue-bound: eliminations, line search -
  // runs one level above the tree passes, whose few instructions wait on LDS round trips either way; the wave that is behind its
  // SIMD partner gets one more level.  Measured (profiles/r04_ab_phase_prio.txt): +1 % on all three biped workloads; the opposite
  // order -0.3 %, without the lagging wave's bonus -1.3 %, finer levels inside the solve (eliminations or line search on top) +-0.
  int prio_lag = 0;  // 1: this wave is behind its SIMD partner (set per step by the rollout loop)
  template <int LEVEL> DEV void phase_prio() const {
    if (AMX_PHASE_PRIO) { if (prio_lag) __builtin_amdgcn_s_setprio(LEVEL + 1); else __builtin_amdgcn_s_setprio(LEVEL); }
  }

  DEV int nA() const { return m.nefc() == 0 ? 0 : m.neq() + m.nfri() + m.nlim(); }  // nefc == 0: constraints disabled
  DEV int nB() const { return m.nefc() == 0 ? 0 : 4 * m.ncon(); }

  DEV Env(const MV<S> &m_, int eb_, int gl_, int gb4_) : m(m_), eb(eb_), gl(gl_), gb4(gb4_) {
    const int lane = gl;
    ebB = eb + m.e_body(); ebD = eb + m.e_dof();
    asm volatile("" : "+v"(ebB), "+v"(ebD));
    const int br = m.c_body() + (lane < m.nb() ? lane : 0) * BODY_S;
    b_parent = CI(br); b_level = lane < m.nb() ? CI(br + 1) : -1; b_root = CI(br + 2); b_jntadr = CI(br + 3); b_jntnum = CI(br + 4);
    b_childadr = CI(br + 5); b_childnum = CI(br + 6) & 255; b_subend = CI(br + 6) >> 8;
    const int dr = m.c_dof() + (lane < m.nv() ? lane : 0) * DOF_S;
    d_body = CI(dr); d_jnt = CI(dr + 1); d_kind = CI(dr + 2); d_axis = CI(dr + 3); d_root = CI(dr + 8); d_jadr = CI(dr + 9); d_limrow = CI(dr + 10); d_frirow = CI(dr + 11);
    {
      const int nch = m.ldm() >> 2, i = lane < m.nv() ? lane : 0;
      m_rot = m.mrot() ? AMX_ROW_ROT(i, nch) : 0;
      m_diag = moff(i, i);
      if (m.mcomp()) {
        const int pk = CI(m.c_mrow() + i), clo = (pk >> 12) & 15, chi = (pk >> 16) & 15;
        m_win = lane < m.nv() ? (2u << chi) - (1u << clo) : 0u;
      }
    }
    aD = aAref = aJaref = aJv = cD = cAref = cJaref = cJv = rJ1 = rJ2 = aFl = 0.0f;
    c2D = c2Aref = c2Jaref = c2Jv = 0.0f; c2act = false;
    r_ma = r_fs = r_as = 0.0f;
    aact = cact = false;
    rd1 = 0; rd2 = -1;
    if (lane < m.neq()) { const int er = m.c_eq() + lane * EQ_S; rd1 = CI(er + 1); rd2 = CI(er + 3); }
    else if (lane < m.neq() + m.nfri()) rd1 = CI(m.c_fri() + (lane - m.neq()) * FRI_S);
    else if (lane < m.neq() + m.nfri() + m.nlim()) rd1 = CI(m.c_lim() + (lane - m.neq() - m.nfri()) * LIM_S + 1);
    vch_base = eb + m.e_vch();
    cEq = false; cBl = false;
    if (lane < 4 * m.ncon()) {
      const int pr_ = CI(m.c_conpair() + (lane >> 2));
      vch_base += pr_ * m.ldc();
      cEq = m.npair() > m.npc() && pr_ >= m.npc() && pr_ < m.npair() - m.nbl();
      cBl = m.nbl() > 0 && pr_ >= m.npair() - m.nbl();
    }
    vch_base2 = eb + m.e_vch();
    if (NBS == 2 && lane + G < 4 * m.ncon()) vch_base2 += CI(m.c_conpair() + ((lane + G) >> 2)) * m.ldc();
    {
      const int nvch = m.npair() * m.ldc();
      vch_dof = CI(m.c_vchdof() + (lane < nvch ? lane : 0));
      vch_dof = lane < nvch ? vch_dof : -1;
    }
  }

  // SoA accessors of the environment slab
  // Records with odd compile-time strides: gl-per-record access is bank-conflict free and every field
  // is an immediate offset from one base address (no per-field address registers).
  DEV float &body(int f, int b) const { return lds[ebB + b * BODY_E + f]; }
  // joint anchor (0-2) and axis (3-5), world frame: fields 6..11 of the record of the joint's first dof `da` (dead solver vectors
  // of the previous step; crb*cdof overwrites them once the motion axes are formed)
  DEV float &jnt(int f, int da) const { return lds[ebD + da * DOF_E + 6 + f]; }
  // float offset of M[i][j] inside the matrix block (rows may store their 16-byte chunks rotated, see amx_plan.h)
  DEV int moff(int i, int j) const {
    if (m.mcomp()) {  // compact storage: < 0 when the entry is not stored (it is zero: neither dof is above the other)
      if (i >= m.mnb0() && j < m.mnb0()) { const int t = i; i = j; j = t; }  // the same number lives in the leading chain's whole row
      const int pk = CI(m.c_mrow() + i), clo = (pk >> 12) & 15, chi = (pk >> 16) & 15;
      return ((j >> 2) < clo || (j >> 2) > chi) ? -1 : (pk & 4095) + j - 4 * clo;
    }
    const int nch = m.ldm() >> 2, rot = m.mrot() ? AMX_ROW_ROT(i, nch) : 0;
    return i * m.ldm() + 4 * (((j >> 2) + rot) % nch) + (j & 3);
  }
  DEV float &dof(int f, int i) const { return lds[ebD + i * DOF_E + f]; }
  DEV float &row(int f, int r) const { return lds[eb + m.e_row() + r * ROW_E + f]; }
  DEV float &con(int f, int c) const { return lds[eb + m.e_con() + c * CON_E + f]; }
  DEV float &geo(int f, int g) const { return lds[eb + m.e_geom() + g * GEO_E + f]; }
  DEV f3 ld_xpos(int b) const { return mk3(body(0, b), body(1, b), body(2, b)); }
  DEV f4 ld_xquat(int b) const { f4 q = {body(3, b), body(4, b), body(5, b), body(6, b)}; return q; }
  DEV f3 ld_com(int r) const { const int o = eb + m.e_com() + 3 * r; return mk3(lds[o], lds[o + 1], lds[o + 2]); }
  DEV sv ld_bsv(int f0, int b) const { sv r = {{body(f0, b), body(f0 + 1, b), body(f0 + 2, b), body(f0 + 3, b), body(f0 + 4, b), body(f0 + 5, b)}}; return r; }
  DEV void st_bsv(int f0, int b, const sv &v) const {
#pragma unroll
    for (int k = 0; k < 6; ++k) body(f0 + k, b) = v.v[k];
  }
  DEV sv ld_dsv(int f0, int i) const { sv r = {{dof(f0, i), dof(f0 + 1, i), dof(f0 + 2, i), dof(f0 + 3, i), dof(f0 + 4, i), dof(f0 + 5, i)}}; return r; }
  DEV void st_dsv(int f0, int i, const sv &v) const {
#pragma unroll
    for (int k = 0; k < 6; ++k) dof(f0 + k, i) = v.v[k];
  }
  // crb (10) is dead once M is built; cvel + cacc (12) reuse its space
  // and cfrc (6) overwrites the body's own cinert once its local force has been formed from it
  enum { F_XPOS = 0, F_XQUAT = 3, F_CINERT = 7, F_CFRC = 7, F_CRB = 17, F_CVEL = 17, F_CACC = 23 };
  // crb*cdof (6) is dead once M is built; the six solver vectors reuse its space
  enum { D_CDOF = 0, D_FCRB = 6, D_MA = 6, D_GRAD = 7, D_MGRAD = 8, D_SEARCH = 9, D_MV = 10, D_FCON = 11,
         D_FSMOOTH = 12, D_ASMOOTH = 13, D_QACC = 14, D_WARM = 15, D_FBIAS = 16 };
  static_assert(D_SEARCH == D_MGRAD + 1, "newton_solve_pair2 writes the search vector into the field behind Mgrad");
  DEV int dofoff(int f) const { return ebD + f; }  // element i of dof field f lives at dofoff(f) + i*DOF_E

  // ------------------------------------------------------------------ A9.1 kinematics
  // Three passes, so that the joint arithmetic (type switch, sin/cos, quaternion products) runs once per step and not once
  // per tree level under a partial exec mask:
  //   1. every body in parallel: its pose and its joints' anchors and axes in the PARENT's frame (a free joint gives the
  //      world pose directly; its parent is the world);
  //   2. level scan from depth 2 down: world pose = parent's world pose o local pose (one quaternion product, one
  //      rotation); depth-1 bodies hang off the world, whose pose is the identity, and are already final;
  //   3. every joint in parallel: anchor and axis taken to the world frame by the parent body's pose.
  static constexpr int KIN_MJ = 2;  // joints per body whose anchors / axes ride in registers through the pose composition
  DEV void kinematics() {
    const int b = gl;
    if (gl == 0) {
      body(0, 0) = 0; body(1, 0) = 0; body(2, 0) = 0;
      body(3, 0) = 1; body(4, 0) = 0; body(5, 0) = 0; body(6, 0) = 0;
    }
    // Bodies with at most one joint each (max_jntnum <= 1: every shipped model): the joint's anchor and axis in the parent's frame
    // stay in the body lane's registers and are taken to the world frame in pass 2, where the parent's world pose is at hand -
    // pass 3 (per joint: body -> parent -> pose, three dependent LDS round trips, and another quaternion-to-matrix) is skipped.
    // (r03: also bodies with two joints - the biped's hips and ankles: a second anchor / axis pair in registers, the joint loop
    // unrolled with a predicate so that both joints' constants and positions are fetched together)
    const bool fuse3 = m.max_jntnum() <= KIN_MJ;
    f3 anc_l[KIN_MJ], ax_l[KIN_MJ];
    int jd0[KIN_MJ];
#pragma unroll
    for (int k = 0; k < KIN_MJ; ++k) { anc_l[k] = mk3(0, 0, 0); ax_l[k] = mk3(0, 0, 1); jd0[k] = 0; }
    if (b > 0 && b < m.nb()) {
      const int br = m.c_body() + b * BODY_S;
      f3 pos = mk3(CF(br + 7), CF(br + 8), CF(br + 9));
      f4 quat = {CF(br + 10), CF(br + 11), CF(br + 12), CF(br + 13)};
      auto one_joint = [&](int k, f3 &anc_o, f3 &ax_o, int &jd_o) {
        const int j = b_jntadr + k, jr = m.c_jnt() + j * JNT_S;
        const int t_ = CI(jr), qa = eb + m.e_qpos() + CI(jr + 1), jd = CI(jr + 2);
        const f3 jpos = mk3(CF(jr + 4), CF(jr + 5), CF(jr + 6)), jax = mk3(CF(jr + 7), CF(jr + 8), CF(jr + 9));
        f3 anchor, axis;
        if (t_ == JFREE) {
          anchor = mk3(lds[qa], lds[qa + 1], lds[qa + 2]);
          axis = mk3(0, 0, 1);
          pos = anchor;
          f4 q = {lds[qa + 3], lds[qa + 4], lds[qa + 5], lds[qa + 6]};
          q = qnormalize(q);
          lds[qa + 3] = q.w; lds[qa + 4] = q.x; lds[qa + 5] = q.y; lds[qa + 6] = q.z;
          quat = q;
        } else {
          anchor = qrot(quat, jpos) + pos;
          axis = qrot(quat, jax);
          if (t_ == JBALL) {
            f4 q = {lds[qa], lds[qa + 1], lds[qa + 2], lds[qa + 3]};
            q = qnormalize(q);
            lds[qa] = q.w; lds[qa + 1] = q.x; lds[qa + 2] = q.y; lds[qa + 3] = q.z;
            quat = qmul(quat, q);
            pos = anchor - qrot1(quat, jpos);
          } else if (t_ == JHINGE) {
            quat = qmul(quat, axisangle(jax, lds[qa] - CF(jr + 10)));
            pos = anchor - qrot1(quat, jpos);
          } else {
            pos = pos + axis * (lds[qa] - CF(jr + 10));
          }
        }
        anc_o = anchor; ax_o = axis; jd_o = jd;
        if (!fuse3 || b_level < 2) {  // final for bodies hanging off the world; local values for pass 3 otherwise
          jnt(0, jd) = anchor.x; jnt(1, jd) = anchor.y; jnt(2, jd) = anchor.z;
          jnt(3, jd) = axis.x; jnt(4, jd) = axis.y; jnt(5, jd) = axis.z;
        }
      };
      if (fuse3) {
#pragma unroll
        for (int k = 0; k < KIN_MJ; ++k) if (k < b_jntnum) one_joint(k, anc_l[k], ax_l[k], jd0[k]);
      } else {
        for (int k = 0; k < b_jntnum; ++k) { f3 a_, x_; int d_; one_joint(k, a_, x_, d_); }
      }
      body(0, b) = pos.x; body(1, b) = pos.y; body(2, b) = pos.z;
      body(3, b) = quat.w; body(4, b) = quat.x; body(5, b) = quat.y; body(6, b) = quat.z;
    }
    WSYNC();
    PROF(28);
    const bool jfuse = fuse3 && b > 0 && b < m.nb() && b_level >= 2 && b_jntnum >= 1;
    if (m.nlevel() > AMX_CHAIN_WALK_MIN_LEVELS && !m.deep_free() && !AMX_LEVEL_SCANS) {  // shallow trees: the level scan is as fast (quadruped, 5 levels: measured)
      // every body walks its own ancestor chain (root first, constants c_anc) and composes the local poses in registers: the
      // same products in the same order as the level-by-level scan, but the loads are independent and nothing waits for a
      // parent's result to come back through LDS.  Chains are padded with the world body, whose pose is the identity (exact
      // no-ops).  All local poses are read before any world pose is stored.
      const bool deep = b > 0 && b < m.nb() && b_level >= 2;
      const int bs = deep ? b : 0, W = m.nlevel() - 2, ar = m.c_anc() + bs * W;
      const f3 lp = ld_xpos(bs);
      const f4 lq = ld_xquat(bs);
      const int a0 = CI(ar);
      f3 ap = ld_xpos(a0);
      f4 aq = ld_xquat(a0);
#pragma unroll
      for (int k = 1; k < W; ++k) {
        const int a = CI(ar + k);
        ap = ap + qrot1(aq, ld_xpos(a));
        aq = qmul(aq, ld_xquat(a));
      }
      const m33 Rp = q2m(aq);  // (ap, aq) is the parent's world pose
      const f3 pos = ap + mulv(Rp, lp);
      const f4 quat = qmul(aq, lq);
      f3 anchor[KIN_MJ], axis[KIN_MJ];
#pragma unroll
      for (int k = 0; k < KIN_MJ; ++k) { anchor[k] = ap + mulv(Rp, anc_l[k]); axis[k] = mulv(Rp, ax_l[k]); }
      WSYNC();
      if (deep) {
        body(0, b) = pos.x; body(1, b) = pos.y; body(2, b) = pos.z;
        body(3, b) = quat.w; body(4, b) = quat.x; body(5, b) = quat.y; body(6, b) = quat.z;
      }
#pragma unroll
      for (int k = 0; k < KIN_MJ; ++k)
        if (jfuse && k < b_jntnum && (k == 0 || m.max_jntnum() > 1)) {
          jnt(0, jd0[k]) = anchor[k].x; jnt(1, jd0[k]) = anchor[k].y; jnt(2, jd0[k]) = anchor[k].z;
          jnt(3, jd0[k]) = axis[k].x; jnt(4, jd0[k]) = axis[k].y; jnt(5, jd0[k]) = axis[k].z;
        }
      WSYNC();
    } else
    for (int L = 2; L < m.nlevel(); ++L) {
      if (b_level == L) {
        const f4 pq = ld_xquat(b_parent);
        const f3 pp = ld_xpos(b_parent);
        const m33 Rp = q2m(pq);
        const f3 pos = pp + mulv(Rp, ld_xpos(b));
        const f4 quat = qmul(pq, ld_xquat(b));
        body(0, b) = pos.x; body(1, b) = pos.y; body(2, b) = pos.z;
        body(3, b) = quat.w; body(4, b) = quat.x; body(5, b) = quat.y; body(6, b) = quat.z;
#pragma unroll
        for (int k = 0; k < KIN_MJ; ++k)
          if (jfuse && k < b_jntnum && (k == 0 || m.max_jntnum() > 1)) {
            const f3 anchor = pp + mulv(Rp, anc_l[k]), axis = mulv(Rp, ax_l[k]);
            jnt(0, jd0[k]) = anchor.x; jnt(1, jd0[k]) = anchor.y; jnt(2, jd0[k]) = anchor.z;
            jnt(3, jd0[k]) = axis.x; jnt(4, jd0[k]) = axis.y; jnt(5, jd0[k]) = axis.z;
          }
      }
      WSYNC();
    }
    PROF(29);
    if (!fuse3 && gl < m.njnt()) {
      const int jr = m.c_jnt() + gl * JNT_S, jd = CI(jr + 2);
      const int p = CI(m.c_body() + CI(jr + 3) * BODY_S);  // parent of the joint's body
      if (p > 0) {  // the world's frame is the identity
        const f4 pq = ld_xquat(p);
        const f3 anchor = ld_xpos(p) + qrot(pq, mk3(jnt(0, jd), jnt(1, jd), jnt(2, jd)));
        const f3 axis = qrot(pq, mk3(jnt(3, jd), jnt(4, jd), jnt(5, jd)));
        jnt(0, jd) = anchor.x; jnt(1, jd) = anchor.y; jnt(2, jd) = anchor.z;
        jnt(3, jd) = axis.x; jnt(4, jd) = axis.y; jnt(5, jd) = axis.z;
      }
    }
    WSYNC();
  }

  // ------------------------------------------------------------------ A9.2 com_pos: tree COM, cinert, cdof (+ collision geom poses)
  DEV void com_pos() {
    const int b = gl, nb = m.nb();
    const bool isb = b > 0 && b < nb;
    const int br = m.c_body() + (isb ? b : 0) * BODY_S;
    f3 xi = mk3(0, 0, 0);
    f4 xq = {1, 0, 0, 0};
    const float ms = isb ? CF(br + 21) : 0.0f;
    if (isb) {
      xq = ld_xquat(b);
      xi = ld_xpos(b) + qrot1(xq, mk3(CF(br + 14), CF(br + 15), CF(br + 16)));
    }
    for (int r = 0; r < m.nroot(); ++r) {
      const bool mine = isb && b_root == r;
#if AMX_TREE_PACK & 1
      const float u_ = gsum_pack4<G>(mine ? xi.x * ms : 0.0f, mine ? xi.y * ms : 0.0f, mine ? ms : 0.0f, mine ? xi.z * ms : 0.0f);  // one packed reduction
      const float sx = quad_value<0>(u_), sy = quad_value<1>(u_), sz = quad_value<2>(u_), sm = quad_value<3>(u_);
#else
      float sx = gsum<G>(mine ? xi.x * ms : 0.0f), sy = gsum<G>(mine ? xi.y * ms : 0.0f), sz = gsum<G>(mine ? xi.z * ms : 0.0f),
            sm = gsum<G>(mine ? ms : 0.0f);
#endif
      const int rb = CI(m.c_rootbody() + r);
      if (b == rb) {  // the root body itself writes the tree's COM
        const int o = eb + m.e_com() + 3 * r;
        if (sm < MINVAL) { lds[o] = xi.x; lds[o + 1] = xi.y; lds[o + 2] = xi.z; }
        else { lds[o] = fdiv(sx, sm); lds[o + 1] = fdiv(sy, sm); lds[o + 2] = fdiv(sz, sm); }  // (three IEEE expansions: 30 instructions)
      }
    }
    WSYNC();
    PROF(31);
    if (b < nb) {
      float r[10];
      if (b == 0) {
#pragma unroll
        for (int k = 0; k < 10; ++k) r[k] = 0;
      } else {
        const f4 iq = {CF(br + 17), CF(br + 18), CF(br + 19), CF(br + 20)};
        const m33 R = q2m(qmul(xq, iq));
        const f3 h = xi - ld_com(b_root);
        const float i0 = CF(br + 22), i1 = CF(br + 23), i2 = CF(br + 24);
#define RIR(i, j) (R.a[3 * i] * i0 * R.a[3 * j] + R.a[3 * i + 1] * i1 * R.a[3 * j + 1] + R.a[3 * i + 2] * i2 * R.a[3 * j + 2])
        r[0] = RIR(0, 0) + ms * (h.y * h.y + h.z * h.z);
        r[1] = RIR(1, 1) + ms * (h.x * h.x + h.z * h.z);
        r[2] = RIR(2, 2) + ms * (h.x * h.x + h.y * h.y);
        r[3] = RIR(0, 1) - ms * h.x * h.y;
        r[4] = RIR(0, 2) - ms * h.x * h.z;
        r[5] = RIR(1, 2) - ms * h.y * h.z;
#undef RIR
        r[6] = ms * h.x; r[7] = ms * h.y; r[8] = ms * h.z; r[9] = ms;
      }
#pragma unroll
      for (int k = 0; k < 10; ++k) body(F_CINERT + k, b) = r[k];
    }
    if (gl < m.nv()) {
      const int i = gl, j = d_jadr;
      const f3 off = ld_com(d_root) - mk3(jnt(0, j), jnt(1, j), jnt(2, j));
      // the four dof kinds without branches: both sources are loaded, the kind selects (a wave holds every kind anyway, so the
      // branches skipped nothing and put a wait in front of each path's loads)
      const m33 R = q2m(ld_xquat(d_body));
      const f3 jax = mk3(jnt(3, j), jnt(4, j), jnt(5, j));
      const f3 e_ = mk3(d_axis == 0 ? 1.0f : 0.0f, d_axis == 1 ? 1.0f : 0.0f, d_axis == 2 ? 1.0f : 0.0f);
      // column d_axis of R by selects: a run-time index would put the matrix in scratch memory
      const f3 rc = d_axis == 0 ? mk3(R.a[0], R.a[3], R.a[6]) : (d_axis == 1 ? mk3(R.a[1], R.a[4], R.a[7]) : mk3(R.a[2], R.a[5], R.a[8]));
      const bool rot = d_kind == 1 || d_kind == 2;  // free/ball rotation, hinge: w = axis, l = w x off
      const f3 wa = d_kind == 1 ? rc : jax;
      const f3 wl = cross(wa, off);
      const f3 w = rot ? wa : mk3(0, 0, 0);
      const f3 l = rot ? wl : (d_kind == 0 ? e_ : jax);
      dof(0, i) = w.x; dof(1, i) = w.y; dof(2, i) = w.z; dof(3, i) = l.x; dof(4, i) = l.y; dof(5, i) = l.z;
    }
    if (gl < m.ngc()) {
      const int g = gl, gr = m.c_geom() + g * GEOM_S, gb = CI(gr);
      const f4 q = ld_xquat(gb);
      const f3 pos = ld_xpos(gb) + qrot1(q, mk3(CF(gr + 2), CF(gr + 3), CF(gr + 4)));
      const f4 gq = {CF(gr + 5), CF(gr + 6), CF(gr + 7), CF(gr + 8)};
      const m33 R = q2m(qmul(q, gq));
      geo(0, g) = pos.x; geo(1, g) = pos.y; geo(2, g) = pos.z;
#pragma unroll
      for (int k = 0; k < 9; ++k) geo(3 + k, g) = R.a[k];
    }
    WSYNC();
  }

  // ------------------------------------------------------------------ A9.3 crb + mass matrix
  DEV void crb_and_M() {
    const int b = gl, nv = m.nv();
    if (m.subtree_contig()) {
      // crb_b = sum of cinert over the subtree of b = the records b .. b_subend-1 (depth-first numbering): every lane adds up
      // its own range, all loads independent - no level-by-level dependence.  A lane past its range reads the world body's
      // record, which is all zeros.  The tree roots (whole trees: the longest ranges by far) wait for their children instead:
      // the common trip count is the largest subtree below a root (quadruped 3, biped 7 instead of 13 and 22).
      const bool isb = b > 0 && b < m.nb(), inner = isb && b_parent > 0;
      float r[10];
#pragma unroll
      for (int k = 0; k < 10; ++k) r[k] = 0.0f;
      auto add_rec = [&](int k) {
        const int d_ = b + k, src = (inner && d_ < b_subend) ? d_ : ((isb && k == 0) ? b : 0);  // a root reads just itself
        float x[10];
#pragma unroll
        for (int q = 0; q < 10; ++q) x[q] = body(F_CINERT + q, src);
        pk_acc<10>(r, x);
      };
      if constexpr (S::is_static) {
        // static trip count, four records' loads in flight at a time (a `#pragma unroll 4` on the run-time form of this loop was not honoured:
        // one LDS round trip per record, seven to nine in a row)
#pragma unroll
        for (int k0 = 0; k0 < S::sub_iter; k0 += 4) {
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) if (k0 + kk < S::sub_iter) add_rec(k0 + kk);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll 4
        for (int k = 0; k < m.sub_iter(); ++k) add_rec(k);
      }
      // a root's composite = itself + the composites of its children = a group sum over the lanes of the root and of the
      // bodies right below it (registers and DPP only; the child-list loop it replaces waited on LDS for every child)
      for (int rt = 0; rt < m.nroot(); ++rt) {
        const bool part = isb && b_root == rt && b_level <= 2;
        float tot[10];
#pragma unroll
        for (int k = 0; k < 10; ++k) tot[k] = part ? r[k] : 0.0f;
#if !(AMX_TREE_PACK & 2)
        for (int k = 0; k < 10; ++k) tot[k] = gsum<G>(tot[k]);
#else
        {  // ten group sums as a packed six and a packed four
          float u_, w_;
          gsum_pack6<G>(tot[0], tot[1], tot[2], tot[3], tot[4], tot[5], u_, w_);
          const float v_ = gsum_pack4<G>(tot[6], tot[7], tot[8], tot[9]);
          tot[0] = quad_value<0>(u_); tot[1] = quad_value<1>(u_); tot[3] = quad_value<2>(u_); tot[2] = quad_value<3>(u_);
          tot[4] = quad_value<0>(w_); tot[5] = quad_value<1>(w_);
          tot[6] = quad_value<0>(v_); tot[7] = quad_value<1>(v_); tot[9] = quad_value<2>(v_); tot[8] = quad_value<3>(v_);
        }
#endif
        if (isb && !inner && b_root == rt) {
#pragma unroll
          for (int k = 0; k < 10; ++k) r[k] = tot[k];
        }
      }
      if (isb) {
#pragma unroll
        for (int k = 0; k < 10; ++k) body(F_CRB + k, b) = r[k];
      }
      WSYNC();
    } else
    for (int L = m.nlevel() - 1; L >= 1; --L) {
      if (b_level == L) {
        float r[10];
#pragma unroll
        for (int k = 0; k < 10; ++k) r[k] = body(F_CINERT + k, b);
        for (int c = 0; c < b_childnum; ++c) {
          const int cb = CI(m.c_child() + b_childadr + c);
#pragma unroll
          for (int k = 0; k < 10; ++k) r[k] += body(F_CRB + k, cb);
        }
#pragma unroll
        for (int k = 0; k < 10; ++k) body(F_CRB + k, b) = r[k];
      }
      WSYNC();
    }
    PROF(20);
    if (gl < nv) {
      float I[10];
#pragma unroll
      for (int k = 0; k < 10; ++k) I[k] = body(F_CRB + k, d_body);
      st_dsv(D_FCRB, gl, inert_mul(I, ld_dsv(D_CDOF, gl)));
    }
    WSYNC();
    const int Mo = eb + m.e_M();
    struct MEnt { int at, at2; float v; };
    item_loop2<G>(m.nM(), gl,
      [&](int e) {
        const unsigned pk = (unsigned)CI(m.c_M() + e);  // offsets of M[i][j] and M[j][i] resolved on the host (rows may be chunk-rotated)
        const int i = (pk >> 22) & 31, j = pk >> 27;
        const sv f = ld_dsv(D_FCRB, i), c = ld_dsv(D_CDOF, j);
        const float arm = CF(m.c_dof() + i * DOF_S + 4);
        float v = sv_dot(c, f);
        v += i == j ? arm : 0.0f;  // a select: as `if`, the armature load sank into a branch of its own with its own wait
        return MEnt{Mo + (int)(pk & 2047u), Mo + (int)((pk >> 11) & 2047u), v};  // j is an ancestor of i: both triangles are stored
      },
      [&](int, const MEnt &r) { lds[r.at] = r.v; lds[r.at2] = r.v; });
    WSYNC();
  }

  // ------------------------------------------------------------------ A9.8 com_vel + A9.10 rne (one down-scan, one up-scan)
  DEV void velocity_and_bias() {
    const int b = gl;
    const int qv = eb + m.e_qvel();
    if (gl == 0) {
      sv z = {{0, 0, 0, 0, 0, 0}};
      st_bsv(F_CVEL, 0, z);
      st_bsv(F_CFRC, 0, z);
      if (!(m.disableflags() & DIS_GRAVITY)) { z.v[3] = -m.gx(); z.v[4] = -m.gy(); z.v[5] = -m.gz(); }
      st_bsv(F_CACC, 0, z);
    }
    WSYNC();
    // The joint arithmetic runs once per step, not once per tree level: with cv_p, ca_p the parent's velocity and
    // acceleration and P_k the body's own partial velocity before joint k,
    //   cvel_b = cv_p + sum_i cdof_i qvel_i                                                =: cv_p + vJ
    //   cacc_b = ca_p + sum_i cross(cv_p + P_k(i), cdof_i) qvel_i = ca_p + cross(cv_p, w) + loc
    // (w = sum over the dofs that have a cdof_dot - every dof but the translations of a free joint; loc = the same sum
    // with P_k alone; the motion cross product is bilinear).  vJ, w, loc need no other body and stay in registers; the
    // level scan is then twelve loads, one cross product and twelve stores.
    sv vJ = {{0, 0, 0, 0, 0, 0}}, w = {{0, 0, 0, 0, 0, 0}}, loc = {{0, 0, 0, 0, 0, 0}};
    sv cv_own = {{0, 0, 0, 0, 0, 0}}, ca_own = {{0, 0, 0, 0, 0, 0}};
    bool own_in_regs = false;
    {
      // Joint sums by dof group (c_bgrp), without branches: every group fetches three dofs (a one-dof group fetches its dof three
      // times and weights two of them by zero; lanes without a body and groups a body does not have weigh everything by zero), so
      // all the loads of all the groups are in flight together.  MJX's cdof_dot of a dof is (velocity before the group) x cdof,
      // summed with the dof's qvel: by bilinearity one cross product with the group's sum.  The free joint's translations add
      // to the velocity only (their cdof_dot is zero and they are not part of w).
      const bool isb = b > 0 && b < m.nb();
      auto one_group = [&](int g) {
        const int gi = isb ? CI(m.c_bgrp() + b * m.max_bgrp() + g) : 0;
        const int st = gi & 255, n = (gi >> 8) & 3;
        const float cf = (gi >> 12) & 1 ? 1.0f : 0.0f;
        const sv c0 = ld_dsv(D_CDOF, st), c1 = ld_dsv(D_CDOF, st + (n > 1 ? 1 : 0)), c2 = ld_dsv(D_CDOF, st + (n > 2 ? 2 : 0));
        const float q0_ = lds[qv + st], q1_ = lds[qv + st + (n > 1 ? 1 : 0)], q2_ = lds[qv + st + (n > 2 ? 2 : 0)];
        const float q0 = n > 0 ? q0_ : 0.0f, q1 = n > 1 ? q1_ : 0.0f, q2 = n > 2 ? q2_ : 0.0f;
        sv add = {{0, 0, 0, 0, 0, 0}};
#pragma unroll
        for (int e = 0; e < 6; ++e) add.v[e] = __builtin_fmaf(c2.v[e], q2, __builtin_fmaf(c1.v[e], q1, c0.v[e] * q0));
        const sv cd = motion_cross(vJ, add);
#pragma unroll
        for (int e = 0; e < 6; ++e) { loc.v[e] = __builtin_fmaf(cd.v[e], cf, loc.v[e]); w.v[e] = __builtin_fmaf(add.v[e], cf, w.v[e]); }
        sv_acc(vJ, add);
      };
      if constexpr (S::is_static) {
#pragma unroll
        for (int g = 0; g < 4; ++g) if (g < S::max_bgrp) one_group(g);
      } else {
        for (int g = 0; g < m.max_bgrp(); ++g) one_group(g);
      }
    }
    PROF(30);
    if (!m.deep_free() && m.nlevel() > AMX_CHAIN_WALK_MIN_LEVELS && !AMX_LEVEL_SCANS) {
      // The same chain walk as in kinematics(): every body stores its own joint sums (vJ, loc), then accumulates them along its
      // ancestor chain in registers - cvel and cacc of every ancestor are recomputed in the lane, in the order the level scan
      // produces them.  w of an ancestor is its vJ (they differ only for free joints, which sit on the first level, where the
      // velocity arriving from the world is zero and the cross product vanishes either way).
      const bool isb = b > 0 && b < m.nb();
      if (isb) { st_bsv(F_CVEL, b, vJ); st_bsv(F_CACC, b, loc); }
      if (gl == 0) { sv z = {{0, 0, 0, 0, 0, 0}}; st_bsv(F_CACC, 0, z); }
      WSYNC();
      const int W = m.nlevel() > 2 ? m.nlevel() - 2 : 0, ar = m.c_anc() + (isb ? b : 0) * (W > 0 ? W : 1);
      sv cv = {{0, 0, 0, 0, 0, 0}}, ca = {{0, 0, 0, 0, 0, 0}};
      if (!(m.disableflags() & DIS_GRAVITY)) { ca.v[3] = -m.gx(); ca.v[4] = -m.gy(); ca.v[5] = -m.gz(); }
#pragma unroll
      for (int k = 0; k < W; ++k) {
        const int a = CI(ar + k);
        const sv va = ld_bsv(F_CVEL, a), la = ld_bsv(F_CACC, a);
        const sv x = motion_cross(cv, va);
#pragma unroll
        for (int e = 0; e < 1; ++e) { sv_acc(ca, x); sv_acc(ca, la); sv_acc(cv, va); }
      }
      const sv x = motion_cross(cv, w);
#pragma unroll
      for (int e = 0; e < 1; ++e) { sv_acc(ca, x); sv_acc(ca, loc); sv_acc(cv, vJ); }
      // the body's own velocity and acceleration are complete in this lane and nobody else reads them (the level scan below hands
      // them from parent to child through LDS; here every body recomputed its ancestors'): the local force is formed straight from
      // the registers - twelve stores, twelve loads and a round trip less
      // (the local force goes to F_CFRC, which overlays the body's own inertia and not the joint sums other lanes still read)
      cv_own = cv; ca_own = ca; own_in_regs = true;
    } else
    for (int L = 1; L < m.nlevel(); ++L) {
      if (b_level == L) {
        const sv cvp = ld_bsv(F_CVEL, b_parent), cap = ld_bsv(F_CACC, b_parent);
        const sv x = motion_cross(cvp, w);
        sv cv, ca;
#pragma unroll
        for (int e = 0; e < 6; ++e) { cv.v[e] = cvp.v[e] + vJ.v[e]; ca.v[e] = cap.v[e] + x.v[e] + loc.v[e]; }
        st_bsv(F_CVEL, b, cv);
        st_bsv(F_CACC, b, ca);
      }
      WSYNC();
    }
    PROF(23);
    if (b > 0 && b < m.nb()) {  // local force of every body in parallel: I cacc + cvel x* (I cvel)
      const sv cv = own_in_regs ? cv_own : ld_bsv(F_CVEL, b), ca = own_in_regs ? ca_own : ld_bsv(F_CACC, b);
      float I[10];
#pragma unroll
      for (int k = 0; k < 10; ++k) I[k] = body(F_CINERT + k, b);
      sv f = inert_mul(I, ca);
      const sv g = motion_cross_force(cv, inert_mul(I, cv));
#pragma unroll
      for (int e = 0; e < 1; ++e) sv_acc(f, g);
      st_bsv(F_CFRC, b, f);
    }
    WSYNC();
    if (m.subtree_contig()) {
      // total force below each body = sum of the local forces over its record range (see crb_and_M); read everything, then
      // store; the roots then add their children's totals
      const bool isb = b > 0 && b < m.nb(), inner = isb && b_parent > 0;
      sv f = {{0, 0, 0, 0, 0, 0}};
      auto add_frc = [&](int k) {
        const int d_ = b + k;
        const sv g = ld_bsv(F_CFRC, (inner && d_ < b_subend) ? d_ : ((isb && k == 0) ? b : 0));
#pragma unroll
        for (int e = 0; e < 1; ++e) sv_acc(f, g);
      };
      if constexpr (S::is_static) {  // (as in crb_and_M: four records in flight at a time)
#pragma unroll
        for (int k0 = 0; k0 < S::sub_iter; k0 += 4) {
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) if (k0 + kk < S::sub_iter) add_frc(k0 + kk);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll 4
        for (int k = 0; k < m.sub_iter(); ++k) add_frc(k);
      }
      for (int rt = 0; rt < m.nroot(); ++rt) {  // roots: group sum over themselves and the bodies right below them
        const bool part = isb && b_root == rt && b_level <= 2;
        sv tot;
#pragma unroll
        for (int e = 0; e < 6; ++e) tot.v[e] = part ? f.v[e] : 0.0f;
#if !(AMX_TREE_PACK & 4)
        for (int e = 0; e < 6; ++e) tot.v[e] = gsum<G>(tot.v[e]);
#else
        {
          float u_, w_;
          gsum_pack6<G>(tot.v[0], tot.v[1], tot.v[2], tot.v[3], tot.v[4], tot.v[5], u_, w_);
          tot.v[0] = quad_value<0>(u_); tot.v[1] = quad_value<1>(u_); tot.v[3] = quad_value<2>(u_); tot.v[2] = quad_value<3>(u_);
          tot.v[4] = quad_value<0>(w_); tot.v[5] = quad_value<1>(w_);
        }
#endif
        if (isb && !inner && b_root == rt) f = tot;
      }
      WSYNC();
      if (isb) st_bsv(F_CFRC, b, f);
      WSYNC();
    } else
    for (int L = m.nlevel() - 2; L >= 1; --L) {  // up-scan: gather children
      if (b_level == L && b_childnum > 0) {
        sv f = ld_bsv(F_CFRC, b);
        for (int c = 0; c < b_childnum; ++c) {
          const sv g = ld_bsv(F_CFRC, CI(m.c_child() + b_childadr + c));
#pragma unroll
          for (int e = 0; e < 1; ++e) sv_acc(f, g);
        }
        st_bsv(F_CFRC, b, f);
      }
      WSYNC();
    }
  }

  // ------------------------------------------------------------------ A9.9 passive, A9.7/A9.11 actuation, A9.12 qfrc_smooth
  DEV void smooth_forces() {
    const int qp = eb + m.e_qpos(), qv = eb + m.e_qvel();
    const bool act_on = m.nu() > 0 && !(m.disableflags() & DIS_ACTUATION);
    if (act_on && gl < m.nu()) {
      // every parameter loaded up front, the option flags are selects (as `if`s each one was a branch around its own loads)
      const int ar = m.c_act() + gl * ACT_S, fl = CI(ar + 2);
      const float gear = CF(ar + 3), gain = CF(ar + 4), b0 = CF(ar + 5), b1 = CF(ar + 6), b2 = CF(ar + 7);
      const float clo = CF(ar + 8), chi = CF(ar + 9), flo = CF(ar + 10), fhi = CF(ar + 11);
      const float qj = lds[qp + CI(ar)], vj = lds[qv + CI(ar + 1)];
      float c = lds[eb + m.e_ctrl() + gl];
      const float cc = fminf(fmaxf(c, clo), chi);
      c = (!(m.disableflags() & DIS_CLAMPCTRL) && (fl & 1)) ? cc : c;
      float force = gain * c;
      const float fb = force + (b0 + b1 * (qj * gear) + b2 * (vj * gear));
      force = (fl & 4) ? fb : force;
      const float fc = fminf(fmaxf(force, flo), fhi);
      force = (fl & 2) ? fc : force;
      lds[eb + m.e_actf() + gl] = gear * force;
    }
    WSYNC();
    if (gl < m.nv()) {
      const int i = gl, dr = m.c_dof() + i * DOF_S;
      const sv c = ld_dsv(D_CDOF, i), f = ld_bsv(F_CFRC, d_body);
      const float bias = sv_dot(c, f);
      float pas = 0;
      if (!(m.disableflags() & DIS_PASSIVE)) {
        const int jr = m.c_jnt() + d_jnt * JNT_S;
        const float k = CF(jr + 11), qj = lds[qp + CI(jr + 1)], qref = CF(jr + 12);
        const float ks = k * (qj - qref);
        pas -= (k != 0.0f && d_kind >= 2) ? ks : 0.0f;
        pas -= CF(dr + 5) * lds[qv + i];
      }
      float act = 0;
      if (act_on) {
        // the first actuator of the dof without a branch (a dof rarely has a second one)
        const int a0 = CI(dr + 6), an = CI(dr + 7);
        const float f0 = lds[eb + m.e_actf() + CI(m.c_dofact() + (an > 0 ? a0 : 0))];
        act = an > 0 ? f0 : 0.0f;
        for (int a = 1; a < an; ++a) act += lds[eb + m.e_actf() + CI(m.c_dofact() + a0 + a)];
      }
      dof(D_FBIAS, i) = bias;
      dof(D_FSMOOTH, i) = pas - bias + act;
      r_fs = pas - bias + act;
    }
    WSYNC();
  }

  // ------------------------------------------------------------------ A9.5 collision (one gl per static geom pair)
  DEV void plane_sphere(f3 n, f3 ppos, f3 spos, float r, float *dist, f3 *pos) const {
    const float d = dot(spos - ppos, n) - r;
    *dist = d;
    *pos = spos - n * (r + 0.5f * d);
  }
  DEV void st_contact(int c, float d, f3 cp) const { con(0, c) = d; con(1, c) = cp.x; con(2, c) = cp.y; con(3, c) = cp.z; }
  DEV void collision() {
    const int np = m.npair();
    // plane - convex pairs with at most 32 hull vertices: one lane per hull vertex (plane_convex_parallel), on groups of 32
    // lanes (one pair per pass) and 64 lanes (two).  The planner found the run of such pairs (pcx_first, npcx).
    const int cx_first = G >= 32 ? m.pcx_first() : 0, cx_num = G >= 32 ? m.npcx() : 0;
    const bool contacts_on = !(m.disableflags() & DIS_CONTACT);
    if (G >= 32 && cx_num > 0 && contacts_on)  // uniform: a property of the model
    {
      if constexpr (S::is_static) plane_convex_parallel_body<G, (S::pcx_unit == 16 && G >= 32) ? 16 : 32>(cx_first, cx_num, m.c_pair(), m.c_vert(), eb + m.e_geom(), eb + m.e_con(), eb + m.e_frame(), gl, gb4);
      else if (m.pcx_unit() == 16) plane_convex_parallel<G, G >= 32 ? 16 : 32>(cx_first, cx_num, m.c_pair(), m.c_vert(), eb + m.e_geom(), eb + m.e_con(), eb + m.e_frame(), gl, gb4);
      else plane_convex_parallel<G, 32>(cx_first, cx_num, m.c_pair(), m.c_vert(), eb + m.e_geom(), eb + m.e_con(), eb + m.e_frame(), gl, gb4);
    }
    if (m.ncx() > 0 && contacts_on)  // uniform: a property of the model
      convex_convex_pairs<G>(m.cx_first(), m.ncx(), m.cx_maxv(), m.c_pair(), eb + m.e_geom(), eb + m.e_con(), eb + m.e_frame(), eb + m.e_cx(), gl, gb4);
    if (m.nbl() > 0 && gl >= np - m.nbl() && gl < np) {
      // A ball joint's limit (MJX constraint._instantiate_limit_ball; oracle: make_constraint): the joint
      // quaternion (normalised by kinematics) as axis * angle, the angle wrapped to (-pi, pi] and made positive by turning the
      // axis; distance = max(range) - angle.  The frame slot carries the axis (the row's Jacobian is -axis on the three dofs).
      const int pr = gl, rec = m.c_pair() + pr * PAIR_S, qa = eb + m.e_qpos() + CI(rec), c0 = CI(rec + 3);
      const float qw = lds[qa], qx = lds[qa + 1], qy = lds[qa + 2], qz = lds[qa + 3];
      const float s_ = sqrtf(qx * qx + qy * qy + qz * qz);
      const float inv = 1.0f / (s_ + (s_ == 0.0f ? 1e-6f : 0.0f));  // math.normalize_with_norm
      float ang = 2.0f * atan2f(s_, qw);
      ang = ang > 3.14159265358979f ? ang - 6.28318530717959f : ang;
      const f3 av = mk3(qx * inv * ang, qy * inv * ang, qz * inv * ang);
      const float n_ = sqrtf(dot(av, av));
      const float inv2 = 1.0f / (n_ + (n_ == 0.0f ? 1e-6f : 0.0f));
      con(0, c0) = CF(rec + 23) - n_;
      const int fo = eb + m.e_frame() + pr * 9;
      lds[fo] = av.x * inv2; lds[fo + 1] = av.y * inv2; lds[fo + 2] = av.z * inv2;
    }
    if (m.npair() - m.nbl() > m.npc() && !(m.disableflags() & DIS_EQUALITY) && gl >= m.npc() && gl < np - m.nbl()) {
      // Row groups of the connect / weld equalities (oracle: make_constraint, connect and weld): the residual of
      // the group and what its Jacobian entries need, while the body poses are still at hand.  Function 8: a point fixed in
      // body1 against a point fixed in body2; function 9 (second group of a weld): the relative orientation against relpose.
      // All rows of one equality share the impedance of the norm of its whole residual: con(0) = that norm, con(1..3) = the
      // group's three residual components; the frame slot carries the two world points (8) or the two quaternions (9).
      const int pr = gl, rec = m.c_pair() + pr * PAIR_S;
      const int b1 = CI(rec), b2 = CI(rec + 1), fn = CI(rec + 2), c0 = CI(rec + 3), weld = CI(rec + 5);
      const float ts = CF(rec + 11);
      const f4 q1 = ld_xquat(b1), q2 = ld_xquat(b2), rq = {CF(rec + 29), CF(rec + 30), CF(rec + 31), CF(rec + 32)};
      const f3 p1 = ld_xpos(b1) + qrot(q1, mk3(CF(rec + 23), CF(rec + 24), CF(rec + 25)));
      const f3 p2 = ld_xpos(b2) + qrot(q2, mk3(CF(rec + 26), CF(rec + 27), CF(rec + 28)));
      const f3 cpos = p1 - p2;
      const f4 q1r = qmul(q1, rq), q2c = {q2.w, -q2.x, -q2.y, -q2.z};
      const f4 qe = qmul(q2c, q1r);
      const f3 crot = mk3(qe.x * ts, qe.y * ts, qe.z * ts);
      const float nrm = sqrtf(dot(cpos, cpos) + (weld ? dot(crot, crot) : 0.0f));
      const f3 res = fn == 8 ? cpos : crot;
      con(0, c0) = nrm; con(1, c0) = res.x; con(2, c0) = res.y; con(3, c0) = res.z;
      const int fo = eb + m.e_frame() + pr * 9;
      if (fn == 8) { lds[fo] = p1.x; lds[fo + 1] = p1.y; lds[fo + 2] = p1.z; lds[fo + 3] = p2.x; lds[fo + 4] = p2.y; lds[fo + 5] = p2.z; }
      else { lds[fo] = q1r.w; lds[fo + 1] = q1r.x; lds[fo + 2] = q1r.y; lds[fo + 3] = q1r.z; lds[fo + 4] = q2c.w; lds[fo + 5] = q2c.x; lds[fo + 6] = q2c.y; lds[fo + 7] = q2c.z; }
    }
    if (gl < m.npc() && !(m.disableflags() & DIS_CONTACT) && !(gl >= cx_first && gl < cx_first + cx_num) && !(gl >= m.cx_first() && gl < m.cx_first() + m.ncx())) {
      const int pr = gl, rec = m.c_pair() + pr * PAIR_S;
      const int g1 = CI(rec), g2 = CI(rec + 1), fn = CI(rec + 2), c0 = CI(rec + 3);
      const f3 pos1 = mk3(geo(0, g1), geo(1, g1), geo(2, g1)), pos2 = mk3(geo(0, g2), geo(1, g2), geo(2, g2));
      m33 R1, R2;
#pragma unroll
      for (int k = 0; k < 9; ++k) { R1.a[k] = geo(3 + k, g1); R2.a[k] = geo(3 + k, g2); }
      const int s1 = m.c_geom() + g1 * GEOM_S + 9, s2 = m.c_geom() + g2 * GEOM_S + 9;
      float fr[9];
      if (fn == 0) {
        const f3 n = mk3(R1.a[2], R1.a[5], R1.a[8]);
        f3 cp;
        float d;
        plane_sphere(n, pos1, pos2, CF(s2), &d, &cp);
        st_contact(c0, d, cp);
        make_frame(n, fr);
      } else if (fn == 1) {
        const f3 n = mk3(R1.a[2], R1.a[5], R1.a[8]), axis = mk3(R2.a[2], R2.a[5], R2.a[8]);
        float bn;
        f3 b = normalize3(axis - n * dot(n, axis), &bn);
        if (bn < 0.5f) b = (n.y > -0.5f && n.y < 0.5f) ? mk3(0, 1, 0) : mk3(0, 0, 1);
        const f3 c = cross(n, b);
        fr[0] = n.x; fr[1] = n.y; fr[2] = n.z; fr[3] = b.x; fr[4] = b.y; fr[5] = b.z; fr[6] = c.x; fr[7] = c.y; fr[8] = c.z;
        const float rad = CF(s2), half = CF(s2 + 1);
        for (int k = 0; k < 2; ++k) {
          f3 cp;
          float d;
          plane_sphere(n, pos1, pos2 + axis * (k == 0 ? half : -half), rad, &d, &cp);
          st_contact(c0 + k, d, cp);
        }
      } else if (fn == 2) {
        const int nvt = CI(rec + 6), va = m.c_vert() + 3 * CI(rec + 5);
        const f3 nw = mk3(R1.a[2], R1.a[5], R1.a[8]);
        const f3 pl = mulTv(R2, pos1 - pos2), n = mulTv(R2, nw);
        float smax = -1e30f;
#define VERT(v) mk3(CF(va + 3 * (v)), CF(va + 3 * (v) + 1), CF(va + 3 * (v) + 2))
        for (int v = 0; v < nvt; ++v) smax = fmaxf(smax, dot(pl - VERT(v), n));
        const float thr = fmaxf(0.0f, smax - 1e-3f);
        int idx[4];
        float best;
#define MASKV(vt) (dot(pl - (vt), n) > thr ? 0.0f : -1e6f)
        idx[0] = 0; best = -1e30f;
        for (int v = 0; v < nvt; ++v) { const float sc = MASKV(VERT(v)); if (sc > best) { best = sc; idx[0] = v; } }
        const f3 a = VERT(idx[0]);
        idx[1] = 0; best = -1e30f;
        for (int v = 0; v < nvt; ++v) { const f3 vt = VERT(v); const f3 d = a - vt; const float sc = dot(d, d) + MASKV(vt); if (sc > best) { best = sc; idx[1] = v; } }
        const f3 bb = VERT(idx[1]);
        const f3 ab = cross(n, a - bb);
        idx[2] = 0; best = -1e30f;
        for (int v = 0; v < nvt; ++v) { const f3 vt = VERT(v); const float sc = fabsf(dot(a - vt, ab)) + MASKV(vt); if (sc > best) { best = sc; idx[2] = v; } }
        const f3 cc = VERT(idx[2]);
        const f3 ac = cross(n, a - cc), bc = cross(n, bb - cc);
        idx[3] = 0; best = -1e30f;
        for (int v = 0; v < nvt; ++v) { const f3 vt = VERT(v); const float sc = fabsf(dot(bb - vt, bc)) + MASKV(vt); if (sc > best) { best = sc; idx[3] = v; } }
        for (int v = 0; v < nvt; ++v) { const f3 vt = VERT(v); const float sc = fabsf(dot(a - vt, ac)) + MASKV(vt); if (sc > best) { best = sc; idx[3] = v; } }
        make_frame(nw, fr);
        for (int k = 0; k < 4; ++k) {
          bool unique = true;
          for (int j = 0; j < k; ++j) unique = unique && (idx[j] != idx[k]);
          const f3 vt = VERT(idx[k]);
          const float d = unique ? -dot(pl - vt, n) : 1.0f;
          st_contact(c0 + k, d, pos2 + mulv(R2, vt) - nw * (0.5f * d));
        }
#undef VERT
#undef MASKV
      } else if (fn == 7) {
        // sphere - convex hull (oracle: sphere_convex): in the hull's frame, the face of largest support (least penetrated, or
        // most separating), then the closest point of that polygon to the centre (projection, or the nearest violated edge)
        const int hB = CI(rec + 26), nfB = CI(hB + 2), vB = CI(hB + 1), nB_ = CI(hB + 3), pB = CI(hB + 4);
#define HV(base, i) mk3(CF((base) + 3 * (i)), CF((base) + 3 * (i) + 1), CF((base) + 3 * (i) + 2))
        const float r = CF(s1);
        const f3 c = mulTv(R2, pos1 - pos2);
        int bf = 0;
        float bs = -1e30f;
        for (int f = 0; f < nfB; ++f) {
          const f3 n = HV(nB_, f), v0 = HV(vB, CI(pB + 9 * f + 1));
          float sup = dot(c - n * r - v0, n);
          if (sup > bs) { bs = sup; bf = f; }
        }
        const f3 n = HV(nB_, bf);
        const int poly = pB + 9 * bf, nvf = CI(poly);
        const f3 v0 = HV(vB, CI(poly + 1));
        f3 pt = c - n * dot(c - v0, n);
        bool inside = true;
        int be = -1;
        float bd = 1e12f;
        for (int k = 0; k < nvf; ++k) {
          const f3 p0 = HV(vB, CI(poly + 1 + (k == 0 ? nvf - 1 : k - 1))), p1 = HV(vB, CI(poly + 1 + k));
          const f3 en = cross(p1 - p0, n);
          const float ed = dot(pt - p0, en);
          if (ed > 0) inside = false;
          const bool degenerate = en.x == 0 && en.y == 0 && en.z == 0;
          if (!degenerate && !(ed < 0) && ed < bd) { bd = ed; be = k; }
        }
        if (!inside && be >= 0) {
          const f3 p0 = HV(vB, CI(poly + 1 + (be == 0 ? nvf - 1 : be - 1))), p1 = HV(vB, CI(poly + 1 + be));
          const f3 e = p1 - p0;
          const float t = fminf(fmaxf(dot(pt - p0, e) / fmaxf(dot(e, e), MINVAL), 0.0f), 1.0f);
          pt = p0 + e * t;
        }
#undef HV
        float dn;
        f3 pn = normalize3(pt - c, &dn);
        if (dn == 0) pn = n * -1.0f;
        const bool below = dot(pn, n) > 0;  // the centre is under the face: deep penetration
        const f3 nl = below ? pn * -1.0f : pn;
        const float d = (below ? -dn : dn) - r;
        const f3 posl = (pt + c + nl * r) * 0.5f;
        st_contact(c0, d, pos2 + mulv(R2, posl));
        make_frame(mulv(R2, nl), fr);
      } else if (fn == 11) {
        // capsule - convex hull (oracle: capsule_convex; r04): in the hull's frame, the face of largest support of the capsule's
        // axis; the axis clipped to that face's side planes -> two contacts, or - an axis that misses the polygon - the closest
        // polygon edge -> one contact, the second slot inactive
        const int hB = CI(rec + 26), nfB = CI(hB + 2), vB = CI(hB + 1), nB_ = CI(hB + 3), pB = CI(hB + 4);
#define HV(base, i) mk3(CF((base) + 3 * (i)), CF((base) + 3 * (i) + 1), CF((base) + 3 * (i) + 2))
        const float r = CF(s1), half = CF(s1 + 1);
        const f3 c = mulTv(R2, pos1 - pos2), ax = mulTv(R2, mk3(R1.a[2], R1.a[5], R1.a[8]));
        const f3 a = c + ax * half, b = c - ax * half;
        int bf = 0;
        float bs = -1e30f;
        for (int f = 0; f < nfB; ++f) {
          const f3 n = HV(nB_, f), v0 = HV(vB, CI(pB + 9 * f + 1));
          const float sup = fminf(dot(a - v0, n), dot(b - v0, n)) - r;
          if (sup > bs) { bs = sup; bf = f; }
        }
        const f3 n = HV(nB_, bf);
        const int poly = pB + 9 * bf, nvf = CI(poly);
        const f3 v0 = HV(vB, CI(poly + 1));
        float t0 = 0.0f, t1 = 1.0f;
        bool empty = false;
        for (int k = 0; k < nvf; ++k) {
          const f3 p0 = HV(vB, CI(poly + 1 + (k == 0 ? nvf - 1 : k - 1))), p1 = HV(vB, CI(poly + 1 + k));
          const f3 en = cross(p1 - p0, n);
          const float da = dot(a - p0, en), db = dot(b - p0, en);
          if (da > 0 && db > 0) { empty = true; continue; }
          if (da > 0) t0 = fmaxf(t0, da / (da - db));
          else if (db > 0) t1 = fminf(t1, da / (da - db));
        }
        if (t0 > t1) empty = true;
        f3 nl, pl0, pl1;
        float dist0, dist1;
        if (!empty) {
          const f3 q0 = a + (b - a) * t0, q1 = a + (b - a) * t1;
          dist0 = dot(q0 - v0, n) - r; dist1 = dot(q1 - v0, n) - r;
          pl0 = q0 - n * (r + 0.5f * dist0); pl1 = q1 - n * (r + 0.5f * dist1);
          nl = n * -1.0f;
        } else {
          float best = 1e30f;
          f3 pa = c, pe = v0;
          for (int k = 0; k < nvf; ++k) {
            const f3 p0 = HV(vB, CI(poly + 1 + (k == 0 ? nvf - 1 : k - 1))), p1 = HV(vB, CI(poly + 1 + k));
            // closest points of the segments [a, b] and [p0, p1] (clamped; the oracle's seg_seg_closest)
            const f3 d1 = b - a, d2 = p1 - p0, rr = a - p0;
            const float aa = dot(d1, d1), ee = dot(d2, d2), ff = dot(d2, rr);
            float sg, tg;
            if (aa <= MINVAL && ee <= MINVAL) { sg = 0; tg = 0; }
            else if (aa <= MINVAL) { sg = 0; tg = fminf(fmaxf(ff / ee, 0.0f), 1.0f); }
            else {
              const float cc = dot(d1, rr);
              if (ee <= MINVAL) { tg = 0; sg = fminf(fmaxf(-cc / aa, 0.0f), 1.0f); }
              else {
                const float bb = dot(d1, d2), denom = aa * ee - bb * bb;
                sg = denom > MINVAL ? fminf(fmaxf((bb * ff - cc * ee) / denom, 0.0f), 1.0f) : 0.0f;
                tg = (bb * sg + ff) / ee;
                if (tg < 0) { tg = 0; sg = fminf(fmaxf(-cc / aa, 0.0f), 1.0f); }
                else if (tg > 1) { tg = 1; sg = fminf(fmaxf((bb - cc) / aa, 0.0f), 1.0f); }
              }
            }
            const f3 c1 = a + d1 * sg, c2 = p0 + d2 * tg, dd = c2 - c1;
            const float d2n = dot(dd, dd);
            if (d2n < best) { best = d2n; pa = c1; pe = c2; }
          }
          float dn;
          nl = normalize3(pe - pa, &dn);
          if (dn == 0) nl = n * -1.0f;
          dist0 = dn - r; dist1 = 1.0f;
          pl0 = (pa + nl * r + pe) * 0.5f; pl1 = pl0;
        }
#undef HV
        st_contact(c0, dist0, pos2 + mulv(R2, pl0));
        st_contact(c0 + 1, dist1, pos2 + mulv(R2, pl1));
        make_frame(mulv(R2, nl), fr);
      } else {
        f3 p1 = pos1, p2 = pos2;
        const float r1 = CF(s1), r2 = CF(s2);
        if (fn == 4) {
          const f3 ax = mk3(R2.a[2], R2.a[5], R2.a[8]);
          const float h2 = CF(s2 + 1);
          p2 = pos2 + ax * fminf(fmaxf(dot(pos1 - pos2, ax), -h2), h2);
        } else if (fn == 5) {
          const f3 a1 = mk3(R1.a[2], R1.a[5], R1.a[8]), a2 = mk3(R2.a[2], R2.a[5], R2.a[8]);
          const float h1 = CF(s1 + 1), h2 = CF(s2 + 1);
          const f3 dv = pos1 - pos2;
          const float a12 = dot(a1, a2), d1 = dot(dv, a1), d2 = dot(dv, a2), det = 1 - a12 * a12;
          float t1 = fabsf(det) < 1e-6f ? 0.0f : fminf(fmaxf((a12 * d2 - d1) / det, -h1), h1);
          const float t2 = fminf(fmaxf(d2 + t1 * a12, -h2), h2);
          t1 = fminf(fmaxf(t2 * a12 - d1, -h1), h1);
          p1 = pos1 + a1 * t1; p2 = pos2 + a2 * t2;
        }
        float len;
        f3 n = normalize3(p2 - p1, &len);
        if (len == 0) n = mk3(1, 0, 0);
        const float d = len - (r1 + r2);
        st_contact(c0, d, p1 + n * (r1 + 0.5f * d));
        make_frame(n, fr);
      }
#pragma unroll
      for (int k = 0; k < 9; ++k) lds[eb + m.e_frame() + pr * 9 + k] = fr[k];
    }
    WSYNC();
  }

  // ------------------------------------------------------------------ A9.6 make_constraint
  // Row table (static): [0,neq) joint equalities, [neq,neq+nlim) joint limits, then 4 pyramid rows per
  // contact slot.  A row that is inactive this step gets D = 0 (the reference zeroes it instead).
  DEV void make_constraint() {
    const int Jo = eb + m.e_J(), ldc = m.ldc();
    const int qp = eb + m.e_qpos(), qv = eb + m.e_qvel();
    const bool con_on = !(m.disableflags() & DIS_CONSTRAINT);
    const int row0 = m.neq() + m.nfri() + m.nlim();
    const bool eq_groups = m.npair() > m.npc();  // a property of the model: connect / weld equalities present
    if (con_on && (!(m.disableflags() & DIS_CONTACT) || eq_groups)) {
      // contact Jacobian entries: one lane per (contact, dof of the pair's ancestor chains); rows are stored
      // chain-compressed: Jc[contact row][slot], slot = position of the dof in the pair's dof list
      struct JEnt { int at; float r0, r1, r2, r3; };
      item_loop2<G>(m.njac(), gl,
        [&](int t) {
          const int pk = CI(m.c_jac() + t), c = pk & 63, slot = (pk >> 6) & 31, i = (pk >> 11) & 31, sg = (pk >> 16) & 3, pr = pk >> 18;  // resolved on the host
          const int rec = m.c_pair() + pr * PAIR_S;
          const float mu = CF(rec + 11);
          const int root1 = CI(rec + 7), root2 = CI(rec + 8);
          const f3 pt = mk3(con(1, c), con(2, c), con(3, c));
          const int fo = eb + m.e_frame() + pr * 9;
          const f3 fn_ = mk3(lds[fo], lds[fo + 1], lds[fo + 2]), ft1 = mk3(lds[fo + 3], lds[fo + 4], lds[fo + 5]),
                   ft2 = mk3(lds[fo + 6], lds[fo + 7], lds[fo + 8]);
          const f3 w = mk3(dof(0, i), dof(1, i), dof(2, i)), l = mk3(dof(3, i), dof(4, i), dof(5, i));
          if (m.nbl() > 0 && pr >= m.npair() - m.nbl())  // a ball joint's limit: -axis on the joint's three dofs (slot = 0, 1, 2), rows 1..3 idle
            return JEnt{Jo + (4 * c) * ldc + slot, -lds[fo + (slot < 3 ? slot : 0)], 0.0f, 0.0f, 0.0f};
          if (eq_groups && pr >= m.npc()) {
            // a connect / weld row group (oracle: make_constraint): body1's Jacobian minus body2's
            const bool s1 = sg & 1, s2 = sg & 2;  // the dof moves body2 / body1
            const int at = Jo + (4 * c) * ldc + slot;
            if (CI(rec + 2) == 8) {  // points: fo holds p1 (fixed in body1), p2 (fixed in body2)
              const f3 p1 = fn_, p2 = ft1;
              const f3 a1 = l + cross(w, p1 - ld_com(root1)), a2 = l + cross(w, p2 - ld_com(root2));
              const f3 jp8 = mk3((s2 ? a1.x : 0.0f) - (s1 ? a2.x : 0.0f), (s2 ? a1.y : 0.0f) - (s1 ? a2.y : 0.0f), (s2 ? a1.z : 0.0f) - (s1 ? a2.z : 0.0f));
              return JEnt{at, jp8.x, jp8.y, jp8.z, 0.0f};
            }
            // orientation: 0.5 * vec(conj(q2) * (0, jacr1 - jacr2) * q1 * relpose) * torquescale; fo holds q1 * relpose and conj(q2)
            const float sgn = (s2 ? 1.0f : 0.0f) - (s1 ? 1.0f : 0.0f);
            const f4 q1r = {lds[fo], lds[fo + 1], lds[fo + 2], lds[fo + 3]}, q2c = {lds[fo + 4], lds[fo + 5], lds[fo + 6], lds[fo + 7]};
            const f4 jw = {0.0f, sgn * w.x, sgn * w.y, sgn * w.z};
            const f4 v = qmul(qmul(q2c, jw), q1r);
            const float h = 0.5f * mu;  // field 11 of an equality group is its torquescale
            return JEnt{at, h * v.x, h * v.y, h * v.z, 0.0f};
          }
          f3 jp;
          if (m.jac_one_sided()) {  // a property of the model: contacts with the ground only - one arm, no selects
            jp = l + cross(w, pt - ld_com(root2));
          } else {
            const f3 com1 = ld_com(root1), com2 = ld_com(root2);  // unconditional loads, selected below
            // both arms computed and selected (as `if`s the loads above sank into two branches, each with its own waits)
            const f3 a2 = l + cross(w, pt - com2), a1 = l + cross(w, pt - com1);
            const bool s1 = sg & 1, s2 = sg & 2;
            jp = mk3(s1 ? a2.x : 0.0f, s1 ? a2.y : 0.0f, s1 ? a2.z : 0.0f);
            jp = mk3(s2 ? jp.x - a1.x : jp.x, s2 ? jp.y - a1.y : jp.y, s2 ? jp.z - a1.z : jp.z);
          }
          const float jn = dot(fn_, jp), j1 = dot(ft1, jp), j2 = dot(ft2, jp);
          // (compile-time: the generic kernels, which never run the three-row form, keep the code - and the roundings - they had)
          if constexpr (S::is_static && S::jfr) return JEnt{Jo + (3 * c) * ldc + slot, jn, mu * j1, mu * j2, 0.0f};  // three rows per contact; the pyramid rows are their sums and differences
          else return JEnt{Jo + (4 * c) * ldc + slot, jn + mu * j1, jn - mu * j1, jn + mu * j2, jn - mu * j2};
        },
        [&](int, const JEnt &r) { lds[r.at] = r.r0; lds[r.at + ldc] = r.r1; lds[r.at + 2 * ldc] = r.r2; if constexpr (!(S::is_static && S::jfr)) lds[r.at + 3 * ldc] = r.r3; });
      stage_chain(qv, 1);  // for the rows' J * qvel below
    }
    WSYNC();
    // slot A: equality / friction-loss / limit row gl
    float dd = 0, ar = 0, fl_ = 0;
    if (con_on && gl < nA()) {
      const int r = gl;
      float pos = 0, invw = 0, jv = 0;
      bool active = false;
      int rec = 0;  // points at the {k,b,dmin,...} block of the row's record
      if (r < m.neq()) {
        if (!(m.disableflags() & DIS_EQUALITY)) {
          active = true;
          const int er = m.c_eq() + r * EQ_S, q1 = CI(er), q2 = CI(er + 2);
          const float pos1 = lds[qp + q1] - CF(er + 19);
          invw = CF(er + 18);
          rJ1 = 1.0f; rJ2 = 0.0f;
          jv = lds[qv + rd1];
          if (q2 >= 0) {
            const float dif = lds[qp + q2] - CF(er + 20);
            const float c0 = CF(er + 4), c1 = CF(er + 5), c2 = CF(er + 6), c3 = CF(er + 7), c4 = CF(er + 8);
            const float dp2 = dif * dif, dp3 = dp2 * dif, dp4 = dp3 * dif;
            const float deriv = c1 + c2 * dif * 2 + c3 * dp2 * 3 + c4 * dp3 * 4;
            pos = pos1 - (c0 + c1 * dif + c2 * dp2 + c3 * dp3 + c4 * dp4);
            rJ2 = -deriv;
            jv += -deriv * lds[qv + rd2];
          } else {
            pos = pos1 - CF(er + 4);
          }
          rec = er + 9;
        }
      } else if (r < m.neq() + m.nfri()) {
        // friction loss (MuJoCo mj_instantiateFriction): J = e_dof, pos = 0, so D and b are constants of the row
        if (!(m.disableflags() & DIS_FRICTIONLOSS)) {
          const int fr_ = m.c_fri() + (r - m.neq()) * FRI_S;
          rJ1 = 1.0f;
          dd = CF(fr_ + 1);
          ar = -CF(fr_ + 2) * lds[qv + rd1];
          fl_ = CF(fr_ + 3);
        }
      } else if (!(m.disableflags() & DIS_LIMIT)) {
        const int lr = m.c_lim() + (r - m.neq() - m.nfri()) * LIM_S;
        const float q = lds[qp + CI(lr)];
        const float dmin = q - CF(lr + 2), dmax = CF(lr + 3) - q;
        pos = fminf(dmin, dmax) - CF(lr + 4);
        active = pos < 0;
        rJ1 = dmin < dmax ? 1.0f : -1.0f;
        jv = rJ1 * lds[qv + rd1];
        invw = CF(lr + 14);
        rec = lr + 5;
      }
      if (active) {
        const float imp = row_impedance(rec, pos);
        dd = fdiv(imp, fmaxf(invw * (1 - imp), MINVAL * imp));  // 1 / max(invw (1 - imp) / imp, MINVAL) with one division
        ar = -CF(rec + 1) * jv - CF(rec) * imp * pos;
      }
    }
    aD = dd;
    aAref = ar;
    aFl = fl_;
    // slot B: contact pyramid row gl, or row gl of a connect / weld group
    dd = 0; ar = 0;
    if (m.nbl() > 0 && cBl) {
      // a ball joint's limit: row 0 of the group, active when the rotation angle is within the margin of the range
      const int cr = gl, c = cr >> 2, k = cr & 3;
      const int prec = m.c_pair() + CI(m.c_conpair() + c) * PAIR_S, rec = prec + 12;
      const float pos = con(0, c) - CF(prec + 21), invw = CF(prec + 22);
      const float jv = chain_dot(Jo + cr * ldc);
      if (con_on && k == 0 && pos < 0 && !(m.disableflags() & DIS_LIMIT)) {
        const float imp = row_impedance(rec, pos);
        dd = fdiv(imp, fmaxf(invw * (1 - imp), MINVAL * imp));  // 1 / max(invw (1 - imp) / imp, MINVAL) with one division
        ar = -CF(rec + 1) * jv - CF(rec) * imp * pos;
      }
    } else
    if (eq_groups && cEq) {
      // three rows per group (the fourth idles), always active, impedance at the norm of the equality's whole residual
      const int cr = gl, c = cr >> 2, k = cr & 3;
      const int prec = m.c_pair() + CI(m.c_conpair() + c) * PAIR_S, rec = prec + 12;
      const float nrm = con(0, c), pos = con(1 + (k < 3 ? k : 0), c), invw = CF(prec + 22);
      const float jv = chain_dot(Jo + cr * ldc);
      if (con_on && k < 3 && !(m.disableflags() & DIS_EQUALITY)) {
        const float imp = row_impedance(rec, nrm);
        dd = fdiv(imp, fmaxf(invw * (1 - imp), MINVAL * imp));  // 1 / max(invw (1 - imp) / imp, MINVAL) with one division
        ar = -CF(rec + 1) * jv - CF(rec) * imp * pos;
      }
    } else
    if (con_on && gl < nB() && !(m.disableflags() & DIS_CONTACT)) {
      const int cr = gl, c = cr >> 2, k = cr & 3;
      const int pr = CI(m.c_conpair() + c), prec = m.c_pair() + pr * PAIR_S;
      const float margin = CF(prec + 21), dist = con(0, c);
      const float pos = dist - margin;
      const float invw = CF(prec + 22);  // (tw + mu^2 tw) 2 mu^2 / impratio, folded by the planner
      const float jv = contact_row_dot(cr, vch_base);  // qvel was staged with the Jacobian entries above
      if (dist < margin) {
        const int rec = prec + 12;
        const float imp = row_impedance(rec, pos);
        dd = fdiv(imp, fmaxf(invw * (1 - imp), MINVAL * imp));  // 1 / max(invw (1 - imp) / imp, MINVAL) with one division
        ar = -CF(rec + 1) * jv - CF(rec) * imp * pos;
      }
    }
    cD = dd;
#ifdef AMX_EXPERIMENT_SQRT_IEEE
    cSd = sqrtf(dd);
#else
    cSd = __builtin_amdgcn_sqrtf(dd);  // once per step: D of a row is fixed over the solver's iterations, only its activity changes
#endif
    cAref = ar;
    if constexpr (NBS == 2) {
      dd = 0; ar = 0;
      if (con_on && validB2() && !(m.disableflags() & DIS_CONTACT)) {
        const int cr = gl + G, c = cr >> 2;
        const int pr = CI(m.c_conpair() + c), prec = m.c_pair() + pr * PAIR_S;
        const float margin = CF(prec + 21), dist = con(0, c);
        const float pos = dist - margin;
        const float invw = CF(prec + 22);
        const float jv = contact_row_dot(cr, vch_base2);
        if (dist < margin) {
          const int rec = prec + 12;
          const float imp = row_impedance(rec, pos);
          dd = fdiv(imp, fmaxf(invw * (1 - imp), MINVAL * imp));
          ar = -CF(rec + 1) * jv - CF(rec) * imp * pos;
        }
      }
      c2D = dd;
      c2Sd = __builtin_amdgcn_sqrtf(dd);
      c2Aref = ar;
    }
  }

  // ------------------------------------------------------------------ A9.13 solver
  // (M v)[gl], v = dof field vf.  v is first staged contiguously in the (idle) H buffer so that both the lane's M row and
  // v stream in as ds_read_b128 (v: one address per group = broadcast); 2 * ldm/4 wide loads replace 2 * nv scalar ones.
  DEV float mul_M_row(int vf) const {
    const int nv = m.nv(), ldm = m.ldm(), i = gl < nv ? gl : 0, Vo = eb + m.e_H();
    const float dv_ = dof(vf, gl < nv ? gl : 0);  // unconditional load, then selects
    if (gl < ldm) lds[Vo + gl] = gl < nv ? dv_ : 0.0f;
    if (nB() > 0) stage_chain(dofoff(vf), DOF_E);  // the same vector in the order of the contact rows' dof lists (row_dot_B follows every M*v)
    WSYNC();
    const float4 *mr = reinterpret_cast<const float4 *>(lds + eb + m.e_M() + i * ldm), *vv = reinterpret_cast<const float4 *>(lds + Vo);
    float acc0 = 0, acc1 = 0;
    const int nch = ldm >> 2;
    if constexpr (S::is_static && S::mcomp) {
      // compact M: the row through m_row_compact, then the same multiply-adds in the same order as the whole-row form below
      float a[S::nv + 1];
      m_row_compact<S, S::nv>(a, eb + m.e_M(), gl, m_diag, m_win, gl < S::nv, false, 0);
      float ax = 0.0f, ay = 0.0f;  // (the whole-row form's two packed lanes)
#pragma unroll
      for (int c = 0; c < (S::nv + 3) / 4; ++c) {
        const float4 v = vv[c];
        const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (4 * c + q < S::nv) { if (q & 1) ay = __builtin_fmaf(a[4 * c + q], e[q], ay); else ax = __builtin_fmaf(a[4 * c + q], e[q], ax); }
      }
      return ax + ay;
    } else
    if constexpr (S::is_static) {  // static chunk count: one straight block, every load in flight
#if AMX_PK_DOT
      // packed fp32 (v_pk_fma_f32: two multiply-adds per instruction): a 16-byte load lands in two aligned register pairs, which
      // is the operand form the packed instructions take - two per chunk instead of four scalar ones
      f2v acc = {0.0f, 0.0f};
#pragma unroll
      for (int c = 0; c < S::ldm / 4; ++c) {
        int pc = c + m_rot;
        pc -= pc >= S::ldm / 4 ? S::ldm / 4 : 0;
        const float4 a = mr[pc], v = vv[c];
        acc = pk_fma(f2v{a.x, a.y}, f2v{v.x, v.y}, acc);
        acc = pk_fma(f2v{a.z, a.w}, f2v{v.z, v.w}, acc);
      }
      return acc.x + acc.y;
#else
#pragma unroll
      for (int c = 0; c < S::ldm / 4; ++c) {
        int pc = c + m_rot;
        pc -= pc >= S::ldm / 4 ? S::ldm / 4 : 0;
        const float4 a = mr[pc], v = vv[c];
        acc0 = __builtin_fmaf(a.x, v.x, __builtin_fmaf(a.y, v.y, acc0));
        acc1 = __builtin_fmaf(a.z, v.z, __builtin_fmaf(a.w, v.w, acc1));
      }
      return acc0 + acc1;
#endif
    }
#pragma unroll 5
    for (int c = 0; c < nch; ++c) {
      int pc = c + m_rot;  // chunk c of the row sits at position (c + rot) mod nch; v is staged unrotated: one address per group
      pc -= pc >= nch ? nch : 0;
      const float4 a = mr[pc], v = vv[c];
      acc0 = __builtin_fmaf(a.x, v.x, __builtin_fmaf(a.y, v.y, acc0));
      acc1 = __builtin_fmaf(a.z, v.z, __builtin_fmaf(a.w, v.w, acc1));
    }
    return acc0 + acc1;
  }
  // J[row,:] . v over the row's static support, for the lane's two row slots
  DEV float row_dot_A(int vf) const {  // unconditional loads (rd1 = 0 on lanes without a row), selects: no branch, no wait of its own
    const int vo = dofoff(vf);
    const float v1 = lds[vo + rd1 * DOF_E], v2 = lds[vo + (rd2 >= 0 ? rd2 : 0) * DOF_E];
    float a = rJ1 * v1;
    const float a2 = __builtin_fmaf(rJ2, v2, a);
    a = rd2 >= 0 ? a2 : a;
    return gl < nA() ? a : 0.0f;
  }
  // Gathers v (element i at vo + i*vs) in the order of every pair's dof list, zeros past a list's end: [npair][ldc] at e_vch.
  // The contact rows' dots then read J and v side by side with immediate offsets (chain_dot) - in place of this pass every row
  // lane extracted a dof index, formed an address and masked the entry for each of its ldc terms (r02: 5 vector instructions per
  // term, 13 terms, five dots per step).  The caller orders the pass before the dots (WSYNC).
  DEV void stage_chain(int vo, int vs) const {
    const int nvch = m.npair() * m.ldc(), out = eb + m.e_vch();
    {
      const float v = lds[vo + (vch_dof < 0 ? 0 : vch_dof) * vs];
      if (gl < nvch) lds[out + gl] = vch_dof < 0 ? 0.0f : v;
    }
    // more slots than lanes (quadruped: 4 pairs x 9; the falls-and-rests biped: 10 x 13): the dof comes from the constants
    if constexpr (S::is_static) {
      // static round count: every round's index load, then every value load, then the stores - two LDS round trips in all.  (As a loop
      // every round was two of its own: four rounds per call and some six calls a step were ~6 k cycles of the falls-and-rests biped's 91 k.)
      constexpr int NV = S::npair * S::ldc, NR = (NV + G - 1) / G - 1;
      if constexpr (NR > 0) {
        int d_[NR];
        float v[NR];
#pragma unroll
        for (int k = 0; k < NR; ++k) { const int t = gl + G * (k + 1); d_[k] = CI(S::c_vchdof + (t < NV ? t : 0)); }
#pragma unroll
        for (int k = 0; k < NR; ++k) v[k] = lds[vo + (d_[k] < 0 ? 0 : d_[k]) * vs];
#pragma unroll
        for (int k = 0; k < NR; ++k) { const int t = gl + G * (k + 1); if (t < NV) lds[out + t] = d_[k] < 0 ? 0.0f : v[k]; }
      }
    } else {
      for (int t = gl + G; t < nvch; t += G) {
        const int d_ = CI(m.c_vchdof() + t);
        const float v = lds[vo + (d_ < 0 ? 0 : d_) * vs];
        lds[out + t] = d_ < 0 ? 0.0f : v;
      }
    }
  }
  // sum over the contact row's dof list of J[t] * v[dof_t]: J at Jr, v staged by stage_chain.  Every term of the row's ldc
  // slots is loaded (slots past the list hold stale - finite - floats of the slab and meet a staged zero), so all the loads are in
  // flight together and each term is one fused multiply-add.  Two partial sums.
  DEV float chain_dot(int Jr) const { return chain_dot(Jr, vch_base); }
  DEV float chain_dot(int Jr, int vch_base) const {
#if AMX_PK_DOT
    if constexpr (S::is_static) {  // pairs of terms as packed multiply-adds (the loads are ds_read2_b32 pairs already)
      f2v acc = {0.0f, 0.0f};
#pragma unroll
      for (int t = 0; t + 1 < S::ldc; t += 2) acc = pk_fma(f2v{lds[Jr + t], lds[Jr + t + 1]}, f2v{lds[vch_base + t], lds[vch_base + t + 1]}, acc);
      float r = acc.x + acc.y;
      if (S::ldc & 1) r = __builtin_fmaf(lds[Jr + S::ldc - 1], lds[vch_base + S::ldc - 1], r);
      return r;
    }
#endif
    float a0 = 0, a1 = 0;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      if (t < m.ldc()) {  // static in the specialised kernels, wave-uniform otherwise
        const float jt = lds[Jr + t], vt = lds[vch_base + t];
        if (t & 1) a1 = __builtin_fmaf(jt, vt, a1); else a0 = __builtin_fmaf(jt, vt, a0);
      }
    }
    if (m.ldc() > 16) {  // dof lists of 17 .. 31 entries (r04: a contact between two limbs of a floating-base robot): one test for everybody else
#pragma unroll
      for (int t = 16; t < 32; ++t) {
        if (t < m.ldc()) {
          const float jt = lds[Jr + t], vt = lds[vch_base + t];
          if (t & 1) a1 = __builtin_fmaf(jt, vt, a1); else a0 = __builtin_fmaf(jt, vt, a0);
        }
      }
    }
    return a0 + a1;
  }
  // contact pyramid row cr times the staged vector.  Three-row Jacobian blocks (MV::jfr): row 4 c + k is normal +- (mu tangent 1 | mu tangent 2),
  // so its product is the sum or difference of two of the block's products
  DEV float contact_row_dot(int cr, int vb) const {
    if constexpr (S::is_static && S::jfr) {
      // lanes 4 c .. 4 c + 3 hold the contact's four pyramid rows: lane k multiplies block row min(k, 2) - one chain_dot, as in the four-row
      // form - and the quad exchanges the three products (DPP quad_perm): row k = normal +- tangent (k >> 1).  The callers' branches and
      // validity tests are whole quads (4 ncon rows), and the four lanes share the pair, hence the staged vector.
      const int c = cr >> 2, k = cr & 3;
      const float x = chain_dot(eb + m.e_J() + (3 * c + (k < 3 ? k : 2)) * S::ldc, vb);
      const float dn = dpp_f<0x00>(x), dt = dpp_f<0xA5>(x);  // quad_perm [0,0,0,0], [1,1,2,2]
      return (k & 1) ? dn - dt : dn + dt;
    }
    return chain_dot(eb + m.e_J() + cr * m.ldc(), vb);
  }
  DEV float row_dot_B(int vf) const {
    const bool valid = gl < nB();
    const float a = contact_row_dot(valid ? gl : 0, vch_base);  // the caller has staged v = dof field vf (mul_M_row does)
    return valid ? a : 0.0f;
  }
  DEV float row_dot_B2(int vf) const {
    if constexpr (NBS != 2) return 0.0f;
    const bool valid = validB2();
    const float a = contact_row_dot(valid ? gl + G : 0, vch_base2);
    return valid ? a : 0.0f;
  }
  // Ma = M v, Jaref = J v - aref  (v = dof field vf)
  DEV void init_context(int vf) {
    const float ma = mul_M_row(vf);
    if (gl < m.nv()) dof(D_MA, gl) = ma;
    r_ma = gl < m.nv() ? ma : 0.0f;
    aJaref = row_dot_A(vf) - aAref;
    cJaref = row_dot_B(vf) - cAref;
    if constexpr (NBS == 2) c2Jaref = row_dot_B2(vf) - c2Aref;
    WSYNC();
  }
  DEV float constraint_cost(int vf, float *gauss_out) const {
    const bool eqB = m.npair() > m.npc() && cEq;  // rows of connect / weld groups are active on both sides
    const bool actA = (gl < nA()) & ((aJaref < 0) | (gl < m.neq())), actB = (gl < nB()) & ((cJaref < 0) | eqB);
    float ca_ = actA ? 0.5f * aD * aJaref * aJaref : 0.0f;
    if (m.nfri() > 0) {  // a property of the model.  Friction-loss rows: quadratic inside |Jaref| < rf = aFl / aD, linear outside
      const float rf = fdiv(aFl, fmaxf(aD, MINVAL)), z = aJaref;
      const float lin = aFl * (fabsf(z) - 0.5f * rf), quad = 0.5f * aD * z * z;
      ca_ = aFl > 0.0f ? (fabsf(z) >= rf ? lin : quad) : ca_;
    }
    float c = ca_ + (actB ? 0.5f * cD * cJaref * cJaref : 0.0f);
    if constexpr (NBS == 2) c += (validB2() & (c2Jaref < 0)) ? 0.5f * c2D * c2Jaref * c2Jaref : 0.0f;
    const int li = gl < m.nv() ? gl : 0;  // unconditional loads, then a select (a lazily evaluated ?: is a branch with its own wait)
    const float gv = (r_ma - r_fs) * (dof(vf, li) - r_as);
    float g = gl < m.nv() ? gv : 0.0f;
    const float r = gsum_pack2<G>(c, g);  // both sums in one packed reduction
    c = quad_value<0>(r);
    g = 0.5f * quad_value<1>(r);
    *gauss_out = g;
    return c + g;
  }
  // qfrc_constraint = J' f with f = -D*Jaref on active rows, grad = Ma - qfrc_smooth - qfrc_constraint;
  // also publishes D*active per row for the Hessian assembly.
  // grad_in_solve(): the specialised fused Newton path takes J'f of the contact rows from the matrix cores (newton_solve_fused
  // GRADCOL), so a call that is followed by newton_direction() only publishes the row records and the limit-row force
  static constexpr bool grad_in_solve() { return S::is_static && (G == 32 || (G == 64 && AMX_GRADCOL_64)) && S::fusedH && S::solver == 2 && S::nv <= 31 && sym_elim_ok<S, G>() && !AMX_NO_GRADCOL; }
  static_assert(!(S::mcomp || S::jfr) || (grad_in_solve() && G == 32 && !S::mfma2), "the planner's dense form lives in the symmetric eliminations and the gradient-column Newton solve of 32-lane groups");
  // the two-pair Newton solve (newton_solve_pair2) also writes the search vector
  static constexpr bool search_from_solve() { return grad_in_solve() && S::mfma2 && AMX_MFMA_PAIR2; }
  float pend_flim;  // limit-row force on this lane's dof, handed from a deferring constraint_force_and_grad to newton_direction
  // x of the single-entry rows on this lane's dof, summed: its limit row and (models with friction loss) its friction-loss row, each
  // fetched from the row's lane through the crossbar; 0 on lanes without a dof or without such rows
  DEV float dof_rows_fetch(float x) const {
    float r = __int_as_float(__builtin_amdgcn_ds_bpermute(gb4 + 4 * (d_limrow < 0 ? 0 : d_limrow), __float_as_int(x)));
    r = (gl < m.nv() && d_limrow >= 0) ? r : 0.0f;
    if (m.nfri() > 0) {
      const float r2 = __int_as_float(__builtin_amdgcn_ds_bpermute(gb4 + 4 * (d_frirow < 0 ? 0 : d_frirow), __float_as_int(x)));
      r += (gl < m.nv() && d_frirow >= 0) ? r2 : 0.0f;
    }
    return r;
  }
  // alive: false for a group whose solve has converged while its wave mate iterates on - its force and gradient are left as
  // they are (recomputing them is idempotent in value but not in bits: the matrix-core sum and the gather round differently,
  // and which of the two ran last must not depend on the wave mate)
  DEV void constraint_force_and_grad(bool defer = false, bool alive = true) {
    const int Jo = eb + m.e_J(), ldc = m.ldc(), row0 = m.neq() + m.nfri() + m.nlim();
    aact = (gl < nA()) & ((aJaref < 0) | (gl < m.neq())) & (aD > 0);
    cact = (gl < nB()) & ((cJaref < 0) | (m.npair() > m.npc() && cEq)) & (cD > 0);
    float f = aact ? -aD * aJaref : 0.0f;
    const float fc = cact ? -cD * cJaref : 0.0f;
    if (m.nfri() > 0) {  // friction-loss rows: the force saturates at +-aFl; only the quadratic zone enters the Hessian
      const bool fri = aFl > 0.0f;
      const float fq = -aD * aJaref, fs = fminf(fmaxf(fq, -aFl), aFl);
      f = fri ? fs : f;
      aact = fri ? (fabsf(fq) < aFl) : aact;
    }
    if (gl < nA()) { row(0, gl) = aact ? aD : 0.0f; row(1, gl) = f; }
    // field 0 of a contact row: what the Hessian's products take - sqrt(D) of an active row (AMX_SQRT_D; D before r05), 0 otherwise
    if (gl < nB()) { row(0, row0 + gl) = cact ? (AMX_SQRT_D ? cSd : cD) : 0.0f; row(1, row0 + gl) = fc; row(2, row0 + gl) = cJaref; }  // field 2: for the augmented J'DJ (newton_solve_fused)
    if constexpr (NBS == 2) {
      c2act = validB2() & (c2Jaref < 0) & (c2D > 0);
      const float fc2 = c2act ? -c2D * c2Jaref : 0.0f;
      if (validB2()) { row(0, row0 + gl + G) = c2act ? (AMX_SQRT_D ? c2Sd : c2D) : 0.0f; row(1, row0 + gl + G) = fc2; row(2, row0 + gl + G) = c2Jaref; }
    }
    WSYNC();
    if (grad_in_solve() && defer) {
      float fl = 0.0f;
      if (nA() > 0) {
        fl = dof_rows_fetch(rJ1 * f);
      }
      pend_flim = fl;
      return;
    }
    // contact rows: dof lane i gathers over the (pair, slot) memberships of its dof.  Straight-line: the trip count is the
    // model's largest membership list (static in the specialised kernels, so the loop unrolls and every load of every term
    // is in flight together); a lane past its own list re-reads term 0 and adds 0.  Two terms per trip, summed in list order.
    {
      const bool isd = gl < m.nv();
      float acc = 0;
      const int mem0 = CI(m.c_dofmemadr() + (isd ? gl : 0)), mem1 = isd ? CI(m.c_dofmemadr() + (isd ? gl : 0) + 1) : mem0;
      auto two_terms = [&](int u) {
        const int t0 = mem0 + u, t1 = t0 + 1;
        const bool w0 = t0 < mem1, w1 = t1 < mem1;
        const int pk0 = CI(m.c_dofmem() + (w0 ? t0 : 0)), pk1 = CI(m.c_dofmem() + (w1 ? t1 : 0));
        const int c0 = pk0 & 255, c1 = pk1 & 255;
        const int Ja = Jo + c0 * ldc + (pk0 >> 8), Fa = eb + m.e_row() + (row0 + c0) * ROW_E + 1;
        const int Jb = Jo + c1 * ldc + (pk1 >> 8), Fb = eb + m.e_row() + (row0 + c1) * ROW_E + 1;
        float ja[4], fa[4], jb[4], fb[4];
        if constexpr (S::is_static && S::jfr) {
          // three-row blocks: c0, c1 are the contacts' first pyramid rows (4 c), their blocks start at row 3 c
          const int Ja3 = Ja - (c0 >> 2) * ldc, Jb3 = Jb - (c1 >> 2) * ldc;
          const float na = lds[Ja3], ta1 = lds[Ja3 + ldc], ta2 = lds[Ja3 + 2 * ldc], nb_ = lds[Jb3], tb1 = lds[Jb3 + ldc], tb2 = lds[Jb3 + 2 * ldc];
          ja[0] = na + ta1; ja[1] = na - ta1; ja[2] = na + ta2; ja[3] = na - ta2;
          jb[0] = nb_ + tb1; jb[1] = nb_ - tb1; jb[2] = nb_ + tb2; jb[3] = nb_ - tb2;
#pragma unroll
          for (int k = 0; k < 4; ++k) { fa[k] = lds[Fa + k * ROW_E]; fb[k] = lds[Fb + k * ROW_E]; }
        } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) { ja[k] = lds[Ja + k * ldc]; fa[k] = lds[Fa + k * ROW_E]; jb[k] = lds[Jb + k * ldc]; fb[k] = lds[Fb + k * ROW_E]; }
        }
        const float sa = (ja[0] * fa[0] + ja[1] * fa[1]) + (ja[2] * fa[2] + ja[3] * fa[3]);
        const float sb = (jb[0] * fb[0] + jb[1] * fb[1]) + (jb[2] * fb[2] + jb[3] * fb[3]);
        acc += w0 ? sa : 0.0f;
        acc += w1 ? sb : 0.0f;
      };
#pragma unroll
      for (int u = 0; u < m.max_dofmem(); u += 2) two_terms(u);
      // (tried: eight terms' loads in flight at a time behind scheduling barriers, for the sixteen-term lists of a contact-rich model - no faster)
      // limit rows have one entry each and a dof at most one limit row: without equality rows the dof lane fetches its row's
      // force through the crossbar instead of the rows scattering with LDS atomics
      if (m.neq() == 0 && nA() > 0) {
        acc += dof_rows_fetch(rJ1 * f);
      }
      if (isd && alive) dof(D_FCON, gl) = acc;
      if (m.neq() == 0) {  // no scattered rows: the gradient is complete in this lane's registers
        if (isd && alive) dof(D_GRAD, gl) = r_ma - r_fs - acc;
        WSYNC();
        return;
      }
    }
    WSYNC();
    // equality / limit rows scatter their one or two entries
    if (m.neq() > 0 && gl < row0 && f != 0.0f && alive) {
      atomicAdd(&dof(D_FCON, rd1), rJ1 * f);
      if (rd2 >= 0) atomicAdd(&dof(D_FCON, rd2), rJ2 * f);
    }
    WSYNC();
    if (gl < m.nv() && alive) dof(D_GRAD, gl) = r_ma - r_fs - dof(D_FCON, gl);
    WSYNC();
  }
  // Mgrad = (M + J' D_active J)^-1 grad, H a full symmetric matrix in LDS.
  // G >= 32: the contact rows' J'DJ comes from the matrix cores and lands in H together with M (build_H_mfma); equality and
  // limit rows, rarely active and one or two entries each, are added with LDS atomics only when one of them is active.
  // G = 16 (four environments per wave do not map onto the two MFMA blocks): copy of M + atomics for every row kind.
  // returns true when (specialised fused path) the stopping test of the next iteration, evaluated inside, stops every group of the
  // wave: the direction was not computed and the caller must not iterate again
  DEV bool newton_direction(bool alive = true, bool early_out = false, bool imp_done = false, bool *group_done = nullptr) {
    const int nv = m.nv(), Mo = eb + m.e_M(), Ho = eb + m.e_H(), Jo = eb + m.e_J(), ldc = m.ldc(), ldm = m.ldm();
    const int row0 = m.neq() + m.nfri() + m.nlim();
    if constexpr (G >= 32) {
      if (m.fusedH()) {
        const unsigned long long act = __ballot(cact);
        unsigned long long rows = G == 32 ? ((act | (act >> 32)) & 0xFFFFFFFFull) : act;  // contact row active in any environment of the wave
        unsigned long long rows2 = 0ull;  // second slot's rows (row gl + G): bits 32 .. 63 of the one mask on 32-lane groups, a mask of their own on 64-lane groups
        if constexpr (NBS == 2) {
          const unsigned long long act2 = __ballot(c2act);
          if (G == 32) rows |= ((act2 | (act2 >> 32)) & 0xFFFFFFFFull) << 32; else rows2 = act2;
        }
        // diagonal term of the limit row on this lane's dof: D of that row if it is active (J = +-1), through the crossbar
        const float dsrc = aact ? aD : 0.0f;
        const float dlim = nA() > 0 ? dof_rows_fetch(dsrc) : 0.0f;  // J = +-1 on these rows: J' D J is D on the diagonal
        PROF(19);
        bool stop = false;
#ifdef AMX_PROFILE
        ProfCtx pctx{&prof_t, prof_row}, *pc = &pctx;
#else
        ProfCtx *pc = nullptr;
#endif
        if constexpr (grad_in_solve() && S::mfma2 && AMX_MFMA_PAIR2)
          stop = newton_solve_pair2<S, S::tree_H>(Mo, rows, eb + m.e_row() + row0 * ROW_E, Jo, dofoff(D_GRAD), dofoff(D_MGRAD), dlim, gl, gb4, m_rot, eb + m.e_H(), pend_flim,
                                                  r_ma, r_fs, dofoff(D_FCON), alive, early_out, imp_done, m.scale(), m.tolerance(), pc, group_done);
        else if constexpr (grad_in_solve())
          stop = newton_solve_fused<S, S::nv, G, S::tree_H, true>(Mo, ldm, nv, rows, eb + m.e_row() + row0 * ROW_E, Jo, ldc, m.c_slot(), dofoff(D_GRAD), dofoff(D_MGRAD), dlim, gl, gb4, m_rot,
                                                                 eb + m.e_H(), pend_flim, r_ma, r_fs, dofoff(D_FCON), alive, early_out, imp_done, m.scale(),
                                                                 m.tolerance(), pc, group_done, rows2, m_diag, m_win);
        else if constexpr (S::is_static) {
          static_assert(!S::mcomp && !S::jfr, "the dense form lives in the gradient-column Newton solve");
          newton_solve_fused<S, S::nv, G, S::tree_H>(Mo, ldm, nv, rows, eb + m.e_row() + row0 * ROW_E, Jo, ldc, m.c_slot(), dofoff(D_GRAD), dofoff(D_MGRAD), dlim, gl, gb4, m_rot, eb + m.e_H());
        }
        else
          newton_fused_dyn<G>(m.nchol(), Mo, ldm, nv, rows, eb + m.e_row() + row0 * ROW_E, Jo, ldc, m.c_slot(), dofoff(D_GRAD), dofoff(D_MGRAD), dlim, gl, gb4, m_rot);
        PROF(14);
        return stop;
      }
    }
    bool scalar_rows = true;  // wave-uniform: some equality / limit row is active
    if (G >= 32) {
      const unsigned long long act = __ballot(cact);
      unsigned long long rows = G == 32 ? ((act | (act >> 32)) & 0xFFFFFFFFull) : act;  // contact row active in any environment of the wave
      unsigned long long rows2 = 0ull;  // second contact-row slot (r05: also on this path - models with joint-equality rows): as in the fused path
      if constexpr (NBS == 2) {
        const unsigned long long act2 = __ballot(c2act);
        if (G == 32) rows |= ((act2 | (act2 >> 32)) & 0xFFFFFFFFull) << 32; else rows2 = act2;
      }
      scalar_rows = wave_any(aact);
      build_H_mfma<G>(Mo, Ho, ldm, nv, rows, m.ncon(), eb + m.e_row() + row0 * ROW_E, Jo, ldc, m.c_slot(), gl, rows2);
    } else {
      lds_copy<G>(Ho, Mo, nv * ldm, gl);
      WSYNC();
    }
    PROF(19);
    if (scalar_rows) {
      if (gl < row0) {  // equality + limit rows: a handful of entries each, from registers
        const float d = row(0, gl);
        if (d != 0.0f) {
          atomicAdd(&lds[Ho + rd1 * ldm + rd1], d * rJ1 * rJ1);
          if (rd2 >= 0) {
            atomicAdd(&lds[Ho + rd2 * ldm + rd2], d * rJ2 * rJ2);
            atomicAdd(&lds[Ho + rd1 * ldm + rd2], d * rJ1 * rJ2);
            atomicAdd(&lds[Ho + rd2 * ldm + rd1], d * rJ1 * rJ2);
          }
        }
      }
      WSYNC();
    }
    PROF(20);
    if (G < 32) {
      // contact rows: one lane per term (H entry, contact): the twelve operands of the contact's four pyramid rows are
      // fetched back to back, then reduced and added to the entry (ds_add_f32; entries are shared between contacts)
      struct HEnt { int at, at2; float v; };
      item_loop2<G>(m.nhent(), gl,
        [&](int t) {
          const int pk = CI(m.c_hent() + t), di = pk & 31, dj = (pk >> 5) & 31, cr0 = (pk >> 10) & 255, sa = (pk >> 18) & 31, sb = pk >> 23;
          const int Jr = Jo + cr0 * ldc, Dr = eb + m.e_row() + (row0 + cr0) * ROW_E;
          float d[4], ja[4], jb[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) { d[k] = lds[Dr + k * ROW_E]; ja[k] = lds[Jr + k * ldc + sa]; jb[k] = lds[Jr + k * ldc + sb]; }
#if AMX_SQRT_D
#pragma unroll
          for (int k = 0; k < 4; ++k) { ja[k] *= d[k]; jb[k] *= d[k]; d[k] = 1.0f; }  // d is sqrt(D): symmetric products, as on the matrix cores
#endif
          return HEnt{Ho + di * ldm + dj, Ho + dj * ldm + di, (d[0] * ja[0] * jb[0] + d[1] * ja[1] * jb[1]) + (d[2] * ja[2] * jb[2] + d[3] * ja[3] * jb[3])};
        },
        [&](int, const HEnt &r) { if (r.v != 0.0f) { atomicAdd(&lds[r.at], r.v); if (r.at2 != r.at) atomicAdd(&lds[r.at2], r.v); } });
      WSYNC();
    }
    PROF(14);
    if constexpr (S::is_static) spd_solve_static<S, G, S::tree_H>(Ho, dofoff(D_GRAD), dofoff(D_MGRAD), gl, m_rot, -1);
    else spd_solve<G>(m.nchol(), Ho, ldm, nv, dofoff(D_GRAD), dofoff(D_MGRAD), gl, gb4, m_rot);
    return false;
  }

  // One trial step length alpha: the quadratic model of the cost along the search direction under the active set at alpha,
  //   cost(alpha) = alpha^2 q2 + alpha q1 + q0,  q_k = sum over the active rows (+ the Gauss terms g_k).
  // ls_terms gives THIS LANE's share of q1 and q2 (both row slots); the group sums of the three trial points of a bracketing
  // iteration are formed together (gsum_pack6).  The bracketing loop only ever compares derivatives, so it needs q1 and q2
  // only; the two bracket ends are evaluated in full (q0 too) once, after the loop - the same expressions as inside MJX's loop,
  // evaluated at the same alphas, so the accepted step is the one the reference's search takes.
  struct LSTerms { float q1, q2; };
  DEV LSTerms ls_terms(float alpha, bool validA, bool validB, bool iseq, float *q0 = nullptr) const {
    const float xA = aJaref + alpha * aJv;
    const bool actA = validA & ((xA < 0) | iseq), actB = validB & ((cJaref + alpha * cJv < 0) | (m.npair() > m.npc() && cEq));
    float a1 = actA ? aJv * aJaref * aD : 0.0f, a2 = actA ? 0.5f * aJv * aJv * aD : 0.0f, a0 = actA ? 0.5f * aJaref * aJaref * aD : 0.0f;
    if (m.nfri() > 0) {  // friction-loss rows: the zone at alpha picks the coefficients (quadratic inside |x| < rf, linear outside)
      const bool fri = aFl > 0.0f;
      const float rf = fdiv(aFl, fmaxf(aD, MINVAL));
      const bool lin = fabsf(xA) >= rf;
      const float sg = xA < 0 ? -1.0f : 1.0f;  // linear branch: cost = aFl (-rf/2 + sg x)
      const float l1 = sg * aFl * aJv, l0 = aFl * (sg * aJaref - 0.5f * rf);
      const float q1q = aJv * aJaref * aD, q2q = 0.5f * aJv * aJv * aD, q0q = 0.5f * aJaref * aJaref * aD;
      a1 = fri ? (lin ? l1 : q1q) : a1;
      a2 = fri ? (lin ? 0.0f : q2q) : a2;
      a0 = fri ? (lin ? l0 : q0q) : a0;
    }
    LSTerms r;
    r.q1 = a1 + (actB ? cJv * cJaref * cD : 0.0f);
    r.q2 = a2 + (actB ? 0.5f * cJv * cJv * cD : 0.0f);
    if (q0) *q0 = a0 + (actB ? 0.5f * cJaref * cJaref * cD : 0.0f);
    if constexpr (NBS == 2) {  // second contact-row slot (a contact row: one-sided)
      const bool actB2 = validB2() & (c2Jaref + alpha * c2Jv < 0);
      r.q1 += actB2 ? c2Jv * c2Jaref * c2D : 0.0f;
      r.q2 += actB2 ? 0.5f * c2Jv * c2Jv * c2D : 0.0f;
      if (q0) *q0 += actB2 ? 0.5f * c2Jaref * c2Jaref * c2D : 0.0f;
    }
    return r;
  }
  struct LSD { float alpha, d0, d1; };
  // first and second derivative from the summed q1, q2 (Gauss terms included).  q2 is a sum of non-negative terms: the
  // reference's `2 q2 + (q2 == 0) * MINVAL` is max(2 q2, MINVAL) for every q2 a float can hold above 5e-16
  static DEV LSD ls_derivs(float alpha, float q1, float q2) { return LSD{alpha, __builtin_fmaf(alpha + alpha, q2, q1), fmaxf(q2 + q2, MINVAL)}; }

  // MJX's bracketing line search.  `alive` freezes the state of a group that has already converged.
  // Returns the cost decrease of the accepted step (0: no step), which is the solver's `improvement` for its stopping test.
  DEV float linesearch(bool alive) {
    const int nv = m.nv();
    const float mv = mul_M_row(D_SEARCH);
    aJv = row_dot_A(D_SEARCH);
    cJv = row_dot_B(D_SEARCH);
    if constexpr (NBS == 2) c2Jv = row_dot_B2(D_SEARCH);
    float sn = 0, g1 = 0, g2 = 0;
    {  // unconditional loads, selects at the end
      const int li = gl < nv ? gl : 0;
      const float si = dof(D_SEARCH, li), ma_ = r_ma, fs_ = r_fs;
      const float sn_ = si * si, g1_ = si * ma_ - si * fs_, g2_ = 0.5f * si * mv;
      sn = gl < nv ? sn_ : 0.0f; g1 = gl < nv ? g1_ : 0.0f; g2 = gl < nv ? g2_ : 0.0f;
    }
    const bool validA = gl < nA(), validB = gl < nB(), iseq = gl < m.neq();
    // the three dof sums and the three row sums of the starting point alpha = 0, in one packed reduction
    float q0_start;  // this lane's constant term at alpha = 0 (see the acceptance test below)
    LSD p0d;
    {
      const LSTerms t = ls_terms(0.0f, validA, validB, iseq, &q0_start);
      float u, w;
      gsum_pack6<G>(g1, g2, t.q1, t.q2, sn, 0.0f, u, w);
      g1 = quad_value<0>(u); g2 = quad_value<1>(u); sn = quad_value<0>(w);
      const float q1 = quad_value<3>(u) + g1, q2 = quad_value<2>(u) + g2;
      p0d = ls_derivs(0.0f, q1, q2);
    }
    PROF(21);
    const float gtol = m.tolerance() * m.ls_tolerance() * __builtin_amdgcn_sqrtf(sn) * m.scale();  // a threshold: 1 ulp is plenty
#define PICK(c, a, b) { (c) ? (a).alpha : (b).alpha, (c) ? (a).d0 : (b).d0, (c) ? (a).d1 : (b).d1 }
    LSD p1;
    {
      const float a1 = p0d.alpha - fdiv_ls(p0d.d0, p0d.d1);
      const LSTerms t = ls_terms(a1, validA, validB, iseq);
      const float r = gsum_pack2<G>(t.q1, t.q2);
      p1 = ls_derivs(a1, quad_value<0>(r) + g1, quad_value<1>(r) + g2);
    }
    const bool first = p1.d0 < p0d.d0;
    LSD lo = PICK(first, p1, p0d), hi = PICK(first, p0d, p1);
    bool swap = true;
    // one bracketing iteration; returns false when every group of the wave has finished its search
    auto bracket_step = [&]() {
      // `|` and `&` on purpose, here and below: the short-circuit forms compile to nested exec-mask branches (s_and_saveexec,
      // s_cbranch_execz per operand - a dozen per iteration, each a scalar round trip on the critical path); the operands are
      // comparisons of values already in registers, so there is nothing to skip
      const bool done = !swap | ((lo.d0 < 0) & (lo.d0 > -gtol)) | ((hi.d0 > 0) & (hi.d0 < gtol));
      if (!wave_any(!done & alive)) return false;
      const float a_lo = lo.alpha - fdiv_ls(lo.d0, lo.d1), a_hi = hi.alpha - fdiv_ls(hi.d0, hi.d1), a_mid = 0.5f * (lo.alpha + hi.alpha);
      const LSTerms t_lo = ls_terms(a_lo, validA, validB, iseq), t_hi = ls_terms(a_hi, validA, validB, iseq), t_mid = ls_terms(a_mid, validA, validB, iseq);
      float u, w;
      gsum_pack6<G>(t_lo.q1, t_lo.q2, t_hi.q1, t_hi.q2, t_mid.q1, t_mid.q2, u, w);
      const LSD lo_next = ls_derivs(a_lo, quad_value<0>(u) + g1, quad_value<1>(u) + g2);
      const LSD hi_next = ls_derivs(a_hi, quad_value<3>(u) + g1, quad_value<2>(u) + g2);
      const LSD mid = ls_derivs(a_mid, quad_value<0>(w) + g1, quad_value<1>(w) + g2);
      // a group whose search has ended keeps its bracket: `done` is part of every replacement test (scalar logic) instead of a
      // second round of selects
      const bool s_lo_next = !done & ((lo.d0 > 0) | (lo.d0 < lo_next.d0));
      LSD nlo = PICK(s_lo_next, lo_next, lo);
      const bool s_lo_mid = !done & (mid.d0 < 0) & (nlo.d0 < mid.d0);
      { const LSD t = PICK(s_lo_mid, mid, nlo); nlo = t; }
      const bool s_hi_next = !done & ((hi.d0 < 0) | (hi.d0 > hi_next.d0));
      LSD nhi = PICK(s_hi_next, hi_next, hi);
      const bool s_hi_mid = !done & (mid.d0 > 0) & (nhi.d0 > mid.d0);
      { const LSD t = PICK(s_hi_mid, mid, nhi); nhi = t; }
      lo = nlo; hi = nhi;
      swap = s_lo_next | s_lo_mid | s_hi_next | s_hi_mid;
      return true;
    };
    // static trip count in the specialised kernels: fully unrolled (+1 %) while the body stays small (the reference pendulum's
    // 10, the shipped 5); MuJoCo's default of 50 stays a loop - unrolled it would be 50 KB of code for iterations the
    // tolerance exit never reaches
    if constexpr (S::is_static && S::ls_iterations > 10) {
#pragma nounroll
      for (int it = 0; it < m.ls_iterations(); ++it) if (!bracket_step()) break;
    } else {
#pragma unroll
      for (int it = 0; it < m.ls_iterations(); ++it) if (!bracket_step()) break;
    }
    PROF(22);
#undef PICK
    // Costs of the two bracket ends RELATIVE TO THE STARTING POINT: cost(alpha) - cost(0) = alpha^2 q2 + alpha q1 + (q0(alpha) - q0(0)),
    // with the difference of the constant terms formed per row before the group sum - it is exactly zero for every row whose
    // activity did not change.  The reference compares cost(alpha) < cost(0) with both sides carrying the full constant term
    // (the Gauss term and every active row's 0.5 D aref^2: 1e3 .. 1e5 for contacts a few centimetres deep), so in single
    // precision an improvement below half an ulp of that constant reads as "no improvement", the step is dropped and the
    // iterate stays where the previous iteration left it (r03, convex-convex draw of tests/test_convex.py: three of 32
    // environments 50 - 110 x their one-ulp input sensitivity away from the fp64 oracle, all three from a dropped last step).
    // Same algebra, same decisions in exact arithmetic; here the comparison resolves improvements down to the rounding of the
    // quadratic terms themselves.
    float lo_delta, hi_delta;
    {
#if AMX_LS_END_FROM_DERIVS
      // the quadratic part of cost(alpha) - cost(0) is already in the bracket ends' derivatives (d1 = 2 q2, d0 = 2 alpha q2 + q1:
      // alpha^2 q2 + alpha q1 = alpha d0 - alpha^2 d1 / 2); only the constant terms' difference needs a sum
      float q0l, q0h;
      ls_terms(lo.alpha, validA, validB, iseq, &q0l);
      ls_terms(hi.alpha, validA, validB, iseq, &q0h);
      const float r = gsum_pack2<G>(q0l - q0_start, q0h - q0_start);
      lo_delta = lo.alpha * (lo.d0 - 0.5f * lo.alpha * lo.d1) + quad_value<0>(r);
      hi_delta = hi.alpha * (hi.d0 - 0.5f * hi.alpha * hi.d1) + quad_value<1>(r);
#else
      float q0l, q0h;
      const LSTerms tl = ls_terms(lo.alpha, validA, validB, iseq, &q0l), th = ls_terms(hi.alpha, validA, validB, iseq, &q0h);
      float u, w;
      gsum_pack6<G>(tl.q1, tl.q2, th.q1, th.q2, q0l - q0_start, q0h - q0_start, u, w);
      const float q1l = quad_value<0>(u) + g1, q2l = quad_value<1>(u) + g2, q1h = quad_value<3>(u) + g1, q2h = quad_value<2>(u) + g2;
      lo_delta = lo.alpha * lo.alpha * q2l + lo.alpha * q1l + quad_value<0>(w);
      hi_delta = hi.alpha * hi.alpha * q2h + hi.alpha * q1h + quad_value<1>(w);
#endif
    }
    const bool improved = alive & ((lo_delta < 0.0f) | (hi_delta < 0.0f));
    const float alpha = lo_delta < hi_delta ? lo.alpha : hi.alpha;
    if (improved) {
      if (gl < nv) { r_ma += alpha * mv; dof(D_QACC, gl) += alpha * dof(D_SEARCH, gl); dof(D_MA, gl) = r_ma; }
      aJaref += alpha * aJv;
      cJaref += alpha * cJv;
      if constexpr (NBS == 2) c2Jaref += alpha * c2Jv;
    }
    WSYNC();
    return improved ? -fminf(lo_delta, hi_delta) : 0.0f;
  }

  // ---- the same search, second form (r04; static kernels whose rows are all one-sided: no equality, friction-loss, connect or weld
  // rows).  Identical trial points and decisions; what changes is the instruction count of a bracketing iteration (101 -> ~80 vector
  // instructions; fifteen iterations a step):
  //  * the Gauss terms ride in the group sums: every lane adds its dof's share (alpha-independent) to its rows' coefficients, so the
  //    reduced values are the whole q1, q2 and the per-point "+ g1", "+ g2" additions are gone;
  //  * coefficients as packed pairs: (q1/2, q2) of a row slot is one register pair and "add the row if it is active at alpha" is one
  //    v_pk_fma_f32 with the activity as a 0 / 1 factor - a compare, a select and one packed multiply-add per row slot and trial
  //    point instead of a compare, two selects and two additions;
  //  * derivatives carried HALVED (hd = d0 / 2 = q1/2 + alpha q2, h = d1 / 2 = max(q2, MINVAL/2)): every comparison of the search is
  //    invariant under the factor (a power of two: exact), d0 / d1 = hd / h bit for bit, and "2 alpha", "2 q2" are never formed;
  //  * the bracket logic on lane MASKS: every comparison is a v_cmp into a scalar register pair, and / or / not run on the scalar
  //    unit, selects take the mask directly (the `bool` form kept `swap` as a 0 / 1 vector register across the unrolled iterations).
  static constexpr bool ls_v2_ok() {
#ifdef AMX_NO_LS_V2
    return false;
#else
    return S::is_static && S::nfri == 0 && S::neq == 0 && S::npair == S::npc + S::nbl && S::ls_iterations <= 10;
#endif
  }
  typedef unsigned long long lmask;
  static DEV lmask LT(float a, float b) { return __builtin_amdgcn_ballot_w64(a < b); }
  static DEV float SEL(lmask mk, float yes, float no) { return __builtin_amdgcn_inverse_ballot_w64(mk) ? yes : no; }
  template <int K> static DEV void fmac_rowb(float &acc, float x, float y) {  // acc += (lane K of the lane's 16-lane row of x) * y
    asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(x), "v"(y), "n"(K));
  }
  // max(lane K of the lane's 16-lane row of x, y) - one instruction (fmaxf adds a canonicalising v_max).  `dep`: an unused operand that
  // orders the instruction behind the compiler-padded DPP read that produced it (r05: as an operand of THIS instruction - carried through y
  // by an empty asm, the constant y was copied afresh for every use: three v_mov per bracketing iteration)
  template <int K> static DEV float max_rowb(float x, float y, float dep) {
    float o;
    asm("v_max_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(o) : "v"(x), "v"(y), "n"(K), "v"(dep));
    return o;
  }
#ifndef AMX_LS_CLAMP_MASK
#define AMX_LS_CLAMP_MASK 1  // 0: a row's activity at a trial point by compare and select (r04), A/B switch
#endif
#ifndef AMX_LS_PACKED_X
#define AMX_LS_PACKED_X 1  // 0: the two row slots' residuals at a trial point as two scalar multiply-adds (A/B switch)
#endif
  struct LSH { float alpha, hd, h; };
  DEV float linesearch_v2(bool alive) {
    const int nv = m.nv();
    const float mv = mul_M_row(D_SEARCH);
    aJv = row_dot_A(D_SEARCH);
    cJv = row_dot_B(D_SEARCH);
    if constexpr (NBS == 2) c2Jv = row_dot_B2(D_SEARCH);
    const bool validA = gl < nA(), validB = gl < nB();
    // (q1 / 2, q2) of the lane's two row slots, zero for a slot without a row; the lane's share of the Gauss terms
    const float tA = validA ? aJv * aD : 0.0f, tB = validB ? cJv * cD : 0.0f;
    const f2v pA = {0.5f * tA * aJaref, 0.5f * tA * aJv}, pB = {0.5f * tB * cJaref, 0.5f * tB * cJv};
    const float tB2 = validB2() ? c2Jv * c2D : 0.0f;  // (the second contact-row slot: zero coefficients without one)
    const f2v pB2 = {0.5f * tB2 * c2Jaref, 0.5f * tB2 * c2Jv};
    f2v gsh;
    float sn_;
    {
      const int li = gl < nv ? gl : 0;
      const float si = dof(D_SEARCH, li);
      const float g1h = 0.5f * (si * r_ma - si * r_fs), g2_ = 0.5f * si * mv;
      gsh = f2v{gl < nv ? g1h : 0.0f, gl < nv ? g2_ : 0.0f};
      sn_ = gl < nv ? si * si : 0.0f;
    }
    // q0 of the lane's rows at alpha = 0 (the acceptance test's constant terms)
    const float a0A = validA ? 0.5f * aD * aJaref * aJaref : 0.0f, a0B = validB ? 0.5f * cD * cJaref * cJaref : 0.0f;
    const float a0B2 = validB2() ? 0.5f * c2D * c2Jaref * c2Jaref : 0.0f;
    const f2v jv2 = {aJv, cJv}, jr2 = {aJaref, cJaref};  // both row slots' residual at a trial point in one packed multiply-add
#if AMX_LS_CLAMP_MASK
    // (r05c) "the row is active at alpha" as the 0 / 1 factor directly: residual x < 0  <=>  clamp(-2^64 x) = 1 for every |x| >= 2^-64, and the
    // scaling by a power of two commutes with the multiply-add's rounding, so one packed multiply-add with the clamp modifier on pre-scaled
    // (-2^64 Jv, -2^64 Jaref) yields both slots' factors: 1 instruction per trial point instead of 1 + 2 compares + 2 selects.  Exact zero,
    // -0, NaN (clamped to 0) and overflow (+-inf -> 1 / 0) decide as the compare does; a residual inside (-2^-64, 0) - a cancellation to twenty
    // digits - would weigh its row by |x| 2^64 < 1 at that one trial point instead of 1.
    const f2v jvS = jv2 * -0x1p64f, jrS = jr2 * -0x1p64f;
    const float c2JvS = c2Jv * -0x1p64f, c2JrS = c2Jaref * -0x1p64f;
    auto terms = [&](float alpha) {
      f2v a2, m2;
      a2.x = alpha;  // (a2.y is left undefined on purpose: op_sel_hi takes the low half twice, and a defined high half costs a v_mov per trial point)
      asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1] clamp" : "=v"(m2) : "v"(a2), "v"(jvS), "v"(jrS));
      f2v r = pk_fma(pA, f2v{m2.x, m2.x}, pk_fma(pB, f2v{m2.y, m2.y}, gsh));
      if constexpr (NBS == 2) {
        float mB2;
        asm("v_fma_f32 %0, %1, %2, %3 clamp" : "=v"(mB2) : "v"(alpha), "v"(c2JvS), "v"(c2JrS));
        r = pk_fma(pB2, f2v{mB2, mB2}, r);
      }
      return r;
    };
#else
    auto terms = [&](float alpha) {  // the lane's (q1 / 2, q2) at alpha, Gauss share included
#if AMX_LS_PACKED_X
      const f2v x2 = pk_fma(f2v{alpha, alpha}, jv2, jr2);
      const float xA = x2.x, xB = x2.y;
#else
      const float xA = __builtin_fmaf(alpha, aJv, aJaref), xB = __builtin_fmaf(alpha, cJv, cJaref);
#endif
      const float mA = xA < 0.0f ? 1.0f : 0.0f, mB = xB < 0.0f ? 1.0f : 0.0f;
      f2v r = pk_fma(pA, f2v{mA, mA}, pk_fma(pB, f2v{mB, mB}, gsh));
      if constexpr (NBS == 2) {
        const float mB2 = __builtin_fmaf(alpha, c2Jv, c2Jaref) < 0.0f ? 1.0f : 0.0f;
        r = pk_fma(pB2, f2v{mB2, mB2}, r);
      }
      return r;
    };
#endif
#if AMX_LS_CLAMP_MASK
    // the same factors weigh the rows' constant terms: 1 * a0 and 0 * a0 are exact and the multiply-add rounds the sum once, as the addition of the
    // two selected terms did - the same bits
    auto q0at = [&](float alpha) {
      f2v a2, m2;
      a2.x = alpha;
      asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1] clamp" : "=v"(m2) : "v"(a2), "v"(jvS), "v"(jrS));
      float r = __builtin_fmaf(m2.x, a0A, m2.y * a0B);
      if constexpr (NBS == 2) {
        float mB2;
        asm("v_fma_f32 %0, %1, %2, %3 clamp" : "=v"(mB2) : "v"(alpha), "v"(c2JvS), "v"(c2JrS));
        r = __builtin_fmaf(mB2, a0B2, r);
      }
      return r;
    };
#else
    auto q0at = [&](float alpha) {
      const float xA = __builtin_fmaf(alpha, aJv, aJaref), xB = __builtin_fmaf(alpha, cJv, cJaref);
      float r = (xA < 0.0f ? a0A : 0.0f) + (xB < 0.0f ? a0B : 0.0f);
      if constexpr (NBS == 2) r += __builtin_fmaf(alpha, c2Jv, c2Jaref) < 0.0f ? a0B2 : 0.0f;
      return r;
    };
#endif
    const float hmin = 0.5f * MINVAL;
    float hmin_r = hmin;
    asm volatile("" : "+v"(hmin_r));  // one register for the whole search (rematerialised per use it is a v_mov each time)
    const float q0_start = q0at(0.0f);
    LSH p0;
    float sn;
    {
      const f2v t = terms(0.0f);
      const float u = gsum_pack4<G>(t.x, t.y, sn_, 0.0f);
      sn = quad_value<3>(u);
      const float hd0 = quad_value<0>(u);  // (compiler-visible DPP read of u first: it carries the wait states the assembly reads below rely on)
      p0 = LSH{0.0f, hd0, max_rowb<4>(u, hmin_r, hd0)};
    }
    PROF(21);
    const float hgtol = 0.5f * (m.tolerance() * m.ls_tolerance() * __builtin_amdgcn_sqrtf(sn) * m.scale());
    LSH p1;
    {
      const float a1 = __builtin_fmaf(-p0.hd, __builtin_amdgcn_rcpf(p0.h), p0.alpha);
      const f2v t = terms(a1);
      const float r = gsum_pack2<G>(t.x, t.y);
      float hd = quad_value<0>(r);
      fmac_rowb<4>(hd, r, a1);
      p1 = LSH{a1, hd, max_rowb<4>(r, hmin_r, hd)};
    }
    const lmask first = LT(p1.hd, p0.hd);
    LSH lo = {SEL(first, p1.alpha, p0.alpha), SEL(first, p1.hd, p0.hd), SEL(first, p1.h, p0.h)};
    LSH hi = {SEL(first, p0.alpha, p1.alpha), SEL(first, p0.hd, p1.hd), SEL(first, p0.h, p1.h)};
    const lmask alivem = __builtin_amdgcn_ballot_w64(alive);
    lmask swapm = ~0ull;
    auto bracket_step = [&]() {
      const lmask dm = ~swapm | (LT(lo.hd, 0.0f) & LT(-hgtol, lo.hd)) | (LT(0.0f, hi.hd) & LT(hi.hd, hgtol));
      const lmask nd = ~dm;
      if ((nd & alivem) == 0ull) return false;
      const float a_lo = __builtin_fmaf(-lo.hd, __builtin_amdgcn_rcpf(lo.h), lo.alpha), a_hi = __builtin_fmaf(-hi.hd, __builtin_amdgcn_rcpf(hi.h), hi.alpha);
      const float a_mid = 0.5f * (lo.alpha + hi.alpha);
      const f2v t_lo = terms(a_lo), t_hi = terms(a_hi), t_mid = terms(a_mid);
      float u, w;
      gsum_pack6<G>(t_lo.x, t_lo.y, t_hi.x, t_hi.y, t_mid.x, t_mid.y, u, w);  // u: quads 0..3 = lo q1/2, lo q2, hi q2, hi q1/2; w: quads 0, 1 = mid q1/2, mid q2
      float hd_lo = quad_value<0>(u), hd_hi = quad_value<3>(u), hd_mid = quad_value<0>(w);
      fmac_rowb<4>(hd_lo, u, a_lo);
      fmac_rowb<8>(hd_hi, u, a_hi);
      fmac_rowb<4>(hd_mid, w, a_mid);
      // (ordered behind the compiler-visible DPP reads of u and w through their results: the two wait states after the lane swaps)
      const float h_lo = max_rowb<4>(u, hmin_r, hd_lo), h_hi = max_rowb<8>(u, hmin_r, hd_hi), h_mid = max_rowb<4>(w, hmin_r, hd_mid);
      const lmask s_lo_next = nd & (LT(0.0f, lo.hd) | LT(lo.hd, hd_lo));
      LSH nlo = {SEL(s_lo_next, a_lo, lo.alpha), SEL(s_lo_next, hd_lo, lo.hd), SEL(s_lo_next, h_lo, lo.h)};
      const lmask s_lo_mid = nd & LT(hd_mid, 0.0f) & LT(nlo.hd, hd_mid);
      nlo = LSH{SEL(s_lo_mid, a_mid, nlo.alpha), SEL(s_lo_mid, hd_mid, nlo.hd), SEL(s_lo_mid, h_mid, nlo.h)};
      const lmask s_hi_next = nd & (LT(hi.hd, 0.0f) | LT(hd_hi, hi.hd));
      LSH nhi = {SEL(s_hi_next, a_hi, hi.alpha), SEL(s_hi_next, hd_hi, hi.hd), SEL(s_hi_next, h_hi, hi.h)};
      const lmask s_hi_mid = nd & LT(0.0f, hd_mid) & LT(hd_mid, nhi.hd);
      nhi = LSH{SEL(s_hi_mid, a_mid, nhi.alpha), SEL(s_hi_mid, hd_mid, nhi.hd), SEL(s_hi_mid, h_mid, nhi.h)};
      lo = nlo; hi = nhi;
      swapm = s_lo_next | s_lo_mid | s_hi_next | s_hi_mid;
      return true;
    };
#pragma unroll
    for (int it = 0; it < S::ls_iterations; ++it) if (!bracket_step()) break;
    PROF(22);
    // costs of the two bracket ends relative to the starting point (see linesearch): 2 alpha (hd - alpha h / 2) is alpha d0 - alpha^2 d1 / 2
    const float r = gsum_pack2<G>(q0at(lo.alpha) - q0_start, q0at(hi.alpha) - q0_start);
    const float lo_delta = 2.0f * lo.alpha * (lo.hd - 0.5f * lo.alpha * lo.h) + quad_value<0>(r);
    const float hi_delta = 2.0f * hi.alpha * (hi.hd - 0.5f * hi.alpha * hi.h) + quad_value<1>(r);
    const bool improved = alive & ((lo_delta < 0.0f) | (hi_delta < 0.0f));
    const float alpha = lo_delta < hi_delta ? lo.alpha : hi.alpha;
    if (improved) {
      if (gl < nv) { r_ma += alpha * mv; dof(D_QACC, gl) += alpha * dof(D_SEARCH, gl); dof(D_MA, gl) = r_ma; }
      aJaref += alpha * aJv;
      cJaref += alpha * cJv;
      if constexpr (NBS == 2) c2Jaref += alpha * c2Jv;
    }
    WSYNC();
    return improved ? -fminf(lo_delta, hi_delta) : 0.0f;
  }

  DEV void solve() {
    const int nv = m.nv();
    float gauss, cost;
    bool use_warm = false;
    float ma_warm = 0, ja_warm = 0, jc_warm = 0, jc2_warm = 0;
    if (!(m.disableflags() & DIS_WARMSTART)) {
      float gw, gs;
      init_context(D_WARM);
      const float cw = constraint_cost(D_WARM, &gw);
      ma_warm = r_ma;
      ja_warm = aJaref;
      jc_warm = cJaref;
      if constexpr (NBS == 2) jc2_warm = c2Jaref;
      WSYNC();
      init_context(D_ASMOOTH);
      const float cs = constraint_cost(D_ASMOOTH, &gs);
      use_warm = cw < cs;
      cost = use_warm ? cw : cs;  // the chosen start's cost and Gauss term are the ones just evaluated
      gauss = use_warm ? gw : gs;
    }
    if (gl < nv) {
      { const float w_ = dof(D_WARM, gl), s_ = dof(D_ASMOOTH, gl); dof(D_QACC, gl) = use_warm ? w_ : s_; }
      if (use_warm) dof(D_MA, gl) = ma_warm;  // context of the chosen start: M*qacc and J*qacc - aref
    }
    r_ma = use_warm ? ma_warm : r_ma;
    if (use_warm) { aJaref = ja_warm; cJaref = jc_warm; if constexpr (NBS == 2) c2Jaref = jc2_warm; }
    WSYNC();
    if (m.disableflags() & DIS_WARMSTART) {
      init_context(D_QACC);
      cost = constraint_cost(D_QACC, &gauss);
    }
    // cost decrease of the last accepted step.  The reference forms it as prev_cost - cost from two full evaluations of the cost;
    // the line search knows it as the value of its quadratic model at the accepted step length (the cost along the search line
    // IS that piecewise quadratic), without the cancellation of two sums of 1e3 .. 1e5 - and the full evaluation per iteration
    // is not needed at all (see linesearch: acceptance test)
    float improvement = INFINITY;
    PROF(9);
    constraint_force_and_grad(m.solver() == 2);
    PROF(10);
    if (m.solver() == 2) newton_direction();
    else chol_solve<G>(m.nchol(), eb + m.e_M(), m.ldm(), nv, eb + m.e_Lb(), dofoff(D_GRAD), dofoff(D_MGRAD), false, gl, gb4);
    if constexpr (!search_from_solve()) {
      if (gl < nv) dof(D_SEARCH, gl) = -dof(D_MGRAD, gl);
      WSYNC();
    }
    PROF(11);
    bool alive = true;  // per group: still iterating
    bool stop = false;  // wave-uniform: the stopping test evaluated inside newton_direction ended the solve
    bool next_done = false, have_next = false;  // the stopping test of the coming iteration, when the fused solve has already made it
    for (int niter = 0; niter < m.iterations(); ++niter) {
      if (stop) break;
      if (m.iterations() != 1) {
        bool done;
        if (grad_in_solve() && have_next) {
          done = next_done;
        } else {
          const float g_ = dof(D_GRAD, gl < nv ? gl : 0);
          const float g = gl < nv ? g_ : 0.0f;
          const float gn = gsum<G>(g * g);
          const float ts_ = m.scale() * m.tolerance();  // the reference's improvement / scale < tol and |grad| / scale < tol, multiplied out
          done = (improvement < ts_) | (gn < ts_ * ts_);
        }
        alive = alive & !done;
      }
      if (!wave_any(alive)) break;
      PROF(12);
      float dec;
      static_assert(NBS == 1 || (S::is_static && S::solver == 2 && G >= 32), "two contact-row slots: per-model Newton kernels on 32- / 64-lane groups");
      if constexpr (ls_v2_ok()) dec = linesearch_v2(alive);
      else dec = linesearch(alive);
      improvement = alive ? dec : improvement;
      PROF(13);
      float pg = 0;  // CG: keep grad.Mgrad of the previous iterate
      if (m.solver() != 2) {
        if (gl < nv) { pg = dof(D_GRAD, gl) * dof(D_MGRAD, gl); lds[eb + m.e_H() + gl] = dof(D_MGRAD, gl); }  // CG scratch: the H buffer is idle
        pg = gsum<G>(pg);
      }
      // a finished group's state is frozen, so recomputing its force / gradient below is idempotent
      constraint_force_and_grad(m.solver() == 2 && niter + 1 < m.iterations(), alive);
      PROF(10);
      if (m.solver() == 2) {
        // the search direction of an iteration that will not run is never read: skip its factorisation
        if (niter + 1 < m.iterations()) {
          // the next iteration's stopping test needs this point's improvement (known) and gradient (formed inside the fused
          // solve): evaluated there, it saves the elimination when the wave is finished
          const bool imp_done = improvement < m.scale() * m.tolerance();
          stop = newton_direction(alive, m.iterations() != 1, imp_done, &next_done);
          have_next = m.iterations() != 1;
          PROF(11);
          if constexpr (!search_from_solve()) { if (gl < nv) dof(D_SEARCH, gl) = -dof(D_MGRAD, gl); }
        }
      } else {
        chol_solve<G>(m.nchol(), eb + m.e_M(), m.ldm(), nv, eb + m.e_Lb(), dofoff(D_GRAD), dofoff(D_MGRAD), false, gl, gb4);
        float num = gl < nv ? dof(D_GRAD, gl) * (dof(D_MGRAD, gl) - lds[eb + m.e_H() + gl]) : 0.0f;
        num = gsum<G>(num);
        const float beta = fmaxf(0.0f, num / fmaxf(MINVAL, pg));
        if (alive && gl < nv) dof(D_SEARCH, gl) = -dof(D_MGRAD, gl) + beta * dof(D_SEARCH, gl);
      }
      WSYNC();
    }
  }

  // ------------------------------------------------------------------ forward (A9.1-A9.13)
  // `after_poses` runs once the body poses are final and before their storage is reused (mjx.forward's xpos / xquat outputs).
  // Phase order: the collision pass precedes crb + M and the constraint rows are assembled last, after the smooth forces -
  // the body records (poses, cinert / cfrc, crb / cvel / cacc) are dead by then and the contact Jacobian rows, the per-row
  // records and the M*v staging vector take over their space (amx_plan.h, region R1).
  DEV void forward() { forward([] {}); }
  template <class Hook> DEV void forward(Hook after_poses) {
    const int nv = m.nv();
    PROF(0);
    phase_prio<0>();
    kinematics();
    PROF(1);
    com_pos();
    PROF(2);
    if (m.ncon() > 0 && !(m.disableflags() & DIS_CONSTRAINT) && (!(m.disableflags() & DIS_CONTACT) || m.npair() > m.npc())) collision();  // contacts, and the connect / weld residuals
    PROF(4);
    crb_and_M();
    PROF(3);
    velocity_and_bias();
    PROF(6);
    smooth_forces();
    PROF(7);
    after_poses();
    if (m.nefc() > 0) make_constraint();
    PROF(5);
    // qacc_smooth = M^-1 qfrc_smooth.  Newton never reuses the factor: one-shot Gauss-Jordan solve;
    // CG keeps a Cholesky factor in Lb as its preconditioner.
    if (m.solver() == 2) {
      if constexpr (S::is_static) spd_solve_static<S, G, true>(eb + m.e_M(), dofoff(D_FSMOOTH), dofoff(D_ASMOOTH), gl, m_rot, eb + m.e_H(), m_diag, m_win);
      else spd_solve<G>(m.nchol(), eb + m.e_M(), m.ldm(), nv, dofoff(D_FSMOOTH), dofoff(D_ASMOOTH), gl, gb4, m_rot);
    }
    else chol_solve<G>(m.nchol(), eb + m.e_M(), m.ldm(), nv, eb + m.e_Lb(), dofoff(D_FSMOOTH), dofoff(D_ASMOOTH), true, gl, gb4);
    { const float t_ = dof(D_ASMOOTH, gl < nv ? gl : 0); r_as = gl < nv ? t_ : 0.0f; }
    PROF(8);
    if (m.nefc() == 0) {
      if (gl < nv) { dof(D_QACC, gl) = dof(D_ASMOOTH, gl); dof(D_FCON, gl) = 0; }
      WSYNC();
      return;
    }
    phase_prio<2>();
    solve();
    if (gl < nv) dof(D_WARM, gl) = dof(D_QACC, gl);
    WSYNC();
    PROF(15);
  }

  // ------------------------------------------------------------------ A9.14 semi-implicit Euler
  DEV void euler() {
    const int nv = m.nv(), Mo = eb + m.e_M(), Ho = eb + m.e_H();
    const int qp = eb + m.e_qpos(), qv = eb + m.e_qvel();
    if (!(m.disableflags() & DIS_EULERDAMP)) {
      // M + dt*diag(damping) in place: this is the last reader of M in the step and the crb pass rewrites it
      if (gl < nv) { dof(D_SEARCH, gl) = dof(D_FSMOOTH, gl) + dof(D_FCON, gl); lds[Mo + m_diag] += m.dt() * CF(m.c_dof() + gl * DOF_S + 5); }
      WSYNC();
      if constexpr (S::is_static) spd_solve_static<S, G, true>(Mo, dofoff(D_SEARCH), dofoff(D_MV), gl, m_rot, eb + m.e_H(), m_diag, m_win);
      else spd_solve<G>(m.nchol(), Mo, m.ldm(), nv, dofoff(D_SEARCH), dofoff(D_MV), gl, gb4, m_rot);
    } else {
      if (gl < nv) dof(D_MV, gl) = dof(D_QACC, gl);
      WSYNC();
    }
    PROF(16);
    if (gl < nv) lds[qv + gl] += dof(D_MV, gl) * m.dt();
    WSYNC();
    if (gl < m.njnt()) {
      const int jr = m.c_jnt() + gl * JNT_S, t_ = CI(jr), qa = qp + CI(jr + 1), da = qv + CI(jr + 2);
      if (t_ == JFREE || t_ == JBALL) {
        int qq = qa, dd = da;
        if (t_ == JFREE) {
          lds[qa] += lds[da] * m.dt(); lds[qa + 1] += lds[da + 1] * m.dt(); lds[qa + 2] += lds[da + 2] * m.dt();
          qq = qa + 3; dd = da + 3;
        }
        float n;
        const f3 w = normalize3(mk3(lds[dd], lds[dd + 1], lds[dd + 2]), &n);
        f4 q = {lds[qq], lds[qq + 1], lds[qq + 2], lds[qq + 3]};
        q = qnormalize(qmul(q, axisangle(w, m.dt() * n)));
        lds[qq] = q.w; lds[qq + 1] = q.x; lds[qq + 2] = q.y; lds[qq + 3] = q.z;
      } else {
        lds[qa] += lds[da] * m.dt();
      }
    }
    WSYNC();
    PROF(17);
  }
};

// =====================================================================================================
// Kernels: AMX_WAVES_PER_BLOCK wavefronts per workgroup, 64/G environments per wavefront.  One kernel per
// entry point of the C ABI so each carries only its own arguments in SGPRs.
// The by-value model descriptor arrives as a few s_load_dwordx16 tuples.  It does not fit the SGPR file next to the
// exec-mask temporaries, and a spilled TUPLE is reloaded whole (16 v_readlane for one field) at every use inside the
// hot loops.  Copying each word through its own s_mov gives the allocator independent 32-bit values: one v_readlane
// per reload.
DEV DevModel split_scalars(const DevModel &in) {
  constexpr int NW = sizeof(DevModel) / 4;
  static_assert(sizeof(DevModel) % 4 == 0, "DevModel is a sequence of 32-bit words");
  int w[NW];
  __builtin_memcpy(w, &in, sizeof(DevModel));
#pragma unroll
  for (int k = 0; k < NW; ++k) asm volatile("s_mov_b32 %0, %1" : "=s"(w[k]) : "s"(w[k]));
  DevModel out;
  __builtin_memcpy(&out, w, sizeof(DevModel));
  return out;
}

// =====================================================================================================
#define AMX_BLOCK (AMX_WAVE * AMX_WAVES_PER_BLOCK)  // largest workgroup; the host launches 1..AMX_WAVES_PER_BLOCK waves per workgroup
#define AMX_KERNEL_PROLOGUE()                                                                               \
  const DevModel md_ = S::is_static ? m_in : split_scalars(m_in); /* a static spec reads a handful of words: no need */ \
  const MV<S> m{md_};                                                                                        \
  const int lane = threadIdx.x & (AMX_WAVE - 1), wave = threadIdx.x / AMX_WAVE;                              \
  const int gl = lane & (G - 1), grp = lane / G, gb4 = 4 * (lane & ~(G - 1));                                \
  const int slot = wave * (AMX_WAVE / G) + grp; /* environment slot inside the workgroup */                 \
  const int wpb = blockDim.x / AMX_WAVE; /* waves in this workgroup, chosen by the host to fill the CU's LDS */ \
  const int e = blockIdx.x * (wpb * (AMX_WAVE / G)) + slot;                                  \
  /* stage the model constants once per workgroup (coalesced, shared by its waves) */                       \
  for (int i = threadIdx.x; i < m.nconst(); i += blockDim.x) lds[i] = m.image()[i];                               \
  __syncthreads(); /* the only workgroup barrier; the waves are independent from here on */                 \
  if (blockIdx.x * (wpb * (AMX_WAVE / G)) + wave * (AMX_WAVE / G) >= nenv) return; /* whole wave idle */ \
  const bool live = e < nenv; /* a trailing group without an environment still runs, on a zero slab, and never touches HBM */ \
  const int eb = m.nconst() + slot * m.env_floats();                                                             \
  const int nq = m.nq(), nv = m.nv(), nu = m.nu(), nx = nq + nv;                                                   \
  /* zero the slab once: J, M and H have static sparsity and are only written on their support */           \
  for (int i = gl; i < m.env_floats(); i += G) lds[eb + i] = 0.0f;                                             \
  if (!live && gl < nq) lds[eb + m.e_qpos() + gl] = CF(m.c_qpos0() + gl); /* keep the idle group's state finite */ \
  WSYNC();                                                                                                   \
  Env<G, S> env(m, eb, gl, gb4);                                                                               \
  const int qp = eb + m.e_qpos(), qv = eb + m.e_qvel();

// vmap(shoot): state stays in LDS for the whole horizon; HBM sees one control row in and one state row out per step.
// MINW: waves per SIMD the register budget is sized for.  2 (256 VGPRs) is what a 4096-environment batch can fill; the
// specialised kernels are also built for 3 (168 VGPRs, a handful of spills), which large batches of small-slab models use:
// quadruped, 32768 environments, 121 -> 161 M env-steps/s (r02).
template <int G, class S, int MINW = AMX_MIN_WAVES_PER_SIMD>
__global__ void __launch_bounds__(AMX_BLOCK, MINW)
amx_rollout_kernel(const DevModel m_in, int nenv, int nsteps, int x0_batched, const float *__restrict__ x0g, const float *__restrict__ usg,
                   float *__restrict__ xs, float *__restrict__ xfinal, float *__restrict__ prof) {
  AMX_KERNEL_PROLOGUE()
#ifdef AMX_PROFILE
  env.prof_t = (unsigned)__builtin_amdgcn_s_memtime();
  env.prof_row = (prof && live) ? prof + (size_t)e * PROF_N : nullptr;
#endif
#if defined(AMX_STAGGER) && AMX_STAGGER > 0
  // A/B experiment: the two waves of a SIMD (waves w and w + 4 of the workgroup) start half a step apart, so that one's
  // latency-bound phases meet the other's issue-bound ones
  if (wave >= 4) for (int i_ = 0; i_ < AMX_STAGGER; ++i_) __builtin_amdgcn_s_sleep(127);
#endif
  const float *x0 = x0g + ((x0_batched && live) ? (size_t)e * nx : 0);
  if (live) {
    for (int i = gl; i < nx; i += G) {
      const float v = x0[i];
      if (i < nq) lds[qp + i] = v; else lds[qv + i - nq] = v;
      if (xs) xs[(size_t)e * (nsteps + 1) * nx + i] = v;
    }
  }
  const float *us = usg + (live ? (size_t)e * nsteps * nu : 0);
  // control row prefetched one step ahead (rows wider than a group are read in place)
  float unext = (live && gl < nu && nsteps > 0) ? us[gl] : 0.0f;
  const float *urow = us;                                                  // control row of the current step
  float *xrow = xs ? xs + (live ? (size_t)e * (nsteps + 1) * nx : 0) : xs;  // state row written last
  WSYNC();
  // t = -1 is shoot()'s forward before the scan (ctrl = 0): it only seeds qacc_warmstart
#ifndef AMX_FAIR_SIMD
#define AMX_FAIR_SIMD 1  // 0: the two waves of a SIMD are left to the hardware's arbitration (A/B switch)
#endif
  // Waves w and w + 4 of an 8-wave workgroup share a SIMD.  With equal priorities its arbiter favours the older one on every
  // contested issue slot: measured (r03, per-wave phase clocks) the older four waves of every workgroup ran a biped step in
  // 81 k cycles and the younger four in 99 k, and a launch ends with its slowest wave.  Each wave posts its step counter in
  // LDS; whichever of the pair is behind runs at raised priority for the step, so both advance at the same pace (~90 k).
  // That waves w and w + 4 sit on one SIMD is how the hardware has dealt a workgroup's waves on every box measured, not a
  // contract: each wave posts the SIMD it runs on (HW_ID bits 5:4, + 1) in the top byte of its counter word, and a wave whose
  // presumed partner reports another SIMD never raises its priority - the pair falls back to the hardware's own arbitration
  // (amx_model_query "simd_pairing" runs the same check as a probe launch and reports it).
  const bool paired = AMX_FAIR_SIMD && wpb == 8;
  const int simd_tag = (int)((__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4) + 1) << 24);  // hwreg(HW_REG_HW_ID, 4, 2) = SIMD_ID
  for (int t = -1; t < nsteps; ++t) {
    if (paired) {
      const int mine_ = t + 1;
      if (lane == 0) lds[m.c_prog() + wave] = __int_as_float(simd_tag | mine_);
      const int other_ = __builtin_amdgcn_readfirstlane(CI(m.c_prog() + (wave ^ 4)));
      if (AMX_PHASE_PRIO) env.prio_lag = (((other_ ^ simd_tag) >> 24) == 0 && (other_ & 0xFFFFFF) > mine_) ? 1 : 0;
      else if (((other_ ^ simd_tag) >> 24) == 0 && (other_ & 0xFFFFFF) > mine_) __builtin_amdgcn_s_setprio(3);
      else __builtin_amdgcn_s_setprio(0);
    }
    if (t >= 0) {
      if (gl < nu) lds[eb + m.e_ctrl() + gl] = unext;
      if (live) for (int i = gl + G; i < nu; i += G) lds[eb + m.e_ctrl() + i] = urow[i];
      urow += nu;  // row t + 1 (a running pointer: the 64-bit row products cost a dozen vector instructions per step)
      if (live && t + 1 < nsteps && gl < nu) unext = urow[gl];
      WSYNC();
    }
    env.forward();
    if (t >= 0) {
      env.euler();
      if (xs && live) {
        xrow += nx;
        for (int i = gl; i < nx; i += G) xrow[i] = i < nq ? lds[qp + i] : lds[qv + i - nq];
      }
    }
#ifdef AMX_PROFILE
    { unsigned t_ = (unsigned)__builtin_amdgcn_s_memtime(); if (gl == 0 && env.prof_row) unsafeAtomicAdd(env.prof_row + 18, (float)(t_ - env.prof_t)); env.prof_t = t_; }
#endif
  }
  if (xfinal && live)
    for (int i = gl; i < nx; i += G) xfinal[(size_t)e * nx + i] = i < nq ? lds[qp + i] : lds[qv + i - nq];
}

// MjxEnv.pipeline_step: nsub physics steps with ctrl held; state comes from and returns to HBM.
// MINW: waves per SIMD the register budget is sized for.  2 (256 VGPRs) is what a 4096-environment batch can fill; the
// specialised kernels are also built for 3 (168 VGPRs, a handful of spills), which large batches of small-slab models use:
// quadruped, 32768 environments, 121 -> 161 M env-steps/s (r02).
template <int G, class S, int MINW = AMX_MIN_WAVES_PER_SIMD>
__global__ void __launch_bounds__(AMX_BLOCK, MINW)
amx_step_kernel(const DevModel m_in, int nenv, int nsub, const float *qpos, const float *qvel, const float *__restrict__ ctrl,
                const float *warm, const float *time, float *qpos_out, float *qvel_out, float *warm_out, float *time_out) {
  // outputs may alias their inputs (amx_step): every input of an environment is read before any of its outputs is written
  AMX_KERNEL_PROLOGUE()
  float t0 = 0.0f;
  if (live) {
    for (int i = gl; i < nq; i += G) lds[qp + i] = qpos[(size_t)e * nq + i];
    if (gl < nv) {
      lds[qv + gl] = qvel[(size_t)e * nv + gl];
      env.dof(Env<G, S>::D_WARM, gl) = warm ? warm[(size_t)e * nv + gl] : 0.0f;
    }
    for (int i = gl; i < nu; i += G) lds[eb + m.e_ctrl() + i] = ctrl[(size_t)e * nu + i];
    if (time && gl == 0) t0 = time[e];
  }
  WSYNC();
  for (int t = 0; t < nsub; ++t) { env.forward(); env.euler(); }
  if (!live) return;
  for (int i = gl; i < nq; i += G) qpos_out[(size_t)e * nq + i] = lds[qp + i];
  if (gl < nv) {
    qvel_out[(size_t)e * nv + gl] = lds[qv + gl];
    if (warm_out) warm_out[(size_t)e * nv + gl] = env.dof(Env<G, S>::D_WARM, gl);
  }
  if (time_out && gl == 0) time_out[e] = t0 + m.dt() * nsub;
}

// The tail of a wrapped env.step (amx_env_step, include/amx.h): one thread per environment, everything after the physics.
// The physics kernel has written qpos / qvel / time and the solver's qacc (= the next warm start) into the *_out buffers; this
// kernel reads them, forms observation and reward, advances the episode counters and overwrites the state of a finished
// environment with its first state.
extern "C" __global__ void __launch_bounds__(256) amx_env_tail_kernel(int nenv, int nq, int nv, int nu, const float *__restrict__ action,
                                                                      const amx_env_tail_desc d, const amx_env_tail_io io) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e == 0 && io.step_counter_out) *io.step_counter_out = (io.step_counter_in ? *io.step_counter_in : 0.0f) + 1.0f;
  if (e >= nenv) return;
  float *qp = io.qpos_out + (size_t)e * nq, *qv = io.qvel_out + (size_t)e * nv;
  const float *u = action + (size_t)e * nu;
  float obs[AMX_TAIL_MAX_OBS];
#pragma unroll
  for (int k = 0; k < AMX_TAIL_MAX_OBS; ++k) {
    if (k < d.nobs) {
      const int kind = d.obs_kind[k], i = d.obs_index[k];
      const float x = kind == 3 ? qv[i] : qp[i];
      obs[k] = kind == 1 ? cosf(x) : (kind == 2 ? sinf(x) : x);
    }
  }
  float reward = 0.0f;
#pragma unroll
  for (int k = 0; k < AMX_TAIL_MAX_REW; ++k) {
    if (k < d.nrew) {
      const int kind = d.rew_kind[k], i = d.rew_index[k];
      float t = kind == 2 ? qv[i] : (kind == 3 ? u[i] : qp[i] - d.rew_target[k]);
      if (kind == 0) t = atan2f(sinf(t), cosf(t));
      reward -= d.rew_weight[k] * t * t;
    }
  }
  const float env_done = 0.0f;  // (the reference env never terminates: swingup.py:115)
  const float steps = io.steps[e] + 1.0f;
  const float trunc = steps >= (float)d.episode_length ? 1.0f : 0.0f;
  const float done = fmaxf(env_done, trunc);
  const bool fin = done > 0.0f;
  const float ret = io.episode_return[e] + reward;
  io.last_episode_return_out[e] = fin ? ret : io.last_episode_return[e];
  io.episodes_out[e] = io.episodes[e] + done;
  io.episode_return_out[e] = fin ? 0.0f : ret;
  io.truncation_out[e] = trunc * (1.0f - env_done);
  io.steps_out[e] = fin ? 0.0f : steps;
  io.reward_out[e] = reward;
  io.done_out[e] = done;
  for (int k = 0; k < d.nobs; ++k) {
    io.terminal_obs_out[(size_t)e * d.nobs + k] = obs[k];
    io.obs_out[(size_t)e * d.nobs + k] = fin ? io.first_obs[(size_t)e * d.nobs + k] : obs[k];
  }
  for (int i = 0; i < nu; ++i) io.ctrl_out[(size_t)e * nu + i] = fin ? io.first_ctrl[(size_t)e * nu + i] : u[i];
  for (int i = 0; i < nv; ++i) io.warm_out[(size_t)e * nv + i] = fin ? io.first_warm[(size_t)e * nv + i] : io.qacc_out[(size_t)e * nv + i];
  if (fin) {
    for (int i = 0; i < nq; ++i) qp[i] = io.first_qpos[(size_t)e * nq + i];
    for (int i = 0; i < nv; ++i) qv[i] = io.first_qvel[(size_t)e * nv + i];
    io.time_out[e] = io.first_time[e];
  }
}

// Which SIMD every wave of an 8-wave workgroup runs on (amx_model_query "simd_pairing": the fair-sharing logic of the rollout
// kernel presumes waves w and w + 4 share one).  out[block * 8 + wave] = SIMD_ID.
extern "C" __global__ void __launch_bounds__(512) amx_simd_probe_kernel(int *out) {
  const int wave = threadIdx.x / AMX_WAVE;
  if ((threadIdx.x & (AMX_WAVE - 1)) == 0) out[blockIdx.x * 8 + wave] = (int)__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4);
}

struct ForwardOut { float *qacc, *qacc_smooth, *qfrc_bias, *qfrc_constraint, *qM, *xpos, *xquat, *cdist; int *nefc; };

// mjx.forward: derived quantities for a given state, no integration.
// MINW: waves per SIMD the register budget is sized for.  2 (256 VGPRs) is what a 4096-environment batch can fill; the
// specialised kernels are also built for 3 (168 VGPRs, a handful of spills), which large batches of small-slab models use:
// quadruped, 32768 environments, 121 -> 161 M env-steps/s (r02).
template <int G, class S, int MINW = AMX_MIN_WAVES_PER_SIMD>
__global__ void __launch_bounds__(AMX_BLOCK, MINW)
amx_forward_kernel(const DevModel m_in, int nenv, const float *__restrict__ qpos, const float *__restrict__ qvel, const float *__restrict__ ctrl,
                   const float *__restrict__ warm, const ForwardOut o) {
  AMX_KERNEL_PROLOGUE()
  if (live) {
    for (int i = gl; i < nq; i += G) lds[qp + i] = qpos[(size_t)e * nq + i];
    if (gl < nv) {
      lds[qv + gl] = qvel[(size_t)e * nv + gl];
      env.dof(Env<G, S>::D_WARM, gl) = warm ? warm[(size_t)e * nv + gl] : 0.0f;
    }
    for (int i = gl; i < nu; i += G) lds[eb + m.e_ctrl() + i] = ctrl[(size_t)e * nu + i];
  }
  WSYNC();
  env.forward([&] {
    if (live && o.xpos)
      for (int k = gl; k < m.nb() * 3; k += G) o.xpos[(size_t)e * m.nb() * 3 + k] = env.body(k % 3, k / 3);
    if (live && o.xquat)
      for (int k = gl; k < m.nb() * 4; k += G) o.xquat[(size_t)e * m.nb() * 4 + k] = env.body(3 + k % 4, k / 4);
    // (the contact distances too: in the planner's dense form the solver's row records take their place afterwards)
    if (live && o.cdist)
      for (int k = gl; k < m.ncon(); k += G) o.cdist[(size_t)e * m.ncon() + k] = env.con(0, k);
    WSYNC();
  });
  const unsigned long long gmask = G == 64 ? ~0ull : ((1ull << (G & 63)) - 1);
  const int nactive = __popcll((__ballot(gl < env.nA() && env.aD > 0) >> (lane & ~(G - 1))) & gmask) +
                      __popcll((__ballot(gl < env.nB() && env.cD > 0) >> (lane & ~(G - 1))) & gmask) +
                      __popcll((__ballot(env.validB2() && env.c2D > 0) >> (lane & ~(G - 1))) & gmask);
  if (!live) return;
  if (gl < nv) {
    if (o.qacc) o.qacc[(size_t)e * nv + gl] = env.dof(Env<G, S>::D_QACC, gl);
    if (o.qacc_smooth) o.qacc_smooth[(size_t)e * nv + gl] = env.dof(Env<G, S>::D_ASMOOTH, gl);
    if (o.qfrc_bias) o.qfrc_bias[(size_t)e * nv + gl] = env.dof(Env<G, S>::D_FBIAS, gl);
    if (o.qfrc_constraint) o.qfrc_constraint[(size_t)e * nv + gl] = env.dof(Env<G, S>::D_FCON, gl);
  }
  if (o.qM)
    for (int k = gl; k < nv * nv; k += G) {
      const int r = k / nv, c = k % nv, hi = r > c ? r : c, lo = r > c ? c : r;
      const int mo = env.moff(hi, lo);
      o.qM[(size_t)e * nv * nv + k] = mo < 0 ? 0.0f : lds[eb + m.e_M() + mo];
    }
  if (o.nefc && gl == 0) o.nefc[e] = nactive;
}

// Batched StaticGoalQuadraticCost.cost (reference ambersim/trajopt/cost.py:62-85), general sizes: one wave per rollout,
// lanes over the matrix entries.  Used when nx or nu exceed the tiled kernel's limits.
extern "C" __global__ void __launch_bounds__(AMX_WAVE) amx_quadcost_kernel(int nenv, int nsteps, int nx, int nu, const float *xs,
                                                                           const float *us, const float *Q, const float *Qf,
                                                                           const float *R, const float *xg, float *cost) {
  const int lane = threadIdx.x, e = blockIdx.x;
  if (e >= nenv) return;
  float acc = 0;
  for (int t = 0; t <= nsteps; ++t) {
    const float *x = xs + ((size_t)e * (nsteps + 1) + t) * nx;
    const float *W = t == nsteps ? Qf : Q;
    for (int k = lane; k < nx * nx; k += AMX_WAVE) {
      const int i = k / nx, j = k - i * nx;
      acc += (x[i] - xg[i]) * W[k] * (x[j] - xg[j]);
    }
  }
  for (int t = 0; t < nsteps; ++t) {
    const float *u = us + ((size_t)e * nsteps + t) * nu;
    for (int k = lane; k < nu * nu; k += AMX_WAVE) {
      const int i = k / nu, j = k - i * nu;
      acc += u[i] * R[k] * u[j];
    }
  }
  acc = wave_sum(acc);
  if (lane == 0) cost[e] = 0.5f * acc;
}

// The same cost as a tiled product on the matrix cores: sum_t y_t' W y_t = sum over a [32 rows x n] tile Y of (Y W) o Y.
// One wave per rollout, four rollouts per workgroup sharing one LDS copy of Q and R (zero-padded to whole 32-column
// blocks and an even number of rows).  Per tile of 32 time steps: the rows stream in coalesced (they are contiguous in
// HBM) minus the goal, v_mfma_f32_32x32x2 accumulates Y W over k pairs for each 32-column block, and the accumulator
// (lane (h, j) holds rows 8q+4h+r of column j) is multiplied elementwise with Y and summed - no per-row bookkeeping,
// because every row of the tile belongs to the same rollout.  The terminal row (Qf) is one row per rollout: lanes over
// matrix entries.  HBM traffic is the trajectories, read once.
#define QC_WAVES 4
#define QC_MAXBLK 3  // column blocks of 32: nx <= 96
#define QC_REGS (32 * 32 * QC_MAXBLK / AMX_WAVE)  // values per lane of one staged tile
struct QuadTile {
  // issue the global loads of a tile: rows*n contiguous floats, value q*64+lane in r[q]
  static DEV void fetch(float (&r)[QC_REGS], const float *__restrict__ src, int total, int lane) {
#pragma unroll
    for (int q = 0; q < QC_REGS; ++q) {
      if (q * AMX_WAVE >= total) break;  // wave-uniform
      const int idx = q * AMX_WAVE + lane;
      r[q] = src[idx < total ? idx : total - 1];
    }
  }
  // registers -> LDS tile Y[32][ldy] minus the goal; the padding columns (>= n) are kept zero by the caller, a short tile
  // zeroes its missing rows
  static DEV void stage(const float (&r)[QC_REGS], int rows, int n, int Gl, int Yl, int ldy, int lane) {
    if (rows < 32) {
      for (int idx = rows * ldy + lane; idx < 32 * ldy; idx += AMX_WAVE) lds[Yl + idx] = 0.0f;
    }
    const int total = rows * n;
    int row = 0, col = lane;
    while (col >= n) { col -= n; ++row; }
#pragma unroll
    for (int q = 0; q < QC_REGS; ++q) {
      if (q * AMX_WAVE >= total) break;  // wave-uniform
      if (q * AMX_WAVE + lane < total) lds[Yl + row * ldy + col] = r[q] - (Gl >= 0 ? lds[Gl + col] : 0.0f);
      col += AMX_WAVE;
      while (col >= n) { col -= n; ++row; }
    }
    WSYNC();
  }
  // sum over the staged tile of y' W y
  static DEV float compute(int n, int Wl, int ldw, int nblk, int Yl, int ldy, int lane) {
    const int np4 = (n + 3) & ~3, i = lane & 31, h = lane >> 5;  // W and Y are zero-padded to a multiple of 4 in k
    float acc = 0.0f;
    for (int jb = 0; jb < nblk; ++jb) {
      v16f C = {};
      for (int k0 = 0; k0 < np4; k0 += 4) {  // two k pairs per trip: four operand loads in flight, two dependent MFMAs
        const float a0 = lds[Yl + i * ldy + k0 + h], a1 = lds[Yl + i * ldy + k0 + 2 + h];                        // Y[row i][k]
        const float b0 = lds[Wl + (k0 + h) * ldw + 32 * jb + i], b1 = lds[Wl + (k0 + 2 + h) * ldw + 32 * jb + i];  // W[k][32 jb + i]
        C = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, C, 0, 0, 0);
        C = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, C, 0, 0, 0);
      }
#pragma unroll
      for (int v = 0; v < 16; ++v) acc = __builtin_fmaf(C[v], lds[Yl + (8 * (v >> 2) + 4 * h + (v & 3)) * ldy + 32 * jb + i], acc);
    }
    WSYNC();  // the tile buffer is rewritten by the next stage()
    return acc;
  }
  // all tiles of one [nrows][n] trajectory block: the loads of tile t+1 are in flight while tile t is on the matrix cores
  static DEV float block(const float *__restrict__ src, int nrows, int n, int Gl, int Wl, int ldw, int nblk, int Yl, int ldy, int lane) {
    float r[QC_REGS], acc = 0.0f;
    if (nrows <= 0) return 0.0f;
    fetch(r, src, (nrows < 32 ? nrows : 32) * n, lane);
    for (int t0 = 0; t0 < nrows; t0 += 32) {
      const int rows = nrows - t0 < 32 ? nrows - t0 : 32;
      stage(r, rows, n, Gl, Yl, ldy, lane);
      const int t1 = t0 + 32;
      if (t1 < nrows) fetch(r, src + (size_t)t1 * n, (nrows - t1 < 32 ? nrows - t1 : 32) * n, lane);
      acc += compute(n, Wl, ldw, nblk, Yl, ldy, lane);
    }
    return acc;
  }
};
extern "C" __global__ void __launch_bounds__(AMX_WAVE * QC_WAVES) amx_quadcost_mfma_kernel(int nenv, int nsteps, int nx, int nu, const float *__restrict__ xs,
                                                                                          const float *__restrict__ us, const float *__restrict__ Q,
                                                                                          const float *__restrict__ Qf, const float *__restrict__ R,
                                                                                          const float *__restrict__ xg, float *__restrict__ cost) {
  const int lane = threadIdx.x & (AMX_WAVE - 1), wave = threadIdx.x / AMX_WAVE, e = blockIdx.x * QC_WAVES + wave;
  const int nbx = (nx + 31) >> 5, nbu = (nu + 31) >> 5, ldwx = 32 * nbx, ldwu = 32 * nbu, nxp = (nx + 3) & ~3, nup = (nu + 3) & ~3;
  const int Ql = 0, Rl = Ql + nxp * ldwx, Gl = Rl + nup * ldwu, tiles = Gl + 32 * nbx;
  const int ldy = 32 * (nbx > nbu ? nbx : nbu) + 5;  // odd, and k may run 3 past the last block: lane-per-row reads are conflict free
  for (int k = threadIdx.x; k < nxp * ldwx; k += blockDim.x) { const int r = k / ldwx, c = k - r * ldwx; lds[Ql + k] = (r < nx && c < nx) ? Q[r * nx + c] : 0.0f; }
  for (int k = threadIdx.x; k < nup * ldwu; k += blockDim.x) { const int r = k / ldwu, c = k - r * ldwu; lds[Rl + k] = (r < nu && c < nu) ? R[r * nu + c] : 0.0f; }
  for (int k = threadIdx.x; k < 32 * nbx; k += blockDim.x) lds[Gl + k] = k < nx ? xg[k] : 0.0f;
  __syncthreads();
  if (e >= nenv) return;
  const int Yl = tiles + wave * 32 * ldy;
  float acc = 0.0f;
  const float *x = xs + (size_t)e * (nsteps + 1) * nx;
  for (int idx = lane; idx < 32 * ldy; idx += AMX_WAVE) lds[Yl + idx] = 0.0f;
  WSYNC();
  acc += QuadTile::block(x, nsteps, nx, Gl, Ql, ldwx, nbx, Yl, ldy, lane);
  if (nu > 0) {
    const float *u = us + (size_t)e * nsteps * nu;
    for (int idx = lane; idx < 32 * ldy; idx += AMX_WAVE) lds[Yl + idx] = 0.0f;  // columns nu..nx still hold state data
    WSYNC();
    acc += QuadTile::block(u, nsteps, nu, -1, Rl, ldwu, nbu, Yl, ldy, lane);
  }
  {  // terminal row with Qf
    const float *xN = x + (size_t)nsteps * nx;
    for (int k = lane; k < nx * nx; k += AMX_WAVE) {
      const int i = k / nx, j = k - i * nx;
      acc = __builtin_fmaf((xN[i] - xg[i]) * Qf[k], xN[j] - xg[j], acc);
    }
  }
  acc = wave_sum(acc);
  if (lane == 0) cost[e] = 0.5f * acc;
}

// ---- finite-difference gradient of the terminal cost (BASELINE.json configs[3]; include/amx.h amx_fd_grad) ----
// us[k,t,i] = nominal[t,i] + (the flat coordinate t*nu+i is sample k's own ? eps[i] : 0); sample 0 is the nominal rollout.
extern "C" __global__ void __launch_bounds__(256) amx_fd_perturb_kernel(int nsamples, int nsteps, int nu, int coord0, const float *__restrict__ nominal,
                                                                       const float *__restrict__ eps, float *__restrict__ us) {
  const int per = nsteps * nu;
  const size_t total = (size_t)nsamples * per;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int k = (int)(idx / per), c = (int)(idx - (size_t)k * per);
    float v = nominal[c];
    if (k >= 1 && c == coord0 + k - 1) v += eps[c % nu];
    us[idx] = v;
  }
}
// One wave per sample: J[k] = 0.5 (x_N - xg)' Qf (x_N - xg) (reference cost.py:62-85 with Q = R = 0, the same entry order
// as the terminal row of amx_quadcost), and the forward difference against sample 0 - every wave evaluates the nominal
// cost itself (same operations, same order: bit-identical in all waves) instead of waiting for another wave's result.
extern "C" __global__ void __launch_bounds__(AMX_WAVE) amx_fd_cost_kernel(int nsamples, int nsteps, int nx, int nu, int coord0, const float *__restrict__ xfinal,
                                                                         const float *__restrict__ Qf, const float *__restrict__ xg,
                                                                         const float *__restrict__ eps, float *__restrict__ J, float *__restrict__ grad) {
  const int lane = threadIdx.x, k = blockIdx.x;
  if (k >= nsamples) return;
  const float *xk = xfinal + (size_t)k * nx;
  float acc = 0.0f, acc0 = 0.0f;
  for (int e = lane; e < nx * nx; e += AMX_WAVE) {
    const int i = e / nx, j = e - i * nx;
    const float w = Qf[e], gi = xg[i], gj = xg[j];
    acc = __builtin_fmaf((xk[i] - gi) * w, xk[j] - gj, acc);
    acc0 = __builtin_fmaf((xfinal[i] - gi) * w, xfinal[j] - gj, acc0);
  }
  acc = 0.5f * wave_sum(acc);
  acc0 = 0.5f * wave_sum(acc0);
  if (lane == 0) {
    J[k] = acc;
    if (k >= 1 && grad) {
      const int c = coord0 + k - 1;
      grad[k - 1] = c < nsteps * nu ? (acc - acc0) / eps[c % nu] : 0.0f;
    }
  }
}
