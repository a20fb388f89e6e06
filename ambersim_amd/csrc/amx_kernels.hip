// amx_kernels.hip — the hot path: batched forward dynamics + constraint solve + integration
// for gfx950 (MI355X), one 64-lane wavefront per environment, per-environment state in LDS.
//
// Replaces mjx.forward / mjx.step under MjxEnv.pipeline_step (reference ambersim/rl/base.py:81-96)
// and shoot (reference ambersim/trajopt/shooting.py:22-48).  Stage names follow SURVEY.md §8(a)
// A9.1-A9.14.  Layout rules:
//   * workgroup = 1 wavefront = 1 environment, so every branch on solver state is wave-uniform
//     and __syncthreads() lowers to a wait on the LDS queue (no s_barrier);
//   * per-body / per-dof / per-row arrays are struct-of-arrays in LDS (component-major), so lane b
//     touching element b is bank-conflict free; dense matrices use an odd leading dimension;
//   * tree scans run level by level (bodies sorted by depth); up-scans gather from a static child
//     list instead of scattering;
//   * the small dense factorisations (M, M+J'DJ, M+dt*damping) hold one matrix row per lane in
//     registers and exchange pivots with v_readlane (no LDS traffic inside the factorisation);
//   * wave reductions use DPP row operations + v_readlane, never LDS;
//   * HBM traffic of a rollout is the control stream in and the state rows out, both coalesced.
#include <hip/hip_runtime.h>
#include "amx_dev.h"

#define DEV __device__ __forceinline__
#define MINVAL 1e-15f
#define MINIMP 0.0001f
#define MAXIMP 0.9999f

enum { JFREE = 0, JBALL = 1, JSLIDE = 2, JHINGE = 3 };
enum { DIS_CONSTRAINT = 1, DIS_EQUALITY = 2, DIS_LIMIT = 8, DIS_CONTACT = 16, DIS_PASSIVE = 32,
       DIS_GRAVITY = 64, DIS_CLAMPCTRL = 128, DIS_WARMSTART = 256, DIS_ACTUATION = 1024,
       DIS_REFSAFE = 2048, DIS_EULERDAMP = 16384 };

#define IP(name, i) (m.ip[m.name + (i)])
#define FP(name, i) (m.fp[m.name + (i)])
#define WSYNC() __syncthreads()

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
DEV f4 qnormalize(f4 q) {
  float n = sqrtf(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
  if (n > 0) { float s = 1.0f / n; q.w *= s; q.x *= s; q.y *= s; q.z *= s; }  // division by n, as a reciprocal multiply
  return q;
}
DEV f3 normalize3(f3 v, float *len) {
  float n = sqrtf(dot(v, v));
  *len = n;
  return n > 0 ? mk3(v.x / n, v.y / n, v.z / n) : v;
}
DEV f4 axisangle(f3 axis, float ang) {
  float s, c;
  sincosf(0.5f * ang, &s, &c);
  f4 r = {c, axis.x * s, axis.y * s, axis.z * s};
  return r;
}
DEV void make_frame(f3 n, float *fr) {  // rows a, b, a x b
  float len;
  f3 a = normalize3(n, &len);
  f3 b = (a.y > -0.5f && a.y < 0.5f) ? mk3(0, 1, 0) : mk3(0, 0, 1);
  b = b - a * dot(a, b);
  b = normalize3(b, &len);
  f3 c = cross(a, b);
  fr[0] = a.x; fr[1] = a.y; fr[2] = a.z; fr[3] = b.x; fr[4] = b.y; fr[5] = b.z; fr[6] = c.x; fr[7] = c.y; fr[8] = c.z;
}

// ---- spatial algebra on [rot(3); lin(3)] vectors and 10-float com-frame inertias
struct sv { float v[6]; };
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
// Sum over the 64 lanes, result uniform.  quad_perm xor1, xor2, row_half_mirror, row_mirror inside each
// 16-lane row (DPP, no LDS), then the four row totals are combined through v_readlane.
DEV float wave_sum(float x) {
  x += dpp_f<0xB1>(x);   // quad_perm [1,0,3,2]
  x += dpp_f<0x4E>(x);   // quad_perm [2,3,0,1]
  x += dpp_f<0x141>(x);  // row_half_mirror
  x += dpp_f<0x140>(x);  // row_mirror
  return (readlane_f(x, 0) + readlane_f(x, 16)) + (readlane_f(x, 32) + readlane_f(x, 48));
}
DEV float wave_max(float x) {
  x = fmaxf(x, dpp_f<0xB1>(x));
  x = fmaxf(x, dpp_f<0x4E>(x));
  x = fmaxf(x, dpp_f<0x141>(x));
  x = fmaxf(x, dpp_f<0x140>(x));
  return fmaxf(fmaxf(readlane_f(x, 0), readlane_f(x, 16)), fmaxf(readlane_f(x, 32), readlane_f(x, 48)));
}

// ---- register-row dense Cholesky: lane i holds row i of A (lower part).  Solves A x = rhs.
// Lb receives L (row-major, leading dimension ld) and is also the staging buffer for the
// transposed read of the back substitution.  When `factor` is false the factor already in Lb is reused.
template <int N>
__device__ __noinline__ void chol_solve_rows(const float *A, int ld, int n, float *Lb, const float *rhs, float *x,
                                             bool factor, int lane) {
  float a[N], invd[N];
  if (factor) {
#pragma unroll
    for (int k = 0; k < N; ++k) a[k] = (lane < n && k < n) ? A[lane * ld + k] : (k == lane ? 1.0f : 0.0f);
#pragma unroll
    for (int j = 0; j < N; ++j) {
      float ajj = readlane_f(a[j], j);
      ajj = ajj > 0 ? ajj : MINVAL;
      float inv = 1.0f / sqrtf(ajj);
      invd[j] = inv;
      float lij = a[j] * inv;
      a[j] = lij;
#pragma unroll
      for (int k = j + 1; k < N; ++k) a[k] -= lij * readlane_f(lij, k);
    }
    if (lane < n) {
#pragma unroll
      for (int k = 0; k < N; ++k)
        if (k < n) Lb[lane * ld + k] = a[k];
    }
  } else {
#pragma unroll
    for (int k = 0; k < N; ++k) a[k] = (lane < n && k < n) ? Lb[lane * ld + k] : (k == lane ? 1.0f : 0.0f);
#pragma unroll
    for (int j = 0; j < N; ++j) invd[j] = 1.0f / readlane_f(a[j], j);
  }
  float b = lane < n ? rhs[lane] : 0.0f;
#pragma unroll
  for (int j = 0; j < N; ++j) {  // forward: L y = rhs
    float yj = readlane_f(b, j) * invd[j];
    b = lane == j ? yj : (lane > j ? b - a[j] * yj : b);
  }
  WSYNC();
#pragma unroll
  for (int k = 0; k < N; ++k) a[k] = (lane < n && k < n && k >= lane) ? Lb[k * ld + lane] : 0.0f;  // column `lane` of L
#pragma unroll
  for (int j = N - 1; j >= 0; --j) {  // backward: L' x = y
    float xj = readlane_f(b, j) * invd[j];
    b = lane == j ? xj : (lane < j ? b - a[j] * xj : b);
  }
  if (lane < n) x[lane] = b;
  WSYNC();
}

DEV void chol_solve(int nchol, const float *A, int ld, int n, float *Lb, const float *rhs, float *x, bool factor, int lane) {
  if (nchol <= 8) chol_solve_rows<8>(A, ld, n, Lb, rhs, x, factor, lane);
  else if (nchol <= 16) chol_solve_rows<16>(A, ld, n, Lb, rhs, x, factor, lane);
  else if (nchol <= 24) chol_solve_rows<24>(A, ld, n, Lb, rhs, x, factor, lane);
  else chol_solve_rows<32>(A, ld, n, Lb, rhs, x, factor, lane);
}

// impedance / reference acceleration parameters of one constraint row [A9.6]
DEV void kbi(const DevModel &m, float sr0, float sr1, const float *si, float pos, float *k, float *b, float *imp) {
  float timeconst = sr0, dampratio = sr1;
  if (!(m.disableflags & DIS_REFSAFE)) timeconst = fmaxf(timeconst, 2 * m.dt);
  float dmin = fminf(fmaxf(si[0], MINIMP), MAXIMP), dmax = fminf(fmaxf(si[1], MINIMP), MAXIMP);
  float width = fmaxf(MINVAL, si[2]), mid = fminf(fmaxf(si[3], MINIMP), MAXIMP), power = fmaxf(1.0f, si[4]);
  float kk = 1.0f / (dmax * dmax * timeconst * timeconst * dampratio * dampratio);
  float bb = 2.0f / (dmax * timeconst);
  if (sr0 <= 0) kk = -sr0 / (dmax * dmax);
  if (sr1 <= 0) bb = -sr1 / dmax;
  float x = fabsf(pos) / width;
  float ia = (1.0f / powf(mid, power - 1)) * powf(x, power);
  float ib = 1.0f - (1.0f / powf(1 - mid, power - 1)) * powf(1 - x, power);
  float y = x < mid ? ia : ib;
  float v = dmin + y * (dmax - dmin);
  v = fminf(fmaxf(v, dmin), dmax);
  if (x > 1.0f) v = dmax;
  *k = kk; *b = bb; *imp = v;
}

struct LSPoint { float alpha, cost, d0, d1; };

// Diagnostic build only (-DAMX_PROFILE, libamx_prof.so): per-phase cycle shares via s_memtime.  Never in the product build.
#ifdef AMX_PROFILE
#define PROF_N 24
#define PROF_DECL unsigned prof_t;
#define PROF_INIT()
#define PROF(id) do { unsigned t_ = (unsigned)__builtin_amdgcn_s_memtime(); if (lane == 0) s[p.prof + (id)] += (float)(t_ - prof_t); prof_t = t_; } while (0)
#else
#define PROF_DECL
#define PROF_INIT()
#define PROF(id)
#endif

// =====================================================================================================
// One environment's physics.  All members index LDS through `s` (this wave's slab).
// =====================================================================================================
struct Env {
  const DevModel &m;
  const LdsPlan &p;
  float *s;
  int lane;
  PROF_DECL

  DEV Env(const DevModel &m_, const LdsPlan &p_, float *s_, int lane_) : m(m_), p(p_), s(s_), lane(lane_) {}

  // ------------------------------------------------------------------ A9.1 kinematics
  DEV void kinematics() {
    const int nb = m.nb, nj = m.njnt;
    float *xpos = s + p.xpos, *xquat = s + p.xquat, *qpos = s + p.qpos;
    if (lane == 0) {
      xpos[0] = 0; xpos[nb] = 0; xpos[2 * nb] = 0;
      xquat[0] = 1; xquat[nb] = 0; xquat[2 * nb] = 0; xquat[3 * nb] = 0;
    }
    WSYNC();
    for (int L = 1; L < m.nlevel; ++L) {
      const int beg = IP(level_adr, L), end = IP(level_adr, L + 1);
      for (int t = beg + lane; t < end; t += AMX_WAVE) {
        const int b = IP(level_body, t), par = IP(body_parent, b);
        f3 ppos = mk3(xpos[par], xpos[nb + par], xpos[2 * nb + par]);
        f4 pq = {xquat[par], xquat[nb + par], xquat[2 * nb + par], xquat[3 * nb + par]};
        f3 pos = ppos + qrot(pq, mk3(FP(body_pos, 3 * b), FP(body_pos, 3 * b + 1), FP(body_pos, 3 * b + 2)));
        f4 bq = {FP(body_quat, 4 * b), FP(body_quat, 4 * b + 1), FP(body_quat, 4 * b + 2), FP(body_quat, 4 * b + 3)};
        f4 quat = qmul(pq, bq);
        const int ja = IP(body_jntadr, b), jn = IP(body_jntnum, b);
        for (int k = 0; k < jn; ++k) {
          const int j = ja + k, t_ = IP(jnt_type, j), qa = IP(jnt_qadr, j);
          f3 jpos = mk3(FP(jnt_pos, 3 * j), FP(jnt_pos, 3 * j + 1), FP(jnt_pos, 3 * j + 2));
          f3 jax = mk3(FP(jnt_axis, 3 * j), FP(jnt_axis, 3 * j + 1), FP(jnt_axis, 3 * j + 2));
          f3 anchor, axis;
          if (t_ == JFREE) {
            anchor = mk3(qpos[qa], qpos[qa + 1], qpos[qa + 2]);
            axis = mk3(0, 0, 1);
            pos = anchor;
            f4 q = {qpos[qa + 3], qpos[qa + 4], qpos[qa + 5], qpos[qa + 6]};
            q = qnormalize(q);
            qpos[qa + 3] = q.w; qpos[qa + 4] = q.x; qpos[qa + 5] = q.y; qpos[qa + 6] = q.z;
            quat = q;
          } else {
            anchor = qrot(quat, jpos) + pos;
            axis = qrot(quat, jax);
            if (t_ == JBALL) {
              f4 q = {qpos[qa], qpos[qa + 1], qpos[qa + 2], qpos[qa + 3]};
              q = qnormalize(q);
              qpos[qa] = q.w; qpos[qa + 1] = q.x; qpos[qa + 2] = q.y; qpos[qa + 3] = q.z;
              quat = qmul(quat, q);
              pos = anchor - qrot(quat, jpos);
            } else if (t_ == JHINGE) {
              quat = qmul(quat, axisangle(jax, qpos[qa] - FP(qpos0, qa)));
              pos = anchor - qrot(quat, jpos);
            } else {
              pos = pos + axis * (qpos[qa] - FP(qpos0, qa));
            }
          }
          float *xa = s + p.xanchor, *xx = s + p.xaxis;
          xa[j] = anchor.x; xa[nj + j] = anchor.y; xa[2 * nj + j] = anchor.z;
          xx[j] = axis.x; xx[nj + j] = axis.y; xx[2 * nj + j] = axis.z;
        }
        xpos[b] = pos.x; xpos[nb + b] = pos.y; xpos[2 * nb + b] = pos.z;
        xquat[b] = quat.w; xquat[nb + b] = quat.x; xquat[2 * nb + b] = quat.y; xquat[3 * nb + b] = quat.z;
      }
      WSYNC();
    }
  }

  DEV f3 ld_xpos(int b) const { const float *x = s + p.xpos; return mk3(x[b], x[m.nb + b], x[2 * m.nb + b]); }
  DEV f4 ld_xquat(int b) const {
    const float *x = s + p.xquat;
    f4 q = {x[b], x[m.nb + b], x[2 * m.nb + b], x[3 * m.nb + b]};
    return q;
  }
  DEV f3 ld_com(int r) const { const float *c = s + p.com; return mk3(c[3 * r], c[3 * r + 1], c[3 * r + 2]); }

  // ------------------------------------------------------------------ A9.2 com_pos: tree COM, cinert, cdof
  DEV void com_pos() {
    const int nb = m.nb, nv = m.nv, nj = m.njnt;
    for (int r = 0; r < m.nroot; ++r) {
      float sx = 0, sy = 0, sz = 0, sm = 0;
      for (int b = 1 + lane; b < nb; b += AMX_WAVE) {
        if (IP(body_root, b) == r) {
          f3 ip = mk3(FP(body_ipos, 3 * b), FP(body_ipos, 3 * b + 1), FP(body_ipos, 3 * b + 2));
          f3 xi = ld_xpos(b) + qrot(ld_xquat(b), ip);
          float ms = FP(body_mass, b);
          sx += xi.x * ms; sy += xi.y * ms; sz += xi.z * ms; sm += ms;
        }
      }
      sx = wave_sum(sx); sy = wave_sum(sy); sz = wave_sum(sz); sm = wave_sum(sm);
      if (lane == 0) {
        float *c = s + p.com + 3 * r;
        if (sm < MINVAL) {
          const int rb = IP(root_body, r);
          f3 ip = mk3(FP(body_ipos, 3 * rb), FP(body_ipos, 3 * rb + 1), FP(body_ipos, 3 * rb + 2));
          f3 xi = ld_xpos(rb) + qrot(ld_xquat(rb), ip);
          c[0] = xi.x; c[1] = xi.y; c[2] = xi.z;
        } else {
          c[0] = sx / sm; c[1] = sy / sm; c[2] = sz / sm;
        }
      }
    }
    WSYNC();
    float *ci = s + p.cinert;
    for (int b = lane; b < nb; b += AMX_WAVE) {
      float r[10];
      if (b == 0) {
#pragma unroll
        for (int k = 0; k < 10; ++k) r[k] = 0;
      } else {
        f4 xq = ld_xquat(b);
        f3 ip = mk3(FP(body_ipos, 3 * b), FP(body_ipos, 3 * b + 1), FP(body_ipos, 3 * b + 2));
        f3 xi = ld_xpos(b) + qrot(xq, ip);
        f4 iq = {FP(body_iquat, 4 * b), FP(body_iquat, 4 * b + 1), FP(body_iquat, 4 * b + 2), FP(body_iquat, 4 * b + 3)};
        m33 R = q2m(qmul(xq, iq));
        f3 h = xi - ld_com(IP(body_root, b));
        const float i0 = FP(body_inertia, 3 * b), i1 = FP(body_inertia, 3 * b + 1), i2 = FP(body_inertia, 3 * b + 2);
        const float ms = FP(body_mass, b);
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
      for (int k = 0; k < 10; ++k) ci[k * nb + b] = r[k];
    }
    float *cd = s + p.cdof;
    const float *xa = s + p.xanchor, *xx = s + p.xaxis;
    for (int i = lane; i < nv; i += AMX_WAVE) {
      const int j = IP(dof_jnt, i), b = IP(dof_body, i), kind = IP(dof_kind, i), ax = IP(dof_axis, i);
      f3 off = ld_com(IP(body_root, b)) - mk3(xa[j], xa[nj + j], xa[2 * nj + j]);
      f3 w = mk3(0, 0, 0), l = mk3(0, 0, 0);
      if (kind == 0) {
        l = mk3(ax == 0 ? 1.0f : 0.0f, ax == 1 ? 1.0f : 0.0f, ax == 2 ? 1.0f : 0.0f);
      } else if (kind == 1) {
        m33 R = q2m(ld_xquat(b));
        w = mk3(R.a[ax], R.a[3 + ax], R.a[6 + ax]);
        l = cross(w, off);
      } else if (kind == 2) {
        w = mk3(xx[j], xx[nj + j], xx[2 * nj + j]);
        l = cross(w, off);
      } else {
        l = mk3(xx[j], xx[nj + j], xx[2 * nj + j]);
      }
      cd[i] = w.x; cd[nv + i] = w.y; cd[2 * nv + i] = w.z; cd[3 * nv + i] = l.x; cd[4 * nv + i] = l.y; cd[5 * nv + i] = l.z;
    }
    WSYNC();
  }

  DEV sv ld_sv(int base, int n, int i) const {
    const float *a = s + base;
    sv r = {{a[i], a[n + i], a[2 * n + i], a[3 * n + i], a[4 * n + i], a[5 * n + i]}};
    return r;
  }
  DEV void st_sv(int base, int n, int i, const sv &v) {
    float *a = s + base;
#pragma unroll
    for (int k = 0; k < 6; ++k) a[k * n + i] = v.v[k];
  }

  // ------------------------------------------------------------------ A9.3 crb + mass matrix
  DEV void crb_and_M() {
    const int nb = m.nb, nv = m.nv, ldm = m.ldm;
    float *crb = s + p.crb;
    const float *ci = s + p.cinert;
    for (int L = m.nlevel - 1; L >= 1; --L) {
      const int beg = IP(level_adr, L), end = IP(level_adr, L + 1);
      for (int t = beg + lane; t < end; t += AMX_WAVE) {
        const int b = IP(level_body, t);
        float r[10];
#pragma unroll
        for (int k = 0; k < 10; ++k) r[k] = ci[k * nb + b];
        for (int c = IP(child_adr, b); c < IP(child_adr, b + 1); ++c) {
          const int cb = IP(child_list, c);
#pragma unroll
          for (int k = 0; k < 10; ++k) r[k] += crb[k * nb + cb];
        }
#pragma unroll
        for (int k = 0; k < 10; ++k) crb[k * nb + b] = r[k];
      }
      WSYNC();
    }
    for (int i = lane; i < nv; i += AMX_WAVE) {
      const int b = IP(dof_body, i);
      float I[10];
#pragma unroll
      for (int k = 0; k < 10; ++k) I[k] = crb[k * nb + b];
      st_sv(p.fcrb, nv, i, inert_mul(I, ld_sv(p.cdof, nv, i)));
    }
    WSYNC();
    float *M = s + p.M;
    for (int e = lane; e < m.nM; e += AMX_WAVE) {
      const int i = IP(M_i, e), j = IP(M_j, e);
      sv f = ld_sv(p.fcrb, nv, i), c = ld_sv(p.cdof, nv, j);
      float v = 0;
#pragma unroll
      for (int k = 0; k < 6; ++k) v += c.v[k] * f.v[k];
      if (i == j) v += FP(dof_armature, i);
      M[i * ldm + j] = v;
      M[j * ldm + i] = v;
    }
    WSYNC();
  }

  // ------------------------------------------------------------------ A9.8 com_vel + A9.10 rne (one down-scan, one up-scan)
  DEV void velocity_and_bias() {
    const int nb = m.nb, nv = m.nv;
    const float *qvel = s + p.qvel, *ci = s + p.cinert;
    if (lane == 0) {
      sv z = {{0, 0, 0, 0, 0, 0}};
      st_sv(p.cvel, nb, 0, z);
      if (!(m.disableflags & DIS_GRAVITY)) { z.v[3] = -m.gx; z.v[4] = -m.gy; z.v[5] = -m.gz; }
      st_sv(p.cacc, nb, 0, z);
      sv zz = {{0, 0, 0, 0, 0, 0}};
      st_sv(p.cfrc, nb, 0, zz);
    }
    WSYNC();
    for (int L = 1; L < m.nlevel; ++L) {
      const int beg = IP(level_adr, L), end = IP(level_adr, L + 1);
      for (int t = beg + lane; t < end; t += AMX_WAVE) {
        const int b = IP(level_body, t), par = IP(body_parent, b);
        sv cv = ld_sv(p.cvel, nb, par), ca = ld_sv(p.cacc, nb, par);
        const int ja = IP(body_jntadr, b), jn = IP(body_jntnum, b);
        for (int k = 0; k < jn; ++k) {
          const int j = ja + k, t_ = IP(jnt_type, j), da = IP(jnt_dadr, j);
          if (t_ == JFREE) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
              sv c = ld_sv(p.cdof, nv, da + i);
              const float q = qvel[da + i];
#pragma unroll
              for (int e = 0; e < 6; ++e) cv.v[e] += c.v[e] * q;
            }
            sv cd3 = motion_cross(cv, ld_sv(p.cdof, nv, da + 3)), cd4 = motion_cross(cv, ld_sv(p.cdof, nv, da + 4)),
               cd5 = motion_cross(cv, ld_sv(p.cdof, nv, da + 5));
#pragma unroll
            for (int e = 0; e < 6; ++e) {
              ca.v[e] += cd3.v[e] * qvel[da + 3] + cd4.v[e] * qvel[da + 4] + cd5.v[e] * qvel[da + 5];
            }
#pragma unroll
            for (int i = 3; i < 6; ++i) {
              sv c = ld_sv(p.cdof, nv, da + i);
              const float q = qvel[da + i];
#pragma unroll
              for (int e = 0; e < 6; ++e) cv.v[e] += c.v[e] * q;
            }
          } else {
            const int n = t_ == JBALL ? 3 : 1;
            sv add = {{0, 0, 0, 0, 0, 0}};
            for (int i = 0; i < n; ++i) {
              sv c = ld_sv(p.cdof, nv, da + i);
              sv cd = motion_cross(cv, c);
              const float q = qvel[da + i];
#pragma unroll
              for (int e = 0; e < 6; ++e) { ca.v[e] += cd.v[e] * q; add.v[e] += c.v[e] * q; }
            }
#pragma unroll
            for (int e = 0; e < 6; ++e) cv.v[e] += add.v[e];
          }
        }
        st_sv(p.cvel, nb, b, cv);
        st_sv(p.cacc, nb, b, ca);
        float I[10];
#pragma unroll
        for (int k = 0; k < 10; ++k) I[k] = ci[k * nb + b];
        sv f = inert_mul(I, ca), g = motion_cross_force(cv, inert_mul(I, cv));
#pragma unroll
        for (int e = 0; e < 6; ++e) f.v[e] += g.v[e];
        st_sv(p.cfrc, nb, b, f);
      }
      WSYNC();
    }
    for (int L = m.nlevel - 2; L >= 1; --L) {  // up-scan: gather children
      const int beg = IP(level_adr, L), end = IP(level_adr, L + 1);
      for (int t = beg + lane; t < end; t += AMX_WAVE) {
        const int b = IP(level_body, t);
        const int c0 = IP(child_adr, b), c1 = IP(child_adr, b + 1);
        if (c1 > c0) {
          sv f = ld_sv(p.cfrc, nb, b);
          for (int c = c0; c < c1; ++c) {
            sv g = ld_sv(p.cfrc, nb, IP(child_list, c));
#pragma unroll
            for (int e = 0; e < 6; ++e) f.v[e] += g.v[e];
          }
          st_sv(p.cfrc, nb, b, f);
        }
      }
      WSYNC();
    }
  }

  // ------------------------------------------------------------------ A9.9 passive, A9.7/A9.11 actuation, A9.12 qfrc_smooth
  DEV void smooth_forces() {
    const int nv = m.nv, nu = m.nu;
    const float *qpos = s + p.qpos, *qvel = s + p.qvel, *ctrl = s + p.ctrl;
    float *actf = s + p.actf;
    const bool act_on = nu > 0 && !(m.disableflags & DIS_ACTUATION);
    if (act_on) {
      for (int u = lane; u < nu; u += AMX_WAVE) {
        const int qa = IP(act_qadr, u), da = IP(act_dadr, u), fl = IP(act_flags, u);
        const float gear = FP(act_gear, u);
        float c = ctrl[u];
        if (!(m.disableflags & DIS_CLAMPCTRL) && (fl & 1)) c = fminf(fmaxf(c, FP(act_ctrlrange, 2 * u)), FP(act_ctrlrange, 2 * u + 1));
        float force = FP(act_gain, u) * c;
        if (fl & 4) force += FP(act_bias, 3 * u) + FP(act_bias, 3 * u + 1) * (qpos[qa] * gear) + FP(act_bias, 3 * u + 2) * (qvel[da] * gear);
        if (fl & 2) force = fminf(fmaxf(force, FP(act_forcerange, 2 * u)), FP(act_forcerange, 2 * u + 1));
        actf[u] = gear * force;
      }
    }
    WSYNC();
    float *fs = s + p.fsmooth;
    for (int i = lane; i < nv; i += AMX_WAVE) {
      const int b = IP(dof_body, i), j = IP(dof_jnt, i);
      sv c = ld_sv(p.cdof, nv, i), f = ld_sv(p.cfrc, m.nb, b);
      float bias = 0;
#pragma unroll
      for (int e = 0; e < 6; ++e) bias += c.v[e] * f.v[e];
      float pas = 0;
      if (!(m.disableflags & DIS_PASSIVE)) {
        const float k = FP(jnt_stiff, j);
        const int kind = IP(dof_kind, i);
        if (k != 0.0f && kind >= 2) { const int qa = IP(jnt_qadr, j); pas -= k * (qpos[qa] - FP(qpos_spring, qa)); }
        pas -= FP(dof_damping, i) * qvel[i];
      }
      float act = 0;
      if (act_on) for (int a = IP(dof_actadr, i); a < IP(dof_actadr, i + 1); ++a) act += actf[IP(dof_actlist, a)];
      s[p.fbias + i] = bias;  // qfrc_bias, kept for amx_forward outputs
      fs[i] = pas - bias + act;
    }
    WSYNC();
  }

  // ------------------------------------------------------------------ A9.5 collision
  DEV void plane_sphere(f3 n, f3 ppos, f3 spos, float r, float *dist, f3 *pos) const {
    float d = dot(spos - ppos, n) - r;
    *dist = d;
    *pos = spos - n * (r + 0.5f * d);
  }
  DEV void collision() {
    const int ngc = m.ngc;
    float *gpos = s + p.gpos, *gmat = s + p.gmat;
    for (int g = lane; g < ngc; g += AMX_WAVE) {
      const int b = IP(gc_body, g);
      f4 xq = ld_xquat(b);
      f3 pos = ld_xpos(b) + qrot(xq, mk3(FP(gc_pos, 3 * g), FP(gc_pos, 3 * g + 1), FP(gc_pos, 3 * g + 2)));
      f4 gq = {FP(gc_quat, 4 * g), FP(gc_quat, 4 * g + 1), FP(gc_quat, 4 * g + 2), FP(gc_quat, 4 * g + 3)};
      m33 R = q2m(qmul(xq, gq));
      gpos[g] = pos.x; gpos[ngc + g] = pos.y; gpos[2 * ngc + g] = pos.z;
#pragma unroll
      for (int k = 0; k < 9; ++k) gmat[k * ngc + g] = R.a[k];
    }
    WSYNC();
    float *cdist = s + p.cdist, *cpos = s + p.cpos, *cframe = s + p.cframe;
    const int nc = m.ncon, np = m.npair;
    for (int pr = lane; pr < np; pr += AMX_WAVE) {
      const int g1 = IP(pair_g1, pr), g2 = IP(pair_g2, pr), fn = IP(pair_fn, pr), c0 = IP(pair_conadr, pr);
      f3 pos1 = mk3(gpos[g1], gpos[ngc + g1], gpos[2 * ngc + g1]), pos2 = mk3(gpos[g2], gpos[ngc + g2], gpos[2 * ngc + g2]);
      m33 R1, R2;
#pragma unroll
      for (int k = 0; k < 9; ++k) { R1.a[k] = gmat[k * ngc + g1]; R2.a[k] = gmat[k * ngc + g2]; }
      f3 size1 = mk3(FP(gc_size, 3 * g1), FP(gc_size, 3 * g1 + 1), FP(gc_size, 3 * g1 + 2));
      f3 size2 = mk3(FP(gc_size, 3 * g2), FP(gc_size, 3 * g2 + 1), FP(gc_size, 3 * g2 + 2));
      float fr[9];
      if (fn == 0) {
        f3 n = mk3(R1.a[2], R1.a[5], R1.a[8]), cp;
        float d;
        plane_sphere(n, pos1, pos2, size2.x, &d, &cp);
        cdist[c0] = d; cpos[c0] = cp.x; cpos[nc + c0] = cp.y; cpos[2 * nc + c0] = cp.z;
        make_frame(n, fr);
      } else if (fn == 1) {
        f3 n = mk3(R1.a[2], R1.a[5], R1.a[8]), axis = mk3(R2.a[2], R2.a[5], R2.a[8]);
        float bn;
        f3 b = normalize3(axis - n * dot(n, axis), &bn);
        if (bn < 0.5f) b = (n.y > -0.5f && n.y < 0.5f) ? mk3(0, 1, 0) : mk3(0, 0, 1);
        f3 c = cross(n, b);
        fr[0] = n.x; fr[1] = n.y; fr[2] = n.z; fr[3] = b.x; fr[4] = b.y; fr[5] = b.z; fr[6] = c.x; fr[7] = c.y; fr[8] = c.z;
        for (int k = 0; k < 2; ++k) {
          f3 cp;
          float d;
          plane_sphere(n, pos1, pos2 + axis * (k == 0 ? size2.y : -size2.y), size2.x, &d, &cp);
          cdist[c0 + k] = d; cpos[c0 + k] = cp.x; cpos[nc + c0 + k] = cp.y; cpos[2 * nc + c0 + k] = cp.z;
        }
      } else if (fn == 2) {
        const int nvt = IP(pair_vertnum, pr), va = IP(pair_vertadr, pr);
        f3 nw = mk3(R1.a[2], R1.a[5], R1.a[8]);
        f3 pl = mulTv(R2, pos1 - pos2), n = mulTv(R2, nw);
        float smax = -1e30f;
        for (int v = 0; v < nvt; ++v) {
          f3 vt = mk3(FP(convex_vert, 3 * (va + v)), FP(convex_vert, 3 * (va + v) + 1), FP(convex_vert, 3 * (va + v) + 2));
          smax = fmaxf(smax, dot(pl - vt, n));
        }
        const float thr = fmaxf(0.0f, smax - 1e-3f);
        int idx[4];
        float best;
#define VERT(v) mk3(FP(convex_vert, 3 * (va + (v))), FP(convex_vert, 3 * (va + (v)) + 1), FP(convex_vert, 3 * (va + (v)) + 2))
#define MASKV(vt) (dot(pl - (vt), n) > thr ? 0.0f : -1e6f)
        idx[0] = 0; best = -1e30f;
        for (int v = 0; v < nvt; ++v) { float sc = MASKV(VERT(v)); if (sc > best) { best = sc; idx[0] = v; } }
        f3 a = VERT(idx[0]);
        idx[1] = 0; best = -1e30f;
        for (int v = 0; v < nvt; ++v) { f3 vt = VERT(v); f3 d = a - vt; float sc = dot(d, d) + MASKV(vt); if (sc > best) { best = sc; idx[1] = v; } }
        f3 bb = VERT(idx[1]);
        f3 ab = cross(n, a - bb);
        idx[2] = 0; best = -1e30f;
        for (int v = 0; v < nvt; ++v) { f3 vt = VERT(v); float sc = fabsf(dot(a - vt, ab)) + MASKV(vt); if (sc > best) { best = sc; idx[2] = v; } }
        f3 cc = VERT(idx[2]);
        f3 ac = cross(n, a - cc), bc = cross(n, bb - cc);
        idx[3] = 0; best = -1e30f;
        for (int v = 0; v < nvt; ++v) { f3 vt = VERT(v); float sc = fabsf(dot(bb - vt, bc)) + MASKV(vt); if (sc > best) { best = sc; idx[3] = v; } }
        for (int v = 0; v < nvt; ++v) { f3 vt = VERT(v); float sc = fabsf(dot(a - vt, ac)) + MASKV(vt); if (sc > best) { best = sc; idx[3] = v; } }
        make_frame(nw, fr);
        for (int k = 0; k < 4; ++k) {
          bool unique = true;
          for (int j = 0; j < k; ++j) unique = unique && (idx[j] != idx[k]);
          f3 vt = VERT(idx[k]);
          float d = unique ? -dot(pl - vt, n) : 1.0f;
          f3 cp = pos2 + mulv(R2, vt) - nw * (0.5f * d);
          cdist[c0 + k] = d; cpos[c0 + k] = cp.x; cpos[nc + c0 + k] = cp.y; cpos[2 * nc + c0 + k] = cp.z;
        }
#undef VERT
#undef MASKV
      } else {
        f3 p1 = pos1, p2 = pos2;
        const float r1 = size1.x, r2 = size2.x;
        if (fn == 4) {
          f3 ax = mk3(R2.a[2], R2.a[5], R2.a[8]);
          float t = fminf(fmaxf(dot(pos1 - pos2, ax), -size2.y), size2.y);
          p2 = pos2 + ax * t;
        } else if (fn == 5) {
          f3 a1 = mk3(R1.a[2], R1.a[5], R1.a[8]), a2 = mk3(R2.a[2], R2.a[5], R2.a[8]);
          const float h1 = size1.y, h2 = size2.y;
          f3 dv = pos1 - pos2;
          float a12 = dot(a1, a2), d1 = dot(dv, a1), d2 = dot(dv, a2), det = 1 - a12 * a12, t1, t2;
          t1 = fabsf(det) < 1e-6f ? 0.0f : fminf(fmaxf((a12 * d2 - d1) / det, -h1), h1);
          t2 = fminf(fmaxf(d2 + t1 * a12, -h2), h2);
          t1 = fminf(fmaxf(t2 * a12 - d1, -h1), h1);
          p1 = pos1 + a1 * t1; p2 = pos2 + a2 * t2;
        }
        float len;
        f3 n = normalize3(p2 - p1, &len);
        if (len == 0) n = mk3(1, 0, 0);
        float d = len - (r1 + r2);
        f3 cp = p1 + n * (r1 + 0.5f * d);
        cdist[c0] = d; cpos[c0] = cp.x; cpos[nc + c0] = cp.y; cpos[2 * nc + c0] = cp.z;
        make_frame(n, fr);
      }
#pragma unroll
      for (int k = 0; k < 9; ++k) cframe[k * np + pr] = fr[k];
    }
    WSYNC();
  }

  // ------------------------------------------------------------------ A9.6 make_constraint
  // Row table (static): [0,neq) joint equalities, [neq,neq+nlim) joint limits, then 4 pyramid rows per
  // contact slot.  A row that is inactive this step gets D = 0 (the reference zeroes it instead).
  DEV void make_constraint() {
    const int nv = m.nv, ldj = m.ldj, nc = m.ncon, np = m.npair;
    const float *qpos = s + p.qpos, *qvel = s + p.qvel;
    float *J = s + p.J, *D = s + p.D, *aref = s + p.aref;
    const bool con_on = !(m.disableflags & DIS_CONSTRAINT);
    // contact Jacobian entries: one lane per (contact, pair-dof)
    if (con_on && !(m.disableflags & DIS_CONTACT)) {
      const float *cd = s + p.cdof, *cpos = s + p.cpos, *cframe = s + p.cframe;
      const int row0 = m.neq + m.nlim;
      const int total = IP(pair_dofadr, np) > 0 ? nc : 0;
      // flatten (contact c, slot t) with a per-contact loop over its pair's dof list
      for (int c = 0; c < total; ++c) {
        const int pr = IP(con_pair, c);
        const int d0 = IP(pair_dofadr, pr), d1 = IP(pair_dofadr, pr + 1);
        const float mu = FP(pair_mu, pr);
        f3 pt = mk3(cpos[c], cpos[nc + c], cpos[2 * nc + c]);
        f3 off2 = pt - ld_com(IP(pair_root2, pr)), off1 = pt - ld_com(IP(pair_root1, pr));
        f3 fn_ = mk3(cframe[pr], cframe[np + pr], cframe[2 * np + pr]);
        f3 ft1 = mk3(cframe[3 * np + pr], cframe[4 * np + pr], cframe[5 * np + pr]);
        f3 ft2 = mk3(cframe[6 * np + pr], cframe[7 * np + pr], cframe[8 * np + pr]);
        for (int t = d0 + lane; t < d1; t += AMX_WAVE) {
          const int i = IP(pair_doflist, t), sg = IP(pair_dofsign, t);
          f3 w = mk3(cd[i], cd[nv + i], cd[2 * nv + i]), l = mk3(cd[3 * nv + i], cd[4 * nv + i], cd[5 * nv + i]);
          f3 jp = mk3(0, 0, 0);
          if (sg & 1) jp = jp + (l + cross(w, off2));
          if (sg & 2) jp = jp - (l + cross(w, off1));
          const float jn = dot(fn_, jp), j1 = dot(ft1, jp), j2 = dot(ft2, jp);
          const int r = row0 + 4 * c;
          J[(r + 0) * ldj + i] = jn + mu * j1;
          J[(r + 1) * ldj + i] = jn - mu * j1;
          J[(r + 2) * ldj + i] = jn + mu * j2;
          J[(r + 3) * ldj + i] = jn - mu * j2;
        }
      }
    }
    WSYNC();
    for (int r = lane; r < m.nefc; r += AMX_WAVE) {
      float pos = 0, invw = 0, sr0 = 0.02f, sr1 = 1.0f, si[5] = {0.9f, 0.95f, 0.001f, 0.5f, 2.0f}, jv = 0;
      bool active = false;
      if (!con_on) {
      } else if (r < m.neq) {
        const int e = r;
        if (!(m.disableflags & DIS_EQUALITY)) {
          active = true;
          const int q1 = IP(eq_q1, e), d1 = IP(eq_d1, e), q2 = IP(eq_q2, e), d2 = IP(eq_d2, e);
          const float *c = m.fp + m.eq_data + 5 * e;
          float pos1 = qpos[q1] - FP(qpos0, q1);
          invw = FP(eq_invw1, e);
          J[r * ldj + d1] = 1.0f;
          jv = qvel[d1];
          if (q2 >= 0) {
            const float dif = qpos[q2] - FP(qpos0, q2);
            const float dp2 = dif * dif, dp3 = dp2 * dif, dp4 = dp3 * dif;
            const float deriv = c[1] + c[2] * dif * 2 + c[3] * dp2 * 3 + c[4] * dp3 * 4;
            pos = pos1 - (c[0] + c[1] * dif + c[2] * dp2 + c[3] * dp3 + c[4] * dp4);
            J[r * ldj + d2] = -deriv;
            jv += -deriv * qvel[d2];
            invw += FP(eq_invw2, e);
          } else {
            pos = pos1 - c[0];
          }
          sr0 = FP(eq_solref, 2 * e); sr1 = FP(eq_solref, 2 * e + 1);
#pragma unroll
          for (int k = 0; k < 5; ++k) si[k] = FP(eq_solimp, 5 * e + k);
        }
      } else if (r < m.neq + m.nlim) {
        const int l = r - m.neq;
        if (!(m.disableflags & DIS_LIMIT)) {
          const int qa = IP(lim_qadr, l), da = IP(lim_dadr, l);
          const float q = qpos[qa];
          const float dmin = q - FP(lim_range, 2 * l), dmax = FP(lim_range, 2 * l + 1) - q;
          pos = fminf(dmin, dmax) - FP(lim_margin, l);
          active = pos < 0;
          const float sgn = dmin < dmax ? 1.0f : -1.0f;
          J[r * ldj + da] = sgn;
          jv = sgn * qvel[da];
          invw = FP(lim_invw, l);
          sr0 = FP(lim_solref, 2 * l); sr1 = FP(lim_solref, 2 * l + 1);
#pragma unroll
          for (int k = 0; k < 5; ++k) si[k] = FP(lim_solimp, 5 * l + k);
        }
      } else {
        const int cr = r - m.neq - m.nlim, c = cr >> 2, k = cr & 3;
        if (!(m.disableflags & DIS_CONTACT)) {
          const int pr = IP(con_pair, c);
          const float margin = FP(pair_margin, pr), dist = s[p.cdist + c];
          active = dist < margin;
          pos = dist - margin;
          const float mu = FP(pair_mu, pr), tw = FP(pair_invw, pr);
          const float f = (k & 1) ? -mu : mu;
          invw = (tw + f * f * tw) * 2 * f * f / m.impratio;
          for (int t = IP(pair_dofadr, pr); t < IP(pair_dofadr, pr + 1); ++t) {
            const int i = IP(pair_doflist, t);
            jv += J[r * ldj + i] * qvel[i];
          }
          sr0 = FP(pair_solref, 2 * pr); sr1 = FP(pair_solref, 2 * pr + 1);
#pragma unroll
          for (int kk = 0; kk < 5; ++kk) si[kk] = FP(pair_solimp, 5 * pr + kk);
        }
      }
      float dd = 0, ar = 0;
      if (active) {
        float k_, b_, imp;
        kbi(m, sr0, sr1, si, pos, &k_, &b_, &imp);
        const float rr = fmaxf(invw * (1 - imp) / imp, MINVAL);
        dd = 1.0f / rr;
        ar = -b_ * jv - k_ * imp * pos;
      }
      D[r] = dd;
      aref[r] = ar;
    }
    WSYNC();
  }

  // ------------------------------------------------------------------ A9.13 solver
  DEV void mul_M(const float *v, float *out) {  // out = M v (dense row per lane)
    const int nv = m.nv, ldm = m.ldm;
    const float *M = s + p.M;
    for (int i = lane; i < nv; i += AMX_WAVE) {
      float acc = 0;
      for (int j = 0; j < nv; ++j) acc += M[i * ldm + j] * v[j];
      out[i] = acc;
    }
  }
  DEV float row_dot(int r, const float *v) const {  // J[r,:] . v over the row's static support
    const float *J = s + p.J;
    const int ldj = m.ldj;
    if (r < m.neq) {
      const int d1 = IP(eq_d1, r), d2 = IP(eq_d2, r);
      float a = J[r * ldj + d1] * v[d1];
      if (d2 >= 0) a += J[r * ldj + d2] * v[d2];
      return a;
    } else if (r < m.neq + m.nlim) {
      const int da = IP(lim_dadr, r - m.neq);
      return J[r * ldj + da] * v[da];
    }
    const int pr = IP(con_pair, (r - m.neq - m.nlim) >> 2);
    float a = 0;
    for (int t = IP(pair_dofadr, pr); t < IP(pair_dofadr, pr + 1); ++t) {
      const int i = IP(pair_doflist, t);
      a += J[r * ldj + i] * v[i];
    }
    return a;
  }
  // Jaref = J qacc - aref, Ma = M qacc, then cost / forces (update_constraint).  Returns cost; gauss via pointer.
  DEV void init_context(const float *qacc) {
    mul_M(qacc, s + p.Ma);
    const float *aref = s + p.aref;
    float *Jaref = s + p.Jaref;
    for (int r = lane; r < m.nefc; r += AMX_WAVE) Jaref[r] = row_dot(r, qacc) - aref[r];
    WSYNC();
  }
  DEV float constraint_cost(const float *qacc, float *gauss_out) {
    const float *D = s + p.D, *Jaref = s + p.Jaref, *Ma = s + p.Ma, *fs = s + p.fsmooth, *as = s + p.asmooth;
    float c = 0, g = 0;
    for (int r = lane; r < m.nefc; r += AMX_WAVE) {
      const float ja = Jaref[r];
      const bool act = ja < 0 || r < m.neq;
      if (act) c += 0.5f * D[r] * ja * ja;
    }
    for (int i = lane; i < m.nv; i += AMX_WAVE) g += (Ma[i] - fs[i]) * (qacc[i] - as[i]);
    c = wave_sum(c);
    g = 0.5f * wave_sum(g);
    *gauss_out = g;
    return c + g;
  }
  // qfrc_constraint = J' f with f = -D*Jaref on active rows, then grad = Ma - qfrc_smooth - qfrc_constraint
  DEV void constraint_force_and_grad() {
    const int nv = m.nv, ldj = m.ldj;
    const float *D = s + p.D, *Jaref = s + p.Jaref, *J = s + p.J;
    float *fcon = s + p.fcon, *grad = s + p.grad, *frow = s + p.jv;  // jv doubles as efc_force scratch here
    for (int r = lane; r < m.nefc; r += AMX_WAVE) {
      const float ja = Jaref[r];
      frow[r] = (ja < 0 || r < m.neq) ? -D[r] * ja : 0.0f;
    }
    WSYNC();
    for (int i = lane; i < nv; i += AMX_WAVE) {
      float acc = 0;
      for (int r = 0; r < m.nefc; ++r) acc += J[r * ldj + i] * frow[r];
      fcon[i] = acc;
      grad[i] = s[p.Ma + i] - s[p.fsmooth + i] - acc;
    }
    WSYNC();
  }
  DEV void newton_direction() {  // Mgrad = (M + J' D_active J)^-1 grad
    const int nv = m.nv, ldm = m.ldm, ldj = m.ldj, np = m.npair;
    const float *M = s + p.M, *D = s + p.D, *Jaref = s + p.Jaref, *J = s + p.J;
    float *H = s + p.H;
    for (int e = lane; e < nv * ldm; e += AMX_WAVE) H[e] = M[e];
    WSYNC();
    // equality + limit rows: a handful of entries each
    for (int r = lane; r < m.neq + m.nlim; r += AMX_WAVE) {
      const float ja = Jaref[r];
      const float d = (ja < 0 || r < m.neq) ? D[r] : 0.0f;
      if (d != 0.0f) {
        if (r < m.neq) {
          const int d1 = IP(eq_d1, r), d2 = IP(eq_d2, r);
          const float j1 = J[r * ldj + d1];
          atomicAdd(&H[d1 * ldm + d1], d * j1 * j1);
          if (d2 >= 0) {
            const float j2 = J[r * ldj + d2];
            atomicAdd(&H[d2 * ldm + d2], d * j2 * j2);
            atomicAdd(&H[d1 * ldm + d2], d * j1 * j2);
            atomicAdd(&H[d2 * ldm + d1], d * j1 * j2);
          }
        } else {
          const int da = IP(lim_dadr, r - m.neq);
          atomicAdd(&H[da * ldm + da], d);  // J = +-1
        }
      }
    }
    // contact rows: per pair, outer products over the pair's dof list
    const int row0 = m.neq + m.nlim;
    for (int pr = 0; pr < np; ++pr) {
      const int d0 = IP(pair_dofadr, pr), nd = IP(pair_dofadr, pr + 1) - d0;
      const int c0 = IP(pair_conadr, pr), c1 = pr + 1 < np ? IP(pair_conadr, pr + 1) : m.ncon;
      const int r0 = row0 + 4 * c0, r1 = row0 + 4 * c1;
      // skip pairs with no active row (uniform)
      float any = 0;
      for (int r = r0 + lane; r < r1; r += AMX_WAVE) any += (Jaref[r] < 0 && D[r] > 0) ? 1.0f : 0.0f;
      if (wave_sum(any) == 0.0f) continue;
      for (int e = lane; e < nd * nd; e += AMX_WAVE) {
        const int a = e / nd, b = e - a * nd;
        if (b > a) continue;
        const int i = IP(pair_doflist, d0 + a), j = IP(pair_doflist, d0 + b);
        float acc = 0;
        for (int r = r0; r < r1; ++r) {
          const float ja = Jaref[r];
          if (ja < 0) acc += D[r] * J[r * ldj + i] * J[r * ldj + j];
        }
        atomicAdd(&H[i * ldm + j], acc);
        if (i != j) atomicAdd(&H[j * ldm + i], acc);
      }
    }
    WSYNC();
    PROF(14);
    chol_solve(m.nchol, H, ldm, nv, s + p.Lb, s + p.grad, s + p.Mgrad, true, lane);
  }

  DEV LSPoint ls_point(float alpha, float ja, float jv, float d, bool iseq, bool valid, float g0, float g1, float g2) const {
    const float x = ja + alpha * jv;
    const bool act = valid && (x < 0 || iseq);
    float q0 = act ? 0.5f * ja * ja * d : 0.0f, q1 = act ? jv * ja * d : 0.0f, q2 = act ? 0.5f * jv * jv * d : 0.0f;
    q0 = wave_sum(q0) + g0; q1 = wave_sum(q1) + g1; q2 = wave_sum(q2) + g2;
    LSPoint r;
    r.alpha = alpha;
    r.cost = alpha * alpha * q2 + alpha * q1 + q0;
    r.d0 = 2 * alpha * q2 + q1;
    r.d1 = 2 * q2 + (q2 == 0.0f ? MINVAL : 0.0f);
    return r;
  }
  // generic (nefc > 64) variant: rows strided over lanes, values re-read from LDS
  DEV LSPoint ls_point_lds(float alpha, float g0, float g1, float g2) const {
    const float *D = s + p.D, *Jaref = s + p.Jaref, *jvv = s + p.jv;
    float q0 = 0, q1 = 0, q2 = 0;
    for (int r = lane; r < m.nefc; r += AMX_WAVE) {
      const float ja = Jaref[r], jv = jvv[r], d = D[r];
      if (ja + alpha * jv < 0 || r < m.neq) { q0 += 0.5f * ja * ja * d; q1 += jv * ja * d; q2 += 0.5f * jv * jv * d; }
    }
    q0 = wave_sum(q0) + g0; q1 = wave_sum(q1) + g1; q2 = wave_sum(q2) + g2;
    LSPoint r;
    r.alpha = alpha;
    r.cost = alpha * alpha * q2 + alpha * q1 + q0;
    r.d0 = 2 * alpha * q2 + q1;
    r.d1 = 2 * q2 + (q2 == 0.0f ? MINVAL : 0.0f);
    return r;
  }

  DEV void linesearch(float gauss) {
    const int nv = m.nv, nefc = m.nefc;
    float *search = s + p.search, *mv = s + p.mv, *jvv = s + p.jv, *Jaref = s + p.Jaref, *qacc = s + p.qacc, *Ma = s + p.Ma;
    const float *fs = s + p.fsmooth, *D = s + p.D;
    mul_M(search, mv);
    for (int r = lane; r < nefc; r += AMX_WAVE) jvv[r] = row_dot(r, search);
    WSYNC();
    float sn = 0, g1 = 0, g2 = 0;
    for (int i = lane; i < nv; i += AMX_WAVE) {
      const float si = search[i];
      sn += si * si;
      g1 += si * Ma[i] - si * fs[i];
      g2 += 0.5f * si * mv[i];
    }
    sn = wave_sum(sn); g1 = wave_sum(g1); g2 = wave_sum(g2);
    const float gtol = m.tolerance * m.ls_tolerance * sqrtf(sn) * m.scale;
    const bool small = nefc <= AMX_WAVE;
    const bool valid = lane < nefc;
    const float ja = (small && valid) ? Jaref[lane] : 0.0f, jv = (small && valid) ? jvv[lane] : 0.0f,
                d = (small && valid) ? D[lane] : 0.0f;
    const bool iseq = lane < m.neq;
#define POINT(al) (small ? ls_point((al), ja, jv, d, iseq, valid, gauss, g1, g2) : ls_point_lds((al), gauss, g1, g2))
    const LSPoint p0 = POINT(0.0f);
    LSPoint lo = POINT(p0.alpha - p0.d0 / p0.d1), hi;
    if (lo.d0 < p0.d0) { hi = p0; } else { hi = lo; lo = p0; }
    bool swap = true;
    int it = 0;
    while (true) {
      bool done = it >= m.ls_iterations;
      done = done || !swap;
      done = done || ((lo.d0 < 0) && (lo.d0 > -gtol));
      done = done || ((hi.d0 > 0) && (hi.d0 < gtol));
      if (done) break;
      const LSPoint lo_next = POINT(lo.alpha - lo.d0 / lo.d1);
      const LSPoint hi_next = POINT(hi.alpha - hi.d0 / hi.d1);
      const LSPoint mid = POINT(0.5f * (lo.alpha + hi.alpha));
      const bool s_lo_next = (lo.d0 > 0) || (lo.d0 < lo_next.d0);
      if (s_lo_next) lo = lo_next;
      const bool s_lo_mid = (mid.d0 < 0) && (lo.d0 < mid.d0);
      if (s_lo_mid) lo = mid;
      const bool s_hi_next = (hi.d0 < 0) || (hi.d0 > hi_next.d0);
      if (s_hi_next) hi = hi_next;
      const bool s_hi_mid = (mid.d0 > 0) && (hi.d0 > mid.d0);
      if (s_hi_mid) hi = mid;
      swap = s_lo_next || s_lo_mid || s_hi_next || s_hi_mid;
      ++it;
    }
#undef POINT
    const bool improved = (lo.cost < p0.cost) || (hi.cost < p0.cost);
    const float alpha = lo.cost < hi.cost ? lo.alpha : hi.alpha;
    if (improved) {
      for (int i = lane; i < nv; i += AMX_WAVE) { qacc[i] += alpha * search[i]; Ma[i] += alpha * mv[i]; }
      for (int r = lane; r < nefc; r += AMX_WAVE) Jaref[r] += alpha * jvv[r];
    }
    WSYNC();
  }

  DEV void solve(int *niter_out) {
    const int nv = m.nv;
    float *qacc = s + p.qacc, *search = s + p.search, *grad = s + p.grad, *Mgrad = s + p.Mgrad;
    const float *as = s + p.asmooth, *warm = s + p.warm;
    float gauss, cost, prev_cost;
    bool use_warm = false;
    if (!(m.disableflags & DIS_WARMSTART)) {
      init_context(warm);
      const float cw = constraint_cost(warm, &gauss);
      init_context(as);
      const float cs = constraint_cost(as, &gauss);
      use_warm = cw < cs;
    }
    for (int i = lane; i < nv; i += AMX_WAVE) qacc[i] = use_warm ? warm[i] : as[i];
    WSYNC();
    if (use_warm || (m.disableflags & DIS_WARMSTART)) init_context(qacc);
    cost = constraint_cost(qacc, &gauss);
    prev_cost = INFINITY;
    PROF(9);
    constraint_force_and_grad();
    PROF(10);
    if (m.solver == 2) newton_direction();
    else chol_solve(m.nchol, s + p.M, m.ldm, nv, s + p.Lb, grad, Mgrad, false, lane);
    for (int i = lane; i < nv; i += AMX_WAVE) search[i] = -Mgrad[i];
    WSYNC();
    PROF(11);
    int niter = 0;
    while (true) {
      if (m.iterations != 1 || niter > 0) {
        float gn = 0;
        for (int i = lane; i < nv; i += AMX_WAVE) gn += grad[i] * grad[i];
        gn = wave_sum(gn);
        bool done = niter >= m.iterations;
        done = done || ((prev_cost - cost) / m.scale < m.tolerance);
        done = done || (sqrtf(gn) / m.scale < m.tolerance);
        if (done) break;
      }
      PROF(12);
      linesearch(gauss);
      PROF(13);
      float pg = 0;  // CG: keep grad.Mgrad of the previous iterate
      if (m.solver != 2) {
        for (int i = lane; i < nv; i += AMX_WAVE) { pg += grad[i] * Mgrad[i]; s[p.tmp + i] = Mgrad[i]; }
        pg = wave_sum(pg);
      }
      prev_cost = cost;
      cost = constraint_cost(qacc, &gauss);
      constraint_force_and_grad();
      PROF(10);
      if (m.solver == 2) {
        newton_direction();
        PROF(11);
        for (int i = lane; i < nv; i += AMX_WAVE) search[i] = -Mgrad[i];
      } else {
        chol_solve(m.nchol, s + p.M, m.ldm, nv, s + p.Lb, grad, Mgrad, false, lane);
        float num = 0;
        for (int i = lane; i < nv; i += AMX_WAVE) num += grad[i] * (Mgrad[i] - s[p.tmp + i]);
        num = wave_sum(num);
        const float beta = fmaxf(0.0f, num / fmaxf(MINVAL, pg));
        for (int i = lane; i < nv; i += AMX_WAVE) search[i] = -Mgrad[i] + beta * search[i];
      }
      WSYNC();
      ++niter;
      if (m.iterations == 1) break;
    }
    *niter_out = niter;
  }

  // ------------------------------------------------------------------ forward (A9.1-A9.13)
  DEV void forward() {
    const int nv = m.nv;
    PROF(0);
    kinematics();
    PROF(1);
    com_pos();
    PROF(2);
    crb_and_M();
    PROF(3);
    if (m.ncon > 0 && !(m.disableflags & (DIS_CONTACT | DIS_CONSTRAINT))) collision();
    PROF(4);
    if (m.nefc > 0) make_constraint();
    PROF(5);
    velocity_and_bias();
    PROF(6);
    smooth_forces();
    PROF(7);
    // qacc_smooth = M^-1 qfrc_smooth; the factor of M stays in Lb for the CG preconditioner
    chol_solve(m.nchol, s + p.M, m.ldm, nv, s + p.Lb, s + p.fsmooth, s + p.asmooth, true, lane);
    PROF(8);
    if (m.nefc == 0) {
      for (int i = lane; i < nv; i += AMX_WAVE) { s[p.qacc + i] = s[p.asmooth + i]; s[p.fcon + i] = 0; }
      WSYNC();
      return;
    }
    int niter;
    solve(&niter);
    for (int i = lane; i < nv; i += AMX_WAVE) s[p.warm + i] = s[p.qacc + i];
    WSYNC();
    PROF(15);
  }

  // ------------------------------------------------------------------ A9.14 semi-implicit Euler
  DEV void euler() {
    const int nv = m.nv, ldm = m.ldm;
    float *qpos = s + p.qpos, *qvel = s + p.qvel;
    const float *qacc = s + p.qacc;
    float *acc = s + p.mv;
    if (!(m.disableflags & DIS_EULERDAMP)) {
      const float *M = s + p.M;
      float *H = s + p.H, *rhs = s + p.search;
      for (int e = lane; e < nv * ldm; e += AMX_WAVE) H[e] = M[e];
      for (int i = lane; i < nv; i += AMX_WAVE) rhs[i] = s[p.fsmooth + i] + s[p.fcon + i];
      WSYNC();
      for (int i = lane; i < nv; i += AMX_WAVE) H[i * ldm + i] += m.dt * FP(dof_damping, i);
      WSYNC();
      chol_solve(m.nchol, H, ldm, nv, s + p.Lb, rhs, acc, true, lane);
    } else {
      for (int i = lane; i < nv; i += AMX_WAVE) acc[i] = qacc[i];
      WSYNC();
    }
    PROF(16);
    for (int i = lane; i < nv; i += AMX_WAVE) qvel[i] += acc[i] * m.dt;
    WSYNC();
    for (int j = lane; j < m.njnt; j += AMX_WAVE) {
      const int t_ = IP(jnt_type, j), qa = IP(jnt_qadr, j), da = IP(jnt_dadr, j);
      if (t_ == JFREE || t_ == JBALL) {
        int qq = qa, dd = da;
        if (t_ == JFREE) {
          qpos[qa] += qvel[da] * m.dt; qpos[qa + 1] += qvel[da + 1] * m.dt; qpos[qa + 2] += qvel[da + 2] * m.dt;
          qq = qa + 3; dd = da + 3;
        }
        float n;
        f3 w = normalize3(mk3(qvel[dd], qvel[dd + 1], qvel[dd + 2]), &n);
        f4 q = {qpos[qq], qpos[qq + 1], qpos[qq + 2], qpos[qq + 3]};
        q = qnormalize(qmul(q, axisangle(w, m.dt * n)));
        qpos[qq] = q.w; qpos[qq + 1] = q.x; qpos[qq + 2] = q.y; qpos[qq + 3] = q.z;
      } else {
        qpos[qa] += qvel[da] * m.dt;
      }
    }
    WSYNC();
    PROF(17);
  }
};

// =====================================================================================================
// Kernel: one workgroup (= one wavefront) per environment.
// =====================================================================================================
extern "C" __global__ void __launch_bounds__(AMX_WAVE) amx_env_kernel(const DevModel m, const LdsPlan p, const LaunchArgs a) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x, e = blockIdx.x;
  if (e >= a.nenv) return;
  float *s = lds;
  const int nq = m.nq, nv = m.nv, nu = m.nu, nx = nq + nv;
  // zero the whole slab once: J, M and H have static sparsity and are only written on their support
  for (int i = lane; i < p.total; i += AMX_WAVE) s[i] = 0.0f;
  WSYNC();
  Env env(m, p, s, lane);
#ifdef AMX_PROFILE
  env.prof_t = (unsigned)__builtin_amdgcn_s_memtime();
#endif
  if (a.mode == AMX_MODE_ROLLOUT) {
    const float *x0 = a.x0 + (a.x0_batched ? (size_t)e * nx : 0);
    for (int i = lane; i < nx; i += AMX_WAVE) {
      const float v = x0[i];
      if (i < nq) s[p.qpos + i] = v; else s[p.qvel + i - nq] = v;
      if (a.xs) a.xs[(size_t)e * (a.nsteps + 1) * nx + i] = v;
    }
    WSYNC();
    env.forward();  // shoot(): one forward before the scan; it seeds qacc_warmstart
    const float *us = a.us + (size_t)e * a.nsteps * nu;
    const bool one_pass = nu <= AMX_WAVE;  // control row fits one coalesced wave load: prefetch it a step ahead
    float unext = (one_pass && lane < nu && a.nsteps > 0) ? us[lane] : 0.0f;
    for (int t = 0; t < a.nsteps; ++t) {
      if (one_pass) {
        if (lane < nu) s[p.ctrl + lane] = unext;
        if (t + 1 < a.nsteps && lane < nu) unext = us[(size_t)(t + 1) * nu + lane];
      } else {
        for (int i = lane; i < nu; i += AMX_WAVE) s[p.ctrl + i] = us[(size_t)t * nu + i];
      }
      WSYNC();
      env.forward();
      env.euler();
      if (a.xs) {
        float *row = a.xs + ((size_t)e * (a.nsteps + 1) + t + 1) * nx;
        for (int i = lane; i < nx; i += AMX_WAVE) row[i] = i < nq ? s[p.qpos + i] : s[p.qvel + i - nq];
      }
#ifdef AMX_PROFILE
      { unsigned t_ = (unsigned)__builtin_amdgcn_s_memtime(); if (lane == 0) s[p.prof + 18] += (float)(t_ - env.prof_t); env.prof_t = t_; }
#endif
    }
#ifdef AMX_PROFILE
    WSYNC();
    if (a.prof && lane < PROF_N) a.prof[(size_t)e * PROF_N + lane] = s[p.prof + lane];
#endif
    if (a.xfinal)
      for (int i = lane; i < nx; i += AMX_WAVE) a.xfinal[(size_t)e * nx + i] = i < nq ? s[p.qpos + i] : s[p.qvel + i - nq];
    return;
  }
  // step / forward: state comes from and returns to HBM
  for (int i = lane; i < nq; i += AMX_WAVE) s[p.qpos + i] = a.qpos[(size_t)e * nq + i];
  for (int i = lane; i < nv; i += AMX_WAVE) {
    s[p.qvel + i] = a.qvel[(size_t)e * nv + i];
    const float *w = a.mode == AMX_MODE_STEP ? a.warm : a.warm_in;
    s[p.warm + i] = w ? w[(size_t)e * nv + i] : 0.0f;
  }
  for (int i = lane; i < nu; i += AMX_WAVE) s[p.ctrl + i] = a.us[(size_t)e * nu + i];
  WSYNC();
  if (a.mode == AMX_MODE_STEP) {
    for (int t = 0; t < a.nsteps; ++t) { env.forward(); env.euler(); }
    for (int i = lane; i < nq; i += AMX_WAVE) a.qpos[(size_t)e * nq + i] = s[p.qpos + i];
    for (int i = lane; i < nv; i += AMX_WAVE) {
      a.qvel[(size_t)e * nv + i] = s[p.qvel + i];
      if (a.warm) a.warm[(size_t)e * nv + i] = s[p.warm + i];
    }
    if (a.time && lane == 0) a.time[e] += m.dt * a.nsteps;
    return;
  }
  env.forward();
  for (int i = lane; i < nv; i += AMX_WAVE) {
    if (a.o_qacc) a.o_qacc[(size_t)e * nv + i] = s[p.qacc + i];
    if (a.o_qacc_smooth) a.o_qacc_smooth[(size_t)e * nv + i] = s[p.asmooth + i];
    if (a.o_qfrc_bias) a.o_qfrc_bias[(size_t)e * nv + i] = s[p.fbias + i];
    if (a.o_qfrc_constraint) a.o_qfrc_constraint[(size_t)e * nv + i] = s[p.fcon + i];
  }
  if (a.o_qM)
    for (int k = lane; k < nv * nv; k += AMX_WAVE) a.o_qM[(size_t)e * nv * nv + k] = s[p.M + (k / nv) * m.ldm + (k % nv)];
  if (a.o_xpos)
    for (int k = lane; k < m.nb * 3; k += AMX_WAVE) a.o_xpos[(size_t)e * m.nb * 3 + k] = s[p.xpos + (k % 3) * m.nb + k / 3];
  if (a.o_xquat)
    for (int k = lane; k < m.nb * 4; k += AMX_WAVE) a.o_xquat[(size_t)e * m.nb * 4 + k] = s[p.xquat + (k % 4) * m.nb + k / 4];
  if (a.o_cdist)
    for (int k = lane; k < m.ncon; k += AMX_WAVE) a.o_cdist[(size_t)e * m.ncon + k] = s[p.cdist + k];
  if (a.o_nefc) {
    float cnt = 0;
    for (int r = lane; r < m.nefc; r += AMX_WAVE) cnt += s[p.D + r] > 0 ? 1.0f : 0.0f;
    cnt = wave_sum(cnt);
    if (lane == 0) a.o_nefc[e] = (int)cnt;
  }
}

// Batched StaticGoalQuadraticCost.cost (reference ambersim/trajopt/cost.py:62-85): one wave per rollout.
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
