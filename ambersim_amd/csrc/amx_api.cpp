// amx_api.cpp — host side of the C ABI in include/amx.h: model upload, launch plan, kernel launches.
// No physics here (that is amx_kernels.hip) and no CPU fallback: every entry point that computes
// launches a HIP kernel or fails with AMX_EHIP.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/amx.h"
#include "amx_dev.h"

#include "amx_specs_gen.h"
#include "amx_kernels.hip"  // single translation unit: kernels + launchers, no relocatable device code

static thread_local char g_err[512] = "";
static int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess) return fail(AMX_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_));    \
  } while (0)

// Every entry point runs on the model's device and leaves the caller's current HIP device as it found it
// (a host that drives several GPUs from one thread must not see its current device change under it).
struct DeviceGuard {
  int prev = -1;
  bool good = true, changed = false;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) { good = hipSetDevice(dev) == hipSuccess; changed = good; }
  }
  ~DeviceGuard() { if (changed && prev >= 0) (void)hipSetDevice(prev); }
  bool ok() const { return good; }
};
#define AMX_ON_DEVICE(dev)                                                        \
  DeviceGuard guard_(dev);                                                        \
  if (!guard_.ok()) return fail(AMX_EHIP, "hipSetDevice(%d) failed", (int)(dev))

struct amx_model {
  DevModel dev;
  int spec;  // index into AMX_SPECS of the compile-time specialisation whose plan equals this model's, -1 = generic kernels
  int device, ncu;
  float *d_image;
  int ncon, nefc, lds_bytes;
};

template <int G> static hipError_t set_lds_limit_g(int bytes) {
  hipError_t e;
  if ((e = hipFuncSetAttribute((const void *)amx_rollout_kernel<G, DynSpec>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes)) != hipSuccess) return e;
  if ((e = hipFuncSetAttribute((const void *)amx_step_kernel<G, DynSpec>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes)) != hipSuccess) return e;
  return hipFuncSetAttribute((const void *)amx_forward_kernel<G, DynSpec>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}
static hipError_t set_lds_limit(int G, int bytes) {
  return G == 16 ? set_lds_limit_g<16>(bytes) : (G == 32 ? set_lds_limit_g<32>(bytes) : set_lds_limit_g<64>(bytes));
}

// Runs f(Spec{}) for the spec with index `spec`; false when there is none (generic kernels).
template <class F> static bool for_spec(int spec, F &&f) {
  int idx = 0;
  bool done = false;
#define X(S) if (!done && spec == idx) { f(S{}); done = true; } idx++;
  AMX_SPECS(X)
#undef X
  return done;
}
template <class S> static hipError_t set_lds_limit_spec(int bytes) {
  hipError_t e;
  if ((e = hipFuncSetAttribute((const void *)amx_rollout_kernel<S::G, S>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes)) != hipSuccess) return e;
  if ((e = hipFuncSetAttribute((const void *)amx_step_kernel<S::G, S>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes)) != hipSuccess) return e;
  return hipFuncSetAttribute((const void *)amx_forward_kernel<S::G, S>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

namespace {
// Builder of the constant image: floats, with ints stored bit-cast.
struct Image {
  std::vector<float> v;
  int size() const { return (int)v.size(); }
  void f(double x) { v.push_back((float)x); }
  void i(int x) { float t; memcpy(&t, &x, 4); v.push_back(t); }
  int reserve(int n) { int o = size(); v.resize(v.size() + (n > 0 ? n : 1), 0.f); return o; }
  void seti(int off, int x) { memcpy(&v[off], &x, 4); }
  int geti(int off) const { int x; memcpy(&x, &v[off], 4); return x; }
  void setf(int off, double x) { v[off] = (float)x; }
};
const int PAIR_NCON[6] = {1, 2, 4, 1, 1, 1};

// {k, b, dmin, dmax, width, mid, power, ca, cb}: the position-independent part of MuJoCo's solref/solimp
// mapping (reference physics: constraint._kbi), evaluated once on the host in double precision.
void write_kbi(Image &im, int off, const double *solref, const double *solimp, double dt, bool refsafe) {
  const double MINIMP = 0.0001, MAXIMP = 0.9999, MINV = 1e-15;
  double timeconst = solref[0], dampratio = solref[1];
  if (refsafe) timeconst = std::max(timeconst, 2 * dt);
  auto clip = [](double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); };
  double dmin = clip(solimp[0], MINIMP, MAXIMP), dmax = clip(solimp[1], MINIMP, MAXIMP);
  double width = std::max(MINV, solimp[2]), mid = clip(solimp[3], MINIMP, MAXIMP), power = std::max(1.0, solimp[4]);
  double k = 1.0 / (dmax * dmax * timeconst * timeconst * dampratio * dampratio), b = 2.0 / (dmax * timeconst);
  if (solref[0] <= 0) k = -solref[0] / (dmax * dmax);
  if (solref[1] <= 0) b = -solref[1] / dmax;
  im.setf(off, k); im.setf(off + 1, b); im.setf(off + 2, dmin); im.setf(off + 3, dmax); im.setf(off + 4, width);
  im.setf(off + 5, mid); im.setf(off + 6, power);
  im.setf(off + 7, 1.0 / pow(mid, power - 1)); im.setf(off + 8, 1.0 / pow(1 - mid, power - 1));
}
}  // namespace

extern "C" int amx_version(void) { return AMX_VERSION; }
extern "C" const char *amx_last_error(void) { return g_err; }

// True when J' D J has the sparsity of the dof tree: every contact pair's dof list and every two-joint equality couple dofs
// that lie on ONE ancestor chain (plane - body contacts do; contacts between two branches and mimic joints across fingers do
// not).  Then H = M + J'DJ is eliminated leaves-first without fill-in, like M (static sparse Gauss-Jordan of a spec).
static bool plan_tree_H(const amx_model_desc *d, const DevModel &m, const Image &im) {
  auto is_anc = [&](int a, int i) { for (int k = i; k >= 0; k = d->dof_parentid[k]) if (k == a) return true; return false; };
  for (int p = 0; p < m.npair; p++) {
    const int o = m.c_pair + p * PAIR_S, d0 = im.geti(o + 9), dn = im.geti(o + 10);
    for (int a = 0; a < dn; a++)
      for (int b = 0; b < a; b++) {
        const int i = im.geti(m.c_pairdof + d0 + a) & 255, j = im.geti(m.c_pairdof + d0 + b) & 255;
        if (!is_anc(j, i) && !is_anc(i, j)) return false;
      }
  }
  for (int r = 0; r < m.neq; r++) {
    const int o = m.c_eq + r * EQ_S, d1 = im.geti(o + 1), d2 = im.geti(o + 3);
    if (d2 >= 0 && !is_anc(d1, d2) && !is_anc(d2, d1)) return false;
  }
  return true;
}

// Host-only part of amx_model_create: validates the descriptor and builds the constant image and the launch plan.
// No HIP call in here (tools/gen_specs.py runs it without a GPU through amx_model_plan_describe).
static int plan_model(const amx_model_desc *d, DevModel &m, Image &im) {
  if (!d) return fail(AMX_EINVAL, "null argument");
  if (d->version != AMX_VERSION) return fail(AMX_EINVAL, "amx_model_desc version %d, library %d", d->version, AMX_VERSION);
  if (d->nbody < 1 || d->nq < 0 || d->nv < 0) return fail(AMX_EINVAL, "bad sizes");
  if (d->integrator != 0) return fail(AMX_EUNSUPPORTED, "only the Euler integrator is supported (integrator=%d)", d->integrator);
  if (d->solver != 1 && d->solver != 2) return fail(AMX_EUNSUPPORTED, "solver must be CG (1) or Newton (2), got %d", d->solver);
  if (d->nv < 1) return fail(AMX_EUNSUPPORTED, "model has no degrees of freedom");
  if (d->nv > AMX_MAX_NV) return fail(AMX_ECAPACITY, "nv=%d exceeds the register-row factorisation capacity %d", d->nv, AMX_MAX_NV);
  const int nb = d->nbody, nv = d->nv, nq = d->nq, nu = d->nu, nj = d->njnt, np = d->npair;
  if (nb > AMX_WAVE || nj > AMX_WAVE || np > AMX_WAVE)
    return fail(AMX_ECAPACITY, "lane-role capacity: nbody=%d njnt=%d npair=%d must each be <= %d", nb, nj, np, AMX_WAVE);

  // ---- the descriptor comes across a public ABI: every index is range-checked before it is used as one
  {
    auto bad = [&](const char *field, int idx, int val, int lo, int hi) {
      return fail(AMX_EINVAL, "amx_model_desc: %s[%d] = %d outside [%d, %d)", field, idx, val, lo, hi);
    };
#define AMX_CHK(field, idx, lo, hi) do { const int v_ = d->field[idx]; if (v_ < (lo) || v_ >= (hi)) return bad(#field, idx, v_, lo, hi); } while (0)
    if (d->nu < 0 || d->njnt < 0 || d->ngeom < 0 || d->neq < 0 || d->npair < 0 || d->nconvexvert < 0) return fail(AMX_EINVAL, "negative size in amx_model_desc");
    for (int b = 0; b < nb; b++) {
      if (b > 0) AMX_CHK(body_parentid, b, 0, b);
      if (d->body_jntnum[b] < 0 || d->body_dofnum[b] < 0) return fail(AMX_EINVAL, "body %d: negative joint/dof count", b);
      if (d->body_jntnum[b] > 0 && (d->body_jntadr[b] < 0 || d->body_jntadr[b] + d->body_jntnum[b] > nj)) return bad("body_jntadr", b, d->body_jntadr[b], 0, nj);
      if (d->body_dofnum[b] > 0 && (d->body_dofadr[b] < 0 || d->body_dofadr[b] + d->body_dofnum[b] > nv)) return bad("body_dofadr", b, d->body_dofadr[b], 0, nv);
    }
    static const int QW[4] = {7, 4, 1, 1}, DW[4] = {6, 3, 1, 1};
    for (int j = 0; j < nj; j++) {
      AMX_CHK(jnt_type, j, 0, 4);
      AMX_CHK(jnt_bodyid, j, 1, nb);
      const int t = d->jnt_type[j];
      if (d->jnt_qposadr[j] < 0 || d->jnt_qposadr[j] + QW[t] > nq) return bad("jnt_qposadr", j, d->jnt_qposadr[j], 0, nq - QW[t] + 1);
      if (d->jnt_dofadr[j] < 0 || d->jnt_dofadr[j] + DW[t] > nv) return bad("jnt_dofadr", j, d->jnt_dofadr[j], 0, nv - DW[t] + 1);
    }
    for (int i = 0; i < nv; i++) {
      AMX_CHK(dof_bodyid, i, 1, nb);
      AMX_CHK(dof_jntid, i, 0, nj);
      AMX_CHK(dof_parentid, i, -1, i);
      const int j = d->dof_jntid[i];
      if (i < d->jnt_dofadr[j] || i >= d->jnt_dofadr[j] + DW[d->jnt_type[j]]) return fail(AMX_EINVAL, "dof %d does not belong to its joint %d", i, j);
    }
    for (int g = 0; g < d->ngeom; g++) AMX_CHK(geom_bodyid, g, 0, nb);
    for (int p = 0; p < np; p++) {
      AMX_CHK(pair_geom1, p, 0, d->ngeom);
      AMX_CHK(pair_geom2, p, 0, d->ngeom);
    }
    for (int e = 0; e < d->neq; e++) {
      AMX_CHK(eq_obj1id, e, 0, nj);
      AMX_CHK(eq_obj2id, e, -1, nj);
    }
#undef AMX_CHK
  }

  memset(&m, 0, sizeof(m));
  m.nq = nq; m.nv = nv; m.nu = nu; m.nb = nb; m.njnt = nj; m.npair = np;
  m.iterations = d->iterations; m.ls_iterations = d->ls_iterations; m.solver = d->solver; m.disableflags = d->disableflags;
  m.dt = (float)d->timestep; m.gx = (float)d->gravity[0]; m.gy = (float)d->gravity[1]; m.gz = (float)d->gravity[2];
  m.tolerance = (float)d->tolerance; m.ls_tolerance = (float)d->ls_tolerance; m.impratio = (float)d->impratio;
  m.scale = (float)(d->meaninertia * (nv > 1 ? nv : 1));
  m.nchol = nv;
  const bool refsafe = !(d->disableflags & 2048);

  // ---- tree structure
  std::vector<int> depth(nb, 0), root_slot(nb, 0), roots;
  for (int b = 1; b < nb; b++) {
    int p = d->body_parentid[b];
    if (p < 0 || p >= b) return fail(AMX_EINVAL, "body %d: parent %d must precede it", b, p);
    depth[b] = depth[p] + 1;
    if (p == 0) { root_slot[b] = (int)roots.size(); roots.push_back(b); }
    else root_slot[b] = root_slot[p];
  }
  m.nlevel = *std::max_element(depth.begin(), depth.end()) + 1;
  m.nroot = (int)roots.size();
  std::vector<int> child_adr(nb, 0), child_num(nb, 0), child_list;
  for (int b = 1; b < nb; b++) {
    child_adr[b] = (int)child_list.size();
    for (int c = b + 1; c < nb; c++) if (d->body_parentid[c] == b) child_list.push_back(c);
    child_num[b] = (int)child_list.size() - child_adr[b];
  }

  // bodies
  m.c_body = im.reserve(nb * BODY_S);
  for (int b = 0; b < nb; b++) {
    int o = m.c_body + b * BODY_S;
    im.seti(o, d->body_parentid[b]); im.seti(o + 1, depth[b]); im.seti(o + 2, root_slot[b]);
    im.seti(o + 3, d->body_jntadr[b] < 0 ? 0 : d->body_jntadr[b]); im.seti(o + 4, d->body_jntnum[b]);
    im.seti(o + 5, child_adr[b]); im.seti(o + 6, child_num[b]);
    for (int k = 0; k < 3; k++) { im.setf(o + 7 + k, d->body_pos[3 * b + k]); im.setf(o + 14 + k, d->body_ipos[3 * b + k]); im.setf(o + 22 + k, d->body_inertia[3 * b + k]); }
    for (int k = 0; k < 4; k++) { im.setf(o + 10 + k, d->body_quat[4 * b + k]); im.setf(o + 17 + k, d->body_iquat[4 * b + k]); }
    im.setf(o + 21, d->body_mass[b]);
  }
  // joints
  m.c_jnt = im.reserve(nj * JNT_S);
  for (int j = 0; j < nj; j++) {
    int o = m.c_jnt + j * JNT_S, t = d->jnt_type[j], qa = d->jnt_qposadr[j];
    if (t <= 1 && d->jnt_stiffness[j] != 0) return fail(AMX_EUNSUPPORTED, "spring on free/ball joint %d", j);
    if (t <= 1 && d->jnt_limited[j]) return fail(AMX_EUNSUPPORTED, "limits on free/ball joint %d", j);
    im.seti(o, t); im.seti(o + 1, qa); im.seti(o + 2, d->jnt_dofadr[j]); im.seti(o + 3, d->jnt_bodyid[j]);
    for (int k = 0; k < 3; k++) { im.setf(o + 4 + k, d->jnt_pos[3 * j + k]); im.setf(o + 7 + k, d->jnt_axis[3 * j + k]); }
    im.setf(o + 10, d->qpos0[qa]); im.setf(o + 11, d->jnt_stiffness[j]); im.setf(o + 12, d->qpos_spring[qa]);
  }
  // actuators (+ per-dof lists)
  std::vector<int> act_dadr(nu);
  m.c_act = im.reserve(nu * ACT_S);
  for (int u = 0; u < nu; u++) {
    int j = d->actuator_trnid[u], o = m.c_act + u * ACT_S;
    if (j < 0 || j >= nj || d->jnt_type[j] < 2) return fail(AMX_EUNSUPPORTED, "actuator %d must drive a hinge or slide joint", u);
    act_dadr[u] = d->jnt_dofadr[j];
    im.seti(o, d->jnt_qposadr[j]); im.seti(o + 1, act_dadr[u]);
    im.seti(o + 2, (d->actuator_ctrllimited[u] ? 1 : 0) | (d->actuator_forcelimited[u] ? 2 : 0) | (d->actuator_biastype[u] == 1 ? 4 : 0));
    im.setf(o + 3, d->actuator_gear[u]); im.setf(o + 4, d->actuator_gainprm[3 * u]);
    for (int k = 0; k < 3; k++) im.setf(o + 5 + k, d->actuator_biasprm[3 * u + k]);
    for (int k = 0; k < 2; k++) { im.setf(o + 8 + k, d->actuator_ctrlrange[2 * u + k]); im.setf(o + 10 + k, d->actuator_forcerange[2 * u + k]); }
  }
  std::vector<int> dofact_adr(nv), dofact_num(nv), dofact;
  for (int i = 0; i < nv; i++) {
    dofact_adr[i] = (int)dofact.size();
    for (int u = 0; u < nu; u++) if (act_dadr[u] == i) dofact.push_back(u);
    dofact_num[i] = (int)dofact.size() - dofact_adr[i];
  }
  // dofs
  m.c_dof = im.reserve(nv * DOF_S);
  for (int i = 0; i < nv; i++) {
    int o = m.c_dof + i * DOF_S, j = d->dof_jntid[i], k = i - d->jnt_dofadr[j], t = d->jnt_type[j];
    int kind = t == 0 ? (k < 3 ? 0 : 1) : (t == 1 ? 1 : (t == 3 ? 2 : 3)), axis = t == 0 ? k % 3 : (t == 1 ? k : 0);
    im.seti(o, d->dof_bodyid[i]); im.seti(o + 1, j); im.seti(o + 2, kind); im.seti(o + 3, axis);
    im.setf(o + 4, d->dof_armature[i]); im.setf(o + 5, d->dof_damping[i]);
    im.seti(o + 6, dofact_adr[i]); im.seti(o + 7, dofact_num[i]); im.seti(o + 8, root_slot[d->dof_bodyid[i]]);
  }
  // collision geoms (only those that appear in a pair)
  std::vector<int> gmap(d->ngeom > 0 ? d->ngeom : 1, -1), gc_src;
  auto use_geom = [&](int g) { if (gmap[g] < 0) { gmap[g] = (int)gc_src.size(); gc_src.push_back(g); } return gmap[g]; };
  for (int p = 0; p < np; p++) { use_geom(d->pair_geom1[p]); use_geom(d->pair_geom2[p]); }
  m.ngc = (int)gc_src.size();
  if (m.ngc > AMX_WAVE) return fail(AMX_ECAPACITY, "%d collision geoms exceed the lane-role capacity %d", m.ngc, AMX_WAVE);
  m.c_geom = im.reserve(m.ngc * GEOM_S);
  for (int g = 0; g < m.ngc; g++) {
    int o = m.c_geom + g * GEOM_S, s = gc_src[g];
    im.seti(o, d->geom_bodyid[s]); im.seti(o + 1, d->geom_type[s]);
    for (int k = 0; k < 3; k++) { im.setf(o + 2 + k, d->geom_pos[3 * s + k]); im.setf(o + 9 + k, d->geom_size[3 * s + k]); }
    for (int k = 0; k < 4; k++) im.setf(o + 5 + k, d->geom_quat[4 * s + k]);
  }
  // pairs, contact slots, per-pair dof lists, Jacobian work items, Hessian work items
  m.c_pair = im.reserve(np * PAIR_S);
  std::vector<int> con_pair, pairdof, jac_items, hent;
  std::vector<std::vector<int>> dofmem(nv), hmembers(nv * (nv + 1) / 2);
  int maxchain = 1;
  int ncon = 0;
  for (int p = 0; p < np; p++) {
    int g1 = d->pair_geom1[p], g2 = d->pair_geom2[p], fn = d->pair_fn[p], o = m.c_pair + p * PAIR_S;
    if (fn < 0 || fn > 5) return fail(AMX_EUNSUPPORTED, "pair %d: unknown collision function %d", p, fn);
    if (fn == 2 && (d->pair_vertnum[p] < 1 || d->pair_vertadr[p] < 0 || d->pair_vertadr[p] + d->pair_vertnum[p] > d->nconvexvert))
      return fail(AMX_EINVAL, "pair %d: bad convex vertex range", p);
    int b1 = d->geom_bodyid[g1], b2 = d->geom_bodyid[g2];
    im.seti(o, gmap[g1]); im.seti(o + 1, gmap[g2]); im.seti(o + 2, fn); im.seti(o + 3, ncon); im.seti(o + 4, PAIR_NCON[fn]);
    im.seti(o + 5, fn == 2 ? d->pair_vertadr[p] : 0); im.seti(o + 6, fn == 2 ? d->pair_vertnum[p] : 0);
    im.seti(o + 7, root_slot[b1]); im.seti(o + 8, root_slot[b2]);
    im.setf(o + 11, d->pair_friction[3 * p]);
    write_kbi(im, o + 12, d->pair_solref + 2 * p, d->pair_solimp + 5 * p, d->timestep, refsafe);
    im.setf(o + 21, d->pair_margin[p]);
    im.setf(o + 22, d->body_invweight0[2 * b1] + d->body_invweight0[2 * b2]);
    // union of the ancestor dof chains of both bodies
    std::vector<int> sign(nv, 0);
    for (int side = 0; side < 2; side++) {
      int b = side == 0 ? b2 : b1;
      while (b > 0 && d->body_dofnum[b] == 0) b = d->body_parentid[b];
      if (b == 0) continue;
      for (int i = d->body_dofadr[b] + d->body_dofnum[b] - 1; i >= 0; i = d->dof_parentid[i]) sign[i] |= (side == 0 ? 1 : 2);
    }
    int d0 = (int)pairdof.size();
    for (int i = 0; i < nv; i++) if (sign[i]) pairdof.push_back(i | (sign[i] << 8));
    int dn = (int)pairdof.size() - d0;
    im.seti(o + 9, d0); im.seti(o + 10, dn);
    if (dn > 16) return fail(AMX_ECAPACITY, "pair %d couples %d dofs (limit 16: the row lanes keep the dof list in registers)", p, dn);
    maxchain = std::max(maxchain, dn);
    for (int a = 0; a < dn; a++) {
      for (int k = 0; k < PAIR_NCON[fn]; k++) dofmem[pairdof[d0 + a] & 255].push_back((4 * (ncon + k)) | (a << 8));
      for (int b = 0; b <= a; b++) {  // dof list ascending, so (dof_a, dof_b) is a lower-triangle entry of H
        const int di = pairdof[d0 + a] & 255, dj = pairdof[d0 + b] & 255;
        for (int k = 0; k < PAIR_NCON[fn]; k++) hmembers[di * (di + 1) / 2 + dj].push_back(di | (dj << 5) | ((4 * (ncon + k)) << 10) | (a << 18) | (b << 23));
      }
    }
    for (int k = 0; k < PAIR_NCON[fn]; k++) {
      for (int t = 0; t < dn; t++)  // contact | slot<<6 | dof<<11 | chain sign<<16 | pair<<18
        jac_items.push_back((ncon + k) | (t << 6) | ((pairdof[d0 + t] & 255) << 11) | ((pairdof[d0 + t] >> 8) << 16) | (p << 18));
      con_pair.push_back(p);
    }
    ncon += PAIR_NCON[fn];
  }
  for (int t = 0; t < nv * (nv + 1) / 2; t++)
    for (int w : hmembers[t]) hent.push_back(w);  // one term per (H entry, contact)
  m.ncon = ncon; m.njac = (int)jac_items.size(); m.nhent = (int)hent.size();
  if (ncon > 63) return fail(AMX_ECAPACITY, "%d contact slots exceed 63", ncon);
  // limit rows, equality rows
  std::vector<int> lim_j;
  for (int j = 0; j < nj; j++) if (d->jnt_limited[j]) lim_j.push_back(j);
  m.nlim = (int)lim_j.size();
  m.c_lim = im.reserve(m.nlim * LIM_S);
  for (int l = 0; l < m.nlim; l++) {
    int j = lim_j[l], o = m.c_lim + l * LIM_S;
    im.seti(o, d->jnt_qposadr[j]); im.seti(o + 1, d->jnt_dofadr[j]);
    im.setf(o + 2, d->jnt_range[2 * j]); im.setf(o + 3, d->jnt_range[2 * j + 1]); im.setf(o + 4, d->jnt_margin[j]);
    write_kbi(im, o + 5, d->jnt_solref + 2 * j, d->jnt_solimp + 5 * j, d->timestep, refsafe);
    im.setf(o + 14, d->dof_invweight0[d->jnt_dofadr[j]]);
  }
  std::vector<int> eq_e;
  for (int e = 0; e < d->neq; e++) if (d->eq_active[e]) eq_e.push_back(e);
  m.neq = (int)eq_e.size();
  m.c_eq = im.reserve(m.neq * EQ_S);
  for (int r = 0; r < m.neq; r++) {
    int e = eq_e[r], o = m.c_eq + r * EQ_S, j1 = d->eq_obj1id[e], j2 = d->eq_obj2id[e];
    if (j1 < 0 || j1 >= nj || d->jnt_type[j1] < 2 || (j2 >= 0 && (j2 >= nj || d->jnt_type[j2] < 2)))
      return fail(AMX_EUNSUPPORTED, "equality %d must couple hinge/slide joints", e);
    im.seti(o, d->jnt_qposadr[j1]); im.seti(o + 1, d->jnt_dofadr[j1]);
    im.seti(o + 2, j2 >= 0 ? d->jnt_qposadr[j2] : -1); im.seti(o + 3, j2 >= 0 ? d->jnt_dofadr[j2] : -1);
    for (int k = 0; k < 5; k++) im.setf(o + 4 + k, d->eq_data[5 * e + k]);
    write_kbi(im, o + 9, d->eq_solref + 2 * e, d->eq_solimp + 5 * e, d->timestep, refsafe);
    im.setf(o + 18, d->dof_invweight0[d->jnt_dofadr[j1]] + (j2 >= 0 ? d->dof_invweight0[d->jnt_dofadr[j2]] : 0.0));
    im.setf(o + 19, d->qpos0[d->jnt_qposadr[j1]]); im.setf(o + 20, j2 >= 0 ? d->qpos0[d->jnt_qposadr[j2]] : 0.0);
  }
  const bool con_off = d->disableflags & 1;
  m.nefc = con_off ? 0 : m.neq + m.nlim + 4 * m.ncon;
  if (m.nefc > AMX_WAVE) return fail(AMX_ECAPACITY, "%d constraint rows exceed the lane-role capacity %d", m.nefc, AMX_WAVE);
  // lanes per environment: the smallest group that holds every lane-role index set
  {
    const int need = std::max({nb, nj, nv, np, m.ngc, m.nefc, 1});
    m.G = need <= 16 ? 16 : (need <= 32 ? 32 : 64);
    if (const char *force = getenv("AMX_FORCE_G")) { int g = atoi(force); if ((g == 16 || g == 32 || g == 64) && g >= m.G) m.G = g; }
  }
  // lists
  auto add_ints = [&](const std::vector<int> &v) { int o = im.reserve((int)v.size()); for (size_t k = 0; k < v.size(); k++) im.seti(o + (int)k, v[k]); return o; };
  m.c_child = add_ints(child_list);
  std::vector<int> Mlist;
  for (int i = 0; i < nv; i++) for (int j = i; j >= 0; j = d->dof_parentid[j]) Mlist.push_back(i | (j << 8));
  m.nM = (int)Mlist.size(); m.c_M = add_ints(Mlist);
  if (m.G >= 32) { hent.clear(); m.nhent = 0; }  // wide groups build the Hessian on the matrix cores; only 16-lane groups walk the term list
  m.c_dofact = add_ints(dofact); m.c_pairdof = add_ints(pairdof); m.c_jac = add_ints(jac_items); m.c_hent = add_ints(hent);
  m.c_conpair = add_ints(con_pair); m.c_rootbody = add_ints(roots);
  {
    std::vector<int> slotmap((size_t)32 * std::max(ncon, 1), -1);
    for (int c = 0; c < ncon; c++) {
      const int o = m.c_pair + con_pair[c] * PAIR_S, d0 = im.geti(o + 9), dn = im.geti(o + 10);
      for (int t = 0; t < dn; t++) slotmap[(size_t)32 * c + (pairdof[d0 + t] & 255)] = t;
    }
    m.c_slot = add_ints(slotmap);
  }
  {
    std::vector<int> adr(nv + 1, 0), flat;
    for (int i = 0; i < nv; i++) { adr[i] = (int)flat.size(); flat.insert(flat.end(), dofmem[i].begin(), dofmem[i].end()); }
    adr[nv] = (int)flat.size();
    m.c_dofmemadr = add_ints(adr); m.c_dofmem = add_ints(flat);
  }
  m.ldc = maxchain | 1;
  m.ldm = ((nv + 3) / 4) * 4;
  if ((m.ldm / 4) % 2 == 0) m.ldm += 4;  // rows 16-byte aligned, ldm/4 odd: 8 lanes reading their rows as b128 cover all 32 banks
  m.c_qpos0 = im.reserve(nq);
  for (int k = 0; k < nq; k++) im.setf(m.c_qpos0 + k, d->qpos0[k]);
  m.c_vert = im.reserve(3 * d->nconvexvert);
  for (int k = 0; k < 3 * d->nconvexvert; k++) im.setf(m.c_vert + k, d->convex_vert[k]);
  while (im.size() & 3) im.reserve(1);  // environment slabs start 16-byte aligned
  m.nconst = im.size();

  // ---- environment slab plan (floats)
  int off = 0;
  auto take = [&](int n) { int o = off; off += (n > 0 ? n : 1); return o; };
  m.e_qpos = take(nq); m.e_qvel = take(nv); m.e_ctrl = take(nu);
  m.e_body = take(BODY_E * nb); m.e_dof = take(DOF_E * nv);
  m.e_com = take(3 * m.nroot); m.e_actf = take(nu);
  const int ntri = nv * (nv + 1) / 2;
  // joint anchors/axes (kinematics -> cdof) are dead before H is first written: they share its space
  off = (off + 3) & ~3;  // M and H are full symmetric [nv][ldm] with 16-byte aligned rows (row loads and M*v use ds_read_b128)
  m.e_M = take(nv * m.ldm);
  m.e_H = take(std::max(std::max(nv * m.ldm, JNT_E * nj), nv + 8)); m.e_jnt = m.e_H; m.e_Lb = d->solver == 1 ? take(ntri) : m.e_H;
  m.e_row = take(ROW_E * m.nefc);
  // collision geom poses (com_pos -> collision) are dead before the contact rows are written: they share their space
  m.e_J = take(std::max(4 * m.ncon * m.ldc, GEO_E * m.ngc)); m.e_geom = m.e_J;
  m.e_con = take(CON_E * m.ncon); m.e_frame = take(9 * np);
#ifdef AMX_PROFILE
  m.e_prof = take(24);
#endif
  m.env_floats = (off + 3) & ~3;  // slabs keep 16-byte alignment
  // waves per workgroup: each workgroup carries one copy of the constants, so pick the size that keeps the most
  // waves resident per CU (160 KiB of LDS, at most AMX_MIN_WAVES_PER_SIMD*4 waves by registers); ties go to the larger
  int best_w = 0;  // largest workgroup (in waves) whose LDS fits; each launch picks 1..best_w (choose_wpb)
  for (int w = 1; w <= AMX_WAVES_PER_BLOCK; w++)
    if ((m.nconst + w * (AMX_WAVE / m.G) * m.env_floats) * (int)sizeof(float) <= 160 * 1024) best_w = w;
  if (best_w == 0)
    return fail(AMX_ECAPACITY, "model needs %d bytes of LDS for one wave (limit 163840)", (m.nconst + (AMX_WAVE / m.G) * m.env_floats) * 4);
  m.wpb = best_w;
  const int envs_per_block = m.wpb * (AMX_WAVE / m.G);
  const int lds_bytes = (m.nconst + envs_per_block * m.env_floats) * (int)sizeof(float);
  if (lds_bytes > 160 * 1024)
    return fail(AMX_ECAPACITY, "model needs %d bytes of LDS per workgroup (%d constants + %d x %d per environment; limit 163840)",
                lds_bytes, m.nconst * 4, envs_per_block, m.env_floats * 4);

  return AMX_OK;
}

// A generated spec applies when the model's whole plan equals it: every integer, the dof tree, the H-sparsity class.
template <class S> static bool spec_matches(const DevModel &m, const amx_model_desc *d, bool tree_H) {
#define X(f) if (m.f != S::f) return false;
  AMX_MODEL_INTS(X)
#undef X
  for (int i = 0; i < m.nv; i++) if (d->dof_parentid[i] != S::dof_parent[i]) return false;
  return tree_H == S::tree_H;
}
static int find_spec(const DevModel &m, const amx_model_desc *d, const Image &im) {
  if (const char *off = getenv("AMX_NO_SPEC")) { if (atoi(off) != 0) return -1; }  // diagnostic: force the generic kernels
  const bool tree_H = plan_tree_H(d, m, im);
  int idx = 0, found = -1;
#define X(S) if (found < 0 && spec_matches<S>(m, d, tree_H)) found = idx; idx++;
  AMX_SPECS(X)
#undef X
  return found;
}

extern "C" int amx_model_create(const amx_model_desc *d, int device, amx_model **out) {
  if (!d || !out) return fail(AMX_EINVAL, "null argument");
  DevModel m;
  Image im;
  if (int rc = plan_model(d, m, im)) return rc;
  const int spec = find_spec(m, d, im);
  const int lds_bytes = (m.nconst + m.wpb * (AMX_WAVE / m.G) * m.env_floats) * (int)sizeof(float);
  amx_model *h = new (std::nothrow) amx_model();
  if (!h) return fail(AMX_ENOMEM, "out of host memory");
  h->ncu = 0;
  h->spec = spec;
  h->device = device; h->ncon = m.ncon; h->nefc = m.nefc; h->lds_bytes = lds_bytes; h->d_image = nullptr;
  hipError_t e;
  DeviceGuard guard(device);
  if ((e = guard.ok() ? hipSuccess : hipErrorInvalidDevice) != hipSuccess || (e = hipMalloc((void **)&h->d_image, im.v.size() * sizeof(float))) != hipSuccess ||
      (e = hipMemcpy(h->d_image, im.v.data(), im.v.size() * sizeof(float), hipMemcpyHostToDevice)) != hipSuccess ||
      (e = set_lds_limit(m.G, lds_bytes)) != hipSuccess ||
      (for_spec(spec, [&](auto sp) { e = set_lds_limit_spec<decltype(sp)>(lds_bytes); }) && e != hipSuccess)) {
    if (h->d_image) (void)hipFree(h->d_image);
    delete h;
    return fail(AMX_EHIP, "model upload failed: %s", hipGetErrorString(e));
  }
  m.image = h->d_image;
  h->dev = m;
  { int n = 0; if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess) h->ncu = n; }
  *out = h;
  return AMX_OK;
}

// Tooling (not part of include/amx.h): the launch plan of a descriptor as "name=value" lines - every AMX_MODEL_INTS
// field, then the dof tree (dof_parent) and whether each contact / equality row couples dofs of one ancestor chain only.
// tools/gen_specs.py writes amx_specs_gen.h from this; tests/test_abi.py checks the committed header is current.
extern "C" int amx_model_plan_describe(const amx_model_desc *d, char *out, size_t cap) {
  if (!d || !out || cap == 0) return fail(AMX_EINVAL, "null argument");
  DevModel m;
  Image im;
  if (int rc = plan_model(d, m, im)) return rc;
  std::string t;
  char line[128];
#define X(f) snprintf(line, sizeof(line), #f "=%d\n", m.f); t += line;
  AMX_MODEL_INTS(X)
#undef X
  t += "dof_parent=";
  for (int i = 0; i < m.nv; i++) { snprintf(line, sizeof(line), "%d%s", d->dof_parentid[i], i + 1 < m.nv ? "," : ""); t += line; }
  t += "\n";
  snprintf(line, sizeof(line), "tree_H=%d\n", plan_tree_H(d, m, im) ? 1 : 0);
  t += line;
  if (t.size() + 1 > cap) return fail(AMX_EINVAL, "amx_model_plan_describe: buffer of %zu bytes, %zu needed", cap, t.size() + 1);
  memcpy(out, t.c_str(), t.size() + 1);
  return AMX_OK;
}

extern "C" int amx_model_destroy(amx_model *m) {
  if (!m) return AMX_OK;
  {
    DeviceGuard guard(m->device);
    (void)hipFree(m->d_image);
  }
  delete m;
  return AMX_OK;
}

extern "C" int amx_model_query(const amx_model *m, const char *name, int *out) {
  if (!m || !name || !out) return fail(AMX_EINVAL, "null argument");
  std::string n(name);
  if (n == "nq") *out = m->dev.nq;
  else if (n == "nv") *out = m->dev.nv;
  else if (n == "nu") *out = m->dev.nu;
  else if (n == "nbody") *out = m->dev.nb;
  else if (n == "ncon") *out = m->ncon;
  else if (n == "nefc_max") *out = m->nefc;
  else if (n == "lds_bytes") *out = m->lds_bytes;
  else if (n == "lanes_per_env") *out = m->dev.G;
  else if (n == "envs_per_block") *out = m->dev.wpb * (AMX_WAVE / m->dev.G);
  else if (n == "env_slab_bytes") *out = m->dev.env_floats * 4;
  else if (n == "const_bytes") *out = m->dev.nconst * 4;
  else if (n == "specialised") *out = m->spec >= 0 ? 1 : 0;
  else return fail(AMX_EINVAL, "unknown query '%s'", name);
  return AMX_OK;
}

// Waves per workgroup for this launch.  Every workgroup carries one copy of the constants, so fewer, larger
// workgroups keep more environments resident per CU; but a batch that does not fill a whole number of residency
// rounds leaves the last round partly empty.  Pick the size with the best fill of its rounds, then the most resident waves.
static int choose_wpb(const amx_model *m, int nenv) {
  if (const char *force = getenv("AMX_FORCE_WPB")) { int w = atoi(force); if (w >= 1 && w <= m->dev.wpb) return w; }
  const int epw = AMX_WAVE / m->dev.G, nwaves = (nenv + epw - 1) / epw, ncu = m->ncu > 0 ? m->ncu : 256;
  int best = 1; double best_fill = -1; int best_res = 0;
  for (int w = 1; w <= m->dev.wpb; w++) {
    const int bytes = (m->dev.nconst + w * epw * m->dev.env_floats) * (int)sizeof(float);
    const int per_cu = std::min((160 * 1024 / bytes) * w, AMX_MIN_WAVES_PER_SIMD * 4);
    if (per_cu < w) continue;
    const long res = (long)(per_cu / w) * w * ncu;
    const long rounds = (nwaves + res - 1) / res;
    const double fill = (double)nwaves / (double)(rounds * res);
    if (fill > best_fill + 1e-9 || (fill > best_fill - 1e-9 && per_cu >= best_res)) { best_fill = fill; best = w; best_res = per_cu; }
  }
  return best;
}
static int pre_launch(const amx_model *m, int nenv, dim3 *grid, int *wpb, int *lds_bytes) {
  *wpb = choose_wpb(m, nenv);
  const int per_block = *wpb * (AMX_WAVE / m->dev.G);
  *lds_bytes = (m->dev.nconst + per_block * m->dev.env_floats) * (int)sizeof(float);
  *grid = dim3((nenv + per_block - 1) / per_block);
  return AMX_OK;
}
// one instantiation per group width; the model picks its own at creation
#define AMX_LAUNCH_G(kernel, m, grid, stream, ...)                                                                       \
  do {                                                                                                                   \
    if ((m)->dev.G == 16) hipLaunchKernelGGL((kernel<16, DynSpec>), grid, dim3(AMX_WAVE * wpb), lds_bytes, (hipStream_t)stream, __VA_ARGS__);      \
    else if ((m)->dev.G == 32) hipLaunchKernelGGL((kernel<32, DynSpec>), grid, dim3(AMX_WAVE * wpb), lds_bytes, (hipStream_t)stream, __VA_ARGS__); \
    else hipLaunchKernelGGL((kernel<64, DynSpec>), grid, dim3(AMX_WAVE * wpb), lds_bytes, (hipStream_t)stream, __VA_ARGS__);                       \
  } while (0)

extern "C" int amx_forward(const amx_model *m, int nenv, const float *qpos, const float *qvel, const float *ctrl,
                           const float *warm, const amx_forward_out *o, void *stream) {
  if (!m || !qpos || !qvel || (m->dev.nu > 0 && !ctrl) || nenv < 0) return fail(AMX_EINVAL, "amx_forward: bad argument");
  if (nenv == 0) return AMX_OK;
  ForwardOut fo;
  memset(&fo, 0, sizeof(fo));
  if (o) {
    fo.qacc = o->qacc_dev; fo.qacc_smooth = o->qacc_smooth_dev; fo.qfrc_bias = o->qfrc_bias_dev;
    fo.qfrc_constraint = o->qfrc_constraint_dev; fo.qM = o->qM_dev; fo.xpos = o->xpos_dev; fo.xquat = o->xquat_dev;
    fo.cdist = o->contact_dist_dev; fo.nefc = o->nefc_dev;
  }
  AMX_ON_DEVICE(m->device);
  dim3 grid;
  int wpb, lds_bytes;
  if (int rc = pre_launch(m, nenv, &grid, &wpb, &lds_bytes)) return rc;
  if (!for_spec(m->spec, [&](auto sp) {
        using S = decltype(sp);
        hipLaunchKernelGGL((amx_forward_kernel<S::G, S>), grid, dim3(AMX_WAVE * wpb), lds_bytes, (hipStream_t)stream, m->dev, nenv, qpos, qvel, ctrl, warm, fo);
      }))
    AMX_LAUNCH_G(amx_forward_kernel, m, grid, stream, m->dev, nenv, qpos, qvel, ctrl, warm, fo);
  HIP_TRY(hipGetLastError());
  return AMX_OK;
}

extern "C" int amx_step(const amx_model *m, int nenv, int nsub, float *qpos, float *qvel, const float *ctrl, float *warm,
                        float *time, void *stream) {
  if (!m || !qpos || !qvel || (m->dev.nu > 0 && !ctrl) || nenv < 0 || nsub < 0) return fail(AMX_EINVAL, "amx_step: bad argument");
  if (nenv == 0) return AMX_OK;
  AMX_ON_DEVICE(m->device);
  dim3 grid;
  int wpb, lds_bytes;
  if (int rc = pre_launch(m, nenv, &grid, &wpb, &lds_bytes)) return rc;
  if (!for_spec(m->spec, [&](auto sp) {
        using S = decltype(sp);
        hipLaunchKernelGGL((amx_step_kernel<S::G, S>), grid, dim3(AMX_WAVE * wpb), lds_bytes, (hipStream_t)stream, m->dev, nenv, nsub, qpos, qvel, ctrl, warm, time);
      }))
    AMX_LAUNCH_G(amx_step_kernel, m, grid, stream, m->dev, nenv, nsub, qpos, qvel, ctrl, warm, time);
  HIP_TRY(hipGetLastError());
  return AMX_OK;
}

#ifdef AMX_PROFILE
static float *g_prof = nullptr;
// Diagnostic build only (libamx_prof.so): where a rollout's cycles go.  Not part of include/amx.h.
extern "C" int amx_prof_set(float *dev_counters /*[nenv,24]*/) { g_prof = dev_counters; return 0; }
#endif

extern "C" int amx_rollout(const amx_model *m, int nenv, int nsteps, const float *x0, int x0_batched, const float *us, float *xs,
                           float *xfinal, void *stream) {
  if (!m || !x0 || nenv < 0 || nsteps < 0 || (nsteps > 0 && m->dev.nu > 0 && !us)) return fail(AMX_EINVAL, "amx_rollout: bad argument");
  if (nenv == 0) return AMX_OK;
  AMX_ON_DEVICE(m->device);
  dim3 grid;
  int wpb, lds_bytes;
  if (int rc = pre_launch(m, nenv, &grid, &wpb, &lds_bytes)) return rc;
  float *prof = nullptr;
#ifdef AMX_PROFILE
  prof = g_prof;
#endif
  if (!for_spec(m->spec, [&](auto sp) {
        using S = decltype(sp);
        hipLaunchKernelGGL((amx_rollout_kernel<S::G, S>), grid, dim3(AMX_WAVE * wpb), lds_bytes, (hipStream_t)stream, m->dev, nenv, nsteps, x0_batched, x0, us, xs, xfinal, prof);
      }))
    AMX_LAUNCH_G(amx_rollout_kernel, m, grid, stream, m->dev, nenv, nsteps, x0_batched, x0, us, xs, xfinal, prof);
  HIP_TRY(hipGetLastError());
  return AMX_OK;
}

extern "C" int amx_quadcost(int nenv, int nsteps, int nx, int nu, const float *xs, const float *us, const float *Q, const float *Qf,
                            const float *R, const float *xg, float *cost, void *stream) {
  if (!xs || !Q || !Qf || !xg || !cost || (nu > 0 && (!us || !R)) || nenv < 0 || nsteps < 0) return fail(AMX_EINVAL, "amx_quadcost: bad argument");
  if (nenv == 0) return AMX_OK;
  // this entry has no model handle: run on the device that owns the trajectories
  int xdev = -1;
  {
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, xs) == hipSuccess) xdev = attr.device;
    else (void)hipGetLastError();
    if (xdev < 0 && hipGetDevice(&xdev) != hipSuccess) return fail(AMX_EHIP, "amx_quadcost: no current HIP device");
  }
  AMX_ON_DEVICE(xdev);
  // tiled matrix-core kernel when Q, R and four 32-row tiles fit in LDS; the entry-per-lane kernel otherwise
  const int nbx = (nx + 31) / 32, nbu = (nu + 31) / 32, ldy = 32 * std::max(nbx, std::max(nbu, 1)) + 5;
  const size_t lds_floats = (size_t)((nx + 3) & ~3) * 32 * nbx + (size_t)((nu + 3) & ~3) * 32 * nbu + 32 * nbx + (size_t)QC_WAVES * 32 * ldy;
  if (nx >= 1 && nbx <= QC_MAXBLK && nbu <= QC_MAXBLK && lds_floats * sizeof(float) <= 64 * 1024) {
    hipLaunchKernelGGL(amx_quadcost_mfma_kernel, dim3((nenv + QC_WAVES - 1) / QC_WAVES), dim3(AMX_WAVE * QC_WAVES), lds_floats * sizeof(float), (hipStream_t)stream,
                       nenv, nsteps, nx, nu, xs, us, Q, Qf, R, xg, cost);
  } else {
    hipLaunchKernelGGL(amx_quadcost_kernel, dim3(nenv), dim3(AMX_WAVE), 0, (hipStream_t)stream, nenv, nsteps, nx, nu, xs, us, Q, Qf, R, xg, cost);
  }
  HIP_TRY(hipGetLastError());
  return AMX_OK;
}

// ---- C4: forward-difference gradient of the terminal cost over the controls
static size_t fd_us_bytes(const amx_model *m, int nsamples, int nsteps) {
  const size_t b = (size_t)nsamples * (size_t)nsteps * (size_t)m->dev.nu * sizeof(float);
  return (b + 255) & ~(size_t)255;
}
extern "C" int amx_fd_workspace_bytes(const amx_model *m, int nsamples, int nsteps, size_t *out) {
  if (!m || !out || nsamples < 1 || nsteps < 0) return fail(AMX_EINVAL, "amx_fd_workspace_bytes: bad argument");
  *out = fd_us_bytes(m, nsamples, nsteps) + (size_t)nsamples * (size_t)(m->dev.nq + m->dev.nv) * sizeof(float);
  return AMX_OK;
}
extern "C" int amx_fd_grad(const amx_model *m, int nsamples, int nsteps, const float *x0, const float *us_nominal, const float *eps, int coord0,
                           const float *Qf, const float *xg, float *J, float *grad, float *xfinal, void *workspace, size_t workspace_bytes,
                           void *stream) {
  if (!m || !x0 || !Qf || !xg || !J || !workspace || nsamples < 1 || nsteps < 0 || coord0 < 0 ||
      (nsteps > 0 && m->dev.nu > 0 && (!us_nominal || !eps)))
    return fail(AMX_EINVAL, "amx_fd_grad: bad argument");
  size_t need = 0;
  if (int rc = amx_fd_workspace_bytes(m, nsamples, nsteps, &need)) return rc;
  if (workspace_bytes < need) return fail(AMX_EINVAL, "amx_fd_grad: workspace of %zu bytes, %zu needed", workspace_bytes, need);
  AMX_ON_DEVICE(m->device);
  float *us = (float *)workspace;
  float *xf = (float *)((char *)workspace + fd_us_bytes(m, nsamples, nsteps));
  const int nx = m->dev.nq + m->dev.nv, nu = m->dev.nu;
  const size_t total = (size_t)nsamples * nsteps * nu;
  if (total > 0) {
    const int blocks = (int)std::min<size_t>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(amx_fd_perturb_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, nsamples, nsteps, nu, coord0, us_nominal, eps, us);
    HIP_TRY(hipGetLastError());
  }
  if (int rc = amx_rollout(m, nsamples, nsteps, x0, /*x0_batched=*/0, us, /*xs=*/nullptr, xf, stream)) return rc;
  hipLaunchKernelGGL(amx_fd_cost_kernel, dim3(nsamples), dim3(AMX_WAVE), 0, (hipStream_t)stream, nsamples, nsteps, nx, nu, coord0, xf, Qf, xg, eps, J, grad);
  HIP_TRY(hipGetLastError());
  if (xfinal) HIP_TRY(hipMemcpyAsync(xfinal, xf, (size_t)nsamples * nx * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return AMX_OK;
}
