// amx_api.cpp — host side of the C ABI in include/amx.h: model upload, launch plan, kernel launches.
// No physics here (that is amx_kernels.hip) and no CPU fallback: every entry point that computes
// launches a HIP kernel or fails with AMX_EHIP.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/amx.h"
#include "amx_dev.h"

#include "amx_kernels.hip"  // single translation unit: kernels + launchers, no relocatable device code

static thread_local char g_err[512] = "";
#ifdef AMX_PROFILE
static float *g_prof = nullptr;
#endif
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

struct amx_model {
  DevModel dev;
  LdsPlan lds;
  int device;
  int *d_ip;
  float *d_fp;
  int ncon, nefc, lds_bytes;
};

namespace {
struct Pools {
  std::vector<int> ip;
  std::vector<float> fp;
  int addi(const std::vector<int> &v) {
    int off = (int)ip.size();
    ip.insert(ip.end(), v.begin(), v.end());
    if (v.empty()) ip.push_back(0);
    return off;
  }
  int addf(const std::vector<float> &v) {
    int off = (int)fp.size();
    fp.insert(fp.end(), v.begin(), v.end());
    if (v.empty()) fp.push_back(0.f);
    return off;
  }
  int addi(const int *p, size_t n) { return addi(std::vector<int>(p, p + n)); }
  int addd(const double *p, size_t n) {
    std::vector<float> v(n);
    for (size_t i = 0; i < n; i++) v[i] = (float)p[i];
    return addf(v);
  }
};
const int PAIR_NCON[6] = {1, 2, 4, 1, 1, 1};
}  // namespace

extern "C" int amx_version(void) { return AMX_VERSION; }
extern "C" const char *amx_last_error(void) { return g_err; }

extern "C" int amx_model_create(const amx_model_desc *d, int device, amx_model **out) {
  if (!d || !out) return fail(AMX_EINVAL, "null argument");
  if (d->version != AMX_VERSION) return fail(AMX_EINVAL, "amx_model_desc version %d, library %d", d->version, AMX_VERSION);
  if (d->nbody < 1 || d->nq < 0 || d->nv < 0) return fail(AMX_EINVAL, "bad sizes");
  if (d->integrator != 0) return fail(AMX_EUNSUPPORTED, "only the Euler integrator is supported (integrator=%d)", d->integrator);
  if (d->solver != 1 && d->solver != 2) return fail(AMX_EUNSUPPORTED, "solver must be CG (1) or Newton (2), got %d", d->solver);
  if (d->nv > AMX_MAX_NV) return fail(AMX_ECAPACITY, "nv=%d exceeds the register-row factorisation capacity %d", d->nv, AMX_MAX_NV);
  if (d->nv < 1) return fail(AMX_EUNSUPPORTED, "model has no degrees of freedom");
  const int nb = d->nbody, nv = d->nv, nq = d->nq, nu = d->nu, nj = d->njnt;

  Pools P;
  DevModel m;
  memset(&m, 0, sizeof(m));
  m.nq = nq; m.nv = nv; m.nu = nu; m.nb = nb; m.njnt = nj; m.neq = 0; m.npair = d->npair;
  m.iterations = d->iterations; m.ls_iterations = d->ls_iterations; m.solver = d->solver; m.disableflags = d->disableflags;
  m.dt = (float)d->timestep; m.gx = (float)d->gravity[0]; m.gy = (float)d->gravity[1]; m.gz = (float)d->gravity[2];
  m.tolerance = (float)d->tolerance; m.ls_tolerance = (float)d->ls_tolerance; m.impratio = (float)d->impratio;
  m.scale = (float)(d->meaninertia * (nv > 1 ? nv : 1));
  m.ldm = nv | 1;
  m.ldj = nv | 1;
  m.nchol = nv;

  // ---- tree structure: depth levels, children, roots
  std::vector<int> depth(nb, 0), root_slot(nb, 0), roots;
  for (int b = 1; b < nb; b++) {
    int p = d->body_parentid[b];
    if (p < 0 || p >= b) return fail(AMX_EINVAL, "body %d: parent %d must precede it", b, p);
    depth[b] = depth[p] + 1;
    if (p == 0) { root_slot[b] = (int)roots.size(); roots.push_back(b); }
    else root_slot[b] = root_slot[p];
  }
  int nlevel = *std::max_element(depth.begin(), depth.end()) + 1;
  std::vector<int> level_adr(nlevel + 1, 0), level_body;
  for (int L = 0; L < nlevel; L++) {
    level_adr[L] = (int)level_body.size();
    for (int b = 0; b < nb; b++) if (depth[b] == L) level_body.push_back(b);
  }
  level_adr[nlevel] = (int)level_body.size();
  std::vector<int> child_adr(nb + 1, 0), child_list;
  for (int b = 0; b < nb; b++) {
    child_adr[b] = (int)child_list.size();
    if (b > 0) for (int c = b + 1; c < nb; c++) if (d->body_parentid[c] == b) child_list.push_back(c);
  }
  child_adr[nb] = (int)child_list.size();
  m.nlevel = nlevel; m.nroot = (int)roots.size();
  m.body_parent = P.addi(d->body_parentid, nb);
  m.body_root = P.addi(root_slot);
  m.body_jntadr = P.addi(d->body_jntadr, nb); m.body_jntnum = P.addi(d->body_jntnum, nb);
  m.body_dofadr = P.addi(d->body_dofadr, nb); m.body_dofnum = P.addi(d->body_dofnum, nb);
  m.level_adr = P.addi(level_adr); m.level_body = P.addi(level_body);
  m.child_adr = P.addi(child_adr); m.child_list = P.addi(child_list);
  m.root_body = P.addi(roots);
  m.body_pos = P.addd(d->body_pos, nb * 3); m.body_quat = P.addd(d->body_quat, nb * 4);
  m.body_ipos = P.addd(d->body_ipos, nb * 3); m.body_iquat = P.addd(d->body_iquat, nb * 4);
  m.body_mass = P.addd(d->body_mass, nb); m.body_inertia = P.addd(d->body_inertia, nb * 3);

  // ---- joints / dofs
  m.jnt_type = P.addi(d->jnt_type, nj); m.jnt_qadr = P.addi(d->jnt_qposadr, nj); m.jnt_dadr = P.addi(d->jnt_dofadr, nj);
  m.jnt_body = P.addi(d->jnt_bodyid, nj);
  m.jnt_pos = P.addd(d->jnt_pos, nj * 3); m.jnt_axis = P.addd(d->jnt_axis, nj * 3); m.jnt_stiff = P.addd(d->jnt_stiffness, nj);
  m.qpos0 = P.addd(d->qpos0, nq); m.qpos_spring = P.addd(d->qpos_spring, nq);
  std::vector<int> dof_kind(nv), dof_axis(nv);
  for (int i = 0; i < nv; i++) {
    int j = d->dof_jntid[i], k = i - d->jnt_dofadr[j], t = d->jnt_type[j];
    if (t == 0) { dof_kind[i] = k < 3 ? 0 : 1; dof_axis[i] = k % 3; }
    else if (t == 1) { dof_kind[i] = 1; dof_axis[i] = k; }
    else if (t == 3) { dof_kind[i] = 2; dof_axis[i] = 0; }
    else { dof_kind[i] = 3; dof_axis[i] = 0; }
    if (t <= 1 && d->jnt_stiffness[j] != 0) return fail(AMX_EUNSUPPORTED, "spring on free/ball joint %d", j);
  }
  m.dof_body = P.addi(d->dof_bodyid, nv); m.dof_jnt = P.addi(d->dof_jntid, nv);
  m.dof_kind = P.addi(dof_kind); m.dof_axis = P.addi(dof_axis);
  m.dof_armature = P.addd(d->dof_armature, nv); m.dof_damping = P.addd(d->dof_damping, nv);
  std::vector<int> Mi, Mj;
  for (int i = 0; i < nv; i++)
    for (int j = i; j >= 0; j = d->dof_parentid[j]) { Mi.push_back(i); Mj.push_back(j); }
  m.nM = (int)Mi.size(); m.M_i = P.addi(Mi); m.M_j = P.addi(Mj);

  // ---- actuators
  std::vector<int> act_qadr(nu), act_dadr(nu), act_flags(nu), dof_actadr(nv + 1, 0), dof_actlist;
  std::vector<float> act_gain(nu);
  for (int u = 0; u < nu; u++) {
    int j = d->actuator_trnid[u];
    if (j < 0 || j >= nj || d->jnt_type[j] < 2) return fail(AMX_EUNSUPPORTED, "actuator %d must drive a hinge or slide joint", u);
    act_qadr[u] = d->jnt_qposadr[j]; act_dadr[u] = d->jnt_dofadr[j];
    act_flags[u] = (d->actuator_ctrllimited[u] ? 1 : 0) | (d->actuator_forcelimited[u] ? 2 : 0) | (d->actuator_biastype[u] == 1 ? 4 : 0);
    act_gain[u] = (float)d->actuator_gainprm[3 * u];
  }
  for (int i = 0; i < nv; i++) {
    dof_actadr[i] = (int)dof_actlist.size();
    for (int u = 0; u < nu; u++) if (act_dadr[u] == i) dof_actlist.push_back(u);
  }
  dof_actadr[nv] = (int)dof_actlist.size();
  m.act_qadr = P.addi(act_qadr); m.act_dadr = P.addi(act_dadr); m.act_flags = P.addi(act_flags);
  m.dof_actadr = P.addi(dof_actadr); m.dof_actlist = P.addi(dof_actlist);
  m.act_gear = P.addd(d->actuator_gear, nu); m.act_gain = P.addf(act_gain); m.act_bias = P.addd(d->actuator_biasprm, nu * 3);
  m.act_ctrlrange = P.addd(d->actuator_ctrlrange, nu * 2); m.act_forcerange = P.addd(d->actuator_forcerange, nu * 2);

  // ---- collision geoms (only those that appear in a pair), pairs, contact slots
  const int np = d->npair;
  std::vector<int> gmap(d->ngeom > 0 ? d->ngeom : 1, -1), gc_body, gc_type;
  std::vector<float> gc_pos, gc_quat, gc_size;
  auto use_geom = [&](int g) {
    if (gmap[g] < 0) {
      gmap[g] = (int)gc_body.size();
      gc_body.push_back(d->geom_bodyid[g]); gc_type.push_back(d->geom_type[g]);
      for (int k = 0; k < 3; k++) { gc_pos.push_back((float)d->geom_pos[3 * g + k]); gc_size.push_back((float)d->geom_size[3 * g + k]); }
      for (int k = 0; k < 4; k++) gc_quat.push_back((float)d->geom_quat[4 * g + k]);
    }
    return gmap[g];
  };
  std::vector<int> pg1(np), pg2(np), pfn(np), pcon(np), pb1(np), pb2(np), pr1(np), pr2(np), pdofadr(np + 1, 0), pdoflist, pdofsign;
  std::vector<int> con_pair, con_slot;
  std::vector<float> pmu(np), pinvw(np);
  int ncon = 0;
  for (int p = 0; p < np; p++) {
    int g1 = d->pair_geom1[p], g2 = d->pair_geom2[p], fn = d->pair_fn[p];
    if (fn < 0 || fn > 5) return fail(AMX_EUNSUPPORTED, "pair %d: unknown collision function %d", p, fn);
    if (fn == 2 && (d->pair_vertnum[p] < 1 || d->pair_vertadr[p] < 0 || d->pair_vertadr[p] + d->pair_vertnum[p] > d->nconvexvert))
      return fail(AMX_EINVAL, "pair %d: bad convex vertex range", p);
    pg1[p] = use_geom(g1); pg2[p] = use_geom(g2); pfn[p] = fn; pcon[p] = ncon;
    int b1 = d->geom_bodyid[g1], b2 = d->geom_bodyid[g2];
    pb1[p] = b1; pb2[p] = b2; pr1[p] = root_slot[b1]; pr2[p] = root_slot[b2];
    pmu[p] = (float)d->pair_friction[3 * p];
    pinvw[p] = (float)(d->body_invweight0[2 * b1] + d->body_invweight0[2 * b2]);
    for (int k = 0; k < PAIR_NCON[fn]; k++) { con_pair.push_back(p); con_slot.push_back(k); }
    ncon += PAIR_NCON[fn];
    // union of the ancestor dof chains of both bodies
    std::vector<int> sign(nv, 0);
    for (int side = 0; side < 2; side++) {
      int b = side == 0 ? b2 : b1;
      while (b > 0 && d->body_dofnum[b] == 0) b = d->body_parentid[b];
      if (b == 0) continue;
      for (int i = d->body_dofadr[b] + d->body_dofnum[b] - 1; i >= 0; i = d->dof_parentid[i]) sign[i] |= (side == 0 ? 1 : 2);
    }
    pdofadr[p] = (int)pdoflist.size();
    for (int i = 0; i < nv; i++) if (sign[i]) { pdoflist.push_back(i); pdofsign.push_back(sign[i]); }
  }
  pdofadr[np] = (int)pdoflist.size();
  m.ngc = (int)gc_body.size(); m.ncon = ncon;
  m.gc_body = P.addi(gc_body); m.gc_type = P.addi(gc_type);
  m.gc_pos = P.addf(gc_pos); m.gc_quat = P.addf(gc_quat); m.gc_size = P.addf(gc_size);
  m.pair_g1 = P.addi(pg1); m.pair_g2 = P.addi(pg2); m.pair_fn = P.addi(pfn); m.pair_conadr = P.addi(pcon);
  m.pair_vertadr = P.addi(d->pair_vertadr, np); m.pair_vertnum = P.addi(d->pair_vertnum, np);
  m.pair_body1 = P.addi(pb1); m.pair_body2 = P.addi(pb2); m.pair_root1 = P.addi(pr1); m.pair_root2 = P.addi(pr2);
  m.pair_dofadr = P.addi(pdofadr); m.pair_doflist = P.addi(pdoflist); m.pair_dofsign = P.addi(pdofsign);
  m.con_pair = P.addi(con_pair); m.con_slot = P.addi(con_slot);
  m.pair_mu = P.addf(pmu); m.pair_invw = P.addf(pinvw);
  m.pair_solref = P.addd(d->pair_solref, np * 2); m.pair_solimp = P.addd(d->pair_solimp, np * 5);
  m.pair_margin = P.addd(d->pair_margin, np);
  m.convex_vert = P.addd(d->convex_vert, (size_t)d->nconvexvert * 3);

  // ---- limit rows and equality rows
  std::vector<int> lim_qadr, lim_dadr;
  std::vector<float> lim_range, lim_margin, lim_solref, lim_solimp, lim_invw;
  for (int j = 0; j < nj; j++) {
    if (!d->jnt_limited[j]) continue;
    if (d->jnt_type[j] < 2) return fail(AMX_EUNSUPPORTED, "limits on free/ball joint %d", j);
    lim_qadr.push_back(d->jnt_qposadr[j]); lim_dadr.push_back(d->jnt_dofadr[j]);
    lim_range.push_back((float)d->jnt_range[2 * j]); lim_range.push_back((float)d->jnt_range[2 * j + 1]);
    lim_margin.push_back((float)d->jnt_margin[j]);
    for (int k = 0; k < 2; k++) lim_solref.push_back((float)d->jnt_solref[2 * j + k]);
    for (int k = 0; k < 5; k++) lim_solimp.push_back((float)d->jnt_solimp[5 * j + k]);
    lim_invw.push_back((float)d->dof_invweight0[d->jnt_dofadr[j]]);
  }
  m.nlim = (int)lim_qadr.size();
  m.lim_qadr = P.addi(lim_qadr); m.lim_dadr = P.addi(lim_dadr);
  m.lim_range = P.addf(lim_range); m.lim_margin = P.addf(lim_margin); m.lim_solref = P.addf(lim_solref);
  m.lim_solimp = P.addf(lim_solimp); m.lim_invw = P.addf(lim_invw);
  std::vector<int> eq_q1, eq_d1, eq_q2, eq_d2;
  std::vector<float> eq_data, eq_solref, eq_solimp, eq_invw1, eq_invw2;
  for (int e = 0; e < d->neq; e++) {
    if (!d->eq_active[e]) continue;
    int j1 = d->eq_obj1id[e], j2 = d->eq_obj2id[e];
    if (j1 < 0 || j1 >= nj || d->jnt_type[j1] < 2 || (j2 >= 0 && (j2 >= nj || d->jnt_type[j2] < 2)))
      return fail(AMX_EUNSUPPORTED, "equality %d must couple hinge/slide joints", e);
    eq_q1.push_back(d->jnt_qposadr[j1]); eq_d1.push_back(d->jnt_dofadr[j1]);
    eq_q2.push_back(j2 >= 0 ? d->jnt_qposadr[j2] : -1); eq_d2.push_back(j2 >= 0 ? d->jnt_dofadr[j2] : -1);
    for (int k = 0; k < 5; k++) { eq_data.push_back((float)d->eq_data[5 * e + k]); eq_solimp.push_back((float)d->eq_solimp[5 * e + k]); }
    for (int k = 0; k < 2; k++) eq_solref.push_back((float)d->eq_solref[2 * e + k]);
    eq_invw1.push_back((float)d->dof_invweight0[d->jnt_dofadr[j1]]);
    eq_invw2.push_back(j2 >= 0 ? (float)d->dof_invweight0[d->jnt_dofadr[j2]] : 0.f);
  }
  m.neq = (int)eq_q1.size();
  m.eq_q1 = P.addi(eq_q1); m.eq_d1 = P.addi(eq_d1); m.eq_q2 = P.addi(eq_q2); m.eq_d2 = P.addi(eq_d2);
  m.eq_data = P.addf(eq_data); m.eq_solref = P.addf(eq_solref); m.eq_solimp = P.addf(eq_solimp);
  m.eq_invw1 = P.addf(eq_invw1); m.eq_invw2 = P.addf(eq_invw2);
  const bool con_off = d->disableflags & 1;
  m.nefc = con_off ? 0 : m.neq + m.nlim + 4 * m.ncon;
  if (d->disableflags & 16) { /* contact disabled: the compiler emits no pairs; keep rows consistent */ }

  // ---- LDS plan (floats)
  LdsPlan L;
  memset(&L, 0, sizeof(L));
  int off = 0;
  auto take = [&](int n) { int o = off; off += (n > 0 ? n : 1); return o; };
  L.qpos = take(nq); L.qvel = take(nv); L.ctrl = take(nu); L.warm = take(nv);
  L.xpos = take(3 * nb); L.xquat = take(4 * nb); L.com = take(3 * m.nroot); L.xanchor = take(3 * nj); L.xaxis = take(3 * nj);
  L.cinert = take(10 * nb); L.crb = take(10 * nb); L.cdof = take(6 * nv); L.fcrb = take(6 * nv);
  L.cvel = take(6 * nb); L.cacc = take(6 * nb); L.cfrc = take(6 * nb);
  L.M = take(nv * m.ldm); L.H = take(nv * m.ldm); L.Lb = take(nv * m.ldm);
  L.fsmooth = take(nv); L.asmooth = take(nv); L.qacc = take(nv); L.Ma = take(nv); L.grad = take(nv); L.Mgrad = take(nv);
  L.search = take(nv); L.mv = take(nv); L.fcon = take(nv); L.tmp = take(nv); L.fbias = take(nv);
  L.actf = take(nu);
  L.D = take(m.nefc); L.aref = take(m.nefc); L.Jaref = take(m.nefc); L.jv = take(m.nefc); L.J = take(m.nefc * m.ldj);
  L.cdist = take(m.ncon); L.cpos = take(3 * m.ncon); L.cframe = take(9 * np); L.gpos = take(3 * m.ngc); L.gmat = take(9 * m.ngc);
  L.prof = take(24);
  L.total = off;
  const int lds_bytes = off * (int)sizeof(float);
  if (lds_bytes > 160 * 1024) return fail(AMX_ECAPACITY, "model needs %d bytes of LDS per environment (limit 163840)", lds_bytes);

  amx_model *h = new (std::nothrow) amx_model();
  if (!h) return fail(AMX_ENOMEM, "out of host memory");
  h->device = device; h->ncon = m.ncon; h->nefc = m.nefc; h->lds_bytes = lds_bytes; h->lds = L;
  h->d_ip = nullptr; h->d_fp = nullptr;
  hipError_t e;
  if ((e = hipSetDevice(device)) != hipSuccess || (e = hipMalloc((void **)&h->d_ip, P.ip.size() * sizeof(int))) != hipSuccess ||
      (e = hipMalloc((void **)&h->d_fp, P.fp.size() * sizeof(float))) != hipSuccess ||
      (e = hipMemcpy(h->d_ip, P.ip.data(), P.ip.size() * sizeof(int), hipMemcpyHostToDevice)) != hipSuccess ||
      (e = hipMemcpy(h->d_fp, P.fp.data(), P.fp.size() * sizeof(float), hipMemcpyHostToDevice)) != hipSuccess ||
      (e = hipFuncSetAttribute((const void *)amx_env_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes)) != hipSuccess) {
    if (h->d_ip) (void)hipFree(h->d_ip);
    if (h->d_fp) (void)hipFree(h->d_fp);
    delete h;
    return fail(AMX_EHIP, "model upload failed: %s", hipGetErrorString(e));
  }
  m.ip = h->d_ip; m.fp = h->d_fp;
  h->dev = m;
  *out = h;
  return AMX_OK;
}

extern "C" int amx_model_destroy(amx_model *m) {
  if (!m) return AMX_OK;
  (void)hipSetDevice(m->device);
  (void)hipFree(m->d_ip);
  (void)hipFree(m->d_fp);
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
  else if (n == "lanes_per_env") *out = AMX_WAVE;
  else if (n == "envs_per_block") *out = 1;
  else return fail(AMX_EINVAL, "unknown query '%s'", name);
  return AMX_OK;
}

static int launch(const amx_model *m, const LaunchArgs &a, void *stream) {
  if (a.nenv <= 0) return AMX_OK;
  HIP_TRY(hipSetDevice(m->device));
  hipLaunchKernelGGL(amx_env_kernel, dim3(a.nenv), dim3(AMX_WAVE), m->lds_bytes, (hipStream_t)stream, m->dev, m->lds, a);
  HIP_TRY(hipGetLastError());
  return AMX_OK;
}

extern "C" int amx_forward(const amx_model *m, int nenv, const float *qpos, const float *qvel, const float *ctrl,
                           const float *warm, const amx_forward_out *o, void *stream) {
  if (!m || !qpos || !qvel || (m->dev.nu > 0 && !ctrl) || nenv < 0) return fail(AMX_EINVAL, "amx_forward: bad argument");
  LaunchArgs a;
  memset(&a, 0, sizeof(a));
  a.mode = AMX_MODE_FORWARD; a.nenv = nenv; a.nsteps = 0;
  a.qpos = const_cast<float *>(qpos); a.qvel = const_cast<float *>(qvel); a.us = ctrl; a.warm_in = warm;
  if (o) {
    a.o_qacc = o->qacc_dev; a.o_qacc_smooth = o->qacc_smooth_dev; a.o_qfrc_bias = o->qfrc_bias_dev;
    a.o_qfrc_constraint = o->qfrc_constraint_dev; a.o_qM = o->qM_dev; a.o_xpos = o->xpos_dev; a.o_xquat = o->xquat_dev;
    a.o_cdist = o->contact_dist_dev; a.o_nefc = o->nefc_dev;
  }
  return launch(m, a, stream);
}

extern "C" int amx_step(const amx_model *m, int nenv, int nsub, float *qpos, float *qvel, const float *ctrl, float *warm,
                        float *time, void *stream) {
  if (!m || !qpos || !qvel || (m->dev.nu > 0 && !ctrl) || nenv < 0 || nsub < 0) return fail(AMX_EINVAL, "amx_step: bad argument");
  LaunchArgs a;
  memset(&a, 0, sizeof(a));
  a.mode = AMX_MODE_STEP; a.nenv = nenv; a.nsteps = nsub;
  a.qpos = qpos; a.qvel = qvel; a.us = ctrl; a.warm = warm; a.time = time;
  return launch(m, a, stream);
}

extern "C" int amx_rollout(const amx_model *m, int nenv, int nsteps, const float *x0, int x0_batched, const float *us, float *xs,
                           float *xfinal, void *stream) {
  if (!m || !x0 || nenv < 0 || nsteps < 0 || (nsteps > 0 && m->dev.nu > 0 && !us)) return fail(AMX_EINVAL, "amx_rollout: bad argument");
  LaunchArgs a;
  memset(&a, 0, sizeof(a));
  a.mode = AMX_MODE_ROLLOUT; a.nenv = nenv; a.nsteps = nsteps; a.x0_batched = x0_batched;
  a.x0 = x0; a.us = us; a.xs = xs; a.xfinal = xfinal;
#ifdef AMX_PROFILE
  a.prof = g_prof;
#endif
  return launch(m, a, stream);
}

#ifdef AMX_PROFILE
// Diagnostic build only (libamx_prof.so): where a rollout's cycles go.  Not part of include/amx.h.
extern "C" int amx_prof_set(float *dev_counters /*[nenv,24]*/) { g_prof = dev_counters; return 0; }
#endif

extern "C" int amx_quadcost(int nenv, int nsteps, int nx, int nu, const float *xs, const float *us, const float *Q, const float *Qf,
                            const float *R, const float *xg, float *cost, void *stream) {
  if (!xs || !Q || !Qf || !xg || !cost || (nu > 0 && (!us || !R)) || nenv < 0 || nsteps < 0) return fail(AMX_EINVAL, "amx_quadcost: bad argument");
  if (nenv == 0) return AMX_OK;
  hipLaunchKernelGGL(amx_quadcost_kernel, dim3(nenv), dim3(AMX_WAVE), 0, (hipStream_t)stream, nenv, nsteps, nx, nu, xs, us, Q, Qf, R, xg, cost);
  HIP_TRY(hipGetLastError());
  return AMX_OK;
}
