// amx_plan.h - host-only planner of the engine: amx_model_desc -> constant image + launch plan (DevModel).
// No HIP in here.  Included by amx_api.cpp (the product library) and built on its own as libamx_plan.so (Makefile), which
// tools/gen_specs.py loads to write amx_specs_gen.h - so the specialisation header can be regenerated before the kernels
// that include it are compiled.
#pragma once
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

static thread_local char g_err[512] = "";
static int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
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
const int PAIR_NCON[12] = {1, 2, 4, 1, 1, 1, 4, 1, 1, 1, 1, 2};  // 8, 9: one row group each (connect / weld-translation, weld-rotation); 10: a ball joint's limit row; 11: capsule - convex

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
  im.setf(off, k); im.setf(off + 1, b); im.setf(off + 2, dmin); im.setf(off + 3, dmax); im.setf(off + 4, 1.0 / width);  // the kernel multiplies
  im.setf(off + 5, mid); im.setf(off + 6, power);
  im.setf(off + 7, 1.0 / pow(mid, power - 1)); im.setf(off + 8, 1.0 / pow(1 - mid, power - 1));
}
}  // namespace


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
// column of the elimination's register row that register k of a pair's block stands for (mfma2; dof, nv = the gradient, -1 none): compile-time
// data of a specialisation record (amx_model_plan_describe prints it), not part of the device model
static thread_local std::vector<int> p2_col;
// > 0: plan the model on a lane group at least this wide (amx_plan_force_lanes; amx_model_create plans a 32-lane model a second time
// at 64 lanes - one environment per wavefront - for batches too small to fill the SIMDs with two-environment waves)
static thread_local int plan_force_lanes = 0;
extern "C" void amx_plan_force_lanes(int g) { plan_force_lanes = g; }
// 1: plan the DENSE form (r05; plan_model tries it for a model whose contact rows alone ask for 64 lanes): two contact rows per lane on a
// 32-lane group, M stored compactly (mcomp) and the solver's row records and staging vector in the space of the contact points, frames and
// geom poses, which nobody reads after make_constraint - what it takes to keep sixteen environments of a contact-rich biped in a CU's LDS
static thread_local int plan_dense = 0;
static thread_local int plan_dense_off = 0;  // 1: the regular plan only (amx_model_create, when the library holds no kernels for the dense form)

static int plan_model_once(const amx_model_desc *d, DevModel &m, Image &im) {
  if (!d) return fail(AMX_EINVAL, "null argument");
  if (d->version != AMX_VERSION) return fail(AMX_EINVAL, "amx_model_desc version %d, library %d", d->version, AMX_VERSION);
  if (d->nbody < 1 || d->nq < 0 || d->nv < 0) return fail(AMX_EINVAL, "bad sizes");
  if (d->integrator != 0) return fail(AMX_EUNSUPPORTED, "only the Euler integrator is supported (integrator=%d)", d->integrator);
  if (d->solver != 1 && d->solver != 2) return fail(AMX_EUNSUPPORTED, "solver must be CG (1) or Newton (2), got %d", d->solver);
  if (d->nv < 1) return fail(AMX_EUNSUPPORTED, "model has no degrees of freedom");
  if (d->nv > AMX_MAX_NV) return fail(AMX_ECAPACITY, "nv=%d exceeds the register-row factorisation capacity %d", d->nv, AMX_MAX_NV);
  // connect / weld equalities ride in the contact rows' machinery as pseudo-pairs behind the collision pairs: one row group per
  // connect (3 rows + 1 idle), two per weld (translation, rotation)
  int neqgroups = 0;
  for (int e = 0; e < d->neq; e++) if (d->eq_active[e] && d->eq_type[e] >= 0 && d->eq_type[e] <= 1) neqgroups += d->eq_type[e] == 1 ? 2 : 1;
  // limited ball joints: one row on the joint's three dofs (MJX _instantiate_limit_ball) - a third kind of pseudo-pair, behind the
  // equality groups: the contact rows' machinery already handles rows over a dof list, single-sided and active by distance
  std::vector<int> ball_j;
  for (int j = 0; j < d->njnt; j++) if (d->jnt_limited[j] && d->jnt_type[j] == 1) ball_j.push_back(j);
  const int nbl = (int)ball_j.size();
  const int nb = d->nbody, nv = d->nv, nq = d->nq, nu = d->nu, nj = d->njnt, npc = d->npair, np = d->npair + neqgroups + nbl;
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
    for (int p = 0; p < npc; p++) {
      AMX_CHK(pair_geom1, p, 0, d->ngeom);
      AMX_CHK(pair_geom2, p, 0, d->ngeom);
    }
    for (int e = 0; e < d->neq; e++) {
      AMX_CHK(eq_type, e, 0, 3);
      if (d->eq_type[e] == 2) { AMX_CHK(eq_obj1id, e, 0, nj); AMX_CHK(eq_obj2id, e, -1, nj); }
      else { AMX_CHK(eq_obj1id, e, 0, nb); AMX_CHK(eq_obj2id, e, 0, nb); }
    }
#undef AMX_CHK
  }

  memset(&m, 0, sizeof(m));
  m.nq = nq; m.nv = nv; m.nu = nu; m.nb = nb; m.njnt = nj; m.npair = np; m.npc = npc; m.nbl = nbl;
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

  // subtree of body b = bodies b .. sub_last[b] when the bodies are numbered depth-first (MuJoCo's order): the up-scans then
  // sum contiguous record ranges instead of walking the tree level by level
  std::vector<int> sub_last(nb);
  for (int b = 0; b < nb; b++) sub_last[b] = b;
  for (int b = nb - 1; b >= 1; b--) sub_last[d->body_parentid[b]] = std::max(sub_last[d->body_parentid[b]], sub_last[b]);
  m.subtree_contig = 1;
  for (int c = 1; c < nb; c++) {  // contiguous iff every body between an ancestor and c is also below that ancestor
    for (int a = d->body_parentid[c]; a > 0; a = d->body_parentid[a])
      for (int x = a + 1; x < c; x++) {
        bool below = false;
        for (int y = x; y > 0; y = d->body_parentid[y]) if (y == a) { below = true; break; }
        if (!below) m.subtree_contig = 0;
      }
  }
  const char *level_env = getenv("AMX_LEVEL_SCANS");  // diagnostic: plan as if the numbering were not depth-first (level-by-level scans)
  const bool force_level = level_env && atoi(level_env) != 0;
  if (force_level) m.subtree_contig = 0;
  m.sub_iter = 1;
  for (int b = 1; b < nb; b++) if (d->body_parentid[b] > 0) m.sub_iter = std::max(m.sub_iter, sub_last[b] - b + 1);
  m.max_jntnum = 0;
  for (int b = 0; b < nb; b++) m.max_jntnum = std::max(m.max_jntnum, d->body_jntnum[b]);
  // bodies
  m.c_body = im.reserve(nb * BODY_S);
  for (int b = 0; b < nb; b++) {
    int o = m.c_body + b * BODY_S;
    im.seti(o, d->body_parentid[b]); im.seti(o + 1, depth[b]); im.seti(o + 2, root_slot[b]);
    im.seti(o + 3, d->body_jntadr[b] < 0 ? 0 : d->body_jntadr[b]); im.seti(o + 4, d->body_jntnum[b]);
    im.seti(o + 5, child_adr[b]); im.seti(o + 6, (child_num[b] & 255) | ((m.subtree_contig ? sub_last[b] + 1 : 0) << 8));
    for (int k = 0; k < 3; k++) { im.setf(o + 7 + k, d->body_pos[3 * b + k]); im.setf(o + 14 + k, d->body_ipos[3 * b + k]); im.setf(o + 22 + k, d->body_inertia[3 * b + k]); }
    for (int k = 0; k < 4; k++) { im.setf(o + 10 + k, d->body_quat[4 * b + k]); im.setf(o + 17 + k, d->body_iquat[4 * b + k]); }
    im.setf(o + 21, d->body_mass[b]);
  }
  // joints
  m.c_jnt = im.reserve(nj * JNT_S);
  for (int j = 0; j < nj; j++) {
    int o = m.c_jnt + j * JNT_S, t = d->jnt_type[j], qa = d->jnt_qposadr[j];
    if (t <= 1 && d->jnt_stiffness[j] != 0) return fail(AMX_EUNSUPPORTED, "spring on free/ball joint %d", j);
    if (t == 0 && d->jnt_limited[j]) return fail(AMX_EUNSUPPORTED, "limits on free joint %d", j);
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
    im.seti(o + 9, d->jnt_dofadr[j]);  // first dof of the dof's joint: its record holds the joint's anchor and axis
    im.seti(o + 10, -1);               // lane of the limit row on this dof (filled below), -1: none
    im.seti(o + 11, -1);               // lane of the friction-loss row on this dof (filled below), -1: none
  }
  // collision geoms (only those that appear in a pair)
  std::vector<int> gmap(d->ngeom > 0 ? d->ngeom : 1, -1), gc_src;
  auto use_geom = [&](int g) { if (gmap[g] < 0) { gmap[g] = (int)gc_src.size(); gc_src.push_back(g); } return gmap[g]; };
  for (int p = 0; p < npc; p++) { use_geom(d->pair_geom1[p]); use_geom(d->pair_geom2[p]); }
  m.ngc = (int)gc_src.size();
  if (m.ngc > AMX_WAVE) return fail(AMX_ECAPACITY, "%d collision geoms exceed the lane-role capacity %d", m.ngc, AMX_WAVE);
  m.c_geom = im.reserve(m.ngc * GEOM_S);
  for (int g = 0; g < m.ngc; g++) {
    int o = m.c_geom + g * GEOM_S, s = gc_src[g];
    im.seti(o, d->geom_bodyid[s]); im.seti(o + 1, d->geom_type[s]);
    for (int k = 0; k < 3; k++) { im.setf(o + 2 + k, d->geom_pos[3 * s + k]); im.setf(o + 9 + k, d->geom_size[3 * s + k]); }
    for (int k = 0; k < 4; k++) im.setf(o + 5 + k, d->geom_quat[4 * s + k]);
  }
  // convex hulls of the convex-convex pairs: record + vertices, face normals, face polygons (count + 8 ids), edges
  std::vector<int> hull_rec(d->nhull > 0 ? d->nhull : 1, 0);
  for (int h = 0; h < d->nhull; h++) {
    const int nvh = d->hull_vertnum[h], nfh = d->hull_facenum[h], neh = d->hull_edgenum[h];
    const int va = d->hull_vertadr[h], fa = d->hull_faceadr[h], ea = d->hull_edgeadr[h];
    if (nvh < 1 || nvh > CX_MAXV || nfh < 1 || nfh > CX_MAXF || neh < 0 || neh > CX_MAXE || va < 0 || va + nvh > d->nhullvert || fa < 0 || fa + nfh > d->nhullface ||
        ea < 0 || ea + neh > d->nhulledge)
      return fail(AMX_ECAPACITY, "hull %d: %d vertices, %d faces, %d edges (limits %d / %d / %d)", h, nvh, nfh, neh, (int)CX_MAXV, (int)CX_MAXF, (int)CX_MAXE);
    const int o = im.reserve(HULL_S);
    hull_rec[h] = o;
    im.seti(o, nvh); im.seti(o + 2, nfh); im.seti(o + 5, neh);
    int q = im.reserve(3 * nvh); im.seti(o + 1, q);
    for (int k = 0; k < 3 * nvh; k++) im.setf(q + k, d->hull_vert[3 * va + k]);
    q = im.reserve(3 * nfh); im.seti(o + 3, q);
    for (int k = 0; k < 3 * nfh; k++) im.setf(q + k, d->hull_face_normal[3 * fa + k]);
    q = im.reserve(9 * nfh); im.seti(o + 4, q);
    for (int f = 0; f < nfh; f++) {
      const int cnt = d->hull_face_vertnum[fa + f], adr = d->hull_face_vertadr[fa + f];
      if (cnt < 3 || cnt > 8 || adr < 0 || adr + cnt > d->nhullfacevert) return fail(AMX_ECAPACITY, "hull %d face %d: %d vertices (3..8)", h, f, cnt);
      im.seti(q + 9 * f, cnt);
      for (int k = 0; k < 8; k++) {
        const int id = d->hull_facevert[adr + (k < cnt ? k : 0)];
        if (id < 0 || id >= nvh) return fail(AMX_EINVAL, "hull %d face %d: vertex id %d outside [0, %d)", h, f, id, nvh);
        im.seti(q + 9 * f + 1 + k, id);
      }
    }
    q = im.reserve(2 * neh); im.seti(o + 6, q);
    for (int k = 0; k < 2 * neh; k++) {
      const int id = d->hull_edge[2 * ea + k];
      if (id < 0 || id >= nvh) return fail(AMX_EINVAL, "hull %d edge %d: vertex id %d outside [0, %d)", h, k / 2, id, nvh);
      im.seti(q + k, id);
    }
  }
  // pairs, contact slots, per-pair dof lists, Jacobian work items, Hessian work items
  m.c_pair = im.reserve(np * PAIR_S);
  std::vector<int> con_pair, pairdof, jac_items, hent;
  std::vector<std::vector<int>> dofmem(nv), hmembers(nv * (nv + 1) / 2);
  int maxchain = 1;
  int ncon = 0;
  // the equality row groups behind the collision pairs: (equality, 8 = points / 9 = orientation)
  std::vector<std::pair<int, int>> eqgroup;
  for (int e = 0; e < d->neq; e++)
    if (d->eq_active[e] && d->eq_type[e] <= 1) { eqgroup.push_back({e, 8}); if (d->eq_type[e] == 1) eqgroup.push_back({e, 9}); }
  for (int p = 0; p < np; p++) {
    const int o = m.c_pair + p * PAIR_S;
    int fn, b1 = 0, b2 = 0, g1 = 0, g2 = 0;
    if (p >= np - nbl) {
      // a ball joint's limit: the record carries where the quaternion and the three dofs sit, the larger end of the range, the
      // margin, the limit's stiffness / impedance constants and the joint's invweight
      const int j = ball_j[p - (np - nbl)];
      fn = 10;
      im.seti(o, d->jnt_qposadr[j]); im.seti(o + 1, d->jnt_dofadr[j]); im.seti(o + 2, fn); im.seti(o + 3, ncon); im.seti(o + 4, 1);
      im.seti(o + 5, 0); im.seti(o + 7, 0); im.seti(o + 8, 0);
      im.setf(o + 11, 0.0);
      write_kbi(im, o + 12, d->jnt_solref + 2 * j, d->jnt_solimp + 5 * j, d->timestep, refsafe);
      im.setf(o + 21, d->jnt_margin[j]);
      im.setf(o + 22, d->dof_invweight0[d->jnt_dofadr[j]]);
      im.setf(o + 23, std::max(d->jnt_range[2 * j], d->jnt_range[2 * j + 1]));
    } else if (p >= npc) {
      // connect / weld: bodies instead of geoms, body-fixed points and the relative orientation instead of shapes
      const int e = eqgroup[p - npc].first;
      fn = eqgroup[p - npc].second;
      b1 = d->eq_obj1id[e]; b2 = d->eq_obj2id[e];
      const double *dat = d->eq_data + 11 * e;
      const bool weld = d->eq_type[e] == 1;
      im.seti(o, b1); im.seti(o + 1, b2); im.seti(o + 2, fn); im.seti(o + 3, ncon); im.seti(o + 4, 1);
      im.seti(o + 5, weld ? 1 : 0);
      im.seti(o + 7, root_slot[b1]); im.seti(o + 8, root_slot[b2]);
      im.setf(o + 11, weld ? dat[10] : 1.0);
      write_kbi(im, o + 12, d->eq_solref + 2 * e, d->eq_solimp + 5 * e, d->timestep, refsafe);
      im.setf(o + 21, 0.0);
      im.setf(o + 22, d->body_invweight0[2 * b1 + (fn == 9)] + d->body_invweight0[2 * b2 + (fn == 9)]);
      // the point fixed in body1 and the one fixed in body2 (eq_data layout: include/amx.h)
      for (int k = 0; k < 3; k++) { im.setf(o + 23 + k, weld ? dat[3 + k] : dat[k]); im.setf(o + 26 + k, weld ? dat[k] : dat[3 + k]); }
      for (int k = 0; k < 4; k++) im.setf(o + 29 + k, weld ? dat[6 + k] : (k == 0 ? 1.0 : 0.0));
    } else {
    g1 = d->pair_geom1[p]; g2 = d->pair_geom2[p]; fn = d->pair_fn[p];
    if (fn < 0 || (fn > 7 && fn != 11)) return fail(AMX_EUNSUPPORTED, "pair %d: unknown collision function %d", p, fn);
    if ((fn == 7 || fn == 11) && (d->pair_hull2[p] < 0 || d->pair_hull2[p] >= d->nhull)) return fail(AMX_EINVAL, "pair %d: sphere / capsule - convex pair without a hull", p);
    if (fn == 6 && (d->pair_hull1[p] < 0 || d->pair_hull1[p] >= d->nhull || d->pair_hull2[p] < 0 || d->pair_hull2[p] >= d->nhull))
      return fail(AMX_EINVAL, "pair %d: convex-convex pair without hulls", p);
    if (fn == 2 && (d->pair_vertnum[p] < 1 || d->pair_vertadr[p] < 0 || d->pair_vertadr[p] + d->pair_vertnum[p] > d->nconvexvert))
      return fail(AMX_EINVAL, "pair %d: bad convex vertex range", p);
    b1 = d->geom_bodyid[g1]; b2 = d->geom_bodyid[g2];
    im.seti(o, gmap[g1]); im.seti(o + 1, gmap[g2]); im.seti(o + 2, fn); im.seti(o + 3, ncon); im.seti(o + 4, PAIR_NCON[fn]);
    im.seti(o + 5, fn == 2 ? d->pair_vertadr[p] : 0); im.seti(o + 6, fn == 2 ? d->pair_vertnum[p] : 0);
    im.seti(o + 7, root_slot[b1]); im.seti(o + 8, root_slot[b2]);
    if (fn == 6) { im.seti(o + 25, hull_rec[d->pair_hull1[p]]); im.seti(o + 26, hull_rec[d->pair_hull2[p]]); }
    if (fn == 7 || fn == 11) im.seti(o + 26, hull_rec[d->pair_hull2[p]]);
    im.setf(o + 11, d->pair_friction[3 * p]);
    write_kbi(im, o + 12, d->pair_solref + 2 * p, d->pair_solimp + 5 * p, d->timestep, refsafe);
    im.setf(o + 21, d->pair_margin[p]);
    {
      // the pyramid rows' diagApprox, the same for all four rows of a contact (f = +-mu enters squared): folded here in double
      const double tw = d->body_invweight0[2 * b1] + d->body_invweight0[2 * b2], mu = d->pair_friction[3 * p];
      im.setf(o + 22, (tw + mu * mu * tw) * 2 * mu * mu / d->impratio);
    }
    }
    // union of the ancestor dof chains of both bodies (a ball limit: the joint's own three dofs)
    std::vector<int> sign(nv, 0);
    if (fn == 10) { const int da = d->jnt_dofadr[ball_j[p - (np - nbl)]]; for (int k = 0; k < 3; k++) sign[da + k] = 1; }
    else
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
    if (dn > 31) return fail(AMX_ECAPACITY, "pair %d couples %d dofs (limit 31: slot indices of a contact row's dof list take five bits)", p, dn);
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
  m.jac_one_sided = 1;
  for (int it : jac_items) if (((it >> 16) & 3) != 1) m.jac_one_sided = 0;
  if (ncon > 32) return fail(AMX_ECAPACITY, "%d contact slots (connect / weld row groups included) exceed 32", ncon);
  {
    int first = -1, last = -1, cnt = 0;
    for (int p = 0; p < npc; p++) if (d->pair_fn[p] == 6) { if (first < 0) first = p; last = p; cnt++; }
    if (cnt > 0 && last - first + 1 != cnt) return fail(AMX_EINVAL, "convex-convex pairs must be contiguous in the pair list (pairs are sorted by function)");
    if (cnt > 0) { m.cx_first = first; m.ncx = cnt; }
    m.cx_maxv = 0;
    for (int p = 0; p < npc; p++)
      if (d->pair_fn[p] == 6)
        for (int h : {d->pair_hull1[p], d->pair_hull2[p]})
          if (h >= 0 && h < d->nhull && d->hull_vertnum[h] > m.cx_maxv) m.cx_maxv = d->hull_vertnum[h];
  }
  // plane - convex pairs with at most 32 hull vertices go to the vertex-parallel routine when they form one contiguous run of
  // pair indices (pairs come grouped by function, so they do unless a larger hull sits among them)
  {
    int first = -1, last = -1, cnt = 0;
    for (int p = 0; p < npc; p++)
      if (d->pair_fn[p] == 2 && d->pair_vertnum[p] <= 32) { if (first < 0) first = p; last = p; cnt++; }
    if (cnt > 0 && last - first + 1 == cnt) { m.pcx_first = first; m.npcx = cnt; }
    m.pcx_unit = 32;
    if (m.npcx > 0) {
      int mx = 0;
      for (int p = m.pcx_first; p < m.pcx_first + m.npcx; p++) mx = std::max(mx, d->pair_vertnum[p]);
      if (mx <= 16) m.pcx_unit = 16;
    }
  }
  // limit rows, equality rows
  std::vector<int> lim_j;
  for (int j = 0; j < nj; j++) if (d->jnt_limited[j] && d->jnt_type[j] >= 2) lim_j.push_back(j);  // slide / hinge (ball: pseudo-pairs above)
  m.nlim = (int)lim_j.size();
  m.c_lim = im.reserve(m.nlim * LIM_S);
  for (int l = 0; l < m.nlim; l++) {
    int j = lim_j[l], o = m.c_lim + l * LIM_S;
    im.seti(o, d->jnt_qposadr[j]); im.seti(o + 1, d->jnt_dofadr[j]);
    im.setf(o + 2, d->jnt_range[2 * j]); im.setf(o + 3, d->jnt_range[2 * j + 1]); im.setf(o + 4, d->jnt_margin[j]);
    write_kbi(im, o + 5, d->jnt_solref + 2 * j, d->jnt_solimp + 5 * j, d->timestep, refsafe);
    im.setf(o + 14, d->dof_invweight0[d->jnt_dofadr[j]]);
  }
  // friction-loss rows: one per dof with frictionloss > 0, after the equality rows and before the limit rows.  pos = 0 always,
  // so the impedance is the constant d(0) = dmin and D, b are folded here
  std::vector<int> fri_i;
  for (int i = 0; i < nv; i++) if (d->dof_frictionloss[i] > 0) fri_i.push_back(i);
  m.nfri = (int)fri_i.size();
  m.c_fri = im.reserve(m.nfri * FRI_S);
  for (int f = 0; f < m.nfri; f++) {
    const int i = fri_i[f], o = m.c_fri + f * FRI_S;
    // the position-independent part of solreffriction / solimpfriction (write_kbi's arithmetic, kept in double)
    const double *sr = d->dof_solref + 2 * i, *si = d->dof_solimp + 5 * i;
    auto clip = [](double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); };
    const double imp = clip(si[0], 0.0001, 0.9999), dmax = clip(si[1], 0.0001, 0.9999);  // impedance at pos = 0 is dmin
    double timeconst = sr[0];
    if (refsafe) timeconst = std::max(timeconst, 2 * d->timestep);
    double b = 2.0 / (dmax * timeconst);
    if (sr[1] <= 0) b = -sr[1] / dmax;
    const double R = std::max(d->dof_invweight0[i] * (1 - imp) / imp, 1e-15);
    im.seti(o, i); im.setf(o + 1, 1.0 / R); im.setf(o + 2, b); im.setf(o + 3, d->dof_frictionloss[i]);
  }
  {
    int neq_active = 0;
    for (int e = 0; e < d->neq; e++) if (d->eq_active[e] && d->eq_type[e] == 2) neq_active++;  // slot A carries the joint equalities only
    for (int f = 0; f < m.nfri; f++) im.seti(m.c_dof + fri_i[f] * DOF_S + 11, neq_active + f);
    for (int l = 0; l < m.nlim; l++) im.seti(m.c_dof + d->jnt_dofadr[lim_j[l]] * DOF_S + 10, neq_active + m.nfri + l);
  }
  std::vector<int> eq_e;
  for (int e = 0; e < d->neq; e++) if (d->eq_active[e] && d->eq_type[e] == 2) eq_e.push_back(e);  // connect / weld: pseudo-pairs above
  m.neq = (int)eq_e.size();
  m.c_eq = im.reserve(m.neq * EQ_S);
  for (int r = 0; r < m.neq; r++) {
    int e = eq_e[r], o = m.c_eq + r * EQ_S, j1 = d->eq_obj1id[e], j2 = d->eq_obj2id[e];
    if (j1 < 0 || j1 >= nj || d->jnt_type[j1] < 2 || (j2 >= 0 && (j2 >= nj || d->jnt_type[j2] < 2)))
      return fail(AMX_EUNSUPPORTED, "equality %d must couple hinge/slide joints", e);
    im.seti(o, d->jnt_qposadr[j1]); im.seti(o + 1, d->jnt_dofadr[j1]);
    im.seti(o + 2, j2 >= 0 ? d->jnt_qposadr[j2] : -1); im.seti(o + 3, j2 >= 0 ? d->jnt_dofadr[j2] : -1);
    for (int k = 0; k < 5; k++) im.setf(o + 4 + k, d->eq_data[11 * e + k]);
    write_kbi(im, o + 9, d->eq_solref + 2 * e, d->eq_solimp + 5 * e, d->timestep, refsafe);
    im.setf(o + 18, d->dof_invweight0[d->jnt_dofadr[j1]] + (j2 >= 0 ? d->dof_invweight0[d->jnt_dofadr[j2]] : 0.0));
    im.setf(o + 19, d->qpos0[d->jnt_qposadr[j1]]); im.setf(o + 20, j2 >= 0 ? d->qpos0[d->jnt_qposadr[j2]] : 0.0);
  }
  const bool con_off = d->disableflags & 1;
  m.nefc = con_off ? 0 : m.neq + m.nfri + m.nlim + 4 * m.ncon;
  if (m.neq + m.nfri + m.nlim > AMX_WAVE) return fail(AMX_ECAPACITY, "%d equality + friction + limit rows exceed the lane-role capacity %d", m.neq + m.nfri + m.nlim, AMX_WAVE);
  // A lane's contact-row slot holds one row, so a 64-lane group carries 16 contacts.  Two slots per lane (rows gl and gl + G; r04) for
  // the models the per-model kernels' fused Newton path and second-form line search cover: contacts only (no equality, friction-loss,
  // connect / weld or ball-limit rows), Newton, nv <= 31, ls_iterations <= 10.  Such a model has no generic kernels (amx_model_create).
  // (r05: joint-equality rows allowed too - the reference's Barrett hand, bh280.xml:195-200; their Newton path stores H and runs the
  //  first-form line search, both of which carry the second slot since r05)
  const bool two_ok = d->solver == 2 && m.nfri == 0 && np == npc && nbl == 0 && nv <= 31 && d->ls_iterations <= 10 && !(d->disableflags & 1);
  m.nbs = 1;
  if (two_ok && m.ncon > 16) m.nbs = 2;
  if (const char *force = getenv("AMX_FORCE_NBS")) { if (atoi(force) == 2 && two_ok && m.ncon > 4) m.nbs = 2; }  // diagnostic: two slots on a narrower group
  if (plan_dense && two_ok && m.ncon > 8) m.nbs = 2;
  if (m.ncon > 16 * m.nbs)
    return fail(AMX_ECAPACITY, "%d contact slots (connect / weld row groups included) exceed %d (4 rows each; two rows per lane of a 64-lane group only for "
                "Newton models with nv <= 31 and ls_iterations <= 10 whose other rows are joint limits and joint equalities)", m.ncon, 16 * m.nbs);
  // lanes per environment: the smallest group that holds every lane-role index set.  A lane carries two constraint rows:
  // one equality / limit row and one contact row (amx_kernels.hip, Env), so the rows ask for max(neq + nlim, 4 ncon) lanes.
  {
    const int need = std::max({nb, nj, nv, np, m.ngc, con_off ? 0 : m.neq + m.nfri + m.nlim, con_off ? 0 : (4 * m.ncon + m.nbs - 1) / m.nbs, 1});
    m.G = need <= 16 ? 16 : (need <= 32 ? 32 : 64);
    if (const char *force = getenv("AMX_FORCE_G")) { int g = atoi(force); if ((g == 16 || g == 32 || g == 64) && g >= m.G) m.G = g; }
    if (plan_force_lanes >= m.G && (plan_force_lanes == 32 || plan_force_lanes == 64)) m.G = plan_force_lanes;  // the wide plan of a model (amx_model_create)
    if (m.nbs == 2 && m.G < 32) m.G = 32;  // the two-slot form lives in the fused Newton path (32- and 64-lane groups)
  }
  // lists
  auto add_ints = [&](const std::vector<int> &v) { int o = im.reserve((int)v.size()); for (size_t k = 0; k < v.size(); k++) im.seti(o + (int)k, v[k]); return o; };
  m.c_child = add_ints(child_list);
  {
    // ancestor chains, root first: the down-scans (poses, velocities) walk a body's own chain in registers instead of waiting
    // level by level for the parent's result to come back through LDS
    const int w = std::max(m.nlevel - 2, 1);
    std::vector<int> anc((size_t)nb * w, 0);
    for (int b = 1; b < nb; b++) {
      std::vector<int> chain;
      for (int a = d->body_parentid[b]; a > 0; a = d->body_parentid[a]) chain.push_back(a);
      for (size_t k = 0; k < chain.size(); k++) anc[(size_t)b * w + k] = chain[chain.size() - 1 - k];
    }
    m.c_anc = add_ints(anc);
    // dof groups of every body for the velocity pass: a free joint is two groups (translations, rotations), every other joint
    // one (ball: 3 dofs, hinge / slide: 1).  Within a group the velocity the cdof_dot cross product is taken with is the same
    // for every dof, and the cross product is bilinear: one product with the group's sum (velocity_and_bias)
    std::vector<std::vector<int>> grp(nb);
    m.max_bgrp = 1;
    for (int b = 1; b < nb; b++) {
      for (int k = 0; k < d->body_jntnum[b]; k++) {
        const int j = d->body_jntadr[b] + k, t = d->jnt_type[j], da = d->jnt_dofadr[j];
        if (t == 0) { grp[b].push_back(da | (3 << 8)); grp[b].push_back((da + 3) | (3 << 8) | (1 << 12)); }
        else grp[b].push_back(da | ((t == 1 ? 3 : 1) << 8) | (1 << 12));
      }
      m.max_bgrp = std::max(m.max_bgrp, (int)grp[b].size());
    }
    std::vector<int> bgrp((size_t)nb * m.max_bgrp, 0);
    for (int b = 1; b < nb; b++) for (size_t g = 0; g < grp[b].size(); g++) bgrp[(size_t)b * m.max_bgrp + g] = grp[b][g];
    m.c_bgrp = add_ints(bgrp);
    m.deep_free = 0;
    for (int j = 0; j < nj; j++) if (d->jnt_type[j] == 0 && depth[d->jnt_bodyid[j]] > 1) m.deep_free = 1;
    if (force_level) m.deep_free = 1;
  }
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
    m.max_dofmem = 0;
    for (int i = 0; i < nv; i++) m.max_dofmem = std::max(m.max_dofmem, (int)dofmem[i].size());
    if (flat.empty()) flat.push_back(0);  // the gather reads entry 0 with weight 0 when a dof has no terms
    m.c_dofmemadr = add_ints(adr); m.c_dofmem = add_ints(flat);
  }
  m.ldc = maxchain | 1;
  {
    // dof of every slot of every pair's dof list: the row dots J*v read v gathered in this order (amx_kernels.hip, stage_chain)
    std::vector<int> vd((size_t)std::max(np, 1) * m.ldc, -1);
    for (int p = 0; p < np; p++) {
      const int o = m.c_pair + p * PAIR_S, d0 = im.geti(o + 9), dn = im.geti(o + 10);
      for (int t = 0; t < dn; t++) vd[(size_t)p * m.ldc + t] = pairdof[d0 + t] & 255;
    }
    m.c_vchdof = add_ints(vd);
  }
  // Newton on 32- and 64-lane groups without equality rows never stores H: the matrix cores' J'DJ, the lane's row of M and the
  // limit rows' diagonal terms meet in registers and are eliminated there (amx_kernels.hip, newton_solve_fused).
  m.fusedH = (m.G >= 32 && d->solver == 2 && m.neq == 0) ? 1 : 0;
  // M (and H where it is stored): full symmetric [nv][ldm], rows 16-byte aligned for ds_read_b128.  Eight lanes reading their
  // rows' chunk c must cover all 32 banks: either ldm/4 is odd (the row stride does it), or - fused path only, where no other
  // code indexes the matrix - ldm = 4 ceil(nv/4) and row i stores its chunks rotated by i (chunk c at position (c + i) mod
  // ldm/4): a 30-dof model keeps 32-float rows instead of 36.
  m.ldm = ((nv + 3) / 4) * 4;
  m.mrot = 0;
  if ((m.ldm / 4) % 2 == 0) { if (m.fusedH) m.mrot = 1; else m.ldm += 4; }
  // Compact M (dense form only; r05).  M[i][j] != 0 only when one of i, j is an ancestor of the other.  A floating-base robot's first mnb0
  // dofs form a chain every other dof descends from: their rows are dense and stay whole.  Any other row i has entries on that chain - which
  // are, M being symmetric, column i of the whole rows - and on the dofs >= mnb0 of its own branch, a narrow range of columns: it stores the
  // 16-byte chunks [clo_i, chi_i] that cover that range, at most mwin of them (amx_kernels.hip, m_row_compact: the lane reads its row's
  // leading entries with mnb0 four-byte loads down the whole rows' column, the others as chunks, a chunk outside its range from the constants'
  // zero block).  The falls-and-rests biped (30 dofs: base, three-dof waist, head, two four-dof arms, two six-dof legs): 6 x 36 + 264 = 480
  // floats instead of 960.  Needs two spare columns: column nv carries the right-hand side to the lane that mirrors it as a row, column
  // ldm - 1 stays zero for the lanes without a row (ldm grows by one chunk where nv leaves fewer than two).
  m.mcomp = 0; m.mnb0 = 0; m.mwin = 0; m.mbs = m.ldm;
  int m_floats = nv * m.ldm;
  std::vector<int> mrow_base(nv, 0), mrow_clo(nv, 0), mrow_chi(nv, m.ldm / 4 - 1);
  const int ldm_c = m.ldm >= nv + 2 ? m.ldm : m.ldm + 4;  // the row length compact M would take: two spare columns (a 27- or 31-dof model gets one more chunk)
  if (plan_dense && m.fusedH && m.nbs == 2 && nv <= 31 && !getenv("AMX_DENSE_NO_MCOMP")) {
    auto is_anc = [&](int a, int i) { for (int j = i; j >= 0; j = d->dof_parentid[j]) if (j == a) return true; return false; };  // a is i or above it
    int nb0 = 0;
    while (nb0 < nv && d->dof_parentid[nb0] == nb0 - 1) {
      bool all = true;
      for (int i = nb0 + 1; i < nv && all; i++) all = is_anc(nb0, i);
      if (!all) break;
      nb0++;
    }
    int wmax = 0;
    for (int i = nb0; i < nv; i++) {
      int lo = i, hi = i;
      for (int j = nb0; j < nv; j++) if (is_anc(j, i) || is_anc(i, j)) { lo = std::min(lo, j); hi = std::max(hi, j); }
      mrow_clo[i] = lo / 4; mrow_chi[i] = hi / 4;
      wmax = std::max(wmax, hi / 4 - lo / 4 + 1);
    }
    int total = nb0 * (ldm_c + 4);
    for (int i = nb0; i < nv; i++) total += 4 * (mrow_chi[i] - mrow_clo[i] + 1);
    if (nb0 >= 1 && nb0 < nv && total < nv * m.ldm) {
      m.ldm = ldm_c;
      m.mcomp = 1; m.mnb0 = nb0; m.mwin = wmax; m.mrot = 0;
      m.mbs = m.ldm + ((m.ldm / 4) % 2 == 0 ? 4 : 0);  // an odd number of chunks per whole row: the leading lanes' sixteen-byte loads do not collide
      int o = 0;
      for (int i = 0; i < nv; i++) {
        if (i < nb0) { mrow_clo[i] = 0; mrow_chi[i] = m.ldm / 4 - 1; }
        mrow_base[i] = o;
        o += i < nb0 ? m.mbs : 4 * (mrow_chi[i] - mrow_clo[i] + 1);
      }
      m_floats = o;
    }
  }
  auto moff = [&](int i, int j) {  // float offset of M[i][j] inside the matrix block
    if (m.mcomp) {
      if (i >= m.mnb0 && j < m.mnb0) std::swap(i, j);  // not stored in row i: the same number lives in the whole row j
      return mrow_base[i] + (j - 4 * mrow_clo[i]);
    }
    const int nch = m.ldm / 4, rot = m.mrot ? AMX_ROW_ROT(i, nch) : 0;
    return i * m.ldm + 4 * ((j / 4 + rot) % nch) + (j & 3);
  };
  {
    std::vector<int> mr(nv, 0);
    for (int i = 0; i < nv; i++) mr[i] = mrow_base[i] | (mrow_clo[i] << 12) | (mrow_chi[i] << 16);
    m.c_mrow = m.mcomp ? add_ints(mr) : 0;
  }
  {
    std::vector<int> Mlist;  // one item per (dof i, ancestor-or-self j): offsets of M[i][j] and M[j][i], and the pair itself
    for (int i = 0; i < nv; i++)
      for (int j = i; j >= 0; j = d->dof_parentid[j]) Mlist.push_back((int)((unsigned)moff(i, j) | ((unsigned)moff(j, i) << 11) | ((unsigned)i << 22) | ((unsigned)j << 27)));
    m.nM = (int)Mlist.size(); m.c_M = add_ints(Mlist);
  }
  // ---- two-pair models on the 4-block matrix instruction (amx_kernels.hip, newton_solve_pair2).  v_mfma_f32_16x16x1_4b has four
  // 16 x 16 blocks: (environment, pair) for a 32-lane group whose contacts all belong to two pairs of four contacts with dof lists
  // of at most 13 entries - a biped's two feet.  Lane 16 f + k of an environment feeds block (env, f): as B operand the Jacobian
  // entry of the dof whose ROW of the block it will hold, as A operand the entry of the dof whose COLUMN register k stands for
  // (k = 15 - ... see p2_col; one more A lane carries the row's residual, which makes the gradient).  Rows are placed so that
  //   * pair 0's dofs below 16 and pair 1's dofs from 16 up sit in their own lanes (home: added in place),
  //   * pair 0's dofs from 16 up / pair 1's dofs below 16 sit in the OTHER 16-lane row at (position + shift): one row exchange
  //     (v_permlane16_swap) and a DPP row shift on the add bring them home;
  // and so that every other lane the adds' bank masks enable holds an all-zero row.  No placement: mfma2 = 0 (32 x 32 path).
  m.mfma2 = 0; m.p2_shl0 = m.p2_shl1 = 0; m.p2_banks = 0;
  p2_col.assign(32, -1);
  {
    std::vector<int> pk(32, 0xFFFF);
    const bool shape = m.G == 32 && m.fusedH && np == 2 && npc == 2 && nbl == 0 && ncon == 8 && nv <= 30 && d->solver == 2 && m.neq == 0 && m.nfri == 0;
    bool ok = shape;
    int dl[2] = {0, 0}, d0s[2] = {0, 0};
    if (ok) for (int f = 0; f < 2 && ok; f++) {
      const int o = m.c_pair + f * PAIR_S;
      d0s[f] = im.geti(o + 9); dl[f] = im.geti(o + 10);
      ok = dl[f] <= 13 && im.geti(o + 3) == 4 * f && im.geti(o + 4) == 4;  // contacts 4 f .. 4 f + 3
    }
    if (ok) {
      // B placement: rowlane[f][pos] = slot of the pair's list whose row lane 16 f + pos holds, -1 none
      int rowlane[2][16], shl[2] = {-1, -1};
      for (int f = 0; f < 2; f++) for (int k = 0; k < 16; k++) rowlane[f][k] = -1;
      for (int f = 0; f < 2 && ok; f++)
        for (int t = 0; t < dl[f]; t++) {  // home rows
          const int dof = pairdof[d0s[f] + t] & 255;
          if ((dof >= 16) == (f == 1)) rowlane[f][dof & 15] = t;
        }
      auto banks_of = [&](int f, bool home, int sh) {  // bank mask of the destination lanes of pair f's home / cross adds
        int b = 0;
        for (int t = 0; t < dl[f]; t++) {
          const int dof = pairdof[d0s[f] + t] & 255;
          if (((dof >= 16) == (f == 1)) == home) b |= 1 << ((dof & 15) >> 2);
        }
        (void)sh;
        return b;
      };
      for (int f = 0; f < 2 && ok; f++) {
        // cross rows of pair f: dest position p = dof & 15 (in the other row), source position p + s in block f
        for (int s_ = 0; s_ < 16 && shl[f] < 0; s_++) {
          int trial[16];
          for (int k = 0; k < 16; k++) trial[k] = rowlane[f][k];
          bool fits = true;
          for (int t = 0; t < dl[f] && fits; t++) {
            const int dof = pairdof[d0s[f] + t] & 255;
            if ((dof >= 16) == (f == 1)) continue;
            const int src = (dof & 15) + s_;
            if (src > 15 || trial[src] >= 0) fits = false; else trial[src] = t;
          }
          if (!fits) continue;
          // junk check: every enabled destination lane (bank granularity) that is not a destination reads an unused or out-of-row lane
          const int cb = banks_of(f, false, s_), hb = banks_of(f, true, 0);
          for (int p = 0; p < 16 && fits; p++) {
            if ((cb >> (p >> 2)) & 1) {
              bool is_dest = false;
              for (int t = 0; t < dl[f]; t++) { const int dof = pairdof[d0s[f] + t] & 255; if ((dof >= 16) != (f == 1) && (dof & 15) == p) is_dest = true; }
              const int src = p + s_;
              if (!is_dest && src <= 15 && trial[src] >= 0) fits = false;
            }
            if ((hb >> (p >> 2)) & 1) {  // home adds read the lane itself: it must hold its own dof's row or nothing
              bool is_home = false;
              for (int t = 0; t < dl[f]; t++) { const int dof = pairdof[d0s[f] + t] & 255; if ((dof >= 16) == (f == 1) && (dof & 15) == p) is_home = true; }
              if (!is_home && trial[p] >= 0) fits = false;
            }
          }
          if (fits) { shl[f] = s_; for (int k = 0; k < 16; k++) rowlane[f][k] = trial[k]; }
        }
        if (shl[f] < 0) ok = false;
      }
      if (ok) {
        m.mfma2 = 1; m.p2_shl0 = shl[0]; m.p2_shl1 = shl[1];
        m.p2_banks = banks_of(0, true, 0) | (banks_of(1, true, 0) << 4) | (banks_of(1, false, 0) << 8) | (banks_of(0, false, 0) << 12);
        for (int f = 0; f < 2; f++)
          for (int k = 0; k < 16; k++) {
            // A operand / column register k of block f: slot k of the list, then the residual (the gradient column)
            const int sa = k < dl[f] ? k : (k == dl[f] ? 0xFE : 0xFF);
            p2_col[16 * f + k] = k < dl[f] ? (pairdof[d0s[f] + k] & 255) : (k == dl[f] ? nv : -1);
            const int sb = rowlane[f][k] >= 0 ? rowlane[f][k] : 0xFF;
            pk[16 * f + k] = sa | (sb << 8);
          }
      }
    }
    m.c_p2 = add_ints(pk);
  }
  m.c_qpos0 = im.reserve(nq);
  for (int k = 0; k < nq; k++) im.setf(m.c_qpos0 + k, d->qpos0[k]);
  m.c_vert = im.reserve(3 * d->nconvexvert);
  for (int k = 0; k < 3 * d->nconvexvert; k++) im.setf(m.c_vert + k, d->convex_vert[k]);
  while (im.size() & 3) im.reserve(1);
  const int nzero = m.mcomp ? 4 * 32 + m.ldm : m.ldm;
  m.c_zero = im.reserve(nzero);  // stays zero: Image::reserve clears
  for (int k = 0; k < nzero; k++) im.setf(m.c_zero + k, 0.0f);
  m.c_prog = im.reserve(AMX_WAVES_PER_BLOCK);  // zeros; written in LDS only (the waves' step counters)
  while (im.size() & 3) im.reserve(1);  // environment slabs start 16-byte aligned
  m.nconst = im.size();

  // ---- environment slab plan (floats).  Phase order of a step (amx_kernels.hip, Env::forward): kinematics, com_pos,
  // collision, crb + M, velocity + bias, smooth forces, make_constraint, solves.  Storage with disjoint lifetimes is shared:
  //   * joint anchors / axes (kinematics -> cdof) live in fields 6..11 of the joint's first dof record (crb*cdof, then the
  //     solver vectors, take them over afterwards);
  //   * region R1 holds the body records (poses, cinert -> cfrc, crb -> cvel/cacc) until the smooth forces are formed, and from
  //     make_constraint on the contact Jacobian rows, the per-row records and the 16-byte aligned staging vector of M*v.
  int off = 0;
  auto take = [&](int n) { int o = off; off += (n > 0 ? n : 1); return o; };
  m.e_qpos = take(nq); m.e_qvel = take(nv); m.e_ctrl = take(nu);
  m.e_dof = take(DOF_E * nv);
  m.e_com = take(3 * m.nroot); m.e_actf = take(nu);
  m.e_con = take(CON_E * m.ncon); m.e_frame = take(9 * np);
  m.e_geom = take(GEO_E * m.ngc);
  m.e_cx = m.ncx > 0 ? take(6 * m.cx_maxv + CX_TAIL) : 0;
  const int ntri = nv * (nv + 1) / 2;
  const int xover_lo = m.e_con, xover_hi = off;  // contact points, frames, geom poses (and the convex scratch): dead once make_constraint has run
  off = (off + 3) & ~3;
  m.e_M = take(m_floats);
  {
    const int r1 = off;
    m.e_body = r1;
    m.e_J = r1;
    // dense form: three rows per contact - the normal's and the two tangents' (the latter times the friction coefficient), of which the four
    // pyramid rows are sums and differences (amx_kernels.hip: Env::jfr)
    m.jfr = (plan_dense && m.fusedH && m.nbs == 2 && m.G == 32 && !m.mfma2 && !getenv("AMX_DENSE_NO_JFR")) ? 1 : 0;  // (the variables: diagnostic)
    int o2 = r1 + std::max((m.jfr ? 3 : 4) * m.ncon * m.ldc, 1);
    // dense form: the row records and the staging vector - first written by the solver, after make_constraint's last read of a contact
    // point - take the place of the contact points / frames / geom poses (rewritten by the next step's kinematics and collision pass)
    const int nrowf = std::max(ROW_E * m.nefc, 1), nstage = std::max(m.ldm, nv + 8);
    int vstage = -1;
    m.e_row = -1;
    if (plan_dense && m.fusedH) {
      int x = (xover_lo + 3) & ~3;
      if (x + nstage <= xover_hi) { vstage = x; x += nstage; }
      if (x + nrowf <= xover_hi) { m.e_row = x; x += nrowf; }
    }
    if (m.e_row < 0) { m.e_row = o2; o2 += nrowf; }
    o2 = (o2 + 3) & ~3;
    if (vstage < 0) { vstage = o2; o2 += nstage; }
    m.e_vch = o2; o2 += std::max(np * m.ldc, 1);
    off = std::max(r1 + BODY_E * nb, o2);
    off = (off + 3) & ~3;
    // H: stored only when it is not eliminated in registers; it doubles as the staging vector and as CG scratch
    m.e_H = m.fusedH ? vstage : take(std::max(nv * m.ldm, nv + 8));
  }
  m.e_Lb = d->solver == 1 ? take(ntri) : m.e_H;
  m.e_prof = 0;  // (unused since r02b: the diagnostic build keeps its counters in HBM)
  m.env_floats = (off + 3) & ~3;  // slabs keep 16-byte alignment
  if (plan_dense && m.G == 32) {
    // the two environments of a wavefront 32 banks apart (of 64), where that still fits.  (Meant to keep their lane-per-record reads at odd
    // strides out of each other's banks; measured afterwards on the headline kernel - slabs 16, 32, 44 banks apart - it makes no difference:
    // a 64-lane four-byte access is served half a wave at a time.  Harmless; kept so that the committed plan records stay as measured.)
    int padded = m.env_floats;
    while ((padded & 63) != 32) padded += 4;
    if ((m.nconst + AMX_WAVES_PER_BLOCK * 2 * padded) * (int)sizeof(float) <= 160 * 1024) m.env_floats = padded;
  }
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

// The plan of a model: the regular one, or - when the contact rows alone push a Newton model whose other rows are joint limits onto 64-lane
// groups (one environment per wavefront) - the dense form (plan_dense), if that keeps more environments resident per CU.  The falls-and-rests
// biped (16 contact slots, 30 dofs): 8 -> 16 environments per CU.  AMX_NO_DENSE=1: the regular plan (A/B switch).
static int plan_model(const amx_model_desc *d, DevModel &m, Image &im) {
  plan_dense = 0;
  const int rc = plan_model_once(d, m, im);
  if (rc != AMX_OK) return rc;
  const char *no = getenv("AMX_NO_DENSE");
  if (m.G == 64 && m.nbs == 1 && m.fusedH && plan_force_lanes == 0 && !plan_dense_off && !(no && atoi(no) != 0)) {
    DevModel m2;
    Image im2;
    const std::vector<int> p2_keep = p2_col;
    char err_keep[sizeof(g_err)];
    memcpy(err_keep, g_err, sizeof(g_err));
    plan_dense = 1;
    const int rc2 = plan_model_once(d, m2, im2);
    plan_dense = 0;
    if (rc2 == AMX_OK && m2.G == 32 && (m2.wpb * (AMX_WAVE / m2.G) > m.wpb * (AMX_WAVE / m.G) || getenv("AMX_DENSE_ALWAYS"))) { m = m2; im = im2; }
    else { p2_col = p2_keep; memcpy(g_err, err_keep, sizeof(g_err)); }
  }
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
  t += "p2_col=";
  for (int k = 0; k < 32; k++) { snprintf(line, sizeof(line), "%d%s", p2_col[k], k + 1 < 32 ? "," : ""); t += line; }
  t += "\n";
  t += "mrow=";  // compact M: per row "offset:first chunk:last chunk" (empty unless mcomp)
  if (m.mcomp) for (int i = 0; i < m.nv; i++) { const int pk = im.geti(m.c_mrow + i); snprintf(line, sizeof(line), "%d:%d:%d%s", pk & 4095, (pk >> 12) & 15, (pk >> 16) & 15, i + 1 < m.nv ? "," : ""); t += line; }
  t += "\n";
  if (t.size() + 1 > cap) return fail(AMX_EINVAL, "amx_model_plan_describe: buffer of %zu bytes, %zu needed", cap, t.size() + 1);
  memcpy(out, t.c_str(), t.size() + 1);
  return AMX_OK;
}

