/* amx_oracle.c — CPU restatement of the reference's step/rollout hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may build, load or call this file.  The product path (ambersim_amd/csrc, HIP) never
 * links or falls back to it.
 *
 * PARITY UNPINNED.  The arithmetic of the path lives in the third-party package `mujoco-mjx`
 * (pinned by the reference only as ">=3.0.0", pyproject.toml:25; the one upstream revision it
 * names is 57e6940f579484adf34eebedc51279a818909f34, ambersim/utils/io_utils.py:233).  That
 * package is absent from /root/reference and from this image, and the reference's tests hold no
 * numeric qpos/qvel vector for this path.  Each function below restates the
 * published MJX algorithm for its stage (module names given as [mjx:<module>.<fn>]) and cites
 * the reference call site that reaches it.  The oracle itself is pinned by closed-form
 * known-answer tests in tests/test_oracle_kat.py (pendulum closed form from
 * ambersim/models/pendulum/pendulum.xml:3-23, energy, momentum, free fall, resting contact).
 * and, stage by stage, by hand-worked closed forms in tests/test_closed_forms.py (limit, contact-pyramid and joint-equality rows:
 * impedance, reference acceleration, regulariser; spring / damping / armature and the implicit-damping step; textbook double
 * pendulum for CRB and RNE; Euler's equations and the quaternion step of a free body) - numbers no code of this repository produced.
 *
 * Call sites restated:  mjx.forward / mjx.step  — ambersim/rl/base.py:85,93;
 *                       ambersim/trajopt/shooting.py:36,41.   shoot — trajopt/shooting.py:22-48.
 *
 * Build: see oracle/Makefile (-DREAL=double -> libamxo_f64.so, -DREAL=float -> libamxo_f32.so; `native`: the fp32 build
 * with -O3 -march=native for bench.py's cpu_baseline leg, built on the machine that times it; `count`: the flop census).
 * Dense M, dense Cholesky, scalar loops in MJX's stage order; one env per call, OpenMP over envs.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "../include/amx.h"

#ifndef REAL
#define REAL double
#endif
#define IS_FLOAT (sizeof(REAL) == 4)
/* Flop census build (oracle/Makefile: libamxo_count.so): this file compiled as C++ with REAL = CountReal
 * (oracle/count_real.hpp), every arithmetic operation on a REAL adds one to amxo_flop_count. */
#ifdef AMXO_COUNT_FLOPS
thread_local double amxo_flop_count = 0;
#define FLOP1() (amxo_flop_count += 1)
extern "C" {
double amxo_flops_read_and_reset(void) { double c = amxo_flop_count; amxo_flop_count = 0; return c; }
#else
#define FLOP1() ((void)0)
#endif
static inline REAL r_sqrt(REAL x) { FLOP1(); return IS_FLOAT ? (REAL)sqrtf((float)x) : (REAL)sqrt((double)x); }
static inline REAL r_sin(REAL x) { FLOP1(); return IS_FLOAT ? (REAL)sinf((float)x) : (REAL)sin((double)x); }
static inline REAL r_cos(REAL x) { FLOP1(); return IS_FLOAT ? (REAL)cosf((float)x) : (REAL)cos((double)x); }
static inline REAL r_pow(REAL x, REAL y) { FLOP1(); return IS_FLOAT ? (REAL)powf((float)x, (float)y) : (REAL)pow((double)x, (double)y); }
static inline REAL r_atan2(REAL y, REAL x) { FLOP1(); return IS_FLOAT ? (REAL)atan2f((float)y, (float)x) : (REAL)atan2((double)y, (double)x); }
static inline REAL r_abs(REAL x) { return x < 0 ? -x : x; }
static inline REAL r_max(REAL a, REAL b) { return a > b ? a : b; }
static inline REAL r_min(REAL a, REAL b) { return a < b ? a : b; }
static inline REAL r_clip(REAL x, REAL lo, REAL hi) { return x < lo ? lo : (x > hi ? hi : x); }

#define MINVAL ((REAL)1e-15)
#define MINIMP ((REAL)0.0001)
#define MAXIMP ((REAL)0.9999)
enum { JFREE = 0, JBALL = 1, JSLIDE = 2, JHINGE = 3 };
enum { DIS_CONSTRAINT = 1, DIS_EQUALITY = 2, DIS_FRICTIONLOSS = 4, DIS_LIMIT = 8, DIS_CONTACT = 16, DIS_PASSIVE = 32,
       DIS_GRAVITY = 64, DIS_CLAMPCTRL = 128, DIS_WARMSTART = 256, DIS_ACTUATION = 1024,
       DIS_REFSAFE = 2048, DIS_EULERDAMP = 16384 };

/* ------------------------------------------------------------------ model (REAL copies) */
typedef struct amxo_model {
  int nq, nv, nu, nbody, njnt, ngeom, neq, npair, nvert, ncon, nefc_max;
  REAL timestep, gravity[3], tolerance, ls_tolerance, impratio, meaninertia;
  int iterations, ls_iterations, solver, integrator, disableflags;
  /* line-search rule (amxo_model_set_int "ls_rule"; 0 = MJX's formulation, the default).  Bit 0: the HIP kernel's ACCEPTANCE rule -
   * a step is taken on the sign of cost(alpha) - cost(0) = alpha^2 q2 + alpha q1 + sum_rows (q0_row(alpha) - q0_row(0)), and that
   * decrease is the solver's `improvement` (ambersim_amd/csrc/amx_kernels.hip: Env::linesearch, Env::solve).  Bit 1: the kernel's
   * ARITHMETIC around it - trial points alpha - d0 * (1 / d1), d1 = max(2 q2, MINVAL), the stopping tests multiplied out, the
   * quadratic part of the decrease taken from the bracket ends' derivatives.  Same algebra, so in exact arithmetic every
   * decision is the same; tests/test_ls_rule.py shows it for fp64 and measures where fp32 parts ways. */
  int ls_rule, ls_stats_on;
  long long ls_stat[12]; /* line searches, accepted under rule 0, accepted under rule 1, accept decisions that differ, Newton iterations,
                         * solver calls, both accept but pick different bracket ends, solver calls with a differing accept decision,
                         * [8] stopping tests on the improvement that differ between `prev_cost - cost` and the line search's own decrease */
  double ls_gap_max;    /* largest |decrease(lo) - decrease(hi)| / max(|decrease|) over the differing bracket-end picks: they are ties */
  double ls_stop_max;   /* largest |(prev_cost - cost) - decrease| / max(|cost|, |prev_cost|, tolerance * scale) over the differing stopping
                         * tests: the cancellation noise of two full costs */
  double ls_acc_max;    /* largest |decrease| / |cost(0)| over the differing accept decisions: improvements the full-cost comparison cannot resolve */
  int *body_parentid, *body_rootid, *body_jntadr, *body_jntnum, *body_dofadr, *body_dofnum;
  REAL *body_pos, *body_quat, *body_ipos, *body_iquat, *body_mass, *body_inertia, *body_invweight0;
  int *jnt_type, *jnt_qposadr, *jnt_dofadr, *jnt_bodyid, *jnt_limited;
  REAL *jnt_pos, *jnt_axis, *jnt_range, *jnt_margin, *jnt_stiffness, *jnt_solref, *jnt_solimp;
  REAL *qpos0, *qpos_spring;
  int *dof_bodyid, *dof_jntid, *dof_parentid;
  REAL *dof_armature, *dof_damping, *dof_invweight0, *dof_frictionloss, *dof_solref, *dof_solimp;
  int *geom_type, *geom_bodyid;
  REAL *geom_pos, *geom_quat, *geom_size;
  int *pair_geom1, *pair_geom2, *pair_fn, *pair_vertadr, *pair_vertnum, *pair_conadr;
  REAL *pair_friction, *pair_solref, *pair_solimp, *pair_margin, *convex_vert;
  int *actuator_trnid, *actuator_biastype, *actuator_ctrllimited, *actuator_forcelimited;
  REAL *actuator_gear, *actuator_gainprm, *actuator_biasprm, *actuator_ctrlrange, *actuator_forcerange;
  int *eq_obj1id, *eq_obj2id, *eq_active, *eq_type;
  REAL *eq_data, *eq_solref, *eq_solimp;
  /* convex hulls of the convex-convex pairs */
  int *pair_hull1, *pair_hull2, *hull_vertadr, *hull_vertnum, *hull_faceadr, *hull_facenum, *hull_edgeadr, *hull_edgenum;
  int *hull_face_vertadr, *hull_face_vertnum, *hull_facevert, *hull_edge;
  REAL *hull_vert, *hull_face_normal;
} amxo_model;

static int *dup_i(const int *src, size_t n) {
  int *p = (int *)malloc((n ? n : 1) * sizeof(int));
  if (n) memcpy(p, src, n * sizeof(int));
  return p;
}
static REAL *dup_r(const double *src, size_t n) {
  REAL *p = (REAL *)malloc((n ? n : 1) * sizeof(REAL));
  for (size_t i = 0; i < n; i++) p[i] = (REAL)src[i];
  return p;
}

static const int PAIR_NCON[12] = {1, 2, 4, 1, 1, 1, 4, 1, 1, 1, 1, 2}; /* 11: capsule - convex (8 - 10 are the planner's row groups, never in pair_fn) */

amxo_model *amxo_model_create(const amx_model_desc *d) {
  amxo_model *m = (amxo_model *)calloc(1, sizeof(amxo_model));
  m->nq = d->nq; m->nv = d->nv; m->nu = d->nu; m->nbody = d->nbody; m->njnt = d->njnt;
  m->ngeom = d->ngeom; m->neq = d->neq; m->npair = d->npair; m->nvert = d->nconvexvert;
  m->timestep = (REAL)d->timestep; m->tolerance = (REAL)d->tolerance; m->ls_tolerance = (REAL)d->ls_tolerance;
  m->impratio = (REAL)d->impratio; m->meaninertia = (REAL)d->meaninertia;
  for (int i = 0; i < 3; i++) m->gravity[i] = (REAL)d->gravity[i];
  m->iterations = d->iterations; m->ls_iterations = d->ls_iterations; m->solver = d->solver;
  m->integrator = d->integrator; m->disableflags = d->disableflags;
  size_t nb = d->nbody, nj = d->njnt, nv = d->nv, ng = d->ngeom, np = d->npair, nu = d->nu, ne = d->neq;
  m->body_parentid = dup_i(d->body_parentid, nb); m->body_rootid = dup_i(d->body_rootid, nb);
  m->body_jntadr = dup_i(d->body_jntadr, nb); m->body_jntnum = dup_i(d->body_jntnum, nb);
  m->body_dofadr = dup_i(d->body_dofadr, nb); m->body_dofnum = dup_i(d->body_dofnum, nb);
  m->body_pos = dup_r(d->body_pos, nb * 3); m->body_quat = dup_r(d->body_quat, nb * 4);
  m->body_ipos = dup_r(d->body_ipos, nb * 3); m->body_iquat = dup_r(d->body_iquat, nb * 4);
  m->body_mass = dup_r(d->body_mass, nb); m->body_inertia = dup_r(d->body_inertia, nb * 3);
  m->body_invweight0 = dup_r(d->body_invweight0, nb * 2);
  m->jnt_type = dup_i(d->jnt_type, nj); m->jnt_qposadr = dup_i(d->jnt_qposadr, nj);
  m->jnt_dofadr = dup_i(d->jnt_dofadr, nj); m->jnt_bodyid = dup_i(d->jnt_bodyid, nj);
  m->jnt_limited = dup_i(d->jnt_limited, nj);
  m->jnt_pos = dup_r(d->jnt_pos, nj * 3); m->jnt_axis = dup_r(d->jnt_axis, nj * 3);
  m->jnt_range = dup_r(d->jnt_range, nj * 2); m->jnt_margin = dup_r(d->jnt_margin, nj);
  m->jnt_stiffness = dup_r(d->jnt_stiffness, nj); m->jnt_solref = dup_r(d->jnt_solref, nj * 2);
  m->jnt_solimp = dup_r(d->jnt_solimp, nj * 5);
  m->qpos0 = dup_r(d->qpos0, d->nq); m->qpos_spring = dup_r(d->qpos_spring, d->nq);
  m->dof_bodyid = dup_i(d->dof_bodyid, nv); m->dof_jntid = dup_i(d->dof_jntid, nv);
  m->dof_parentid = dup_i(d->dof_parentid, nv);
  m->dof_armature = dup_r(d->dof_armature, nv); m->dof_damping = dup_r(d->dof_damping, nv);
  m->dof_frictionloss = dup_r(d->dof_frictionloss, nv); m->dof_solref = dup_r(d->dof_solref, nv * 2); m->dof_solimp = dup_r(d->dof_solimp, nv * 5);
  m->dof_invweight0 = dup_r(d->dof_invweight0, nv);
  m->geom_type = dup_i(d->geom_type, ng); m->geom_bodyid = dup_i(d->geom_bodyid, ng);
  m->geom_pos = dup_r(d->geom_pos, ng * 3); m->geom_quat = dup_r(d->geom_quat, ng * 4);
  m->geom_size = dup_r(d->geom_size, ng * 3);
  m->pair_geom1 = dup_i(d->pair_geom1, np); m->pair_geom2 = dup_i(d->pair_geom2, np);
  m->pair_fn = dup_i(d->pair_fn, np); m->pair_vertadr = dup_i(d->pair_vertadr, np);
  m->pair_vertnum = dup_i(d->pair_vertnum, np);
  m->pair_friction = dup_r(d->pair_friction, np * 3); m->pair_solref = dup_r(d->pair_solref, np * 2);
  m->pair_solimp = dup_r(d->pair_solimp, np * 5); m->pair_margin = dup_r(d->pair_margin, np);
  m->convex_vert = dup_r(d->convex_vert, (size_t)d->nconvexvert * 3);
  m->actuator_trnid = dup_i(d->actuator_trnid, nu); m->actuator_biastype = dup_i(d->actuator_biastype, nu);
  m->actuator_ctrllimited = dup_i(d->actuator_ctrllimited, nu);
  m->actuator_forcelimited = dup_i(d->actuator_forcelimited, nu);
  m->actuator_gear = dup_r(d->actuator_gear, nu); m->actuator_gainprm = dup_r(d->actuator_gainprm, nu * 3);
  m->actuator_biasprm = dup_r(d->actuator_biasprm, nu * 3);
  m->actuator_ctrlrange = dup_r(d->actuator_ctrlrange, nu * 2);
  m->actuator_forcerange = dup_r(d->actuator_forcerange, nu * 2);
  m->eq_obj1id = dup_i(d->eq_obj1id, ne); m->eq_obj2id = dup_i(d->eq_obj2id, ne);
  m->eq_active = dup_i(d->eq_active, ne);
  m->eq_type = dup_i(d->eq_type, ne);
  m->eq_data = dup_r(d->eq_data, ne * 11); m->eq_solref = dup_r(d->eq_solref, ne * 2);
  m->eq_solimp = dup_r(d->eq_solimp, ne * 5);
  {
    size_t nh = d->nhull, nf = d->nhullface;
    m->pair_hull1 = dup_i(d->pair_hull1, np); m->pair_hull2 = dup_i(d->pair_hull2, np);
    m->hull_vertadr = dup_i(d->hull_vertadr, nh); m->hull_vertnum = dup_i(d->hull_vertnum, nh);
    m->hull_faceadr = dup_i(d->hull_faceadr, nh); m->hull_facenum = dup_i(d->hull_facenum, nh);
    m->hull_edgeadr = dup_i(d->hull_edgeadr, nh); m->hull_edgenum = dup_i(d->hull_edgenum, nh);
    m->hull_vert = dup_r(d->hull_vert, (size_t)d->nhullvert * 3); m->hull_face_normal = dup_r(d->hull_face_normal, nf * 3);
    m->hull_face_vertadr = dup_i(d->hull_face_vertadr, nf); m->hull_face_vertnum = dup_i(d->hull_face_vertnum, nf);
    m->hull_facevert = dup_i(d->hull_facevert, d->nhullfacevert); m->hull_edge = dup_i(d->hull_edge, (size_t)d->nhulledge * 2);
  }
  m->pair_conadr = (int *)malloc((np ? np : 1) * sizeof(int));
  m->ncon = 0;
  for (size_t p = 0; p < np; p++) { m->pair_conadr[p] = m->ncon; m->ncon += PAIR_NCON[m->pair_fn[p]]; }
  int nlim = 0;
  for (size_t j = 0; j < nj; j++) nlim += (m->jnt_limited[j] && (m->jnt_type[j] == JSLIDE || m->jnt_type[j] == JHINGE || m->jnt_type[j] == JBALL));
  int nfri = 0, neqrows = 0;
  for (size_t i = 0; i < nv; i++) nfri += m->dof_frictionloss[i] > 0;
  for (int e = 0; e < m->neq; e++) neqrows += m->eq_type[e] == 0 ? 3 : (m->eq_type[e] == 1 ? 6 : 1);
  m->nefc_max = neqrows + nfri + nlim + 4 * m->ncon;
  return m;
}

void amxo_model_destroy(amxo_model *m) {
  if (!m) return;
  free(m->pair_hull1); free(m->pair_hull2); free(m->hull_vertadr); free(m->hull_vertnum); free(m->hull_faceadr); free(m->hull_facenum);
  free(m->hull_edgeadr); free(m->hull_edgenum); free(m->hull_vert); free(m->hull_face_normal); free(m->hull_face_vertadr);
  free(m->hull_face_vertnum); free(m->hull_facevert); free(m->hull_edge);
  void **ptrs[] = {(void **)&m->body_parentid, (void **)&m->body_rootid, (void **)&m->body_jntadr, (void **)&m->body_jntnum,
    (void **)&m->body_dofadr, (void **)&m->body_dofnum, (void **)&m->body_pos, (void **)&m->body_quat, (void **)&m->body_ipos,
    (void **)&m->body_iquat, (void **)&m->body_mass, (void **)&m->body_inertia, (void **)&m->body_invweight0,
    (void **)&m->jnt_type, (void **)&m->jnt_qposadr, (void **)&m->jnt_dofadr, (void **)&m->jnt_bodyid, (void **)&m->jnt_limited,
    (void **)&m->jnt_pos, (void **)&m->jnt_axis, (void **)&m->jnt_range, (void **)&m->jnt_margin, (void **)&m->jnt_stiffness,
    (void **)&m->jnt_solref, (void **)&m->jnt_solimp, (void **)&m->qpos0, (void **)&m->qpos_spring, (void **)&m->dof_bodyid,
    (void **)&m->dof_jntid, (void **)&m->dof_parentid, (void **)&m->dof_armature, (void **)&m->dof_damping, (void **)&m->dof_frictionloss, (void **)&m->dof_solref, (void **)&m->dof_solimp,
    (void **)&m->dof_invweight0, (void **)&m->geom_type, (void **)&m->geom_bodyid, (void **)&m->geom_pos, (void **)&m->geom_quat,
    (void **)&m->geom_size, (void **)&m->pair_geom1, (void **)&m->pair_geom2, (void **)&m->pair_fn, (void **)&m->pair_vertadr,
    (void **)&m->pair_vertnum, (void **)&m->pair_conadr, (void **)&m->pair_friction, (void **)&m->pair_solref,
    (void **)&m->pair_solimp, (void **)&m->pair_margin, (void **)&m->convex_vert, (void **)&m->actuator_trnid,
    (void **)&m->actuator_biastype, (void **)&m->actuator_ctrllimited, (void **)&m->actuator_forcelimited,
    (void **)&m->actuator_gear, (void **)&m->actuator_gainprm, (void **)&m->actuator_biasprm, (void **)&m->actuator_ctrlrange,
    (void **)&m->actuator_forcerange, (void **)&m->eq_obj1id, (void **)&m->eq_obj2id, (void **)&m->eq_active,
    (void **)&m->eq_data, (void **)&m->eq_solref, (void **)&m->eq_solimp, (void **)&m->eq_type};
  for (size_t i = 0; i < sizeof(ptrs) / sizeof(ptrs[0]); i++) free(*ptrs[i]);
  free(m);
}

/* ------------------------------------------------------------------ data */
typedef struct amxo_data {
  REAL *qpos, *qvel, *ctrl, *qacc_warmstart, time;
  REAL *xpos, *xquat, *xmat, *xipos, *ximat, *xanchor, *xaxis, *geom_xpos, *geom_xmat;
  REAL *subtree_com, *cinert, *cdof, *crb, *qM, *qL;
  REAL *cvel, *cdof_dot, *cacc, *cfrc;
  REAL *qfrc_passive, *qfrc_bias, *qfrc_actuator, *qfrc_smooth, *qacc_smooth, *qacc, *qfrc_constraint;
  REAL *actuator_moment_gear; /* per actuator: gear on its dof */
  REAL *con_dist, *con_pos, *con_frame; int *con_pair;
  int nefc, ne, nf; /* rows: [0, ne) equality, [ne, ne + nf) friction loss, then limits, then contacts */
  REAL *efc_J, *efc_D, *efc_aref, *efc_pos, *efc_force, *efc_floss; /* efc_floss: the friction rows' force bound, 0 elsewhere */
  int solver_niter;
  long long ls_stat[12];
  double ls_gap_max, ls_acc_max, ls_stop_max; int ls_differed;
  /* scratch */
  REAL *w_a, *w_b, *w_c, *w_d, *w_e, *w_f, *w_g, *w_H, *w_L, *w_jv, *w_jaref, *w_mass, *w_quad;
  REAL *w_support;
} amxo_data;

#define ALLOC(field, n) d->field = (REAL *)calloc((size_t)((n) > 0 ? (n) : 1), sizeof(REAL))

amxo_data *amxo_data_create(const amxo_model *m) {
  amxo_data *d = (amxo_data *)calloc(1, sizeof(amxo_data));
  int nb = m->nbody, nv = m->nv, nq = m->nq, nj = m->njnt, ng = m->ngeom, nc = m->ncon, ne = m->nefc_max;
  ALLOC(qpos, nq); ALLOC(qvel, nv); ALLOC(ctrl, m->nu); ALLOC(qacc_warmstart, nv);
  ALLOC(xpos, nb * 3); ALLOC(xquat, nb * 4); ALLOC(xmat, nb * 9); ALLOC(xipos, nb * 3); ALLOC(ximat, nb * 9);
  ALLOC(xanchor, nj * 3); ALLOC(xaxis, nj * 3); ALLOC(geom_xpos, ng * 3); ALLOC(geom_xmat, ng * 9);
  ALLOC(subtree_com, nb * 3); ALLOC(cinert, nb * 10); ALLOC(cdof, nv * 6); ALLOC(crb, nb * 10);
  ALLOC(qM, nv * nv); ALLOC(qL, nv * nv);
  ALLOC(cvel, nb * 6); ALLOC(cdof_dot, nv * 6); ALLOC(cacc, nb * 6); ALLOC(cfrc, nb * 6);
  ALLOC(qfrc_passive, nv); ALLOC(qfrc_bias, nv); ALLOC(qfrc_actuator, nv); ALLOC(qfrc_smooth, nv);
  ALLOC(qacc_smooth, nv); ALLOC(qacc, nv); ALLOC(qfrc_constraint, nv); ALLOC(actuator_moment_gear, m->nu);
  ALLOC(con_dist, nc); ALLOC(con_pos, nc * 3); ALLOC(con_frame, nc * 9);
  d->con_pair = (int *)calloc((size_t)(nc > 0 ? nc : 1), sizeof(int));
  ALLOC(efc_J, ne * nv); ALLOC(efc_D, ne); ALLOC(efc_aref, ne); ALLOC(efc_pos, ne); ALLOC(efc_force, ne); ALLOC(efc_floss, ne);
  ALLOC(w_a, nv); ALLOC(w_b, nv); ALLOC(w_c, nv); ALLOC(w_d, nv); ALLOC(w_e, nv); ALLOC(w_f, nv); ALLOC(w_g, nv);
  ALLOC(w_H, nv * nv + 3 * nv); ALLOC(w_L, nv * nv);
  ALLOC(w_jv, ne); ALLOC(w_jaref, ne); ALLOC(w_mass, nb * 4); ALLOC(w_quad, ne * 3);
  ALLOC(w_support, m->nvert);
  memcpy(d->qpos, m->qpos0, sizeof(REAL) * nq);
  return d;
}

void amxo_data_destroy(amxo_data *d) {
  if (!d) return;
  REAL **ptrs[] = {&d->qpos, &d->qvel, &d->ctrl, &d->qacc_warmstart, &d->xpos, &d->xquat, &d->xmat, &d->xipos, &d->ximat,
    &d->xanchor, &d->xaxis, &d->geom_xpos, &d->geom_xmat, &d->subtree_com, &d->cinert, &d->cdof, &d->crb, &d->qM, &d->qL,
    &d->cvel, &d->cdof_dot, &d->cacc, &d->cfrc, &d->qfrc_passive, &d->qfrc_bias, &d->qfrc_actuator, &d->qfrc_smooth,
    &d->qacc_smooth, &d->qacc, &d->qfrc_constraint, &d->actuator_moment_gear, &d->con_dist, &d->con_pos, &d->con_frame,
    &d->efc_J, &d->efc_D, &d->efc_aref, &d->efc_pos, &d->efc_force, &d->efc_floss, &d->w_a, &d->w_b, &d->w_c, &d->w_d, &d->w_e, &d->w_f, &d->w_g, &d->w_H, &d->w_L,
    &d->w_jv, &d->w_jaref, &d->w_mass, &d->w_quad, &d->w_support};
  for (size_t i = 0; i < sizeof(ptrs) / sizeof(ptrs[0]); i++) free(*ptrs[i]);
  free(d->con_pair);
  free(d);
}

/* ------------------------------------------------------------------ small math [mjx:math] */
static void quat_mul(REAL *r, const REAL *a, const REAL *b) {
  REAL t0 = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  REAL t1 = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  REAL t2 = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  REAL t3 = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  r[0] = t0; r[1] = t1; r[2] = t2; r[3] = t3;
}
static void quat_to_mat(REAL *m, const REAL *q) {
  REAL w = q[0], x = q[1], y = q[2], z = q[3];
  m[0] = w * w + x * x - y * y - z * z; m[1] = 2 * (x * y - w * z); m[2] = 2 * (x * z + w * y);
  m[3] = 2 * (x * y + w * z); m[4] = w * w - x * x + y * y - z * z; m[5] = 2 * (y * z - w * x);
  m[6] = 2 * (x * z - w * y); m[7] = 2 * (y * z + w * x); m[8] = w * w - x * x - y * y + z * z;
}
static void rotate(REAL *r, const REAL *v, const REAL *q) { /* r = R(q) v */
  REAL m[9]; quat_to_mat(m, q);
  REAL a = m[0] * v[0] + m[1] * v[1] + m[2] * v[2];
  REAL b = m[3] * v[0] + m[4] * v[1] + m[5] * v[2];
  REAL c = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
  r[0] = a; r[1] = b; r[2] = c;
}
static void cross3(REAL *r, const REAL *a, const REAL *b) {
  REAL x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
static REAL dot3(const REAL *a, const REAL *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static REAL normalize_n(REAL *v, int n) {
  REAL s = 0; for (int i = 0; i < n; i++) s += v[i] * v[i];
  s = r_sqrt(s);
  if (s > 0) for (int i = 0; i < n; i++) v[i] /= s;
  return s;
}
static void axis_angle_quat(REAL *q, const REAL *axis, REAL angle) {
  REAL s = r_sin(angle * (REAL)0.5);
  q[0] = r_cos(angle * (REAL)0.5); q[1] = axis[0] * s; q[2] = axis[1] * s; q[3] = axis[2] * s;
}
static void mat_mul_vec3(REAL *r, const REAL *m, const REAL *v) {
  REAL a = m[0] * v[0] + m[1] * v[1] + m[2] * v[2], b = m[3] * v[0] + m[4] * v[1] + m[5] * v[2],
       c = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
  r[0] = a; r[1] = b; r[2] = c;
}
static void matT_mul_vec3(REAL *r, const REAL *m, const REAL *v) {
  REAL a = m[0] * v[0] + m[3] * v[1] + m[6] * v[2], b = m[1] * v[0] + m[4] * v[1] + m[7] * v[2],
       c = m[2] * v[0] + m[5] * v[1] + m[8] * v[2];
  r[0] = a; r[1] = b; r[2] = c;
}
/* spatial algebra: vectors are [rotational(3); translational(3)] */
static void inert_mul(REAL *res, const REAL *i, const REAL *v) {
  res[0] = i[0] * v[0] + i[3] * v[1] + i[4] * v[2] - i[8] * v[4] + i[7] * v[5];
  res[1] = i[3] * v[0] + i[1] * v[1] + i[5] * v[2] + i[8] * v[3] - i[6] * v[5];
  res[2] = i[4] * v[0] + i[5] * v[1] + i[2] * v[2] - i[7] * v[3] + i[6] * v[4];
  res[3] = i[8] * v[1] - i[7] * v[2] + i[9] * v[3];
  res[4] = i[6] * v[2] - i[8] * v[0] + i[9] * v[4];
  res[5] = i[7] * v[0] - i[6] * v[1] + i[9] * v[5];
}
static void motion_cross(REAL *res, const REAL *vel, const REAL *v) {
  REAL t[6];
  cross3(t, vel, v);
  REAL a[3], b[3];
  cross3(a, vel, v + 3); cross3(b, vel + 3, v);
  t[3] = a[0] + b[0]; t[4] = a[1] + b[1]; t[5] = a[2] + b[2];
  memcpy(res, t, sizeof(t));
}
static void motion_cross_force(REAL *res, const REAL *vel, const REAL *f) {
  REAL a[3], b[3], c[3];
  cross3(a, vel, f); cross3(b, vel + 3, f + 3); cross3(c, vel, f + 3);
  res[0] = a[0] + b[0]; res[1] = a[1] + b[1]; res[2] = a[2] + b[2];
  res[3] = c[0]; res[4] = c[1]; res[5] = c[2];
}
static void make_frame(REAL *frame, const REAL *normal) { /* rows: a, b, a x b  [mjx:math.make_frame] */
  REAL a[3] = {normal[0], normal[1], normal[2]};
  normalize_n(a, 3);
  REAL b[3] = {0, 0, 0};
  if (a[1] > (REAL)-0.5 && a[1] < (REAL)0.5) b[1] = 1; else b[2] = 1;
  REAL ab = dot3(a, b);
  for (int i = 0; i < 3; i++) b[i] -= a[i] * ab;
  normalize_n(b, 3);
  memcpy(frame, a, sizeof(a)); memcpy(frame + 3, b, sizeof(b));
  cross3(frame + 6, a, b);
}

/* ------------------------------------------------------------------ A9.1 kinematics [mjx:smooth.kinematics] */
static void kinematics(const amxo_model *m, amxo_data *d) {
  REAL *xpos = d->xpos, *xquat = d->xquat;
  xpos[0] = xpos[1] = xpos[2] = 0; xquat[0] = 1; xquat[1] = xquat[2] = xquat[3] = 0;
  for (int b = 1; b < m->nbody; b++) {
    int p = m->body_parentid[b];
    REAL pos[3], quat[4];
    rotate(pos, m->body_pos + 3 * b, xquat + 4 * p);
    for (int i = 0; i < 3; i++) pos[i] += xpos[3 * p + i];
    quat_mul(quat, xquat + 4 * p, m->body_quat + 4 * b);
    for (int k = 0; k < m->body_jntnum[b]; k++) {
      int j = m->body_jntadr[b] + k, qa = m->jnt_qposadr[j], t = m->jnt_type[j];
      REAL *anchor = d->xanchor + 3 * j, *axis = d->xaxis + 3 * j;
      if (t == JFREE) {
        for (int i = 0; i < 3; i++) anchor[i] = d->qpos[qa + i];
        axis[0] = 0; axis[1] = 0; axis[2] = 1;
        for (int i = 0; i < 3; i++) pos[i] = d->qpos[qa + i];
        normalize_n(d->qpos + qa + 3, 4);
        for (int i = 0; i < 4; i++) quat[i] = d->qpos[qa + 3 + i];
      } else {
        rotate(anchor, m->jnt_pos + 3 * j, quat);
        for (int i = 0; i < 3; i++) anchor[i] += pos[i];
        rotate(axis, m->jnt_axis + 3 * j, quat);
        if (t == JBALL) {
          normalize_n(d->qpos + qa, 4);
          quat_mul(quat, quat, d->qpos + qa);
          REAL off[3]; rotate(off, m->jnt_pos + 3 * j, quat);
          for (int i = 0; i < 3; i++) pos[i] = anchor[i] - off[i];
        } else if (t == JHINGE) {
          REAL ql[4]; axis_angle_quat(ql, m->jnt_axis + 3 * j, d->qpos[qa] - m->qpos0[qa]);
          quat_mul(quat, quat, ql);
          REAL off[3]; rotate(off, m->jnt_pos + 3 * j, quat);
          for (int i = 0; i < 3; i++) pos[i] = anchor[i] - off[i];
        } else { /* slide */
          REAL s = d->qpos[qa] - m->qpos0[qa];
          for (int i = 0; i < 3; i++) pos[i] += axis[i] * s;
        }
      }
    }
    memcpy(xpos + 3 * b, pos, sizeof(pos)); memcpy(xquat + 4 * b, quat, sizeof(quat));
  }
  for (int b = 0; b < m->nbody; b++) {
    quat_to_mat(d->xmat + 9 * b, xquat + 4 * b);
    REAL t[3], q[4];
    rotate(t, m->body_ipos + 3 * b, xquat + 4 * b);
    for (int i = 0; i < 3; i++) d->xipos[3 * b + i] = xpos[3 * b + i] + t[i];
    quat_mul(q, xquat + 4 * b, m->body_iquat + 4 * b);
    quat_to_mat(d->ximat + 9 * b, q);
  }
  for (int g = 0; g < m->ngeom; g++) {
    int b = m->geom_bodyid[g];
    REAL t[3], q[4];
    rotate(t, m->geom_pos + 3 * g, xquat + 4 * b);
    for (int i = 0; i < 3; i++) d->geom_xpos[3 * g + i] = xpos[3 * b + i] + t[i];
    quat_mul(q, xquat + 4 * b, m->geom_quat + 4 * g);
    quat_to_mat(d->geom_xmat + 9 * g, q);
  }
}

/* ------------------------------------------------------------------ A9.2 com_pos [mjx:smooth.com_pos] */
static void com_pos(const amxo_model *m, amxo_data *d) {
  int nb = m->nbody;
  REAL *acc = d->w_mass; /* [nb,4]: mass-weighted position sum, mass sum */
  for (int b = 0; b < nb; b++) {
    for (int i = 0; i < 3; i++) acc[4 * b + i] = d->xipos[3 * b + i] * m->body_mass[b];
    acc[4 * b + 3] = m->body_mass[b];
  }
  for (int b = nb - 1; b > 0; b--) {
    int p = m->body_parentid[b];
    for (int i = 0; i < 4; i++) acc[4 * p + i] += acc[4 * b + i];
  }
  for (int b = 0; b < nb; b++)
    for (int i = 0; i < 3; i++)
      d->subtree_com[3 * b + i] = acc[4 * b + 3] < MINVAL ? d->xipos[3 * b + i] : acc[4 * b + i] / acc[4 * b + 3];
  for (int b = 0; b < nb; b++) { /* cinert about the root's subtree com */
    const REAL *com = d->subtree_com + 3 * m->body_rootid[b], *mat = d->ximat + 9 * b, *in = m->body_inertia + 3 * b;
    REAL h[3], mass = m->body_mass[b], *r = d->cinert + 10 * b;
    for (int i = 0; i < 3; i++) h[i] = d->xipos[3 * b + i] - com[i];
    REAL I[9];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++)
        I[3 * i + j] = mat[3 * i] * in[0] * mat[3 * j] + mat[3 * i + 1] * in[1] * mat[3 * j + 1] + mat[3 * i + 2] * in[2] * mat[3 * j + 2];
    r[0] = I[0] + mass * (h[1] * h[1] + h[2] * h[2]);
    r[1] = I[4] + mass * (h[0] * h[0] + h[2] * h[2]);
    r[2] = I[8] + mass * (h[0] * h[0] + h[1] * h[1]);
    r[3] = I[1] - mass * h[0] * h[1]; r[4] = I[2] - mass * h[0] * h[2]; r[5] = I[5] - mass * h[1] * h[2];
    r[6] = mass * h[0]; r[7] = mass * h[1]; r[8] = mass * h[2]; r[9] = mass;
  }
  for (int j = 0; j < m->njnt; j++) { /* cdof */
    int b = m->jnt_bodyid[j], da = m->jnt_dofadr[j], t = m->jnt_type[j];
    const REAL *com = d->subtree_com + 3 * m->body_rootid[b];
    REAL off[3];
    for (int i = 0; i < 3; i++) off[i] = com[i] - d->xanchor[3 * j + i];
    if (t == JFREE || t == JBALL) {
      int ra = da;
      if (t == JFREE) {
        for (int k = 0; k < 3; k++) { REAL *c = d->cdof + 6 * (da + k); memset(c, 0, 6 * sizeof(REAL)); c[3 + k] = 1; }
        ra = da + 3;
      }
      for (int k = 0; k < 3; k++) {
        REAL ax[3] = {d->xmat[9 * b + k], d->xmat[9 * b + 3 + k], d->xmat[9 * b + 6 + k]};
        REAL *c = d->cdof + 6 * (ra + k);
        memcpy(c, ax, sizeof(ax)); cross3(c + 3, ax, off);
      }
    } else if (t == JHINGE) {
      REAL *c = d->cdof + 6 * da;
      memcpy(c, d->xaxis + 3 * j, 3 * sizeof(REAL)); cross3(c + 3, d->xaxis + 3 * j, off);
    } else {
      REAL *c = d->cdof + 6 * da;
      c[0] = c[1] = c[2] = 0; memcpy(c + 3, d->xaxis + 3 * j, 3 * sizeof(REAL));
    }
  }
}

/* ------------------------------------------------------------------ A9.3 crb, A9.4 factor_m [mjx:smooth.crb / factor_m] */
static void crb(const amxo_model *m, amxo_data *d) {
  int nb = m->nbody, nv = m->nv;
  memcpy(d->crb, d->cinert, sizeof(REAL) * 10 * nb);
  for (int b = nb - 1; b > 0; b--) {
    int p = m->body_parentid[b];
    if (p > 0) for (int i = 0; i < 10; i++) d->crb[10 * p + i] += d->crb[10 * b + i];
  }
  memset(d->crb, 0, sizeof(REAL) * 10);
  memset(d->qM, 0, sizeof(REAL) * nv * nv);
  for (int i = 0; i < nv; i++) {
    REAL f[6];
    inert_mul(f, d->crb + 10 * m->dof_bodyid[i], d->cdof + 6 * i);
    for (int j = i; j >= 0; j = m->dof_parentid[j]) {
      REAL s = 0;
      for (int k = 0; k < 6; k++) s += d->cdof[6 * j + k] * f[k];
      d->qM[i * nv + j] = d->qM[j * nv + i] = s;
    }
    d->qM[i * nv + i] += m->dof_armature[i];
  }
}
/* dense Cholesky A = L L^T (lower), returns 0 on success */
static int chol_factor(REAL *L, const REAL *A, int n) {
  memcpy(L, A, sizeof(REAL) * n * n);
  for (int j = 0; j < n; j++) {
    REAL s = L[j * n + j];
    for (int k = 0; k < j; k++) s -= L[j * n + k] * L[j * n + k];
    if (!(s > 0)) s = MINVAL;
    REAL dj = r_sqrt(s);
    L[j * n + j] = dj;
    for (int i = j + 1; i < n; i++) {
      REAL t = L[i * n + j];
      for (int k = 0; k < j; k++) t -= L[i * n + k] * L[j * n + k];
      L[i * n + j] = t / dj;
    }
  }
  return 0;
}
static void chol_solve(const REAL *L, REAL *x, const REAL *b, int n) {
  for (int i = 0; i < n; i++) {
    REAL t = b[i];
    for (int k = 0; k < i; k++) t -= L[i * n + k] * x[k];
    x[i] = t / L[i * n + i];
  }
  for (int i = n - 1; i >= 0; i--) {
    REAL t = x[i];
    for (int k = i + 1; k < n; k++) t -= L[k * n + i] * x[k];
    x[i] = t / L[i * n + i];
  }
}
static void mul_m(const amxo_model *m, const amxo_data *d, REAL *res, const REAL *v) {
  int nv = m->nv;
  for (int i = 0; i < nv; i++) {
    REAL s = 0;
    for (int j = 0; j < nv; j++) s += d->qM[i * nv + j] * v[j];
    res[i] = s;
  }
}

/* ------------------------------------------------------------------ A9.5 collision [mjx:collision_driver / collision_primitive / collision_convex] */
static void plane_sphere_core(const REAL *n, const REAL *ppos, const REAL *spos, REAL radius, REAL *dist, REAL *pos) {
  REAL dv[3] = {spos[0] - ppos[0], spos[1] - ppos[1], spos[2] - ppos[2]};
  *dist = dot3(dv, n) - radius;
  for (int i = 0; i < 3; i++) pos[i] = spos[i] - n[i] * (radius + (REAL)0.5 * (*dist));
}
static int argmax_r(const REAL *v, int n) { int k = 0; for (int i = 1; i < n; i++) if (v[i] > v[k]) k = i; return k; }

/* ------------------------------------------------------------------ A9.5c convex-convex [mjx:collision_convex.convex_convex]
 * Restated from the published description (mujoco-mjx is absent here): separating-axis test over the face
 * normals of both hulls and the cross products of their edge pairs; the axis of least penetration decides between a face
 * contact - reference face, incident face, Sutherland-Hodgman clipping of the incident polygon against the side planes of
 * the reference face, 4-point manifold chosen like plane_convex's - and an edge-edge contact (closest points of the two
 * edges; an edge pair is a candidate only if both edges are supporting edges along its axis).  Ties go to the lowest axis
 * index (faces of hull 1, faces of hull 2, edge pairs edge1-major).  Four contact
 * slots per pair; unused ones report dist = 1.  Contact normal points from geom1 to geom2. */
#define CX_MAXV 64
#define CX_MAXP 16
#ifdef __cplusplus
#define AMXO_TLS thread_local
#else
#define AMXO_TLS _Thread_local
#endif
static AMXO_TLS REAL cx_dbg[32]; /* this thread's last convex_convex call: np, idx[4], depth[16] (debugging aid for tools/dbg) */
typedef struct { int nv, nf, ne; const REAL *vert, *normal; const int *fvadr, *fvnum, *fvert, *edge; } hull_view;
static void hull_get(const amxo_model *m, int h, hull_view *H) {
  H->nv = m->hull_vertnum[h]; H->nf = m->hull_facenum[h]; H->ne = m->hull_edgenum[h];
  H->vert = m->hull_vert + 3 * m->hull_vertadr[h]; H->normal = m->hull_face_normal + 3 * m->hull_faceadr[h];
  H->fvadr = m->hull_face_vertadr + m->hull_faceadr[h]; H->fvnum = m->hull_face_vertnum + m->hull_faceadr[h];
  H->fvert = m->hull_facevert; H->edge = m->hull_edge + 2 * m->hull_edgeadr[h];
}
static void support_range(const REAL (*V)[3], int n, const REAL *ax, REAL *lo, REAL *hi) {
  REAL a = (REAL)1e30, b = (REAL)-1e30;
  for (int i = 0; i < n; i++) { REAL s = dot3(V[i], ax); if (s < a) a = s; if (s > b) b = s; }
  *lo = a; *hi = b;
}
static void convex_convex(const amxo_model *m, amxo_data *d, int p) {
  int g1 = m->pair_geom1[p], g2 = m->pair_geom2[p], c0 = m->pair_conadr[p];
  const REAL *pos1 = d->geom_xpos + 3 * g1, *mat1 = d->geom_xmat + 9 * g1, *pos2 = d->geom_xpos + 3 * g2, *mat2 = d->geom_xmat + 9 * g2;
  hull_view A, B;
  hull_get(m, m->pair_hull1[p], &A); hull_get(m, m->pair_hull2[p], &B);
  REAL VA[CX_MAXV][3], VB[CX_MAXV][3];
  for (int i = 0; i < A.nv; i++) { mat_mul_vec3(VA[i], mat1, A.vert + 3 * i); for (int k = 0; k < 3; k++) VA[i][k] += pos1[k]; }
  for (int i = 0; i < B.nv; i++) { mat_mul_vec3(VB[i], mat2, B.vert + 3 * i); for (int k = 0; k < 3; k++) VB[i][k] += pos2[k]; }
  /* --- separating axes.  Selections here and in the manifold are "the lowest index within a tolerance of the extreme", not
   * "the first strict extreme": parallel faces, and the manifold heuristic on a parallelogram of clipped points, produce exact
   * ties that rounding would otherwise break differently in every precision and instruction order. */
#define CX_TOL(x) ((REAL)1e-4 * r_abs(x) + (REAL)1e-7)
  int naxes = A.nf + B.nf + A.ne * B.ne, bt = 0;
  REAL best = (REAL)1e30, bsgn = 1, bn[3] = {0, 0, 1};
  for (int pass = 0; pass < 2; pass++) {
    for (int t = 0; t < naxes; t++) {
      REAL n[3];
      if (t < A.nf) mat_mul_vec3(n, mat1, A.normal + 3 * t);
      else if (t < A.nf + B.nf) mat_mul_vec3(n, mat2, B.normal + 3 * (t - A.nf));
      else {
        int ea = (t - A.nf - B.nf) / B.ne, eb = (t - A.nf - B.nf) % B.ne;
        REAL da[3], db[3];
        for (int k = 0; k < 3; k++) { da[k] = VA[A.edge[2 * ea + 1]][k] - VA[A.edge[2 * ea]][k]; db[k] = VB[B.edge[2 * eb + 1]][k] - VB[B.edge[2 * eb]][k]; }
        normalize_n(da, 3); normalize_n(db, 3);
        cross3(n, da, db);
        REAL c2 = dot3(n, n);
        if (c2 < (REAL)1e-6) continue; /* (nearly) parallel edges span no axis */
        REAL inv = 1 / r_sqrt(c2);
        for (int k = 0; k < 3; k++) n[k] *= inv;
      }
      REAL loA, hiA, loB, hiB;
      support_range(VA, A.nv, n, &loA, &hiA); support_range(VB, B.nv, n, &loB, &hiB);
      REAL d1 = hiA - loB, d2 = hiB - loA;
      REAL pen = d1 < d2 ? d1 : d2, sg = d1 > d2 ? (REAL)-1 : (REAL)1;
      if (t >= A.nf + B.nf) {
        /* an edge pair counts only if both edges are supporting edges along the axis (the edges lie in planes normal to it, so
         * one endpoint each decides): among parallel edges that singles out the pair that actually touches */
        int ea = (t - A.nf - B.nf) / B.ne, eb = (t - A.nf - B.nf) % B.ne;
        REAL sa = dot3(VA[A.edge[2 * ea]], n), sb = dot3(VB[B.edge[2 * eb]], n);
        REAL supA = sg > 0 ? hiA : loA, supB = sg > 0 ? loB : hiB;
        if (r_abs(sa - supA) > (REAL)1e-5 || r_abs(sb - supB) > (REAL)1e-5) continue;
      }
      if (pass == 0) { if (pen < best) best = pen; }
      else if (pen <= best + CX_TOL(best)) { bt = t; bsgn = sg; memcpy(bn, n, sizeof(bn)); break; }
    }
  }
  REAL N[3] = {bsgn * bn[0], bsgn * bn[1], bsgn * bn[2]}; /* from geom1 to geom2 */
  REAL frame_n[3] = {N[0], N[1], N[2]};
  for (int k = 0; k < 4; k++) { d->con_dist[c0 + k] = 1; for (int i = 0; i < 3; i++) d->con_pos[3 * (c0 + k) + i] = (REAL)0.5 * (pos1[i] + pos2[i]); }
  if (bt >= A.nf + B.nf) { /* --- edge-edge: closest points of the two segments */
    int ea = (bt - A.nf - B.nf) / B.ne, eb = (bt - A.nf - B.nf) % B.ne;
    const REAL *a0 = VA[A.edge[2 * ea]], *a1 = VA[A.edge[2 * ea + 1]], *b0 = VB[B.edge[2 * eb]], *b1 = VB[B.edge[2 * eb + 1]];
    REAL u[3], v[3], r[3];
    for (int k = 0; k < 3; k++) { u[k] = a1[k] - a0[k]; v[k] = b1[k] - b0[k]; r[k] = a0[k] - b0[k]; }
    REAL a = dot3(u, u), e = dot3(v, v), f = dot3(v, r), c = dot3(u, r), b = dot3(u, v), den = a * e - b * b;
    REAL s = den > (REAL)1e-12 * a * e ? r_clip((b * f - c * e) / den, 0, 1) : 0;
    REAL t = (b * s + f) / e;
    if (t < 0) { t = 0; s = r_clip(-c / a, 0, 1); }
    else if (t > 1) { t = 1; s = r_clip((b - c) / a, 0, 1); }
    REAL pa[3], pb[3], dv[3];
    for (int k = 0; k < 3; k++) { pa[k] = a0[k] + s * u[k]; pb[k] = b0[k] + t * v[k]; dv[k] = pb[k] - pa[k]; }
    d->con_dist[c0] = dot3(dv, N);
    for (int k = 0; k < 3; k++) d->con_pos[3 * c0 + k] = (REAL)0.5 * (pa[k] + pb[k]);
  } else { /* --- face contact */
    /* reference face: the face of either hull best aligned with the axis (hull 1 along +N, hull 2 along -N; ties: hull 1) */
    int fa = 0, fb = 0;
    REAL da = (REAL)-1e30, db = (REAL)-1e30, nw[3];
    for (int f = 0; f < A.nf; f++) { mat_mul_vec3(nw, mat1, A.normal + 3 * f); REAL s = dot3(nw, N); if (s > da) { da = s; fa = f; } }
    for (int f = 0; f < B.nf; f++) { mat_mul_vec3(nw, mat2, B.normal + 3 * f); REAL s = -dot3(nw, N); if (s > db) { db = s; fb = f; } }
    int refA = da >= db - (REAL)1e-6;
    const hull_view *R = refA ? &A : &B, *I = refA ? &B : &A;
    const REAL (*VR)[3] = refA ? VA : VB, (*VI)[3] = refA ? VB : VA;
    const REAL *matR = refA ? mat1 : mat2, *matI = refA ? mat2 : mat1;
    int rf = refA ? fa : fb;
    REAL nr[3];
    mat_mul_vec3(nr, matR, R->normal + 3 * rf);
    int fi = 0; REAL dmin = (REAL)1e30; /* incident face: most anti-parallel to the reference normal */
    for (int f = 0; f < I->nf; f++) { mat_mul_vec3(nw, matI, I->normal + 3 * f); REAL s = dot3(nw, nr); if (s < dmin) { dmin = s; fi = f; } }
    REAL poly[2][CX_MAXP][3];
    int np_ = I->fvnum[fi], cur = 0;
    for (int i = 0; i < np_; i++) memcpy(poly[0][i], VI[I->fvert[I->fvadr[fi] + i]], 3 * sizeof(REAL));
    int nr_ = R->fvnum[rf];
    for (int k = 0; k < nr_ && np_ > 0; k++) { /* Sutherland-Hodgman against the side plane through reference edge k */
      const REAL *r0 = VR[R->fvert[R->fvadr[rf] + k]], *r1 = VR[R->fvert[R->fvadr[rf] + (k + 1) % nr_]];
      REAL e[3] = {r1[0] - r0[0], r1[1] - r0[1], r1[2] - r0[2]}, mk[3];
      cross3(mk, e, nr); /* outward of the reference polygon */
      int no = 0;
      for (int i = 0; i < np_; i++) {
        const REAL *pc = poly[cur][i], *pn = poly[cur][(i + 1) % np_];
        REAL dc[3] = {pc[0] - r0[0], pc[1] - r0[1], pc[2] - r0[2]}, dn[3] = {pn[0] - r0[0], pn[1] - r0[1], pn[2] - r0[2]};
        REAL sc = dot3(dc, mk), sn = dot3(dn, mk);
        if (sc <= 0 && no < CX_MAXP) { memcpy(poly[1 - cur][no], pc, 3 * sizeof(REAL)); no++; }
        if ((sc <= 0) != (sn <= 0) && no < CX_MAXP) {
          REAL tt = sc / (sc - sn);
          for (int q = 0; q < 3; q++) poly[1 - cur][no][q] = pc[q] + tt * (pn[q] - pc[q]);
          no++;
        }
      }
      np_ = no; cur = 1 - cur;
    }
    if (np_ > 0) {
      const REAL *r0 = VR[R->fvert[R->fvadr[rf]]];
      REAL depth[CX_MAXP], dmax = (REAL)-1e30, msk[CX_MAXP];
      for (int i = 0; i < np_; i++) {
        REAL dv[3] = {r0[0] - poly[cur][i][0], r0[1] - poly[cur][i][1], r0[2] - poly[cur][i][2]};
        depth[i] = dot3(dv, nr);
        if (depth[i] > dmax) dmax = depth[i];
      }
      /* candidates: the penetrating points; none penetrates: the deepest alone (margin contacts) */
      for (int i = 0; i < np_; i++) msk[i] = (dmax >= 0 ? depth[i] >= 0 : depth[i] == dmax) ? (REAL)0 : (REAL)-1e6;
      const REAL (*P)[3] = (const REAL (*)[3])poly[cur];
      int idx[4] = {0, 0, 0, 0};
      REAL sc[CX_MAXP], sc2[CX_MAXP], bs, bs2;
#define CX_PICK(arr, out) do { bs = (REAL)-1e30; for (int i = 0; i < np_; i++) if (arr[i] > bs) bs = arr[i]; \
                               for (int i = 0; i < np_; i++) if (arr[i] >= bs - CX_TOL(bs)) { out = i; break; } } while (0)
      CX_PICK(msk, idx[0]);
      const REAL *a = P[idx[0]];
      for (int i = 0; i < np_; i++) { REAL q[3] = {a[0] - P[i][0], a[1] - P[i][1], a[2] - P[i][2]}; sc[i] = dot3(q, q) + msk[i]; }
      CX_PICK(sc, idx[1]);
      const REAL *bp = P[idx[1]];
      REAL amb[3] = {a[0] - bp[0], a[1] - bp[1], a[2] - bp[2]}, ab[3];
      cross3(ab, nr, amb);
      for (int i = 0; i < np_; i++) { REAL q[3] = {a[0] - P[i][0], a[1] - P[i][1], a[2] - P[i][2]}; sc[i] = r_abs(dot3(q, ab)) + msk[i]; }
      CX_PICK(sc, idx[2]);
      const REAL *cp = P[idx[2]];
      REAL amc[3] = {a[0] - cp[0], a[1] - cp[1], a[2] - cp[2]}, bmc[3] = {bp[0] - cp[0], bp[1] - cp[1], bp[2] - cp[2]}, ac[3], bc[3];
      cross3(ac, nr, amc); cross3(bc, nr, bmc);
      bs2 = (REAL)-1e30;
      for (int i = 0; i < np_; i++) {
        REAL q[3] = {bp[0] - P[i][0], bp[1] - P[i][1], bp[2] - P[i][2]}, q2[3] = {a[0] - P[i][0], a[1] - P[i][1], a[2] - P[i][2]};
        sc[i] = r_abs(dot3(q, bc)) + msk[i]; sc2[i] = r_abs(dot3(q2, ac)) + msk[i];
      }
      bs = (REAL)-1e30;
      for (int i = 0; i < np_; i++) { if (sc[i] > bs) bs = sc[i]; if (sc2[i] > bs2) bs2 = sc2[i]; }
      if (bs2 > bs + CX_TOL(bs)) CX_PICK(sc2, idx[3]); else CX_PICK(sc, idx[3]); /* farthest from edge bc, or - clearly farther - from edge ac */
      cx_dbg[0] = (REAL)np_;
      for (int k = 0; k < 4; k++) cx_dbg[1 + k] = (REAL)idx[k];
      for (int i = 0; i < np_; i++) cx_dbg[5 + i] = depth[i];
      for (int k = 0; k < 4; k++) {
        int unique = 1;
        for (int j = 0; j < k; j++) if (idx[j] == idx[k]) unique = 0;
        if (!unique) continue;
        d->con_dist[c0 + k] = -depth[idx[k]];
        for (int q = 0; q < 3; q++) d->con_pos[3 * (c0 + k) + q] = P[idx[k]][q] + (REAL)0.5 * depth[idx[k]] * nr[q];
      }
    }
    for (int q = 0; q < 3; q++) frame_n[q] = refA ? nr[q] : -nr[q];
  }
  REAL frame[9];
  make_frame(frame, frame_n);
  for (int k = 0; k < 4; k++) memcpy(d->con_frame + 9 * (c0 + k), frame, sizeof(frame));
}

/* ------------------------------------------------------------------ sphere-convex [mjx:collision_convex.sphere_convex]
 * In the hull's frame: support of every face = signed height of the sphere's lowest point over the face plane; the face
 * with the largest support (the one the sphere penetrates least, or the most separating one when it is clear of the hull);
 * the closest point of that face's polygon to the sphere centre (its projection, or the closest point of the nearest
 * violated edge); distance and normal from there.
 * DEVIATION, recorded in DESIGN.md 4c: the published routine, as recalled (mujoco-mjx is not in this image), first replaces
 * every support >= 0 by -1e12, which for a sphere clear of the hull selects a far face and reports a deep penetration. The
 * two selections are the same whenever all supports are negative, i.e. whenever the sphere reaches below every face plane. */
static void sphere_convex(const amxo_model *m, amxo_data *d, int p) {
  int g1 = m->pair_geom1[p], g2 = m->pair_geom2[p], c0 = m->pair_conadr[p];
  const REAL *pos1 = d->geom_xpos + 3 * g1, *pos2 = d->geom_xpos + 3 * g2, *mat2 = d->geom_xmat + 9 * g2;
  REAL r = m->geom_size[3 * g1];
  hull_view H;
  hull_get(m, m->pair_hull2[p], &H);
  REAL rel[3] = {pos1[0] - pos2[0], pos1[1] - pos2[1], pos1[2] - pos2[2]}, c[3];
  matT_mul_vec3(c, mat2, rel);
  int bf = 0; REAL bs = (REAL)-1e30;
  for (int f = 0; f < H.nf; f++) {
    const REAL *n = H.normal + 3 * f, *v0 = H.vert + 3 * H.fvert[H.fvadr[f]];
    REAL q[3] = {c[0] - n[0] * r - v0[0], c[1] - n[1] * r - v0[1], c[2] - n[2] * r - v0[2]};
    REAL sup = dot3(q, n);
    if (sup > bs) { bs = sup; bf = f; }
  }
  const REAL *n = H.normal + 3 * bf;
  int nvf = H.fvnum[bf];
  const int *fv = H.fvert + H.fvadr[bf];
  const REAL *v0 = H.vert + 3 * fv[0];
  REAL cv[3] = {c[0] - v0[0], c[1] - v0[1], c[2] - v0[2]}, h = dot3(cv, n), pt[3] = {c[0] - n[0] * h, c[1] - n[1] * h, c[2] - n[2] * h};
  int inside = 1, be = -1; REAL bd = (REAL)1e12;
  for (int k = 0; k < nvf; k++) { /* edge k: from vertex k-1 to vertex k */
    const REAL *p0 = H.vert + 3 * fv[(k + nvf - 1) % nvf], *p1 = H.vert + 3 * fv[k];
    REAL e[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]}, en[3], w[3] = {pt[0] - p0[0], pt[1] - p0[1], pt[2] - p0[2]};
    cross3(en, e, n);
    REAL ed = dot3(w, en);
    if (ed > 0) inside = 0;
    int degenerate = en[0] == 0 && en[1] == 0 && en[2] == 0;
    if (!degenerate && !(ed < 0) && ed < bd) { bd = ed; be = k; }
  }
  if (!inside && be >= 0) { /* closest point of the nearest violated edge */
    const REAL *p0 = H.vert + 3 * fv[(be + nvf - 1) % nvf], *p1 = H.vert + 3 * fv[be];
    REAL e[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]}, w[3] = {pt[0] - p0[0], pt[1] - p0[1], pt[2] - p0[2]};
    REAL t = r_clip(dot3(w, e) / r_max(dot3(e, e), MINVAL), 0, 1);
    for (int i = 0; i < 3; i++) pt[i] = p0[i] + t * e[i];
  }
  REAL pn[3] = {pt[0] - c[0], pt[1] - c[1], pt[2] - c[2]};
  REAL dn = normalize_n(pn, 3);
  if (dn == 0) { pn[0] = -n[0]; pn[1] = -n[1]; pn[2] = -n[2]; }
  int below = dot3(pn, n) > 0; /* the centre is under the face: deep penetration */
  REAL nl[3], sgn = below ? (REAL)-1 : (REAL)1;
  for (int i = 0; i < 3; i++) nl[i] = below ? -pn[i] : pn[i];
  REAL dist = dn * sgn - r, posl[3], nw[3], pw[3];
  for (int i = 0; i < 3; i++) posl[i] = (REAL)0.5 * (pt[i] + c[i] + nl[i] * r);
  mat_mul_vec3(nw, mat2, nl); mat_mul_vec3(pw, mat2, posl);
  d->con_dist[c0] = dist;
  for (int i = 0; i < 3; i++) d->con_pos[3 * c0 + i] = pw[i] + pos2[i];
  make_frame(d->con_frame + 9 * c0, nw);
}

/* closest points of the segments [p1, q1] and [p2, q2] (clamped; Ericson, Real-Time Collision Detection 5.1.9): s, t in [0, 1] */
static void seg_seg_closest(const REAL *p1, const REAL *q1, const REAL *p2, const REAL *q2, REAL *c1, REAL *c2) {
  REAL d1[3], d2[3], r[3];
  for (int i = 0; i < 3; i++) { d1[i] = q1[i] - p1[i]; d2[i] = q2[i] - p2[i]; r[i] = p1[i] - p2[i]; }
  REAL a = dot3(d1, d1), e = dot3(d2, d2), f = dot3(d2, r), s, t;
  if (a <= MINVAL && e <= MINVAL) { s = 0; t = 0; }
  else if (a <= MINVAL) { s = 0; t = r_clip(f / e, 0, 1); }
  else {
    REAL c = dot3(d1, r);
    if (e <= MINVAL) { t = 0; s = r_clip(-c / a, 0, 1); }
    else {
      REAL b = dot3(d1, d2), denom = a * e - b * b;
      s = denom > MINVAL ? r_clip((b * f - c * e) / denom, 0, 1) : 0;
      t = (b * s + f) / e;
      if (t < 0) { t = 0; s = r_clip(-c / a, 0, 1); }
      else if (t > 1) { t = 1; s = r_clip((b - c) / a, 0, 1); }
    }
  }
  for (int i = 0; i < 3; i++) { c1[i] = p1[i] + d1[i] * s; c2[i] = p2[i] + d2[i] * t; }
}

/* ------------------------------------------------------------------ A9.5 capsule - convex hull (pair function 11, r04)
 * The published routine (mjx collision_convex.capsule_convex) is restated from its description, like sphere_convex: in the hull's
 * frame, the face of largest support of the capsule's axis (the height of its LOWER end over the face plane, minus the radius);
 * the axis clipped to that face's side planes gives two contacts (the clipped end points: depth = height - radius, normal = minus the
 * face normal); an axis that misses the polygon altogether touches the hull at the polygon edge closest to it: one contact
 * (closest points of the two segments), the second slot inactive.  Pinned by closed forms (tests/test_convex.py: flat on a box,
 * overhanging, end-on, against an edge; a capsule at rest on a box carries its weight). */
static void capsule_convex(const amxo_model *m, amxo_data *d, int p) {
  int g1 = m->pair_geom1[p], g2 = m->pair_geom2[p], c0 = m->pair_conadr[p];
  const REAL *pos1 = d->geom_xpos + 3 * g1, *mat1 = d->geom_xmat + 9 * g1, *pos2 = d->geom_xpos + 3 * g2, *mat2 = d->geom_xmat + 9 * g2;
  REAL r = m->geom_size[3 * g1], half = m->geom_size[3 * g1 + 1];
  hull_view H;
  hull_get(m, m->pair_hull2[p], &H);
  REAL rel[3] = {pos1[0] - pos2[0], pos1[1] - pos2[1], pos1[2] - pos2[2]}, axw[3] = {mat1[2], mat1[5], mat1[8]}, c[3], ax[3], a[3], b[3];
  matT_mul_vec3(c, mat2, rel); matT_mul_vec3(ax, mat2, axw);
  for (int i = 0; i < 3; i++) { a[i] = c[i] + ax[i] * half; b[i] = c[i] - ax[i] * half; }
  int bf = 0; REAL bs = (REAL)-1e30;
  for (int f = 0; f < H.nf; f++) {
    const REAL *n = H.normal + 3 * f, *v0 = H.vert + 3 * H.fvert[H.fvadr[f]];
    REAL qa[3] = {a[0] - v0[0], a[1] - v0[1], a[2] - v0[2]}, qb[3] = {b[0] - v0[0], b[1] - v0[1], b[2] - v0[2]};
    REAL sup = r_min(dot3(qa, n), dot3(qb, n)) - r;
    if (sup > bs) { bs = sup; bf = f; }
  }
  const REAL *n = H.normal + 3 * bf;
  int nvf = H.fvnum[bf];
  const int *fv = H.fvert + H.fvadr[bf];
  const REAL *v0 = H.vert + 3 * fv[0];
  REAL t0 = 0, t1 = 1; int empty = 0;
  for (int k = 0; k < nvf; k++) { /* side plane of edge k (vertex k-1 -> k): outside where (x - p0) . (e x n) > 0 */
    const REAL *p0 = H.vert + 3 * fv[(k + nvf - 1) % nvf], *p1 = H.vert + 3 * fv[k];
    REAL e[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]}, en[3], wa[3] = {a[0] - p0[0], a[1] - p0[1], a[2] - p0[2]}, wb[3] = {b[0] - p0[0], b[1] - p0[1], b[2] - p0[2]};
    cross3(en, e, n);
    REAL da = dot3(wa, en), db = dot3(wb, en);
    if (da > 0 && db > 0) { empty = 1; continue; }
    if (da > 0) t0 = r_max(t0, da / (da - db));
    else if (db > 0) t1 = r_min(t1, da / (da - db));
  }
  if (t0 > t1) empty = 1;
  REAL nl[3], posl[2][3], dist[2];
  if (!empty) {
    REAL tt[2] = {t0, t1};
    for (int k = 0; k < 2; k++) {
      REAL pt[3], w[3];
      for (int i = 0; i < 3; i++) { pt[i] = a[i] + (b[i] - a[i]) * tt[k]; w[i] = pt[i] - v0[i]; }
      REAL h = dot3(w, n);
      dist[k] = h - r;
      for (int i = 0; i < 3; i++) posl[k][i] = pt[i] - n[i] * (r + (REAL)0.5 * dist[k]);
    }
    for (int i = 0; i < 3; i++) nl[i] = -n[i];
  } else {
    REAL best = (REAL)1e30, pa[3] = {c[0], c[1], c[2]}, pe[3] = {v0[0], v0[1], v0[2]};
    for (int k = 0; k < nvf; k++) {
      const REAL *p0 = H.vert + 3 * fv[(k + nvf - 1) % nvf], *p1 = H.vert + 3 * fv[k];
      REAL c1[3], c2[3];
      seg_seg_closest(a, b, p0, p1, c1, c2);
      REAL dd[3] = {c2[0] - c1[0], c2[1] - c1[1], c2[2] - c1[2]}, d2 = dot3(dd, dd);
      if (d2 < best) { best = d2; memcpy(pa, c1, sizeof(pa)); memcpy(pe, c2, sizeof(pe)); }
    }
    for (int i = 0; i < 3; i++) nl[i] = pe[i] - pa[i];
    REAL dn = normalize_n(nl, 3);
    if (dn == 0) { nl[0] = -n[0]; nl[1] = -n[1]; nl[2] = -n[2]; }
    dist[0] = dn - r; dist[1] = 1;
    for (int i = 0; i < 3; i++) { posl[0][i] = (REAL)0.5 * (pa[i] + nl[i] * r + pe[i]); posl[1][i] = posl[0][i]; }
  }
  REAL nw[3];
  mat_mul_vec3(nw, mat2, nl);
  for (int k = 0; k < 2; k++) {
    REAL pw[3];
    mat_mul_vec3(pw, mat2, posl[k]);
    d->con_dist[c0 + k] = dist[k];
    for (int i = 0; i < 3; i++) d->con_pos[3 * (c0 + k) + i] = pw[i] + pos2[i];
    make_frame(d->con_frame + 9 * (c0 + k), nw);
  }
}

static void collision(const amxo_model *m, amxo_data *d) {
  for (int p = 0; p < m->npair; p++) {
    int g1 = m->pair_geom1[p], g2 = m->pair_geom2[p], c0 = m->pair_conadr[p], fn = m->pair_fn[p];
    const REAL *pos1 = d->geom_xpos + 3 * g1, *mat1 = d->geom_xmat + 9 * g1;
    const REAL *pos2 = d->geom_xpos + 3 * g2, *mat2 = d->geom_xmat + 9 * g2;
    const REAL *size1 = m->geom_size + 3 * g1, *size2 = m->geom_size + 3 * g2;
    for (int k = 0; k < PAIR_NCON[fn]; k++) d->con_pair[c0 + k] = p;
    if (fn == 6) { convex_convex(m, d, p); continue; }
    if (fn == 7) { sphere_convex(m, d, p); continue; }
    if (fn == 11) { capsule_convex(m, d, p); continue; }
    if (fn == 0) { /* plane-sphere */
      REAL n[3] = {mat1[2], mat1[5], mat1[8]};
      plane_sphere_core(n, pos1, pos2, size2[0], d->con_dist + c0, d->con_pos + 3 * c0);
      make_frame(d->con_frame + 9 * c0, n);
    } else if (fn == 1) { /* plane-capsule: two sphere contacts, frame aligned with the capsule axis */
      REAL n[3] = {mat1[2], mat1[5], mat1[8]}, axis[3] = {mat2[2], mat2[5], mat2[8]};
      REAL na = dot3(n, axis), b[3];
      for (int i = 0; i < 3; i++) b[i] = axis[i] - n[i] * na;
      REAL bn = normalize_n(b, 3);
      if (bn < (REAL)0.5) {
        b[0] = 0; b[1] = 0; b[2] = 0;
        if (n[1] > (REAL)-0.5 && n[1] < (REAL)0.5) b[1] = 1; else b[2] = 1;
      }
      REAL frame[9];
      memcpy(frame, n, sizeof(n)); memcpy(frame + 3, b, sizeof(b)); cross3(frame + 6, n, b);
      for (int k = 0; k < 2; k++) {
        REAL sgn = k == 0 ? (REAL)1 : (REAL)-1, sp[3];
        for (int i = 0; i < 3; i++) sp[i] = pos2[i] + sgn * axis[i] * size2[1];
        plane_sphere_core(n, pos1, sp, size2[0], d->con_dist + c0 + k, d->con_pos + 3 * (c0 + k));
        memcpy(d->con_frame + 9 * (c0 + k), frame, sizeof(frame));
      }
    } else if (fn == 2) { /* plane-convex: 4-point manifold among the deepest vertices */
      int nvt = m->pair_vertnum[p];
      const REAL *vert = m->convex_vert + 3 * m->pair_vertadr[p];
      REAL nw[3] = {mat1[2], mat1[5], mat1[8]}, rel[3] = {pos1[0] - pos2[0], pos1[1] - pos2[1], pos1[2] - pos2[2]};
      REAL pl[3], n[3];
      matT_mul_vec3(pl, mat2, rel); matT_mul_vec3(n, mat2, nw);
      REAL *sup = d->w_support, smax = (REAL)-1e30;
      for (int v = 0; v < nvt; v++) {
        REAL dv[3] = {pl[0] - vert[3 * v], pl[1] - vert[3 * v + 1], pl[2] - vert[3 * v + 2]};
        sup[v] = dot3(dv, n);
        if (sup[v] > smax) smax = sup[v];
      }
      REAL thr = r_max((REAL)0, smax - (REAL)1e-3);
      /* _manifold_points: a = first masked, b farthest from a, c farthest from line ab, d farthest from edges ac, bc */
      int idx[4];
      REAL best; int bi;
#define MASK(v) (sup[v] > thr ? (REAL)0 : (REAL)-1e6)
      bi = 0; best = MASK(0);
      for (int v = 1; v < nvt; v++) if (MASK(v) > best) { best = MASK(v); bi = v; }
      idx[0] = bi;
      const REAL *a = vert + 3 * idx[0];
      bi = 0; best = (REAL)-1e30;
      for (int v = 0; v < nvt; v++) {
        REAL dx = a[0] - vert[3 * v], dy = a[1] - vert[3 * v + 1], dz = a[2] - vert[3 * v + 2];
        REAL s = dx * dx + dy * dy + dz * dz + MASK(v);
        if (s > best) { best = s; bi = v; }
      }
      idx[1] = bi;
      const REAL *bp = vert + 3 * idx[1];
      REAL amb[3] = {a[0] - bp[0], a[1] - bp[1], a[2] - bp[2]}, ab[3];
      cross3(ab, n, amb);
      bi = 0; best = (REAL)-1e30;
      for (int v = 0; v < nvt; v++) {
        REAL ap[3] = {a[0] - vert[3 * v], a[1] - vert[3 * v + 1], a[2] - vert[3 * v + 2]};
        REAL s = r_abs(dot3(ap, ab)) + MASK(v);
        if (s > best) { best = s; bi = v; }
      }
      idx[2] = bi;
      const REAL *cp = vert + 3 * idx[2];
      REAL amc[3] = {a[0] - cp[0], a[1] - cp[1], a[2] - cp[2]}, bmc[3] = {bp[0] - cp[0], bp[1] - cp[1], bp[2] - cp[2]}, ac[3], bc[3];
      cross3(ac, n, amc); cross3(bc, n, bmc);
      bi = 0; best = (REAL)-1e30; /* argmax over concat(dist_bp, dist_ap) % nvt: first half has priority on ties */
      for (int v = 0; v < nvt; v++) {
        REAL q[3] = {bp[0] - vert[3 * v], bp[1] - vert[3 * v + 1], bp[2] - vert[3 * v + 2]};
        REAL s = r_abs(dot3(q, bc)) + MASK(v);
        if (s > best) { best = s; bi = v; }
      }
      for (int v = 0; v < nvt; v++) {
        REAL q[3] = {a[0] - vert[3 * v], a[1] - vert[3 * v + 1], a[2] - vert[3 * v + 2]};
        REAL s = r_abs(dot3(q, ac)) + MASK(v);
        if (s > best) { best = s; bi = v; }
      }
      idx[3] = bi;
#undef MASK
      REAL frame[9]; make_frame(frame, nw);
      for (int k = 0; k < 4; k++) {
        int unique = 1;
        for (int j = 0; j < k; j++) if (idx[j] == idx[k]) unique = 0;
        REAL dist = unique ? -sup[idx[k]] : (REAL)1, pw[3];
        mat_mul_vec3(pw, mat2, vert + 3 * idx[k]);
        for (int i = 0; i < 3; i++) d->con_pos[3 * (c0 + k) + i] = pos2[i] + pw[i] - (REAL)0.5 * dist * nw[i];
        d->con_dist[c0 + k] = dist;
        memcpy(d->con_frame + 9 * (c0 + k), frame, sizeof(frame));
      }
    } else if (fn == 3 || fn == 4 || fn == 5) { /* sphere-sphere, sphere-capsule, capsule-capsule */
      REAL p1[3] = {pos1[0], pos1[1], pos1[2]}, p2[3] = {pos2[0], pos2[1], pos2[2]}, r1 = size1[0], r2 = size2[0];
      if (fn == 4) { /* closest point on capsule segment to the sphere centre */
        REAL ax[3] = {mat2[2], mat2[5], mat2[8]}, dv[3] = {pos1[0] - pos2[0], pos1[1] - pos2[1], pos1[2] - pos2[2]};
        REAL t = r_clip(dot3(dv, ax), -size2[1], size2[1]);
        for (int i = 0; i < 3; i++) p2[i] = pos2[i] + ax[i] * t;
      } else if (fn == 5) { /* closest points between two segments */
        REAL a1[3] = {mat1[2], mat1[5], mat1[8]}, a2[3] = {mat2[2], mat2[5], mat2[8]}, h1 = size1[1], h2 = size2[1];
        REAL dv[3] = {pos1[0] - pos2[0], pos1[1] - pos2[1], pos1[2] - pos2[2]};
        REAL a12 = dot3(a1, a2), d1 = dot3(dv, a1), d2 = dot3(dv, a2), det = 1 - a12 * a12, t1, t2;
        if (r_abs(det) < (REAL)1e-6) { t1 = 0; }
        else t1 = r_clip((a12 * d2 - d1) / det, -h1, h1);
        t2 = r_clip(d2 + t1 * a12, -h2, h2);
        t1 = r_clip(t2 * a12 - d1, -h1, h1);
        for (int i = 0; i < 3; i++) { p1[i] = pos1[i] + a1[i] * t1; p2[i] = pos2[i] + a2[i] * t2; }
      }
      REAL n[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
      REAL len = normalize_n(n, 3);
      if (len == 0) { n[0] = 1; n[1] = 0; n[2] = 0; }
      d->con_dist[c0] = len - (r1 + r2);
      for (int i = 0; i < 3; i++) d->con_pos[3 * c0 + i] = p1[i] + n[i] * (r1 + (REAL)0.5 * d->con_dist[c0]);
      make_frame(d->con_frame + 9 * c0, n);
    }
  }
}

/* ------------------------------------------------------------------ A9.6 make_constraint [mjx:constraint.make_constraint] */
static void kbi(const amxo_model *m, const REAL *solref, const REAL *solimp, REAL pos, REAL *k, REAL *b, REAL *imp) {
  REAL timeconst = solref[0], dampratio = solref[1];
  if (!(m->disableflags & DIS_REFSAFE)) timeconst = r_max(timeconst, 2 * m->timestep);
  REAL dmin = r_clip(solimp[0], MINIMP, MAXIMP), dmax = r_clip(solimp[1], MINIMP, MAXIMP);
  REAL width = r_max(MINVAL, solimp[2]), mid = r_clip(solimp[3], MINIMP, MAXIMP), power = r_max((REAL)1, solimp[4]);
  *k = 1 / (dmax * dmax * timeconst * timeconst * dampratio * dampratio);
  *b = 2 / (dmax * timeconst);
  if (solref[0] <= 0) *k = -solref[0] / (dmax * dmax);
  if (solref[1] <= 0) *b = -solref[1] / dmax;
  REAL x = r_abs(pos) / width;
  REAL ia = (1 / r_pow(mid, power - 1)) * r_pow(x, power);
  REAL ib = 1 - (1 / r_pow(1 - mid, power - 1)) * r_pow(1 - x, power);
  REAL y = x < mid ? ia : ib;
  REAL v = dmin + y * (dmax - dmin);
  v = r_clip(v, dmin, dmax);
  if (x > 1) v = dmax;
  *imp = v;
}
static void finish_row(const amxo_model *m, amxo_data *d, int row, REAL pos, REAL invweight, const REAL *solref, const REAL *solimp) {
  REAL k, b, imp, jv = 0;
  kbi(m, solref, solimp, pos, &k, &b, &imp);
  for (int i = 0; i < m->nv; i++) jv += d->efc_J[row * m->nv + i] * d->qvel[i];
  REAL r = r_max(invweight * (1 - imp) / imp, MINVAL);
  d->efc_D[row] = 1 / r;
  d->efc_aref[row] = -b * jv - k * imp * pos;
  d->efc_pos[row] = pos;
  d->efc_floss[row] = 0;
}
/* a row whose impedance is evaluated at `pos_imp` (the norm of a multi-row residual) instead of its own position */
static void finish_row_imp(const amxo_model *m, amxo_data *d, int row, REAL pos, REAL pos_imp, REAL invweight, const REAL *solref, const REAL *solimp) {
  REAL k, b, imp, jv = 0;
  kbi(m, solref, solimp, pos_imp, &k, &b, &imp);
  for (int i = 0; i < m->nv; i++) jv += d->efc_J[row * m->nv + i] * d->qvel[i];
  REAL r = r_max(invweight * (1 - imp) / imp, MINVAL);
  d->efc_D[row] = 1 / r;
  d->efc_aref[row] = -b * jv - k * imp * pos;
  d->efc_pos[row] = pos;
  d->efc_floss[row] = 0;
}
/* rotational Jacobian of body b (angular parts of the motion axes along its chain) into jac[3*nv] [mjx:support.jac] */
static void jac_rot(const amxo_model *m, const amxo_data *d, REAL *jac, int b, REAL sign) {
  while (b > 0 && m->body_dofnum[b] == 0) b = m->body_parentid[b];
  if (b == 0) return;
  for (int i = m->body_dofadr[b] + m->body_dofnum[b] - 1; i >= 0; i = m->dof_parentid[i])
    for (int k = 0; k < 3; k++) jac[k * m->nv + i] += sign * d->cdof[6 * i + k];
}
/* translational Jacobian of a world point fixed to body b, into jac[3*nv] (row-major 3 x nv) [mjx:support.jac] */
static void jac_point(const amxo_model *m, const amxo_data *d, REAL *jac, const REAL *point, int b, REAL sign) {
  if (b == 0) return;
  const REAL *com = d->subtree_com + 3 * m->body_rootid[b];
  REAL off[3] = {point[0] - com[0], point[1] - com[1], point[2] - com[2]};
  while (b > 0 && m->body_dofnum[b] == 0) b = m->body_parentid[b];
  if (b == 0) return;
  for (int i = m->body_dofadr[b] + m->body_dofnum[b] - 1; i >= 0; i = m->dof_parentid[i]) {
    const REAL *c = d->cdof + 6 * i;
    REAL t[3]; cross3(t, c, off);
    for (int k = 0; k < 3; k++) jac[k * m->nv + i] += sign * (c[3 + k] + t[k]);
  }
}
static void make_constraint(const amxo_model *m, amxo_data *d) {
  int nv = m->nv, row = 0;
  d->ne = 0;
  if (m->disableflags & DIS_CONSTRAINT) { d->nefc = 0; return; }
  if (!(m->disableflags & DIS_EQUALITY)) {
    /* connect and weld [mjx:constraint._instantiate_equality_connect / _weld; MuJoCo mj_instantiateEquality]: the difference of two
     * body-fixed points (and, for a weld, of the orientations) held at zero.  All rows of one equality share the impedance of the
     * norm of its residual; translational rows are regularised by the bodies' translational invweight0, rotational ones by the
     * rotational. */
    for (int e = 0; e < m->neq; e++) {
      if (!m->eq_active[e] || m->eq_type[e] > 1) continue;
      const int weld = m->eq_type[e] == 1, b1 = m->eq_obj1id[e], b2 = m->eq_obj2id[e], nr = weld ? 6 : 3;
      const REAL *dat = m->eq_data + 11 * e;
      /* connect: anchor in body1 = dat[0:3], in body2 = dat[3:6]; weld: weld point in body2 = dat[0:3], in body1 = dat[3:6] */
      const REAL *a1 = weld ? dat + 3 : dat, *a2 = weld ? dat : dat + 3;
      REAL p1[3], p2[3], pos[6], t[3];
      mat_mul_vec3(t, d->xmat + 9 * b1, a1); for (int k = 0; k < 3; k++) p1[k] = d->xpos[3 * b1 + k] + t[k];
      mat_mul_vec3(t, d->xmat + 9 * b2, a2); for (int k = 0; k < 3; k++) p2[k] = d->xpos[3 * b2 + k] + t[k];
      for (int k = 0; k < 3; k++) pos[k] = p1[k] - p2[k];
      REAL *J0 = d->efc_J + row * nv;
      memset(J0, 0, sizeof(REAL) * nr * nv);
      jac_point(m, d, J0, p1, b1, (REAL)1);
      jac_point(m, d, J0, p2, b2, (REAL)-1);
      if (weld) {
        /* orientation residual: vector part of conj(q2) * q1 * relpose, times torquescale; its Jacobian is
         * 0.5 * vec(conj(q2) * (0, jacr1 - jacr2) * q1 * relpose), times torquescale */
        const REAL ts = dat[10];
        REAL q1r[4], q2c[4], qe[4];
        quat_mul(q1r, d->xquat + 4 * b1, dat + 6);
        q2c[0] = d->xquat[4 * b2]; q2c[1] = -d->xquat[4 * b2 + 1]; q2c[2] = -d->xquat[4 * b2 + 2]; q2c[3] = -d->xquat[4 * b2 + 3];
        quat_mul(qe, q2c, q1r);
        for (int k = 0; k < 3; k++) pos[3 + k] = qe[1 + k] * ts;
        REAL *Jr = J0 + 3 * nv;
        jac_rot(m, d, Jr, b1, (REAL)1);
        jac_rot(m, d, Jr, b2, (REAL)-1);
        for (int i = 0; i < nv; i++) {
          REAL w[4] = {0, Jr[i], Jr[nv + i], Jr[2 * nv + i]}, u[4], v[4];
          if (w[1] == 0 && w[2] == 0 && w[3] == 0) continue;
          quat_mul(u, q2c, w); quat_mul(v, u, q1r);
          for (int k = 0; k < 3; k++) Jr[k * nv + i] = (REAL)0.5 * v[1 + k] * ts;
        }
      }
      REAL nrm = 0;
      for (int k = 0; k < nr; k++) nrm += pos[k] * pos[k];
      nrm = r_sqrt(nrm);
      for (int k = 0; k < nr; k++) {
        const REAL invw = m->body_invweight0[2 * b1 + (k >= 3)] + m->body_invweight0[2 * b2 + (k >= 3)];
        finish_row_imp(m, d, row, pos[k], nrm, invw, m->eq_solref + 2 * e, m->eq_solimp + 5 * e);
        row++;
      }
    }
    for (int e = 0; e < m->neq; e++) { /* joint polycoef equality */
      if (!m->eq_active[e] || m->eq_type[e] != 2) continue;
      int j1 = m->eq_obj1id[e], j2 = m->eq_obj2id[e];
      const REAL *c = m->eq_data + 11 * e;
      REAL *J = d->efc_J + row * nv;
      memset(J, 0, sizeof(REAL) * nv);
      REAL pos1 = d->qpos[m->jnt_qposadr[j1]] - m->qpos0[m->jnt_qposadr[j1]], pos, invw = m->dof_invweight0[m->jnt_dofadr[j1]];
      if (j2 >= 0) {
        REAL dif = d->qpos[m->jnt_qposadr[j2]] - m->qpos0[m->jnt_qposadr[j2]];
        REAL dp[5] = {1, dif, dif * dif, dif * dif * dif, dif * dif * dif * dif};
        REAL deriv = c[1] * dp[0] * 1 + c[2] * dp[1] * 2 + c[3] * dp[2] * 3 + c[4] * dp[3] * 4;
        pos = pos1 - (c[0] * dp[0] + c[1] * dp[1] + c[2] * dp[2] + c[3] * dp[3] + c[4] * dp[4]);
        J[m->jnt_dofadr[j2]] = -deriv;
        invw += m->dof_invweight0[m->jnt_dofadr[j2]];
      } else {
        pos = pos1 - c[0];
      }
      J[m->jnt_dofadr[j1]] = 1;
      finish_row(m, d, row, pos, invw, m->eq_solref + 2 * e, m->eq_solimp + 5 * e);
      row++;
    }
  }
  d->ne = row;
  /* friction loss [mjx:constraint._instantiate_friction; MuJoCo engine_core_constraint.c mj_instantiateFriction]: one row per
   * dof with frictionloss > 0: J = e_i, pos = 0 (aref = -b * qvel_i), D from the dof's invweight0 and solreffriction /
   * solimpfriction; the force is bounded by +-frictionloss (update_constraint below) */
  if (!(m->disableflags & DIS_FRICTIONLOSS)) {
    for (int i = 0; i < nv; i++) {
      if (!(m->dof_frictionloss[i] > 0)) continue;
      REAL *J = d->efc_J + row * nv;
      memset(J, 0, sizeof(REAL) * nv);
      J[i] = 1;
      finish_row(m, d, row, 0, m->dof_invweight0[i], m->dof_solref + 2 * i, m->dof_solimp + 5 * i);
      d->efc_floss[row] = m->dof_frictionloss[i];
      row++;
    }
  }
  d->nf = row - d->ne;
  if (!(m->disableflags & DIS_LIMIT)) {
    /* ball joints [mjx:constraint._instantiate_limit_ball; MuJoCo engine_core_constraint.c mj_instantiateLimit, mjJNT_BALL]: the
     * joint quaternion as axis * angle (angle wrapped to (-pi, pi], then made positive by flipping the axis), one row
     * J = -axis on the joint's three dofs, pos = max(range) - angle - margin.  math.normalize_with_norm divides by
     * n + 1e-6 (n == 0): a zero rotation gives a zero axis and angle 0 */
    for (int j = 0; j < m->njnt; j++) {
      if (!m->jnt_limited[j] || m->jnt_type[j] != JBALL) continue;
      const REAL *q = d->qpos + m->jnt_qposadr[j];
      REAL s = r_sqrt(q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
      REAL inv = 1 / (s + (s == 0 ? (REAL)1e-6 : 0));
      REAL ax[3] = {q[1] * inv, q[2] * inv, q[3] * inv};
      REAL angle = 2 * r_atan2(s, q[0]);
      if (angle > (REAL)3.14159265358979323846) angle -= 2 * (REAL)3.14159265358979323846;
      REAL av[3] = {ax[0] * angle, ax[1] * angle, ax[2] * angle};
      REAL n = r_sqrt(av[0] * av[0] + av[1] * av[1] + av[2] * av[2]);
      REAL inv2 = 1 / (n + (n == 0 ? (REAL)1e-6 : 0));
      REAL rmax = r_max(m->jnt_range[2 * j], m->jnt_range[2 * j + 1]);
      REAL pos = rmax - n - m->jnt_margin[j];
      if (!(pos < 0)) continue;
      REAL *J = d->efc_J + row * nv;
      memset(J, 0, sizeof(REAL) * nv);
      for (int k = 0; k < 3; k++) J[m->jnt_dofadr[j] + k] = -av[k] * inv2;
      finish_row(m, d, row, pos, m->dof_invweight0[m->jnt_dofadr[j]], m->jnt_solref + 2 * j, m->jnt_solimp + 5 * j);
      row++;
    }
    for (int j = 0; j < m->njnt; j++) {
      if (!m->jnt_limited[j] || (m->jnt_type[j] != JSLIDE && m->jnt_type[j] != JHINGE)) continue;
      REAL q = d->qpos[m->jnt_qposadr[j]];
      REAL dmin = q - m->jnt_range[2 * j], dmax = m->jnt_range[2 * j + 1] - q;
      REAL pos = r_min(dmin, dmax) - m->jnt_margin[j];
      if (!(pos < 0)) continue; /* inactive rows are all-zero in MJX and contribute nothing */
      REAL *J = d->efc_J + row * nv;
      memset(J, 0, sizeof(REAL) * nv);
      J[m->jnt_dofadr[j]] = dmin < dmax ? (REAL)1 : (REAL)-1;
      finish_row(m, d, row, pos, m->dof_invweight0[m->jnt_dofadr[j]], m->jnt_solref + 2 * j, m->jnt_solimp + 5 * j);
      row++;
    }
  }
  if (!(m->disableflags & DIS_CONTACT)) {
    REAL *jac = d->w_H; /* 3 x nv scratch */
    for (int c = 0; c < m->ncon; c++) {
      int p = d->con_pair[c];
      REAL margin = m->pair_margin[p];
      if (!(d->con_dist[c] < margin)) continue;
      int b1 = m->geom_bodyid[m->pair_geom1[p]], b2 = m->geom_bodyid[m->pair_geom2[p]];
      memset(jac, 0, sizeof(REAL) * 3 * nv);
      jac_point(m, d, jac, d->con_pos + 3 * c, b2, (REAL)1);
      jac_point(m, d, jac, d->con_pos + 3 * c, b1, (REAL)-1);
      const REAL *fr = d->con_frame + 9 * c;
      REAL t = m->body_invweight0[2 * b1] + m->body_invweight0[2 * b2];
      REAL mu = m->pair_friction[3 * p]; /* condim 3: both tangent directions use the sliding coefficient */
      REAL pos = d->con_dist[c] - margin;
      for (int k = 0; k < 4; k++) {
        int tan = 1 + k / 2;
        REAL f = (k % 2 == 0) ? mu : -mu;
        REAL *J = d->efc_J + row * nv;
        for (int i = 0; i < nv; i++) {
          REAL jn = fr[0] * jac[i] + fr[1] * jac[nv + i] + fr[2] * jac[2 * nv + i];
          REAL jt = fr[3 * tan] * jac[i] + fr[3 * tan + 1] * jac[nv + i] + fr[3 * tan + 2] * jac[2 * nv + i];
          J[i] = jn + jt * f;
        }
        REAL invw = (t + f * f * t) * 2 * f * f / m->impratio;
        finish_row(m, d, row, pos, invw, m->pair_solref + 2 * p, m->pair_solimp + 5 * p);
        row++;
      }
    }
  }
  d->nefc = row;
}

/* ------------------------------------------------------------------ A9.8 com_vel [mjx:smooth.com_vel] */
static void com_vel(const amxo_model *m, amxo_data *d) {
  memset(d->cvel, 0, sizeof(REAL) * 6);
  for (int b = 1; b < m->nbody; b++) {
    REAL cv[6];
    memcpy(cv, d->cvel + 6 * m->body_parentid[b], sizeof(cv));
    for (int k = 0; k < m->body_jntnum[b]; k++) {
      int j = m->body_jntadr[b] + k, da = m->jnt_dofadr[j], t = m->jnt_type[j];
      if (t == JFREE) {
        for (int i = 0; i < 3; i++) { memset(d->cdof_dot + 6 * (da + i), 0, 6 * sizeof(REAL)); for (int c = 0; c < 6; c++) cv[c] += d->cdof[6 * (da + i) + c] * d->qvel[da + i]; }
        for (int i = 3; i < 6; i++) motion_cross(d->cdof_dot + 6 * (da + i), cv, d->cdof + 6 * (da + i));
        for (int i = 3; i < 6; i++) for (int c = 0; c < 6; c++) cv[c] += d->cdof[6 * (da + i) + c] * d->qvel[da + i];
      } else {
        int n = t == JBALL ? 3 : 1;
        for (int i = 0; i < n; i++) motion_cross(d->cdof_dot + 6 * (da + i), cv, d->cdof + 6 * (da + i));
        for (int i = 0; i < n; i++) for (int c = 0; c < 6; c++) cv[c] += d->cdof[6 * (da + i) + c] * d->qvel[da + i];
      }
    }
    memcpy(d->cvel + 6 * b, cv, sizeof(cv));
  }
}

/* ------------------------------------------------------------------ A9.9 passive [mjx:passive.passive] */
static void passive(const amxo_model *m, amxo_data *d) {
  memset(d->qfrc_passive, 0, sizeof(REAL) * m->nv);
  if (m->disableflags & DIS_PASSIVE) return;
  for (int j = 0; j < m->njnt; j++) {
    if (m->jnt_stiffness[j] == 0) continue;
    if (m->jnt_type[j] == JSLIDE || m->jnt_type[j] == JHINGE)
      d->qfrc_passive[m->jnt_dofadr[j]] -= m->jnt_stiffness[j] * (d->qpos[m->jnt_qposadr[j]] - m->qpos_spring[m->jnt_qposadr[j]]);
  }
  for (int i = 0; i < m->nv; i++) d->qfrc_passive[i] -= m->dof_damping[i] * d->qvel[i];
}

/* ------------------------------------------------------------------ A9.10 rne [mjx:smooth.rne] */
static void rne(const amxo_model *m, amxo_data *d) {
  int nb = m->nbody;
  memset(d->cacc, 0, sizeof(REAL) * 6);
  if (!(m->disableflags & DIS_GRAVITY)) for (int i = 0; i < 3; i++) d->cacc[3 + i] = -m->gravity[i];
  for (int b = 1; b < nb; b++) {
    REAL *ca = d->cacc + 6 * b;
    memcpy(ca, d->cacc + 6 * m->body_parentid[b], 6 * sizeof(REAL));
    for (int k = 0; k < m->body_dofnum[b]; k++) {
      int i = m->body_dofadr[b] + k;
      for (int c = 0; c < 6; c++) ca[c] += d->cdof_dot[6 * i + c] * d->qvel[i];
    }
  }
  for (int b = 0; b < nb; b++) {
    REAL f[6], iv[6], g[6];
    inert_mul(f, d->cinert + 10 * b, d->cacc + 6 * b);
    inert_mul(iv, d->cinert + 10 * b, d->cvel + 6 * b);
    motion_cross_force(g, d->cvel + 6 * b, iv);
    for (int c = 0; c < 6; c++) d->cfrc[6 * b + c] = f[c] + g[c];
  }
  for (int b = nb - 1; b > 0; b--) {
    int p = m->body_parentid[b];
    for (int c = 0; c < 6; c++) d->cfrc[6 * p + c] += d->cfrc[6 * b + c];
  }
  for (int i = 0; i < m->nv; i++) {
    REAL s = 0;
    for (int c = 0; c < 6; c++) s += d->cdof[6 * i + c] * d->cfrc[6 * m->dof_bodyid[i] + c];
    d->qfrc_bias[i] = s;
  }
}

/* ------------------------------------------------------------------ A9.7 + A9.11 transmission / fwd_actuation [mjx:forward.fwd_actuation] */
static void fwd_actuation(const amxo_model *m, amxo_data *d) {
  memset(d->qfrc_actuator, 0, sizeof(REAL) * m->nv);
  if (m->nu == 0 || (m->disableflags & DIS_ACTUATION)) return;
  for (int u = 0; u < m->nu; u++) {
    int j = m->actuator_trnid[u], qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
    REAL gear = m->actuator_gear[u], ctrl = d->ctrl[u];
    if (!(m->disableflags & DIS_CLAMPCTRL) && m->actuator_ctrllimited[u])
      ctrl = r_clip(ctrl, m->actuator_ctrlrange[2 * u], m->actuator_ctrlrange[2 * u + 1]);
    REAL length = d->qpos[qa] * gear, velocity = d->qvel[da] * gear;
    REAL force = m->actuator_gainprm[3 * u] * ctrl;
    if (m->actuator_biastype[u] == 1)
      force += m->actuator_biasprm[3 * u] + m->actuator_biasprm[3 * u + 1] * length + m->actuator_biasprm[3 * u + 2] * velocity;
    if (m->actuator_forcelimited[u]) force = r_clip(force, m->actuator_forcerange[2 * u], m->actuator_forcerange[2 * u + 1]);
    d->qfrc_actuator[da] += gear * force;
  }
}

/* ------------------------------------------------------------------ A9.12 fwd_acceleration [mjx:forward.fwd_acceleration] */
static void fwd_acceleration(const amxo_model *m, amxo_data *d) {
  for (int i = 0; i < m->nv; i++) d->qfrc_smooth[i] = d->qfrc_passive[i] - d->qfrc_bias[i] + d->qfrc_actuator[i];
  chol_solve(d->qL, d->qacc_smooth, d->qfrc_smooth, m->nv);
}

/* ------------------------------------------------------------------ A9.13 solver [mjx:solver.solve] */
typedef struct { REAL *qacc, *Ma, *grad, *Mgrad, *search, *Jaref; REAL gauss, cost, prev_cost; } sctx;
typedef struct { REAL alpha, cost, deriv0, deriv1, q1, q2; } lspoint;

static void update_constraint(const amxo_model *m, amxo_data *d, sctx *c) {
  int nv = m->nv, ne = d->nefc;
  REAL cost = 0;
  memset(d->qfrc_constraint, 0, sizeof(REAL) * nv);
  for (int r = 0; r < ne; r++) {
    int active = (c->Jaref[r] < 0) || r < d->ne;
    if (d->efc_floss[r] > 0) {
      /* friction row: quadratic while |D Jaref| < floss, linear beyond (the force saturates at +-floss)
       * [MuJoCo mj_constraintUpdate: LINEARNEG / LINEARPOS / QUADRATIC] */
      REAL f = d->efc_floss[r], rf = f / d->efc_D[r], z = c->Jaref[r];
      if (z <= -rf) { d->efc_force[r] = f; cost += f * (-(REAL)0.5 * rf - z); }
      else if (z >= rf) { d->efc_force[r] = -f; cost += f * (-(REAL)0.5 * rf + z); }
      else { d->efc_force[r] = -d->efc_D[r] * z; cost += (REAL)0.5 * d->efc_D[r] * z * z; }
    } else {
      d->efc_force[r] = active ? -d->efc_D[r] * c->Jaref[r] : 0;
      if (active) cost += (REAL)0.5 * d->efc_D[r] * c->Jaref[r] * c->Jaref[r];
    }
    if (d->efc_force[r] != 0)
      for (int i = 0; i < nv; i++) d->qfrc_constraint[i] += d->efc_J[r * nv + i] * d->efc_force[r];
  }
  REAL g = 0;
  for (int i = 0; i < nv; i++) g += (c->Ma[i] - d->qfrc_smooth[i]) * (c->qacc[i] - d->qacc_smooth[i]);
  c->gauss = (REAL)0.5 * g;
  c->prev_cost = c->cost;
  c->cost = cost + c->gauss;
}
static void update_gradient(const amxo_model *m, amxo_data *d, sctx *c) {
  int nv = m->nv, ne = d->nefc;
  for (int i = 0; i < nv; i++) c->grad[i] = c->Ma[i] - d->qfrc_smooth[i] - d->qfrc_constraint[i];
  if (m->solver == 1) { chol_solve(d->qL, c->Mgrad, c->grad, nv); return; }
  REAL *H = d->w_H;
  memcpy(H, d->qM, sizeof(REAL) * nv * nv);
  for (int r = 0; r < ne; r++) {
    int active = (c->Jaref[r] < 0) || r < d->ne;
    if (d->efc_floss[r] > 0) { REAL rf = d->efc_floss[r] / d->efc_D[r]; active = c->Jaref[r] > -rf && c->Jaref[r] < rf; } /* quadratic zone only */
    if (!active) continue;
    const REAL *J = d->efc_J + r * nv;
    for (int i = 0; i < nv; i++) {
      if (J[i] == 0) continue;
      REAL s = J[i] * d->efc_D[r];
      for (int j = 0; j < nv; j++) H[i * nv + j] += s * J[j];
    }
  }
  chol_factor(d->w_L, H, nv);
  chol_solve(d->w_L, c->Mgrad, c->grad, nv);
}
static lspoint ls_point(const amxo_data *d, const sctx *c, const REAL *jv, const REAL *qg, REAL alpha) {
  REAL q0 = qg[0], q1 = qg[1], q2 = qg[2];
  for (int r = 0; r < d->nefc; r++) {
    REAL x = c->Jaref[r] + alpha * jv[r];
    if (d->efc_floss[r] > 0) { /* friction row: the zone at alpha decides the quadratic's coefficients [mjx:solver._while_loop ls point] */
      REAL f = d->efc_floss[r], rf = f / d->efc_D[r];
      if (x <= -rf) { q0 += f * (-(REAL)0.5 * rf - c->Jaref[r]); q1 += -f * jv[r]; continue; }
      if (x >= rf) { q0 += f * (-(REAL)0.5 * rf + c->Jaref[r]); q1 += f * jv[r]; continue; }
      q0 += (REAL)0.5 * c->Jaref[r] * c->Jaref[r] * d->efc_D[r]; q1 += jv[r] * c->Jaref[r] * d->efc_D[r]; q2 += (REAL)0.5 * jv[r] * jv[r] * d->efc_D[r];
      continue;
    }
    if (x < 0 || r < d->ne) {
      q0 += (REAL)0.5 * c->Jaref[r] * c->Jaref[r] * d->efc_D[r];
      q1 += jv[r] * c->Jaref[r] * d->efc_D[r];
      q2 += (REAL)0.5 * jv[r] * jv[r] * d->efc_D[r];
    }
  }
  lspoint p;
  p.alpha = alpha;
  p.cost = alpha * alpha * q2 + alpha * q1 + q0;
  p.deriv0 = 2 * alpha * q2 + q1;
  p.deriv1 = 2 * q2 + (q2 == 0 ? MINVAL : 0);
  p.q1 = q1; p.q2 = q2;
  return p;
}
/* sum over the rows of q0_row(alpha) - q0_row(0), each difference formed before it is added: exactly zero for every row whose
 * activity (friction rows: zone) is the same at alpha as at 0 [amx_kernels.hip: Env::linesearch, "q0l - q0_start"] */
static REAL ls_q0_delta(const amxo_data *d, const sctx *c, const REAL *jv, REAL alpha) {
  REAL s = 0;
  for (int r = 0; r < d->nefc; r++) {
    REAL z = c->Jaref[r], x = z + alpha * jv[r], D = d->efc_D[r], qa, qz;
    if (d->efc_floss[r] > 0) {
      REAL f = d->efc_floss[r], rf = f / D;
      qa = x <= -rf ? f * (-(REAL)0.5 * rf - z) : (x >= rf ? f * (-(REAL)0.5 * rf + z) : (REAL)0.5 * z * z * D);
      qz = z <= -rf ? f * (-(REAL)0.5 * rf - z) : (z >= rf ? f * (-(REAL)0.5 * rf + z) : (REAL)0.5 * z * z * D);
    } else {
      REAL q = (REAL)0.5 * z * z * D;
      qa = (x < 0 || r < d->ne) ? q : 0;
      qz = (z < 0 || r < d->ne) ? q : 0;
    }
    s += qa - qz;
  }
  return s;
}
/* the next trial point of the 1-D Newton iteration on the derivative; rule bit 1: reciprocal and multiply, d1 clamped from below */
static inline REAL ls_next_alpha(const amxo_model *m, const lspoint *p) {
  if (m->ls_rule & 2) { REAL d1 = r_max(p->deriv1, MINVAL); return p->alpha - p->deriv0 * ((REAL)1 / d1); }
  return p->alpha - p->deriv0 / p->deriv1;
}
/* returns the cost decrease of the accepted step as the search itself knows it (0: no step) - the solver's `improvement` under rule bit 0 */
static REAL linesearch(const amxo_model *m, amxo_data *d, sctx *c) {
  int nv = m->nv, ne = d->nefc;
  REAL *mv = d->w_e, *jv = d->w_jv;
  REAL sn = 0;
  for (int i = 0; i < nv; i++) sn += c->search[i] * c->search[i];
  REAL smag = r_sqrt(sn) * m->meaninertia * (REAL)(nv > 1 ? nv : 1);
  REAL gtol = m->tolerance * m->ls_tolerance * smag;
  mul_m(m, d, mv, c->search);
  for (int r = 0; r < ne; r++) {
    REAL s = 0;
    for (int i = 0; i < nv; i++) s += d->efc_J[r * nv + i] * c->search[i];
    jv[r] = s;
  }
  REAL qg[3] = {c->gauss, 0, 0};
  for (int i = 0; i < nv; i++) {
    qg[1] += c->search[i] * c->Ma[i] - c->search[i] * d->qfrc_smooth[i];
    qg[2] += (REAL)0.5 * c->search[i] * mv[i];
  }
  lspoint p0 = ls_point(d, c, jv, qg, 0);
  lspoint lo = ls_point(d, c, jv, qg, ls_next_alpha(m, &p0)), hi;
  if (lo.deriv0 < p0.deriv0) { hi = p0; } else { hi = lo; lo = p0; }
  int swap = 1, it = 0;
  while (1) {
    int done = it >= m->ls_iterations;
    done |= !swap;
    done |= (lo.deriv0 < 0) && (lo.deriv0 > -gtol);
    done |= (hi.deriv0 > 0) && (hi.deriv0 < gtol);
    if (done) break;
    lspoint lo_next = ls_point(d, c, jv, qg, ls_next_alpha(m, &lo));
    lspoint hi_next = ls_point(d, c, jv, qg, ls_next_alpha(m, &hi));
    lspoint mid = ls_point(d, c, jv, qg, (REAL)0.5 * (lo.alpha + hi.alpha));
    int s_lo_next = (lo.deriv0 > 0) || (lo.deriv0 < lo_next.deriv0);
    if (s_lo_next) lo = lo_next;
    int s_lo_mid = (mid.deriv0 < 0) && (lo.deriv0 < mid.deriv0);
    if (s_lo_mid) lo = mid;
    int s_hi_next = (hi.deriv0 < 0) || (hi.deriv0 > hi_next.deriv0);
    if (s_hi_next) hi = hi_next;
    int s_hi_mid = (mid.deriv0 > 0) && (hi.deriv0 > mid.deriv0);
    if (s_hi_mid) hi = mid;
    swap = s_lo_next | s_lo_mid | s_hi_next | s_hi_mid;
    it++;
  }
  /* rule 0 [mjx:solver._linesearch]: both sides carry the whole constant term */
  int improved0 = (lo.cost < p0.cost) || (hi.cost < p0.cost);
  int pick_lo0 = lo.cost < hi.cost;
  /* rule bit 0 [amx_kernels.hip: Env::linesearch]: the decrease itself, constant terms differenced per row */
  int improved1 = improved0, pick_lo1 = pick_lo0;
  REAL lo_delta = lo.cost - p0.cost, hi_delta = hi.cost - p0.cost;
  REAL dl1 = 0, dh1 = 0;
  if ((m->ls_rule & 1) || m->ls_stats_on) {
    REAL ql, qh;
    if (m->ls_rule & 2) { /* alpha^2 q2 + alpha q1 = alpha d0 - alpha^2 d1 / 2, from the ends' own derivatives */
      ql = lo.alpha * (lo.deriv0 - (REAL)0.5 * lo.alpha * lo.deriv1);
      qh = hi.alpha * (hi.deriv0 - (REAL)0.5 * hi.alpha * hi.deriv1);
    } else { /* the same quadratic from the ends' summed coefficients */
      ql = lo.alpha * lo.alpha * lo.q2 + lo.alpha * lo.q1;
      qh = hi.alpha * hi.alpha * hi.q2 + hi.alpha * hi.q1;
    }
    REAL dl = ql + ls_q0_delta(d, c, jv, lo.alpha), dh = qh + ls_q0_delta(d, c, jv, hi.alpha);
    improved1 = (dl < 0) || (dh < 0);
    pick_lo1 = dl < dh;
    dl1 = dl; dh1 = dh;
    if (m->ls_rule & 1) { lo_delta = dl; hi_delta = dh; }
  }
  if (m->ls_stats_on) {
    d->ls_stat[0]++; d->ls_stat[1] += improved0; d->ls_stat[2] += improved1;
    d->ls_stat[3] += improved0 != improved1;
    d->ls_differed |= improved0 != improved1;
    if (improved0 != improved1) {
      double dec = fabs((double)r_min(dl1, dh1)) / fmax(fabs((double)p0.cost), 1e-300);
      if (dec > d->ls_acc_max) d->ls_acc_max = dec;
    }
    if (improved0 && improved1 && pick_lo0 != pick_lo1 && lo.alpha != hi.alpha) {
      double a = (double)(lo.cost - p0.cost), b = (double)(hi.cost - p0.cost);
      double gap = fabs(a - b) / fmax(fmax(fabs(a), fabs(b)), 1e-300);
      d->ls_stat[6]++;
      if (gap > d->ls_gap_max) d->ls_gap_max = gap;
    }
  }
  int improved = (m->ls_rule & 1) ? improved1 : improved0;
  int pick_lo = (m->ls_rule & 1) ? pick_lo1 : pick_lo0;
  REAL alpha = pick_lo ? lo.alpha : hi.alpha;
  if (improved) {
    for (int i = 0; i < nv; i++) { c->qacc[i] += alpha * c->search[i]; c->Ma[i] += alpha * mv[i]; }
    for (int r = 0; r < ne; r++) c->Jaref[r] += alpha * jv[r];
  }
  return improved ? -r_min(lo_delta, hi_delta) : 0;
}
static REAL ctx_cost_at(const amxo_model *m, amxo_data *d, sctx *c, const REAL *qacc) {
  int nv = m->nv;
  memcpy(c->qacc, qacc, sizeof(REAL) * nv);
  mul_m(m, d, c->Ma, c->qacc);
  for (int r = 0; r < d->nefc; r++) {
    REAL s = 0;
    for (int i = 0; i < nv; i++) s += d->efc_J[r * nv + i] * c->qacc[i];
    c->Jaref[r] = s - d->efc_aref[r];
  }
  c->cost = (REAL)INFINITY; c->prev_cost = 0;
  update_constraint(m, d, c);
  return c->cost;
}
static void solve(const amxo_model *m, amxo_data *d) {
  int nv = m->nv;
  sctx c;
  c.qacc = d->qacc; c.Ma = d->w_a; c.grad = d->w_b; c.Mgrad = d->w_c; c.search = d->w_d; c.Jaref = d->w_jaref;
  REAL *start = d->qacc_smooth;
  if (!(m->disableflags & DIS_WARMSTART)) {
    REAL cw = ctx_cost_at(m, d, &c, d->qacc_warmstart);
    REAL cs = ctx_cost_at(m, d, &c, d->qacc_smooth);
    if (cw < cs) start = d->qacc_warmstart;
  }
  ctx_cost_at(m, d, &c, start);
  update_gradient(m, d, &c);
  for (int i = 0; i < nv; i++) c.search[i] = -c.Mgrad[i];
  REAL scale = m->meaninertia * (REAL)(nv > 1 ? nv : 1);
  REAL *prev_grad = d->w_f, *prev_Mgrad = d->w_g;
  int niter = 0;
  REAL improvement = (REAL)INFINITY; /* rule bit 0: the accepted step's decrease as the line search knows it, instead of prev_cost - cost */
  while (1) {
    if (m->iterations != 1 || niter > 0) { /* iterations==1 runs the body once unconditionally */
      REAL gn = 0;
      for (int i = 0; i < nv; i++) gn += c.grad[i] * c.grad[i];
      int done = niter >= m->iterations;
      REAL imp = (m->ls_rule & 1) ? improvement : c.prev_cost - c.cost;
      if (m->ls_stats_on && niter > 0 && niter < m->iterations) { /* both forms of the improvement against the threshold */
        REAL i0 = c.prev_cost - c.cost, i1 = improvement, ts = scale * m->tolerance;
        if ((i0 < ts) != (i1 < ts)) {
          double gap = fabs((double)(i0 - i1)) / fmax(fmax(fabs((double)c.cost), fabs((double)c.prev_cost)), (double)ts);
          d->ls_stat[8]++;
          if (gap > d->ls_stop_max) d->ls_stop_max = gap;
        }
      }
      if (m->ls_rule & 2) { /* multiplied out [amx_kernels.hip: Env::solve] */
        REAL ts = scale * m->tolerance;
        done |= imp < ts;
        done |= gn < ts * ts;
      } else {
        done |= (imp / scale) < m->tolerance;
        done |= (r_sqrt(gn) / scale) < m->tolerance;
      }
      if (done) break;
    }
    improvement = linesearch(m, d, &c);
    memcpy(prev_grad, c.grad, sizeof(REAL) * nv); memcpy(prev_Mgrad, c.Mgrad, sizeof(REAL) * nv);
    update_constraint(m, d, &c);
    update_gradient(m, d, &c);
    if (m->solver == 2) {
      for (int i = 0; i < nv; i++) c.search[i] = -c.Mgrad[i];
    } else { /* Polak-Ribiere */
      REAL num = 0, den = 0;
      for (int i = 0; i < nv; i++) { num += c.grad[i] * (c.Mgrad[i] - prev_Mgrad[i]); den += prev_grad[i] * prev_Mgrad[i]; }
      REAL beta = r_max((REAL)0, num / r_max(MINVAL, den));
      for (int i = 0; i < nv; i++) c.search[i] = -c.Mgrad[i] + beta * c.search[i];
    }
    niter++;
    if (m->iterations == 1) break;
  }
  d->solver_niter = niter;
  if (m->ls_stats_on) { d->ls_stat[4] += niter; d->ls_stat[5]++; d->ls_stat[7] += d->ls_differed; d->ls_differed = 0; }
  memcpy(d->qacc_warmstart, d->qacc, sizeof(REAL) * nv);
}

/* ------------------------------------------------------------------ forward / step [mjx:forward.forward / step / _euler / _advance] */
void amxo_forward(const amxo_model *m, amxo_data *d) {
  kinematics(m, d);
  com_pos(m, d);
  crb(m, d);
  chol_factor(d->qL, d->qM, m->nv);
  if (!(m->disableflags & (DIS_CONTACT | DIS_CONSTRAINT))) collision(m, d);
  make_constraint(m, d);
  com_vel(m, d);
  passive(m, d);
  rne(m, d);
  fwd_actuation(m, d);
  fwd_acceleration(m, d);
  memset(d->qfrc_constraint, 0, sizeof(REAL) * m->nv);
  d->solver_niter = 0;
  if (m->nefc_max == 0) { /* static: model has no constraint rows at all -> solver skipped */
    memcpy(d->qacc, d->qacc_smooth, sizeof(REAL) * m->nv);
    return;
  }
  solve(m, d);
}

static void quat_integrate(REAL *q, const REAL *w, REAL dt) {
  REAL v[3] = {w[0], w[1], w[2]};
  REAL n = normalize_n(v, 3);
  REAL r[4], out[4];
  axis_angle_quat(r, v, dt * n);
  quat_mul(out, q, r);
  normalize_n(out, 4);
  memcpy(q, out, sizeof(out));
}

void amxo_step(const amxo_model *m, amxo_data *d) {
  int nv = m->nv;
  amxo_forward(m, d);
  REAL *qacc = d->w_a;
  memcpy(qacc, d->qacc, sizeof(REAL) * nv);
  if (!(m->disableflags & DIS_EULERDAMP)) { /* implicit in joint damping: (M + dt*diag(damping)) qacc = qfrc_smooth + qfrc_constraint */
    REAL *H = d->w_H, *L = d->w_L, *rhs = d->w_b;
    memcpy(H, d->qM, sizeof(REAL) * nv * nv);
    for (int i = 0; i < nv; i++) { H[i * nv + i] += m->timestep * m->dof_damping[i]; rhs[i] = d->qfrc_smooth[i] + d->qfrc_constraint[i]; }
    chol_factor(L, H, nv);
    chol_solve(L, qacc, rhs, nv);
  }
  for (int i = 0; i < nv; i++) d->qvel[i] += qacc[i] * m->timestep;
  for (int j = 0; j < m->njnt; j++) {
    int qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j], t = m->jnt_type[j];
    if (t == JFREE) {
      for (int i = 0; i < 3; i++) d->qpos[qa + i] += d->qvel[da + i] * m->timestep;
      quat_integrate(d->qpos + qa + 3, d->qvel + da + 3, m->timestep);
    } else if (t == JBALL) {
      quat_integrate(d->qpos + qa, d->qvel + da, m->timestep);
    } else {
      d->qpos[qa] += d->qvel[da] * m->timestep;
    }
  }
  d->time += m->timestep;
}

/* ------------------------------------------------------------------ accessors */
void amxo_set_state(const amxo_model *m, amxo_data *d, const REAL *qpos, const REAL *qvel, const REAL *ctrl, const REAL *warm) {
  if (qpos) memcpy(d->qpos, qpos, sizeof(REAL) * m->nq);
  if (qvel) memcpy(d->qvel, qvel, sizeof(REAL) * m->nv);
  if (ctrl) memcpy(d->ctrl, ctrl, sizeof(REAL) * m->nu);
  if (warm) memcpy(d->qacc_warmstart, warm, sizeof(REAL) * m->nv);
}
int amxo_sizeof_real(void) { return (int)sizeof(REAL); }
int amxo_get(const amxo_model *m, const amxo_data *d, const char *name, REAL *out, int cap) {
  const REAL *src = 0; int n = 0;
  int nb = m->nbody, nv = m->nv;
#define F(nm, ptr, cnt) if (!strcmp(name, nm)) { src = (ptr); n = (cnt); }
  F("qpos", d->qpos, m->nq) F("qvel", d->qvel, nv) F("qacc", d->qacc, nv) F("qacc_smooth", d->qacc_smooth, nv)
  F("qacc_warmstart", d->qacc_warmstart, nv) F("qfrc_bias", d->qfrc_bias, nv) F("qfrc_passive", d->qfrc_passive, nv)
  F("qfrc_actuator", d->qfrc_actuator, nv) F("qfrc_smooth", d->qfrc_smooth, nv) F("qfrc_constraint", d->qfrc_constraint, nv)
  F("qM", d->qM, nv * nv) F("xpos", d->xpos, nb * 3) F("xquat", d->xquat, nb * 4) F("xmat", d->xmat, nb * 9)
  F("xipos", d->xipos, nb * 3) F("subtree_com", d->subtree_com, nb * 3) F("cinert", d->cinert, nb * 10)
  F("cdof", d->cdof, nv * 6) F("cvel", d->cvel, nb * 6) F("geom_xpos", d->geom_xpos, m->ngeom * 3)
  F("con_dist", d->con_dist, m->ncon) F("con_pos", d->con_pos, m->ncon * 3) F("con_frame", d->con_frame, m->ncon * 9)
  F("efc_J", d->efc_J, d->nefc * nv) F("efc_D", d->efc_D, d->nefc) F("efc_aref", d->efc_aref, d->nefc)
  F("efc_pos", d->efc_pos, d->nefc) F("efc_force", d->efc_force, d->nefc)
#undef F
  if (!strcmp(name, "cx_dbg")) { for (int i = 0; i < 21 && i < cap; i++) out[i] = cx_dbg[i]; return 21; }
  if (!strcmp(name, "nefc")) { if (cap > 0) out[0] = (REAL)d->nefc; return 1; }
  if (!strcmp(name, "solver_niter")) { if (cap > 0) out[0] = (REAL)d->solver_niter; return 1; }
  if (!strcmp(name, "time")) { if (cap > 0) out[0] = d->time; return 1; }
  if (!src) return -1;
  if (n > cap) return -n;
  memcpy(out, src, sizeof(REAL) * n);
  return n;
}
/* ls_rule / ls_stats: test-side switches of a model handle (not thread-safe against a running rollout) */
int amxo_model_set_int(amxo_model *m, const char *name, int v) {
  if (!strcmp(name, "ls_rule")) { m->ls_rule = v; return 0; }
  if (!strcmp(name, "ls_stats")) { m->ls_stats_on = v; memset(m->ls_stat, 0, sizeof(m->ls_stat)); m->ls_gap_max = 0; m->ls_acc_max = 0; m->ls_stop_max = 0; return 0; }
  return -1;
}
void amxo_ls_stats(const amxo_model *m, long long *out, double *gap_max) { memcpy(out, m->ls_stat, sizeof(m->ls_stat)); if (gap_max) { gap_max[0] = m->ls_gap_max; gap_max[1] = m->ls_acc_max; gap_max[2] = m->ls_stop_max; } }
/* folds one data block's counters into its model's (amxo_rollout does it per thread; single-env users call it themselves) */
void amxo_ls_stats_flush(amxo_model *m, amxo_data *d) {
#ifdef _OPENMP
#pragma omp critical(amxo_ls_stats)
#endif
  {
    for (int i = 0; i < 12; i++) { m->ls_stat[i] += d->ls_stat[i]; d->ls_stat[i] = 0; }
    if (d->ls_stop_max > m->ls_stop_max) m->ls_stop_max = d->ls_stop_max;
    d->ls_stop_max = 0;
    if (d->ls_gap_max > m->ls_gap_max) m->ls_gap_max = d->ls_gap_max;
    if (d->ls_acc_max > m->ls_acc_max) m->ls_acc_max = d->ls_acc_max;
    d->ls_gap_max = 0; d->ls_acc_max = 0;
  }
}
int amxo_model_int(const amxo_model *m, const char *name) {
  if (!strcmp(name, "ls_rule")) return m->ls_rule;
  if (!strcmp(name, "ncon")) return m->ncon;
  if (!strcmp(name, "nefc_max")) return m->nefc_max;
  if (!strcmp(name, "nq")) return m->nq;
  if (!strcmp(name, "nv")) return m->nv;
  if (!strcmp(name, "nu")) return m->nu;
  return -1;
}

/* ------------------------------------------------------------------ shoot, batched [reference ambersim/trajopt/shooting.py:22-48,152]
 * xs[e,0]=x0[e]; forward; then N x {ctrl=us[e,t]; step; xs[e,t+1]=concat(qpos,qvel)}.  OpenMP over envs. */
int amxo_rollout(const amxo_model *m, int nenv, int nsteps, const REAL *x0, int x0_batched, const REAL *us, REAL *xs, int nthreads) {
  int nx = m->nq + m->nv;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel
#endif
  {
    amxo_data *d = amxo_data_create(m);
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
    for (int e = 0; e < nenv; e++) {
      const REAL *x = x0 + (x0_batched ? (size_t)e * nx : 0);
      REAL *out = xs ? xs + (size_t)e * (nsteps + 1) * nx : 0;
      memcpy(d->qpos, x, sizeof(REAL) * m->nq); memcpy(d->qvel, x + m->nq, sizeof(REAL) * m->nv);
      memset(d->ctrl, 0, sizeof(REAL) * m->nu); memset(d->qacc_warmstart, 0, sizeof(REAL) * m->nv);
      d->time = 0;
      if (out) memcpy(out, x, sizeof(REAL) * nx);
      amxo_forward(m, d); /* shoot() calls forward once before the scan; it leaves qacc_warmstart = qacc */
      for (int t = 0; t < nsteps; t++) {
        memcpy(d->ctrl, us + ((size_t)e * nsteps + t) * m->nu, sizeof(REAL) * m->nu);
        amxo_step(m, d);
        if (out) { memcpy(out + (size_t)(t + 1) * nx, d->qpos, sizeof(REAL) * m->nq); memcpy(out + (size_t)(t + 1) * nx + m->nq, d->qvel, sizeof(REAL) * m->nv); }
      }
    }
    if (m->ls_stats_on) amxo_ls_stats_flush((amxo_model *)m, d);
    amxo_data_destroy(d);
  }
  return 0;
}
int amxo_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
#ifdef AMXO_COUNT_FLOPS
} /* extern "C" */
#endif
