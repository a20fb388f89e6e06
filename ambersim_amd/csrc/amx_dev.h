// amx_dev.h — device-side model/plan description shared by amx_api.cpp (host) and amx_kernels.hip.
// Internal to ambersim_amd/csrc; the public C ABI is include/amx.h.
#pragma once
#include <stdint.h>

#define AMX_WAVE 64
#define AMX_MAX_NV 32  // register-row Cholesky capacity (one matrix row per lane, <=32 columns)

// One pool of ints and one of floats in HBM; every array is an offset into its pool.
struct DevModel {
  const int *ip;
  const float *fp;
  // sizes
  int nq, nv, nu, nb, njnt, ngc, npair, ncon, neq, nlim, nefc, nlevel, nM, nroot, ldm, ldj, nchol;
  int iterations, ls_iterations, solver, disableflags;
  float dt, gx, gy, gz, tolerance, ls_tolerance, impratio, scale /* meaninertia*max(1,nv) */;
  // --- int pool offsets
  int body_parent, body_root /*slot into com[]*/, body_jntadr, body_jntnum, body_dofadr, body_dofnum;
  int level_adr /*[nlevel+1]*/, level_body /*[nb] bodies sorted by depth*/;
  int child_adr /*[nb+1]*/, child_list;
  int root_body /*[nroot]*/;
  int jnt_type, jnt_qadr, jnt_dadr, jnt_body;
  int dof_body, dof_jnt, dof_kind /*0 free-trans k,1 free/ball-rot,2 hinge,3 slide*/, dof_axis /*k within joint*/;
  int dof_actadr /*[nv+1]*/, dof_actlist;
  int M_i, M_j /*[nM] sparse entries (i, ancestor j)*/;
  int act_qadr, act_dadr, act_flags /*bit0 ctrllimited bit1 forcelimited bit2 affine bias*/;
  int gc_body, gc_type;
  int pair_g1, pair_g2 /*indices into the collision-geom list*/, pair_fn, pair_conadr, pair_vertadr, pair_vertnum,
      pair_body1, pair_body2, pair_dofadr /*[npair+1]*/, pair_doflist /*union of both ancestor chains*/,
      pair_dofsign /*bit0: in chain of body2 (+), bit1: in chain of body1 (-)*/, pair_root1, pair_root2;
  int con_pair /*[ncon]*/, con_slot /*[ncon] index within its pair*/;
  int lim_qadr, lim_dadr;
  int eq_q1, eq_d1, eq_q2, eq_d2 /* -1 when absent */;
  // --- float pool offsets
  int body_pos, body_quat, body_ipos, body_iquat, body_mass, body_inertia;
  int jnt_pos, jnt_axis, jnt_stiff;
  int qpos0, qpos_spring;
  int dof_armature, dof_damping;
  int act_gear, act_gain, act_bias /*[nu,3]*/, act_ctrlrange, act_forcerange;
  int gc_pos, gc_quat, gc_size;
  int pair_mu, pair_solref, pair_solimp, pair_margin, pair_invw /*body_invweight0 sum*/, convex_vert;
  int lim_range, lim_margin, lim_solref, lim_solimp, lim_invw;
  int eq_data, eq_solref, eq_solimp, eq_invw1, eq_invw2;
};

// LDS layout of one environment (float offsets), computed on the host.
struct LdsPlan {
  int qpos, qvel, ctrl, warm;
  int xpos, xquat, com, xanchor, xaxis;
  int cinert, crb, cdof, fcrb;
  int cvel, cacc, cfrc;
  int M, H, Lb;
  int fsmooth, asmooth, qacc, Ma, grad, Mgrad, search, mv, fcon, tmp, fbias;
  int actf;
  int D, aref, Jaref, jv, J;
  int cdist, cpos, cframe, gpos, gmat;
  int prof;   // 24 floats, used by the diagnostic build only
  int total;  // floats per environment
};

enum { AMX_MODE_ROLLOUT = 0, AMX_MODE_STEP = 1, AMX_MODE_FORWARD = 2 };

struct LaunchArgs {
  int mode, nenv, nsteps, x0_batched;
  const float *x0;     // rollout: [nenv,nq+nv] or [nq+nv]
  const float *us;     // rollout: [nenv,nsteps,nu]; step/forward: ctrl [nenv,nu]
  float *xs;           // rollout: [nenv,nsteps+1,nq+nv] or NULL
  float *xfinal;       // rollout: [nenv,nq+nv] or NULL
  float *qpos, *qvel, *warm, *time;  // step/forward: in place state
  const float *warm_in;              // forward: optional warm start
  // forward outputs (any may be NULL)
  float *o_qacc, *o_qacc_smooth, *o_qfrc_bias, *o_qfrc_constraint, *o_qM, *o_xpos, *o_xquat, *o_cdist;
  int *o_nefc;
  float *prof;  // diagnostic build only: per-env per-phase cycle sums [nenv,24]
};
