/* amx.h — C ABI of the MI355X batched rigid-body rollout engine.
 *
 * This is the drop-in boundary for the hot path of Caltech-AMBER/ambersim: everything the
 * reference reaches through `mujoco.mjx` is bound here.  The reference has no FFI of its own
 * (it is pure Python over JAX); each entry point cites the Python call site it replaces:
 *
 *   amx_model_create   <- mjx.device_put(mj_model)        ambersim/utils/io_utils.py:225, rl/base.py:52
 *   amx_forward        <- mjx.forward(sys, data)          ambersim/rl/base.py:85, trajopt/shooting.py:36
 *   amx_step(_to)      <- lax.scan of mjx.step            ambersim/rl/base.py:88-96  (MjxEnv.pipeline_step)
 *   amx_rollout        <- shoot(m, x0, us) under vmap     ambersim/trajopt/shooting.py:22-48,152
 *   amx_quadcost       <- StaticGoalQuadraticCost.cost    ambersim/trajopt/cost.py:62-85 (batched over rollouts)
 *   amx_fd_grad        <- shoot + terminal cost, forward-differenced over ctrl (BASELINE.json configs[3]; composes
 *                         ambersim/trajopt/shooting.py:22-48 and ambersim/trajopt/cost.py:62-85 with Q = R = 0)
 *
 * Conventions: plain pointers and sizes only.  Pointers named *_dev are device (HBM)
 * addresses owned by the caller (e.g. torch tensors); every other pointer is host memory.
 * All batched arrays are env-major, contiguous, float32 (the reference physics' precision).
 * Calls are asynchronous and ordered on the given HIP stream (`stream` is a hipStream_t
 * passed as void*; NULL = the default stream).  Functions return 0 on success and a negative
 * AMX_E* code otherwise; amx_last_error() gives a thread-local message.  Nothing throws
 * across this boundary.  A model handle is immutable and may be shared between threads.
 */
#ifndef AMX_H
#define AMX_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AMX_VERSION 3 /* 2: convex hull tables for convex-convex pairs; 3: joint friction loss (appended fields) */

enum {
  AMX_OK = 0,
  AMX_EINVAL = -1,      /* bad argument / inconsistent model description      -> ValueError */
  AMX_EUNSUPPORTED = -2,/* feature outside the supported physics subset       -> NotImplementedError */
  AMX_ENOMEM = -3,      /* host or device allocation failed                   -> MemoryError */
  AMX_EHIP = -4,        /* HIP runtime error (message has hipGetErrorString)  -> RuntimeError */
  AMX_ECAPACITY = -5    /* model exceeds a kernel capacity (LDS / lanes)      -> NotImplementedError */
};

/* Compiled host model: MuJoCo's field names, float64/int32 host arrays.  Produced by
 * ambersim_amd/mjcf.py (the counterpart of mujoco.MjModel.from_xml_path). */
typedef struct amx_model_desc {
  int version;          /* AMX_VERSION */
  int nq, nv, nu, nbody, njnt, ngeom, neq, npair, nconvexvert;
  int nhull, nhullvert, nhullface, nhullfacevert, nhulledge; /* convex hulls used by convex-convex pairs (below) */
  /* option / statistic */
  double timestep, gravity[3], tolerance, ls_tolerance, impratio, meaninertia;
  int iterations, ls_iterations, solver /*1 CG, 2 Newton*/, integrator /*0 Euler*/, disableflags;
  /* bodies [nbody] */
  const int *body_parentid, *body_rootid, *body_jntadr, *body_jntnum, *body_dofadr, *body_dofnum;
  const double *body_pos /*[nbody,3]*/, *body_quat /*[nbody,4]*/, *body_ipos, *body_iquat,
      *body_mass, *body_inertia /*[nbody,3]*/, *body_invweight0 /*[nbody,2]*/;
  /* joints [njnt] */
  const int *jnt_type /*0 free 1 ball 2 slide 3 hinge*/, *jnt_qposadr, *jnt_dofadr, *jnt_bodyid, *jnt_limited;
  const double *jnt_pos, *jnt_axis, *jnt_range /*[njnt,2]*/, *jnt_margin, *jnt_stiffness,
      *jnt_solref /*[njnt,2]*/, *jnt_solimp /*[njnt,5]*/;
  const double *qpos0 /*[nq]*/, *qpos_spring /*[nq]*/;
  /* dofs [nv] */
  const int *dof_bodyid, *dof_jntid, *dof_parentid;
  const double *dof_armature, *dof_damping, *dof_invweight0;
  /* geoms [ngeom] */
  const int *geom_type, *geom_bodyid;
  const double *geom_pos, *geom_quat, *geom_size /*[ngeom,3]*/;
  /* static collision pairs [npair] (contype/conaffinity/parent filters already applied) */
  const int *pair_geom1, *pair_geom2, *pair_fn /*0 plane-sphere 1 plane-capsule 2 plane-convex
      3 sphere-sphere 4 sphere-capsule 5 capsule-capsule 6 convex-convex (box / mesh hulls) 7 sphere-convex 11 capsule-convex*/, *pair_vertadr, *pair_vertnum;
  const double *pair_friction /*[npair,3]*/, *pair_solref /*[npair,2]*/, *pair_solimp /*[npair,5]*/,
      *pair_margin /*[npair] margin-gap*/, *convex_vert /*[nconvexvert,3] geom-local hull vertices*/;
  /* convex hulls of the box / mesh geoms in convex-convex pairs (pair_fn 6): polygon faces (coplanar triangles merged,
   * vertices counter-clockwise seen from outside), outward unit face normals, unique edges; everything geom-local.
   * Limits: <= 64 vertices, <= 128 faces of <= 8 vertices, <= 192 edges per hull (a triangulated 64-vertex hull has 124 faces, 186 edges). */
  const int *pair_hull1, *pair_hull2 /*[npair] hull of geom1 / geom2 for pair_fn 6 (pair_fn 7: of geom2 only), else -1*/;
  const int *hull_vertadr, *hull_vertnum, *hull_faceadr, *hull_facenum, *hull_edgeadr, *hull_edgenum /*[nhull]*/;
  const double *hull_vert /*[nhullvert,3]*/, *hull_face_normal /*[nhullface,3]*/;
  const int *hull_face_vertadr, *hull_face_vertnum /*[nhullface] into hull_facevert*/, *hull_facevert /*[nhullfacevert] vertex ids local to the hull*/,
      *hull_edge /*[nhulledge,2] vertex ids local to the hull*/;
  /* actuators [nu] (joint transmissions) */
  const int *actuator_trnid, *actuator_biastype /*0 none 1 affine*/, *actuator_ctrllimited, *actuator_forcelimited;
  const double *actuator_gear, *actuator_gainprm /*[nu,3]*/, *actuator_biasprm /*[nu,3]*/,
      *actuator_ctrlrange /*[nu,2]*/, *actuator_forcerange /*[nu,2]*/;
  /* equalities [neq]: joint (polycoef), connect, weld - see eq_type below */
  const int *eq_obj1id, *eq_obj2id, *eq_active;
  const double *eq_data /*[neq,11]*/, *eq_solref /*[neq,2]*/, *eq_solimp /*[neq,5]*/;
  /* dry joint friction [nv] (version 3): a dof with dof_frictionloss > 0 carries one friction-loss constraint row (force
   * clamped to +-frictionloss: quadratic inside, linear outside), regularised by dof_solref / dof_solimp
   * (MJCF solreffriction / solimpfriction).  Reaches the reference through URDF <dynamics friction=...>. */
  /* equality types (version 3): eq_type[e] = 0 connect, 1 weld, 2 joint (MuJoCo's mjtEq).  obj ids are joint ids for joint
   * equalities and body ids for connect / weld (0 = the world).  eq_data columns:
   *   joint   [0:5] polycoef
   *   connect [0:3] anchor in body1's frame, [3:6] the same point in body2's frame at qpos0
   *   weld    [0:3] the weld point in body2's frame, [3:6] the same point in body1's frame at the relative pose, [6:10] relpose
   *           quaternion (orientation of body2 in body1's frame), [10] torquescale */
  const int *eq_type /*[neq]*/;
  const double *dof_frictionloss /*[nv]*/, *dof_solref /*[nv,2]*/, *dof_solimp /*[nv,5]*/;
} amx_model_desc;

typedef struct amx_model amx_model; /* opaque: device-resident constant block + launch plan */

/* Derived outputs of amx_forward; any pointer may be NULL to skip that field. */
typedef struct amx_forward_out {
  float *qacc_dev;            /* [nenv,nv]  constrained acceleration                        */
  float *qacc_smooth_dev;     /* [nenv,nv]  unconstrained acceleration                      */
  float *qfrc_bias_dev;       /* [nenv,nv]  Coriolis/centrifugal/gravity (RNE)              */
  float *qfrc_constraint_dev; /* [nenv,nv]  J^T * efc_force                                 */
  float *qM_dev;              /* [nenv,nv,nv] dense joint-space inertia (CRBA)              */
  float *xpos_dev;            /* [nenv,nbody,3] body frame origins                          */
  float *xquat_dev;           /* [nenv,nbody,4] body frame orientations                     */
  float *contact_dist_dev;    /* [nenv,ncon] signed distance per contact slot               */
  int *nefc_dev;              /* [nenv] number of active constraint rows                    */
} amx_forward_out;

/* Library / device probes (no GPU work). */
int amx_version(void);
const char *amx_last_error(void);

/* Uploads a compiled model to `device` and builds the launch plan (tree levels, ancestor
 * chains, constraint row tables).  Replaces mjx.device_put(mj_model). */
int amx_model_create(const amx_model_desc *desc, int device, amx_model **out);
int amx_model_destroy(amx_model *m);
/* Static sizes of a created model: name in {"nq","nv","nu","nbody","ncon","nefc_max",
 * "lds_bytes","lanes_per_env","envs_per_block","env_slab_bytes","const_bytes","specialised"}.
 * "simd_pairing" is a property of the device, measured by one probe launch (cached): 1 when waves w and w + 4 of an 8-wave
 * workgroup share a SIMD (what the rollout kernel's fair sharing of a SIMD presumes), 0 when they do not (the kernel then leaves
 * the arbitration to the hardware; it re-checks on every step), -1 when the probe could not run. */
int amx_model_query(const amx_model *m, const char *name, int *out);

/* mjx.forward: derived quantities for the given state, no integration.  `qacc_warmstart_dev`
 * may be NULL (cold start from qacc_smooth). */
int amx_forward(const amx_model *m, int nenv, const float *qpos_dev, const float *qvel_dev,
                const float *ctrl_dev, const float *qacc_warmstart_dev, const amx_forward_out *out,
                void *stream);

/* MjxEnv.pipeline_step: `nsub` physics steps with ctrl held, in place on
 * qpos[nenv,nq], qvel[nenv,nv], qacc_warmstart[nenv,nv]; time[nenv] may be NULL. */
int amx_step(const amx_model *m, int nenv, int nsub, float *qpos_dev, float *qvel_dev,
             const float *ctrl_dev, float *qacc_warmstart_dev, float *time_dev, void *stream);

/* The same step out of place, as mjx.step is (it returns a new Data, reference ambersim/rl/base.py:93): reads
 * qpos/qvel/qacc_warmstart/time, writes the *_out buffers (same shapes).  An output may be its input (that is amx_step);
 * a functional host layer needs no copies in front of the launch.  qacc_warmstart_dev NULL: cold start; time_dev NULL:
 * time counts from 0; qacc_warmstart_out_dev / time_out_dev NULL: not written. */
int amx_step_to(const amx_model *m, int nenv, int nsub, const float *qpos_dev, const float *qvel_dev,
                const float *ctrl_dev, const float *qacc_warmstart_dev, const float *time_dev,
                float *qpos_out_dev, float *qvel_out_dev, float *qacc_warmstart_out_dev, float *time_out_dev,
                void *stream);

/* A whole wrapped env.step in one call (two launches on `stream`: the physics of amx_step_to, then ONE elementwise kernel for
 * everything the reference's env and its training wrappers do after it).  Replaces, for envs whose observation and reward fit
 * the descriptor below, the body of PendulumSwingupEnv.step (reference ambersim/rl/pendulum/swingup.py:62-122: observation
 * (cos q, sin q, qvel), reward -w_theta*wrap(q - pi)^2 - w_dtheta*qvel^2 - w_u*ctrl^2, done = 0) and what brax's EpisodeWrapper and
 * AutoResetWrapper add around it inside ppo.train (reference examples/rl/pendulum/ex_swingup.py:25-43 passes episode_length):
 * step counter, truncation at the episode length, done, episode returns, and the select that restarts a finished env from the
 * state it was first reset to.  The reference's env.step is one jitted XLA program; torch would run these as dozens of small
 * kernels per step.
 *   observation term k: kind 0 qpos[i], 1 cos(qpos[i]), 2 sin(qpos[i]), 3 qvel[i]                        (i = obs_index[k])
 *   reward = - sum_k rew_weight[k] * t_k^2, term kind 0 wrap_to_pi(qpos[i] - target), 1 qpos[i] - target, 2 qvel[i], 3 ctrl[i]
 *   (wrap_to_pi(x) = atan2(sin x, cos x); ctrl is the action as given, not clamped: Data.ctrl of the reference) */
#define AMX_TAIL_MAX_OBS 16
#define AMX_TAIL_MAX_REW 8
typedef struct amx_env_tail_desc {
  int nobs, obs_kind[AMX_TAIL_MAX_OBS], obs_index[AMX_TAIL_MAX_OBS];
  int nrew, rew_kind[AMX_TAIL_MAX_REW], rew_index[AMX_TAIL_MAX_REW];
  float rew_target[AMX_TAIL_MAX_REW], rew_weight[AMX_TAIL_MAX_REW];
  int episode_length; /* control steps after which an episode is truncated */
} amx_env_tail_desc;
/* Device pointers, fp32, env-major.  Inputs: the first (reset) state and observation of every env and the episode counters
 * before the step.  Outputs (written by the call, must not alias the inputs of the same call): the state after the step with
 * finished envs restarted (qpos, qvel, ctrl, qacc_warmstart, time), qacc (the solver's result before the restart select),
 * obs (of the restarted state), terminal_obs (before the restart), reward, done, truncation and the counters after the step.
 * step_counter_{in,out}: one float shared by all envs (the env's own step count), may be NULL. */
typedef struct amx_env_tail_io {
  const float *first_qpos, *first_qvel, *first_ctrl, *first_warm, *first_time, *first_obs;
  const float *steps, *episode_return, *last_episode_return, *episodes, *step_counter_in;
  float *qpos_out, *qvel_out, *ctrl_out, *warm_out, *time_out, *qacc_out;
  float *obs_out, *terminal_obs_out, *reward_out, *done_out, *truncation_out;
  float *steps_out, *episode_return_out, *last_episode_return_out, *episodes_out, *step_counter_out;
} amx_env_tail_io;
int amx_env_step(const amx_model *m, int nenv, int nsub, const float *qpos_dev, const float *qvel_dev, const float *action_dev,
                 const float *qacc_warmstart_dev, const float *time_dev, const amx_env_tail_desc *tail, const amx_env_tail_io *io,
                 void *stream);

/* vmap(shoot): xs[e,0,:] = x0[e], xs[e,t+1,:] = concat(qpos,qvel) after step t under us[e,t,:].
 * x0 is [nenv,nq+nv] when x0_batched != 0, else one shared [nq+nv] row.  State stays on chip
 * for the whole horizon; HBM traffic is us in, xs out.  xs_dev may be NULL when only the final
 * state is wanted (xfinal_dev [nenv,nq+nv], may also be NULL). */
int amx_rollout(const amx_model *m, int nenv, int nsteps, const float *x0_dev, int x0_batched,
                const float *us_dev, float *xs_dev, float *xfinal_dev, void *stream);

/* Batched StaticGoalQuadraticCost.cost over rollouts: cost[e] = 0.5*(sum_t<N (x_t-xg)'Q(x_t-xg)
 * + (x_N-xg)'Qf(x_N-xg) + sum_t u_t'R u_t).  Q,Qf are [nx,nx], R is [nu,nu], xg is [nx]. */
int amx_quadcost(int nenv, int nsteps, int nx, int nu, const float *xs_dev, const float *us_dev,
                 const float *Q_dev, const float *Qf_dev, const float *R_dev, const float *xg_dev,
                 float *cost_dev, void *stream);

/* Finite-difference gradient of the terminal cost of a shooting rollout with respect to the controls
 * (BASELINE.json configs[3], SURVEY.md 8(d) "C4 inputs"; the reference itself has no finite-difference code, this
 * composes its shoot() and StaticGoalQuadraticCost.cost with Q = R = 0).  `nsamples` rollouts of `nsteps` steps start
 * from ONE shared state x0[nq+nv]: rollout 0 runs under us_nominal[nsteps,nu]; rollout k >= 1 adds eps[i] to the single
 * control coordinate c = coord0 + k - 1 (step t = c / nu, actuator i = c % nu; coordinates >= nsteps*nu leave the
 * rollout unperturbed and give grad = 0).  Outputs: J[k] = 0.5 (x_N - xg)' Qf (x_N - xg) for every rollout and
 * grad[k-1] = (J[k] - J[0]) / eps[i].  One call = one perturbation kernel, one fused rollout (final states only, no
 * trajectory traffic) and one cost/difference kernel, all on `stream`.  xfinal_dev ([nsamples,nq+nv]) may be NULL.
 * workspace_dev must hold amx_fd_workspace_bytes() bytes (perturbed controls + final states). */
int amx_fd_workspace_bytes(const amx_model *m, int nsamples, int nsteps, size_t *out);
int amx_fd_grad(const amx_model *m, int nsamples, int nsteps, const float *x0_dev, const float *us_nominal_dev,
                const float *eps_dev /*[nu]*/, int coord0, const float *Qf_dev /*[nx,nx]*/, const float *xg_dev /*[nx]*/,
                float *J_dev /*[nsamples]*/, float *grad_dev /*[nsamples-1]*/, float *xfinal_dev,
                void *workspace_dev, size_t workspace_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* AMX_H */
