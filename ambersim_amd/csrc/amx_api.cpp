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
#include <mutex>
#include <vector>

#include "../../include/amx.h"
#include "amx_dev.h"
#include "amx_plan.h"  // host-only planner: descriptor -> constant image + launch plan (also built alone as libamx_plan.so)

#ifndef AMX_SPECS_HEADER
#define AMX_SPECS_HEADER "amx_specs_gen.h"  // the shipped models; a per-model build (ambersim_amd/_spec.py) points this at its own record
#endif
#include AMX_SPECS_HEADER
#include "amx_kernels.hip"  // single translation unit: kernels + launchers, no relocatable device code

extern "C" int amx_version(void) { return AMX_VERSION; }
extern "C" const char *amx_last_error(void) { return g_err; }

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
  amx_model *wide = nullptr;  // the same model planned on a 64-lane group (static kernels only), for batches of at most 4 * ncu environments
  DevModel dev;
  int spec;  // index into AMX_SPECS of the compile-time specialisation whose plan equals this model's, -1 = generic kernels
  int device, ncu;
  float *d_image;
  int ncon, nefc, lds_bytes;
};

// AMX_NO_GENERIC (per-model builds, ambersim_amd/_spec.py): the library holds only the kernels compiled for its one model - the
// generic kernels are two thirds of the compile time and such a library is loaded for that model's handle alone.
#ifndef AMX_NO_GENERIC
#define AMX_NO_GENERIC 0
#endif
#if AMX_NO_GENERIC
static hipError_t set_lds_limit(int, int) { return hipSuccess; }
#else
template <int G> static hipError_t set_lds_limit_g(int bytes) {
  hipError_t e;
  if ((e = hipFuncSetAttribute((const void *)amx_rollout_kernel<G, DynSpec>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes)) != hipSuccess) return e;
  if ((e = hipFuncSetAttribute((const void *)amx_step_kernel<G, DynSpec>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes)) != hipSuccess) return e;
  return hipFuncSetAttribute((const void *)amx_forward_kernel<G, DynSpec>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}
static hipError_t set_lds_limit(int G, int bytes) {
  return G == 16 ? set_lds_limit_g<16>(bytes) : (G == 32 ? set_lds_limit_g<32>(bytes) : set_lds_limit_g<64>(bytes));
}
#endif

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
  if ((e = hipFuncSetAttribute((const void *)amx_rollout_kernel<S::G, S, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes)) != hipSuccess) return e;
  if ((e = hipFuncSetAttribute((const void *)amx_step_kernel<S::G, S>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes)) != hipSuccess) return e;
  return hipFuncSetAttribute((const void *)amx_forward_kernel<S::G, S>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

// A generated spec applies when the model's whole plan equals it: every integer, the dof tree, the H-sparsity class.
template <class S> static bool spec_matches(const DevModel &m, const amx_model_desc *d, bool tree_H) {
#define X(f) if (m.f != S::f) return false;
  AMX_MODEL_INTS(X)
#undef X
  for (int i = 0; i < m.nv; i++) if (d->dof_parentid[i] != S::dof_parent[i]) return false;
  // the two-pair Hessian (newton_solve_pair2) reads S::p2_col at compile time to decide which register each accumulator column is added to;
  // the columns come from the pairs' dof lists, which none of the integers above fixes (two models can share the tree, the list lengths and
  // the banks / shifts and still put their feet on sibling branches): compare the planner's list for THIS model (plan_model left it in the
  // thread-local p2_col) - ADVICE r04
  if (m.mfma2) for (int k = 0; k < 32; k++) if (p2_col[k] != S::p2_col[k]) return false;
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

// Tooling (not part of include/amx.h; no device call): the index of the compiled-in specialisation the library would run this
// model on (-1: none - the generic kernels), as amx_model_create decides it.  poke_k >= 0 overrides entry poke_k of the planner's
// two-pair column list with poke_v before the comparison - tests/test_abi.py checks with it that a model whose pair dof lists
// differ from a record's is not matched to that record even when every integer of the plan agrees.
extern "C" int amx_spec_index_of(const amx_model_desc *d, int poke_k, int poke_v) {
  if (!d) return -2;
  DevModel m;
  Image im;
  if (plan_model(d, m, im)) return -2;
  if (poke_k >= 0 && poke_k < 32) p2_col[poke_k] = poke_v;
  int spec = find_spec(m, d, im);
  if (spec < 0 && (m.mcomp || m.jfr)) {  // as amx_model_create: without kernels for the dense form, the regular plan
    plan_dense_off = 1;
    const int rc = plan_model(d, m, im);
    plan_dense_off = 0;
    if (rc) return -2;
    if (poke_k >= 0 && poke_k < 32) p2_col[poke_k] = poke_v;
    spec = find_spec(m, d, im);
  }
  return spec;
}

static int model_create_impl(const amx_model_desc *d, int device, amx_model **out, bool wide_plan) {
  if (!d || !out) return fail(AMX_EINVAL, "null argument");
  DevModel m;
  Image im;
  if (int rc = plan_model(d, m, im)) return rc;
  int spec = find_spec(m, d, im);
  if (spec < 0 && (m.mcomp || m.jfr)) {
    // the planner chose its dense form (r05), which only per-model kernels carry, and this library holds none for the model: the regular
    // plan, which the generic kernels run (a per-model build of the library - put_model's default - is compiled for the dense form)
    plan_dense_off = 1;
    const int rc = plan_model(d, m, im);
    plan_dense_off = 0;
    if (rc) return rc;
    spec = find_spec(m, d, im);
  }
  if (wide_plan && (spec < 0 || m.G != 64)) return AMX_EUNSUPPORTED;  // only kept when the library holds static kernels for it
  if (AMX_NO_GENERIC && spec < 0) return fail(AMX_EUNSUPPORTED, "this build of the library holds only the kernels of the model it was compiled for");
  if (m.nbs == 2 && spec < 0)
    return fail(AMX_EUNSUPPORTED, "needs per-model kernels: %d contact slots take two contact rows per lane, which the generic kernels do not carry - "
                "put_model(..., specialise=True) or AMX_SPECIALISE=1 builds them (hipcc)", m.ncon);
  const int lds_bytes = (m.nconst + m.wpb * (AMX_WAVE / m.G) * m.env_floats) * (int)sizeof(float);
  // The dynamic-LDS limit is an attribute of the KERNEL, shared by every model handle of the process (the generic kernels
  // serve any number of models): it is set to the device's 160 KiB, never to this model's own size - a smaller model created
  // after a larger one must not lower the limit under the earlier handle.  (plan_model has checked that lds_bytes fits.)
  const int lds_limit = 160 * 1024;
  amx_model *h = new (std::nothrow) amx_model();
  if (!h) return fail(AMX_ENOMEM, "out of host memory");
  h->ncu = 0;
  h->spec = spec;
  h->device = device; h->ncon = m.ncon; h->nefc = m.nefc; h->lds_bytes = lds_bytes; h->d_image = nullptr;
  hipError_t e;
  DeviceGuard guard(device);
  if ((e = guard.ok() ? hipSuccess : hipErrorInvalidDevice) != hipSuccess || (e = hipMalloc((void **)&h->d_image, im.v.size() * sizeof(float))) != hipSuccess ||
      (e = hipMemcpy(h->d_image, im.v.data(), im.v.size() * sizeof(float), hipMemcpyHostToDevice)) != hipSuccess ||
      (e = set_lds_limit(m.G, lds_limit)) != hipSuccess ||
      (for_spec(spec, [&](auto sp) { e = set_lds_limit_spec<decltype(sp)>(lds_limit); }) && e != hipSuccess)) {
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

extern "C" int amx_model_create(const amx_model_desc *d, int device, amx_model **out) {
  if (int rc = model_create_impl(d, device, out, false)) return rc;
  // A 32-lane model with static kernels gets a second plan on a 64-lane group when the library holds static kernels for that too
  // (tools/gen_specs.py: the *_w64 records): one environment per wavefront, for batches that cannot give every SIMD a wave at two
  // environments per wave (C4: 1024 rollouts = 512 waves on 1024 SIMDs).  Opt-in (AMX_WIDE_SMALL_BATCH=1, see pick_plan): it
  // measured slower than the regular plan, and by default every batch size runs one kernel (results bit-identical across launch sizes).
  amx_model *h = *out;
  // (the wide plan is opt-in at run time - pick_plan, AMX_WIDE_SMALL_BATCH=1 - so its device image is only built when that is set)
  const char *wide_env = getenv("AMX_WIDE_SMALL_BATCH");
  if (h->dev.G == 32 && h->spec >= 0 && !getenv("AMX_NO_WIDE") && wide_env && atoi(wide_env) != 0) {
    amx_plan_force_lanes(64);
    amx_model *w = nullptr;
    const int rc = model_create_impl(d, device, &w, true);
    amx_plan_force_lanes(0);
    if (rc == AMX_OK) h->wide = w;
  }
  return AMX_OK;
}

// the handle a batch of `nenv` environments runs on: the wide plan when the caller has asked for it (AMX_WIDE_SMALL_BATCH=1, read
// per call), there is one, and every environment can have a wave of its own.  OFF by default - measured (profiles/r04_c4_fd_grad.txt):
// C4 on the wide plan 36.5 M env-steps/s against 38.7 M on the regular one.  The premise did not hold: a one-environment 64-lane
// wave-step is not shorter than a two-environment 32-lane one (it loses the two-pair matrix-core form and the gradient out of the
// product, both tied to 32-lane groups), so twice as many waves of the same length gain nothing once every SIMD has one.
static const amx_model *pick_plan(const amx_model *m, int nenv) {
  const int ncu = m->ncu > 0 ? m->ncu : 256;
  if (!m->wide || nenv > 4 * ncu) return m;
  const char *e = getenv("AMX_WIDE_SMALL_BATCH");
  return (e && atoi(e) != 0) ? m->wide : m;
}

extern "C" int amx_model_destroy(amx_model *m) {
  if (!m) return AMX_OK;
  if (m->wide) { amx_model_destroy(m->wide); m->wide = nullptr; }
  {
    DeviceGuard guard(m->device);
    (void)hipFree(m->d_image);
  }
  delete m;
  return AMX_OK;
}

// "simd_pairing": do waves w and w + 4 of an 8-wave workgroup run on one SIMD on this device?  One probe launch per device (one
// workgroup per CU, 512 threads), cached: 1 yes on every workgroup, 0 no (the rollout kernel then leaves the pair to the
// hardware's arbitration: it re-checks per step), -1 the probe could not run.
static int simd_pairing(int device, int ncu) {
  // one probe per device and process, under a lock (handles may be queried from several threads); the probe runs on a stream of its
  // own and waits for that stream only - a null-stream launch + hipDeviceSynchronize from a query would invalidate a stream capture
  // in progress on the caller's side (the PPO collector captures its unroll as a HIP graph)
  static std::mutex mu;
  static int cache[64];
  static bool have[64];
  if (device < 0 || device >= 64) return -1;
  std::lock_guard<std::mutex> lock(mu);
  if (have[device]) return cache[device];
  int verdict = -1;
  const int nwg = ncu > 0 ? ncu : 256;
  int *d = nullptr;
  hipStream_t st = nullptr;
  std::vector<int> h((size_t)nwg * 8, -1);
  if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess) {
    if (hipMalloc(&d, h.size() * sizeof(int)) == hipSuccess) {
      hipLaunchKernelGGL(amx_simd_probe_kernel, dim3(nwg), dim3(512), 0, st, d);
      if (hipGetLastError() == hipSuccess && hipMemcpyAsync(h.data(), d, h.size() * sizeof(int), hipMemcpyDeviceToHost, st) == hipSuccess &&
          hipStreamSynchronize(st) == hipSuccess) {
        verdict = 1;
        for (int b = 0; b < nwg; b++)
          for (int w = 0; w < 4; w++)
            if (h[(size_t)b * 8 + w] < 0 || h[(size_t)b * 8 + w] != h[(size_t)b * 8 + w + 4]) verdict = 0;
      }
      (void)hipFree(d);
    }
    (void)hipStreamDestroy(st);
  }
  cache[device] = verdict;
  have[device] = true;
  return verdict;
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
  else if (n == "wide_plan") *out = m->wide ? 1 : 0;
  else if (n == "simd_pairing") { AMX_ON_DEVICE(m->device); *out = simd_pairing(m->device, m->ncu); }
  else return fail(AMX_EINVAL, "unknown query '%s'", name);
  return AMX_OK;
}

// Waves per workgroup for this launch.  Every workgroup carries one copy of the constants, so fewer, larger workgroups keep
// more environments resident per CU; a batch that does not fill a whole number of residency rounds leaves the last round
// partly empty; and a small batch runs faster spread thinly over all CUs than packed onto a few (a wave-step is latency
// bound, but co-resident waves still share the SIMD's issue slots: 8 waves per CU cost 1.26x the step time of 2, measured on
// the 1024-rollout gradient launch).  Cost model: sum over residency rounds of 1 + 0.05 * (waves per CU in that round);
// ties go to the larger workgroup.
static int choose_wpb(const amx_model *m, int nenv, int minw = AMX_MIN_WAVES_PER_SIMD, double *cost_out = nullptr) {
  if (const char *force = getenv("AMX_FORCE_WPB")) { int w = atoi(force); if (w >= 1 && w <= m->dev.wpb) return w; }
  const int epw = AMX_WAVE / m->dev.G, nwaves = (nenv + epw - 1) / epw, ncu = m->ncu > 0 ? m->ncu : 256;
  int best = 1;
  double best_cost = 1e30;
  for (int w = 1; w <= m->dev.wpb; w++) {
    // powers of two only: a workgroup's waves go round the CU's four SIMDs, so 3, 5, 6 or 7 of them load the SIMDs unevenly and
    // the register file of the fuller ones caps the residency (measured: 6-wave workgroups at 3 waves/SIMD ran at 6 per CU)
    if (w & (w - 1)) continue;
    const int bytes = (m->dev.nconst + w * epw * m->dev.env_floats) * (int)sizeof(float);
    const int wg_per_cu = std::min(160 * 1024 / bytes, minw * 4 / w);  // LDS and register residency
    if (wg_per_cu < 1) continue;
    const long nwg = (nwaves + w - 1) / w, per_round = (long)wg_per_cu * ncu;
    const long full = nwg / per_round, rest = nwg % per_round;
    double cost = full * (1.0 + 0.05 * wg_per_cu * w);
    if (rest) cost += 1.0 + 0.05 * w * (double)((rest + ncu - 1) / ncu);
    if (cost <= best_cost + 1e-9) { best_cost = cost; best = w; }
  }
  if (cost_out) *cost_out = best_cost;
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
#if AMX_NO_GENERIC
#define AMX_LAUNCH_G(kernel, m, grid, stream, ...) return fail(AMX_EUNSUPPORTED, "this build of the library holds only the kernels of the model it was compiled for")
#else
#define AMX_LAUNCH_G(kernel, m, grid, stream, ...)                                                                       \
  do {                                                                                                                   \
    if ((m)->dev.G == 16) hipLaunchKernelGGL((kernel<16, DynSpec>), grid, dim3(AMX_WAVE * wpb), lds_bytes, (hipStream_t)stream, __VA_ARGS__);      \
    else if ((m)->dev.G == 32) hipLaunchKernelGGL((kernel<32, DynSpec>), grid, dim3(AMX_WAVE * wpb), lds_bytes, (hipStream_t)stream, __VA_ARGS__); \
    else hipLaunchKernelGGL((kernel<64, DynSpec>), grid, dim3(AMX_WAVE * wpb), lds_bytes, (hipStream_t)stream, __VA_ARGS__);                       \
  } while (0)
#endif

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

extern "C" int amx_step_to(const amx_model *m, int nenv, int nsub, const float *qpos, const float *qvel, const float *ctrl, const float *warm,
                           const float *time, float *qpos_out, float *qvel_out, float *warm_out, float *time_out, void *stream) {
  if (!m || !qpos || !qvel || !qpos_out || !qvel_out || (m->dev.nu > 0 && !ctrl) || nenv < 0 || nsub < 0) return fail(AMX_EINVAL, "amx_step: bad argument");
  if (nenv == 0) return AMX_OK;
  AMX_ON_DEVICE(m->device);
  dim3 grid;
  int wpb, lds_bytes;
  if (int rc = pre_launch(m, nenv, &grid, &wpb, &lds_bytes)) return rc;
  if (!for_spec(m->spec, [&](auto sp) {
        using S = decltype(sp);
        hipLaunchKernelGGL((amx_step_kernel<S::G, S>), grid, dim3(AMX_WAVE * wpb), lds_bytes, (hipStream_t)stream, m->dev, nenv, nsub, qpos, qvel, ctrl, warm, time,
                           qpos_out, qvel_out, warm_out, time_out);
      }))
    AMX_LAUNCH_G(amx_step_kernel, m, grid, stream, m->dev, nenv, nsub, qpos, qvel, ctrl, warm, time, qpos_out, qvel_out, warm_out, time_out);
  HIP_TRY(hipGetLastError());
  return AMX_OK;
}
extern "C" int amx_step(const amx_model *m, int nenv, int nsub, float *qpos, float *qvel, const float *ctrl, float *warm,
                        float *time, void *stream) {
  return amx_step_to(m, nenv, nsub, qpos, qvel, ctrl, warm, time, qpos, qvel, warm, time, stream);
}

extern "C" int amx_env_step(const amx_model *m, int nenv, int nsub, const float *qpos, const float *qvel, const float *action, const float *warm,
                            const float *time, const amx_env_tail_desc *tail, const amx_env_tail_io *io, void *stream) {
  if (!m || !tail || !io) return fail(AMX_EINVAL, "amx_env_step: null argument");
  if (tail->nobs < 0 || tail->nobs > AMX_TAIL_MAX_OBS || tail->nrew < 0 || tail->nrew > AMX_TAIL_MAX_REW || tail->episode_length < 1)
    return fail(AMX_EINVAL, "amx_env_step: descriptor out of range (nobs %d, nrew %d, episode_length %d)", tail->nobs, tail->nrew, tail->episode_length);
  for (int k = 0; k < tail->nobs; k++) {
    const int kind = tail->obs_kind[k], i = tail->obs_index[k];
    if (kind < 0 || kind > 3 || i < 0 || i >= (kind == 3 ? m->dev.nv : m->dev.nq)) return fail(AMX_EINVAL, "amx_env_step: observation term %d (kind %d, index %d) is out of range", k, kind, i);
  }
  for (int k = 0; k < tail->nrew; k++) {
    const int kind = tail->rew_kind[k], i = tail->rew_index[k];
    if (kind < 0 || kind > 3 || i < 0 || i >= (kind == 2 ? m->dev.nv : (kind == 3 ? m->dev.nu : m->dev.nq))) return fail(AMX_EINVAL, "amx_env_step: reward term %d (kind %d, index %d) is out of range", k, kind, i);
  }
  const void *need[] = {io->first_qpos, io->first_qvel, io->first_warm, io->first_time, io->first_obs, io->steps, io->episode_return, io->last_episode_return, io->episodes,
                        io->qpos_out, io->qvel_out, io->warm_out, io->time_out, io->qacc_out, io->obs_out, io->terminal_obs_out, io->reward_out, io->done_out,
                        io->truncation_out, io->steps_out, io->episode_return_out, io->last_episode_return_out, io->episodes_out};
  for (const void *p : need) if (!p) return fail(AMX_EINVAL, "amx_env_step: a required buffer of amx_env_tail_io is NULL");
  if (m->dev.nu > 0 && (!io->first_ctrl || !io->ctrl_out || !action)) return fail(AMX_EINVAL, "amx_env_step: ctrl buffers are NULL");
  if (!time) return fail(AMX_EINVAL, "amx_env_step: time_dev is NULL");
  // the physics: state in -> qpos_out, qvel_out, time_out and the solver's result (the next warm start) in qacc_out
  if (int rc = amx_step_to(m, nenv, nsub, qpos, qvel, action, warm, time, io->qpos_out, io->qvel_out, io->qacc_out, io->time_out, stream)) return rc;
  if (nenv == 0) return AMX_OK;
  AMX_ON_DEVICE(m->device);
  hipLaunchKernelGGL(amx_env_tail_kernel, dim3((nenv + 255) / 256), dim3(256), 0, (hipStream_t)stream, nenv, m->dev.nq, m->dev.nv, m->dev.nu, action, *tail, *io);
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
  if (m->wide && pick_plan(m, nenv) != m) return amx_rollout(m->wide, nenv, nsteps, x0, x0_batched, us, xs, xfinal, stream);
  AMX_ON_DEVICE(m->device);
  dim3 grid;
  int wpb, lds_bytes;
  if (int rc = pre_launch(m, nenv, &grid, &wpb, &lds_bytes)) return rc;
  float *prof = nullptr;
#ifdef AMX_PROFILE
  prof = g_prof;
#endif
  // large batches of a specialised small-slab model: the build sized for 3 waves per SIMD, when the residency model says the
  // third wave pays (fewer rounds outweigh the denser CU)
  bool three = false;
  if (m->spec >= 0 && !getenv("AMX_NO_W3")) {
    double c2 = 0, c3 = 0;
    (void)choose_wpb(m, nenv, 2, &c2);
    const int w3 = choose_wpb(m, nenv, 3, &c3);
    const char *f3 = getenv("AMX_FORCE_W3");
    if (f3 ? atoi(f3) != 0 : c3 < c2 - 1e-6) {  // measured: the 168-register build is never slower at equal residency, faster whenever it saves a round
      three = true;
      wpb = w3;
      const int per_block = wpb * (AMX_WAVE / m->dev.G);
      lds_bytes = (m->dev.nconst + per_block * m->dev.env_floats) * (int)sizeof(float);
      grid = dim3((nenv + per_block - 1) / per_block);
    }
  }
  if (!for_spec(m->spec, [&](auto sp) {
        using S = decltype(sp);
        if (three) hipLaunchKernelGGL((amx_rollout_kernel<S::G, S, 3>), grid, dim3(AMX_WAVE * wpb), lds_bytes, (hipStream_t)stream, m->dev, nenv, nsteps, x0_batched, x0, us, xs, xfinal, prof);
        else hipLaunchKernelGGL((amx_rollout_kernel<S::G, S>), grid, dim3(AMX_WAVE * wpb), lds_bytes, (hipStream_t)stream, m->dev, nenv, nsteps, x0_batched, x0, us, xs, xfinal, prof);
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
  if (int rc = amx_rollout(pick_plan(m, nsamples), nsamples, nsteps, x0, /*x0_batched=*/0, us, /*xs=*/nullptr, xf, stream)) return rc;
  hipLaunchKernelGGL(amx_fd_cost_kernel, dim3(nsamples), dim3(AMX_WAVE), 0, (hipStream_t)stream, nsamples, nsteps, nx, nu, coord0, xf, Qf, xg, eps, J, grad);
  HIP_TRY(hipGetLastError());
  if (xfinal) HIP_TRY(hipMemcpyAsync(xfinal, xf, (size_t)nsamples * nx * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return AMX_OK;
}
