/* A C host of the drop-in boundary (include/amx.h), no Python: uploads the pendulum (examples/c_host/pendulum_desc.h, written by
 * tools/dump_desc_c.py from ambersim_amd/models/pendulum/pendulum.xml), rolls 64 environments out for 100 steps with
 * amx_rollout and checks every state against the closed form of the reference's pendulum (M = 0.337959,
 * m g l = 4.905, torque clamped to +-2, semi-implicit Euler, dt = 0.02).
 *
 * Build (the library is linked with -no-hip-rt, so the host brings the HIP runtime):
 *   make -C examples/c_host        # gcc + -lamx -lamdhip64
 * Run on a GPU box: examples/c_host/pendulum_host */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include "amx.h"
#include "pendulum_desc.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)
#define AK(x) do { int rc_ = (x); if (rc_ != AMX_OK) { fprintf(stderr, "amx error %d: %s (line %d)\n", rc_, amx_last_error(), __LINE__); return 3; } } while (0)

int main(void) {
  enum { NENV = 64, N = 100, NX = 2, NU = 1 };
  amx_model_desc desc = {0};
  pendulum_fill_desc(&desc);
  amx_model *model = NULL;
  AK(amx_model_create(&desc, 0, &model));
  int lanes = 0;
  AK(amx_model_query(model, "lanes_per_env", &lanes));

  static float x0[NENV][NX], us[NENV][N][NU], xs[NENV][N + 1][NX];
  unsigned s = 12345u;
  for (int e = 0; e < NENV; ++e) {
    s = s * 1664525u + 1013904223u; x0[e][0] = -3.14159f + 6.28318f * (float)(s >> 8) / 16777216.0f;
    s = s * 1664525u + 1013904223u; x0[e][1] = -2.0f + 4.0f * (float)(s >> 8) / 16777216.0f;
    for (int t = 0; t < N; ++t) { s = s * 1664525u + 1013904223u; us[e][t][0] = -3.0f + 6.0f * (float)(s >> 8) / 16777216.0f; }  /* beyond the +-2 clamp on purpose */
  }
  float *x0_d, *us_d, *xs_d;
  CK(hipMalloc((void **)&x0_d, sizeof x0)); CK(hipMalloc((void **)&us_d, sizeof us)); CK(hipMalloc((void **)&xs_d, sizeof xs));
  CK(hipMemcpy(x0_d, x0, sizeof x0, hipMemcpyHostToDevice)); CK(hipMemcpy(us_d, us, sizeof us, hipMemcpyHostToDevice));
  AK(amx_rollout(model, NENV, N, x0_d, 1, us_d, xs_d, NULL, NULL));
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(xs, xs_d, sizeof xs, hipMemcpyDeviceToHost));

  double worst = 0.0;
  for (int e = 0; e < NENV; ++e) {
    double th = x0[e][0], om = x0[e][1];
    for (int t = 0; t < N; ++t) {
      double u = us[e][t][0];
      u = u > 2.0 ? 2.0 : (u < -2.0 ? -2.0 : u);
      const double acc = (u - 4.905 * sin(th)) / 0.337959;
      om += 0.02 * acc; th += 0.02 * om;
      const double e0 = fabs(xs[e][t + 1][0] - th), e1 = fabs(xs[e][t + 1][1] - om);
      const double sc = fmax(1.0, fmax(fabs(th), fabs(om)));
      if (e0 / sc > worst) worst = e0 / sc;
      if (e1 / sc > worst) worst = e1 / sc;
    }
  }
  printf("pendulum, %d envs x %d steps through the C ABI (%d lanes per env): max relative error vs closed form %.3e\n", NENV, N, lanes, worst);
  AK(amx_model_destroy(model));
  hipFree(x0_d); hipFree(us_d); hipFree(xs_d);
  return worst < 3e-5 ? 0 : 1;  /* fp32 drift over 100 steps, same bound as tests/test_gpu_parity.py::test_pendulum_c1_closed_form */
}
