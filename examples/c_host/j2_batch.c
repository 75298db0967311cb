/* A plain C host driving the constitutive update through include/dxmat.h: no Python, no torch.
 *
 *   make -C examples/c_host && ./examples/c_host/j2_batch [npoints]
 *
 * Replays what QuadratureMap.update() / advance() do around material.integrate()
 * (dolfinx_materials/quadrature_map.py:297-360) for a J2 law with linear hardening on a batch of
 * Gauss points under proportional uniaxial-strain loading, and checks the result against the closed
 * form of the radial return (tests/mfront/IsotropicLinearHardeningPlasticity.mfront:49-77) on the host. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "dxmat.h"

#define CHECK(call)                                                                  \
  do {                                                                               \
    int rc_ = (call);                                                                \
    if (rc_ < 0) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, dxm_last_error()); return 1; } \
  } while (0)

int main(int argc, char** argv) {
  const long long n = argc > 1 ? atoll(argv[1]) : 100000;
  const double E = 70e3, nu = 0.3, sig0 = 250.0, H = 5e3;
  const double params[4] = {E, nu, sig0, H};
  if (dxm_device_count() < 1) { fprintf(stderr, "no HIP device: %s\n", dxm_last_error()); return 2; }
  dxm_material* m = dxm_create(DXM_LAW_J2_LINEAR, params, 4, n, 0);
  if (!m) { fprintf(stderr, "dxm_create: %s\n", dxm_last_error()); return 1; }

  double* eps = (double*)dxm_host_alloc(sizeof(double) * n * 6);   /* page-locked: full PCIe rate */
  double* sig = (double*)dxm_host_alloc(sizeof(double) * n * 6);
  double* isv = (double*)dxm_host_alloc(sizeof(double) * n * 7);
  double* ct = (double*)dxm_host_alloc(sizeof(double) * n * 36);
  if (!eps || !sig || !isv || !ct) { fprintf(stderr, "dxm_host_alloc failed\n"); return 1; }

  const double lam = E * nu / ((1 + nu) * (1 - 2 * nu)), mu = E / (2 * (1 + nu));
  double worst = 0.0;
  dxm_stats st;
  for (int step = 1; step <= 4; ++step) {
    const double exx = 2e-3 * step;   /* uniaxial strain eps_xx, every point scaled a little differently */
    for (long long i = 0; i < n; ++i) {
      double* e = eps + 6 * i;
      e[0] = exx * (1.0 + 1e-3 * (double)(i % 100));
      e[1] = e[2] = e[3] = e[4] = e[5] = 0.0;
    }
    CHECK(dxm_integrate(m, eps, 0.0, sig, isv, ct, &st));   /* QuadratureMap.update()  */
    /* closed form for monotonic uniaxial strain from a virgin state: s_trial = 2 mu dev(eps), seq = 2 mu exx,
     * p = (seq - sig0) / (3 mu + H), sigma = lambda tr(eps) 1 + 2 mu (eps - p n), n = sqrt(3/2)-normalised */
    for (long long i = 0; i < n; i += (n > 1000 ? n / 1000 : 1)) {
      const double ex = eps[6 * i];
      const double seq = 2.0 * mu * ex;
      const double p = seq > sig0 ? (seq - sig0) / (3.0 * mu + H) : 0.0;
      const double sxx = lam * ex + 2.0 * mu * (ex - p);          /* n_xx = 1 for this direction (3/2 * 2/3) */
      const double err = fabs(sig[6 * i] - sxx) / fabs(sxx);
      const double perr = fabs(isv[7 * i] - p);
      if (err > worst) worst = err;
      if (perr > worst) worst = perr;
    }
    printf("step %d  eps_xx = %.4f  sigma_xx[0] = %10.4f  p[0] = %.6f  plastic points = %lld  Ct_00[0] = %.2f\n", step,
           eps[0], sig[0], isv[0], (long long)st.n_plastic, ct[0]);
    CHECK(dxm_advance(m));                                    /* QuadratureMap.advance() */
  }
  printf("largest deviation from the closed form: %.3e  (kernel %s)\n", worst, dxm_kernel_name(m));

  /* A map over a SUBSET of the cells (quadrature_map.py:66-73 with `cells`): the same points are rows 2 i + 1 of fields twice
   * as long, in ordinary malloc memory; dxm_integrate_rows delivers stress and tangent block of point i into that row.
   * Same update once more from the same initial state: the rows must equal what dxm_integrate returns, bit for bit. */
  long long* rows = (long long*)malloc(sizeof(long long) * (size_t)n);
  double* sig2 = (double*)malloc(sizeof(double) * (size_t)n * 2 * 6);
  double* ct2 = (double*)malloc(sizeof(double) * (size_t)n * 2 * 36);
  if (!rows || !sig2 || !ct2) { fprintf(stderr, "malloc failed\n"); return 1; }
  for (long long i = 0; i < n; ++i) rows[i] = 2 * i + 1;
  for (long long k = 0; k < n * 2 * 6; ++k) sig2[k] = -1.0;
  for (long long k = 0; k < n * 2 * 36; ++k) ct2[k] = -1.0;
  CHECK(dxm_integrate(m, eps, 0.0, sig, NULL, ct, &st));
  CHECK(dxm_integrate_rows(m, eps, 0.0, sig2, ct2, (const int64_t*)rows, &st));
  long long differing = 0;
  for (long long i = 0; i < n; ++i) {
    for (int c = 0; c < 6; ++c) differing += sig2[(2 * i + 1) * 6 + c] != sig[i * 6 + c] || sig2[2 * i * 6 + c] != -1.0;
    for (int c = 0; c < 36; ++c) differing += ct2[(2 * i + 1) * 36 + c] != ct[i * 36 + c] || ct2[2 * i * 36 + c] != -1.0;
  }
  printf("dxm_integrate_rows into every second row of fields over 2 x %lld points: %lld entries differ\n", n, differing);
  free(rows); free(sig2); free(ct2);
  dxm_host_free(eps); dxm_host_free(sig); dxm_host_free(isv); dxm_host_free(ct);
  dxm_destroy(m);
  return worst < 1e-10 && differing == 0 ? 0 : 3;
}
