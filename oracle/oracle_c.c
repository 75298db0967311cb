/*
 * Plain-C restatement of the per-Gauss-point constitutive updates (CPU oracle).
 *
 * TEST INFRASTRUCTURE ONLY: used by tests/ and by the `cpu_baseline` leg of bench.py ("port").
 * The product (dolfinx_materials_amd/, libdxmat.so) never links or calls this file.
 *
 * Follows, line by line where the reference has a line to follow:
 *   python_materials/elasticity.py:12-24                          (isotropic elasticity)
 *   tests/mfront/IsotropicLinearHardeningPlasticity.mfront:49-77  (radial return + tangent)
 *   tests/test_FeFp_jax.py:14-15                                  (Voce law)
 * Layout: AoS row-major (n, dim) fp64 everywhere, like the reference's (N, dim) numpy arrays
 * (generic.py:219-240).  Scalar code, one point at a time (the reference's own CPU path is a
 * Python loop over points: generic.py:77-79); `nthreads` > 1 splits the range with OpenMP.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_MAXIT 25

static void lame(double E, double nu, double* lambda, double* mu) {
  *lambda = E * nu / (1 + nu) / (1 - 2 * nu); /* elasticity.py:12-13 */
  *mu = E / 2 / (1 + nu);
}

/* law 0: sigma = C eps, Ct = C  (elasticity.py:15-24) */
void orc_elastic_iso(int64_t n, const double* eps, double E, double nu, double* sig, double* ct,
                     int nthreads) {
  double lambda, mu;
  lame(E, nu, &lambda, &mu);
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads) schedule(static)
#endif
  for (int64_t p = 0; p < n; ++p) {
    const double* e = eps + 6 * p;
    double* s = sig + 6 * p;
    double* c = ct + 36 * p;
    const double ltr = lambda * (e[0] + e[1] + e[2]);
    for (int i = 0; i < 6; ++i) s[i] = (i < 3 ? ltr : 0.0) + 2 * mu * e[i];
    for (int i = 0; i < 6; ++i)
      for (int j = 0; j < 6; ++j)
        c[6 * i + j] = ((i < 3 && j < 3) ? lambda : 0.0) + ((i == j) ? 2 * mu : 0.0);
  }
  (void)nthreads;
}

/* hardening: kind 0 linear R = s0 + H p (h1 = H); kind 1 Voce R = s0 + (su-s0)(1-exp(-b p)) */
static double hard_R(int kind, double s0, double h1, double h2, double p) {
  return kind == 0 ? s0 + h1 * p : s0 + (h1 - s0) * (1.0 - exp(-h2 * p));
}
static double hard_dR(int kind, double s0, double h1, double h2, double p) {
  return kind == 0 ? h1 : (h1 - s0) * h2 * exp(-h2 * p);
}

/* laws 1/2: small-strain J2 with isotropic hardening, (eps_p, p)-state form of mfront:49-77.
 * Returns the number of points whose local Newton did not converge. */
int64_t orc_j2(int64_t n, const double* eps, const double* epsp_n, const double* p_n, double E,
               double nu, int kind, double s0, double h1, double h2, double rtol, double* sig,
               double* epsp, double* p_out, double* ct, int64_t* n_plastic, int nthreads) {
  double lambda, mu;
  lame(E, nu, &lambda, &mu);
  int64_t notconv = 0, nplast = 0;
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads) schedule(static) reduction(+ : notconv, nplast)
#endif
  for (int64_t q = 0; q < n; ++q) {
    double e[6], se[6], nn[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 6; ++i) e[i] = eps[6 * q + i] - epsp_n[6 * q + i]; /* mfront:52 */
    const double tr = e[0] + e[1] + e[2];
    for (int i = 0; i < 6; ++i) se[i] = 2 * mu * (e[i] - (i < 3 ? tr / 3.0 : 0.0)); /* :53 */
    double nrm2 = 0;
    for (int i = 0; i < 6; ++i) nrm2 += se[i] * se[i];
    const double seq = sqrt(1.5 * nrm2); /* :54 */
    const double pn = p_n[q];
    double c1 = lambda, c2 = 2 * mu, c3 = 0.0, dp = 0.0;
    if (seq - hard_R(kind, s0, h1, h2, pn) > 0.0) { /* :55 */
      ++nplast;
      if (kind == 0) {
        dp = (seq - s0 - h1 * pn) / (h1 + 3 * mu); /* :62-63 */
      } else {
        for (int it = 0;; ++it) {
          const double r = seq - 3 * mu * dp - hard_R(kind, s0, h1, h2, pn + dp);
          if (fabs(r) <= fmax(rtol * fmax(fabs(s0), 2e-8 * mu), rtol * seq)) break;
          if (it >= ORC_MAXIT) { ++notconv; break; }
          dp -= r / (-3 * mu - hard_dR(kind, s0, h1, h2, pn + dp));
        }
      }
      for (int i = 0; i < 6; ++i) nn[i] = 1.5 * se[i] / seq; /* :61 */
      const double beta = dp / seq;
      const double gamma = 1.0 / (hard_dR(kind, s0, h1, h2, pn + dp) + 3 * mu);
      c1 = lambda + 2 * mu * mu * beta; /* :66-69 with M = 3/2 Id - 1/2 IxI */
      c2 = 2 * mu - 6 * mu * mu * beta;
      c3 = 4 * mu * mu * (beta - gamma);
    }
    for (int i = 0; i < 6; ++i) {
      epsp[6 * q + i] = epsp_n[6 * q + i] + dp * nn[i];
      e[i] -= dp * nn[i]; /* :64 */
    }
    p_out[q] = pn + dp;
    const double ltr = lambda * (e[0] + e[1] + e[2]);
    for (int i = 0; i < 6; ++i) sig[6 * q + i] = (i < 3 ? ltr : 0.0) + 2 * mu * e[i]; /* :76 */
    double* c = ct + 36 * q;
    for (int i = 0; i < 6; ++i)
      for (int j = 0; j < 6; ++j)
        c[6 * i + j] = ((i < 3 && j < 3) ? c1 : 0.0) + ((i == j) ? c2 : 0.0) + (c3 * nn[i]) * nn[j];
  }
  if (n_plastic) *n_plastic = nplast;
  (void)nthreads;
  return notconv;
}

int orc_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* ------------------------------------------------------------------------------------------
 * laws 3/4: finite-strain FeFp J2 plasticity, Voce (kind 1: su, b) or linear (kind 0: su = H) hardening (PARITY UNPINNED: the reference
 * only fixes the interface, jaxmat.py:170-186 / tests/test_FeFp_jax.py:7-31; the algorithm is the
 * build's own choice, restated from oracle/constitutive_np.py::fefp_update -- same 2x2 Newton in
 * (dp, Ie), tangent by nine hand-written JVPs through the algorithm).
 * F9, P9 in the order [11,22,33,12,21,13,31,23,32] (utils.py:168-190); Mandel 6-vectors for
 * cpinv (isochoric Cp^-1) and be_bar.
 * ------------------------------------------------------------------------------------------ */
static const int NS_I[9] = {0, 1, 2, 0, 1, 0, 2, 1, 2};
static const int NS_J[9] = {0, 1, 2, 1, 0, 2, 0, 2, 1};

static double det3(const double A[3][3]) {
  return A[0][0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1]) - A[0][1] * (A[1][0] * A[2][2] - A[1][2] * A[2][0]) +
         A[0][2] * (A[1][0] * A[2][1] - A[1][1] * A[2][0]);
}
static void cof3(const double A[3][3], double C[3][3]) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      C[i][j] = A[i1][j1] * A[i2][j2] - A[i1][j2] * A[i2][j1];
    }
}
static void mm3(const double A[3][3], const double B[3][3], double C[3][3]) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) C[i][j] = A[i][0] * B[0][j] + A[i][1] * B[1][j] + A[i][2] * B[2][j];
}
static void mm3t(const double A[3][3], const double B[3][3], double C[3][3]) { /* A B^T */
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) C[i][j] = A[i][0] * B[j][0] + A[i][1] * B[j][1] + A[i][2] * B[j][2];
}
static void mandel_to_t(const double* v, double T[3][3]) {
  const double r = 0.70710678118654752440;
  T[0][0] = v[0]; T[1][1] = v[1]; T[2][2] = v[2];
  T[0][1] = T[1][0] = v[3] * r; T[0][2] = T[2][0] = v[4] * r; T[1][2] = T[2][1] = v[5] * r;
}
static void t_to_mandel(const double T[3][3], double* v) {
  const double s = 1.4142135623730950488;
  v[0] = T[0][0]; v[1] = T[1][1]; v[2] = T[2][2];
  v[3] = s * 0.5 * (T[0][1] + T[1][0]); v[4] = s * 0.5 * (T[0][2] + T[2][0]); v[5] = s * 0.5 * (T[1][2] + T[2][1]);
}

int64_t orc_fefp(int64_t n, const double* F9, const double* cpinv_n, const double* p_n, double E, double nu,
                 int kind, double s0, double su, double b, double rtol, double* P9, double* be_bar, double* cpinv,
                 double* p_out, double* ct, int64_t* n_plastic, int nthreads) {
  double lambda, mu;
  lame(E, nu, &lambda, &mu);
  const double kappa = lambda + 2 * mu / 3;
  const double SQ32 = sqrt(1.5), SQ23 = sqrt(2.0 / 3.0), SQ6 = sqrt(6.0);
  int64_t notconv = 0, nplast = 0;
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads) schedule(static) reduction(+ : notconv, nplast)
#endif
  for (int64_t q = 0; q < n; ++q) {
    double F[3][3], G[3][3], Fi[3][3], cf[3][3], GFt[3][3], btr[3][3], d[3][3], sh[3][3] = {{0}}, be[3][3];
    for (int t = 0; t < 9; ++t) F[NS_I[t]][NS_J[t]] = F9[9 * q + t];
    mandel_to_t(cpinv_n + 6 * q, G);
    const double J = det3(F);
    cof3(F, cf);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) Fi[i][j] = cf[j][i] / J;
    const double Jm23 = pow(J, -2.0 / 3.0);
    mm3t(G, F, GFt);
    mm3(F, GFt, btr);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) btr[i][j] *= Jm23;
    const double Itr = (btr[0][0] + btr[1][1] + btr[2][2]) / 3.0;
    double atr2 = 0;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) { d[i][j] = btr[i][j] - (i == j ? Itr : 0.0); atr2 += d[i][j] * d[i][j]; }
    const double atr = sqrt(atr2);
    const double pn = p_n[q];
    const int plastic = SQ32 * mu * atr - hard_R(kind, s0, su, b, pn) > 0.0;
    double dp = 0, Ie = Itr, a = atr, delta = 0;
    if (plastic) {
      ++nplast;
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) sh[i][j] = d[i][j] / atr;
      delta = det3(sh);
      const double tol1 = fmax(rtol * fmax(fabs(s0), 2e-8 * mu), rtol * (SQ32 * mu * atr));
      for (int it = 0;; ++it) {
        const double aa = SQ23 * hard_R(kind, s0, su, b, pn + dp) / mu;
        const double r1 = atr - aa - SQ6 * dp * Ie;
        const double r2 = Ie * Ie * Ie - 0.5 * aa * aa * Ie + aa * aa * aa * delta - 1.0;
        if (fabs(SQ32 * mu * r1) <= tol1 && fabs(r2) <= 1e-14) break;
        if (it >= ORC_MAXIT) { ++notconv; break; }
        const double ap = SQ23 * hard_dR(kind, s0, su, b, pn + dp) / mu;
        const double j11 = -ap - SQ6 * Ie, j12 = -SQ6 * dp;
        const double j21 = (-aa * Ie + 3 * aa * aa * delta) * ap, j22 = 3 * Ie * Ie - 0.5 * aa * aa;
        const double det = j11 * j22 - j12 * j21;
        const double ddp = (-r1 * j22 + r2 * j12) / det, dIe = (-j11 * r2 + j21 * r1) / det;
        dp += ddp;
        Ie += dIe;
      }
      a = SQ23 * hard_R(kind, s0, su, b, pn + dp) / mu;
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) be[i][j] = (i == j ? Ie : 0.0) + a * sh[i][j];
    } else {
      memcpy(be, btr, sizeof(be));
    }
    const double Ib = (be[0][0] + be[1][1] + be[2][2]) / 3.0;
    double tau[3][3], P[3][3], t1[3][3], gn[3][3];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) tau[i][j] = mu * (be[i][j] - (i == j ? Ib : 0.0)) + (i == j ? 0.5 * kappa * (J * J - 1.0) : 0.0);
    mm3t(tau, Fi, P);
    for (int t = 0; t < 9; ++t) P9[9 * q + t] = P[NS_I[t]][NS_J[t]];
    p_out[q] = pn + dp;
    t_to_mandel(be, be_bar + 6 * q);
    mm3t(be, Fi, t1);
    mm3(Fi, t1, gn);
    const double J23 = pow(J, 2.0 / 3.0);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) gn[i][j] *= J23;
    t_to_mandel(gn, cpinv + 6 * q);
    if (!ct) continue;
    /* tangent: column (k,l) = derivative of the algorithm in the direction dF = e_k (x) e_l */
    const double p1 = pn + dp;
    const double ap = SQ23 * hard_dR(kind, s0, su, b, p1) / mu;
    double cs[3][3];
    cof3(sh, cs);
    const double gI = 3 * Ie * Ie - 0.5 * a * a;
    const double dIe_da = (a * Ie - 3 * a * a * delta) / gI, dIe_dd = -(a * a * a) / gI;
    const double r_dp = -ap - SQ6 * Ie - SQ6 * dp * dIe_da * ap, r_dd = -SQ6 * dp * dIe_dd;
    for (int col = 0; col < 9; ++col) {
      const int k = NS_I[col], l = NS_J[col];
      const double trFidF = Fi[l][k];
      const double dJ = J * trFidF;
      double dB[3][3] = {{0}}, dd[3][3], ds[3][3], dtau[3][3], dP[3][3];
      for (int j = 0; j < 3; ++j) { dB[k][j] += GFt[l][j]; dB[j][k] += GFt[l][j]; }
      double dItr = 0;
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) { dd[i][j] = Jm23 * dB[i][j] - (2.0 / 3.0) * trFidF * btr[i][j]; if (i == j) dItr += dd[i][j] / 3.0; }
      for (int i = 0; i < 3; ++i) dd[i][i] -= dItr;
      if (plastic) {
        double datr = 0, ddel = 0, dsh[3][3];
        for (int i = 0; i < 3; ++i)
          for (int j = 0; j < 3; ++j) datr += sh[i][j] * dd[i][j];
        for (int i = 0; i < 3; ++i)
          for (int j = 0; j < 3; ++j) { dsh[i][j] = (dd[i][j] - sh[i][j] * datr) / atr; ddel += cs[i][j] * dsh[i][j]; }
        const double ddp = -(datr + r_dd * ddel) / r_dp, da = ap * ddp;
        for (int i = 0; i < 3; ++i)
          for (int j = 0; j < 3; ++j) ds[i][j] = mu * (da * sh[i][j] + a * dsh[i][j]);
      } else {
        for (int i = 0; i < 3; ++i)
          for (int j = 0; j < 3; ++j) ds[i][j] = mu * dd[i][j];
      }
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) dtau[i][j] = ds[i][j] + (i == j ? kappa * J * dJ : 0.0);
      mm3t(dtau, Fi, dP);
      /* - P dF^T F^-T : (P dF^T)[i][m] = P[i][l] delta_mk  ->  -P[i][l] Fi[j][k] */
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) dP[i][j] -= P[i][l] * Fi[j][k];
      for (int t = 0; t < 9; ++t) ct[81 * q + 9 * t + col] = dP[NS_I[t]][NS_J[t]];
    }
  }
  if (n_plastic) *n_plastic = nplast;
  (void)nthreads;
  return notconv;
}
