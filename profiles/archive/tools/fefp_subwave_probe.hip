// Sub-wave probe for the FeFp return mapping (north_star: "one wavefront (or sub-wave) per Gauss point ... wavefront shuffles
// for the small dense solve"; SURVEY.md 2.2: "evaluate 8- or 16-lane sub-wave variant against 1 thread/pt").
//
// The shipped fefp_kernel already IS sub-wave where the work is wide: its tangent epilogue maps lane = (point slot, column), 9
// lanes per point, from LDS-staged records.  What it keeps at one thread per point is the return mapping (trial state, 2x2
// Newton, PK1, new state).  This file measures that choice: the SAME return mapping, stress and state update, no tangent,
//   rm_thread_kernel   one thread per point (the shipped mapping; plain AoS loads, no LDS),
//   rm_subwave_kernel  nine lanes per point, lane (i, j) owns entry (i, j) of every 3x3 matrix, cross-lane reads by wavefront
//                      shuffles (ds_bpermute), scalars replicated; 7 points per wave, lane 63 idle,
// on identical inputs, outputs compared by tools/fefp_subwave_probe.py, counters by rocprofv3 --pmc.
// Measurement infrastructure: not part of libdxmat.
#include <hip/hip_runtime.h>
#include <stdint.h>

struct Prm { double mu, kappa, sig0, sigu, b, tol, rtol; int maxit; };

__device__ __forceinline__ void voce(const Prm& q, double p, double& R, double& dR) {
  const double ex = exp(-q.b * p);
  R = q.sig0 + (q.sigu - q.sig0) * (1.0 - ex);
  dR = (q.sigu - q.sig0) * q.b * ex;
}

// the scalar part both mappings share: 2x2 Newton on (dp, Ie)  (fefp.hpp step 3)
__device__ __forceinline__ void return_scalars(const Prm& q, double atr, double delta, double Itr, double p_n, double R_n, double dR_n,
                                               double& dp, double& Ie, double& R_1) {
  const double SQ32 = 1.2247448713915890491, SQ23 = 0.81649658092772603273, SQ6 = 2.4494897427831780982;
  const double imu = 1.0 / q.mu;
  dp = 0.0;
  Ie = Itr;
  double R_k = R_n, dR_k = dR_n;
  const double tol1 = fmax(q.tol, q.rtol * (SQ32 * q.mu * atr));
  for (int it = 0;; ++it) {
    const double aa = SQ23 * R_k * imu;
    const double r1 = atr - aa - SQ6 * dp * Ie;
    const double r2 = Ie * Ie * Ie - 0.5 * aa * aa * Ie + aa * aa * aa * delta - 1.0;
    if (fabs(SQ32 * q.mu * r1) <= tol1 && fabs(r2) <= 1e-14) break;
    if (it >= q.maxit) break;
    const double ap = SQ23 * dR_k * imu;
    const double j11 = -ap - SQ6 * Ie, j12 = -SQ6 * dp;
    const double j21 = (-aa * Ie + 3.0 * aa * aa * delta) * ap, j22 = 3.0 * Ie * Ie - 0.5 * aa * aa;
    const double idet = 1.0 / (j11 * j22 - j12 * j21);
    dp += (-r1 * j22 + r2 * j12) * idet;
    Ie += (-j11 * r2 + j21 * r1) * idet;
    voce(q, p_n + dp, R_k, dR_k);
  }
  R_1 = R_k;
}

#define SYM(i, j) ((i) == (j) ? (i) : ((i) + (j) + 2))

// ---- one thread per point ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) rm_thread_kernel(const Prm q, int64_t n, const double* __restrict__ Fin, const double* __restrict__ s0,
                                                        double* __restrict__ s1, int64_t ld, double* __restrict__ Pout) {
  const double SQ32 = 1.2247448713915890491, SQ23 = 0.81649658092772603273, RS2 = 0.70710678118654752440, SQ2 = 1.4142135623730950488;
  for (int64_t gi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; gi < n; gi += (int64_t)gridDim.x * blockDim.x) {
    const double* f = Fin + gi * 9;
    double F[9];
    F[0] = f[0]; F[4] = f[1]; F[8] = f[2]; F[1] = f[3]; F[3] = f[4]; F[2] = f[5]; F[6] = f[6]; F[5] = f[7]; F[7] = f[8];
    const double p_n = s0[gi];
    double G[6];
    for (int c = 0; c < 6; ++c) G[c] = s0[(int64_t)(7 + c) * ld + gi] * (c < 3 ? 1.0 : RS2);
    double cf[9];
    cf[0] = F[4] * F[8] - F[5] * F[7]; cf[1] = F[5] * F[6] - F[3] * F[8]; cf[2] = F[3] * F[7] - F[4] * F[6];
    cf[3] = F[7] * F[2] - F[8] * F[1]; cf[4] = F[8] * F[0] - F[6] * F[2]; cf[5] = F[6] * F[1] - F[7] * F[0];
    cf[6] = F[1] * F[5] - F[2] * F[4]; cf[7] = F[2] * F[3] - F[0] * F[5]; cf[8] = F[0] * F[4] - F[1] * F[3];
    const double J = F[0] * cf[0] + F[1] * cf[1] + F[2] * cf[2];
    const double iJ = 1.0 / J;
    double Fi[9];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) Fi[r * 3 + c] = cf[c * 3 + r] * iJ;
    const double J23 = cbrt(J * J), Jm23 = 1.0 / J23;
    double h[9];
    for (int L = 0; L < 3; ++L)
      for (int m = 0; m < 3; ++m) h[L * 3 + m] = Jm23 * (G[SYM(L, 0)] * F[m * 3] + G[SYM(L, 1)] * F[m * 3 + 1] + G[SYM(L, 2)] * F[m * 3 + 2]);
    double d[6];
    {
      const int SI[6] = {0, 1, 2, 0, 0, 1}, SJ[6] = {0, 1, 2, 1, 2, 2};
      for (int t = 0; t < 6; ++t) d[t] = F[SI[t] * 3] * h[SJ[t]] + F[SI[t] * 3 + 1] * h[3 + SJ[t]] + F[SI[t] * 3 + 2] * h[6 + SJ[t]];
    }
    const double Itr = (d[0] + d[1] + d[2]) / 3.0;
    d[0] -= Itr; d[1] -= Itr; d[2] -= Itr;
    const double atr = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + 2.0 * (d[3] * d[3] + d[4] * d[4] + d[5] * d[5]));
    double R_n, dR_n;
    voce(q, p_n, R_n, dR_n);
    double dp = 0.0, Ie = Itr, sdev[6];
    if (SQ32 * q.mu * atr - R_n > 0.0) {
      double sh[6], cs0, cs3, cs4;
      const double iatr = 1.0 / atr;
      for (int k = 0; k < 6; ++k) sh[k] = d[k] * iatr;
      cs0 = sh[1] * sh[2] - sh[5] * sh[5]; cs3 = sh[4] * sh[5] - sh[3] * sh[2]; cs4 = sh[3] * sh[5] - sh[4] * sh[1];
      const double delta = sh[0] * cs0 + sh[3] * cs3 + sh[4] * cs4;
      double R_1;
      return_scalars(q, atr, delta, Itr, p_n, R_n, dR_n, dp, Ie, R_1);
      const double a = SQ23 * R_1 / q.mu;
      for (int k = 0; k < 6; ++k) sdev[k] = a * sh[k];
    } else {
      for (int k = 0; k < 6; ++k) sdev[k] = d[k];
    }
    const double pr = 0.5 * q.kappa * (J * J - 1.0);
    double tau[6], P[9];
    for (int k = 0; k < 6; ++k) tau[k] = q.mu * sdev[k];
    tau[0] += pr; tau[1] += pr; tau[2] += pr;
    for (int i = 0; i < 3; ++i)
      for (int Jx = 0; Jx < 3; ++Jx) P[i * 3 + Jx] = tau[SYM(i, 0)] * Fi[Jx * 3] + tau[SYM(i, 1)] * Fi[Jx * 3 + 1] + tau[SYM(i, 2)] * Fi[Jx * 3 + 2];
    double gn[6];
    {
      double t[9];
      const double cI = Ie - pr / q.mu;
      for (int i = 0; i < 3; ++i)
        for (int Jx = 0; Jx < 3; ++Jx) t[i * 3 + Jx] = P[i * 3 + Jx] / q.mu + cI * Fi[Jx * 3 + i];
      const int SI[6] = {0, 1, 2, 0, 0, 1}, SJ[6] = {0, 1, 2, 1, 2, 2};
      for (int k = 0; k < 6; ++k) gn[k] = J23 * (Fi[SI[k] * 3] * t[SJ[k]] + Fi[SI[k] * 3 + 1] * t[3 + SJ[k]] + Fi[SI[k] * 3 + 2] * t[6 + SJ[k]]);
    }
    double* po = Pout + gi * 9;
    po[0] = P[0]; po[1] = P[4]; po[2] = P[8]; po[3] = P[1]; po[4] = P[3]; po[5] = P[2]; po[6] = P[6]; po[7] = P[5]; po[8] = P[7];
    s1[gi] = p_n + dp;
    for (int c = 0; c < 6; ++c) {
      s1[(int64_t)(1 + c) * ld + gi] = (c < 3 ? sdev[c] + Ie : SQ2 * sdev[c]);
      s1[(int64_t)(7 + c) * ld + gi] = (c < 3 ? gn[c] : SQ2 * gn[c]);
    }
  }
}

// ---- nine lanes per point ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ double shf(double v, int lane) { return __shfl(v, lane, 64); }

__global__ void __launch_bounds__(256) rm_subwave_kernel(const Prm q, int64_t n, const double* __restrict__ Fin, const double* __restrict__ s0,
                                                         double* __restrict__ s1, int64_t ld, double* __restrict__ Pout) {
  const double SQ32 = 1.2247448713915890491, SQ23 = 0.81649658092772603273, RS2 = 0.70710678118654752440, SQ2 = 1.4142135623730950488;
  const int lane = threadIdx.x & 63;
  const int g = lane / 9, c = lane - g * 9;      // group (point slot) 0..6 (7 = the idle lane 63), entry
  const int i = c / 3, j = c - i * 3;
  const int gb = g * 9;                           // first lane of my group
  const bool live = lane < 63;
  // position of entry (i, j) in the 9-vector [11,22,33,12,21,13,31,23,32]
  const int pos9 = (i == j) ? i : ((i == 0 && j == 1) ? 3 : (i == 1 && j == 0) ? 4 : (i == 0 && j == 2) ? 5 : (i == 2 && j == 0) ? 6 : (i == 1 && j == 2) ? 7 : 8);
  const int sym = SYM(i, j);                      // slot of (i, j) in [xx,yy,zz,xy,xz,yz]
  const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int64_t ntiles = (n + 6) / 7;
  auto E = [&](double v, int r, int s) { return shf(v, gb + 3 * r + s); };   // entry (r, s) of a matrix held one entry per lane
  for (int64_t tile = wave; tile < ntiles; tile += nwaves) {
    const int64_t gi = tile * 7 + g;
    const bool valid = live && gi < n;
    const int64_t gs = valid ? gi : 0;
    double F = Fin[gs * 9 + pos9];
    if (!valid) F = (i == j) ? 1.0 : 0.0;
    const double p_n = s0[gs];
    double G = s0[(int64_t)(7 + sym) * ld + gs] * (i == j ? 1.0 : RS2);
    if (!valid) G = (i == j) ? 1.0 : 0.0;
    // cofactor entry (i, j): cyclic minors carry the sign
    const double cf = E(F, i1, j1) * E(F, i2, j2) - E(F, i1, j2) * E(F, i2, j1);
    const double fc = F * cf;
    const double J = shf(fc, gb) + shf(fc, gb + 1) + shf(fc, gb + 2);
    const double iJ = 1.0 / J;
    const double Fi = E(cf, j, i) * iJ;                                       // inverse: transposed cofactor
    const double J23 = cbrt(J * J), Jm23 = 1.0 / J23;
    // h[L][m] = J^(-2/3) sum_N G[L][N] F[m][N]          (lane (i, j) = (L, m))
    const double h = Jm23 * (E(G, i, 0) * E(F, j, 0) + E(G, i, 1) * E(F, j, 1) + E(G, i, 2) * E(F, j, 2));
    // be_tr = F h, then its deviator
    double d = E(F, i, 0) * E(h, 0, j) + E(F, i, 1) * E(h, 1, j) + E(F, i, 2) * E(h, 2, j);
    const double Itr = (E(d, 0, 0) + E(d, 1, 1) + E(d, 2, 2)) / 3.0;
    if (i == j) d -= Itr;
    double nrm2 = 0.0;
#pragma unroll
    for (int k = 0; k < 9; ++k) { const double v = shf(d, gb + k); nrm2 += v * v; }
    const double atr = sqrt(nrm2);
    double R_n, dR_n;
    voce(q, p_n, R_n, dR_n);
    double dp = 0.0, Ie = Itr, sdev = d;
    {   // (branching on `plastic` diverges only between the 7 points of a wave; the shuffles run for all lanes)
      const bool plastic = SQ32 * q.mu * atr - R_n > 0.0;
      const double iatr = 1.0 / atr;
      const double sh = d * iatr;
      const double cs = E(sh, i1, j1) * E(sh, i2, j2) - E(sh, i1, j2) * E(sh, i2, j1);
      const double sc = sh * cs;
      const double delta = shf(sc, gb) + shf(sc, gb + 1) + shf(sc, gb + 2);
      if (plastic) {
        double R_1;
        return_scalars(q, atr, delta, Itr, p_n, R_n, dR_n, dp, Ie, R_1);
        sdev = (SQ23 * R_1 / q.mu) * sh;
      }
    }
    const double pr = 0.5 * q.kappa * (J * J - 1.0);
    const double tau = q.mu * sdev + (i == j ? pr : 0.0);
    // P[i][J] = sum_k tau[i][k] Fi[J][k]
    const double P = E(tau, i, 0) * E(Fi, j, 0) + E(tau, i, 1) * E(Fi, j, 1) + E(tau, i, 2) * E(Fi, j, 2);
    const double cI = Ie - pr / q.mu;
    const double t = P / q.mu + cI * E(Fi, j, i);
    const double gn = J23 * (E(Fi, i, 0) * E(t, 0, j) + E(Fi, i, 1) * E(t, 1, j) + E(Fi, i, 2) * E(t, 2, j));
    if (valid) {
      Pout[gi * 9 + pos9] = P;
      if (c == 0) s1[gi] = p_n + dp;
      if (i <= j) {
        s1[(int64_t)(1 + sym) * ld + gi] = (i == j) ? sdev + Ie : SQ2 * sdev;
        s1[(int64_t)(7 + sym) * ld + gi] = (i == j) ? gn : SQ2 * gn;
      }
    }
  }
}

extern "C" int rm_probe_launch(int which, const double* prm7, int maxit, int64_t n, const void* Fin, const void* s0, void* s1, int64_t ld,
                               void* Pout, int blocks, void* stream) {
  Prm q{prm7[0], prm7[1], prm7[2], prm7[3], prm7[4], prm7[5], prm7[6], maxit};
  if (which == 0)
    hipLaunchKernelGGL(rm_thread_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, q, n, (const double*)Fin, (const double*)s0, (double*)s1, ld, (double*)Pout);
  else
    hipLaunchKernelGGL(rm_subwave_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, q, n, (const double*)Fin, (const double*)s0, (double*)s1, ld, (double*)Pout);
  return (int)hipGetLastError();
}
