// Finite-strain FeFp J2 plasticity (gradient F (9), flux PK1 (9), 9x9 tangent dP/dF) for gfx950.
//
// Interface fixed by the reference: jaxmat.py:170-186 (F / PK1, 9-vectors in the order
// [11,22,33,12,21,13,31,23,32] of utils.py:168-190), ISVs p and be_bar (identity initially:
// demos/jax/finite_strain_elastoplasticity/finite_strain_elastoplasticity.py:181), parameters
// tests/test_FeFp_jax.py:7-15.  The arithmetic lives in the absent third-party jaxmat package;
// the algorithm here is the build's own documented choice (DESIGN.md "FeFp", restated on the CPU
// in oracle/constitutive_np.py::fefp_update): Simo's multiplicative J2 model with an exactly
// isochoric radial return, reduced to a 2x2 local Newton in (dp, Ie = tr(be_bar)/3).
//
// Mapping: one thread per Gauss point, one wave per tile of 64 points.  F in and PK1 out move as
// 16 B-per-lane coalesced accesses through wave-private LDS.  The 81-entry tangent (648 of the
// 976 B/point) is never held per thread: with Fi = F^-1 it has the closed form
//   A[(i,J),(k,L)] = V[k][L] Fi[J][i] + U[i][L] Fi[J][k] + W[k][L] Sd[i][J] + (i==k) g[L][J]
// (derivation in DESIGN.md), so each point stages 54 doubles in LDS and the whole wave evaluates
// the entries in output order and stores them as contiguous 1 KiB wave stores, PPR points per
// round to bound the LDS footprint.
#pragma once
#include "dxm_common.hpp"

namespace dxm {

constexpr int FEFP_SLOT_P = 0;    // p
constexpr int FEFP_SLOT_BE = 1;   // be_bar, Mandel (user-visible ISV)
constexpr int FEFP_SLOT_CPI = 7;  // isochoric Cp^-1, Mandel (hidden state)
constexpr int FEFP_NSLOTS = 13;

constexpr int FEFP_PPR = 32;                 // points per tangent round
constexpr int FEFP_NCOEF = 54;               // staged doubles per point
constexpr int FEFP_STAGE = 64 * 9;           // F in / PK1 out staging (doubles per wave)
constexpr int FEFP_LDS_PER_WAVE = FEFP_STAGE + FEFP_PPR * FEFP_NCOEF;

// (row, col) of entry t of the 9-vector, packed 2 bits each: rows [0,1,2,0,1,0,2,1,2],
// cols [0,1,2,1,0,2,0,2,1]  (utils.py:168-190)
constexpr unsigned FEFP_ROWS = 0u | (1u << 2) | (2u << 4) | (0u << 6) | (1u << 8) | (0u << 10) | (2u << 12) | (1u << 14) | (2u << 16);
constexpr unsigned FEFP_COLS = 0u | (1u << 2) | (2u << 4) | (1u << 6) | (0u << 8) | (2u << 10) | (0u << 12) | (2u << 14) | (1u << 16);

__device__ __forceinline__ double det3(const double* A) {
  return A[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (A[3] * A[8] - A[5] * A[6]) +
         A[2] * (A[3] * A[7] - A[4] * A[6]);
}
// cofactor matrix C[i][j] = d det / dA[i][j]
__device__ __forceinline__ void cof3(const double* A, double* C) {
  C[0] = A[4] * A[8] - A[5] * A[7];
  C[1] = A[5] * A[6] - A[3] * A[8];
  C[2] = A[3] * A[7] - A[4] * A[6];
  C[3] = A[7] * A[2] - A[8] * A[1];
  C[4] = A[8] * A[0] - A[6] * A[2];
  C[5] = A[6] * A[1] - A[7] * A[0];
  C[6] = A[1] * A[5] - A[2] * A[4];
  C[7] = A[2] * A[3] - A[0] * A[5];
  C[8] = A[0] * A[4] - A[1] * A[3];
}
// C = A * B
__device__ __forceinline__ void mm(const double* A, const double* B, double* C) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) C[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
}
// C = A * B^T
__device__ __forceinline__ void mmt(const double* A, const double* B, double* C) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) C[i * 3 + j] = A[i * 3] * B[j * 3] + A[i * 3 + 1] * B[j * 3 + 1] + A[i * 3 + 2] * B[j * 3 + 2];
}

__device__ __forceinline__ double voce_R(const LawParams& prm, double p) {
  return prm.sig0 + (prm.h1 - prm.sig0) * (1.0 - exp(-prm.h2 * p));
}
__device__ __forceinline__ double voce_dR(const LawParams& prm, double p) {
  return (prm.h1 - prm.sig0) * prm.h2 * exp(-prm.h2 * p);
}

__global__ void __launch_bounds__(BLOCK, 2)
fefp_kernel(const LawParams prm, const int64_t n, const double* __restrict__ Fin,
            const double* __restrict__ s0, double* __restrict__ s1, const int64_t ld,
            double* __restrict__ Pout, double* __restrict__ ct, BlockStats* __restrict__ stats) {
  __shared__ __attribute__((aligned(16))) double lds_all[WAVES_PER_BLOCK * FEFP_LDS_PER_WAVE];
  __shared__ unsigned long long red[4 * WAVES_PER_BLOCK];

  const int lane = threadIdx.x & (WAVE - 1);
  const int wid = threadIdx.x >> 6;
  double* stage = lds_all + wid * FEFP_LDS_PER_WAVE;
  double* coef = stage + FEFP_STAGE;
  double2_t* stage2 = reinterpret_cast<double2_t*>(stage);

  const int64_t ntiles = (n + WAVE - 1) / WAVE;
  const int64_t tile_stride = (int64_t)gridDim.x * WAVES_PER_BLOCK;
  unsigned long long c_plastic = 0, c_notconv = 0, c_nan = 0, c_maxit = 0;

  const double mu = prm.mu, kappa = prm.kappa;
  const double imu = 1.0 / mu;
  const double SQ32 = 1.2247448713915890491;   // sqrt(3/2)
  const double SQ23 = 0.81649658092772603273;  // sqrt(2/3)
  const double SQ6 = 2.4494897427831780982;    // sqrt(6)
  const double RS2 = 0.70710678118654752440;   // 1/sqrt(2)
  const double SQ2 = 1.4142135623730950488;

  for (int64_t tile = (int64_t)blockIdx.x * WAVES_PER_BLOCK + wid; tile < ntiles; tile += tile_stride) {
    const int64_t base = tile * WAVE;
    const int npts = (n - base) < WAVE ? (int)(n - base) : WAVE;
    const bool valid = lane < npts;
    const int64_t gi = base + lane;

    // ---- 1. coalesced load of F (64 x 9 doubles = 288 double2 per tile) ------------------------
    if (npts == WAVE) {
      const double2_t* gsrc = reinterpret_cast<const double2_t*>(Fin + base * 9);
      double2_t v[5];
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        const int idx = k * WAVE + lane;
        v[k] = (idx < 288) ? gsrc[idx] : double2_t{0.0, 0.0};
      }
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        const int idx = k * WAVE + lane;
        if (idx < 288) stage2[idx] = v[k];
      }
    } else {  // ragged last tile: 8-byte accesses, identity for the missing points
      const double* gsrc = Fin + base * 9;
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const int idx = k * WAVE + lane;
        const int c = idx % 9;
        stage[idx] = (idx < npts * 9) ? gsrc[idx] : (c < 3 ? 1.0 : 0.0);
      }
    }
    double p_n = 0.0, g6[6] = {1, 1, 1, 0, 0, 0};
    if (valid) {
      p_n = s0[(int64_t)FEFP_SLOT_P * ld + gi];
#pragma unroll
      for (int c = 0; c < 6; ++c) g6[c] = s0[(int64_t)(FEFP_SLOT_CPI + c) * ld + gi];
    }
    wave_lds_sync();
    double F[9];
    {
      const double* f = stage + lane * 9;
      F[0] = f[0]; F[4] = f[1]; F[8] = f[2]; F[1] = f[3]; F[3] = f[4];
      F[2] = f[5]; F[6] = f[6]; F[5] = f[7]; F[7] = f[8];
    }
    wave_lds_sync();

    // ---- 2. trial state ------------------------------------------------------------------------
    double G[9];
    G[0] = g6[0]; G[4] = g6[1]; G[8] = g6[2];
    G[1] = G[3] = g6[3] * RS2; G[2] = G[6] = g6[4] * RS2; G[5] = G[7] = g6[5] * RS2;
    const double J = det3(F);
    double Fi[9];
    {
      double cf[9];
      cof3(F, cf);
      const double iJ = 1.0 / J;
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) Fi[r * 3 + c] = cf[c * 3 + r] * iJ;
    }
    const double J23 = cbrt(J * J);
    const double Jm23 = 1.0 / J23;
    double h[9];  // h[L][m] = J^(-2/3) G[L][N] F[m][N]
    mmt(G, F, h);
#pragma unroll
    for (int k = 0; k < 9; ++k) h[k] *= Jm23;
    double d[9];  // be_bar_trial = F h, then its deviator
    mm(F, h, d);
    const double Itr = (d[0] + d[4] + d[8]) / 3.0;  // exact for F = I (zero stress)
    d[0] -= Itr; d[4] -= Itr; d[8] -= Itr;
    double atr2 = 0.0;
#pragma unroll
    for (int k = 0; k < 9; ++k) atr2 += d[k] * d[k];
    const double atr = sqrt(atr2);
    const double f_tr = SQ32 * mu * atr - voce_R(prm, p_n);

    // ---- 3. return mapping ---------------------------------------------------------------------
    double dp = 0.0, Ie = Itr, a = atr, theta = 1.0;
    double Q[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // Q[k][L], scaled by mu/atr below
    double sdev[9];                             // dev(be_bar)
    const bool plastic = f_tr > 0.0;
    if (plastic) {
      double sh[9];
      const double iatr = 1.0 / atr;
#pragma unroll
      for (int k = 0; k < 9; ++k) sh[k] = d[k] * iatr;
      const double delta = det3(sh);
      const double tol1 = (prm.tol / fabs(prm.sig0)) * fmax(fabs(prm.sig0), SQ32 * mu * atr);
      unsigned iters = 0;
      for (int it = 0;; ++it) {
        const double ex = exp(-prm.h2 * (p_n + dp));
        const double aa = SQ23 * (prm.sig0 + (prm.h1 - prm.sig0) * (1.0 - ex)) * imu;
        const double r1 = atr - aa - SQ6 * dp * Ie;
        const double r2 = Ie * Ie * Ie - 0.5 * aa * aa * Ie + aa * aa * aa * delta - 1.0;
        if (fabs(SQ32 * mu * r1) <= tol1 && fabs(r2) <= 1e-14) break;
        if (it >= prm.maxit) { if (valid) ++c_notconv; break; }
        const double ap = SQ23 * ((prm.h1 - prm.sig0) * prm.h2 * ex) * imu;
        const double j11 = -ap - SQ6 * Ie;
        const double j12 = -SQ6 * dp;
        const double j21 = (-aa * Ie + 3.0 * aa * aa * delta) * ap;
        const double j22 = 3.0 * Ie * Ie - 0.5 * aa * aa;
        const double idet = 1.0 / (j11 * j22 - j12 * j21);
        dp += (-r1 * j22 + r2 * j12) * idet;
        Ie += (-j11 * r2 + j21 * r1) * idet;
        ++iters;
      }
      const double p_new = p_n + dp;
      const double exn = exp(-prm.h2 * p_new);
      a = SQ23 * (prm.sig0 + (prm.h1 - prm.sig0) * (1.0 - exn)) * imu;
      const double ap = SQ23 * ((prm.h1 - prm.sig0) * prm.h2 * exn) * imu;
      theta = a * iatr;
#pragma unroll
      for (int k = 0; k < 9; ++k) sdev[k] = a * sh[k];
      // implicit differentiation of (r1, r2) = 0 with respect to (atr, delta)
      const double igI = 1.0 / (3.0 * Ie * Ie - 0.5 * a * a);
      const double dIe_da = (a * Ie - 3.0 * a * a * delta) * igI;
      const double dIe_dd = -(a * a * a) * igI;
      const double r_dp = -ap - SQ6 * Ie - SQ6 * dp * dIe_da * ap;
      const double r_dd = -SQ6 * dp * dIe_dd;
      double cs[9], hs[9], hc[9];
      cof3(sh, cs);
      mm(h, sh, hs);  // (h s)[L][k]
      mm(h, cs, hc);  // (h cof)[L][k]
      const double trc = cs[0] + cs[4] + cs[8];
      const double ir_dp = 1.0 / r_dp;
#pragma unroll
      for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int L = 0; L < 3; ++L) {
          const double n1 = 2.0 * hs[L * 3 + k] - (2.0 / 3.0) * atr * Fi[L * 3 + k];
          const double n2 = (2.0 * hc[L * 3 + k] - (2.0 / 3.0) * trc * h[L * 3 + k] -
                             2.0 * delta * atr * Fi[L * 3 + k] - 3.0 * delta * n1) * iatr;
          const double np_ = -(n1 + r_dd * n2) * ir_dp;
          Q[k * 3 + L] = (ap * np_ - theta * n1) * (mu * iatr);
        }
      if (valid) {
        ++c_plastic;
        c_maxit = iters > c_maxit ? iters : c_maxit;
      }
    } else {
#pragma unroll
      for (int k = 0; k < 9; ++k) sdev[k] = d[k];
      Ie = Itr;
    }
    const double p_new = p_n + dp;

    // ---- 4. stress, new state --------------------------------------------------------------------
    double tau[9];
    const double pr = 0.5 * kappa * (J * J - 1.0);
#pragma unroll
    for (int k = 0; k < 9; ++k) tau[k] = mu * sdev[k];
    tau[0] += pr; tau[4] += pr; tau[8] += pr;
    double P[9];
    mmt(tau, Fi, P);  // P = tau F^-T
    double be[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) be[k] = sdev[k];
    be[0] += Ie; be[4] += Ie; be[8] += Ie;
    {
      double chk = p_new;
#pragma unroll
      for (int k = 0; k < 9; ++k) chk += P[k];
      if (valid && !(fabs(chk) <= 1.79769313486231570e308)) ++c_nan;
    }
    if (valid) {
      double t[9], gn[9];
      mmt(be, Fi, t);   // be F^-T
      mm(Fi, t, gn);    // F^-1 be F^-T
      s1[(int64_t)FEFP_SLOT_P * ld + gi] = p_new;
      s1[(int64_t)(FEFP_SLOT_BE + 0) * ld + gi] = be[0];
      s1[(int64_t)(FEFP_SLOT_BE + 1) * ld + gi] = be[4];
      s1[(int64_t)(FEFP_SLOT_BE + 2) * ld + gi] = be[8];
      s1[(int64_t)(FEFP_SLOT_BE + 3) * ld + gi] = SQ2 * be[1];
      s1[(int64_t)(FEFP_SLOT_BE + 4) * ld + gi] = SQ2 * be[2];
      s1[(int64_t)(FEFP_SLOT_BE + 5) * ld + gi] = SQ2 * be[5];
      s1[(int64_t)(FEFP_SLOT_CPI + 0) * ld + gi] = J23 * gn[0];
      s1[(int64_t)(FEFP_SLOT_CPI + 1) * ld + gi] = J23 * gn[4];
      s1[(int64_t)(FEFP_SLOT_CPI + 2) * ld + gi] = J23 * gn[8];
      s1[(int64_t)(FEFP_SLOT_CPI + 3) * ld + gi] = SQ2 * J23 * 0.5 * (gn[1] + gn[3]);
      s1[(int64_t)(FEFP_SLOT_CPI + 4) * ld + gi] = SQ2 * J23 * 0.5 * (gn[2] + gn[6]);
      s1[(int64_t)(FEFP_SLOT_CPI + 5) * ld + gi] = SQ2 * J23 * 0.5 * (gn[5] + gn[7]);
    }

    // ---- 5. PK1 through LDS, coalesced store -------------------------------------------------------
    {
      double* f = stage + lane * 9;
      f[0] = P[0]; f[1] = P[4]; f[2] = P[8]; f[3] = P[1]; f[4] = P[3];
      f[5] = P[2]; f[6] = P[6]; f[7] = P[5]; f[8] = P[7];
    }
    wave_lds_sync();
    if (npts == WAVE) {
      double2_t* gdst = reinterpret_cast<double2_t*>(Pout + base * 9);
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        const int idx = k * WAVE + lane;
        if (idx < 288) gdst[idx] = stage2[idx];
      }
    } else {
      double* gdst = Pout + base * 9;
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const int idx = k * WAVE + lane;
        if (idx < npts * 9) gdst[idx] = stage[idx];
      }
    }

    // ---- 6. tangent: per-point coefficients, then cooperative entry evaluation ----------------------
    const double mt = mu * theta;
    const double c0 = kappa * J * J;
    double V[9], U[9], Sd[9];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int L = 0; L < 3; ++L) {
        V[k * 3 + L] = c0 * Fi[L * 3 + k] - (2.0 / 3.0) * mt * h[L * 3 + k];
        U[k * 3 + L] = mt * h[L * 3 + k] - P[k * 3 + L];                 // U[i][L], i = k here
        Q[k * 3 + L] = Q[k * 3 + L] - (2.0 / 3.0) * mt * Fi[L * 3 + k];  // W[k][L]
      }
    mmt(d, Fi, Sd);  // Sd[i][J] = d[i][m] Fi[J][m]
    const double gs = mt * Jm23;

#pragma unroll 1
    for (int round = 0; round < WAVE / FEFP_PPR; ++round) {
      if ((lane / FEFP_PPR) == round) {
        double2_t* c2 = reinterpret_cast<double2_t*>(coef + (lane % FEFP_PPR) * FEFP_NCOEF);
        // layout: Fi 0..8 | V 9..17 | U 18..26 | W 27..35 | Sd 36..44 | g 45..53
        double buf[FEFP_NCOEF];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
          buf[k] = Fi[k]; buf[9 + k] = V[k]; buf[18 + k] = U[k]; buf[27 + k] = Q[k];
          buf[36 + k] = Sd[k]; buf[45 + k] = gs * G[k];
        }
#pragma unroll
        for (int k = 0; k < FEFP_NCOEF / 2; ++k) c2[k] = double2_t{buf[2 * k], buf[2 * k + 1]};
      }
      wave_lds_sync();
      const int p0 = round * FEFP_PPR;              // first point of the round
      int np_round = npts - p0;
      np_round = np_round < 0 ? 0 : (np_round > FEFP_PPR ? FEFP_PPR : np_round);
      const int nent = np_round * 81;               // entries of this round
      double* gct = ct + (base + p0) * 81;
      constexpr int NITER = (FEFP_PPR * 81 + 2 * WAVE - 1) / (2 * WAVE);
      // (base + p0) * 81 is even: every round starts 16 B aligned
#pragma unroll 2
      for (int it = 0; it < NITER; ++it) {
        const int e0 = (it * WAVE + lane) * 2;
        double v[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int e = e0 + u;
          const int q = e / 81;
          const int m = e - q * 81;
          const int row = m / 9, col = m - row * 9;
          const int i = (FEFP_ROWS >> (2 * row)) & 3, Jx = (FEFP_COLS >> (2 * row)) & 3;
          const int k = (FEFP_ROWS >> (2 * col)) & 3, L = (FEFP_COLS >> (2 * col)) & 3;
          const double* c = coef + (q < FEFP_PPR ? q : 0) * FEFP_NCOEF;
          double x = c[9 + k * 3 + L] * c[Jx * 3 + i] + c[18 + i * 3 + L] * c[Jx * 3 + k] +
                     c[27 + k * 3 + L] * c[36 + i * 3 + Jx];
          if (i == k) x += c[45 + L * 3 + Jx];
          v[u] = x;
        }
        if (e0 + 1 < nent) {
          *reinterpret_cast<double2_t*>(gct + e0) = double2_t{v[0], v[1]};
        } else if (e0 < nent) {
          gct[e0] = v[0];
        }
      }
      wave_lds_sync();
    }
  }
  store_block_stats(stats, c_plastic, c_notconv, c_nan, c_maxit, red);
}

}  // namespace dxm
