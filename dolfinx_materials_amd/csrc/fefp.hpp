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
// (derivation in DESIGN.md).  Each point stages 54 doubles in LDS, 14 points per round; lane
// (point slot, tangent column) then evaluates the 9 rows of its column with compile-time row
// indices (4 FMAs per entry, no per-entry index arithmetic), the results are transposed through
// an LDS out-tile and leave as contiguous 1 KiB wave stores (full 64 B HBM write requests).
#pragma once
#include "dxm_common.hpp"
#include "gradient.hpp"

namespace dxm {

constexpr int FEFP_SLOT_P = 0;    // p
constexpr int FEFP_SLOT_BE = 1;   // be_bar, Mandel (user-visible ISV)
constexpr int FEFP_SLOT_CPI = 7;  // isochoric Cp^-1, Mandel (hidden state)
constexpr int FEFP_NSLOTS = 13;

constexpr int FEFP_STAGE = 64 * 9;           // F in / PK1 out staging (doubles per wave)
constexpr int F2_REC = 73;   // record stride: 54 staged doubles per point, padded to 146 dwords = 18 mod 64,
                             // so that the owner lanes write conflict-free (18 l mod 32 distinct) and the
                             // point slots of a 32-lane read group land on (almost) disjoint banks
// Wave-private LDS map (doubles): [out-tile, aliasing the F / PK1 staging | records].  A tangent round handles
// F2_PPR points in steps of 7 point slots x 9 columns.
// 16 points per round: a round's 16 x 648 B of tangent start on a 128 B line of the (N, 81) array, so every 1 KiB wave
// store covers whole lines.  With 14 points per round (the mapping's natural 2 x 7 point slots) the rounds started 112 /
// 96 / 80 / 64 B past a line and every store instruction straddled two partial lines: an arithmetic-free kernel with this
// kernel's streams, occupancy and compute gaps runs 1.63 ms per 1e7 points with 16-point rounds against 1.77 with 14
// (tools/fefp_shape_probe.py, profiles/archive/r03_fefp_shape_probe.md).  The price is a third, mostly idle, step per round
// (7 + 7 + 2 point slots): 12 instead of 10 tangent steps per tile.
// (A/B knobs of the residency study, profiles/r04_fefp_third_wave_and_subwave.md: 8-point rounds cost 9 %, a 168-register budget
// spills 78 VGPRs and costs 80 %; the defaults are what ships)
#ifndef DXM_FEFP_PPR
#define DXM_FEFP_PPR 16
#endif
#ifndef DXM_FEFP_WGS
#define DXM_FEFP_WGS 2
#endif
constexpr int F2_PPR = DXM_FEFP_PPR;
constexpr int F2_STEPS = (F2_PPR + 6) / 7;
constexpr int F2_NIT = (F2_PPR * 81 + 2 * WAVE - 1) / (2 * WAVE);   // 1 KiB wave stores per round (the last one partial)
constexpr int F2_OUT = F2_PPR * 81;                                 // out-tile; the copy-out's last, partial KiB reads on into the records
constexpr int F2_COEF = ((F2_PPR * F2_REC + 1) / 2) * 2;
constexpr int F2_LDS_PER_WAVE = F2_OUT + F2_COEF;                   // 2464 doubles: 78.8 KB per workgroup, 2 per CU
static_assert(F2_OUT % 2 == 0 && (F2_NIT * 2 * WAVE - F2_OUT) <= F2_COEF, "16 B aligned records; the over-read stays inside the wave's region");
static_assert(4 * F2_LDS_PER_WAVE * 8 <= 80 * 1024, "two workgroups per CU");
static_assert(F2_OUT >= FEFP_STAGE, "the out-tile aliases the F / PK1 staging region");
static_assert(8 * HEX_FUSED_REC <= F2_COEF, "the 8 cell records of a fused tile live in the coefficient region");

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

// hardening law and its slope, sharing the exponential
template <int HARD>
__device__ __forceinline__ void hardening(const LawParams& prm, double p, double& R, double& dR) {
  if constexpr (HARD == 0) {
    R = prm.sig0 + prm.h1 * p;
    dR = prm.h1;
  } else {
#ifdef DXM_CUSTOM_HARDENING
    R = custom_R(prm, p);
    dR = custom_dR(prm, p);
#else
    const double ex = exp(DXM_MUL(-prm.h2, p));
    R = prm.sig0 + DXM_MUL(prm.h1 - prm.sig0, 1.0 - ex);
    dR = DXM_MUL((prm.h1 - prm.sig0) * prm.h2, ex);
#endif
  }
}

// symmetric 3x3 matrices are kept as 6 values [xx, yy, zz, xy, xz, yz]; SYM(i, j) is the slot of entry (i, j)
#define DXM_SYM(i, j) ((i) == (j) ? (i) : ((i) + (j) + 2))

// HARD: 0 = linear hardening R = sig0 + H p (prm.h1 = H), 1 = Voce (prm.h1 = sigu, prm.h2 = b)
// GRAD: 0 = F comes from the (N,9) array Fin; 1 / 2 / 3 = F = I + grad u is evaluated in the kernel from the
//       displacement vector of a hex8 mesh with 8 Gauss points per cell / of a tet4 mesh / of a straight-sided
//       simplex mesh with a Lagrange displacement of any order (`src`, see small_strain.hpp)
//
// What was measured for this kernel in round 2 (profiles/archive/r02_fefp_ab_*.jsonl; builds side by side in one process,
// five handles each, every handle with its state placement searched: the kernel is placement-sensitive like the J2
// one and comparisons of single un-tuned handles are dominated by that):
//   * register diet (symmetric storage, Q = h M form of the tangent coefficients, per-tile re-derivation of the
//     lane invariants): 256 VGPRs + 27 spilled -> 255 VGPRs, no scratch                                   1.90 -> 1.85 ms
//   * tile bookkeeping in scalar registers (readfirstlane of the wave index: every "uniform" branch had been an
//     exec-mask sequence), one hardening evaluation per Newton iterate                                       -> 1.76 ms
//   * record / out-tile bases as opaque indices (LDS accesses become base + immediate), straight-line copy-out for
//     the two round shapes of a full tile                                                                      -> 1.74 ms
//   * NOT shipped: requesting the next tile's inputs ahead of this tile's stores by LDS-DMA (global_load_lds into a
//     spare LDS region, counted s_waitcnt at the top of the next tile; needs 7-point rounds to make room): +6 %, of
//     which +5 % is the 7-point rounds alone -- load latency and the in-order completion of a wave's memory
//     operations are not what the kernel waits for; unrolling the two steps of a round: no gain.
// TLF: 0 = the 9x9 tangent (81 doubles per point); 1 = its 54 building blocks per point (the record of step 6 below,
//      432 instead of 648 B: what the host-buffer form moves over PCIe before rebuilding the block on the host with the
//      same four-term expression, dxmat.hip::expand_fefp_tangent)
constexpr int FEFP_REC = 54;
template <int HARD, int GRAD = 0, int TLF = 0>
__global__ void __launch_bounds__(BLOCK, DXM_FEFP_WGS)
fefp_kernel(const LawParams prm, const int64_t n, const double* __restrict__ Fin,
            const double* __restrict__ s0, double* __restrict__ s1, const int64_t ld,
            double* __restrict__ Pout, double* __restrict__ ct, BlockStats* __restrict__ stats,
            const MeshSource src) {
  __shared__ __attribute__((aligned(16))) double lds_all[WAVES_PER_BLOCK * F2_LDS_PER_WAVE];

  const int lane0 = threadIdx.x & (WAVE - 1);
  // wave-uniform by construction: told to the compiler, so that the tile index, the point count of the tile and
  // every branch on them live in scalar registers (as a VGPR value each "uniform" branch costs an exec-mask dance)
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double* stage = lds_all + wid * F2_LDS_PER_WAVE;   // F in / PK1 out staging ...
  double* outt = stage;                              // ... reused as the tangent out-tile
  double* coef = stage + F2_OUT;
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

  constexpr int TI[9] = {0, 1, 2, 0, 1, 0, 2, 1, 2};  // row index of entry t of the 9-vector
  constexpr int TJ[9] = {0, 1, 2, 1, 0, 2, 0, 2, 1};  // column index           (utils.py:168-190)
  int lane = lane0;

  for (int64_t tile = (int64_t)blockIdx.x * WAVES_PER_BLOCK + wid; tile < ntiles; tile += tile_stride) {
    const int64_t base = tile * WAVE;
    const int npts = (n - base) < WAVE ? (int)(n - base) : WAVE;
    // Everything derived from the lane index is re-derived per tile from an opaque copy: hoisted out of the
    // tile loop these per-lane invariants (LDS addresses, column indices, masks) occupy ~40 VGPRs across the
    // whole body and the kernel spills at its 256-register budget; recomputing them costs ~30 integer ops.
    asm volatile("" : "+v"(lane));
    lane &= WAVE - 1;   // gives the value range back to the compiler
    const bool valid = lane < npts;
    const int64_t gi = base + lane;
    // tangent epilogue: lane = (point slot ps, tangent column cc); lane 63 idles
    const int ps = lane / 9;
    const int cc = lane - ps * 9;
    const int kk = (0x26124 >> (2 * cc)) & 3;   // TI[cc] packed 2 bits each: 0,1,2,0,1,0,2,1,2
    const int LL = (0x18864 >> (2 * cc)) & 3;   // TJ[cc]: 0,1,2,1,0,2,0,2,1
    const double mk0 = kk == 0 ? 1.0 : 0.0, mk1 = kk == 1 ? 1.0 : 0.0, mk2 = kk == 2 ? 1.0 : 0.0;

    double F[9];
    double p_n = 0.0, g6[6] = {1, 1, 1, 0, 0, 0};
    if constexpr (GRAD == 0) {
      // ---- 1. F through LDS (64 x 9 doubles = 288 double2 per tile) ---------------------------------
      if (npts == WAVE) {
        const double2_t* gsrc = reinterpret_cast<const double2_t*>(Fin + base * 9);
        double2_t v[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
          const int idx = k * WAVE + lane;
          v[k] = (idx < 288) ? stream_load<2>(gsrc + idx) : double2_t{0.0, 0.0};
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
      if (valid) {
        p_n = stream_load<3>(s0 + (int64_t)FEFP_SLOT_P * ld + gi);
#pragma unroll
        for (int c = 0; c < 6; ++c) g6[c] = stream_load<3>(s0 + (int64_t)(FEFP_SLOT_CPI + c) * ld + gi);
      }
      wave_lds_sync();
      {
        const double* f = stage + lane * 9;
        F[0] = f[0]; F[4] = f[1]; F[8] = f[2]; F[1] = f[3]; F[3] = f[4];
        F[2] = f[5]; F[6] = f[6]; F[5] = f[7]; F[7] = f[8];
      }
      wave_lds_sync();
    } else {
      if constexpr (GRAD == 1) {
        // ---- 1'. one (cell, corner) per lane: node -> wave-private record in the coefficient region
        {
          const int64_t cell = ((src.point0 + base) >> 3) + (lane >> 3);
          double2_t r0 = {0.0, 0.0}, r1 = {0.0, 0.0}, r2 = {0.0, 0.0};
          if (cell < src.ncells) {
            const int64_t nd = src.conn[cell * 8 + (lane & 7)];
            r0 = double2_t{src.coords[3 * nd], src.coords[3 * nd + 1]};
            r1 = double2_t{src.coords[3 * nd + 2], src.u[3 * nd]};
            r2 = double2_t{src.u[3 * nd + 1], src.u[3 * nd + 2]};
          }
          double2_t* d = reinterpret_cast<double2_t*>(coef + (lane >> 3) * HEX_FUSED_REC + (lane & 7) * 6);
          d[0] = r0; d[1] = r1; d[2] = r2;
        }
        wave_lds_sync();
        {
          const double2_t* rec = reinterpret_cast<const double2_t*>(coef + (lane >> 3) * HEX_FUSED_REC);
          auto node = [&](int m, double* X, double* U) {
            const double2_t a = rec[m * 3], b = rec[m * 3 + 1], c = rec[m * 3 + 2];
            X[0] = a.x; X[1] = a.y; X[2] = b.x;
            U[0] = b.y; U[1] = c.x; U[2] = c.y;
          };
          const int q = lane & 7;
          if (valid) {
            hex8_disp_grad(src.xi[q][0], src.xi[q][1], src.xi[q][2], node, F);   // H[i][j], row-major like F
          } else {
#pragma unroll
            for (int k = 0; k < 9; ++k) F[k] = 0.0;
          }
        }
        wave_lds_sync();  // the coefficient region is rewritten by the tangent rounds
      } else {
        if (valid) {
          const int64_t cell = (src.point0 + gi) / src.nqp;
          if constexpr (GRAD == 2) tet4_cell_disp_grad(src.coords, src.conn, src.u, cell, F);
          else simplex_disp_grad(src, cell, (int)(src.point0 + gi - cell * src.nqp), F);
        } else {
#pragma unroll
          for (int k = 0; k < 9; ++k) F[k] = 0.0;
        }
      }
      F[0] += 1.0; F[4] += 1.0; F[8] += 1.0;
      if (valid) {
        p_n = stream_load<3>(s0 + (int64_t)FEFP_SLOT_P * ld + gi);
#pragma unroll
        for (int c = 0; c < 6; ++c) g6[c] = stream_load<3>(s0 + (int64_t)(FEFP_SLOT_CPI + c) * ld + gi);
      }
    }

    // ---- 2. trial state ------------------------------------------------------------------------
    double G[6];   // Cp_bar^-1 (symmetric)
    G[0] = g6[0]; G[1] = g6[1]; G[2] = g6[2];
    G[3] = g6[3] * RS2; G[4] = g6[4] * RS2; G[5] = g6[5] * RS2;
    double Fi[9], J;
    {
      double cf[9];
      cof3(F, cf);
      J = F[0] * cf[0] + F[1] * cf[1] + F[2] * cf[2];
      // an inverted or flat point (det F <= 0) has no answer: J^(-2/3) is NaN in the reference's arithmetic and
      // QuadratureMap.update asserts on it (quadrature_map.py:322-324); cbrt(J^2) below would quietly accept it
      J = J > 0.0 ? J : __builtin_nan("");
      const double iJ = fast_rcp(J);
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) Fi[r * 3 + c] = cf[c * 3 + r] * iJ;
    }
    const double J23 = cbrt(J * J);
    const double Jm23 = fast_rcp(J23);
    double h[9];  // h[L][m] = J^(-2/3) G[L][N] F[m][N]
#pragma unroll
    for (int L = 0; L < 3; ++L)
#pragma unroll
      for (int m = 0; m < 3; ++m)
        h[L * 3 + m] = Jm23 * (G[DXM_SYM(L, 0)] * F[m * 3] + G[DXM_SYM(L, 1)] * F[m * 3 + 1] + G[DXM_SYM(L, 2)] * F[m * 3 + 2]);
    double d[6];  // be_bar_trial = F h (symmetric), then its deviator
    {
      constexpr int SI[6] = {0, 1, 2, 0, 0, 1}, SJ[6] = {0, 1, 2, 1, 2, 2};
#pragma unroll
      for (int t = 0; t < 6; ++t)
        d[t] = F[SI[t] * 3] * h[SJ[t]] + F[SI[t] * 3 + 1] * h[3 + SJ[t]] + F[SI[t] * 3 + 2] * h[6 + SJ[t]];
    }
    const double Itr = (d[0] + d[1] + d[2]) / 3.0;  // exact for F = I (zero stress)
    d[0] -= Itr; d[1] -= Itr; d[2] -= Itr;
    const double atr = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + 2.0 * (d[3] * d[3] + d[4] * d[4] + d[5] * d[5]));
    double R_n, dR_n;
    hardening<HARD>(prm, p_n, R_n, dR_n);
    const double f_tr = SQ32 * mu * atr - R_n;

    // ---- 3. return mapping ---------------------------------------------------------------------
    // Plastic: dev be = a s_hat, tr be / 3 = Ie from the 2x2 system (r1, r2); the tangent needs
    //   Q[k][L] = (mu / atr) (a' dDp/dF - theta datr/dF)[k][L] = (h M)[L][k] + eq Fi[L][k]
    // with the symmetric M = ms s_hat + mc cof(s_hat) + m1 1 (implicit differentiation of (r1, r2) with
    // respect to (atr, det s_hat)).
    double dp = 0.0, Ie = Itr, theta = 1.0, eq = 0.0;
    double M[6] = {0, 0, 0, 0, 0, 0};
    double sdev[6];                             // dev(be_bar)
    const bool plastic = f_tr > 0.0;
    if (plastic) {
      double sh[6], cs[6];
      const double iatr = fast_rcp(atr);
#pragma unroll
      for (int k = 0; k < 6; ++k) sh[k] = d[k] * iatr;
      cs[0] = sh[1] * sh[2] - sh[5] * sh[5];
      cs[1] = sh[0] * sh[2] - sh[4] * sh[4];
      cs[2] = sh[0] * sh[1] - sh[3] * sh[3];
      cs[3] = sh[4] * sh[5] - sh[3] * sh[2];
      cs[4] = sh[3] * sh[5] - sh[4] * sh[1];
      cs[5] = sh[3] * sh[4] - sh[0] * sh[5];
      const double delta = sh[0] * cs[0] + sh[3] * cs[3] + sh[4] * cs[4];
      const double tol1 = fmax(prm.tol, prm.rtol * (SQ32 * mu * atr));
      unsigned iters = 0;
      // the hardening law (an exp for Voce, anything for a traced law) is evaluated once per iterate: the first
      // iterate (dp = 0) reuses the values of the yield test, the converged ones are reused after the loop
      double R_k = R_n, dR_k = dR_n;
      for (int it = 0;; ++it) {
        const double aa = SQ23 * R_k * imu;
        const double r1 = atr - aa - SQ6 * dp * Ie;
        const double r2 = Ie * Ie * Ie - 0.5 * aa * aa * Ie + aa * aa * aa * delta - 1.0;
        if (fabs(SQ32 * mu * r1) <= tol1 && fabs(r2) <= 1e-14) break;
        if (it >= prm.maxit) { if (valid) ++c_notconv; break; }
        const double ap = SQ23 * dR_k * imu;
        const double j11 = -ap - SQ6 * Ie;
        const double j12 = -SQ6 * dp;
        const double j21 = (-aa * Ie + 3.0 * aa * aa * delta) * ap;
        const double j22 = 3.0 * Ie * Ie - 0.5 * aa * aa;
        const double idet = fast_rcp(j11 * j22 - j12 * j21);
        dp += (-r1 * j22 + r2 * j12) * idet;
        Ie += (-j11 * r2 + j21 * r1) * idet;
        ++iters;
        hardening<HARD>(prm, p_n + dp, R_k, dR_k);
      }
      const double R_1 = R_k, dR_1 = dR_k;
      const double a = SQ23 * R_1 * imu;
      const double ap = SQ23 * dR_1 * imu;
      theta = a * iatr;
#pragma unroll
      for (int k = 0; k < 6; ++k) sdev[k] = a * sh[k];
      // implicit differentiation of (r1, r2) = 0 with respect to (atr, delta)
      const double igI = fast_rcp(3.0 * Ie * Ie - 0.5 * a * a);
      const double dIe_da = (a * Ie - 3.0 * a * a * delta) * igI;
      const double dIe_dd = -(a * a * a) * igI;
      const double r_dp = -ap - SQ6 * Ie - SQ6 * dp * dIe_da * ap;
      const double r_dd = -SQ6 * dp * dIe_dd;
      const double ir_dp = fast_rcp(r_dp);
      const double ca = -ap * ir_dp - theta;          // coefficient of N1 = datr/dF
      const double cb = -ap * r_dd * ir_dp * iatr;    // coefficient of atr N2 = atr ddelta/dF
      const double mia = mu * iatr;
      const double ms = mia * (2.0 * ca - 6.0 * delta * cb);
      const double mc = mia * 2.0 * cb;
      const double m1 = -mia * (2.0 / 3.0) * (cs[0] + cs[1] + cs[2]) * cb;
#pragma unroll
      for (int k = 0; k < 6; ++k) M[k] = ms * sh[k] + mc * cs[k];
      M[0] += m1; M[1] += m1; M[2] += m1;
      eq = -(2.0 / 3.0) * mu * ca;
      if (valid) {
        ++c_plastic;
        c_maxit = iters > c_maxit ? iters : c_maxit;
      }
    } else {
#pragma unroll
      for (int k = 0; k < 6; ++k) sdev[k] = d[k];
    }
    const double p_new = p_n + dp;

    // ---- 4. stress, new state --------------------------------------------------------------------
    const double pr = 0.5 * kappa * (J * J - 1.0);
    double P[9];   // P = tau F^-T, tau = mu dev(be_bar) + pr 1
    {
      double tau[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) tau[k] = mu * sdev[k];
      tau[0] += pr; tau[1] += pr; tau[2] += pr;
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int Jx = 0; Jx < 3; ++Jx)
          P[i * 3 + Jx] = tau[DXM_SYM(i, 0)] * Fi[Jx * 3] + tau[DXM_SYM(i, 1)] * Fi[Jx * 3 + 1] + tau[DXM_SYM(i, 2)] * Fi[Jx * 3 + 2];
    }
    bool nonfinite;   // one bit per lane (an SGPR pair): settled after the tangent's building blocks exist (step 6)
    {
      double chk = p_new;
#pragma unroll
      for (int k = 0; k < 9; ++k) chk += P[k];
      nonfinite = !(fabs(chk) <= 1.79769313486231570e308);
    }
    // new isochoric Cp^-1 = J^(2/3) F^-1 be F^-T with be F^-T = P / mu + (Ie - pr / mu) F^-T
    double gn[6];
    {
      double t[9];
      const double cI = Ie - pr * imu;
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int Jx = 0; Jx < 3; ++Jx) t[i * 3 + Jx] = P[i * 3 + Jx] * imu + cI * Fi[Jx * 3 + i];
      constexpr int SI[6] = {0, 1, 2, 0, 0, 1}, SJ[6] = {0, 1, 2, 1, 2, 2};
#pragma unroll
      for (int k = 0; k < 6; ++k)
        gn[k] = J23 * (Fi[SI[k] * 3] * t[SJ[k]] + Fi[SI[k] * 3 + 1] * t[3 + SJ[k]] + Fi[SI[k] * 3 + 2] * t[6 + SJ[k]]);
    }

    if (valid) {
      stream_store<1>(s1 + (int64_t)FEFP_SLOT_P * ld + gi, p_new);
      stream_store<1>(s1 + (int64_t)(FEFP_SLOT_BE + 0) * ld + gi, sdev[0] + Ie);
      stream_store<1>(s1 + (int64_t)(FEFP_SLOT_BE + 1) * ld + gi, sdev[1] + Ie);
      stream_store<1>(s1 + (int64_t)(FEFP_SLOT_BE + 2) * ld + gi, sdev[2] + Ie);
      stream_store<1>(s1 + (int64_t)(FEFP_SLOT_BE + 3) * ld + gi, SQ2 * sdev[3]);
      stream_store<1>(s1 + (int64_t)(FEFP_SLOT_BE + 4) * ld + gi, SQ2 * sdev[4]);
      stream_store<1>(s1 + (int64_t)(FEFP_SLOT_BE + 5) * ld + gi, SQ2 * sdev[5]);
      stream_store<1>(s1 + (int64_t)(FEFP_SLOT_CPI + 0) * ld + gi, gn[0]);
      stream_store<1>(s1 + (int64_t)(FEFP_SLOT_CPI + 1) * ld + gi, gn[1]);
      stream_store<1>(s1 + (int64_t)(FEFP_SLOT_CPI + 2) * ld + gi, gn[2]);
      stream_store<1>(s1 + (int64_t)(FEFP_SLOT_CPI + 3) * ld + gi, SQ2 * gn[3]);
      stream_store<1>(s1 + (int64_t)(FEFP_SLOT_CPI + 4) * ld + gi, SQ2 * gn[4]);
      stream_store<1>(s1 + (int64_t)(FEFP_SLOT_CPI + 5) * ld + gi, SQ2 * gn[5]);
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
        if (idx < 288) stream_store<0>(gdst + idx, stage2[idx]);
      }
    } else {
      double* gdst = Pout + base * 9;
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const int idx = k * WAVE + lane;
        if (idx < npts * 9) gdst[idx] = stage[idx];
      }
    }

    // ---- 6. tangent.  Per point 54 doubles are staged in LDS:
    //   Fi[J][i] 0..8 | Vc[col] 9..17 | U[i][L] 18..26 | Wc[col] 27..35 | Sr[row] 36..44 | g[L][J] 45..53
    // A[row=(i,J)][col=(k,L)] = Vc[col] Fi[J][i] + Wc[col] Sr[row] + U[i][L] Fi[J][k] + (i==k) g[L][J].
    // Lane (ps, cc) owns tangent COLUMN cc of point slot ps: k, L and the (i==k) masks are lane
    // constants, the 9 rows are unrolled with compile-time (i, J), so an entry costs 4 FMAs and
    // ~3 LDS reads with immediate offsets.  Results are transposed through an LDS out-tile of
    // F2_PPR points and leave as contiguous 1 KiB wave stores (full 64 B HBM write requests).
    const double mt = mu * theta;
    const double c0 = kappa * J * J;
    const double eqw = eq - (2.0 / 3.0) * mt;
    const double gs = mt * Jm23;
    double V[9], U[9], W[9], Sd[9];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int L = 0; L < 3; ++L) {
        V[k * 3 + L] = c0 * Fi[L * 3 + k] - (2.0 / 3.0) * mt * h[L * 3 + k];
        U[k * 3 + L] = mt * h[L * 3 + k] - P[k * 3 + L];                 // U[i][L], i = k here
        W[k * 3 + L] = h[L * 3] * M[DXM_SYM(0, k)] + h[L * 3 + 1] * M[DXM_SYM(1, k)] + h[L * 3 + 2] * M[DXM_SYM(2, k)] +
                       eqw * Fi[L * 3 + k];
        // Sd[i][J] = d[i][m] Fi[J][m]   (i = k, J = L here)
        Sd[k * 3 + L] = d[DXM_SYM(k, 0)] * Fi[L * 3] + d[DXM_SYM(k, 1)] * Fi[L * 3 + 1] + d[DXM_SYM(k, 2)] * Fi[L * 3 + 2];
      }
    {
      // the tangent counts too (quadrature_map.py:324): W carries everything that depends on the hardening SLOPE (M, eq), which
      // the stress never sees -- a slope that is not finite at the returned state leaves PK1 finite and the tangent not
      const double wsum = ((W[0] + W[1]) + (W[2] + W[3])) + ((W[4] + W[5]) + (W[6] + W[7])) + W[8];
      if (valid && (nonfinite || !(fabs(wsum) <= 1.79769313486231570e308))) ++c_nan;
    }

    if constexpr (TLF == 1) {
      // building blocks only: every lane leaves its own record (rare path: the transfer behind it is 50x the kernel)
      if (valid) {
        double* rec = ct + gi * FEFP_REC;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          rec[t] = Fi[t];
          rec[9 + t] = V[TI[t] * 3 + TJ[t]];
          rec[18 + t] = U[t];
          rec[27 + t] = W[TI[t] * 3 + TJ[t]];
          rec[36 + t] = Sd[TI[t] * 3 + TJ[t]];
          rec[45 + t] = gs * G[DXM_SYM(t / 3, t % 3)];
        }
      }
    } else
#pragma unroll 1
    for (int rd = 0; rd < (WAVE + F2_PPR - 1) / F2_PPR; ++rd) {
      const int p0 = rd * F2_PPR;                               // first point of the round
      const int cnt = (WAVE - p0) < F2_PPR ? (WAVE - p0) : F2_PPR;  // points staged this round
      if (lane >= p0 && lane < p0 + cnt) {
        // record base as an opaque index: every access below is base + small immediate (as a foldable constant the
        // 9 KiB offset of the record region exceeds the offset field of ds_write2_b64 and costs one v_add per pair)
        int ro = F2_OUT + (lane - p0) * F2_REC;
        asm volatile("" : "+v"(ro));
        double* rec = stage + ro;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          rec[t] = Fi[t];
          rec[9 + t] = V[TI[t] * 3 + TJ[t]];
          rec[18 + t] = U[t];
          rec[27 + t] = W[TI[t] * 3 + TJ[t]];
          rec[36 + t] = Sd[TI[t] * 3 + TJ[t]];
          rec[45 + t] = gs * G[DXM_SYM(t / 3, t % 3)];
        }
      }
      wave_lds_sync();
#pragma unroll 1   // (unrolled: no spill, no gain: 1.758 vs 1.755 ms)
      for (int st = 0; st < F2_STEPS; ++st) {
        const int ql = st * 7 + ps;                              // point inside the round
        if (lane < 63 && ql < cnt) {
          int ro = F2_OUT + ql * F2_REC;
          asm volatile("" : "+v"(ro));
          const double* rec = stage + ro;
          // all LDS reads first (the out-tile writes below may alias them for the compiler)
          double fi[9], sr[9];
#pragma unroll
          for (int t = 0; t < 9; ++t) { fi[t] = rec[t]; sr[t] = rec[36 + t]; }
          const double Vc = rec[9 + cc], Wc = rec[27 + cc];
          const double U0 = rec[18 + LL], U1 = rec[21 + LL], U2 = rec[24 + LL];
          const double g0 = rec[45 + LL * 3 + 0], g1 = rec[45 + LL * 3 + 1], g2 = rec[45 + LL * 3 + 2];
          const double F0k = kk == 0 ? fi[0] : (kk == 1 ? fi[1] : fi[2]);
          const double F1k = kk == 0 ? fi[3] : (kk == 1 ? fi[4] : fi[5]);
          const double F2k = kk == 0 ? fi[6] : (kk == 1 ? fi[7] : fi[8]);
          double x[9];
#pragma unroll
          for (int r = 0; r < 9; ++r) {
            const int i = TI[r], Jx = TJ[r];
            const double Ui = i == 0 ? U0 : (i == 1 ? U1 : U2);
            const double FJk = Jx == 0 ? F0k : (Jx == 1 ? F1k : F2k);
            const double gJ = Jx == 0 ? g0 : (Jx == 1 ? g1 : g2);
            const double mi = i == 0 ? mk0 : (i == 1 ? mk1 : mk2);
            double t = Vc * fi[Jx * 3 + i];
            t += Wc * sr[r];
            t += Ui * FJk;
            t += mi * gJ;
            x[r] = t;
          }
          double* o = outt + ql * 81 + cc;
#pragma unroll
          for (int r = 0; r < 9; ++r) o[r * 9] = x[r];
        }
      }
      wave_lds_sync();
      {
        int nv = npts - p0;                                      // valid points of this round
        nv = nv < 0 ? 0 : (nv > cnt ? cnt : nv);
        const int nent = nv * 81;                                // wave-uniform
        double* gct = ct + (base + p0) * 81;                     // 16 B aligned: (base + p0) * 81 is even
        const double2_t* o2 = reinterpret_cast<const double2_t*>(outt);
        constexpr int NIT = F2_NIT;
        double2_t v[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) v[it] = o2[it * WAVE + lane];   // out-tile is padded to NIT KiB
        double2_t* g2p = reinterpret_cast<double2_t*>(gct) + lane;
        // the two shapes every full tile consists of: straight-line stores, no per-KiB bookkeeping
        constexpr int E_FULL = F2_PPR * 81, E_LAST = (WAVE % F2_PPR) * 81;
        if (nent == E_FULL) {
#pragma unroll
          for (int it = 0; it < E_FULL / (2 * WAVE); ++it) stream_store<0>(g2p + it * WAVE, v[it]);
          if constexpr (E_FULL % (2 * WAVE) != 0) {
            static_assert(E_FULL % 2 == 0, "whole 16 B elements");
            if (lane < (E_FULL % (2 * WAVE)) / 2) stream_store<0>(g2p + (E_FULL / (2 * WAVE)) * WAVE, v[E_FULL / (2 * WAVE)]);
          }
        } else if (E_LAST > 0 && nent == E_LAST) {
#pragma unroll
          for (int it = 0; it < E_LAST / (2 * WAVE); ++it) stream_store<0>(g2p + it * WAVE, v[it]);
          if constexpr (E_LAST % (2 * WAVE) != 0) {
            static_assert(E_LAST % 2 == 0, "whole 16 B elements");
            if (lane < (E_LAST % (2 * WAVE)) / 2) stream_store<0>(g2p + (E_LAST / (2 * WAVE)) * WAVE, v[E_LAST / (2 * WAVE)]);
          }
        } else {   // ragged tile: element-wise bounds
#pragma unroll
          for (int it = 0; it < NIT; ++it) {
            const int e0 = (it * WAVE + lane) * 2;
            if ((it + 1) * 2 * WAVE <= nent) {                     // scalar branch: whole KiB valid
              stream_store<0>(reinterpret_cast<double2_t*>(gct + e0), v[it]);
            } else if (e0 + 1 < nent) {
              stream_store<0>(reinterpret_cast<double2_t*>(gct + e0), v[it]);
            } else if (e0 < nent) {
              stream_store<0>(gct + e0, v[it].x);
            }
          }
        }
      }
      wave_lds_sync();
    }
  }
  // the workgroup reduction borrows the first words of every wave's own region (the tile loop is over)
  store_block_stats(stats, c_plastic, c_notconv, c_nan, c_maxit, reinterpret_cast<unsigned long long*>(lds_all), F2_LDS_PER_WAVE);
}

}  // namespace dxm
