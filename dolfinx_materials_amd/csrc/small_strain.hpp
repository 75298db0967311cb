// Small-strain constitutive updates for gfx950: isotropic elasticity, J2 plasticity with linear or
// Voce isotropic hardening.  One fused kernel per law: trial state, yield test, local Newton,
// stress, consistent tangent and state write-back.
//
// Replaces, per Gauss point, what the reference obtains from
//   vmap(jacfwd(behavior.constitutive_update))      dolfinx_materials/jaxmat.py:147-164
// Arithmetic spec (the only in-tree statement of the return mapping):
//   tests/mfront/IsotropicLinearHardeningPlasticity.mfront:49-77
//   python_materials/elasticity.py:12-24 (elastic), tests/test_FeFp_jax.py:14-15 (Voce law)
//
// Mapping (HBM-bound streaming kernel, ~1 flop/B fp64, no MFMA):
//   * one thread per Gauss point, one wave per tile of 64 points, grid-stride over tiles;
//   * the AoS boundary arrays (strain (N,6) in, stress (N,6) and tangent (N,36) out: the memory
//     of the dolfinx quadrature Functions) are moved with 16 B-per-lane, fully coalesced
//     accesses and re-distributed between lanes through a wave-private LDS region (no s_barrier);
//   * persistent state (p, eps_p) is SoA in HBM: 8 B-per-lane coalesced loads/stores;
//   * the 6x6 tangent is never materialised per thread: each point stages 9 doubles
//     (c1, c2, c3, n[6]) in LDS and the whole wave then evaluates
//         Ct = c1 1x1 + c2 I + c3 n x n
//     entry by entry in output order, so the dominant 288 B/point stream leaves as contiguous
//     1 KiB wave stores (SYM = true: only the 21 entries of the upper triangle, 168 B/point).
#pragma once
#include "dxm_common.hpp"
#include "gradient.hpp"

namespace dxm {

enum { LAW_ELASTIC = 0, LAW_J2_LINEAR = 1, LAW_J2_VOCE = 2 };

// state slots (SoA, leading dimension ld): 0 = p, 1..6 = eps_p (Mandel)
constexpr int SS_NSLOTS = 7;

constexpr int SS_STAGE = 64 * 6;   // doubles per wave: strain in / stress out staging
constexpr int SS_COEF = 64 * 9;    // doubles per wave: (c1,c2,c3,n0..n5) per point
constexpr int SS_LDS_PER_WAVE = SS_STAGE + SS_COEF;
static_assert(8 * HEX_FUSED_REC <= SS_COEF, "the 8 cell records of a fused tile live in the coefficient region");

template <int LAW>
__device__ __forceinline__ double hardening_R(const LawParams& prm, double p) {
  if constexpr (LAW == LAW_J2_LINEAR) {
    return prm.sig0 + prm.h1 * p;
  } else {
#ifdef DXM_CUSTOM_HARDENING
    return custom_R(prm, p);
#else
    return prm.sig0 + DXM_MUL(prm.h1 - prm.sig0, 1.0 - exp(DXM_MUL(-prm.h2, p)));
#endif
  }
}
template <int LAW>
__device__ __forceinline__ double hardening_dR(const LawParams& prm, double p) {
  if constexpr (LAW == LAW_J2_LINEAR) {
    return prm.h1;
  } else {
#ifdef DXM_CUSTOM_HARDENING
    return custom_dR(prm, p);
#else
    return DXM_MUL((prm.h1 - prm.sig0) * prm.h2, exp(DXM_MUL(-prm.h2, p)));
#endif
  }
}

// GRAD = 0: the strain comes from the (N,6) array `eps`.  GRAD = 1: it is evaluated in the kernel from
// the displacement vector of a hex8 mesh with 8 Gauss points per cell (`src`; `eps` unused): lane
// (cell c of the tile, corner k) gathers one node into a wave-private LDS record, every lane then
// evaluates the isoparametric gradient at its own point -- the strain array (48 B/point written by the
// gradient kernel and read back here) never exists.  GRAD = 2: tet4 mesh, every lane gathers the 4
// nodes of its own cell (the gradient is constant per cell).  GRAD = 3: straight-sided simplices with a Lagrange
// displacement of any order (tet10, tri6 in plane strain, ...: gradient.hpp::simplex_disp_grad), also gathered per lane.
// Entries (i, j) and (i, j+1) of Ct = c1 1x1 + c2 I + c3 n x n from the nine staged numbers cf = (c1, c2, c3, n[6]).
// k3 (ni nj), not (k3 ni) nj: the product ni nj commutes bit for bit, so the block is EXACTLY symmetric and can be
// rebuilt from its coefficients with this very expression elsewhere (host path: dxmat.hip::expand_coef_tangent;
// after an all-gather of coefficients: expand_tangent_kernel below).
__device__ __forceinline__ double2_t tangent_pair(const double* cf, int i, int j) {
  const double k1 = cf[0], k2 = cf[1], k3 = cf[2];
  const double ni = cf[3 + i], nj0 = cf[3 + j], nj1 = cf[4 + j];
  const double t0 = ((i < 3 && j < 3) ? k1 : 0.0) + ((i == j) ? k2 : 0.0);
  const double t1 = ((i < 3 && j + 1 < 3) ? k1 : 0.0) + ((i == j + 1) ? k2 : 0.0);
  double2_t v;
  v.x = t0 + k3 * (ni * nj0);
  v.y = t1 + k3 * (ni * nj1);
  return v;
}

// TL: layout of the tangent output.  TL_FULL the 6x6 block, row-major (what jacobian_flatten holds,
// quadrature_map.py:83-105); TL_SYM its 21 upper-triangle entries; TL_COEF the 9 coefficients
// (c1, c2, c3, n[6]) of Ct = c1 1x1 + c2 I + c3 n x n themselves (72 B/point); TL_PACK4 only (c1, c2, c3, w): the flow
// direction is n = dev(sigma) w by definition (below), so a consumer that receives the stress anyway -- the host-buffer
// form, dxmat.hip::expand_pack4_tangent -- rebuilds n and the block from 32 B/point, bit for bit.
enum { TL_FULL = 0, TL_SYM = 1, TL_COEF = 2, TL_PACK4 = 3 };
constexpr double SS_THIRD = 1.0 / 3.0;

template <int LAW, int TL, int GRAD = 0>
__global__ void __launch_bounds__(BLOCK, 4)  // 4 waves per SIMD (the launcher pads the LDS of the J2 kernels so that a fifth never fits)
small_strain_kernel(const LawParams prm, const int64_t n, const double* __restrict__ eps,
                    const double* __restrict__ s0, double* __restrict__ s1, const int64_t ld,
                    double* __restrict__ sig, double* __restrict__ ct,
                    BlockStats* __restrict__ stats, const MeshSource src) {
  __shared__ __attribute__((aligned(16))) double lds_all[WAVES_PER_BLOCK * SS_LDS_PER_WAVE];
  __shared__ unsigned long long red[4 * WAVES_PER_BLOCK];

  int lane = threadIdx.x & (WAVE - 1);
  // (not made scalar with readfirstlane as in fefp.hpp: measured in one process, three handles each, that build is
  // 0.45 % slower -- 96 instead of 103 VGPRs makes a fifth wave per SIMD resident, which this kernel does not like)
  const int wid = threadIdx.x >> 6;
  double* stage = lds_all + wid * SS_LDS_PER_WAVE;
  double* coef = stage + SS_STAGE;
  double2_t* stage2 = reinterpret_cast<double2_t*>(stage);

  const int64_t ntiles = (n + WAVE - 1) / WAVE;
  const int64_t tile_stride = (int64_t)gridDim.x * WAVES_PER_BLOCK;

  unsigned long long c_plastic = 0, c_notconv = 0, c_nan = 0, c_maxit = 0;

  const double lambda = prm.lambda, mu = prm.mu;

  for (int64_t tile = (int64_t)blockIdx.x * WAVES_PER_BLOCK + wid; tile < ntiles;
       tile += tile_stride) {
    const int64_t base = tile * WAVE;
    const int npts = (n - base) < WAVE ? (int)(n - base) : WAVE;
    if constexpr (LAW == LAW_J2_VOCE) {
      // the lane index is re-read through an opaque copy once per tile: per-lane invariants hoisted out of the
      // tile loop otherwise push the Voce kernels over their 128-register budget (2-8 spilled VGPRs)
      asm volatile("" : "+v"(lane));
      lane &= WAVE - 1;
    }
    const bool valid = lane < npts;
    const int64_t gi = base + lane;

    double e[6];
    double p_n = 0.0, ep[6] = {0, 0, 0, 0, 0, 0};
    if constexpr (GRAD == 0) {
      // ---- 1. coalesced strain load (3 x 1 KiB per wave) into LDS ------------------------------
      {
        const double2_t* gsrc = reinterpret_cast<const double2_t*>(eps + base * 6);
        double2_t v[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const int idx = k * WAVE + lane;
          v[k] = (idx < npts * 3) ? stream_load<2>(gsrc + idx) : double2_t{0.0, 0.0};
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) stage2[k * WAVE + lane] = v[k];
      }
      // ---- old state, SoA (issued before the LDS round trip completes) -------------------------
      if constexpr (LAW != LAW_ELASTIC) {
        if (valid) {
          p_n = stream_load<3>(s0 + gi);
#pragma unroll
          for (int c = 0; c < 6; ++c) ep[c] = stream_load<3>(s0 + (int64_t)(1 + c) * ld + gi);
        }
      }
      wave_lds_sync();
      // ---- 2. my point's strain ----------------------------------------------------------------
      {
        const double2_t a = stage2[lane * 3 + 0], b = stage2[lane * 3 + 1], c = stage2[lane * 3 + 2];
        e[0] = a.x; e[1] = a.y; e[2] = b.x; e[3] = b.y; e[4] = c.x; e[5] = c.y;
      }
      wave_lds_sync();  // staging region is reused for the stress below
    } else {
      double Hd[9];
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
        // ---- 2'. displacement gradient at my Gauss point (point q = lane & 7 of cell lane >> 3) -----
        {
          const double2_t* rec = reinterpret_cast<const double2_t*>(coef + (lane >> 3) * HEX_FUSED_REC);
          auto node = [&](int m, double* X, double* U) {
            const double2_t a = rec[m * 3], b = rec[m * 3 + 1], c = rec[m * 3 + 2];
            X[0] = a.x; X[1] = a.y; X[2] = b.x;
            U[0] = b.y; U[1] = c.x; U[2] = c.y;
          };
          const int q = lane & 7;
          if (valid) {
            hex8_disp_grad(src.xi[q][0], src.xi[q][1], src.xi[q][2], node, Hd);
          } else {
#pragma unroll
            for (int k = 0; k < 9; ++k) Hd[k] = 0.0;
          }
        }
        wave_lds_sync();  // the coefficient region is rewritten in step 5
      } else {
        if (valid) {
          const int64_t cell = (src.point0 + gi) / src.nqp;
          if constexpr (GRAD == 2) tet4_cell_disp_grad(src.coords, src.conn, src.u, cell, Hd);
          else simplex_disp_grad(src, cell, (int)(src.point0 + gi - cell * src.nqp), Hd);
        } else {
#pragma unroll
          for (int k = 0; k < 9; ++k) Hd[k] = 0.0;
        }
      }
      {
        const double r = 0.70710678118654752440;
        e[0] = Hd[0]; e[1] = Hd[4]; e[2] = Hd[8];
        e[3] = r * (Hd[1] + Hd[3]); e[4] = r * (Hd[2] + Hd[6]); e[5] = r * (Hd[5] + Hd[7]);
      }
      // old state only now: 14 registers fewer live through the gradient evaluation
      if constexpr (LAW != LAW_ELASTIC) {
        if (valid) {
          p_n = stream_load<3>(s0 + gi);
#pragma unroll
          for (int c = 0; c < 6; ++c) ep[c] = stream_load<3>(s0 + (int64_t)(1 + c) * ld + gi);
        }
      }
    }

    // ---- 3. constitutive update --------------------------------------------------------------
    double c1 = lambda, c2 = 2.0 * mu, c3 = 0.0;
    double wn = 0.0;   // n = dev(sigma) wn: the direction the tangent is built with (0 for an elastic point)
    double p_new = p_n;
    if constexpr (LAW != LAW_ELASTIC) {
      // trial elastic strain                                   mfront:52  eel += deto
#pragma unroll
      for (int c = 0; c < 6; ++c) e[c] -= ep[c];
      const double tr = e[0] + e[1] + e[2];
      const double third = tr / 3.0;
      double se[6];
      se[0] = 2.0 * mu * (e[0] - third);
      se[1] = 2.0 * mu * (e[1] - third);
      se[2] = 2.0 * mu * (e[2] - third);
      se[3] = 2.0 * mu * e[3];
      se[4] = 2.0 * mu * e[4];
      se[5] = 2.0 * mu * e[5];                                // mfront:53
      double nrm2 = 0.0;
#pragma unroll
      for (int c = 0; c < 6; ++c) nrm2 += se[c] * se[c];
      const double seq = sqrt(1.5 * nrm2);                    // mfront:54
      const double f = seq - hardening_R<LAW>(prm, p_n);      // mfront:55
      if (f > 0.0) {
        double dp;
        unsigned iters = 0;
        if constexpr (LAW == LAW_J2_LINEAR) {
          dp = f / (prm.h1 + 3.0 * mu);                       // mfront:62-63
        } else {
          // r(dp) = seq - 3 mu dp - R(p_n + dp) = 0, monotone Newton from dp = 0
          // tolerance relative to the larger of the yield stress and the trial stress the residual is made of
          dp = 0.0;
          const double tolp = fmax(prm.tol, prm.rtol * seq);
          for (int it = 0;; ++it) {
            const double r = seq - 3.0 * mu * dp - hardening_R<LAW>(prm, p_n + dp);
            if (fabs(r) <= tolp) break;
            if (it >= prm.maxit) { if (valid) ++c_notconv; break; }
            const double dr = -3.0 * mu - hardening_dR<LAW>(prm, p_n + dp);
            dp -= r / dr;
            ++iters;
          }
        }
        const double iseq = 1.0 / seq;
        double nn[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) nn[c] = 1.5 * se[c] * iseq;   // mfront:61
        const double beta = dp * iseq;
        // The return is radial: dev(sigma) = (1 - 3 mu beta) s_e, so n = 3/2 s_e / seq = dev(sigma) wn with
        // wn = 3/2 / (seq (1 - 3 mu beta)).  The TANGENT is built with n in that form (step 5), so that whoever holds
        // the stress and wn holds n, to the bit.
        {
          const double rho = 1.0 - 3.0 * mu * beta;
          wn = rho > 0.0 ? 1.5 * iseq / rho : 0.0;   // (a division: with the 5-instruction reciprocal the fused tet4 Voce variant spills 2 registers)
          // rho = R(p) / seq of the returned state: <= 0 only for a yield stress that is not positive there (a softening law
          // driven to zero, an overshooting iterate).  The direction is then undefined (wn = 0 drops the n x n term): reported
          // as a point that did not converge, never silently.  Linear hardening can get there with H < 0 only: one scalar
          // compare on the kernel's parameters keeps the per-lane test out of the H >= 0 launches (the headline)
          if constexpr (LAW != LAW_J2_LINEAR) { if (valid && !(rho > 0.0)) ++c_notconv; }
          else if (prm.h1 < 0.0) { if (valid && !(rho > 0.0)) ++c_notconv; }
        }
        const double gamma = 1.0 / (hardening_dR<LAW>(prm, p_n + dp) + 3.0 * mu);
        // Dt = lambda IxI + 2mu Id - 4mu^2 [beta (M - n^n) + gamma n^n]      mfront:66-69
        c1 = lambda + 2.0 * mu * mu * beta;
        c2 = 2.0 * mu - 6.0 * mu * mu * beta;
        c3 = 4.0 * mu * mu * (beta - gamma);
        p_new = p_n + dp;
#pragma unroll
        for (int c = 0; c < 6; ++c) {
          ep[c] += dp * nn[c];
          e[c] -= dp * nn[c];                                  // mfront:64
        }
        if (valid) {
          ++c_plastic;
          c_maxit = iters > c_maxit ? iters : c_maxit;
        }
      }
    }
    // sigma = lambda tr(eel) 1 + 2 mu eel                                      mfront:76
    const double ltr = lambda * (e[0] + e[1] + e[2]);
    double s[6];
    s[0] = ltr + 2.0 * mu * e[0];
    s[1] = ltr + 2.0 * mu * e[1];
    s[2] = ltr + 2.0 * mu * e[2];
    s[3] = 2.0 * mu * e[3];
    s[4] = 2.0 * mu * e[4];
    s[5] = 2.0 * mu * e[5];
    {
      // stress, p and what the tangent is made of (quadrature_map.py:322-324 asserts on flux, state and Ct): a hardening
      // slope that is not finite at the returned state leaves the stress finite and c3 not
      const double chk = s[0] + s[1] + s[2] + s[3] + s[4] + s[5] + p_new + ((c1 + c2) + (c3 + wn));
      if (valid && !(fabs(chk) <= 1.79769313486231570e308)) ++c_nan;
    }

    // ---- 4. new state, SoA -------------------------------------------------------------------
    if constexpr (LAW != LAW_ELASTIC) {
      if (valid) {
        stream_store<1>(s1 + gi, p_new);
#pragma unroll
        for (int c = 0; c < 6; ++c) stream_store<1>(s1 + (int64_t)(1 + c) * ld + gi, ep[c]);
      }
    }

    // ---- 5. stage stress and tangent coefficients in LDS --------------------------------------
    stage2[lane * 3 + 0] = double2_t{s[0], s[1]};
    stage2[lane * 3 + 1] = double2_t{s[2], s[3]};
    stage2[lane * 3 + 2] = double2_t{s[4], s[5]};
    if constexpr (LAW != LAW_ELASTIC && TL == TL_PACK4) {
      double2_t* c4 = reinterpret_cast<double2_t*>(coef) + lane * 2;
      c4[0] = double2_t{c1, c2};
      c4[1] = double2_t{c3, wn};
    } else if constexpr (LAW != LAW_ELASTIC) {
      double* cf = coef + lane * 9;
      cf[0] = c1; cf[1] = c2; cf[2] = c3;
      // n = dev(sigma) wn, every operation individually rounded (the host rebuilds it with the same three lines)
      const double third = opaque((s[0] + s[1] + s[2]) * SS_THIRD);
      cf[3] = (s[0] - third) * wn; cf[4] = (s[1] - third) * wn; cf[5] = (s[2] - third) * wn;
      cf[6] = s[3] * wn; cf[7] = s[4] * wn; cf[8] = s[5] * wn;
    }
    wave_lds_sync();

    // ---- 6. coalesced stress store (3 x 1 KiB) -----------------------------------------------
    {
      double2_t* gdst = reinterpret_cast<double2_t*>(sig + base * 6);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int idx = k * WAVE + lane;
        if (idx < npts * 3) {
          stream_store<0>(gdst + idx, stage2[idx]);
        }
      }
    }
    // ---- 7. coalesced tangent store: entry pair (i, j..j+1) of point q ---------------------------
    if constexpr (TL == TL_PACK4) {
      static_assert(LAW != LAW_ELASTIC, "the elastic tangent is a constant: nothing to write");
      if (npts == WAVE) {   // 64 x 4 doubles: two 1 KiB wave stores
        double2_t* gct = reinterpret_cast<double2_t*>(ct + base * 4);
        const double2_t* c4 = reinterpret_cast<const double2_t*>(coef);
        stream_store<0>(gct + lane, c4[lane]);
        stream_store<0>(gct + WAVE + lane, c4[WAVE + lane]);
      } else {
        double2_t* gct = reinterpret_cast<double2_t*>(ct + base * 4);
        const double2_t* c4 = reinterpret_cast<const double2_t*>(coef);
        if (lane < npts * 2) stream_store<0>(gct + lane, c4[lane]);
        if (WAVE + lane < npts * 2) stream_store<0>(gct + WAVE + lane, c4[WAVE + lane]);
      }
    } else if constexpr (TL == TL_COEF) {
      // the staged coefficients as they are: 64 x 9 doubles, contiguous (4.5 KiB per tile)
      static_assert(LAW != LAW_ELASTIC, "the elastic tangent is a constant: nothing to write");
      if (npts == WAVE) {
        double2_t* gct = reinterpret_cast<double2_t*>(ct + base * 9);
        const double2_t* c2 = reinterpret_cast<const double2_t*>(coef);
#pragma unroll
        for (int k = 0; k < 5; ++k) {
          const int idx = k * WAVE + lane;
          if (idx < 288) stream_store<0>(gct + idx, c2[idx]);
        }
      } else {
        double* gct = ct + base * 9;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
          const int idx = k * WAVE + lane;
          if (idx < npts * 9) stream_store<0>(gct + idx, coef[idx]);
        }
      }
    } else if constexpr (TL == TL_FULL) {
      // full 6x6, row-major (quadrature_map.py:83-105): 18 x 1 KiB per tile, 18 pairs per point.
      // The (q, i, j) of a lane's pair advance by a fixed pattern from one iteration to the next
      // (64 pairs = 3 points + 10 pairs), so they are carried instead of re-divided; for a full tile
      // (wave-uniform) the stores are unpredicated and the LDS reads of 3 iterations are issued
      // together, ahead of the arithmetic (groups of 3: larger groups spill at 128 VGPRs).
      double2_t* gct = reinterpret_cast<double2_t*>(ct + base * 36);
      auto entry = [&](int q, int i, int j) -> double2_t {
        double2_t v;
        if constexpr (LAW == LAW_ELASTIC) {
          v.x = ((i < 3 && j < 3) ? lambda : 0.0) + ((i == j) ? 2.0 * mu : 0.0);
          v.y = ((i < 3 && j + 1 < 3) ? lambda : 0.0) + ((i == j + 1) ? 2.0 * mu : 0.0);
        } else {
          v = tangent_pair(coef + q * 9, i, j);
        }
        return v;
      };
      if (npts == WAVE) {
#pragma unroll 1  // a fully unrolled loop gets its (loop-invariant) index arithmetic hoisted out of the
                  // tile loop: 18 x (q, i, j) live across tiles, which spills at 128 VGPRs
        for (int g = 0; g < 6; ++g) {
          double2_t v[3];
#pragma unroll
          for (int u = 0; u < 3; ++u) {
            const int k = (g * 3 + u) * WAVE + lane;
            const int q = k / 18;
            const int r = k - q * 18;
            const int i = r / 3;
            v[u] = entry(q, i, (r - i * 3) * 2);
          }
#pragma unroll
          for (int u = 0; u < 3; ++u) stream_store<0>(gct + (g * 3 + u) * WAVE + lane, v[u]);
        }
      } else {
        const int lim = npts * 18;
#pragma unroll 1
        for (int it = 0; it < 18; ++it) {
          const int k = it * WAVE + lane;
          const int q = k / 18;
          const int r = k - q * 18;
          const int i = r / 3;
          if (k < lim) stream_store<0>(gct + k, entry(q, i, (r - i * 3) * 2));
        }
      }
    } else {
      // symmetric-packed: the 21 entries (i <= j) of the upper triangle, row-major, per point
      // (the J2 tangent is symmetric).  10.5 x 1 KiB per tile.
      constexpr unsigned long long IP = 0x0ull | (1ull << 18) | (1ull << 21) | (1ull << 24) | (1ull << 27) | (1ull << 30) |
                                        (2ull << 33) | (2ull << 36) | (2ull << 39) | (2ull << 42) | (3ull << 45) |
                                        (3ull << 48) | (3ull << 51) | (4ull << 54) | (4ull << 57) | (5ull << 60);
      constexpr unsigned long long JP = (0ull << 0) | (1ull << 3) | (2ull << 6) | (3ull << 9) | (4ull << 12) | (5ull << 15) |
                                        (1ull << 18) | (2ull << 21) | (3ull << 24) | (4ull << 27) | (5ull << 30) |
                                        (2ull << 33) | (3ull << 36) | (4ull << 39) | (5ull << 42) | (3ull << 45) |
                                        (4ull << 48) | (5ull << 51) | (4ull << 54) | (5ull << 57) | (5ull << 60);
      double* gct = ct + base * 21;
      const int lim = npts * 21;
#pragma unroll 4
      for (int it = 0; it < 11; ++it) {
        const int e0 = (it * WAVE + lane) * 2;
        double v[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int e = e0 + u;
          const int q = e / 21;
          const int t = e - q * 21;
          const int i = (int)((IP >> (3 * t)) & 7ull), j = (int)((JP >> (3 * t)) & 7ull);
          double x;
          if constexpr (LAW == LAW_ELASTIC) {
            x = ((i < 3 && j < 3) ? lambda : 0.0) + ((i == j) ? 2.0 * mu : 0.0);
          } else {
            const double* cf = coef + (q < WAVE ? q : 0) * 9;
            x = (((i < 3 && j < 3) ? cf[0] : 0.0) + ((i == j) ? cf[1] : 0.0)) + cf[2] * (cf[3 + i] * cf[3 + j]);
          }
          v[u] = x;
        }
        if (e0 + 1 < lim) stream_store<0>(reinterpret_cast<double2_t*>(gct + e0), double2_t{v[0], v[1]});
        else if (e0 < lim) stream_store<0>(gct + e0, v[0]);
      }
    }
    wave_lds_sync();  // LDS regions are rewritten by the next tile
  }

  store_block_stats(stats, c_plastic, c_notconv, c_nan, c_maxit, red);
}

// coefficients (N, 9) -> full tangent (N, 36), both in HBM: what a rank runs after all-gathering the 72 B/point
// coefficient form of the J2 tangent instead of its 288 B/point block (sharding.allgather_tangent): 360 B/point of
// HBM traffic buy 216 B/point less on the xGMI links.  Same staging and store loops as step 7 of the update kernel.
__global__ void __launch_bounds__(BLOCK, 4)
expand_tangent_kernel(const int64_t n, const double* __restrict__ cin, double* __restrict__ ct) {
  __shared__ __attribute__((aligned(16))) double lds_all[WAVES_PER_BLOCK * SS_COEF];
  const int lane = threadIdx.x & (WAVE - 1);
  const int wid = threadIdx.x >> 6;
  double* coef = lds_all + wid * SS_COEF;
  const int64_t ntiles = (n + WAVE - 1) / WAVE;
  for (int64_t tile = (int64_t)blockIdx.x * WAVES_PER_BLOCK + wid; tile < ntiles; tile += (int64_t)gridDim.x * WAVES_PER_BLOCK) {
    const int64_t base = tile * WAVE;
    const int npts = (n - base) < WAVE ? (int)(n - base) : WAVE;
    if (npts == WAVE) {
      const double2_t* g = reinterpret_cast<const double2_t*>(cin + base * 9);
      double2_t v[5];
#pragma unroll
      for (int k = 0; k < 5; ++k) v[k] = (k * WAVE + lane < 288) ? g[k * WAVE + lane] : double2_t{0.0, 0.0};
#pragma unroll
      for (int k = 0; k < 5; ++k)
        if (k * WAVE + lane < 288) reinterpret_cast<double2_t*>(coef)[k * WAVE + lane] = v[k];
    } else {
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const int idx = k * WAVE + lane;
        coef[idx] = idx < npts * 9 ? cin[base * 9 + idx] : 0.0;
      }
    }
    wave_lds_sync();
    double2_t* gct = reinterpret_cast<double2_t*>(ct + base * 36);
    const int lim = npts * 18;
#pragma unroll 1
    for (int g = 0; g < 6; ++g) {
      double2_t v[3];
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int k = (g * 3 + u) * WAVE + lane;
        const int q = k / 18;
        const int r = k - q * 18;
        const int i = r / 3;
        v[u] = tangent_pair(coef + q * 9, i, (r - i * 3) * 2);
      }
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int k = (g * 3 + u) * WAVE + lane;
        if (k < lim) stream_store<0>(gct + k, v[u]);
      }
    }
    wave_lds_sync();
  }
}

// (stress (N, 6), (c1, c2, c3, w) (N, 4)) -> full tangent (N, 36): every lane forms the nine staged numbers of its point with the
// update kernel's own three lines (step 5), then the same store loops.  408 B/point of HBM traffic.
__global__ void __launch_bounds__(BLOCK, 4)
expand_pack4_kernel(const int64_t n, const double* __restrict__ sig, const double* __restrict__ cw, double* __restrict__ ct) {
  constexpr int PER_WAVE = 64 * 6 + 64 * 4;   // staging of a tile's stress rows and packs; the 64 x 9 coefficients reuse it
  static_assert(PER_WAVE >= SS_COEF, "the coefficient records live in the staging region");
  __shared__ __attribute__((aligned(16))) double lds_all[WAVES_PER_BLOCK * PER_WAVE];
  const int lane = threadIdx.x & (WAVE - 1);
  const int wid = threadIdx.x >> 6;
  double* coef = lds_all + wid * PER_WAVE;
  const int64_t ntiles = (n + WAVE - 1) / WAVE;
  for (int64_t tile = (int64_t)blockIdx.x * WAVES_PER_BLOCK + wid; tile < ntiles; tile += (int64_t)gridDim.x * WAVES_PER_BLOCK) {
    const int64_t base = tile * WAVE;
    const int npts = (n - base) < WAVE ? (int)(n - base) : WAVE;
    // stage the tile's stress rows (3 KiB) and packs (2 KiB) with 16 B-per-lane loads, then one point per lane
    double2_t* st2 = reinterpret_cast<double2_t*>(coef);            // [0, 192) pairs: stress, [192, 320): packs
    {
      const double2_t* g = reinterpret_cast<const double2_t*>(sig + base * 6);
      const double2_t* h = reinterpret_cast<const double2_t*>(cw + base * 4);
      double2_t v[5];
#pragma unroll
      for (int k = 0; k < 3; ++k) v[k] = (k * WAVE + lane < npts * 3) ? g[k * WAVE + lane] : double2_t{0.0, 0.0};
#pragma unroll
      for (int k = 0; k < 2; ++k) v[3 + k] = (k * WAVE + lane < npts * 2) ? h[k * WAVE + lane] : double2_t{0.0, 0.0};
#pragma unroll
      for (int k = 0; k < 5; ++k) st2[k * WAVE + lane] = v[k];
    }
    wave_lds_sync();
    double s[6], c1, c2, c3, wn;
    {
      const double2_t a = st2[lane * 3], b = st2[lane * 3 + 1], c = st2[lane * 3 + 2];
      s[0] = a.x; s[1] = a.y; s[2] = b.x; s[3] = b.y; s[4] = c.x; s[5] = c.y;
      const double2_t u = st2[192 + lane * 2], w = st2[192 + lane * 2 + 1];
      c1 = u.x; c2 = u.y; c3 = w.x; wn = w.y;
    }
    wave_lds_sync();
    {
      double* cf = coef + lane * 9;
      cf[0] = c1; cf[1] = c2; cf[2] = c3;
      const double third = opaque((s[0] + s[1] + s[2]) * SS_THIRD);
      cf[3] = (s[0] - third) * wn; cf[4] = (s[1] - third) * wn; cf[5] = (s[2] - third) * wn;
      cf[6] = s[3] * wn; cf[7] = s[4] * wn; cf[8] = s[5] * wn;
    }
    wave_lds_sync();
    double2_t* gct = reinterpret_cast<double2_t*>(ct + base * 36);
    const int lim = npts * 18;
#pragma unroll 1
    for (int g = 0; g < 6; ++g) {
      double2_t v[3];
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int k = (g * 3 + u) * WAVE + lane;
        const int q = k / 18;
        const int r = k - q * 18;
        const int i = r / 3;
        v[u] = tangent_pair(coef + q * 9, i, (r - i * 3) * 2);
      }
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int k = (g * 3 + u) * WAVE + lane;
        if (k < lim) stream_store<0>(gct + k, v[u]);
      }
    }
    wave_lds_sync();
  }
}

}  // namespace dxm
