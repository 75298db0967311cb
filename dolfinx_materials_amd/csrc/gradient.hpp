// Gradient evaluation at the Gauss points on the device, for first-order hexahedra and tetrahedra: the step
// immediately BEFORE the hot path (reference: QuadratureExpression.eval -> fem.Expression.eval,
// dolfinx_materials/quadrature_function.py:45-51, called from quadrature_map.py:247-253), so that
// only the displacement vector (24 B/node) crosses PCIe instead of the strain array (48 B/point).
//
// One thread per Gauss point: gathers the 8 nodes of its cell (coordinates and displacements;
// the 8 threads of a cell hit the same lines), builds the isoparametric Jacobian, and writes the
// Mandel strain (6) or the deformation gradient F = I + grad u (9) in the AoS layout the
// constitutive kernels consume.  Memory-light next to them (the mesh is read through L2).
#pragma once
#include "dxm_common.hpp"

namespace dxm {

// reference corner signs of the trilinear hexahedron, node order
// (-,-,-) (+,-,-) (+,+,-) (-,+,-) (-,-,+) (+,-,+) (+,+,+) (-,+,+)
__device__ __constant__ const signed char HEX_SX[8] = {-1, 1, 1, -1, -1, 1, 1, -1};
__device__ __constant__ const signed char HEX_SY[8] = {-1, -1, 1, 1, -1, -1, 1, 1};
__device__ __constant__ const signed char HEX_SZ[8] = {-1, -1, -1, -1, 1, 1, 1, 1};

struct QuadPoints { int nqp; double xi[27][3]; };

// kind 0: Mandel strain (6)  [utils.py:146-165];  kind 1: F = I + grad u (9) [utils.py:168-190]
template <int KIND>
__global__ void __launch_bounds__(256)
hex8_gradient_kernel(const double* __restrict__ coords, const int32_t* __restrict__ conn,
                     const double* __restrict__ u, const int64_t ncells, const QuadPoints qp,
                     double* __restrict__ grad) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t npts = ncells * qp.nqp;
  if (gid >= npts) return;
  const int64_t cell = gid / qp.nqp;
  const int q = (int)(gid - cell * qp.nqp);
  const double x = qp.xi[q][0], y = qp.xi[q][1], z = qp.xi[q][2];
  double Jm[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // Jm[a][d] = dX_a / dxi_d
  double dN[8][3];
  int32_t nd[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    nd[m] = conn[cell * 8 + m];
    const double sx = HEX_SX[m], sy = HEX_SY[m], sz = HEX_SZ[m];
    dN[m][0] = 0.125 * sx * (1 + sy * y) * (1 + sz * z);
    dN[m][1] = 0.125 * sy * (1 + sx * x) * (1 + sz * z);
    dN[m][2] = 0.125 * sz * (1 + sx * x) * (1 + sy * y);
    const double X0 = coords[3 * (int64_t)nd[m]], X1 = coords[3 * (int64_t)nd[m] + 1], X2 = coords[3 * (int64_t)nd[m] + 2];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      Jm[0 + d] += X0 * dN[m][d];
      Jm[3 + d] += X1 * dN[m][d];
      Jm[6 + d] += X2 * dN[m][d];
    }
  }
  // inverse of Jm: Ji[d][a] = dxi_d / dX_a
  double Ji[9];
  {
    const double c00 = Jm[4] * Jm[8] - Jm[5] * Jm[7], c01 = Jm[5] * Jm[6] - Jm[3] * Jm[8], c02 = Jm[3] * Jm[7] - Jm[4] * Jm[6];
    const double idet = 1.0 / (Jm[0] * c00 + Jm[1] * c01 + Jm[2] * c02);
    Ji[0] = c00 * idet; Ji[3] = c01 * idet; Ji[6] = c02 * idet;
    Ji[1] = (Jm[2] * Jm[7] - Jm[1] * Jm[8]) * idet;
    Ji[4] = (Jm[0] * Jm[8] - Jm[2] * Jm[6]) * idet;
    Ji[7] = (Jm[1] * Jm[6] - Jm[0] * Jm[7]) * idet;
    Ji[2] = (Jm[1] * Jm[5] - Jm[2] * Jm[4]) * idet;
    Ji[5] = (Jm[2] * Jm[3] - Jm[0] * Jm[5]) * idet;
    Ji[8] = (Jm[0] * Jm[4] - Jm[1] * Jm[3]) * idet;
  }
  double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // H[i][j] = du_i / dX_j
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    double g[3];  // physical gradient of shape function m
#pragma unroll
    for (int a = 0; a < 3; ++a) g[a] = dN[m][0] * Ji[0 + a] + dN[m][1] * Ji[3 + a] + dN[m][2] * Ji[6 + a];
    const double u0 = u[3 * (int64_t)nd[m]], u1 = u[3 * (int64_t)nd[m] + 1], u2 = u[3 * (int64_t)nd[m] + 2];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      H[0 + a] += u0 * g[a];
      H[3 + a] += u1 * g[a];
      H[6 + a] += u2 * g[a];
    }
  }
  if constexpr (KIND == 0) {
    const double r = 0.70710678118654752440;  // sqrt(2) * (1/2)
    double* o = grad + gid * 6;
    o[0] = H[0]; o[1] = H[4]; o[2] = H[8];
    o[3] = r * (H[1] + H[3]); o[4] = r * (H[2] + H[6]); o[5] = r * (H[5] + H[7]);
  } else {
    double* o = grad + gid * 9;
    o[0] = 1.0 + H[0]; o[1] = 1.0 + H[4]; o[2] = 1.0 + H[8];
    o[3] = H[1]; o[4] = H[3]; o[5] = H[2]; o[6] = H[6]; o[7] = H[5]; o[8] = H[7];
  }
}

// First-order tetrahedra (affine): the displacement gradient is constant per cell,
//   H = sum_m u_m (x) grad N_m,  grad N from the inverse of the edge matrix [X1-X0, X2-X0, X3-X0].
// One thread per Gauss point (the nqp points of a cell repeat the cell value, as a dolfinx
// quadrature Function of degree > 1 would hold it).
template <int KIND>
__global__ void __launch_bounds__(256)
tet4_gradient_kernel(const double* __restrict__ coords, const int32_t* __restrict__ conn,
                     const double* __restrict__ u, const int64_t ncells, const int nqp,
                     double* __restrict__ grad) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= ncells * nqp) return;
  const int64_t cell = gid / nqp;
  int64_t nd[4];
  double X[4][3], U[4][3];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    nd[m] = conn[cell * 4 + m];
#pragma unroll
    for (int a = 0; a < 3; ++a) { X[m][a] = coords[3 * nd[m] + a]; U[m][a] = u[3 * nd[m] + a]; }
  }
  // A[a][d] = X_{d+1}[a] - X_0[a]  (dX_a / dxi_d for the reference tetrahedron)
  double A[9];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int d = 0; d < 3; ++d) A[a * 3 + d] = X[d + 1][a] - X[0][a];
  double Ai[9];  // Ai[d][a] = dxi_d / dX_a
  {
    const double c00 = A[4] * A[8] - A[5] * A[7], c01 = A[5] * A[6] - A[3] * A[8], c02 = A[3] * A[7] - A[4] * A[6];
    const double idet = 1.0 / (A[0] * c00 + A[1] * c01 + A[2] * c02);
    Ai[0] = c00 * idet; Ai[3] = c01 * idet; Ai[6] = c02 * idet;
    Ai[1] = (A[2] * A[7] - A[1] * A[8]) * idet;
    Ai[4] = (A[0] * A[8] - A[2] * A[6]) * idet;
    Ai[7] = (A[1] * A[6] - A[0] * A[7]) * idet;
    Ai[2] = (A[1] * A[5] - A[2] * A[4]) * idet;
    Ai[5] = (A[2] * A[3] - A[0] * A[5]) * idet;
    Ai[8] = (A[0] * A[4] - A[1] * A[3]) * idet;
  }
  // H[i][a] = sum_d (U_{d+1}[i] - U_0[i]) Ai[d][a]
  double H[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int a = 0; a < 3; ++a)
      H[i * 3 + a] = (U[1][i] - U[0][i]) * Ai[0 + a] + (U[2][i] - U[0][i]) * Ai[3 + a] + (U[3][i] - U[0][i]) * Ai[6 + a];
  if constexpr (KIND == 0) {
    const double r = 0.70710678118654752440;
    double* o = grad + gid * 6;
    o[0] = H[0]; o[1] = H[4]; o[2] = H[8];
    o[3] = r * (H[1] + H[3]); o[4] = r * (H[2] + H[6]); o[5] = r * (H[5] + H[7]);
  } else {
    double* o = grad + gid * 9;
    o[0] = 1.0 + H[0]; o[1] = 1.0 + H[4]; o[2] = 1.0 + H[8];
    o[3] = H[1]; o[4] = H[3]; o[5] = H[2]; o[6] = H[6]; o[7] = H[5]; o[8] = H[7];
  }
}

}  // namespace dxm
