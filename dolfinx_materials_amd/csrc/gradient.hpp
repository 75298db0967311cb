// Gradient evaluation at the Gauss points on the device, for first-order hexahedra and tetrahedra and for
// Lagrange elements of any order on straight-sided simplices (tet10 / tri6 of the reference's demos): the step
// immediately BEFORE the hot path (reference: QuadratureExpression.eval -> fem.Expression.eval,
// dolfinx_materials/quadrature_function.py:45-51, called from quadrature_map.py:247-253), so that
// only the displacement vector (24 B/node) crosses PCIe instead of the strain array (48 B/point).
//
// One thread per Gauss point.  Hexahedra with >= 4 points per cell: the block gathers the nodal data of
// its cells once into LDS (one (cell, corner) per thread) and the points read their cell's record
// from there; otherwise every thread gathers its 8 nodes itself (the points of a cell hit the same
// lines).  The thread builds the isoparametric Jacobian and writes the Mandel strain (6) or the
// deformation gradient F = I + grad u (9) in the AoS layout the constitutive kernels consume.
// 0.25 ms per 1e7 points staged vs 0.32 ms direct (profiles/archive/r01_bench_gradient_v2.jsonl).
#pragma once
#include "dxm_common.hpp"

namespace dxm {

// reference corner signs of the trilinear hexahedron, node order
// (-,-,-) (+,-,-) (+,+,-) (-,+,-) (-,-,+) (+,-,+) (+,+,+) (-,+,+)
__device__ __constant__ const signed char HEX_SX[8] = {-1, 1, 1, -1, -1, 1, 1, -1};
__device__ __constant__ const signed char HEX_SY[8] = {-1, -1, 1, 1, -1, -1, 1, 1};
__device__ __constant__ const signed char HEX_SZ[8] = {-1, -1, -1, -1, 1, 1, 1, 1};

struct QuadPoints { int nqp; double xi[27][3]; };

// What the fused displacement -> update kernels read instead of a gradient array.
//   kind 1: hex8 mesh with exactly 8 Gauss points per cell (one 64-point tile = 8 cells, one (cell, corner) per lane)
//   kind 2: tet4 mesh, any number of points per cell (every lane gathers the 4 nodes of its own cell)
//   kind 3: straight-sided simplices (tdim 2 or 3) with a Lagrange displacement of any order: geometry from the
//           tdim+1 vertices (`conn`), displacement through its own dofmap (nd dofs per cell) and the tabulated
//           reference derivatives of the nd shape functions at the nqp points (`dphi`, in device memory)
struct MeshSource {
  const double* coords;
  const int32_t* conn;
  const double* u;
  int64_t ncells;
  int64_t point0;   // first Gauss point of the launched range (chunked host path), else 0
  int32_t kind;
  int32_t nqp;
  double xi[8][3];  // hex8 only
  const int32_t* dofmap;   // kind 3: (ncells, nd)
  const double* dphi;      // kind 3: (nqp, nd, tdim), dphi[q][m][d] = d N_m / d xi_d at point q
  int32_t nd, tdim;
};
constexpr int HEX_FUSED_REC = 50;   // doubles per staged cell record (8 corners x 6, padded: see the staged kernel)

// Displacement gradient of one Gauss point of a trilinear hexahedron.  `node(m, X, U)` hands over
// coordinates and displacement of corner m (from global memory or from an LDS stage).
#ifndef HEX_UNROLL
#define HEX_UNROLL 2
#endif
// reference gradient of shape function m at (x, y, z)
__device__ __forceinline__ void hex8_dN(int m, double x, double y, double z, double* dN) {
  const double sx = HEX_SX[m], sy = HEX_SY[m], sz = HEX_SZ[m];
  dN[0] = 0.125 * sx * (1 + sy * y) * (1 + sz * z);
  dN[1] = 0.125 * sy * (1 + sx * x) * (1 + sz * z);
  dN[2] = 0.125 * sz * (1 + sx * x) * (1 + sy * y);
}

// Two passes over the corners (coordinates, then displacements) with the shape-function gradients
// recomputed in the second one: ~70 VGPRs live instead of ~110 when dN and the nodal values of all 8
// corners are kept (matters inside the fused update kernels, which run at 128).
template <class NodeFn>
__device__ __forceinline__ void hex8_disp_grad(const double x, const double y, const double z, NodeFn node,
                                               double* __restrict__ H /* H[i][j] = du_i / dX_j */) {
  double Jm[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // Jm[a][d] = dX_a / dxi_d
#pragma unroll HEX_UNROLL
  for (int m = 0; m < 8; ++m) {
    double dN[3], X[3], U[3];
    hex8_dN(m, x, y, z, dN);
    node(m, X, U);
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      Jm[0 + d] += X[0] * dN[d];
      Jm[3 + d] += X[1] * dN[d];
      Jm[6 + d] += X[2] * dN[d];
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
#pragma unroll
  for (int k = 0; k < 9; ++k) H[k] = 0.0;
#pragma unroll HEX_UNROLL
  for (int m = 0; m < 8; ++m) {
    double dN[3], X[3], U[3], g[3];  // g: physical gradient of shape function m
    hex8_dN(m, x, y, z, dN);
    node(m, X, U);
#pragma unroll
    for (int a = 0; a < 3; ++a) g[a] = dN[0] * Ji[0 + a] + dN[1] * Ji[3 + a] + dN[2] * Ji[6 + a];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      H[0 + a] += U[0] * g[a];
      H[3 + a] += U[1] * g[a];
      H[6 + a] += U[2] * g[a];
    }
  }
}

// kind 0: Mandel strain (6)  [utils.py:146-165];  kind 1: F = I + grad u (9) [utils.py:168-190]
template <int KIND, class NodeFn>
__device__ __forceinline__ void hex8_point(const double x, const double y, const double z, NodeFn node,
                                           double* __restrict__ o) {
  double H[9];
  hex8_disp_grad(x, y, z, node, H);
  if constexpr (KIND == 0) {
    const double r = 0.70710678118654752440;  // sqrt(2) * (1/2)
    double2_t* o2 = reinterpret_cast<double2_t*>(o);   // 48 B per point: 16 B aligned
    o2[0] = double2_t{H[0], H[4]};
    o2[1] = double2_t{H[8], r * (H[1] + H[3])};
    o2[2] = double2_t{r * (H[2] + H[6]), r * (H[5] + H[7])};
  } else {
    o[0] = 1.0 + H[0]; o[1] = 1.0 + H[4]; o[2] = 1.0 + H[8];
    o[3] = H[1]; o[4] = H[3]; o[5] = H[2]; o[6] = H[6]; o[7] = H[5]; o[8] = H[7];
  }
}

// Direct variant (any nqp): every thread gathers the 8 nodes of its cell from global memory.
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
  auto node = [&](int m, double* X, double* U) {
    const int64_t nd = conn[cell * 8 + m];
    X[0] = coords[3 * nd]; X[1] = coords[3 * nd + 1]; X[2] = coords[3 * nd + 2];
    U[0] = u[3 * nd]; U[1] = u[3 * nd + 1]; U[2] = u[3 * nd + 2];
  };
  hex8_point<KIND>(qp.xi[q][0], qp.xi[q][1], qp.xi[q][2], node, grad + gid * (KIND == 0 ? 6 : 9));
}

// Staged variant (nqp >= 4, i.e. at most 65 cells per 256-point block): the points of a cell share
// its 8 nodes, so the block first gathers every (cell, corner) once -- one corner per thread: 1
// connectivity entry + 6 doubles instead of 8 + 48 per thread -- into LDS, and the points then read
// their cell's record from there (the 8 lanes of a cell read the same addresses: LDS broadcast).
// Record stride 50 doubles = 100 dwords: the 8 cells of a wave start 36 dwords apart modulo 64 banks.
constexpr int HEX_STAGE_CELLS = 66;
constexpr int HEX_STAGE_REC = 50;
template <int KIND>
__global__ void __launch_bounds__(256)
hex8_gradient_staged_kernel(const double* __restrict__ coords, const int32_t* __restrict__ conn,
                            const double* __restrict__ u, const int64_t ncells, const QuadPoints qp,
                            double* __restrict__ grad) {
  __shared__ __attribute__((aligned(16))) double nod[HEX_STAGE_CELLS * HEX_STAGE_REC];
  const int64_t npts = ncells * qp.nqp;
  const int64_t p0 = (int64_t)blockIdx.x * 256;
  const int64_t plast = (p0 + 255 < npts ? p0 + 255 : npts - 1);
  const int64_t c0 = p0 / qp.nqp;
  const int ncb = (int)(plast / qp.nqp - c0) + 1;
  for (int e = threadIdx.x; e < ncb * 8; e += 256) {
    const int64_t nd = conn[c0 * 8 + e];   // the block's connectivity rows are contiguous
    double2_t* d = reinterpret_cast<double2_t*>(nod + (e >> 3) * HEX_STAGE_REC + (e & 7) * 6);
    const double X0 = coords[3 * nd], X1 = coords[3 * nd + 1], X2 = coords[3 * nd + 2];
    const double U0 = u[3 * nd], U1 = u[3 * nd + 1], U2 = u[3 * nd + 2];
    d[0] = double2_t{X0, X1};
    d[1] = double2_t{X2, U0};
    d[2] = double2_t{U1, U2};
  }
  __syncthreads();
  const int64_t gid = p0 + threadIdx.x;
  if (gid >= npts) return;
  const int64_t cell = gid / qp.nqp;
  const int q = (int)(gid - cell * qp.nqp);
  const double2_t* rec = reinterpret_cast<const double2_t*>(nod + (int)(cell - c0) * HEX_STAGE_REC);
  auto node = [&](int m, double* X, double* U) {
    const double2_t a = rec[m * 3], b = rec[m * 3 + 1], c = rec[m * 3 + 2];
    X[0] = a.x; X[1] = a.y; X[2] = b.x;
    U[0] = b.y; U[1] = c.x; U[2] = c.y;
  };
  hex8_point<KIND>(qp.xi[q][0], qp.xi[q][1], qp.xi[q][2], node, grad + gid * (KIND == 0 ? 6 : 9));
}

// First-order tetrahedra (affine): the displacement gradient is constant per cell,
//   H = sum_m u_m (x) grad N_m,  grad N from the inverse of the edge matrix [X1-X0, X2-X0, X3-X0].
// One thread per Gauss point (the nqp points of a cell repeat the cell value, as a dolfinx
// quadrature Function of degree > 1 would hold it).
// Displacement gradient of a linear tetrahedron from the coordinates X and displacements U of its 4 nodes.
__device__ __forceinline__ void tet4_disp_grad(const double (*X)[3], const double (*U)[3], double* __restrict__ H) {
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
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int a = 0; a < 3; ++a)
      H[i * 3 + a] = (U[1][i] - U[0][i]) * Ai[0 + a] + (U[2][i] - U[0][i]) * Ai[3 + a] + (U[3][i] - U[0][i]) * Ai[6 + a];
}

// gathers the 4 nodes of `cell` and evaluates its (constant) displacement gradient
__device__ __forceinline__ void tet4_cell_disp_grad(const double* __restrict__ coords, const int32_t* __restrict__ conn,
                                                    const double* __restrict__ u, int64_t cell, double* __restrict__ H) {
  double X[4][3], U[4][3];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int64_t nd = conn[cell * 4 + m];
#pragma unroll
    for (int a = 0; a < 3; ++a) { X[m][a] = coords[3 * nd + a]; U[m][a] = u[3 * nd + a]; }
  }
  tet4_disp_grad(X, U, H);
}

// Lagrange element of any order on a straight-sided simplex: the geometry map is affine (vertices only), so
//   H[i][a] = sum_d (sum_m u_m[i] dphi[q][m][d]) Ai[d][a],   Ai = inverse of the edge matrix,
// with dphi the tabulated reference derivatives (what basix hands out as element.tabulate(1, points)[1:]).
// tdim = 2: u has two components per dof, H is embedded in 3x3 with zeros (plane strain: eps_zz = 0, F_zz = 1).
// Every lane gathers its own cell; the nqp lanes of a cell ask for the same addresses (merged by the
// texture addresser), the table is a few hundred bytes and stays in the vector L1.
__device__ __forceinline__ void simplex_disp_grad(const MeshSource& s, const int64_t cell, const int q,
                                                  double* __restrict__ H) {
  double Ai[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};   // Ai[d][a] = dxi_d / dX_a
  if (s.tdim == 3) {
    double X[4][3];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int64_t v = s.conn[cell * 4 + m];
#pragma unroll
      for (int a = 0; a < 3; ++a) X[m][a] = s.coords[3 * v + a];
    }
    double A[9];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int d = 0; d < 3; ++d) A[a * 3 + d] = X[d + 1][a] - X[0][a];
    const double c00 = A[4] * A[8] - A[5] * A[7], c01 = A[5] * A[6] - A[3] * A[8], c02 = A[3] * A[7] - A[4] * A[6];
    const double idet = 1.0 / (A[0] * c00 + A[1] * c01 + A[2] * c02);
    Ai[0] = c00 * idet; Ai[3] = c01 * idet; Ai[6] = c02 * idet;
    Ai[1] = (A[2] * A[7] - A[1] * A[8]) * idet;
    Ai[4] = (A[0] * A[8] - A[2] * A[6]) * idet;
    Ai[7] = (A[1] * A[6] - A[0] * A[7]) * idet;
    Ai[2] = (A[1] * A[5] - A[2] * A[4]) * idet;
    Ai[5] = (A[2] * A[3] - A[0] * A[5]) * idet;
    Ai[8] = (A[0] * A[4] - A[1] * A[3]) * idet;
  } else {
    double X[3][2];
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      const int64_t v = s.conn[cell * 3 + m];
      X[m][0] = s.coords[3 * v]; X[m][1] = s.coords[3 * v + 1];
    }
    const double a00 = X[1][0] - X[0][0], a01 = X[2][0] - X[0][0], a10 = X[1][1] - X[0][1], a11 = X[2][1] - X[0][1];
    const double idet = 1.0 / (a00 * a11 - a01 * a10);
    Ai[0] = a11 * idet; Ai[1] = -a01 * idet;
    Ai[3] = -a10 * idet; Ai[4] = a00 * idet;
  }
  // B[i][d] = sum_m u_m[i] dphi[q][m][d] (du_i / dxi_d: 9 fused multiply-adds per dof), then H = B Ai once
  double B[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  const int32_t* dofs = s.dofmap + cell * s.nd;
  const double* tab = s.dphi + (int64_t)q * s.nd * s.tdim;
  if (s.tdim == 3) {
#pragma unroll 2
    for (int m = 0; m < s.nd; ++m) {
      const int64_t dof = dofs[m];
      const double t0 = tab[3 * m], t1 = tab[3 * m + 1], t2 = tab[3 * m + 2];
      const double U0 = s.u[3 * dof], U1 = s.u[3 * dof + 1], U2 = s.u[3 * dof + 2];
      B[0] += U0 * t0; B[1] += U0 * t1; B[2] += U0 * t2;
      B[3] += U1 * t0; B[4] += U1 * t1; B[5] += U1 * t2;
      B[6] += U2 * t0; B[7] += U2 * t1; B[8] += U2 * t2;
    }
  } else {
#pragma unroll 2
    for (int m = 0; m < s.nd; ++m) {
      const int64_t dof = dofs[m];
      const double t0 = tab[2 * m], t1 = tab[2 * m + 1];
      const double U0 = s.u[2 * dof], U1 = s.u[2 * dof + 1];
      B[0] += U0 * t0; B[1] += U0 * t1;
      B[3] += U1 * t0; B[4] += U1 * t1;
    }
  }
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int a = 0; a < 3; ++a) H[i * 3 + a] = B[i * 3] * Ai[a] + B[i * 3 + 1] * Ai[3 + a] + B[i * 3 + 2] * Ai[6 + a];
}

template <int KIND>
__global__ void __launch_bounds__(256)
simplex_gradient_kernel(const MeshSource src, double* __restrict__ grad) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= src.ncells * src.nqp) return;
  const int64_t cell = gid / src.nqp;
  double H[9];
  simplex_disp_grad(src, cell, (int)(gid - cell * src.nqp), H);
  if constexpr (KIND == 0) {
    const double r = 0.70710678118654752440;
    double2_t* o2 = reinterpret_cast<double2_t*>(grad + gid * 6);
    o2[0] = double2_t{H[0], H[4]};
    o2[1] = double2_t{H[8], r * (H[1] + H[3])};
    o2[2] = double2_t{r * (H[2] + H[6]), r * (H[5] + H[7])};
  } else {
    double* o = grad + gid * 9;
    o[0] = 1.0 + H[0]; o[1] = 1.0 + H[4]; o[2] = 1.0 + H[8];
    o[3] = H[1]; o[4] = H[3]; o[5] = H[2]; o[6] = H[6]; o[7] = H[5]; o[8] = H[7];
  }
}

template <int KIND>
__global__ void __launch_bounds__(256)
tet4_gradient_kernel(const double* __restrict__ coords, const int32_t* __restrict__ conn,
                     const double* __restrict__ u, const int64_t ncells, const int nqp,
                     double* __restrict__ grad) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= ncells * nqp) return;
  double H[9];
  tet4_cell_disp_grad(coords, conn, u, gid / nqp, H);
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
