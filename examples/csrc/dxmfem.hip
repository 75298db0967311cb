// libdxmfem.so -- matrix-free assembly-side operators on a hex8 mesh (8 Gauss points per cell, small strain), for the
// device-resident stand-in FE loop of examples/device_fem.py.
//
// NOT part of the product: SURVEY.md section 2 row 7 / section 8 keep FEM assembly on the host (dolfinx), and libdxmat.so exports
// the constitutive update and nothing else.  These kernels are what a caller that keeps everything on the GPU does with the
// stress / tangent arrays the update produced -- the residual form `dot(sig, strain(v)) * dx` and its derivative
// (tests/uniaxial_tension.py:62-67, quadrature_map.py:132-158 of the reference) restated matrix-free.  They lived in
// libdxmat.so until round 4 (dxm_mesh_internal_force_device & co.); the library is stateless now: every array is the
// caller's (torch tensors in examples/fem_operators.py), so nothing of libdxmat's internals is shared.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

namespace dxf {

using double2_t = double2;

// reference corner signs of the trilinear hexahedron, node order
// (-,-,-) (+,-,-) (+,+,-) (-,+,-) (-,-,+) (+,-,+) (+,+,+) (-,+,+)   (the order of dxm_mesh_create_hex8)
__device__ __constant__ const signed char HEX_SX[8] = {-1, 1, 1, -1, -1, 1, 1, -1};
__device__ __constant__ const signed char HEX_SY[8] = {-1, -1, 1, 1, -1, -1, 1, 1};
__device__ __constant__ const signed char HEX_SZ[8] = {-1, -1, -1, -1, 1, 1, 1, 1};

// reference gradient of shape function m at (x, y, z)
__device__ __forceinline__ void hex8_dN(int m, double x, double y, double z, double* dN) {
  const double sx = HEX_SX[m], sy = HEX_SY[m], sz = HEX_SZ[m];
  dN[0] = 0.125 * sx * (1 + sy * y) * (1 + sz * z);
  dN[1] = 0.125 * sy * (1 + sx * x) * (1 + sz * z);
  dN[2] = 0.125 * sz * (1 + sx * x) * (1 + sy * y);
}

constexpr int HEX_STAGE_REC = 50;   // doubles per staged cell record (8 corners x 6, padded to an even, bank-friendly stride)

// ---- assembly-side consumers on the device (hex8 with 8 Gauss points per cell, small strain) -----------------
// What dolfinx assembly does with the quadrature Functions after QuadratureMap.update (the residual form
// `dot(sig, strain(v)) * dx` and its derivative, tests/uniaxial_tension.py:62-67, quadrature_map.py:132-158),
// restated matrix-free for a caller that keeps everything on the GPU:
//   OP_FORCE     f  = sum_q w detJ  B_q^T sigma_q                        (internal force from the stress array)
//   OP_APPLY     y  = sum_q w detJ  B_q^T Ct_q B_q x                     (tangent operator times a vector)
//   OP_DIAGONAL  d  = diag( sum_q w detJ  B_q^T Ct_q B_q )               (Jacobi preconditioner)
// with Ct either the nine coefficients of Ct = c1 1x1 + c2 I + c3 n x n (TL_COEF, 72 B/point) or the full block.
// Two deterministic passes, no atomics: hex8_element_kernel writes the 24 element values of every cell
// (lane (cell, q) evaluates its Gauss point and leaves inverse Jacobian + weighted stress in LDS, lane (cell, corner)
// then sums B^T over the cell's 8 points), node_gather_kernel adds up the <= 8 element contributions of every node
// through a node -> (cell, corner) table.
enum { OP_FORCE = 0, OP_APPLY = 1, OP_DIAGONAL = 2 };
struct HexOperatorArgs {
  const double* coords;
  const int32_t* conn;
  const double* x;       // OP_APPLY: nodal vector (n_nodes * 3)
  const double* field;   // OP_FORCE: stress (npoints, 6) Mandel; otherwise the tangent in `layout`
  int64_t ncells;
  int32_t layout;        // 0 full (npoints, 36), 2 coefficients (npoints, 9)
  double xi[8][3];
  double w[8];
};
constexpr int HEX_OP_CELLS = 32;    // cells per 256-thread block
constexpr int HEX_OP_PT_DIAG = 21, HEX_OP_PT = 9;   // doubles per Gauss-point record (see phase A), odd strides

template <int OP>
__global__ void __launch_bounds__(256)
hex8_element_kernel(const HexOperatorArgs a, double* __restrict__ fe /* (8, ncells, 3): corner-major */) {
  constexpr int PT = OP == OP_DIAGONAL ? HEX_OP_PT_DIAG : HEX_OP_PT;
  static_assert(256 * PT >= HEX_OP_CELLS * HEX_STAGE_REC, "the point records reuse the node staging region");
  __shared__ __attribute__((aligned(16))) double lds[256 * PT];
  __shared__ double dnt[8 * 8 * 3];   // reference shape-function gradients dN_m / dxi_d at the 8 points: [q][m][d]
  const int lane8 = threadIdx.x & 7, cl = threadIdx.x >> 3;
  const int64_t cell = (int64_t)blockIdx.x * HEX_OP_CELLS + cl;
  const bool live = cell < a.ncells;
  if (threadIdx.x < 64) {
    double dN[3];
    hex8_dN(lane8, a.xi[cl][0], a.xi[cl][1], a.xi[cl][2], dN);   // here cl = q, lane8 = m
    dnt[threadIdx.x * 3] = dN[0]; dnt[threadIdx.x * 3 + 1] = dN[1]; dnt[threadIdx.x * 3 + 2] = dN[2];
  }
  // ---- stage the nodes of the block's cells: one (cell, corner) per thread
  {
    double2_t r0 = {0.0, 0.0}, r1 = {0.0, 0.0}, r2 = {0.0, 0.0};
    if (live) {
      const int64_t nd = a.conn[cell * 8 + lane8];
      r0 = double2_t{a.coords[3 * nd], a.coords[3 * nd + 1]};
      r1.x = a.coords[3 * nd + 2];
      if constexpr (OP == OP_APPLY) { r1.y = a.x[3 * nd]; r2 = double2_t{a.x[3 * nd + 1], a.x[3 * nd + 2]}; }
    }
    double2_t* d = reinterpret_cast<double2_t*>(lds + cl * HEX_STAGE_REC + lane8 * 6);
    d[0] = r0; d[1] = r1; d[2] = r2;
  }
  __syncthreads();
  // ---- phase A: lane (cell, q).  Record left for phase B:
  //   OP_FORCE / OP_APPLY   T[i][d] = w detJ sum_a S[i][a] Ji[d][a]  (9): the stress-like tensor pulled back to the
  //                         reference cell, so that phase B needs the REFERENCE gradients only: f_m[i] = T[i][:] . dN_m
  //   OP_DIAGONAL           Ji (9), coefficients (9), w detJ
  double rec[PT];
  {
    const double2_t* nrec = reinterpret_cast<const double2_t*>(lds + cl * HEX_STAGE_REC);
    const double* dq = dnt + lane8 * 24;
    double Jm[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll 2
    for (int m = 0; m < 8; ++m) {
      const double d0 = dq[3 * m], d1 = dq[3 * m + 1], d2 = dq[3 * m + 2];
      const double2_t p0 = nrec[m * 3], p1 = nrec[m * 3 + 1];
      Jm[0] += p0.x * d0; Jm[1] += p0.x * d1; Jm[2] += p0.x * d2;
      Jm[3] += p0.y * d0; Jm[4] += p0.y * d1; Jm[5] += p0.y * d2;
      Jm[6] += p1.x * d0; Jm[7] += p1.x * d1; Jm[8] += p1.x * d2;
    }
    const double c00 = Jm[4] * Jm[8] - Jm[5] * Jm[7], c01 = Jm[5] * Jm[6] - Jm[3] * Jm[8], c02 = Jm[3] * Jm[7] - Jm[4] * Jm[6];
    const double det = Jm[0] * c00 + Jm[1] * c01 + Jm[2] * c02;
    const double idet = live ? 1.0 / det : 0.0;
    double Ji[9];
    Ji[0] = c00 * idet; Ji[3] = c01 * idet; Ji[6] = c02 * idet;
    Ji[1] = (Jm[2] * Jm[7] - Jm[1] * Jm[8]) * idet;
    Ji[4] = (Jm[0] * Jm[8] - Jm[2] * Jm[6]) * idet;
    Ji[7] = (Jm[1] * Jm[6] - Jm[0] * Jm[7]) * idet;
    Ji[2] = (Jm[1] * Jm[5] - Jm[2] * Jm[4]) * idet;
    Ji[5] = (Jm[2] * Jm[3] - Jm[0] * Jm[5]) * idet;
    Ji[8] = (Jm[0] * Jm[4] - Jm[1] * Jm[3]) * idet;
    const double wdet = live ? a.w[lane8] * det : 0.0;
    const int64_t pt = cell * 8 + lane8;
    const double r = 0.70710678118654752440;
    if constexpr (OP == OP_DIAGONAL) {
#pragma unroll
      for (int k = 0; k < 9; ++k) rec[k] = Ji[k];
#pragma unroll
      for (int k = 0; k < 9; ++k) rec[9 + k] = live ? a.field[pt * 9 + k] : 0.0;
      rec[18] = wdet;
    } else {
      double s[6] = {0, 0, 0, 0, 0, 0};
      if constexpr (OP == OP_FORCE) {
        if (live) {
#pragma unroll
          for (int k = 0; k < 6; ++k) s[k] = a.field[pt * 6 + k];
        }
      } else {
        // du_i / dxi_d first (reference gradients), then H = B Ji
        double B[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll 2
        for (int m = 0; m < 8; ++m) {
          const double d0 = dq[3 * m], d1 = dq[3 * m + 1], d2 = dq[3 * m + 2];
          const double2_t p1 = nrec[m * 3 + 1], p2 = nrec[m * 3 + 2];
          B[0] += p1.y * d0; B[1] += p1.y * d1; B[2] += p1.y * d2;
          B[3] += p2.x * d0; B[4] += p2.x * d1; B[5] += p2.x * d2;
          B[6] += p2.y * d0; B[7] += p2.y * d1; B[8] += p2.y * d2;
        }
        double H[9];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int c = 0; c < 3; ++c) H[i * 3 + c] = B[i * 3] * Ji[c] + B[i * 3 + 1] * Ji[3 + c] + B[i * 3 + 2] * Ji[6 + c];
        const double e[6] = {H[0], H[4], H[8], r * (H[1] + H[3]), r * (H[2] + H[6]), r * (H[5] + H[7])};
        if (live) {
          if (a.layout == 2) {
            const double* cf = a.field + pt * 9;
            const double k1 = cf[0], k2 = cf[1], k3 = cf[2];
            const double tr = e[0] + e[1] + e[2];
            double nd = 0.0;
#pragma unroll
            for (int k = 0; k < 6; ++k) nd += cf[3 + k] * e[k];
#pragma unroll
            for (int k = 0; k < 6; ++k) s[k] = k2 * e[k] + (k3 * nd) * cf[3 + k] + (k < 3 ? k1 * tr : 0.0);
          } else {
            const double* ct = a.field + pt * 36;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
              double acc = 0.0;
#pragma unroll
              for (int j = 0; j < 6; ++j) acc += ct[i * 6 + j] * e[j];
              s[i] = acc;
            }
          }
        }
      }
      // weighted symmetric tensor S, then T[i][d] = sum_a S[i][a] Ji[d][a]
      const double S[9] = {wdet * s[0], wdet * r * s[3], wdet * r * s[4],
                           wdet * r * s[3], wdet * s[1], wdet * r * s[5],
                           wdet * r * s[4], wdet * r * s[5], wdet * s[2]};
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int d = 0; d < 3; ++d) rec[i * 3 + d] = S[i * 3] * Ji[d * 3] + S[i * 3 + 1] * Ji[d * 3 + 1] + S[i * 3 + 2] * Ji[d * 3 + 2];
    }
  }
  __syncthreads();   // everybody is done with the node records: the region becomes the point records
  {
    double* dst = lds + threadIdx.x * PT;
    constexpr int NW = OP == OP_DIAGONAL ? 19 : 9;
#pragma unroll
    for (int k = 0; k < NW; ++k) dst[k] = rec[k];
  }
  __syncthreads();
  // ---- phase B: lane (cell, corner m) sums over the cell's 8 points
  double acc[3] = {0.0, 0.0, 0.0};
#pragma unroll 2
  for (int q = 0; q < 8; ++q) {
    const double* pr = lds + (cl * 8 + q) * PT;   // the 8 lanes of a cell read the same record: broadcast
    const double* dm = dnt + (q * 8 + lane8) * 3;
    const double d0 = dm[0], d1 = dm[1], d2 = dm[2];
    if constexpr (OP == OP_DIAGONAL) {
      double g[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) g[c] = d0 * pr[c] + d1 * pr[3 + c] + d2 * pr[6 + c];
      const double k1 = pr[9], k2 = pr[10], k3 = pr[11], wd = pr[18];
      const double r = 0.70710678118654752440;
      const double gg = g[0] * g[0] + g[1] * g[1] + g[2] * g[2];
      // unit displacement of this node in direction i: eps = sym(e_i (x) g); |eps|^2 = (g_i^2 + |g|^2) / 2, tr = g_i
      const double n0 = pr[12], n1 = pr[13], n2 = pr[14], n3 = pr[15], n4 = pr[16], n5 = pr[17];
      const double ne[3] = {n0 * g[0] + r * (n3 * g[1] + n4 * g[2]), n1 * g[1] + r * (n3 * g[0] + n5 * g[2]),
                            n2 * g[2] + r * (n4 * g[0] + n5 * g[1])};
#pragma unroll
      for (int i = 0; i < 3; ++i)
        acc[i] += wd * (k1 * g[i] * g[i] + k2 * 0.5 * (g[i] * g[i] + gg) + k3 * ne[i] * ne[i]);
    } else {
      acc[0] += pr[0] * d0 + pr[1] * d1 + pr[2] * d2;
      acc[1] += pr[3] * d0 + pr[4] * d1 + pr[5] * d2;
      acc[2] += pr[6] * d0 + pr[7] * d1 + pr[8] * d2;
    }
  }
  // corner-major: neighbouring nodes of a structured mesh find the values of the same corner slot side by side
  if (live) {
    double* o = fe + ((int64_t)lane8 * a.ncells + cell) * 3;
    o[0] = acc[0]; o[1] = acc[1]; o[2] = acc[2];
  }
}

// y[node] = sum of the element values of the (cell, corner) pairs that are this node; adj[k] = corner * ncells + cell,
// per node in ascending order (a fixed summation order: the result is reproducible bit for bit)
__global__ void __launch_bounds__(256)
node_gather_kernel(const int64_t nnodes, const int64_t* __restrict__ ptr, const int32_t* __restrict__ adj,
                   const double* __restrict__ fe, double* __restrict__ y) {
  const int64_t nd = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (nd >= nnodes) return;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0;
  for (int64_t k = ptr[nd]; k < ptr[nd + 1]; ++k) {
    const double* f = fe + 3 * (int64_t)adj[k];
    a0 += f[0]; a1 += f[1]; a2 += f[2];
  }
  y[3 * nd] = a0; y[3 * nd + 1] = a1; y[3 * nd + 2] = a2;
}

}  // namespace dxf

static thread_local char g_err[256] = "";
static int fail(int rc, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return rc;
}

extern "C" {

const char* dxf_last_error(void) { return g_err; }

// op 0: f = sum_q w detJ B_q^T sigma_q            field = stress (npoints, 6) Mandel, x unused
// op 1: y = sum_q w detJ B_q^T Ct_q B_q x         field = tangent, layout 0 full (npoints, 36) / 2 coefficients (npoints, 9)
// op 2: d = diag(sum_q w detJ B_q^T Ct_q B_q)     field = coefficients (npoints, 9), x unused
// Every pointer but xi (8 x 3 reference points in [-1, 1]^3) and w (8 weights) is device memory of `device`:
// coords (n_nodes, 3), conn (n_cells, 8), fe scratch (8 * n_cells * 3), node_ptr (n_nodes + 1) / node_adj (8 * n_cells): for
// every node the entries corner * n_cells + cell of the (cell, corner) pairs that are this node, ascending.
// Deterministic (element values, then a node gather; no atomics), asynchronous on hip_stream.
int dxf_hex8_operator(int op, int device, const double* coords_dev, const int32_t* conn_dev, int64_t n_cells, int64_t n_nodes,
                      const double* xi, const double* w, const double* field_dev, int layout, const double* x_dev,
                      double* fe_dev, const int64_t* node_ptr_dev, const int32_t* node_adj_dev, double* y_dev, void* hip_stream) {
  using namespace dxf;
  if (!coords_dev || !conn_dev || !xi || !w || !field_dev || !fe_dev || !node_ptr_dev || !node_adj_dev || !y_dev)
    return fail(-1, "null argument");
  if (op < OP_FORCE || op > OP_DIAGONAL) return fail(-1, "op must be 0 (force), 1 (apply) or 2 (diagonal)");
  if (op == OP_APPLY && (!x_dev || x_dev == y_dev)) return fail(-1, "tangent apply needs x, and x and y must not alias");
  if (op == OP_APPLY && layout != 0 && layout != 2) return fail(-1, "tangent layout must be 0 (full) or 2 (coefficients)");
  if (n_cells <= 0 || n_nodes <= 0 || n_cells * 8 > INT32_MAX) return fail(-1, "invalid mesh sizes");
  int prev = 0;
  if (hipGetDevice(&prev) != hipSuccess || hipSetDevice(device) != hipSuccess) return fail(-2, "no usable HIP device %d", device);
  HexOperatorArgs a{};
  a.coords = coords_dev; a.conn = conn_dev; a.x = x_dev; a.field = field_dev; a.ncells = n_cells;
  a.layout = op == OP_DIAGONAL ? 2 : layout;
  for (int q = 0; q < 8; ++q) {
    for (int d = 0; d < 3; ++d) a.xi[q][d] = xi[3 * q + d];
    a.w[q] = w[q];
  }
  hipStream_t st = (hipStream_t)hip_stream;
  const int blocks = (int)((n_cells + HEX_OP_CELLS - 1) / HEX_OP_CELLS);
  if (op == OP_FORCE) hipLaunchKernelGGL(hex8_element_kernel<OP_FORCE>, dim3(blocks), dim3(256), 0, st, a, fe_dev);
  else if (op == OP_APPLY) hipLaunchKernelGGL(hex8_element_kernel<OP_APPLY>, dim3(blocks), dim3(256), 0, st, a, fe_dev);
  else hipLaunchKernelGGL(hex8_element_kernel<OP_DIAGONAL>, dim3(blocks), dim3(256), 0, st, a, fe_dev);
  hipLaunchKernelGGL(node_gather_kernel, dim3((int)((n_nodes + 255) / 256)), dim3(256), 0, st, n_nodes, node_ptr_dev, node_adj_dev,
                     fe_dev, y_dev);
  const hipError_t e = hipGetLastError();
  (void)hipSetDevice(prev);
  if (e != hipSuccess) return fail(-3, "HIP error: %s", hipGetErrorString(e));
  return 0;
}

}  // extern "C"
