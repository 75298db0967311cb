// Shared device-side definitions for the gfx950 constitutive-update kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dxm {

constexpr int WAVE = 64;            // gfx950 wavefront
constexpr int BLOCK = 256;          // 4 waves per workgroup
constexpr int WAVES_PER_BLOCK = BLOCK / WAVE;

// Material parameters as the kernels consume them (built on the host from E, nu, ...).
struct LawParams {
  double lambda;   // first Lame coefficient          python_materials/elasticity.py:12-13
  double mu;       // shear modulus
  double kappa;    // bulk modulus (finite strain)
  double sig0;     // initial yield stress
  double h1;       // linear: H           | Voce: sigu
  double h2;       // linear: unused      | Voce: b
  double tol;      // absolute residual tolerance of the local Newton (= rtol * max(|sig0|, 2e-8 mu))
  double rtol;     // the relative one it was built from
  int32_t maxit;   // local Newton iteration cap
  int32_t pad;
  double c[6];     // parameters of a user-supplied hardening law (JIT builds only, see below)
};

// A library built with -DDXM_CUSTOM_HARDENING replaces the Voce law of the "voce" kernels by two
// user-supplied C expressions DXM_CUSTOM_R and DXM_CUSTOM_DR in the variables `p` (cumulated
// plastic strain), `sig0` and `c[0..5]`: R(p) and dR/dp of an arbitrary isotropic hardening law
// (what a Python `yield_stress(p)` callable is to jaxmat: tests/test_FeFp_jax.py:14-15).  The
// Python layer compiles such a library on demand (dolfinx_materials_amd/_lib.py::load_custom).
// A value the optimiser cannot look through: a product passed through this is never contracted into an
// FMA with a neighbouring add.  The Voce law and every traced Python law (tracing.py emits DXM_MUL for
// each product) are evaluated with these plain, individually rounded operations, so that (i) a traced
// callable and the built-in law with the same formula give bit-identical results whatever the
// surrounding code looks like, and (ii) the device evaluates R(p) with the roundings of the Python
// callable itself (up to the exp / pow implementations).
__device__ __forceinline__ double opaque(double x) {
  asm("" : "+v"(x));
  return x;
}
#define DXM_MUL(a, b) (::dxm::opaque((a) * (b)))

#ifdef DXM_CUSTOM_HARDENING
__device__ __forceinline__ double custom_R(const LawParams& prm, double p) {
  const double sig0 = prm.sig0;
  const double* c = prm.c;
  (void)sig0; (void)c;
  return (DXM_CUSTOM_R);
}
__device__ __forceinline__ double custom_dR(const LawParams& prm, double p) {
  const double sig0 = prm.sig0;
  const double* c = prm.c;
  (void)sig0; (void)c;
  return (DXM_CUSTOM_DR);
}
#endif

// One record per workgroup, summed on the host on demand (no atomics on the hot path: a
// same-address atomic fan-in of a few thousand workgroups costs ~10 ns each at kernel end).
struct BlockStats {
  unsigned long long n_plastic;
  unsigned long long n_not_converged;
  unsigned long long n_nan;
  unsigned long long max_iters;
};

// LDS traffic between lanes of ONE wave: DS operations of a wave execute in issue order, so no
// s_barrier is needed; the fences only stop the compiler from moving LDS accesses across.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The same ordering point without memory-model fences, for code that has LDS-DMA requests
// (`global_load_lds`) in flight: a fence, even at wavefront scope, makes the compiler drain the vector
// memory counter (`s_waitcnt vmcnt(0)`: the DMA is a pending LDS write) and with it every global store
// issued before.  A compiler-level memory barrier keeps the LDS accesses on their side of the point; the
// hardware executes one wave's DS operations in order.
__device__ __forceinline__ void wave_lds_order() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  asm volatile("" ::: "memory");
}

__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, WAVE);
  return v;
}
__device__ __forceinline__ unsigned long long wave_max(unsigned long long v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    unsigned long long o = __shfl_down(v, off, WAVE);
    v = o > v ? o : v;
  }
  return v;
}

// Reduce per-thread counters over the workgroup and let thread 0 store the block record.  `red` is LDS:
// wave w uses the 4 words at red[w * stride ...] (its own wave-private region once the tile loop is over:
// a kernel with LDS-DMA in flight must keep ALL its LDS in one __shared__ object -- with a second one the
// compiler drains `vmcnt(0)` before the first LDS write that follows a DMA request).
__device__ __forceinline__ void store_block_stats(BlockStats* out, unsigned long long n_plastic,
                                                  unsigned long long n_notconv,
                                                  unsigned long long n_nan,
                                                  unsigned long long max_it,
                                                  unsigned long long* red, int stride = 4) {
  const int lane = threadIdx.x & (WAVE - 1);
  const int wid = threadIdx.x >> 6;
  n_plastic = wave_sum(n_plastic);
  n_notconv = wave_sum(n_notconv);
  n_nan = wave_sum(n_nan);
  max_it = wave_max(max_it);
  if (lane == 0) {
    red[wid * stride + 0] = n_plastic;
    red[wid * stride + 1] = n_notconv;
    red[wid * stride + 2] = n_nan;
    red[wid * stride + 3] = max_it;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    BlockStats s = {0, 0, 0, 0};
    for (int w = 0; w < WAVES_PER_BLOCK; ++w) {
      s.n_plastic += red[w * stride + 0];
      s.n_not_converged += red[w * stride + 1];
      s.n_nan += red[w * stride + 2];
      s.max_iters = red[w * stride + 3] > s.max_iters ? red[w * stride + 3] : s.max_iters;
    }
    out[blockIdx.x] = s;
  }
}

// 1/x from the hardware approximation plus two Newton-Raphson steps (<= 1 ulp for normal x):
// 5 instructions instead of the ~14 of the IEEE division sequence; used where the operand is a
// well-scaled positive quantity (J, |dev be|, Jacobian determinants, stress ratios).
__device__ __forceinline__ double fast_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
  return r;
}

typedef double double2_t __attribute__((ext_vector_type(2)));

// Cache policy of the streaming accesses, fixed at build time by the bits of DXM_NT:
//   bit 0 = non-temporal stores of flux and tangent      bit 1 = of the new state
//   bit 2 = non-temporal loads of the gradient            bit 3 = of the old state
// Shipped: 1.  Flux and tangent (336 of the 496 B/point of J2) are not read again on the GPU, so
// their lines need not displace the gradient and the state in L2 / Infinity Cache: same-process A/B
// (tools/ab_inproc.py, profiles/archive/r01_nt_ab.jsonl) -26 % kernel time at 1e6 points, -9 % at 1e5,
// -2 % at 1e7, never slower.  The new state IS read back by the next update (advance), and a
// non-temporal state store costs +18 % at 3e5 points; non-temporal loads only lose where the
// gradient was just written by the displacement-gradient kernel.  So bits 1-3 stay off.
#ifndef DXM_NT
#define DXM_NT 1
#endif
template <int BIT, class T>
__device__ __forceinline__ void stream_store(T* p, T v) {
  if constexpr ((DXM_NT >> BIT) & 1) __builtin_nontemporal_store(v, p);
  else *p = v;
}
template <int BIT, class T>
__device__ __forceinline__ T stream_load(const T* p) {
  if constexpr ((DXM_NT >> BIT) & 1) return __builtin_nontemporal_load(p);
  else return *p;
}

}  // namespace dxm
