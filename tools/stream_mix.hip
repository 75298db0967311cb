// Speed-of-light probe for the J2 kernel's traffic mix: reads 104 B and writes 392 B per "point"
// (the algorithmic bytes of the constitutive update) as two perfectly linear
// 16 B-per-lane streams with no arithmetic.  Whatever this reaches on a given box is the ceiling a
// 1 : 3.8 read : write streaming kernel can reach there; tools/stream_mix.py compares the J2
// kernel against it in the same process.  Measurement infrastructure, not part of libdxmat.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef double double2_t __attribute__((ext_vector_type(2)));

// rpairs / wpairs: 16-byte pairs read / written per 64-point tile (J2: 416 / 1568; elastic: 192 / 1344;
// FeFp: 512 / 3296)
template <bool NT>
__global__ void __launch_bounds__(256) stream_mix_kernel(const double2_t* __restrict__ rbuf,
                                                         double2_t* __restrict__ wbuf, int64_t ntiles,
                                                         int rpairs, int wpairs) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t t = wave; t < ntiles; t += nwaves) {
    const double2_t* r = rbuf + t * rpairs;
    double2_t acc = {0.0, 0.0};
#pragma unroll 4
    for (int idx = lane; idx < rpairs; idx += 64) acc += r[idx];
    double2_t* w = wbuf + t * wpairs;
#pragma unroll 8
    for (int idx = lane; idx < wpairs; idx += 64) {
      if constexpr (NT) __builtin_nontemporal_store(acc, w + idx);   // the cache policy libdxmat ships for flux / tangent
      else w[idx] = acc;
    }
  }
}

// Software-pipelined variant: the next tile's reads are issued BEFORE the current tile's writes, so
// that waiting for them (vmcnt is in order) does not also wait for the writes to drain.
__global__ void __launch_bounds__(256) stream_mix_pipelined_kernel(const double2_t* __restrict__ rbuf,
                                                                   double2_t* __restrict__ wbuf, int64_t ntiles,
                                                                   int rpairs, int wpairs) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  double2_t nxt = {0.0, 0.0};
  if (wave < ntiles) {
    const double2_t* r = rbuf + wave * rpairs;
    for (int idx = lane; idx < rpairs; idx += 64) nxt += r[idx];
  }
  for (int64_t t = wave; t < ntiles; t += nwaves) {
    const double2_t acc = nxt;
    nxt = double2_t{0.0, 0.0};
    if (t + nwaves < ntiles) {
      const double2_t* r = rbuf + (t + nwaves) * rpairs;
#pragma unroll 4
      for (int idx = lane; idx < rpairs; idx += 64) nxt += r[idx];
    }
    double2_t* w = wbuf + t * wpairs;
#pragma unroll 8
    for (int idx = lane; idx < wpairs; idx += 64) w[idx] = acc;
  }
}

// J2-shaped variant: the same 104 B in / 392 B out per point, but split like the real kernel:
// strain 48 B (AoS, 16 B/lane) + 7 SoA state slots of 8 B/lane in; stress 48 B + 7 SoA slots +
// tangent 288 B out.  Measures what the 17 concurrent streams cost against the 2-stream probe.
__global__ void __launch_bounds__(256) stream_mix_j2_shape_kernel(const double* __restrict__ eps, const double* __restrict__ s0,
                                                                  double* __restrict__ s1, int64_t ld, double* __restrict__ sig,
                                                                  double* __restrict__ ct, int64_t ntiles) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t t = wave; t < ntiles; t += nwaves) {
    const int64_t base = t * 64;
    const double2_t* e2 = reinterpret_cast<const double2_t*>(eps + base * 6);
    double2_t acc = e2[lane] + e2[64 + lane] + e2[128 + lane];
    double a = 0.0;
#pragma unroll
    for (int c = 0; c < 7; ++c) a += s0[c * ld + base + lane];
    acc.x += a;
#pragma unroll
    for (int c = 0; c < 7; ++c) s1[c * ld + base + lane] = acc.x;
    double2_t* g2 = reinterpret_cast<double2_t*>(sig + base * 6);
#pragma unroll
    for (int k = 0; k < 3; ++k) g2[k * 64 + lane] = acc;
    double2_t* c2 = reinterpret_cast<double2_t*>(ct + base * 36);
#pragma unroll
    for (int k = 0; k < 18; ++k) c2[k * 64 + lane] = acc;
  }
}

extern "C" int stream_mix_j2_shape_launch(const void* eps, const void* s0, void* s1, int64_t ld, void* sig, void* ct,
                                          int64_t npoints, int blocks, void* stream) {
  hipLaunchKernelGGL(stream_mix_j2_shape_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const double*)eps,
                     (const double*)s0, (double*)s1, ld, (double*)sig, (double*)ct, npoints / 64);
  return (int)hipGetLastError();
}

// Elastic-shaped variant: the elastic kernel's three streams and nothing else -- strain 48 B in (AoS, 16 B per lane), stress 48 B
// and tangent 288 B out with the non-temporal stores libdxmat ships for flux / tangent (DXM_NT bit 0); no state.  384 B/point.
__global__ void __launch_bounds__(256) stream_mix_elastic_shape_kernel(const double* __restrict__ eps, double* __restrict__ sig,
                                                                       double* __restrict__ ct, int64_t ntiles) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t t = wave; t < ntiles; t += nwaves) {
    const int64_t base = t * 64;
    const double2_t* e2 = reinterpret_cast<const double2_t*>(eps + base * 6);
    const double2_t acc = e2[lane] + e2[64 + lane] + e2[128 + lane];
    double2_t* g2 = reinterpret_cast<double2_t*>(sig + base * 6);
#pragma unroll
    for (int k = 0; k < 3; ++k) __builtin_nontemporal_store(acc, g2 + k * 64 + lane);
    double2_t* c2 = reinterpret_cast<double2_t*>(ct + base * 36);
#pragma unroll
    for (int k = 0; k < 18; ++k) __builtin_nontemporal_store(acc, c2 + k * 64 + lane);
  }
}

extern "C" int stream_mix_elastic_shape_launch(const void* eps, void* sig, void* ct, int64_t npoints, int blocks, void* stream) {
  hipLaunchKernelGGL(stream_mix_elastic_shape_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const double*)eps, (double*)sig,
                     (double*)ct, npoints / 64);
  return (int)hipGetLastError();
}

extern "C" int stream_mix_pipelined_launch(const void* rbuf, void* wbuf, int64_t npoints, int read_bytes_per_point,
                                           int write_bytes_per_point, int blocks, void* stream) {
  hipLaunchKernelGGL(stream_mix_pipelined_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                     (const double2_t*)rbuf, (double2_t*)wbuf, npoints / 64, read_bytes_per_point * 4,
                     write_bytes_per_point * 4);
  return (int)hipGetLastError();
}

// Same probe with `lds_bytes` of dynamic LDS per workgroup, only to cap residency (e.g. 70 KiB ->
// 2 workgroups = 8 waves per CU, the FeFp kernel's occupancy).
extern "C" int stream_mix_capped_launch(const void* rbuf, void* wbuf, int64_t npoints, int read_bytes_per_point,
                                        int write_bytes_per_point, int blocks, int lds_bytes, void* stream) {
  hipLaunchKernelGGL(stream_mix_kernel<false>, dim3(blocks), dim3(256), lds_bytes, (hipStream_t)stream,
                     (const double2_t*)rbuf, (double2_t*)wbuf, npoints / 64, read_bytes_per_point * 4,
                     write_bytes_per_point * 4);
  return (int)hipGetLastError();
}

extern "C" int stream_mix_launch(const void* rbuf, void* wbuf, int64_t npoints, int read_bytes_per_point,
                                 int write_bytes_per_point, int blocks, void* stream) {
  hipLaunchKernelGGL(stream_mix_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                     (const double2_t*)rbuf, (double2_t*)wbuf, npoints / 64, read_bytes_per_point * 4,
                     write_bytes_per_point * 4);
  return (int)hipGetLastError();
}

extern "C" int stream_mix_nt_launch(const void* rbuf, void* wbuf, int64_t npoints, int read_bytes_per_point,
                                    int write_bytes_per_point, int blocks, void* stream) {
  hipLaunchKernelGGL(stream_mix_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                     (const double2_t*)rbuf, (double2_t*)wbuf, npoints / 64, read_bytes_per_point * 4,
                     write_bytes_per_point * 4);
  return (int)hipGetLastError();
}

// State-only variant: the 7 SoA slots in and 7 out of the J2 kernel, nothing else (placement studies).
__global__ void __launch_bounds__(256) stream_mix_state_only_kernel(const double* __restrict__ s0, double* __restrict__ s1,
                                                                    int64_t ld, int64_t ntiles) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t t = wave; t < ntiles; t += nwaves) {
    const int64_t base = t * 64;
    double a = 0.0;
#pragma unroll
    for (int c = 0; c < 7; ++c) a += s0[c * ld + base + lane];
#pragma unroll
    for (int c = 0; c < 7; ++c) s1[c * ld + base + lane] = a;
  }
}

extern "C" int stream_mix_state_only_launch(const void* s0, void* s1, int64_t ld, int64_t npoints, int blocks, void* stream) {
  hipLaunchKernelGGL(stream_mix_state_only_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const double*)s0,
                     (double*)s1, ld, npoints / 64);
  return (int)hipGetLastError();
}

// FeFp-shaped variant with the kernel's TIME structure: per 64-point tile F (AoS, 288 pairs) + 7 SoA state slots in; a
// "per-point phase" of `pre` dependent FMAs; 13 SoA slots + PK1 (288 pairs) out; then the 81-entry tangent in rounds of
// `ppr` points, each round preceded by `per_round` dependent FMAs (the tangent evaluation of the round) and stored as
// 1 KiB wave stores.  `prefetch` != 0 issues the next tile's loads before this tile's stores.  Residency is capped
// with dynamic LDS by the launcher (70 KiB -> the FeFp kernel's 2 workgroups per CU).  No arithmetic of the law:
// what this reaches is the ceiling of the kernel's memory shape + occupancy + compute gaps.
__device__ __forceinline__ double spin(double x, int n) {
#pragma unroll 1
  for (int i = 0; i < n; ++i) x = __builtin_fma(x, 1.0000001, 1e-9);
  return x;
}

// STATE: 0 = SoA slots, 8 B per lane (the shipped layout: 7 read + 13 written streams of 512 B wave accesses);
//        1 = tile-blocked [tile][slot][64], moved as linear 16 B-per-lane accesses (same bytes, 2 streams);
//        2 = no state traffic at all (what the other streams reach alone)
template <bool PREFETCH, int STATE, bool ALIGNED = false>
__global__ void __launch_bounds__(256) stream_mix_fefp_shape_kernel(const double* __restrict__ F, const double* __restrict__ s0,
                                                                    double* __restrict__ s1, int64_t ld, double* __restrict__ P,
                                                                    double* __restrict__ ct, int64_t ntiles, int pre, int per_round, int ppr) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  double2_t f[5];
  double st[7];
  double2_t st2[4];
  auto load = [&](int64_t t) {
    const int64_t base = t * 64;
    const double2_t* f2 = reinterpret_cast<const double2_t*>(F + base * 9);
#pragma unroll
    for (int k = 0; k < 5; ++k) f[k] = (k * 64 + lane < 288) ? f2[k * 64 + lane] : double2_t{0.0, 0.0};
    if constexpr (STATE == 0) {
#pragma unroll
      for (int c = 0; c < 7; ++c) st[c] = s0[c * ld + base + lane];
    } else if constexpr (STATE == 1) {
      const double2_t* b = reinterpret_cast<const double2_t*>(s0 + t * (7 * 64));   // 224 pairs per tile
#pragma unroll
      for (int k = 0; k < 4; ++k) st2[k] = (k * 64 + lane < 224) ? b[k * 64 + lane] : double2_t{0.0, 0.0};
    }
  };
  if (PREFETCH && wave < ntiles) load(wave);
  for (int64_t t = wave; t < ntiles; t += nwaves) {
    const int64_t base = t * 64;
    if (!PREFETCH) load(t);
    double2_t acc = f[0] + f[1] + f[2] + f[3] + f[4];
    if constexpr (STATE == 0) {
#pragma unroll
      for (int c = 0; c < 7; ++c) acc.x += st[c];
    } else if constexpr (STATE == 1) {
      acc += st2[0] + st2[1] + st2[2] + st2[3];
    }
    acc.x = spin(acc.x, pre);
    if (PREFETCH && t + nwaves < ntiles) load(t + nwaves);   // ahead of this tile's stores
    if constexpr (STATE == 0) {
#pragma unroll
      for (int c = 0; c < 13; ++c) s1[c * ld + base + lane] = acc.x;
    } else if constexpr (STATE == 1) {
      double2_t* b = reinterpret_cast<double2_t*>(s1 + t * (13 * 64));   // 416 pairs per tile
#pragma unroll
      for (int k = 0; k < 7; ++k)
        if (k * 64 + lane < 416) b[k * 64 + lane] = acc;
    }
    double2_t* p2 = reinterpret_cast<double2_t*>(P + base * 9);
#pragma unroll
    for (int k = 0; k < 5; ++k)
      if (k * 64 + lane < 288) __builtin_nontemporal_store(acc, p2 + k * 64 + lane);
    double2_t* c2 = reinterpret_cast<double2_t*>(ct + base * 81);
    const int total = 64 * 81 / 2;   // 2592 pairs per tile
    const int per = ppr * 81 / 2;    // pairs per round (ppr even)
    for (int o = 0; o < total; o += per) {
      acc.y = spin(acc.y, per_round);
      const int end = o + per < total ? o + per : total;
      if (ppr < 0 || !ALIGNED) {
        for (int idx = o + lane; idx < end; idx += 64) __builtin_nontemporal_store(acc, c2 + idx);
      } else {   // every wave store covers one 1 KiB-aligned KiB of the tile's tangent block: partial first / last store per round
        for (int idx = (o & ~63) + lane; idx < end; idx += 64)
          if (idx >= o) __builtin_nontemporal_store(acc, c2 + idx);
      }
    }
  }
}

extern "C" int stream_mix_fefp_shape_launch(const void* F, const void* s0, void* s1, int64_t ld, void* P, void* ct, int64_t npoints,
                                            int blocks, int lds_bytes, int pre, int per_round, int ppr, int prefetch, int state, void* stream) {
#define DXM_SHAPE(PF, ST, AL)                                                                                                      \
  hipLaunchKernelGGL((stream_mix_fefp_shape_kernel<PF, ST, AL>), dim3(blocks), dim3(256), lds_bytes, (hipStream_t)stream, (const double*)F, \
                     (const double*)s0, (double*)s1, ld, (double*)P, (double*)ct, npoints / 64, pre, per_round, ppr)
  if (state == 3) { DXM_SHAPE(false, 0, true); }   // SoA state, tangent stores on 1 KiB boundaries
  else if (prefetch) { if (state == 0) DXM_SHAPE(true, 0, false); else if (state == 1) DXM_SHAPE(true, 1, false); else DXM_SHAPE(true, 2, false); }
  else               { if (state == 0) DXM_SHAPE(false, 0, false); else if (state == 1) DXM_SHAPE(false, 1, false); else DXM_SHAPE(false, 2, false); }
#undef DXM_SHAPE
  return (int)hipGetLastError();
}

// Clock-stamped variant of the non-temporal probe (box survey): workgroup 0..gridDim-1 records the shader-clock counter
// (s_memtime) and the constant 100 MHz counter (s_memrealtime) on entry and exit into stamps[4 * block .. +3], memory that
// nothing else reads.  Host side: shader clock under THIS load = d(memtime) / d(memrealtime) x 100 MHz, median over
// workgroups (MI355X_MICROARCH.md, DVFS give-back, item 6).
__global__ void __launch_bounds__(256) stream_mix_clock_kernel(const double2_t* __restrict__ rbuf, double2_t* __restrict__ wbuf,
                                                               int64_t ntiles, int rpairs, int wpairs,
                                                               unsigned long long* __restrict__ stamps) {
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t t = wave; t < ntiles; t += nwaves) {
    const double2_t* r = rbuf + t * rpairs;
    double2_t acc = {0.0, 0.0};
#pragma unroll 4
    for (int idx = lane; idx < rpairs; idx += 64) acc += r[idx];
    double2_t* w = wbuf + t * wpairs;
#pragma unroll 8
    for (int idx = lane; idx < wpairs; idx += 64) __builtin_nontemporal_store(acc, w + idx);
  }
  __builtin_amdgcn_s_waitcnt(0);
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) {
    unsigned long long* s = stamps + 4 * (int64_t)blockIdx.x;
    s[0] = c0; s[1] = r0; s[2] = c1; s[3] = r1;
  }
}

extern "C" int stream_mix_clock_launch(const void* rbuf, void* wbuf, int64_t npoints, int read_bytes_per_point,
                                       int write_bytes_per_point, int blocks, void* stamps, void* stream) {
  hipLaunchKernelGGL(stream_mix_clock_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const double2_t*)rbuf,
                     (double2_t*)wbuf, npoints / 64, read_bytes_per_point * 4, write_bytes_per_point * 4,
                     (unsigned long long*)stamps);
  return (int)hipGetLastError();
}
