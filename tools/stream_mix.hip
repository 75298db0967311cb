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
