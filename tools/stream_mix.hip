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
    for (int idx = lane; idx < wpairs; idx += 64) w[idx] = acc;
  }
}

extern "C" int stream_mix_launch(const void* rbuf, void* wbuf, int64_t npoints, int read_bytes_per_point,
                                 int write_bytes_per_point, int blocks, void* stream) {
  hipLaunchKernelGGL(stream_mix_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                     (const double2_t*)rbuf, (double2_t*)wbuf, npoints / 64, read_bytes_per_point * 4,
                     write_bytes_per_point * 4);
  return (int)hipGetLastError();
}
