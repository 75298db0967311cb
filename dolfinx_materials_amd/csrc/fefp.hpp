// Finite-strain FeFp J2 plasticity kernel (placeholder until the kernel lands).
#pragma once
#include "dxm_common.hpp"
namespace dxm {
constexpr int FEFP_SLOT_P = 0;    // p
constexpr int FEFP_SLOT_BE = 1;   // be_bar, Mandel (user-visible ISV)
constexpr int FEFP_SLOT_CPI = 7;  // isochoric Cp^-1, Mandel (hidden)
constexpr int FEFP_NSLOTS = 13;
__global__ void __launch_bounds__(BLOCK)
fefp_kernel(const LawParams prm, const int64_t n, const double* __restrict__ F,
            const double* __restrict__ s0, double* __restrict__ s1, const int64_t ld,
            double* __restrict__ P, double* __restrict__ ct, BlockStats* __restrict__ stats) {}
}  // namespace dxm
