// Host-side speed of the tangent rebuild alone (no GPU, no PCIe): the (N,9) coefficient form expanded to the (N,36)
// block with the expression of dxmat.hip::expand_coef_tangent, T threads on contiguous slices, non-temporal stores.
// Tells whether the PCIe-inclusive host path is bounded by this loop or by what shares the memory system with it.
//   /opt/rocm/lib/llvm/bin/clang++ -O3 -std=c++17 -pthread -o host_expand_probe tools/host_expand_probe.cpp   (the host compiler of hipcc) && ./host_expand_probe 10000000 8 16 32
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

typedef double double2_t __attribute__((ext_vector_type(2)));

__attribute__((target("fma"))) static void expand(const double* __restrict__ s, double* __restrict__ d, int64_t n) {
  for (int64_t p = 0; p < n; ++p, s += 9, d += 36) {
    const double k1 = s[0], k2 = s[1], k3 = s[2];
    const double* nv = s + 3;
    double o[36];
    for (int i = 0; i < 6; ++i)
      for (int j = 0; j < 6; ++j) {
        const double t0 = ((i < 3 && j < 3) ? k1 : 0.0) + ((i == j) ? k2 : 0.0);
        o[i * 6 + j] = __builtin_fma(k3, nv[i] * nv[j], t0);
      }
    for (int k = 0; k < 36; k += 2)
      __builtin_nontemporal_store(double2_t{o[k], o[k + 1]}, reinterpret_cast<double2_t*>(d + k));
  }
}

int main(int argc, char** argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 10000000;
  double* src = static_cast<double*>(aligned_alloc(64, sizeof(double) * n * 9));
  double* dst = static_cast<double*>(aligned_alloc(64, sizeof(double) * n * 36));
  for (int64_t k = 0; k < n * 9; ++k) src[k] = 1.0 + 1e-9 * (double)(k % 1000);
  for (int a = 2; a < argc; ++a) {
    const int T = atoi(argv[a]);
    double best = 1e30;
    for (int rep = 0; rep < 6; ++rep) {
      const auto t0 = std::chrono::steady_clock::now();
      std::vector<std::thread> th;
      const int64_t per = (n + T - 1) / T;
      for (int t = 0; t < T; ++t) {
        const int64_t lo = t * per, hi = lo + per < n ? lo + per : n;
        if (lo < hi) th.emplace_back([=] { expand(src + lo * 9, dst + lo * 36, hi - lo); });
      }
      for (auto& x : th) x.join();
      const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      if (rep > 0 && dt < best) best = dt;
    }
    printf("{\"points\": %lld, \"threads\": %d, \"ms\": %.3f, \"Mpoints_per_s\": %.1f, \"GBs_written\": %.1f, \"GBs_read_plus_written\": %.1f}\n",
           (long long)n, T, best * 1e3, n / best / 1e6, n * 288.0 / best / 1e9, n * 360.0 / best / 1e9);
  }
  free(src);
  free(dst);
  return 0;
}
