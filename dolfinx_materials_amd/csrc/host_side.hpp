// host_side.hpp -- the part of libdxmat's host code that needs no GPU: plain C++17, no HIP header.
//
// dxmat.hip includes it for the product; tests/host_side_harness.cpp includes the SAME file and is built twice with
// clang++ -fsanitize=thread and -fsanitize=address,undefined (tests/test_host_side_sanitizers.py, `-m "not gpu"`), so that the
// worker pool, the chunk / staging-ring arithmetic, the page-locked range table, the three bit-exact tangent rebuilds, the
// threaded row moves and the upload-route state machine run under the sanitizers on the CPU box.
//
// What it stands behind, in the reference: the (N, 6, 6) / (N, 9, 9) arrays `integrate` hands back
// (dolfinx_materials/jaxmat.py:231-234, quadrature_map.py:321) and `_update_vals(field, values, cells)`
// (utils.py:136-143), the scatter of results into the quadrature Functions.
#pragma once

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

namespace dxm_host {

typedef double double2_h __attribute__((ext_vector_type(2)));

constexpr double THIRD = 1.0 / 3.0;   // == SS_THIRD of small_strain.hpp (static_assert in dxmat.hip)
constexpr int MAX_CHUNKS = 64;        // launches one host-buffer call is cut into, at most
constexpr int RING = 16;              // slots of the page-locked staging ring
constexpr int FEFP_RECORD = 54;       // building blocks per point of the FeFp tangent (== FEFP_REC of fefp.hpp)

// ------------------------------------------------------------------------------------------
// packed tangent -> full block, bit-identical to what the full-tangent kernels store
// ------------------------------------------------------------------------------------------
// The small-strain tangent is Ct = c1 1x1 + c2 I + c3 n x n: nine numbers per point.  The expression is the kernel's own
// (small_strain.hpp, step 7: t0 + k3 (ni nj) as one fused multiply-add).
#if !defined(__HIP_DEVICE_COMPILE__)
__attribute__((target("fma")))
#endif
inline void expand_coef_tangent(const double* __restrict__ s, double* __restrict__ d, int64_t n) {
  const bool aligned = (reinterpret_cast<uintptr_t>(d) & 15) == 0;
  for (int64_t p = 0; p < n; ++p, s += 9, d += 36) {
    const double k1 = s[0], k2 = s[1], k3 = s[2];
    const double* nv = s + 3;
    double o[36];
    for (int i = 0; i < 6; ++i)
      for (int j = 0; j < 6; ++j) {
        const double t0 = ((i < 3 && j < 3) ? k1 : 0.0) + ((i == j) ? k2 : 0.0);
        o[i * 6 + j] = __builtin_fma(k3, nv[i] * nv[j], t0);
      }
    if (aligned) {   // streaming stores: the block is not read again by these threads
      for (int k = 0; k < 36; k += 2)
        __builtin_nontemporal_store(double2_h{o[k], o[k + 1]}, reinterpret_cast<double2_h*>(d + k));
    } else {
      for (int k = 0; k < 36; ++k) d[k] = o[k];
    }
  }
}

// The same block from 32 B/point: (c1, c2, c3, w) and the STRESS, which crosses PCIe anyway.  The kernel builds its
// tangent with n = dev(sigma) w (small_strain.hpp, steps 3 and 5); the three lines that form n are repeated here with
// every operation individually rounded (no contraction), so the block is the kernel's, bit for bit.
// With `rows` (a map over a subset of the cells): point p belongs in row rows[p] of the caller's arrays -- its block goes to
// dbase + rows[p] * 36 and its stress, which landed in the library's own page-locked area, to fdst + rows[p] * 6.
#if !defined(__HIP_DEVICE_COMPILE__)
__attribute__((target("fma")))
#endif
inline void expand_pack4_tangent(const double* __restrict__ sg, const double* __restrict__ cw, double* __restrict__ dbase, int64_t n,
                                 const int64_t* __restrict__ rows = nullptr, double* __restrict__ fdst = nullptr) {
#pragma clang fp contract(off)
  const bool aligned = (reinterpret_cast<uintptr_t>(dbase) & 15) == 0;
  for (int64_t p = 0; p < n; ++p, sg += 6, cw += 4) {
    double* d = dbase + (rows ? rows[p] : p) * 36;
    if (rows) {
      double* f = fdst + rows[p] * 6;
      for (int k = 0; k < 6; ++k) f[k] = sg[k];
    }
    const double k1 = cw[0], k2 = cw[1], k3 = cw[2], w = cw[3];
    const double third = (sg[0] + sg[1] + sg[2]) * THIRD;
    double nv[6];
    nv[0] = (sg[0] - third) * w; nv[1] = (sg[1] - third) * w; nv[2] = (sg[2] - third) * w;
    nv[3] = sg[3] * w; nv[4] = sg[4] * w; nv[5] = sg[5] * w;
    double o[36];
    for (int i = 0; i < 6; ++i)
      for (int j = 0; j < 6; ++j) {
        const double t0 = ((i < 3 && j < 3) ? k1 : 0.0) + ((i == j) ? k2 : 0.0);
        const double nij = nv[i] * nv[j];
        o[i * 6 + j] = __builtin_fma(k3, nij, t0);
      }
    if (aligned) {
      for (int k = 0; k < 36; k += 2)
        __builtin_nontemporal_store(double2_h{o[k], o[k + 1]}, reinterpret_cast<double2_h*>(d + k));
    } else {
      for (int k = 0; k < 36; ++k) d[k] = o[k];
    }
  }
}

// The 21 upper-triangle entries (i <= j, row by row: the kernels' TL_SYM order) of the same block from the same 32 B/point: what a
// handle with the "sym" tangent layout receives in its host-buffer calls instead of 168 B/point over PCIe.  Entry (i, j) is the
// expression of expand_pack4_tangent, operation for operation, so the 21 numbers are those 21 of the full block, bit for bit.
#if !defined(__HIP_DEVICE_COMPILE__)
__attribute__((target("fma")))
#endif
inline void expand_pack4_tangent_sym(const double* __restrict__ sg, const double* __restrict__ cw, double* __restrict__ d, int64_t n) {
#pragma clang fp contract(off)
  for (int64_t p = 0; p < n; ++p, sg += 6, cw += 4, d += 21) {
    const double k1 = cw[0], k2 = cw[1], k3 = cw[2], w = cw[3];
    const double third = (sg[0] + sg[1] + sg[2]) * THIRD;
    double nv[6];
    nv[0] = (sg[0] - third) * w; nv[1] = (sg[1] - third) * w; nv[2] = (sg[2] - third) * w;
    nv[3] = sg[3] * w; nv[4] = sg[4] * w; nv[5] = sg[5] * w;
    int t = 0;
    for (int i = 0; i < 6; ++i)
      for (int j = i; j < 6; ++j) {
        const double t0 = ((i < 3 && j < 3) ? k1 : 0.0) + ((i == j) ? k2 : 0.0);
        const double nij = nv[i] * nv[j];
        d[t++] = __builtin_fma(k3, nij, t0);
      }
  }
}

// FeFp: the 9x9 block from its 54 building blocks (fefp.hpp step 6):
//   A[row=(i,J)][col=(k,L)] = Vc[col] Fi[J][i] + Wc[col] Sr[row] + U[i][L] Fi[J][k] + (i==k) g[L][J]
// evaluated as the kernel evaluates it (one product, three fused multiply-adds, the Kronecker delta as a 0/1 factor).
// (rows / fdst / pk as in expand_pack4_tangent: block to dbase + rows[p] * 81, the stress pk[p] to fdst + rows[p] * 9)
#if !defined(__HIP_DEVICE_COMPILE__)
__attribute__((target("fma")))
#endif
inline void expand_fefp_tangent(const double* __restrict__ s, double* __restrict__ dbase, int64_t n, const int64_t* __restrict__ rows = nullptr,
                                double* __restrict__ fdst = nullptr, const double* __restrict__ pk = nullptr) {
  static const int TI[9] = {0, 1, 2, 0, 1, 0, 2, 1, 2}, TJ[9] = {0, 1, 2, 1, 0, 2, 0, 2, 1};
  for (int64_t p = 0; p < n; ++p, s += FEFP_RECORD) {
    double* d = dbase + (rows ? rows[p] : p) * 81;
    if (rows) {
      double* f = fdst + rows[p] * 9;
      for (int k = 0; k < 9; ++k) f[k] = pk[p * 9 + k];
    }
    const double* fi = s;
    for (int r = 0; r < 9; ++r) {
      const int i = TI[r], J = TJ[r];
      for (int c = 0; c < 9; ++c) {
        const int k = TI[c], L = TJ[c];
        double t = s[9 + c] * fi[J * 3 + i];
        t = __builtin_fma(s[27 + c], s[36 + r], t);
        t = __builtin_fma(s[18 + i * 3 + L], fi[J * 3 + k], t);
        t = __builtin_fma(i == k ? 1.0 : 0.0, s[45 + L * 3 + J], t);
        d[r * 9 + c] = t;
      }
    }
  }
}

// elastic law: the same constant block for every point (python_materials/elasticity.py:15-19)
inline void fill_const_tangent(const double* __restrict__ s /* lambda, mu */, double* __restrict__ dbase, int64_t n, const int64_t* __restrict__ rows = nullptr,
                               double* __restrict__ fdst = nullptr, const double* __restrict__ sg = nullptr) {
  double o[36];
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) o[i * 6 + j] = ((i < 3 && j < 3) ? s[0] : 0.0) + ((i == j) ? 2.0 * s[1] : 0.0);
  for (int64_t p = 0; p < n; ++p) {
    double* d = dbase + (rows ? rows[p] : p) * 36;
    for (int k = 0; k < 36; ++k) d[k] = o[k];
    if (rows)
      for (int k = 0; k < 6; ++k) fdst[rows[p] * 6 + k] = sg[p * 6 + k];
  }
}

// ------------------------------------------------------------------------------------------
// worker pool: a few persistent threads per handle (created on the first host-path call that needs them)
// ------------------------------------------------------------------------------------------
struct HostPool {
  // stride 9: J2 coefficients -> 6x6, 4: (c1, c2, c3, w) + the stress rows `aux` -> 6x6, -4: the same source -> the 21
  // upper-triangle entries, 54: FeFp building blocks -> 9x9, 0: constant block, -1: plain copy of n BYTES, -8: rows of `tag`
  // doubles from a contiguous block to rows `rows` of `dst` (a state field into the Function of a map over a subset of the cells)
  struct Job { const double* src; double* dst; int64_t n; int stride; int tag; const double* aux; const int64_t* rows; double* dst2; };
  std::vector<std::thread> threads;
  std::mutex mu;
  std::condition_variable cv, cv_done, cv_copy;
  std::deque<Job> queue;
  int pending = 0;
  int pending_copy[MAX_CHUNKS] = {};   // plain copies still running, per tag (the chunk they belong to)
  bool stop = false;
  explicit HostPool(int nthreads) {
    for (int t = 0; t < nthreads; ++t) threads.emplace_back([this] { run(); });
  }
  HostPool(const HostPool&) = delete;
  HostPool& operator=(const HostPool&) = delete;
  ~HostPool() {
    { std::lock_guard<std::mutex> lk(mu); stop = true; }
    cv.notify_all();
    for (auto& t : threads) t.join();
  }
  void run() {
    for (;;) {
      Job j;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [this] { return stop || !queue.empty(); });
        if (queue.empty()) return;
        j = queue.front();
        queue.pop_front();
      }
      if (j.stride == 9) expand_coef_tangent(j.src, j.dst, j.n);
      else if (j.stride == 4) expand_pack4_tangent(j.aux, j.src, j.dst, j.n, j.rows, j.dst2);
      else if (j.stride == -4) expand_pack4_tangent_sym(j.aux, j.src, j.dst, j.n);
      else if (j.stride == FEFP_RECORD) expand_fefp_tangent(j.src, j.dst, j.n, j.rows, j.dst2, j.aux);
      else if (j.stride == -1) memcpy(j.dst, j.src, (size_t)j.n);
      else if (j.stride == -8) {
        const int w = j.tag;
        for (int64_t p = 0; p < j.n; ++p)
          for (int k = 0; k < w; ++k) j.dst[j.rows[p] * w + k] = j.src[p * w + k];
      }
      else fill_const_tangent(j.src, j.dst, j.n, j.rows, j.dst2, j.aux);
      {
        std::lock_guard<std::mutex> lk(mu);
        if (j.stride == -1) { if (--pending_copy[j.tag] == 0) cv_copy.notify_all(); }
        else if (--pending == 0) cv_done.notify_all();
      }
    }
  }
  // rows [0, n) of one chunk, cut into one piece per thread
  // (rows != nullptr: dst / dst2 are the BASES of the caller's tangent / flux arrays, rows the index of this chunk, aux the
  // stress of the chunk where it landed)
  void submit(const double* src, double* dst, int64_t n, int stride, const double* aux = nullptr, const int64_t* rows = nullptr, double* dst2 = nullptr) {
    if (n <= 0) return;
    const int64_t pieces = (int64_t)threads.size();
    const int64_t per = (n + pieces - 1) / pieces;
    const int nf = stride == FEFP_RECORD ? 9 : 6;
    const int in_stride = stride == -4 ? 4 : stride;                                      // doubles per point of the source
    const int out_stride = stride == FEFP_RECORD ? 81 : (stride == -4 ? 21 : 36);         // ... and of the destination
    std::lock_guard<std::mutex> lk(mu);
    for (int64_t o = 0; o < n; o += per) {
      queue.push_back(Job{src + o * in_stride, rows ? dst : dst + o * out_stride, std::min(per, n - o), stride, 0, aux ? aux + o * nf : nullptr,
                          rows ? rows + o : nullptr, dst2});
      ++pending;
    }
    cv.notify_all();
  }
  // point p of a contiguous (n, width) block -> row rows[p] of the caller's (M, width) array, cut over the threads
  void submit_scatter(const double* src, double* dst_base, const int64_t* rows, int64_t n, int width) {
    if (n <= 0) return;
    const int64_t pieces = (int64_t)threads.size();
    const int64_t per = (n + pieces - 1) / pieces;
    std::lock_guard<std::mutex> lk(mu);
    for (int64_t o = 0; o < n; o += per) {
      queue.push_back(Job{src + o * width, dst_base, std::min(per, n - o), -8, width, nullptr, rows + o, nullptr});
      ++pending;
    }
    cv.notify_all();
  }
  void wait() {
    std::unique_lock<std::mutex> lk(mu);
    cv_done.wait(lk, [this] { return pending == 0; });
  }
  // `bytes` from src to dst, cut over the threads and queued AHEAD of any rebuild work; wait_copy(tag) returns when
  // every piece submitted under that tag (0 .. MAX_CHUNKS-1) has been copied
  void copy_async(const void* src, void* dst, size_t bytes, int tag) {
    if (bytes <= (256u << 10)) {   // waking the threads costs more than copying this much
      memcpy(dst, src, bytes);
      return;
    }
    const size_t pieces = std::min<size_t>(threads.size(), 8);
    const size_t per = ((bytes + pieces - 1) / pieces + 63) / 64 * 64;
    {
      std::lock_guard<std::mutex> lk(mu);
      for (size_t o = 0; o < bytes; o += per) {
        queue.push_front(Job{reinterpret_cast<const double*>(static_cast<const char*>(src) + o),
                             reinterpret_cast<double*>(static_cast<char*>(dst) + o), (int64_t)std::min(per, bytes - o), -1, tag, nullptr, nullptr, nullptr});
        ++pending_copy[tag];
      }
    }
    cv.notify_all();
  }
  void wait_copy(int tag) {
    std::unique_lock<std::mutex> lk(mu);
    cv_copy.wait(lk, [this, tag] { return pending_copy[tag] == 0; });
  }
};

// ------------------------------------------------------------------------------------------
// chunk planner of the host-buffer form, ring slots, status-record capacity
// ------------------------------------------------------------------------------------------
// Large batches are cut into up to MAX_CHUNKS chunks (multiples of 256 points; dxmat.hip caps the three-stream scheme at 24).  The
// last chunk's host expansion is not hidden behind any transfer: many small chunks keep that tail short.
struct ChunkPlan {
  int nchunks;
  int64_t csize;   // points per chunk (a multiple of 256); chunk c covers [c * csize, min(n, (c + 1) * csize))
  int64_t offset(int c) const { return (int64_t)c * csize; }
  int64_t count(int c, int64_t n) const {
    const int64_t o = offset(c);
    return o >= n ? 0 : ((n - o) < csize ? (n - o) : csize);
  }
  int issued(int64_t n) const { return csize > 0 ? (int)std::min<int64_t>(nchunks, (n + csize - 1) / csize) : 0; }   // chunks with at least one point
};

inline ChunkPlan plan_chunks(int64_t n, bool packed, bool staged_upload, int max_chunks, bool pipeline) {
  int nchunks = (int)std::min<int64_t>(n / (packed ? (n >= 2097152 ? 65536 : 32768) : 131072), MAX_CHUNKS * 1024);
  if (nchunks < 1) nchunks = 1;
  if (!packed && nchunks > (staged_upload ? 32 : 8)) nchunks = staged_upload ? 32 : 8;   // staged uploads start later: shorter chunks
  if (nchunks > max_chunks) nchunks = max_chunks;
  if (nchunks > MAX_CHUNKS) nchunks = MAX_CHUNKS;
  if (!pipeline) nchunks = 1;
  const int64_t csize = ((n + nchunks - 1) / nchunks + 255) / 256 * 256;
  return ChunkPlan{nchunks, csize};
}

// slot of the page-locked ring that stages chunk c of a pageable gradient array; chunk c may be staged once the copy
// kernel of chunk c - RING has read the slot
inline int ring_slot(int chunk) { return chunk % RING; }

// One status record per workgroup and launch.  A single launch has at most num_cu * 256 workgroups (the largest grid
// option "blocks_per_cu" allows); the chunked host path appends the records of up to MAX_CHUNKS launches, each of
// min(ceil(chunk / 256), num_cu * blocks_per_cu) workgroups: never more than one record per 256 points plus one partial
// block per chunk.
inline int stats_capacity(int num_cu, int64_t npoints) {
  return (int)std::min<int64_t>(INT32_MAX, std::max<int64_t>((int64_t)num_cu * 256, (npoints + 255) / 256 + MAX_CHUNKS));
}
// workgroups (= records) of one launch over cnt points
inline int launch_grid(int64_t cnt, int num_cu, int blocks_per_cu) {
  const int64_t tiles = (cnt + 255) / 256;
  const int64_t cap = (int64_t)num_cu * blocks_per_cu;
  return (int)std::max<int64_t>(1, std::min<int64_t>(tiles, cap));
}

// ------------------------------------------------------------------------------------------
// host ranges this library has page-locked itself (dxm_host_alloc, dxm_host_register): start -> bytes
// ------------------------------------------------------------------------------------------
struct LockedTable {
  std::mutex mu;
  std::map<uintptr_t, size_t> ranges;
  void note(const void* p, size_t bytes) {
    std::lock_guard<std::mutex> lk(mu);
    ranges[reinterpret_cast<uintptr_t>(p)] = bytes;
  }
  void forget(const void* p) {
    std::lock_guard<std::mutex> lk(mu);
    ranges.erase(reinterpret_cast<uintptr_t>(p));
  }
  // the whole of [host, host + bytes) inside ONE noted range
  bool contains(const void* host, size_t bytes) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(host);
    std::lock_guard<std::mutex> lk(mu);
    auto it = ranges.upper_bound(a);
    if (it == ranges.begin()) return false;
    --it;
    return a >= it->first && a + bytes <= it->first + it->second;
  }
};

// ------------------------------------------------------------------------------------------
// threaded host copies
// ------------------------------------------------------------------------------------------
inline void host_copy(void* dst, const void* src, uint64_t bytes, int threads) {
  if (bytes == 0) return;
  int nt = threads > 0 ? threads : 8;
  if (nt > 64) nt = 64;
  if (bytes < (uint64_t)(4u << 20) || nt == 1) { memcpy(dst, src, bytes); return; }
  const uint64_t per = ((bytes + nt - 1) / nt + 4095) / 4096 * 4096;
  std::vector<std::thread> pool;
  for (uint64_t o = per; o < bytes; o += per)
    pool.emplace_back([=] { memcpy(static_cast<char*>(dst) + o, static_cast<const char*>(src) + o, (size_t)std::min<uint64_t>(per, bytes - o)); });
  memcpy(dst, src, (size_t)std::min<uint64_t>(per, bytes));
  for (auto& t : pool) t.join();
}

// rows of `width` doubles moved through an index, the i-range cut over `threads` threads: what a QuadratureMap over a SUBSET of
// the cells does with every result array per update (utils.py:136-143 `array[index] = values`; numpy's fancy assignment runs on
// one core: 1 s per 1e7 x 36 doubles)
inline void move_rows(bool scatter, double* dst, const double* src, const int64_t* rows, int64_t n, int width, int threads) {
  if (n <= 0 || width <= 0) return;
  int nt = threads > 0 ? threads : 8;
  if (nt > 64) nt = 64;
  if ((uint64_t)n * width < (uint64_t)(1u << 18)) nt = 1;
  auto work = [=](int64_t a, int64_t b) {
    const size_t bytes = sizeof(double) * width;
    for (int64_t i = a; i < b; ++i) {
      if (scatter) memcpy(dst + rows[i] * width, src + i * width, bytes);
      else memcpy(dst + i * width, src + rows[i] * width, bytes);
    }
  };
  const int64_t per = (n + nt - 1) / nt;
  std::vector<std::thread> pool;
  for (int64_t a = per; a < n; a += per) pool.emplace_back(work, a, std::min<int64_t>(a + per, n));
  work(0, std::min<int64_t>(per, n));
  for (auto& t : pool) t.join();
}

// smallest and largest entry of an index (the range check of dxm_integrate_rows: rows outside the caller's arrays would
// make the worker threads store 288 B blocks out of bounds), on a few threads: ~1 ms per 1e7 entries
inline void index_min_max(const int64_t* rows, int64_t n, int threads, int64_t* lo, int64_t* hi) {
  *lo = INT64_MAX;
  *hi = INT64_MIN;
  if (n <= 0) return;
  int nt = threads > 0 ? threads : 8;
  if (nt > 64) nt = 64;
  if (n < (1 << 18)) nt = 1;
  const int64_t per = (n + nt - 1) / nt;
  std::vector<int64_t> los(nt, INT64_MAX), his(nt, INT64_MIN);
  auto work = [&](int t, int64_t a, int64_t b) {
    int64_t l = INT64_MAX, h = INT64_MIN;
    for (int64_t i = a; i < b; ++i) { l = std::min(l, rows[i]); h = std::max(h, rows[i]); }
    los[t] = l; his[t] = h;
  };
  std::vector<std::thread> pool;
  int t = 1;
  for (int64_t a = per; a < n; a += per, ++t) pool.emplace_back(work, t, a, std::min<int64_t>(a + per, n));
  work(0, 0, std::min<int64_t>(per, n));
  for (auto& th : pool) th.join();
  for (int k = 0; k < nt; ++k) { *lo = std::min(*lo, los[k]); *hi = std::max(*hi, his[k]); }
}

// ------------------------------------------------------------------------------------------
// which way a pageable gradient array goes up (option register_input)
// ------------------------------------------------------------------------------------------
// Page-locking the caller's array for the call and uploading by DMA (way 1) is 2-3 ms ahead of staging it through the ring
// (way 2) on most boxes of the pool and 5-9 ms behind on some.  With option register_input = 1 the handle finds out: its first
// call is not judged (one-time set-up), calls 2-5 alternate between the two ways, the faster one (by the call's whole
// duration; page-locking wins a tie within 5 %) is kept, and the other gets one call in 32 to prove itself.
// 2 = always page-lock, 0 = always stage.  Registrations that turn out expensive (small pages: above 10 ms/GB, three in a row)
// or are refused send the next 20 calls through the ring.
struct UploadChooser {
  int opt = 1;
  int calls = 0, pref = 1, since_probe = 0;
  bool probing = false;
  double ms[3] = {0.0, 0.0, 0.0};   // [1] page-locked for the call, [2] staged: best of the calibration calls, then a running mean
  int register_skip = 0, register_calls = 0, register_slow = 0;

  void set_option(int value) {
    opt = value;
    calls = 0; pref = 1; since_probe = 0; probing = false; ms[1] = ms[2] = 0.0;
  }
  // 0: staging is the only route (option 0); 1: try to page-lock; 2: stage
  int choose() {
    if (opt == 0) return 0;
    int way = 1;
    if (opt == 1) {
      const int k = calls;
      if (k >= 1 && k <= 4) way = (k % 2 == 1) ? 1 : 2;
      else if (k > 4) {
        way = pref;
        if (++since_probe >= 32) { way = 3 - pref; probing = true; since_probe = 0; }
      }
    }
    if (way == 1 && register_skip > 0) { --register_skip; way = 2; }
    return way;
  }
  // a registration of `bytes` succeeded after ms_lock (+ what the previous call paid to release its range)
  void registered(double ms_total, size_t bytes) {
    const double ms_per_gb = ms_total > 0.5 ? (ms_total - 0.5) / ((double)bytes / 1e9) : 0.0;   // half a millisecond of fixed cost is fine for any size
    // (the first registration of a handle also pays for one-time set-up in the runtime and is not judged)
    if (register_calls++ > 0 && ms_per_gb > 10.0) {
      if (++register_slow >= 3) { register_skip = 20; register_slow = 0; }
    } else {
      register_slow = 0;
    }
  }
  void refused() { register_skip = 20; }
  // the call that went `way` took call_ms as a whole; returns true when the preferred way changed
  bool record(int way, double call_ms) {
    if (opt != 1 || way == 0) return false;
    const int k = calls++;
    if (k == 0) return false;
    const int before = pref;
    if (k <= 4) {
      ms[way] = ms[way] == 0.0 ? call_ms : std::min(ms[way], call_ms);
      if (k == 4) pref = (ms[2] > 0.0 && ms[1] > 0.0 && ms[2] < 0.95 * ms[1]) ? 2 : 1;
    } else if (probing) {
      probing = false;
      const bool better = way == 2 ? call_ms < 0.95 * ms[pref] : call_ms < 1.05 * ms[pref];
      if (way != pref && better) pref = way;
      ms[way] = call_ms;
    } else {
      ms[way] = 0.75 * ms[way] + 0.25 * call_ms;
    }
    return pref != before;
  }
};

}  // namespace dxm_host
