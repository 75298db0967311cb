// Sanitizer harness for dolfinx_materials_amd/csrc/host_side.hpp (the GPU-free host code of libdxmat.so).
//
//   clang++ -std=c++17 -O1 -g -fsanitize=thread            host_side_harness.cpp -o harness_tsan
//   clang++ -std=c++17 -O1 -g -fsanitize=address,undefined host_side_harness.cpp -o harness_asan
//   harness_* <input.bin> <output.bin>
//
// Built and run by tests/test_host_side_sanitizers.py (`-m "not gpu"`), which writes the input arrays, compares the rebuilt
// tangent blocks in <output.bin> with oracle/host_rebuild_np.py and fails on any sanitizer report.  The "GPU" of the chunk
// pipeline is played by plain memcpy on the calling thread; everything else is the code the product runs.
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>

#include "../dolfinx_materials_amd/csrc/host_side.hpp"

using namespace dxm_host;

static int g_failures = 0;
#define CHECK(cond, ...)                                    \
  do {                                                      \
    if (!(cond)) {                                          \
      ++g_failures;                                         \
      fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__);  \
      fprintf(stderr, __VA_ARGS__);                         \
      fprintf(stderr, "\n");                                \
    }                                                       \
  } while (0)

struct Input {
  int64_t n = 0, M = 0;
  std::vector<double> coef, sg, cw, rec, pk, lm;
  std::vector<int64_t> rows;
};

static bool read_input(const char* path, Input& in) {
  FILE* f = fopen(path, "rb");
  if (!f) return false;
  int64_t hdr[2];
  if (fread(hdr, sizeof(int64_t), 2, f) != 2) { fclose(f); return false; }
  in.n = hdr[0];
  in.M = hdr[1];
  auto rd = [&](std::vector<double>& v, size_t count) { v.resize(count); return count == 0 || fread(v.data(), sizeof(double), count, f) == count; };
  bool ok = rd(in.coef, in.n * 9) && rd(in.sg, in.n * 6) && rd(in.cw, in.n * 4) && rd(in.rec, in.n * 54) && rd(in.pk, in.n * 9) && rd(in.lm, 2);
  in.rows.resize(in.n);
  ok = ok && (in.n == 0 || fread(in.rows.data(), sizeof(int64_t), in.n, f) == (size_t)in.n);
  fclose(f);
  return ok;
}

static bool same_bits(const std::vector<double>& a, const std::vector<double>& b) {
  return a.size() == b.size() && (a.empty() || memcmp(a.data(), b.data(), a.size() * sizeof(double)) == 0);
}

// ---- A. the chunk pipeline of run_and_download (dxmat.hip) with the device played by memcpy ---------------------------------
// Per chunk: stage it into its ring slot on the worker threads (`ahead` chunks ahead), wait for the copy, "upload" the slot,
// hand the chunk's packed tangent to the workers.  Returns the rebuilt blocks; checks the staged gradient arrived intact.
struct Rebuilt { std::vector<double> ct_coef, ct_pack4, ct_fefp, ct_const, ct_sym; };

static Rebuilt run_pipeline(const Input& in, int threads, int max_chunks, int ahead, HostPool* shared = nullptr) {
  const int64_t n = in.n;
  Rebuilt out;
  out.ct_coef.assign(n * 36, -1.0);
  out.ct_pack4.assign(n * 36, -1.0);
  out.ct_fefp.assign(n * 81, -1.0);
  out.ct_const.assign(n * 36, -1.0);
  out.ct_sym.assign(n * 21, -1.0);
  HostPool* own = shared ? nullptr : new HostPool(threads);
  HostPool& pool = shared ? *shared : *own;
  const ChunkPlan plan = plan_chunks(n, true, true, max_chunks, true);
  const int ng = 6;
  std::vector<double> ring((size_t)plan.csize * ng * RING, 0.0), d_grad((size_t)n * ng, 0.0);
  const double* host_grad = in.sg.data();   // any (n, 6) array serves as the caller's pageable gradient
  bool slot_busy[RING] = {};
  auto stage = [&](int p) {
    const int64_t cnt = plan.count(p, n);
    if (p >= plan.nchunks || cnt == 0) return;
    CHECK(!slot_busy[ring_slot(p)], "ring slot %d still holds chunk %d when chunk %d is staged", ring_slot(p), p - RING, p);
    slot_busy[ring_slot(p)] = true;
    pool.copy_async(host_grad + plan.offset(p) * ng, ring.data() + (size_t)ring_slot(p) * plan.csize * ng, sizeof(double) * cnt * ng, p);
  };
  if (!shared) pool.submit(in.lm.data(), out.ct_const.data(), n, 0);   // the constant block: nothing to wait for
  for (int p = 0; p < ahead; ++p) stage(p);
  for (int c = 0; c < plan.nchunks; ++c) {
    const int64_t off = plan.offset(c), cnt = plan.count(c, n);
    if (cnt == 0) break;
    stage(c + ahead);
    pool.wait_copy(c);
    memcpy(d_grad.data() + off * ng, ring.data() + (size_t)ring_slot(c) * plan.csize * ng, sizeof(double) * cnt * ng);   // the copy kernel
    slot_busy[ring_slot(c)] = false;                                                                                      // ring_done[slot]
    pool.submit(in.coef.data() + off * 9, out.ct_coef.data() + off * 36, cnt, 9);
    pool.submit(in.cw.data() + off * 4, out.ct_pack4.data() + off * 36, cnt, 4, in.sg.data() + off * 6);
    pool.submit(in.cw.data() + off * 4, out.ct_sym.data() + off * 21, cnt, -4, in.sg.data() + off * 6);   // "sym" handles: 21 entries per point
    pool.submit(in.rec.data() + off * 54, out.ct_fefp.data() + off * 81, cnt, 54);
  }
  if (shared) {
    pool.submit(in.lm.data(), out.ct_const.data(), n, 0);
  }
  for (int t = 0; t < MAX_CHUNKS; ++t) pool.wait_copy(t);
  pool.wait();
  CHECK(n == 0 || memcmp(d_grad.data(), host_grad, sizeof(double) * n * ng) == 0, "staged gradient differs (max_chunks %d, ahead %d)", max_chunks, ahead);
  delete own;
  return out;
}

static void test_pipeline(const Input& in, FILE* fout) {
  const int64_t n = in.n;
  // serial reference: the same routines on this thread
  Rebuilt ref;
  ref.ct_coef.assign(n * 36, 0.0); ref.ct_pack4.assign(n * 36, 0.0); ref.ct_fefp.assign(n * 81, 0.0); ref.ct_const.assign(n * 36, 0.0);
  expand_coef_tangent(in.coef.data(), ref.ct_coef.data(), n);
  expand_pack4_tangent(in.sg.data(), in.cw.data(), ref.ct_pack4.data(), n);
  expand_fefp_tangent(in.rec.data(), ref.ct_fefp.data(), n);
  fill_const_tangent(in.lm.data(), ref.ct_const.data(), n);
  // the 21-entry form is the upper triangle of the 36-entry one, bit for bit
  ref.ct_sym.assign(n * 21, 0.0);
  expand_pack4_tangent_sym(in.sg.data(), in.cw.data(), ref.ct_sym.data(), n);
  for (int64_t p = 0; p < n; ++p) {
    int t = 0;
    for (int i = 0; i < 6; ++i)
      for (int j = i; j < 6; ++j, ++t)
        CHECK(memcmp(&ref.ct_sym[p * 21 + t], &ref.ct_pack4[p * 36 + i * 6 + j], 8) == 0 && memcmp(&ref.ct_pack4[p * 36 + i * 6 + j], &ref.ct_pack4[p * 36 + j * 6 + i], 8) == 0,
              "sym rebuild: entry (%d, %d) of point %" PRId64, i, j, p);
    if (g_failures > 5) break;
  }
  const int chunk_caps[] = {1, 2, 3, 7, 16, 17, 31, 33, 64};
  for (int cap : chunk_caps)
    for (int ahead : {1, 3, RING - 2})
      for (int threads : {1, 5, 16}) {
        if (threads == 16 && cap != 64 && cap != 17) continue;   // keep the sanitizer runs short
        Rebuilt got = run_pipeline(in, threads, cap, ahead);
        CHECK(same_bits(got.ct_coef, ref.ct_coef) && same_bits(got.ct_pack4, ref.ct_pack4) && same_bits(got.ct_fefp, ref.ct_fefp) && same_bits(got.ct_const, ref.ct_const) &&
                  same_bits(got.ct_sym, ref.ct_sym),
              "pipeline result differs from the serial rebuild (max_chunks %d, ahead %d, %d threads)", cap, ahead, threads);
      }
  // rows mode (dxm_integrate_rows): blocks and stress delivered through the index into arrays of M rows
  const int64_t M = in.M;
  std::vector<double> ct_rows(M * 36, 7.0), fx_rows(M * 6, 7.0), ct9_rows(M * 81, 7.0), fx9_rows(M * 9, 7.0), ctc_rows(M * 36, 7.0), fxc_rows(M * 6, 7.0), fld6_rows(M * 6, 7.0), fld1_rows(M, 7.0);
  {
    HostPool pool(16);
    const ChunkPlan plan = plan_chunks(n, true, false, 64, true);
    for (int c = 0; c < plan.nchunks; ++c) {
      const int64_t off = plan.offset(c), cnt = plan.count(c, n);
      if (cnt == 0) break;
      pool.submit(in.cw.data() + off * 4, ct_rows.data(), cnt, 4, in.sg.data() + off * 6, in.rows.data() + off, fx_rows.data());
      pool.submit(in.rec.data() + off * 54, ct9_rows.data(), cnt, 54, in.pk.data() + off * 9, in.rows.data() + off, fx9_rows.data());
      pool.submit(in.lm.data(), ctc_rows.data(), cnt, 0, in.sg.data() + off * 6, in.rows.data() + off, fxc_rows.data());
      pool.submit_scatter(in.sg.data() + off * 6, fld6_rows.data(), in.rows.data() + off, cnt, 6);      // a bound state field of 6 components
      pool.submit_scatter(in.cw.data() + off, fld1_rows.data(), in.rows.data() + off, cnt, 1);          // ... and a scalar one (the first n doubles of cw as an (n, 1) field)
    }
    pool.wait();
  }
  std::vector<char> hit(M, 0);
  for (int64_t p = 0; p < n; ++p) {
    const int64_t r = in.rows[p];
    hit[r] = 1;
    CHECK(memcmp(ct_rows.data() + r * 36, ref.ct_pack4.data() + p * 36, 288) == 0 && memcmp(fx_rows.data() + r * 6, in.sg.data() + p * 6, 48) == 0, "rows mode (pack4): point %" PRId64, p);
    CHECK(memcmp(ct9_rows.data() + r * 81, ref.ct_fefp.data() + p * 81, 648) == 0 && memcmp(fx9_rows.data() + r * 9, in.pk.data() + p * 9, 72) == 0, "rows mode (fefp): point %" PRId64, p);
    CHECK(memcmp(ctc_rows.data() + r * 36, ref.ct_const.data() + p * 36, 288) == 0 && memcmp(fxc_rows.data() + r * 6, in.sg.data() + p * 6, 48) == 0, "rows mode (constant): point %" PRId64, p);
    CHECK(memcmp(fld6_rows.data() + r * 6, in.sg.data() + p * 6, 48) == 0 && fld1_rows[r] == in.cw[p], "rows mode (state fields): point %" PRId64, p);
    if (g_failures > 5) break;
  }
  for (int64_t r = 0; r < M; ++r)
    if (!hit[r]) { CHECK(ct_rows[r * 36] == 7.0 && fx_rows[r * 6] == 7.0 && ct9_rows[r * 81 + 80] == 7.0, "rows mode wrote row %" PRId64 " that no point maps to", r); if (g_failures > 5) break; }
  if (fout) {
    fwrite(ref.ct_coef.data(), sizeof(double), ref.ct_coef.size(), fout);
    fwrite(ref.ct_pack4.data(), sizeof(double), ref.ct_pack4.size(), fout);
    fwrite(ref.ct_fefp.data(), sizeof(double), ref.ct_fefp.size(), fout);
    fwrite(ref.ct_const.data(), sizeof(double), ref.ct_const.size(), fout);
  }
  printf("ok pipeline (n = %" PRId64 ")\n", n);
}

// ---- B. many producers ------------------------------------------------------------------------------------------------------
static void test_many_producers(const Input& in) {
  Rebuilt ref = run_pipeline(in, 4, 16, 3);
  // one pool per producer (HIPMaterial(devices=[...]): one handle, one pool per GPU, side by side)
  {
    std::vector<std::thread> producers;
    std::vector<Rebuilt> got(6);
    for (int t = 0; t < 6; ++t) producers.emplace_back([&, t] { got[t] = run_pipeline(in, 3 + t, 8 + 9 * t, 1 + t % 3); });
    for (auto& th : producers) th.join();
    for (int t = 0; t < 6; ++t)
      CHECK(same_bits(got[t].ct_coef, ref.ct_coef) && same_bits(got[t].ct_pack4, ref.ct_pack4) && same_bits(got[t].ct_fefp, ref.ct_fefp), "producer %d with its own pool", t);
  }
  // several producers on ONE pool, each with its own copy tags (tags are per chunk: disjoint ranges of them here)
  {
    HostPool pool(8);
    const int64_t n = in.n;
    std::vector<std::vector<double>> outs(4, std::vector<double>(n * 36, 0.0));
    std::vector<std::vector<double>> copies(4, std::vector<double>(n * 6, 0.0));
    std::vector<std::thread> producers;
    for (int t = 0; t < 4; ++t)
      producers.emplace_back([&, t] {
        const int64_t per = (n + 7) / 8;
        for (int c = 0; c < 8; ++c) {
          const int64_t off = c * per, cnt = std::max<int64_t>(0, std::min(per, n - off));
          if (cnt == 0) break;
          pool.copy_async(in.sg.data() + off * 6, copies[t].data() + off * 6, sizeof(double) * cnt * 6, t * 16 + c);
          pool.submit(in.cw.data() + off * 4, outs[t].data() + off * 36, cnt, 4, in.sg.data() + off * 6);
          pool.wait_copy(t * 16 + c);
        }
        pool.wait();   // waits for everybody's rebuild jobs: later than needed, never earlier
      });
    for (auto& th : producers) th.join();
    for (int t = 0; t < 4; ++t)
      CHECK(same_bits(outs[t], ref.ct_pack4) && memcmp(copies[t].data(), in.sg.data(), sizeof(double) * n * 6) == 0, "producer %d on the shared pool", t);
  }
  // pools created and destroyed while idle, with work queued, and right after work
  // (the destructor finishes whatever is still queued before it joins: the output array must outlive the pool)
  for (int k = 0; k < 50; ++k) {
    std::vector<double> o(64 * 36);
    HostPool pool(1 + k % 7);
    if (k % 3) pool.submit(in.coef.data(), o.data(), std::min<int64_t>(64, in.n), 9);
    if (k % 3 == 1) pool.wait();
  }
  printf("ok many producers\n");
}

// ---- C. page-locked range table -------------------------------------------------------------------------------------------
static void test_locked_table() {
  LockedTable t;
  std::vector<char> a(4096), b(4096);
  t.note(a.data(), 1000);
  CHECK(t.contains(a.data(), 1000) && t.contains(a.data() + 10, 990) && t.contains(a.data() + 999, 1), "sub-ranges of a noted range");
  CHECK(!t.contains(a.data(), 1001) && !t.contains(a.data() + 999, 2) && !t.contains(a.data() + 1000, 1), "ranges that extend past the noted one");
  CHECK(!t.contains(b.data(), 1) && !t.contains(a.data() - 1, 2), "foreign memory");
  t.note(a.data() + 1000, 1000);   // adjacent: a range that spans both is NOT inside one noted range
  CHECK(t.contains(a.data() + 1000, 1000) && !t.contains(a.data() + 500, 1000), "adjacent ranges are not merged");
  t.forget(a.data());
  CHECK(!t.contains(a.data(), 1) && t.contains(a.data() + 1500, 10), "forget removes exactly one range");
  t.note(a.data(), 50);            // the address comes back with a shorter length
  CHECK(t.contains(a.data(), 50) && !t.contains(a.data(), 51), "re-noted with a shorter length");
  t.forget(a.data());
  t.forget(a.data() + 1000);
  t.forget(b.data());              // never noted: no effect
  CHECK(t.ranges.empty(), "table empty at the end");
  // concurrent note / contains / forget on disjoint and on shared addresses
  std::vector<std::vector<char>> bufs(8, std::vector<char>(1 << 12));
  std::atomic<int> wrong{0};
  std::vector<std::thread> ths;
  for (int k = 0; k < 8; ++k)
    ths.emplace_back([&, k] {
      for (int it = 0; it < 2000; ++it) {
        t.note(bufs[k].data(), 1 << 12);
        if (!t.contains(bufs[k].data() + (it % 4000), 8)) ++wrong;
        (void)t.contains(bufs[(k + 1) % 8].data(), 16);   // somebody else's range: either answer, no race
        t.forget(bufs[k].data());
        if (t.contains(bufs[k].data(), 1)) ++wrong;
      }
    });
  for (auto& th : ths) th.join();
  CHECK(wrong.load() == 0, "%d wrong answers under concurrency", wrong.load());
  printf("ok locked table\n");
}

// ---- D. threaded copies and row moves -------------------------------------------------------------------------------------
static void test_copies() {
  std::mt19937_64 rng(5);
  for (int64_t n : {(int64_t)0, (int64_t)1, (int64_t)63, (int64_t)7283, (int64_t)300001}) {
    for (int width : {1, 6, 9, 36}) {
      if (n > 100000 && width == 9) continue;
      const int64_t M = 2 * n + 3;
      std::vector<int64_t> rows(n);
      std::vector<int64_t> perm(M);
      for (int64_t i = 0; i < M; ++i) perm[i] = i;
      std::shuffle(perm.begin(), perm.end(), rng);
      for (int64_t i = 0; i < n; ++i) rows[i] = perm[i];
      std::vector<double> src(n * width), dst(M * width, -3.0), ref(M * width, -3.0), back(n * width, 0.0);
      for (auto& v : src) v = (double)(rng() % 100000) / 7.0;
      for (int64_t i = 0; i < n; ++i) memcpy(ref.data() + rows[i] * width, src.data() + i * width, sizeof(double) * width);
      for (int threads : {1, 16, 64, 0}) {
        std::fill(dst.begin(), dst.end(), -3.0);
        move_rows(true, dst.data(), src.data(), rows.data(), n, width, threads);
        CHECK(same_bits(dst, ref), "scatter n=%" PRId64 " width=%d threads=%d", n, width, threads);
        move_rows(false, back.data(), dst.data(), rows.data(), n, width, threads);
        CHECK(same_bits(back, src), "gather n=%" PRId64 " width=%d threads=%d", n, width, threads);
      }
      int64_t lo, hi;
      index_min_max(rows.data(), n, 16, &lo, &hi);
      if (n > 0) {
        CHECK(lo == *std::min_element(rows.begin(), rows.end()) && hi == *std::max_element(rows.begin(), rows.end()), "index_min_max n=%" PRId64, n);
      } else {
        CHECK(lo == INT64_MAX && hi == INT64_MIN, "index_min_max of an empty index");
      }
    }
  }
  for (uint64_t bytes : {(uint64_t)0, (uint64_t)1, (uint64_t)4095, (uint64_t)(4u << 20), (uint64_t)(4u << 20) + 1, (uint64_t)37000001}) {
    std::vector<char> a(bytes + 1, 0), b(bytes + 1, 0x55);
    for (uint64_t i = 0; i < bytes; ++i) a[i] = (char)(i * 131 + 7);
    for (int threads : {0, 1, 3, 16, 200}) {
      std::fill(b.begin(), b.end(), 0x55);
      host_copy(b.data(), a.data(), bytes, threads);
      CHECK((bytes == 0 || memcmp(a.data(), b.data(), bytes) == 0) && b[bytes] == 0x55, "host_copy of %" PRIu64 " bytes on %d threads", bytes, threads);
    }
  }
  printf("ok copies and row moves\n");
}

// ---- E. chunk planner, ring slots, status-record capacity -----------------------------------------------------------------
static void test_planner() {
  std::vector<int64_t> sizes = {0, 1, 63, 64, 65, 255, 256, 257, 32767, 32768, 32769, 65535, 65536, 131071, 131072, 131073, 1000003, 2097151, 2097152, 2097153,
                                9999999, 10000000, 10000001, 12500000, 19999999, 20000000, 100000000};
  std::mt19937_64 rng(11);
  for (int k = 0; k < 400; ++k) sizes.push_back((int64_t)(rng() % 20000001));
  int64_t checked = 0;
  for (int64_t n : sizes)
    for (int cap = 1; cap <= MAX_CHUNKS; ++cap)
      for (int flags = 0; flags < 8; ++flags) {
        const bool packed = flags & 1, staged = flags & 2, pipeline = flags & 4;
        const ChunkPlan p = plan_chunks(n, packed, staged, cap, pipeline);
        ++checked;
        CHECK(p.nchunks >= 1 && p.nchunks <= cap && p.nchunks <= MAX_CHUNKS && (pipeline || p.nchunks == 1), "nchunks %d (n %" PRId64 ", cap %d, flags %d)", p.nchunks, n, cap, flags);
        CHECK(p.csize % 256 == 0 && (n == 0 || p.csize > 0), "chunk size %" PRId64 " not a multiple of 256", p.csize);
        int64_t covered = 0;
        int64_t records[4] = {0, 0, 0, 0};
        const int bpc[4] = {1, 5, 32, 256};
        for (int c = 0; c < p.nchunks; ++c) {
          const int64_t cnt = p.count(c, n);
          CHECK(cnt == 0 || p.offset(c) == covered, "chunk %d starts at %" PRId64 ", expected %" PRId64, c, p.offset(c), covered);
          CHECK(p.offset(c) % 256 == 0, "chunk offset not a multiple of 256");
          covered += cnt;
          if (cnt > 0)
            for (int b = 0; b < 4; ++b) records[b] += launch_grid(cnt, 256, bpc[b]);
        }
        CHECK(covered == n, "chunks cover %" PRId64 " of %" PRId64 " points", covered, n);
        CHECK(p.issued(n) <= p.nchunks && (n == 0 || p.count(p.issued(n) - 1, n) > 0) && p.count(p.issued(n), n) == 0, "issued() disagrees with count()");
        for (int b = 0; b < 4; ++b)
          CHECK(records[b] <= stats_capacity(256, n), "status records %" PRId64 " exceed the capacity %d (n %" PRId64 ", cap %d, flags %d, %d blocks/CU)", records[b], stats_capacity(256, n), n, cap, flags, bpc[b]);
        if (g_failures > 20) return;
      }
  // a slot is reused by chunk c + RING and by no chunk in between; `ahead` <= RING - 2 chunks in flight never collide
  for (int c = 0; c < 4 * RING; ++c)
    for (int d = 1; d < RING; ++d) CHECK(ring_slot(c) != ring_slot(c + d), "ring slot collision between chunks %d and %d", c, c + d);
  CHECK(ring_slot(5) == ring_slot(5 + RING), "ring period");
  CHECK(stats_capacity(256, 0) == 256 * 256 && stats_capacity(304, (int64_t)1 << 40) == INT32_MAX, "capacity limits");
  printf("ok planner (%" PRId64 " plans)\n", checked);
}

// ---- F. upload-route state machine ------------------------------------------------------------------------------------------
static void test_chooser() {
  {   // option 1: call 1 not judged, calls 2-5 alternate, the faster way is kept, the other is probed every 32nd call
    UploadChooser u;
    std::vector<int> ways;
    auto call = [&](double ms1, double ms2) {
      int w = u.choose();
      if (w == 1) u.registered(1.0, (size_t)480e6);
      ways.push_back(w);
      u.record(w, w == 1 ? ms1 : ms2);
      return w;
    };
    for (int k = 0; k < 5; ++k) call(24.0, 31.0);
    CHECK((ways == std::vector<int>{1, 1, 2, 1, 2}), "calibration sequence");
    CHECK(u.pref == 1 && u.ms[1] == 24.0 && u.ms[2] == 31.0, "page-locking kept (24 vs 31 ms)");
    int probes = 0;
    for (int k = 0; k < 64; ++k) probes += call(24.0, 31.0) == 2;
    CHECK(probes == 2 && u.pref == 1, "two probes of the other way in 64 calls, preference unchanged");
    for (int k = 0; k < 40; ++k) call(40.0, 30.0);   // the host changed: staging is now faster; the next probe finds out
    CHECK(u.pref == 2, "switched to staging after a probe (pref %d)", u.pref);
  }
  {   // a tie within 5 % goes to page-locking
    UploadChooser u;
    for (int k = 0; k < 5; ++k) { int w = u.choose(); u.record(w, w == 1 ? 25.0 : 24.0); }
    CHECK(u.pref == 1, "tie within 5 %% keeps page-locking");
    UploadChooser v;
    for (int k = 0; k < 5; ++k) { int w = v.choose(); v.record(w, w == 1 ? 25.0 : 23.0); }
    CHECK(v.pref == 2, "8 %% faster staging wins");
  }
  {   // expensive registrations (small pages): three in a row after the first send 20 calls through the ring
    UploadChooser u;
    u.set_option(2);
    int staged = 0, total = 0;
    for (int k = 0; k < 30; ++k) {
      int w = u.choose();
      ++total;
      if (w == 1) u.registered(12.0, (size_t)480e6); else ++staged;   // 24 ms/GB
      u.record(w, 30.0);
    }
    // call 1 registers and is not judged, calls 2-4 are slow -> calls 5-24 staged, calls 25-27 slow again -> 28-30 staged
    CHECK(staged == 23 && total == 30, "23 of 30 calls staged around two runs of three slow registrations (got %d)", staged);
    UploadChooser r;
    r.set_option(2);
    CHECK(r.choose() == 1, "option 2 page-locks");
    r.refused();
    for (int k = 0; k < 20; ++k) CHECK(r.choose() == 2, "refusal: staged for 20 calls");
    CHECK(r.choose() == 1, "then one more try");
  }
  {   // option 0 never registers; set_option resets the calibration
    UploadChooser u;
    u.set_option(0);
    for (int k = 0; k < 10; ++k) { CHECK(u.choose() == 0, "option 0"); u.record(0, 20.0); }
    CHECK(u.calls == 0, "option 0 does not count calls");
    u.set_option(1);
    for (int k = 0; k < 7; ++k) { int w = u.choose(); u.record(w, 20.0); }
    u.set_option(1);
    CHECK(u.calls == 0 && u.pref == 1 && u.ms[1] == 0.0 && u.ms[2] == 0.0 && !u.probing, "set_option resets the calibration");
  }
  printf("ok upload chooser\n");
}

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: %s input.bin [output.bin]\n", argv[0]); return 2; }
  Input in;
  if (!read_input(argv[1], in)) { fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
  FILE* fout = argc > 2 ? fopen(argv[2], "wb") : nullptr;
  test_pipeline(in, fout);
  if (fout) fclose(fout);
  test_many_producers(in);
  test_locked_table();
  test_copies();
  test_planner();
  test_chooser();
  // ragged and empty batches through the same pipeline
  for (int64_t n : {(int64_t)0, (int64_t)1, (int64_t)255, (int64_t)257}) {
    Input small = in;
    small.n = std::min(n, in.n);
    small.coef.resize(small.n * 9); small.sg.resize(small.n * 6); small.cw.resize(small.n * 4); small.rec.resize(small.n * 54); small.pk.resize(small.n * 9); small.rows.resize(small.n);
    test_pipeline(small, nullptr);
  }
  if (g_failures) { fprintf(stderr, "%d check(s) failed\n", g_failures); return 1; }
  printf("all ok\n");
  return 0;
}
