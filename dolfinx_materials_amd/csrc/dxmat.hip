// libdxmat.so -- C ABI (include/dxmat.h) over the gfx950 constitutive-update kernels.
//
// Host side of the batch dispatch that the reference performs in
//   JAXMaterial.integrate / DataManager           dolfinx_materials/jaxmat.py:30-43, :208-234
//   Material.integrate / MaterialStateManager     dolfinx_materials/generic.py:176-295
// with the state kept device-resident in SoA form instead of dicts of (N, dim) arrays.
// There is deliberately no CPU implementation behind this ABI.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <thread>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/dxmat.h"
#include "dxm_common.hpp"
#include "fefp.hpp"
#include "gradient.hpp"
#include "small_strain.hpp"
#include "host_side.hpp"

using namespace dxm;
static_assert(TL_FULL == DXM_TANGENT_FULL && TL_SYM == DXM_TANGENT_SYM && TL_COEF == DXM_TANGENT_COEF && TL_PACK4 == DXM_TANGENT_PACK4, "kernel and ABI layout ids");

// ------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------
static thread_local std::string g_last_error;

static int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}

// A failing runtime call is reported through the library's own channel (return code + dxm_last_error) and CONSUMED here: the
// runtime's per-thread "last error" is shared with the host application (torch checks it after its own launches), which must not
// find an error of this library in it (tests/conftest.py checks the state after every GPU test).
#define HIP_TRY(expr)                                                                      \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess) {                                                                \
      (void)hipGetLastError();                                                             \
      return fail(-2, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    }                                                                                      \
  } while (0)

// Selects the handle's device for the duration of a call and restores the caller's current
// device afterwards (the host application, e.g. torch, owns the thread's current device).
struct DeviceGuard {
  int prev = -1;
  bool ok = true;
  explicit DeviceGuard(int device) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != device) ok = (hipSetDevice(device) == hipSuccess);
  }
  ~DeviceGuard() {
    int cur = -1;
    if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
  }
};
#define DEVICE_GUARD(m)                                                        \
  DeviceGuard _guard((m)->device);                                             \
  if (!_guard.ok) return fail(-2, "hipSetDevice(%d) failed", (m)->device)

// ------------------------------------------------------------------------------------------
// law descriptors
// ------------------------------------------------------------------------------------------
struct LawDesc {
  int n_grad, n_flux, n_params;
  int n_isv_fields;                       // user-visible
  int n_fields;                           // incl. hidden state fields (addressable by set/get_state)
  int isv_dim[DXM_MAX_STATE_FIELDS];
  const char* isv_name[DXM_MAX_STATE_FIELDS];
  int isv_slot[DXM_MAX_STATE_FIELDS];     // first SoA slot of the field
  int n_slots;                            // SoA slots incl. hidden ones
  int alg_bytes;
  const char* kernel;
};

static const LawDesc kLaws[DXM_LAW_COUNT] = {
    {6, 6, 2, 0, 0, {0, 0, 0, 0}, {nullptr, nullptr, nullptr, nullptr}, {0, 0, 0, 0}, 0, 384,
     "small_strain_kernel<0"},
    {6, 6, 4, 2, 2, {1, 6, 0, 0}, {"p", "epsp", nullptr, nullptr}, {0, 1, 0, 0}, SS_NSLOTS, 496,
     "small_strain_kernel<1"},
    {6, 6, 5, 2, 2, {1, 6, 0, 0}, {"p", "epsp", nullptr, nullptr}, {0, 1, 0, 0}, SS_NSLOTS, 496,
     "small_strain_kernel<2"},
    // field 2 is hidden state: the isochoric inverse plastic right Cauchy-Green tensor
    {9, 9, 5, 2, 3, {1, 6, 6, 0}, {"p", "be_bar", "cp_bar_inv", nullptr}, {FEFP_SLOT_P, FEFP_SLOT_BE, FEFP_SLOT_CPI, 0}, FEFP_NSLOTS, 976,
     "fefp_kernel<1"},
    {9, 9, 4, 2, 3, {1, 6, 6, 0}, {"p", "be_bar", "cp_bar_inv", nullptr}, {FEFP_SLOT_P, FEFP_SLOT_BE, FEFP_SLOT_CPI, 0}, FEFP_NSLOTS, 976,
     "fefp_kernel<0"},
};

static int tangent_size(const dxm_material* m);

static int isv_total(const LawDesc& d) {
  int t = 0;
  for (int f = 0; f < d.n_isv_fields; ++f) t += d.isv_dim[f];
  return t;
}

// ------------------------------------------------------------------------------------------
// host side that needs no GPU (worker pool, tangent rebuilds, chunk planner, locked-range table, row moves, upload-route
// state machine): host_side.hpp, shared with the sanitizer harness of the CPU test suite
// ------------------------------------------------------------------------------------------
using dxm_host::HostPool;
using dxm_host::expand_coef_tangent;
using dxm_host::expand_pack4_tangent;
using dxm_host::expand_fefp_tangent;
using dxm_host::fill_const_tangent;
static_assert(dxm_host::THIRD == SS_THIRD && dxm_host::FEFP_RECORD == FEFP_REC, "host rebuilds and kernels share these constants");

// ------------------------------------------------------------------------------------------
// handle
// ------------------------------------------------------------------------------------------
constexpr int DXM_MAX_CHUNKS = dxm_host::MAX_CHUNKS;
constexpr int DXM_RING = dxm_host::RING;   // slots of the page-locked staging ring

struct dxm_material {
  int law = 0;
  int device = 0;
  int64_t n = 0;
  int64_t ld = 0;  // SoA leading dimension (n rounded up to 256, plus a 32-double stagger)
  LawParams prm{};
  std::vector<double> raw_params;
  int maxit = 25;
  double rtol = 1e-14;
  double* state[2] = {nullptr, nullptr};  // [n_slots][ld] each
  double* state_base = nullptr;           // one allocation holds s0 and s1
  size_t s1_skew = 0;
  bool s1_alias = false;  // after advance()/revert() s1 == s0 until the next integrate: no copy is made
  BlockStats* d_stats = nullptr;
  BlockStats* h_stats = nullptr;   // page-locked landing of the block records (no DMA into pageable memory)
  int stats_capacity = 0;
  int last_grid = 0;                      // stats records written by the last integrate
  hipStream_t pipe_stream = nullptr;      // second stream of the chunk-pipelined host path
  hipStream_t down_stream2 = nullptr;     // option split_streams: the second of the two download streams (pipe_stream is the first)
  // completion of the last launch: an event owned by the handle, recorded on the caller's stream (the
  // stream itself may be gone by the time the handle is asked to wait: e.g. a torch side stream)
  hipEvent_t last_event = nullptr;
  bool last_event_recorded = false;       // false while the last launch was stream-captured (no event then)
  bool launched = false;
  hipStream_t own_stream = nullptr;
  // launch configuration identity (dxm_launch_generation): epoch changes with parameters / layout /
  // placement / options, parity with every advance that swaps the two state buffers
  uint64_t epoch = 0;
  int parity = 0;
  // options (dxm_set_option)
  bool opt_pipeline = true;               // chunk-pipelined host path
  bool opt_split_streams = true;          // host path, page-locked gradient: uploads + kernels on one stream, downloads on two others (else: whole chunks alternate on two)
  int opt_packed_transfer = 2;            // host path: 0 move the full tangent; 1 its 9 coefficients (J2) / 54 building blocks (FeFp), block rebuilt
                                          // on the host; 2 (small strain) only (c1, c2, c3, w), the direction rebuilt from the stress
  bool opt_fused_gradient = true;         // displacement form: evaluate the gradient inside the update kernel
  bool opt_staged_gradient = true;        // hex8 gradient kernel: nodal data through LDS
  bool opt_verbose = false;
  dxm_host::UploadChooser up;             // option register_input (host path, pageable gradient array): 2 page-lock it for the call (DMA upload), 0 stage it, 1 measure and keep the faster
  int last_upload = DXM_UPLOAD_NONE;      // dxm_stats.upload of the last host-buffer call
  int opt_host_threads = 16;
  int opt_pageable_dma = 0;   // 1: hand pageable host arrays to the runtime (faster uploads; see upload_from_host)
  int64_t opt_packed_min_points = 32768;   // below: waking the workers costs what the bytes save (r02_hostpath_v2.jsonl)
  int opt_max_chunks = DXM_MAX_CHUNKS;
  HostPool* pool = nullptr;
  double* h_coef = nullptr;               // page-locked landing area of the tangent coefficients, h_coef_per_point doubles per point
  int h_coef_per_point = 0;
  double* h_flux = nullptr;               // page-locked (n, 6) landing area of the stress (dxm_integrate_rows)
  double* h_isv = nullptr;                // page-locked landing area of the bound state fields in the rows forms, field after field
  double elastic_lm[2] = {0.0, 0.0};      // lambda, mu handed to the constant-block fill
  hipEvent_t chunk_done[DXM_MAX_CHUNKS] = {};
  hipEvent_t kernel_done[DXM_MAX_CHUNKS] = {};   // option split_streams: what the download stream waits for, per chunk
  int num_cu = 256;
  int blocks_per_cu = 5;
  int tangent_layout = 0;    // DXM_TANGENT_FULL (36 / 81) | DXM_TANGENT_SYM (21) | DXM_TANGENT_COEF (9)
  // host-path staging (device side), allocated on first dxm_integrate
  double* d_grad = nullptr;
  double* d_flux = nullptr;
  // option keep_initial_io: gradient / flux of the initial state s0, kept by swapping with d_grad / d_flux at dxm_advance
  double* d_grad0 = nullptr;
  double* d_flux0 = nullptr;
  int io1_valid = 0;          // bit 0: d_grad holds the gradient of s1, bit 1: d_flux its flux (a completed host-buffer call)
  int io0_valid = 0;          // the same for d_grad0 / d_flux0 and s0
  int opt_keep_initial_io = 0;
  double* d_isv = nullptr;
  double* isv_out[DXM_MAX_STATE_FIELDS] = {};   // dxm_bind_isv_output: host rows the host-buffer forms deliver each field of s1 into
  double* d_isv_fields = nullptr;   // field-major scratch of those deliveries in a call that ALSO fills isv_aos (d_isv holds the AoS rows then)
  double* d_ct = nullptr;
  double* d_field = nullptr;  // (n, <= 6) AoS scratch for set/get_state of one field
  // host-buffer form, strain in ordinary memory: page-locked ring the kernels read the chunks from (zero-copy)
  double* h_grad_ring = nullptr;
  int64_t ring_slot_doubles = 0;
  hipEvent_t ring_done[DXM_RING] = {};
  int opt_stage_ahead = 3;
  double unregister_ms = 0.0;   // what releasing the call-scoped page-lock of the gradient array took in the last call
};

static int sync_last(dxm_material* m);

static void free_state(dxm_material* m) {
  if (!m->state_base) return;
  (void)hipFree(m->state_base);
  m->state_base = nullptr;
}

// number of parameters a law takes in THIS build: a JIT build with a user-supplied hardening law
// (DXM_CUSTOM_HARDENING) takes [E, nu, sig0, c0..c5] for the two "voce" slots
static int n_params_of(int law) {
#ifdef DXM_CUSTOM_HARDENING
  if (law == DXM_LAW_J2_VOCE || law == DXM_LAW_FEFP_J2_VOCE) return 9;
#endif
  return kLaws[law].n_params;
}

static int build_params(dxm_material* m, const double* p, int np) {
  const int expect = n_params_of(m->law);
  if (np != expect) return fail(-1, "law %d expects %d parameters, got %d", m->law, expect, np);
  const double E = p[0], nu = p[1];
  if (!(E > 0.0) || !(nu > -1.0 && nu < 0.5)) return fail(-1, "invalid elastic constants E=%g nu=%g", E, nu);
  LawParams q{};
  q.lambda = E * nu / (1 + nu) / (1 - 2 * nu);  // python_materials/elasticity.py:12-13
  q.mu = E / 2 / (1 + nu);
  q.kappa = q.lambda + 2.0 * q.mu / 3.0;
  q.sig0 = 1.0;
  switch (m->law) {
    case DXM_LAW_ELASTIC_ISO: break;
    case DXM_LAW_J2_LINEAR:
    case DXM_LAW_FEFP_J2_LINEAR: q.sig0 = p[2]; q.h1 = p[3]; break;
    case DXM_LAW_J2_VOCE:
    case DXM_LAW_FEFP_J2_VOCE:
#ifdef DXM_CUSTOM_HARDENING
      q.sig0 = p[2];
      for (int k = 0; k < 6; ++k) q.c[k] = p[3 + k];
#else
      q.sig0 = p[2]; q.h1 = p[3]; q.h2 = p[4];
#endif
      break;
  }
  q.maxit = m->maxit;
  // relative to the initial yield stress, floored so that a law with R(0) = 0 keeps a reachable tolerance
  q.rtol = m->rtol;
  q.tol = m->rtol * fmax(fabs(q.sig0), 2e-8 * q.mu);
  m->prm = q;
  m->raw_params.assign(p, p + np);
  return 0;
}

static int tangent_size(const dxm_material* m) {
  const LawDesc& d = kLaws[m->law];
  if (m->tangent_layout == DXM_TANGENT_SYM) return d.n_flux * (d.n_flux + 1) / 2;
  if (m->tangent_layout == DXM_TANGENT_COEF) return 9;
  if (m->tangent_layout == DXM_TANGENT_PACK4) return 4;
  return d.n_flux * d.n_grad;
}

// ------------------------------------------------------------------------------------------
// small helper kernels (not on the hot path)
// ------------------------------------------------------------------------------------------
__global__ void fill_slot_kernel(double* dst, int64_t count, double value) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) dst[i] = value;
}

struct PackMap { int n; int slot[16]; };

// SoA state -> AoS (n, total) internal-state-variable array (jaxmat.py:46-58 `_hcat_mixed`).
// Strain chunks from the page-locked ring to HBM: a copy kernel on a few workgroups (the PCIe link, not the CUs, bounds
// it), so that the rest of the chip stays free for the update kernel and the downloads of the other stream.
__global__ void __launch_bounds__(256) ring_upload_kernel(const double2_t* __restrict__ src, double2_t* __restrict__ dst, int64_t n16) {
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n16; k += (int64_t)gridDim.x * blockDim.x)
    dst[k] = __builtin_nontemporal_load(src + k);
}

__global__ void pack_isv_kernel(const double* __restrict__ soa, int64_t ld, int64_t n,
                                double* __restrict__ aos, PackMap map) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = n * map.n;
  if (t >= total) return;
  const int64_t i = t / map.n;
  const int k = (int)(t - i * map.n);
  aos[t] = soa[(int64_t)map.slot[k] * ld + i];
}

// AoS (n, map.n) -> SoA state slots (set_initial_state_dict upload).
__global__ void unpack_isv_kernel(double* __restrict__ soa, int64_t ld, int64_t n,
                                  const double* __restrict__ aos, PackMap map) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = n * map.n;
  if (t >= total) return;
  const int64_t i = t / map.n;
  const int k = (int)(t - i * map.n);
  soa[(int64_t)map.slot[k] * ld + i] = aos[t];
}

// ------------------------------------------------------------------------------------------
// ABI
// ------------------------------------------------------------------------------------------
extern "C" {

int dxm_abi_version(void) { return DXM_ABI_VERSION; }

int dxm_has_custom_hardening(void) {
#ifdef DXM_CUSTOM_HARDENING
  return 1;
#else
  return 0;
#endif
}

const char* dxm_last_error(void) { return g_last_error.c_str(); }

int dxm_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) return fail(-2, "hipGetDeviceCount: %s", hipGetErrorString(e));
  return n;
}

int dxm_law_info_get(int law, dxm_law_info* out) {
  if (law < 0 || law >= DXM_LAW_COUNT || !out) return fail(-1, "unknown law id %d", law);
  const LawDesc& d = kLaws[law];
  memset(out, 0, sizeof(*out));
  out->n_grad = d.n_grad;
  out->n_flux = d.n_flux;
  out->n_params = n_params_of(law);
  out->n_isv_fields = d.n_isv_fields;
  for (int f = 0; f < DXM_MAX_STATE_FIELDS; ++f) {
    out->isv_dim[f] = d.isv_dim[f];
    out->isv_name[f] = d.isv_name[f];
  }
  out->n_isv_total = isv_total(d);
  out->algorithmic_bytes_per_point = d.alg_bytes;
  return 0;
}

static int init_state(dxm_material* m) {
  const LawDesc& d = kLaws[m->law];
  if (d.n_slots == 0) return 0;
  const size_t bytes = (size_t)d.n_slots * m->ld * sizeof(double);
  for (int w = 0; w < 2; ++w) {
    HIP_TRY(hipMemsetAsync(m->state[w], 0, bytes, m->own_stream));
    if (kLaws[m->law].n_grad == 9) {
      // be_bar = Cp^-1 = identity: unstressed natural configuration
      // (demos/jax/finite_strain_elastoplasticity/finite_strain_elastoplasticity.py:181)
      const int ones[6] = {FEFP_SLOT_BE + 0, FEFP_SLOT_BE + 1, FEFP_SLOT_BE + 2,
                           FEFP_SLOT_CPI + 0, FEFP_SLOT_CPI + 1, FEFP_SLOT_CPI + 2};
      for (int s : ones) {
        const int blocks = (int)((m->ld + 255) / 256);
        hipLaunchKernelGGL(fill_slot_kernel, dim3(blocks), dim3(256), 0, m->own_stream,
                           m->state[w] + (size_t)s * m->ld, m->ld, 1.0);
      }
    }
  }
  HIP_TRY(hipStreamSynchronize(m->own_stream));
  return 0;
}

dxm_material* dxm_create(int law, const double* params, int n_params, int64_t npoints, int device) {
  if (law < 0 || law >= DXM_LAW_COUNT) { fail(-1, "unknown law id %d", law); return nullptr; }
  if (npoints < 0) { fail(-1, "negative point count"); return nullptr; }
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) {
    fail(-2, "no usable HIP device (hipGetDeviceCount: %s, count %d); libdxmat has no CPU fallback",
         hipGetErrorString(e), ndev);
    return nullptr;
  }
  if (device < 0 || device >= ndev) { fail(-1, "device %d out of range [0,%d)", device, ndev); return nullptr; }
  dxm_material* m = new dxm_material();
  m->law = law;
  m->device = device;
  m->n = npoints;
  // SoA leading dimension: N rounded up to 256 plus 32 doubles.  With a slot stride that is a
  // multiple of 2 KiB the s0 read streams and s1 write streams of all slots stay congruent and the
  // kernel falls into a slow mode (0.91 vs 0.83 ms at 1e7 points, bimodal by allocation address);
  // a 256 B stagger per slot removes it (profiles/archive/r01_tune_state_stride.txt).
  m->ld = ((npoints + 255) / 256) * 256 + 32;
  auto bail = [&](void) -> dxm_material* { dxm_destroy(m); return nullptr; };
  if (build_params(m, params, n_params) != 0) return bail();
  DeviceGuard guard(device);
  if (!guard.ok) { fail(-2, "hipSetDevice(%d) failed", device); return bail(); }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) m->num_cu = prop.multiProcessorCount;
  if (hipStreamCreateWithFlags(&m->own_stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&m->last_event, hipEventDisableTiming) != hipSuccess) {
    fail(-2, "hipStreamCreate / hipEventCreate failed"); return bail();
  }
  const LawDesc& d = kLaws[law];
  if (d.n_slots > 0) {
    const size_t bytes = (size_t)d.n_slots * m->ld * sizeof(double);
    {
      if (hipMalloc(&m->state_base, 2 * bytes + m->s1_skew) != hipSuccess) {
        fail(-3, "hipMalloc of %zu state bytes failed", 2 * bytes + m->s1_skew); return bail();
      }
      m->state[0] = m->state_base;
      m->state[1] = reinterpret_cast<double*>(reinterpret_cast<char*>(m->state_base) + bytes + m->s1_skew);
    }
  }
  // grid size, in workgroups per CU.  FeFp: the workgroups that are resident at once (persistent
  // grid).  Small strain: 32 = 8 x the resident 4: a grid of exactly-resident workgroups starts all
  // waves together and keeps them in lockstep (everybody loads, then everybody stores); workgroups
  // that are dispatched as others retire spread those phases, 3-6 % faster at 1e7 points in the fast
  // placement mode and 10 % in the slow one (tools/grid_sweep.py, profiles/archive/r01_grid_sweep.jsonl).
  {
    int occ = 0;
    const void* fn = nullptr;
    switch (law) {
      case DXM_LAW_ELASTIC_ISO: fn = (const void*)small_strain_kernel<LAW_ELASTIC, TL_FULL>; break;
      case DXM_LAW_J2_LINEAR: fn = (const void*)small_strain_kernel<LAW_J2_LINEAR, TL_FULL>; break;
      case DXM_LAW_J2_VOCE: fn = (const void*)small_strain_kernel<LAW_J2_VOCE, TL_FULL>; break;
      case DXM_LAW_FEFP_J2_LINEAR: fn = (const void*)fefp_kernel<0, 0>; break;
      default: fn = (const void*)fefp_kernel<1, 0>; break;
    }
    // residency from the kernel's own resources (the occupancy API over-reports on ROCm 7.2):
    // waves/SIMD by allocated VGPRs (512-entry file, granule 8), workgroups by LDS (160 KiB/CU)
    hipFuncAttributes fa;
    if (hipFuncGetAttributes(&fa, fn) == hipSuccess && fa.numRegs > 0) {
      const int alloc = ((fa.numRegs + 7) / 8) * 8;
      int waves_simd = 512 / alloc;
      if (waves_simd > 8) waves_simd = 8;
      int by_vgpr = waves_simd * 4 / WAVES_PER_BLOCK;
      int by_lds = fa.sharedSizeBytes > 0 ? (int)(160 * 1024 / fa.sharedSizeBytes) : 8;
      occ = by_vgpr < by_lds ? by_vgpr : by_lds;
      if (occ < 1) occ = 1;
      m->blocks_per_cu = occ;
    }
    if (law == DXM_LAW_ELASTIC_ISO || law == DXM_LAW_J2_LINEAR) m->blocks_per_cu = 32;
    // The laws with a local Newton iteration (Voce or traced hardening, FeFp) do a point-dependent amount of work per tile: a
    // grid of one workgroup per 256 points (up to 256 per CU, the cap; a grid-stride loop beyond) lets the dispatcher balance
    // it -- Voce +3 %, FeFp +2.5 % over the persistent grids above at 1e7 points (profiles/archive/r03_grid_size_by_law.txt); the
    // linear-hardening kernel, whose tiles all cost the same, loses 2.5 % with it and keeps 32.
    if (law == DXM_LAW_J2_VOCE || law == DXM_LAW_FEFP_J2_VOCE || law == DXM_LAW_FEFP_J2_LINEAR) m->blocks_per_cu = 256;
  }
  // one record per workgroup and launch.  A single launch has at most num_cu * 256 workgroups (the largest grid
  // dxm_set_option("blocks_per_cu") allows); the chunked host path appends the records of up to DXM_MAX_CHUNKS
  // launches, each of min(ceil(chunk / 256), num_cu * blocks_per_cu) workgroups: never more than one record per
  // 256 points plus one partial block per chunk
  m->stats_capacity = dxm_host::stats_capacity(m->num_cu, npoints);
  if (hipMalloc(&m->d_stats, sizeof(BlockStats) * m->stats_capacity) != hipSuccess) {
    fail(-3, "hipMalloc of stats failed"); return bail();
  }
  if (init_state(m) != 0) return bail();
  g_last_error.clear();
  return m;
}

int dxm_destroy(dxm_material* m) {
  if (!m) return 0;
  DeviceGuard guard(m->device);
  (void)sync_last(m);
  delete m->pool;
  if (m->h_coef) (void)hipHostFree(m->h_coef);
  if (m->h_flux) (void)hipHostFree(m->h_flux);
  if (m->h_isv) (void)hipHostFree(m->h_isv);
  for (hipEvent_t e : m->chunk_done) if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : m->kernel_done) if (e) (void)hipEventDestroy(e);
  if (m->last_event) (void)hipEventDestroy(m->last_event);
  free_state(m);
  if (m->d_stats) (void)hipFree(m->d_stats);
  if (m->h_stats) (void)hipHostFree(m->h_stats);
  if (m->d_grad) (void)hipFree(m->d_grad);
  if (m->d_flux) (void)hipFree(m->d_flux);
  if (m->d_grad0) (void)hipFree(m->d_grad0);
  if (m->d_flux0) (void)hipFree(m->d_flux0);
  if (m->d_isv) (void)hipFree(m->d_isv);
  if (m->d_isv_fields) (void)hipFree(m->d_isv_fields);
  if (m->d_ct) (void)hipFree(m->d_ct);
  if (m->d_field) (void)hipFree(m->d_field);
  if (m->h_grad_ring) (void)hipHostFree(m->h_grad_ring);
  for (hipEvent_t e : m->ring_done) if (e) (void)hipEventDestroy(e);
  if (m->own_stream) (void)hipStreamDestroy(m->own_stream);
  if (m->pipe_stream) (void)hipStreamDestroy(m->pipe_stream);
  if (m->down_stream2) (void)hipStreamDestroy(m->down_stream2);
  delete m;
  return 0;
}

int64_t dxm_npoints(const dxm_material* m) { return m ? m->n : -1; }
int dxm_law(const dxm_material* m) { return m ? m->law : -1; }

int dxm_set_params(dxm_material* m, const double* params, int n_params) {
  if (!m || !params) return fail(-1, "null argument");
  if (int rc = build_params(m, params, n_params)) return rc;   // nothing changed: captured graphs stay valid
  ++m->epoch;
  return 0;
}

int dxm_set_tangent_layout(dxm_material* m, int layout) {
  if (!m) return fail(-1, "null handle");
  if (layout != DXM_TANGENT_FULL && layout != DXM_TANGENT_SYM && layout != DXM_TANGENT_COEF && layout != DXM_TANGENT_PACK4)
    return fail(-1, "unknown tangent layout %d", layout);
  if (layout != DXM_TANGENT_FULL && kLaws[m->law].n_grad == 9)
    return fail(-1, "the FeFp tangent dP/dF is not symmetric: only DXM_TANGENT_FULL is available");
  if ((layout == DXM_TANGENT_COEF || layout == DXM_TANGENT_PACK4) && m->law == DXM_LAW_ELASTIC_ISO)
    return fail(-1, "the elastic tangent is the constant lambda 1x1 + 2 mu I: there are no per-point coefficients");
  m->tangent_layout = layout;
  ++m->epoch;
  return 0;
}

int dxm_tangent_size(const dxm_material* m) { return m ? tangent_size(m) : -1; }

int dxm_set_newton(dxm_material* m, int maxit, double rtol) {
  if (!m) return fail(-1, "null handle");
  if (maxit < 1 || !(rtol > 0.0)) return fail(-1, "invalid Newton controls maxit=%d rtol=%g", maxit, rtol);
  m->maxit = maxit;
  m->rtol = rtol;
  if (int rc = build_params(m, m->raw_params.data(), (int)m->raw_params.size())) return rc;
  ++m->epoch;
  return 0;
}

static int check_field(const dxm_material* m, int which, int field) {
  if (!m) return fail(-1, "null handle");
  if (which != DXM_S0 && which != DXM_S1) return fail(-1, "state selector must be DXM_S0 or DXM_S1");
  const LawDesc& d = kLaws[m->law];
  if (field < 0 || field >= d.n_fields) return fail(-1, "law %d has no state field %d", m->law, field);
  return 0;
}

// Wait for the last launch on the handle.  The event belongs to the handle; if the last launch went
// into a stream capture (no event can be recorded there) the whole device is waited for.
static int sync_last(dxm_material* m) {
  if (!m->launched) return 0;
  if (m->last_event_recorded) {
    HIP_TRY(hipEventSynchronize(m->last_event));
  } else {
    HIP_TRY(hipDeviceSynchronize());
  }
  return 0;
}

static double* state_of(const dxm_material* m, int which) {
  return m->state[(which == DXM_S1 && !m->s1_alias) ? 1 : 0];
}

// A write to s1 while it is served from s0 needs its own storage first.
static int materialize_s1(dxm_material* m) {
  if (!m->s1_alias) return 0;
  const LawDesc& d = kLaws[m->law];
  if (d.n_slots > 0)
    HIP_TRY(hipMemcpy(m->state[1], m->state[0], (size_t)d.n_slots * m->ld * sizeof(double),
                      hipMemcpyDeviceToDevice));
  m->s1_alias = false;
  return 0;
}


// No DMA to or from pageable memory.  The runtime page-locks such a range on the fly and keeps the mapping in a cache
// keyed by address; when the owner has freed that memory since and a later allocation lands on the same address, the
// cached mapping is stale and the copy dies with "Memory access fault by GPU ... on address <host address>" -- as a
// write into a read-only page (download into a range once pinned as an upload source) or as a read through a dead
// mapping (upload).  Seen about once in 25 runs of the GPU test suite, whose tests allocate and free large numpy
// arrays all the time, as QuadratureMap.update does with its gradient arrays.  So: page-locked or registered memory
// is used directly; anything else goes through this page-locked staging (two halves in flight) and a CPU copy.
constexpr size_t BOUNCE_BYTES = 16u << 20;
// Host ranges this library has page-locked itself (dxm_host_alloc, dxm_host_register): start -> bytes.
static dxm_host::LockedTable g_locked;
static std::atomic<int> g_query_foreign{1};   // ask the runtime about pointers that are not in the table (option "query_foreign_pointers")
static void note_locked(const void* p, size_t bytes) { g_locked.note(p, bytes); }
static void forget_locked(const void* p) { g_locked.forget(p); }
static bool in_locked_table(const void* host, size_t bytes) { return g_locked.contains(host, bytes); }
static bool page_locked_byte(const void* host) {
  hipPointerAttribute_t attr{};
  const bool locked = host && hipPointerGetAttributes(&attr, host) == hipSuccess && attr.type == hipMemoryTypeHost;
  (void)hipGetLastError();   // "not a registered pointer" is the expected answer for ordinary memory
  return locked;
}
// Both ends of [host, host + bytes): an array that starts inside a registered / page-locked block but extends past it
// (a view into a larger buffer, a dxm_host_register of a shorter length) must take the staged route too.
static bool page_locked(const void* host, size_t bytes) {
  if (!host) return false;
  if (in_locked_table(host, bytes)) return true;
  if (!g_query_foreign.load()) return false;
  if (!page_locked_byte(host)) return false;
  return bytes <= 1 || page_locked_byte(static_cast<const char*>(host) + bytes - 1);
}
struct Staging {
  char* buf = nullptr;
  hipEvent_t done[2] = {nullptr, nullptr};
};
static std::mutex g_staging_mu[64];   // one per device: handles on different GPUs stage concurrently
static Staging g_staging[64];
static int current_device_index(int* dev) {
  HIP_TRY(hipGetDevice(dev));
  if (*dev < 0 || *dev >= 64) return fail(-1, "device index %d out of range", *dev);
  return 0;
}
static int staging_for_current_device(Staging** out) {
  int dev = 0;
  if (int rc = current_device_index(&dev)) return rc;
  Staging& s = g_staging[dev];
  if (!s.buf) {
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&s.buf), 2 * BOUNCE_BYTES, hipHostMallocDefault));
    for (hipEvent_t& e : s.done) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  *out = &s;
  return 0;
}
static size_t chunk_size(size_t c, size_t nchunks, size_t bytes) { return c + 1 < nchunks ? BOUNCE_BYTES : bytes - c * BOUNCE_BYTES; }

// device -> caller-owned host memory; complete on return
static int download_to_host(void* host, const void* dev, size_t bytes, hipStream_t st) {
  if (bytes == 0) return 0;
  if (page_locked(host, bytes)) {
    HIP_TRY(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return 0;
  }
  int devi = 0;
  if (int rc = current_device_index(&devi)) return rc;
  std::lock_guard<std::mutex> lk(g_staging_mu[devi]);
  Staging* s = nullptr;
  if (int rc = staging_for_current_device(&s)) return rc;
  const size_t nchunks = (bytes + BOUNCE_BYTES - 1) / BOUNCE_BYTES;
  for (size_t c = 0; c <= nchunks; ++c) {
    if (c < nchunks) {
      HIP_TRY(hipMemcpyAsync(s->buf + (c & 1) * BOUNCE_BYTES, static_cast<const char*>(dev) + c * BOUNCE_BYTES,
                             chunk_size(c, nchunks, bytes), hipMemcpyDeviceToHost, st));
      HIP_TRY(hipEventRecord(s->done[c & 1], st));
    }
    if (c > 0) {   // the previous chunk lands while this one is in flight
      HIP_TRY(hipEventSynchronize(s->done[(c - 1) & 1]));
      memcpy(static_cast<char*>(host) + (c - 1) * BOUNCE_BYTES, s->buf + ((c - 1) & 1) * BOUNCE_BYTES, chunk_size(c - 1, nchunks, bytes));
    }
  }
  return 0;
}

// caller-owned host memory -> device; the host range may be reused on return (the device copy is complete too)
static int upload_from_host(void* dev, const void* host, size_t bytes, hipStream_t st) {
  if (bytes == 0) return 0;
  if (page_locked(host, bytes)) {
    HIP_TRY(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    return 0;
  }
  int devi = 0;
  if (int rc = current_device_index(&devi)) return rc;
  std::lock_guard<std::mutex> lk(g_staging_mu[devi]);
  Staging* s = nullptr;
  if (int rc = staging_for_current_device(&s)) return rc;
  const size_t nchunks = (bytes + BOUNCE_BYTES - 1) / BOUNCE_BYTES;
  for (size_t c = 0; c < nchunks; ++c) {
    if (c >= 2) HIP_TRY(hipEventSynchronize(s->done[c & 1]));   // this half has left for the device
    memcpy(s->buf + (c & 1) * BOUNCE_BYTES, static_cast<const char*>(host) + c * BOUNCE_BYTES, chunk_size(c, nchunks, bytes));
    HIP_TRY(hipMemcpyAsync(static_cast<char*>(dev) + c * BOUNCE_BYTES, s->buf + (c & 1) * BOUNCE_BYTES, chunk_size(c, nchunks, bytes),
                           hipMemcpyHostToDevice, st));
    HIP_TRY(hipEventRecord(s->done[c & 1], st));
  }
  HIP_TRY(hipStreamSynchronize(st));
  return 0;
}

int dxm_set_state(dxm_material* m, int which, int field, const double* host_aos) {
  if (int rc = check_field(m, which, field)) return rc;
  if (!host_aos) return fail(-1, "null host pointer");
  DEVICE_GUARD(m);
  if (int rc = sync_last(m)) return rc;
  const LawDesc& d = kLaws[m->law];
  const int dim = d.isv_dim[field];
  const int64_t n = m->n;
  if (n == 0) return 0;
  // s1 served from s0 (after advance / revert) gets its own storage before EITHER is written: writing s1 must not
  // touch s0, and writing s0 must not change what s1 shows (the reference rebinds `data_manager.s0` and leaves s1
  // alone, generic.py:200-201 / jaxmat.py:205-206; found by tests/test_gpu_fuzz_protocol.py: advance, set, advance)
  if (int rc = materialize_s1(m)) return rc;
  // upload the AoS block and transpose on the device (a host-side transposition of 6 x 1e7
  // doubles costs more than the PCIe transfer)
  if (!m->d_field) HIP_TRY(hipMalloc(&m->d_field, sizeof(double) * n * 6));
  PackMap map{};
  map.n = dim;
  for (int c = 0; c < dim; ++c) map.slot[c] = d.isv_slot[field] + c;
  hipStream_t st = m->own_stream;
  if (int rc = upload_from_host(m->d_field, host_aos, sizeof(double) * n * dim, st)) return rc;
  const int blocks = (int)((n * dim + 255) / 256);
  hipLaunchKernelGGL(unpack_isv_kernel, dim3(blocks), dim3(256), 0, st, state_of(m, which), m->ld, n,
                     m->d_field, map);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(st));
  return 0;
}

int dxm_get_state(dxm_material* m, int which, int field, double* host_aos) {
  if (int rc = check_field(m, which, field)) return rc;
  if (!host_aos) return fail(-1, "null host pointer");
  DEVICE_GUARD(m);
  if (int rc = sync_last(m)) return rc;
  const LawDesc& d = kLaws[m->law];
  const int dim = d.isv_dim[field];
  const int64_t n = m->n;
  if (n == 0) return 0;
  if (!m->d_field) HIP_TRY(hipMalloc(&m->d_field, sizeof(double) * n * 6));
  PackMap map{};
  map.n = dim;
  for (int c = 0; c < dim; ++c) map.slot[c] = d.isv_slot[field] + c;
  hipStream_t st = m->own_stream;
  const int blocks = (int)((n * dim + 255) / 256);
  hipLaunchKernelGGL(pack_isv_kernel, dim3(blocks), dim3(256), 0, st, state_of(m, which), m->ld, n,
                     m->d_field, map);
  HIP_TRY(hipGetLastError());
  return download_to_host(host_aos, m->d_field, sizeof(double) * n * dim, st);
}

int dxm_advance(dxm_material* m) {
  if (!m) return fail(-1, "null handle");
  // s0 <- s1 (generic.py:212-213 copies, jaxmat.py:39-40 aliases).  Done by swapping the two
  // device buffers; until the next integrate rewrites s1 in full, s1 reads are served from s0
  // (QuadratureMap.advance reads the final state right after update(): quadrature_map.py:355-356).
  if (!m->s1_alias) {
    double* t = m->state[0];
    m->state[0] = m->state[1];
    m->state[1] = t;
    m->s1_alias = true;
    m->parity ^= 1;   // launches now read / write the other buffer: dxm_launch_generation changes
    // Gradient and flux of the accepted state stay on the device as those of s0 (option keep_initial_io): the buffers the
    // last host-buffer call filled become d_grad0 / d_flux0 and the next call fills the other pair -- no copy.  Without a
    // new state in between (advance twice, advance after revert: s1 is served from s0 and this branch is not entered) s0
    // keeps what it has.  A state that came from a form of call without host arrays (device pointers; a fused displacement
    // for the gradient) or that got its own storage back after revert (dxm_set_state: materialize_s1) brings NO copies
    // along, and s0 then holds none (io0_valid below): the Python layer's lazy s0 mirrors are settled before this call.
    if (m->opt_keep_initial_io && m->io1_valid) {
      if (m->io1_valid & 1) { double* g = m->d_grad0; m->d_grad0 = m->d_grad; m->d_grad = g; }
      if (m->io1_valid & 2) { double* f = m->d_flux0; m->d_flux0 = m->d_flux; m->d_flux = f; }
    }
    // the copies held for the state that has just been replaced belong to nobody now: s0 holds what the accepted state
    // brought along and nothing else (a state accepted from a device-pointer call brings nothing)
    m->io0_valid = m->opt_keep_initial_io ? m->io1_valid : 0;
    m->io1_valid = 0;
  }
  return 0;
}

// which copies the handle holds for state `which`: s1 shows those of s0 while it is served from it (after advance / revert)
static int io_mask(const dxm_material* m, int which) { return (which == DXM_S0 || m->s1_alias) ? m->io0_valid : m->io1_valid; }

int dxm_io_held(const dxm_material* m, int which) {
  if (!m || (which != DXM_S0 && which != DXM_S1)) return -1;
  return io_mask(m, which);
}

int dxm_get_io(dxm_material* m, int which, int kind, double* host_aos) {
  if (!m) return fail(-1, "null handle");
  if (which != DXM_S0 && which != DXM_S1) return fail(-1, "state selector must be DXM_S0 or DXM_S1");
  if (kind != 0 && kind != 1) return fail(-1, "kind must be 0 (gradient) or 1 (flux)");
  if (!(io_mask(m, which) & (1 << kind)))
    return fail(-1, "the %s of that state is not held on the device (a host-buffer integrate; for s0 also option keep_initial_io before dxm_advance)", kind ? "flux" : "gradient");
  if (m->n == 0) return 0;
  if (!host_aos) return fail(-1, "null host pointer");
  DEVICE_GUARD(m);
  if (int rc = sync_last(m)) return rc;
  const LawDesc& d = kLaws[m->law];
  const bool first = which == DXM_S0 || m->s1_alias;
  const double* src = kind ? (first ? m->d_flux0 : m->d_flux) : (first ? m->d_grad0 : m->d_grad);
  return download_to_host(host_aos, src, sizeof(double) * m->n * (kind ? d.n_flux : d.n_grad), m->own_stream);
}

int dxm_revert(dxm_material* m) {
  if (!m) return fail(-1, "null handle");
  m->s1_alias = true;  // s1 <- s0 (generic.py:215-216, jaxmat.py:42-43)
  m->io1_valid = 0;    // d_grad / d_flux belong to the state that was dropped
  return 0;
}

}  // extern "C"

// ---- launch -------------------------------------------------------------------------------
// One launch over the point range [off, off + cnt) (off a multiple of 256): the chunk-pipelined host
// path issues several of these on alternating streams; everything else launches the whole batch.
// The block records of the launch go to m->d_stats[stats_off ...).
template <int LAW>
static void launch_small_strain(dxm_material* m, int grid, hipStream_t st, int64_t off, int64_t cnt,
                                const double* grad, double* flux, double* ct, int stats_off,
                                const MeshSource* fused, int tl) {
  const double* s0 = m->state[0] + off;
  double* s1 = m->state[1] + off;
  BlockStats* bs = m->d_stats + stats_off;
  const MeshSource none{};
  const MeshSource& src = fused ? *fused : none;
  const int g = fused ? fused->kind : 0;   // where the strain comes from: array / hex8 x 8 / tet4 / Lagrange simplex
  // J2 kernels: 2.25 KiB of unused dynamic LDS on top of the static 30.1 KiB keep FOUR workgroups (16 waves) per CU.  The
  // linear-hardening kernel needs 95 VGPRs since the flow direction of the tangent comes from the stress (it was 103):
  // a fifth wave per SIMD would fit and costs 0.65 % (0.8169 vs 0.8116 / 0.8123 ms per 1e7 points in one process,
  // profiles/archive/r03_j2_ab_pack4.jsonl); the elastic kernel keeps its five.
  constexpr int dyn_lds = LAW == LAW_ELASTIC ? 0 : 2304;
#define DXM_LAUNCH_SS(TL, G)                                                                              \
  hipLaunchKernelGGL((small_strain_kernel<LAW, TL, G>), dim3(grid), dim3(BLOCK), dyn_lds, st, m->prm, cnt, grad, s0, s1, \
                     m->ld, flux, ct, bs, src)
#define DXM_LAUNCH_SS_G(TL) do { if (g == 0) DXM_LAUNCH_SS(TL, 0); else if (g == 1) DXM_LAUNCH_SS(TL, 1); \
                                 else if (g == 2) DXM_LAUNCH_SS(TL, 2); else DXM_LAUNCH_SS(TL, 3); } while (0)
  if (tl == TL_SYM) DXM_LAUNCH_SS_G(TL_SYM);
  else if (tl == TL_FULL) DXM_LAUNCH_SS_G(TL_FULL);
  else if constexpr (LAW != LAW_ELASTIC) { if (tl == TL_PACK4) DXM_LAUNCH_SS_G(TL_PACK4); else DXM_LAUNCH_SS_G(TL_COEF); }
#undef DXM_LAUNCH_SS_G
#undef DXM_LAUNCH_SS
}

// tl: tangent layout of THIS launch (the host path may ask for the coefficient form although the handle's
// layout is the full block: it rebuilds the block on the host)
static int launch_range(dxm_material* m, int64_t off, int64_t cnt, const double* grad, double* flux,
                        double* ct, hipStream_t st, int stats_off, int* grid_out,
                        const MeshSource* fused, int tl) {
  if (((uintptr_t)grad | (uintptr_t)flux | (uintptr_t)ct) & 15)
    return fail(-1, "gradient / flux / tangent device arrays must be 16-byte aligned");
  m->io1_valid = 0;   // s1 is being rewritten; a host-buffer call that completes sets it again
  static_assert(WAVE * WAVES_PER_BLOCK == 256, "dxm_host::launch_grid counts workgroups of 256 points");
  const int64_t blocks = dxm_host::launch_grid(cnt, m->num_cu, m->blocks_per_cu);
  if (stats_off + blocks > m->stats_capacity) return fail(-1, "internal: stats buffer too small");
  const int grid = (int)blocks;
  switch (m->law) {
    case DXM_LAW_ELASTIC_ISO: launch_small_strain<LAW_ELASTIC>(m, grid, st, off, cnt, grad, flux, ct, stats_off, fused, tl); break;
    case DXM_LAW_J2_LINEAR: launch_small_strain<LAW_J2_LINEAR>(m, grid, st, off, cnt, grad, flux, ct, stats_off, fused, tl); break;
    case DXM_LAW_J2_VOCE: launch_small_strain<LAW_J2_VOCE>(m, grid, st, off, cnt, grad, flux, ct, stats_off, fused, tl); break;
    case DXM_LAW_FEFP_J2_VOCE:
    case DXM_LAW_FEFP_J2_LINEAR: {
      const double* s0 = m->state[0] + off;
      double* s1 = m->state[1] + off;
      BlockStats* bs = m->d_stats + stats_off;
      const MeshSource none{};
      const bool voce = m->law == DXM_LAW_FEFP_J2_VOCE;
      const MeshSource& fsrc = fused ? *fused : none;
      const int g = fused ? fused->kind : 0;
#define DXM_LAUNCH_FEFP(HARD, G, T) \
  hipLaunchKernelGGL((fefp_kernel<HARD, G, T>), dim3(grid), dim3(BLOCK), 0, st, m->prm, cnt, grad, s0, s1, m->ld, flux, ct, bs, fsrc)
#define DXM_LAUNCH_FEFP_G(HARD, T) do { if (g == 0) DXM_LAUNCH_FEFP(HARD, 0, T); else if (g == 1) DXM_LAUNCH_FEFP(HARD, 1, T); \
                                       else if (g == 2) DXM_LAUNCH_FEFP(HARD, 2, T); else DXM_LAUNCH_FEFP(HARD, 3, T); } while (0)
      // tl == TL_COEF: the 54 building blocks of the tangent per point instead of its 81 entries (host-buffer form)
      if (tl == TL_COEF) { if (voce) DXM_LAUNCH_FEFP_G(1, 1); else DXM_LAUNCH_FEFP_G(0, 1); }
      else               { if (voce) DXM_LAUNCH_FEFP_G(1, 0); else DXM_LAUNCH_FEFP_G(0, 0); }
#undef DXM_LAUNCH_FEFP_G
#undef DXM_LAUNCH_FEFP
      break;
    }
    default: return fail(-1, "law %d not launchable", m->law);
  }
  HIP_TRY(hipGetLastError());
  *grid_out = grid;
  return 0;
}

static int launch(dxm_material* m, const double* grad, double* flux, double* ct, hipStream_t st,
                  const MeshSource* fused = nullptr) {
  if (m->n == 0) { m->last_grid = 0; return 0; }
  int grid = 0;
  if (int rc = launch_range(m, 0, m->n, grad, flux, ct, st, 0, &grid, fused, m->tangent_layout)) return rc;
  m->last_grid = grid;
  m->launched = true;
  m->s1_alias = false;  // the kernel rewrites every slot of s1
  // completion marker owned by the handle (not recordable while the stream is being captured into a graph)
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
  m->last_event_recorded = false;
  if (cap == hipStreamCaptureStatusNone) {
    HIP_TRY(hipEventRecord(m->last_event, st));
    m->last_event_recorded = true;
  }
  return 0;
}

extern "C" {

int dxm_integrate_device(dxm_material* m, const double* grad_dev, double dt, double* flux_dev,
                         double* ct_dev, void* hip_stream) {
  (void)dt;  // rate-independent laws; QuadratureMap never forwards dt (quadrature_map.py:321)
  if (!m) return fail(-1, "null handle");
  if (m->n > 0 && (!grad_dev || !flux_dev || !ct_dev)) return fail(-1, "null device pointer");
  DEVICE_GUARD(m);
  return launch(m, grad_dev, flux_dev, ct_dev, (hipStream_t)hip_stream);
}

int dxm_get_stats(dxm_material* m, dxm_stats* stats) {
  if (!m) return fail(-1, "null handle");
  dxm_stats s{};
  s.n_points = m->n;
  if (m->launched && m->last_grid > 0) {
    DEVICE_GUARD(m);
    if (int rc = sync_last(m)) return rc;
    if (!m->h_stats) HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&m->h_stats), sizeof(BlockStats) * m->stats_capacity, hipHostMallocDefault));
    HIP_TRY(hipMemcpy(m->h_stats, m->d_stats, sizeof(BlockStats) * m->last_grid, hipMemcpyDeviceToHost));
    for (int k = 0; k < m->last_grid; ++k) {
      const BlockStats& b = m->h_stats[k];
      s.n_plastic += (int64_t)b.n_plastic;
      s.n_not_converged += (int64_t)b.n_not_converged;
      s.n_nan += (int64_t)b.n_nan;
      if ((int32_t)b.max_iters > s.max_local_iters) s.max_local_iters = (int32_t)b.max_iters;
    }
  }
  if (stats) *stats = s;
  if (s.n_not_converged > 0)
    return (int)(s.n_not_converged > 0x7fffffff ? 0x7fffffff : s.n_not_converged);
  return 0;
}

static int pack_isv_range(dxm_material* m, int which, int64_t off, int64_t cnt, double* isv_aos_dev,
                          hipStream_t st) {
  const LawDesc& d = kLaws[m->law];
  const int total = isv_total(d);
  PackMap map{};
  map.n = total;
  int k = 0;
  for (int f = 0; f < d.n_isv_fields; ++f)
    for (int c = 0; c < d.isv_dim[f]; ++c) map.slot[k++] = d.isv_slot[f] + c;
  const int64_t work = cnt * total;
  const int blocks = (int)((work + 255) / 256);
  hipLaunchKernelGGL(pack_isv_kernel, dim3(blocks), dim3(256), 0, st, state_of(m, which) + off, m->ld,
                     cnt, isv_aos_dev, map);
  HIP_TRY(hipGetLastError());
  return 0;
}

// one user-visible field of s1 for the point range [off, off + cnt): AoS (cnt, dim) at dst_dev
static int pack_isv_field_range(dxm_material* m, int field, int64_t off, int64_t cnt, double* dst_dev, hipStream_t st) {
  const LawDesc& d = kLaws[m->law];
  PackMap map{};
  map.n = d.isv_dim[field];
  for (int c = 0; c < map.n; ++c) map.slot[c] = d.isv_slot[field] + c;
  const int blocks = (int)((cnt * map.n + 255) / 256);
  hipLaunchKernelGGL(pack_isv_kernel, dim3(blocks), dim3(256), 0, st, m->state[1] + off, m->ld, cnt, dst_dev, map);
  HIP_TRY(hipGetLastError());
  return 0;
}

int dxm_bind_isv_output(dxm_material* m, int field, double* host_aos) {
  if (!m) return fail(-1, "null handle");
  const LawDesc& d = kLaws[m->law];
  if (field < 0 || field >= d.n_isv_fields) return fail(-1, "state field %d out of range", field);
  if (host_aos && m->n > 0 && !page_locked(host_aos, sizeof(double) * m->n * d.isv_dim[field]))
    return fail(-1, "the array must be page-locked (dxm_host_alloc / dxm_host_register): it is written by DMA inside the chunk pipeline");
  m->isv_out[field] = host_aos;
  return 0;
}

int dxm_isv_device(dxm_material* m, int which, double* isv_aos_dev, void* hip_stream) {
  if (!m) return fail(-1, "null handle");
  if (which != DXM_S0 && which != DXM_S1) return fail(-1, "state selector must be DXM_S0 or DXM_S1");
  const LawDesc& d = kLaws[m->law];
  const int total = isv_total(d);
  if (total == 0 || m->n == 0) return 0;
  if (!isv_aos_dev) return fail(-1, "null device pointer");
  DEVICE_GUARD(m);
  return pack_isv_range(m, which, 0, m->n, isv_aos_dev, (hipStream_t)hip_stream);
}

static int ensure_host_path_buffers(dxm_material* m, bool need_grad = true) {
  const LawDesc& d = kLaws[m->law];
  const int64_t n = m->n;
  const int total = isv_total(d);
  if (need_grad && !m->d_grad) HIP_TRY(hipMalloc(&m->d_grad, sizeof(double) * n * d.n_grad));
  if (!m->d_flux) HIP_TRY(hipMalloc(&m->d_flux, sizeof(double) * n * d.n_flux));
  if (!m->d_ct) HIP_TRY(hipMalloc(&m->d_ct, sizeof(double) * n * d.n_flux * d.n_grad));  // sized for the full layout
  if (total > 0 && !m->d_isv) HIP_TRY(hipMalloc(&m->d_isv, sizeof(double) * n * total));
  return 0;
}

}  // extern "C"

// Host-buffer form, shared by dxm_integrate and dxm_integrate_displacement.
//   upload(off, cnt, stream) enqueues whatever produces m->d_grad[off .. off+cnt) on `stream`.
// Large batches are cut into chunks (multiples of 256 points): the H2D and the kernel of chunk c+1 overlap the D2H of
// chunk c (PCIe is full duplex and the bytes coming back dominate).  Page-locked gradient arrays: up to 24 chunks, uploads +
// kernels on one stream, downloads on two others (option split_streams, see below); staged uploads and the displacement
// forms: up to 64 whole chunks alternating on two streams.  Each chunk is one launch over a point range; its block-stat
// records are appended after the previous chunk's.
//
// What crosses PCIe on the way back, per point: the flux (48 / 72 B), the tangent, and -- only when the
// caller passes a destination -- the internal state variables (they are consumed at advance(), not per
// Newton iteration: the Python layer fetches them on demand).  For the small-strain laws with the full
// (N, 6, 6) tangent requested, the kernel writes the 9 coefficients of the tangent instead of its 36
// entries, 72 instead of 288 B/point are moved into a page-locked landing area and worker threads rebuild
// the block in the caller's array chunk by chunk, behind the transfer of the following chunks
// (bit-identical to the full kernel; the elastic block is a constant and is only filled in).
// host_grad: the caller's gradient array when it is in ordinary (pageable) memory, else nullptr.  It is never handed
// to the runtime: worker threads copy chunk c into slot c % 8 of a page-locked ring and a small copy kernel moves it to
// HBM over PCIe (what the runtime's own pageable path does with its staging buffers; no DMA engine is shared with the
// downloads).
template <class Upload>
static int run_and_download(dxm_material* m, Upload upload, double* flux_aos, double* isv_aos,
                            double* ct_aos, dxm_stats* stats, const MeshSource* fused = nullptr,
                            const double* host_grad = nullptr, const int64_t* rows = nullptr) {
  const auto t_enter = std::chrono::steady_clock::now();
  const LawDesc& d = kLaws[m->law];
  const int64_t n = m->n;
  const int total = isv_total(d);
  // rows (dxm_integrate_rows: J2 laws, full layout, flux and tangent requested): flux_aos / ct_aos are the BASES of larger
  // arrays, point i is their row rows[i].  Always the 32 B/point form; the stress lands in the library's own page-locked
  // area and the worker threads that rebuild the blocks put both where they belong -- the caller's arrays see CPU stores only.
  const bool rowmode = rows != nullptr;
  // ... and so are the bound state fields (dxm_bind_isv_output): in the rows forms the bound pointers are the BASES of the Functions
  // over all cells; the fields land in the library's page-locked area and the worker threads put point i into row rows[i]
  bool isv_rows = false;
  if (rowmode) for (int f = 0; f < d.n_isv_fields; ++f) isv_rows = isv_rows || m->isv_out[f] != nullptr;
  const bool fefp = d.n_grad == 9;
  // a J2 handle with the "sym" layout: (c1, c2, c3, w) cross PCIe like for the full layout (32 instead of 168 B/point) and the
  // workers rebuild the 21 upper-triangle entries from them and the stress (expand_pack4_tangent_sym); needs the stress in
  // page-locked memory like the pack4 form below, else the kernel's own 21 entries are downloaded
  const bool sym_packed = !rowmode && m->opt_packed_transfer >= 2 && m->tangent_layout == DXM_TANGENT_SYM && m->law != DXM_LAW_ELASTIC_ISO && !fefp &&
                          ct_aos != nullptr && flux_aos != nullptr && n >= m->opt_packed_min_points &&
                          (m->opt_pageable_dma || page_locked(flux_aos, sizeof(double) * n * d.n_flux));
  const bool packed = rowmode || sym_packed || (m->opt_packed_transfer && m->tangent_layout == DXM_TANGENT_FULL && ct_aos != nullptr && n >= m->opt_packed_min_points);
  // the rows forms of a handle whose OWN layout is packed (sym / coef / pack4): the kernel writes that layout, it lands in the
  // library's page-locked area like the stress, and the worker threads MOVE point i to row rows[i] -- nothing is rebuilt
  const bool rows_plain = rowmode && m->tangent_layout != DXM_TANGENT_FULL;
  const bool constant = packed && !rows_plain && m->law == DXM_LAW_ELASTIC_ISO;
  // small strain: (c1, c2, c3, w) only -- the direction n is rebuilt from the stress, which the caller receives in
  // page-locked memory as part of the same chunk -- else the nine coefficients
  const bool pack4 = packed && !constant && !fefp && !rows_plain && (rowmode || (m->opt_packed_transfer >= 2 && flux_aos != nullptr &&
                     (m->opt_pageable_dma || page_locked(flux_aos, sizeof(double) * n * d.n_flux))));
  const int tl = rows_plain ? m->tangent_layout : (packed && !constant ? (pack4 ? TL_PACK4 : TL_COEF) : m->tangent_layout);   // layout of this call's launches
  const int np = rows_plain ? tangent_size(m) : (fefp ? FEFP_REC : (pack4 ? 4 : 9));   // doubles per point of what lands in h_coef
  const int nfull = sym_packed ? 21 : d.n_flux * d.n_grad;      // doubles per point of what the workers rebuild in the caller's array
  const int job = sym_packed ? -4 : np;                         // HostPool job code of that rebuild
  const int nt = packed && !constant ? np : tangent_size(m);   // doubles per point in d_ct: the packed form of this call, else the handle's layout
  if (!m->pipe_stream) HIP_TRY(hipStreamCreateWithFlags(&m->pipe_stream, hipStreamNonBlocking));
  if (m->opt_split_streams && !m->down_stream2) HIP_TRY(hipStreamCreateWithFlags(&m->down_stream2, hipStreamNonBlocking));
  if (packed || host_grad) {
    const int land = fefp ? FEFP_REC : (np > 9 ? np : 9);   // (9 covers both packed forms of the J2 laws; a "sym" handle in the rows forms lands 21)
    if (packed && !constant && m->h_coef_per_point < land) {
      if (m->h_coef) HIP_TRY(hipHostFree(m->h_coef));
      m->h_coef = nullptr;
      m->h_coef_per_point = 0;
      HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&m->h_coef), sizeof(double) * n * land, hipHostMallocDefault));
      m->h_coef_per_point = land;
    }
    if (rowmode && !m->h_flux) HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&m->h_flux), sizeof(double) * n * d.n_flux, hipHostMallocDefault));
    if (isv_rows && !m->h_isv) HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&m->h_isv), sizeof(double) * n * total, hipHostMallocDefault));
    if (!m->pool || (int)m->pool->threads.size() != m->opt_host_threads) {
      delete m->pool;
      m->pool = new HostPool(m->opt_host_threads);
    }
    m->elastic_lm[0] = m->prm.lambda;
    m->elastic_lm[1] = m->prm.mu;
  }
  // short chunks (up to 64) whenever little crosses PCIe per point -- the packed forms of this call AND a handle whose own layout
  // is packed (sym / coef / pack4: 168 / 72 / 32 B/point of tangent): with 8 long chunks the first result lands after 5 of the
  // 28 ms of a 1e7-point pack4 call (profiles/r06_packed_update.md); the full 288 B/point block keeps its 8
  const bool short_chunks = packed || (m->tangent_layout != DXM_TANGENT_FULL && ct_aos != nullptr);
  // Three streams instead of two (option split_streams, default on) when every chunk starts with a DMA upload from page-locked
  // memory: uploads + kernels of all chunks on one stream, the downloads of chunk c on one of two others behind the chunk's
  // kernel_done event.  An upload queued on the stream that also carries a chunk's downloads costs the device-to-host
  // direction -- 80 to 136 B/point against 48 up -- 4-8 ms per 1e7 points (raw HIP calls: 27-32 ms against 24-25); split,
  // the 1.36 GB of a pack4 update with its state fields land in 25.4 ms (53 GB/s) instead of 27.4-30.6, the 0.8 GB of the lazy
  // mode in 15.6 ms instead of 20-24, and the times stop moving from call to call.  With too many chunks the same scheme runs at
  // a third of the link rate (1e7 points: from 32 chunks of four downloads, or 64 of two, on; profiles/r06_packed_update.md):
  // the number of chunks is capped below.
  // Staged uploads (a pageable gradient array through the ring) and the fused displacement form (no per-chunk upload at
  // all) keep the two alternating streams.
  const bool split_ok = m->opt_split_streams && m->opt_pipeline && host_grad == nullptr && fused == nullptr;
  // How many chunks the three-stream scheme takes before it turns slower than the two alternating streams grows with the batch
  // (profiles/r06_hostpath_split_chunk_sweep.jsonl, 3e5 ... 1e7 points x 2 ... 24 chunks): 6 at 3e5 points, 8 at 1e6, 12 at
  // 2-3e6, 16 at 5e6, 24 at 1e7 -- one chunk more and the call takes up to 1.7 x as long.  7 sqrt(n / 1e6) stays on the good
  // side at every size measured (4, 7, 9-12, 15, 22), where the scheme beats alternating chunks by 8-25 %.
  int split_cap = (int)(7.0 * std::sqrt((double)n / 1e6));
  split_cap = split_cap < 1 ? 1 : (split_cap > 24 ? 24 : split_cap);
  const dxm_host::ChunkPlan plan = dxm_host::plan_chunks(n, short_chunks, host_grad != nullptr, split_ok && m->opt_max_chunks > split_cap ? split_cap : m->opt_max_chunks, m->opt_pipeline);
  const int nchunks = plan.nchunks;
  for (int c = 0; c < nchunks; ++c)
    if (!m->chunk_done[c]) HIP_TRY(hipEventCreateWithFlags(&m->chunk_done[c], hipEventDisableTiming));
  const bool split = split_ok && nchunks > 1;
  if (split)
    for (int c = 0; c < nchunks; ++c)
      if (!m->kernel_done[c]) HIP_TRY(hipEventCreateWithFlags(&m->kernel_done[c], hipEventDisableTiming));
  const int64_t csize = plan.csize;
  if (host_grad) {
    const int64_t need = csize * d.n_grad;
    if (m->ring_slot_doubles < need) {
      if (m->h_grad_ring) HIP_TRY(hipHostFree(m->h_grad_ring));
      m->h_grad_ring = nullptr;
      HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&m->h_grad_ring), sizeof(double) * need * DXM_RING, hipHostMallocDefault));
      m->ring_slot_doubles = need;
    }
    for (hipEvent_t& e : m->ring_done) if (!e) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  double* field_scratch = m->d_isv;
  if (isv_aos && total > 0) {
    bool delivering = false;
    for (int f = 0; f < d.n_isv_fields; ++f) delivering = delivering || m->isv_out[f] != nullptr;
    if (delivering) {
      if (!m->d_isv_fields) HIP_TRY(hipMalloc(&m->d_isv_fields, sizeof(double) * n * total));
      field_scratch = m->d_isv_fields;
    }
  }
  const bool any = m->opt_pageable_dma;
  const bool flux_locked = rowmode || any || page_locked(flux_aos, sizeof(double) * n * d.n_flux);   // rowmode: into h_flux
  const bool isv_locked = any || page_locked(isv_aos, sizeof(double) * n * total);
  const bool ct_locked = rowmode || any || page_locked(ct_aos, sizeof(double) * n * tangent_size(m));
  int stats_off = 0, issued = 0, submitted = 0;
  hipStream_t streams[2] = {m->own_stream, m->pipe_stream};
  // An early return (a failing HIP call part-way through the chunk loop) leaves kernels, ring copies and downloads of
  // the earlier chunks in flight on both streams, into the caller's arrays: wait for them before returning, so that the
  // caller may free or reuse its arrays, and leave the handle as after a launch whose completion is unknown.
  struct InFlight {
    dxm_material* m;
    bool completed = false;
    ~InFlight() {
      if (completed) return;
      (void)hipStreamSynchronize(m->own_stream);
      if (m->pipe_stream) (void)hipStreamSynchronize(m->pipe_stream);
      if (m->down_stream2) (void)hipStreamSynchronize(m->down_stream2);
      (void)hipGetLastError();
      m->launched = true;
      m->last_event_recorded = false;   // sync_last falls back to the whole device
      m->s1_alias = false;              // parts of s1 have been rewritten
      m->last_grid = 0;
    }
  } inflight{m};
  // no worker may still be writing into the caller's array when this function returns, error paths included
  struct PoolDrain {
    HostPool* p;
    ~PoolDrain() {
      if (!p) return;
      for (int t = 0; t < 64; ++t) p->wait_copy(t);   // staging copies still read the caller's gradient array
      p->wait();
    }
  } drain{(packed || host_grad) ? m->pool : nullptr};
  if (constant && !rowmode) m->pool->submit(m->elastic_lm, ct_aos, n, 0);   // nothing to wait for
  // chunk p of the caller's pageable gradient array -> its ring slot, by the worker threads, asynchronously
  auto stage_chunk = [&](int p) -> int {
    const int64_t o = (int64_t)p * csize;
    if (!host_grad || p >= nchunks || o >= n) return 0;
    if (p >= DXM_RING) HIP_TRY(hipEventSynchronize(m->ring_done[dxm_host::ring_slot(p)]));   // the copy kernel of chunk p - DXM_RING has read this slot
    m->pool->copy_async(host_grad + o * d.n_grad, m->h_grad_ring + (int64_t)dxm_host::ring_slot(p) * m->ring_slot_doubles,
                        sizeof(double) * ((n - o) < csize ? (n - o) : csize) * d.n_grad, p);
    return 0;
  };
  // rows forms: the bound state fields of chunk [o, o + cnt) from the landing area to their rows, on the worker threads
  auto scatter_fields = [&](int64_t o, int64_t cnt) {
    for (int f = 0, before = 0; f < d.n_isv_fields; before += d.isv_dim[f], ++f)
      if (m->isv_out[f]) m->pool->submit_scatter(m->h_isv + n * before + o * d.isv_dim[f], m->isv_out[f], rows + o, cnt, d.isv_dim[f]);
  };
  double ms_wait_copy = 0.0, ms_first_copy = 0.0;   // option verbose: time the issue loop spent waiting for staging copies
  const int ahead = m->opt_stage_ahead;
  for (int p = 0; p < ahead; ++p) if (int rc = stage_chunk(p)) return rc;
  for (int c = 0; c < nchunks; ++c) {
    const int64_t off = (int64_t)c * csize;
    if (off >= n) break;
    const int64_t cnt = (n - off) < csize ? (n - off) : csize;
    hipStream_t st = split ? m->own_stream : streams[c & 1];                         // upload + kernels of this chunk
    hipStream_t sd = split ? ((c & 1) ? m->down_stream2 : m->pipe_stream) : st;      // its downloads (split: two download streams take turns)
    if (int rc = upload(off, cnt, st)) return rc;
    const double* gptr = fused ? m->d_flux : m->d_grad + off * d.n_grad;
    if (host_grad) {
      const int slot = dxm_host::ring_slot(c);
      double* dst = m->h_grad_ring + (int64_t)slot * m->ring_slot_doubles;
      if (int rc = stage_chunk(c + ahead)) return rc;   // keep `ahead` chunks ahead of the launches
      {
        const auto tw = std::chrono::steady_clock::now();
        m->pool->wait_copy(c);
        const double w = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tw).count();
        ms_wait_copy += w;
        if (c == 0) ms_first_copy = w;
      }
      // page-locked host memory is addressable from the device under the same pointer
      const int64_t n16 = cnt * d.n_grad / 2;      // cnt is a multiple of 256 except in the last chunk; n_grad 6 or 9:
      const int64_t tail = cnt * d.n_grad - 2 * n16;   // an odd count leaves one double for a plain copy
      // (on the chunk's own stream: a third stream carrying all uploads ahead of the kernels was measured at 46-62 ms per
      // 1e7 points against 31-33, profiles/archive/r03_hostpath_fresh_array.md)
      hipLaunchKernelGGL(ring_upload_kernel, dim3(32), dim3(256), 0, st, reinterpret_cast<const double2_t*>(dst),
                         reinterpret_cast<double2_t*>(m->d_grad + off * d.n_grad), n16);
      if (tail) HIP_TRY(hipMemcpyAsync(m->d_grad + off * d.n_grad + 2 * n16, dst + 2 * n16, sizeof(double), hipMemcpyHostToDevice, st));
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipEventRecord(m->ring_done[slot], st));   // the slot is free once the copy kernel has read it
    }
    int grid = 0;
    MeshSource src{};
    if (fused) { src = *fused; src.point0 = off; }
    if (int rc = launch_range(m, off, cnt, gptr, m->d_flux + off * d.n_flux,
                              m->d_ct + off * nt, st, stats_off, &grid, fused ? &src : nullptr, tl))
      return rc;
    stats_off += grid;
    // device side of the chunk first (pack kernels of the internal state variables included), then its downloads (split: on
    // one of the two download streams, behind the chunk's kernel_done event)
    if (isv_aos && total > 0) {
      // the kernel wrote state[1]; s1_alias is cleared below, address it directly
      const bool alias = m->s1_alias;
      m->s1_alias = false;
      int rc = pack_isv_range(m, DXM_S1, off, cnt, m->d_isv + off * total, st);
      m->s1_alias = alias;
      if (rc) return rc;
    }
    // fields of the final state bound to host rows (dxm_bind_isv_output: the x.array of the ISV Functions): an (N, total)
    // device scratch holds them field after field, [n * sum of the dims before f] + off * dim_f -- d_isv itself when the call
    // has no isv_aos, its own scratch when d_isv carries the interleaved rows of isv_aos (two layouts cannot share one area:
    // the chunks overlap on two streams and a pageable isv_aos is downloaded from d_isv after the loop)
    for (int f = 0, before = 0; f < d.n_isv_fields; before += d.isv_dim[f], ++f) {
      if (!m->isv_out[f]) continue;
      if (int rc = pack_isv_field_range(m, f, off, cnt, field_scratch + n * before + off * d.isv_dim[f], st)) return rc;
    }
    if (split) {
      HIP_TRY(hipEventRecord(m->kernel_done[c], st));
      HIP_TRY(hipStreamWaitEvent(sd, m->kernel_done[c], 0));
    }
    if (flux_aos && flux_locked)
      HIP_TRY(hipMemcpyAsync((rowmode ? m->h_flux : flux_aos) + off * d.n_flux, m->d_flux + off * d.n_flux,
                             sizeof(double) * cnt * d.n_flux, hipMemcpyDeviceToHost, sd));
    if (ct_aos && !constant && (packed || ct_locked)) {
      double* dst = packed ? m->h_coef + off * np : ct_aos + off * nt;
      HIP_TRY(hipMemcpyAsync(dst, m->d_ct + off * nt, sizeof(double) * cnt * nt, hipMemcpyDeviceToHost, sd));
    }
    if (isv_aos && total > 0 && isv_locked)
      HIP_TRY(hipMemcpyAsync(isv_aos + off * total, m->d_isv + off * total, sizeof(double) * cnt * total, hipMemcpyDeviceToHost, sd));
    for (int f = 0, before = 0; f < d.n_isv_fields; before += d.isv_dim[f], ++f) {
      if (!m->isv_out[f]) continue;
      double* land = isv_rows ? m->h_isv + n * before + off * d.isv_dim[f] : m->isv_out[f] + off * d.isv_dim[f];
      HIP_TRY(hipMemcpyAsync(land, field_scratch + n * before + off * d.isv_dim[f], sizeof(double) * cnt * d.isv_dim[f], hipMemcpyDeviceToHost, sd));
    }
    HIP_TRY(hipEventRecord(m->chunk_done[c], sd));
    issued = c + 1;
    // a pageable upload blocks this thread for its whole duration, so earlier chunks land while the later ones are
    // still being issued: hand them to the workers now, not after the loop
    if (packed && (!constant || rowmode))
      while (submitted < issued && hipEventQuery(m->chunk_done[submitted]) == hipSuccess) {
        const int64_t o = (int64_t)submitted * csize;
        if (rowmode) {
          const int64_t cn = (n - o) < csize ? (n - o) : csize;
          if (rows_plain) {
            m->pool->submit_scatter(m->h_flux + o * d.n_flux, flux_aos, rows + o, cn, d.n_flux);
            m->pool->submit_scatter(m->h_coef + o * np, ct_aos, rows + o, cn, np);
          }
          else m->pool->submit(constant ? m->elastic_lm : m->h_coef + o * np, ct_aos, cn, constant ? 0 : np, m->h_flux + o * d.n_flux, rows + o, flux_aos);
          if (isv_rows) scatter_fields(o, cn);
        }
        else m->pool->submit(m->h_coef + o * np, ct_aos + o * nfull, (n - o) < csize ? (n - o) : csize, job, pack4 ? flux_aos + o * d.n_flux : nullptr);
        ++submitted;
      }
  }
  (void)hipGetLastError();   // hipEventQuery reports "not ready" through the error state
  m->last_grid = stats_off;
  m->launched = true;
  m->s1_alias = false;  // every slot of s1 has been rewritten
  // the chunks complete in issue order on their streams; rebuild each block as soon as it has landed
  const auto t_issued = std::chrono::steady_clock::now();
  auto ms_since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); };
  for (int c = 0; c < issued; ++c) {
    HIP_TRY(hipEventSynchronize(m->chunk_done[c]));
    if (m->opt_verbose && (c % 8 == 7 || c == 0)) fprintf(stderr, "[dxm host path] chunk %d landed at +%.2f ms after issue (issue loop took %.2f ms)\n", c, ms_since(t_issued), std::chrono::duration<double, std::milli>(t_issued - t_enter).count());
    if (packed && (!constant || rowmode) && c >= submitted) {
      const int64_t off = (int64_t)c * csize;
      const int64_t cnt = (n - off) < csize ? (n - off) : csize;
      if (rowmode) {
        if (rows_plain) {
          m->pool->submit_scatter(m->h_flux + off * d.n_flux, flux_aos, rows + off, cnt, d.n_flux);
          m->pool->submit_scatter(m->h_coef + off * np, ct_aos, rows + off, cnt, np);
        }
        else m->pool->submit(constant ? m->elastic_lm : m->h_coef + off * np, ct_aos, cnt, constant ? 0 : np, m->h_flux + off * d.n_flux, rows + off, flux_aos);
        if (isv_rows) scatter_fields(off, cnt);
      }
      else m->pool->submit(m->h_coef + off * np, ct_aos + off * nfull, cnt, job, pack4 ? flux_aos + off * d.n_flux : nullptr);
      submitted = c + 1;
    }
  }
  // destinations in ordinary (pageable) memory were left out above: they are filled through the page-locked staging now
  // (download_to_host; slower, and only a C caller that did not use dxm_host_alloc / dxm_host_register gets here)
  if (flux_aos && !flux_locked)
    if (int rc = download_to_host(flux_aos, m->d_flux, sizeof(double) * n * d.n_flux, m->own_stream)) return rc;
  if (isv_aos && total > 0 && !isv_locked)
    if (int rc = download_to_host(isv_aos, m->d_isv, sizeof(double) * n * total, m->own_stream)) return rc;
  if (ct_aos && !constant && !packed && !ct_locked)
    if (int rc = download_to_host(ct_aos, m->d_ct, sizeof(double) * n * nt, m->own_stream)) return rc;
  HIP_TRY(hipEventRecord(m->last_event, m->own_stream));   // everything of this call is complete already
  m->last_event_recorded = true;
  const auto t_landed = std::chrono::steady_clock::now();
  if (packed) m->pool->wait();
  if (m->opt_verbose && host_grad) fprintf(stderr, "[dxm host path] %d chunks of %lld points; issue loop waited %.2f ms for staging copies (first chunk %.2f ms)\n", nchunks, (long long)csize, ms_wait_copy, ms_first_copy);
  if (m->opt_verbose) fprintf(stderr, "[dxm host path] all landed at +%.2f ms, workers done %.2f ms later\n", std::chrono::duration<double, std::milli>(t_landed - t_issued).count(), ms_since(t_landed));
  inflight.completed = true;
  m->io1_valid = (fused ? 0 : 1) | 2;   // d_grad (unless the strain never existed as an array) and d_flux are those of s1
  const auto t_stats = std::chrono::steady_clock::now();
  const int rc_stats = dxm_get_stats(m, stats);
  if (stats) stats->upload = host_grad ? DXM_UPLOAD_STAGED : m->last_upload;
  if (m->opt_verbose) fprintf(stderr, "[dxm host path] status records summed in %.3f ms; %.2f ms since entry\n", ms_since(t_stats), ms_since(t_enter));
  return rc_stats;
}

extern "C" {

// upload_choice: 0 = this call had no choice to make (page-locked input, small batch, option off), else the way the pageable
// gradient array went up (1 page-locked for the call + DMA, 2 staged by the worker threads); see integrate_host below
static int integrate_host_impl(dxm_material* m, const double* grad_aos, double* flux_aos, double* isv_aos, double* ct_aos,
                               dxm_stats* stats, const int64_t* rows, int* upload_choice) {
  const LawDesc& d = kLaws[m->law];
  const int64_t n = m->n;
  DEVICE_GUARD(m);
  // a page-locked gradient array is uploaded by DMA; so is a pageable one if the caller asked for it (option pageable_dma)
  const auto t_in = std::chrono::steady_clock::now();
  bool locked_in = m->opt_pageable_dma || page_locked(grad_aos, sizeof(double) * n * d.n_grad);
  // An array in ordinary memory is page-locked HERE, for the duration of this call, and uploaded by DMA like a page-locked
  // one: registering a 480 MB array takes ~1 ms (profiles/archive/r03_hostpath_register.md) where staging it through the ring
  // costs the worker threads 5-10 ms of a 24 ms call.  This is not the runtime's implicit path for pageable memory (whose
  // cache of on-the-fly mappings outlives the caller's array: DESIGN.md section 1): the range is unregistered before this
  // function returns, error paths included, while the caller still owns the array.  A refusal (a range that overlaps a
  // registered one, memory that cannot be pinned) falls back to the staging ring.
  struct TempRegistration {
    void* p = nullptr;
    dxm_material* m = nullptr;
    ~TempRegistration() {
      if (!p) return;
      const auto t0 = std::chrono::steady_clock::now();
      forget_locked(p);
      (void)hipHostUnregister(p);
      (void)hipGetLastError();
      m->unregister_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      if (m->opt_verbose) fprintf(stderr, "[dxm host path] releasing the page-lock of the gradient array took %.3f ms\n", m->unregister_ms);
    }
  } temp;
  temp.m = m;
  if (!locked_in && (size_t)n * d.n_grad * sizeof(double) >= ((size_t)1 << 20)) {
    // Which of the two is faster depends on the host (where the caller's pages sit relative to the GPU, what else runs on the
    // machine): the handle measures it (dxm_host::UploadChooser, option register_input).
    const int way = m->up.choose();   // 0: option off, 1: page-lock for the call, 2: stage through the ring
    *upload_choice = way;
    if (way == 1) {
      void* p = const_cast<double*>(grad_aos);
      const size_t bytes = sizeof(double) * n * d.n_grad;
      const auto t0 = std::chrono::steady_clock::now();
      if (hipHostRegister(p, bytes, hipHostRegisterDefault) == hipSuccess) {
        temp.p = p;
        locked_in = true;
        // 0.9 ms per 480 MB on transparent huge pages (what numpy asks for), 7-17 ms on 4 KiB pages, where the whole call
        // then takes 43 instead of 28 ms: such arrays go through the staging ring for the next 20 calls, then one more try
        const double ms_lock = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (m->opt_verbose) fprintf(stderr, "[dxm host path] page-locking the gradient array took %.3f ms (releasing it after the previous call: %.3f ms)\n", ms_lock, m->unregister_ms);
        m->up.registered(ms_lock + m->unregister_ms, bytes);   // what the previous call paid to release its range counts too
      } else {
        (void)hipGetLastError();
        m->up.refused();
        *upload_choice = 2;
      }
    }
  }
  m->last_upload = m->opt_pageable_dma ? DXM_UPLOAD_RUNTIME : (temp.p ? DXM_UPLOAD_REGISTERED : (locked_in ? DXM_UPLOAD_PAGE_LOCKED : DXM_UPLOAD_STAGED));
  const auto t_q = std::chrono::steady_clock::now();
  if (int rc = ensure_host_path_buffers(m)) return rc;
  if (int rc = sync_last(m)) return rc;
  if (m->opt_verbose)
    fprintf(stderr, "[dxm host path] classifying the gradient pointer took %.3f ms, buffers + wait for the previous call %.3f ms\n",
            std::chrono::duration<double, std::milli>(t_q - t_in).count(),
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_q).count());
  const int ng = d.n_grad;
  auto upload = [&](int64_t off, int64_t cnt, hipStream_t st) -> int {
    if (locked_in)
      HIP_TRY(hipMemcpyAsync(m->d_grad + off * ng, grad_aos + off * ng, sizeof(double) * cnt * ng,
                             hipMemcpyHostToDevice, st));
    return 0;
  };
  return run_and_download(m, upload, flux_aos, isv_aos, ct_aos, stats, nullptr, locked_in ? nullptr : grad_aos, rows);
}

static int integrate_host(dxm_material* m, const double* grad_aos, double* flux_aos, double* isv_aos, double* ct_aos,
                          dxm_stats* stats, const int64_t* rows) {
  if (!m) return fail(-1, "null handle");
  if (m->n == 0) {
    if (stats) memset(stats, 0, sizeof(*stats));
    return 0;
  }
  if (!grad_aos) return fail(-1, "null gradient pointer");
  int way = 0;
  const auto t0 = std::chrono::steady_clock::now();
  const int rc = integrate_host_impl(m, grad_aos, flux_aos, isv_aos, ct_aos, stats, rows, &way);
  if (rc < 0) return rc;
  // the adaptive choice between page-locking the caller's gradient array for the call and staging it (see above)
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  const bool changed = m->up.record(way, ms);
  if (m->opt_verbose && way && m->up.opt == 1 && (changed || m->up.calls == 5))
    fprintf(stderr, "[dxm host path] gradient upload: page-locked for the call %.2f ms, staged %.2f ms per call -> %s\n",
            m->up.ms[1], m->up.ms[2], m->up.pref == 1 ? "page-locking" : "staging");
  return rc;
}

int dxm_integrate(dxm_material* m, const double* grad_aos, double dt, double* flux_aos,
                  double* isv_aos, double* ct_aos, dxm_stats* stats) {
  (void)dt;
  return integrate_host(m, grad_aos, flux_aos, isv_aos, ct_aos, stats, nullptr);
}

int dxm_integrate_rows(dxm_material* m, const double* grad_aos, double dt, double* flux_rows, double* ct_rows,
                       const int64_t* rows, dxm_stats* stats) {
  (void)dt;
  if (!m) return fail(-1, "null handle");
  if (m->n > 0 && (!flux_rows || !ct_rows || !rows)) return fail(-1, "dxm_integrate_rows needs the flux array, the tangent array and the row index");
  return integrate_host(m, grad_aos, flux_rows, nullptr, ct_rows, stats, rows);
}

void* dxm_host_alloc(uint64_t bytes) {
  void* p = nullptr;
  if (bytes == 0) bytes = 8;
  hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocDefault);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    fail(-3, "hipHostMalloc(%llu) failed: %s", (unsigned long long)bytes, hipGetErrorString(e));
    return nullptr;
  }
  note_locked(p, bytes);
  return p;
}

int dxm_host_free(void* p) {
  if (p) {
    forget_locked(p);
    HIP_TRY(hipHostFree(p));
  }
  return 0;
}

// ---- gradient evaluation on device ----------------------------------------------------------
struct dxm_mesh {
  int nodes_per_cell = 8;   // 8: trilinear hexahedron, 4: linear tetrahedron, 0: Lagrange simplex (fields below)
  int device = 0;
  int64_t n_nodes = 0, n_cells = 0;
  int64_t u_len = 0;        // doubles in the displacement vector (3 per node; tdim per dof for a Lagrange simplex)
  QuadPoints qp{};
  double* d_coords = nullptr;
  int32_t* d_conn = nullptr;
  double* d_u = nullptr;
  hipEvent_t grad_done = nullptr;
  // Lagrange simplex: geometry through d_conn (tdim + 1 vertices per cell), displacement through its own dofmap
  int tdim = 3, nd = 0;
  int64_t n_dofs = 0;
  int32_t* d_dofmap = nullptr;
  double* d_dphi = nullptr;
};

static dxm_mesh* mesh_create(int npc, const double* coords, int64_t n_nodes, const int32_t* conn,
                             int64_t n_cells, const double* qpoints, int nqp, int device) {
  if (!coords || !conn || n_nodes <= 0 || n_cells <= 0 || nqp <= 0 || nqp > 27 || (npc == 8 && !qpoints)) {
    fail(-1, "invalid mesh arguments");
    return nullptr;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
    fail(-2, "no usable HIP device %d (libdxmat has no CPU fallback)", device);
    return nullptr;
  }
  for (int64_t k = 0; k < n_cells * npc; ++k)
    if (conn[k] < 0 || conn[k] >= n_nodes) { fail(-1, "connectivity entry %lld out of range", (long long)k); return nullptr; }
  dxm_mesh* mesh = new dxm_mesh();
  mesh->nodes_per_cell = npc;
  mesh->device = device;
  mesh->n_nodes = n_nodes;
  mesh->n_cells = n_cells;
  mesh->u_len = 3 * n_nodes;
  mesh->qp.nqp = nqp;
  if (qpoints)
    for (int q = 0; q < nqp; ++q)
      for (int d = 0; d < 3; ++d) mesh->qp.xi[q][d] = qpoints[3 * q + d];
  DeviceGuard guard(device);
  bool ok = guard.ok;
  ok = ok && hipMalloc(&mesh->d_coords, sizeof(double) * 3 * n_nodes) == hipSuccess;
  ok = ok && hipMalloc(&mesh->d_conn, sizeof(int32_t) * npc * n_cells) == hipSuccess;
  ok = ok && hipMalloc(&mesh->d_u, sizeof(double) * 3 * n_nodes) == hipSuccess;
  ok = ok && upload_from_host(mesh->d_coords, coords, sizeof(double) * 3 * n_nodes, nullptr) == 0;
  ok = ok && upload_from_host(mesh->d_conn, conn, sizeof(int32_t) * npc * n_cells, nullptr) == 0;
  if (!ok) {
    fail(-3, "device allocation / upload of the mesh failed");
    dxm_mesh_destroy(mesh);
    return nullptr;
  }
  return mesh;
}

dxm_mesh* dxm_mesh_create_hex8(const double* coords, int64_t n_nodes, const int32_t* conn,
                               int64_t n_cells, const double* qpoints, int nqp, int device) {
  return mesh_create(8, coords, n_nodes, conn, n_cells, qpoints, nqp, device);
}

dxm_mesh* dxm_mesh_create_tet4(const double* coords, int64_t n_nodes, const int32_t* conn,
                               int64_t n_cells, int nqp, int device) {
  return mesh_create(4, coords, n_nodes, conn, n_cells, nullptr, nqp, device);
}

dxm_mesh* dxm_mesh_create_simplex(int tdim, const double* coords, int64_t n_vertices, const int32_t* geom_conn,
                                  int64_t n_cells, const int32_t* dofmap, int nd, int64_t n_dofs,
                                  const double* dphi, int nqp, int device) {
  if ((tdim != 2 && tdim != 3) || !coords || !geom_conn || !dofmap || !dphi || n_vertices <= 0 || n_cells <= 0 ||
      nd < tdim + 1 || nd > 64 || n_dofs <= 0 || nqp <= 0 || nqp > 64) {
    fail(-1, "invalid simplex mesh arguments");
    return nullptr;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
    fail(-2, "no usable HIP device %d (libdxmat has no CPU fallback)", device);
    return nullptr;
  }
  const int nv = tdim + 1;
  for (int64_t k = 0; k < n_cells * nv; ++k)
    if (geom_conn[k] < 0 || geom_conn[k] >= n_vertices) { fail(-1, "geometry connectivity entry %lld out of range", (long long)k); return nullptr; }
  for (int64_t k = 0; k < n_cells * nd; ++k)
    if (dofmap[k] < 0 || dofmap[k] >= n_dofs) { fail(-1, "dofmap entry %lld out of range", (long long)k); return nullptr; }
  // a flat or inverted reference map would put Inf / NaN into every gradient of the cell
  for (int64_t c = 0; c < n_cells; ++c) {
    const double* X0 = coords + 3 * (int64_t)geom_conn[c * nv];
    double A[9];
    for (int d = 0; d < tdim; ++d) {
      const double* Xd = coords + 3 * (int64_t)geom_conn[c * nv + d + 1];
      for (int a = 0; a < tdim; ++a) A[a * 3 + d] = Xd[a] - X0[a];
    }
    const double det = tdim == 2 ? A[0] * A[4] - A[1] * A[3]
                                 : A[0] * (A[4] * A[8] - A[5] * A[7]) + A[1] * (A[5] * A[6] - A[3] * A[8]) + A[2] * (A[3] * A[7] - A[4] * A[6]);
    if (!(det != 0.0) || !std::isfinite(det)) { fail(-1, "cell %lld is degenerate (zero volume)", (long long)c); return nullptr; }
  }
  dxm_mesh* mesh = new dxm_mesh();
  mesh->nodes_per_cell = 0;
  mesh->device = device;
  mesh->n_nodes = n_vertices;
  mesh->n_cells = n_cells;
  mesh->qp.nqp = nqp;
  mesh->tdim = tdim;
  mesh->nd = nd;
  mesh->n_dofs = n_dofs;
  mesh->u_len = (int64_t)tdim * n_dofs;
  DeviceGuard guard(device);
  bool ok = guard.ok;
  ok = ok && hipMalloc(&mesh->d_coords, sizeof(double) * 3 * n_vertices) == hipSuccess;
  ok = ok && hipMalloc(&mesh->d_conn, sizeof(int32_t) * nv * n_cells) == hipSuccess;
  ok = ok && hipMalloc(&mesh->d_dofmap, sizeof(int32_t) * nd * n_cells) == hipSuccess;
  ok = ok && hipMalloc(&mesh->d_dphi, sizeof(double) * nqp * nd * tdim) == hipSuccess;
  ok = ok && hipMalloc(&mesh->d_u, sizeof(double) * mesh->u_len) == hipSuccess;
  ok = ok && upload_from_host(mesh->d_coords, coords, sizeof(double) * 3 * n_vertices, nullptr) == 0;
  ok = ok && upload_from_host(mesh->d_conn, geom_conn, sizeof(int32_t) * nv * n_cells, nullptr) == 0;
  ok = ok && upload_from_host(mesh->d_dofmap, dofmap, sizeof(int32_t) * nd * n_cells, nullptr) == 0;
  ok = ok && upload_from_host(mesh->d_dphi, dphi, sizeof(double) * nqp * nd * tdim, nullptr) == 0;
  if (!ok) {
    fail(-3, "device allocation / upload of the mesh failed");
    dxm_mesh_destroy(mesh);
    return nullptr;
  }
  return mesh;
}

int dxm_mesh_destroy(dxm_mesh* mesh) {
  if (!mesh) return 0;
  DeviceGuard guard(mesh->device);
  if (mesh->d_coords) (void)hipFree(mesh->d_coords);
  if (mesh->d_conn) (void)hipFree(mesh->d_conn);
  if (mesh->d_u) (void)hipFree(mesh->d_u);
  if (mesh->d_dofmap) (void)hipFree(mesh->d_dofmap);
  if (mesh->d_dphi) (void)hipFree(mesh->d_dphi);
  if (mesh->grad_done) (void)hipEventDestroy(mesh->grad_done);
  delete mesh;
  return 0;
}

int64_t dxm_mesh_npoints(const dxm_mesh* mesh) { return mesh ? mesh->n_cells * mesh->qp.nqp : -1; }
int64_t dxm_mesh_displacement_size(const dxm_mesh* mesh) { return mesh ? mesh->u_len : -1; }

static MeshSource mesh_source(const dxm_mesh* mesh, const double* u_dev);

int dxm_mesh_gradient_device(dxm_mesh* mesh, const double* u_dev, int kind, double* grad_dev,
                             void* hip_stream) {
  if (!mesh || !u_dev || !grad_dev) return fail(-1, "null argument");
  if (kind != 0 && kind != 1) return fail(-1, "gradient kind must be 0 (strain) or 1 (F)");
  DEVICE_GUARD(mesh);
  const int64_t npts = mesh->n_cells * mesh->qp.nqp;
  const int blocks = (int)((npts + 255) / 256);
  hipStream_t st = (hipStream_t)hip_stream;
  if (mesh->nodes_per_cell == 0) {
    const MeshSource src = mesh_source(mesh, u_dev);
    if (kind == 0) hipLaunchKernelGGL(simplex_gradient_kernel<0>, dim3(blocks), dim3(256), 0, st, src, grad_dev);
    else hipLaunchKernelGGL(simplex_gradient_kernel<1>, dim3(blocks), dim3(256), 0, st, src, grad_dev);
  } else if (mesh->nodes_per_cell == 4) {
    if (kind == 0)
      hipLaunchKernelGGL(tet4_gradient_kernel<0>, dim3(blocks), dim3(256), 0, st, mesh->d_coords, mesh->d_conn,
                         u_dev, mesh->n_cells, mesh->qp.nqp, grad_dev);
    else
      hipLaunchKernelGGL(tet4_gradient_kernel<1>, dim3(blocks), dim3(256), 0, st, mesh->d_coords, mesh->d_conn,
                         u_dev, mesh->n_cells, mesh->qp.nqp, grad_dev);
  } else if (mesh->qp.nqp >= 4) {   // nodal data staged through LDS once per cell
    if (kind == 0)
      hipLaunchKernelGGL(hex8_gradient_staged_kernel<0>, dim3(blocks), dim3(256), 0, st, mesh->d_coords, mesh->d_conn,
                         u_dev, mesh->n_cells, mesh->qp, grad_dev);
    else
      hipLaunchKernelGGL(hex8_gradient_staged_kernel<1>, dim3(blocks), dim3(256), 0, st, mesh->d_coords, mesh->d_conn,
                         u_dev, mesh->n_cells, mesh->qp, grad_dev);
  } else if (kind == 0) {
    hipLaunchKernelGGL(hex8_gradient_kernel<0>, dim3(blocks), dim3(256), 0, st, mesh->d_coords, mesh->d_conn,
                       u_dev, mesh->n_cells, mesh->qp, grad_dev);
  } else {
    hipLaunchKernelGGL(hex8_gradient_kernel<1>, dim3(blocks), dim3(256), 0, st, mesh->d_coords, mesh->d_conn,
                       u_dev, mesh->n_cells, mesh->qp, grad_dev);
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

// kind of in-kernel gradient evaluation this mesh allows: 1 hex8 x 8 points, 2 tet4, 3 Lagrange simplex, 0 none
static int fused_kind(const dxm_mesh* mesh) {
  if (mesh->nodes_per_cell == 8) return mesh->qp.nqp == 8 ? 1 : 0;
  if (mesh->nodes_per_cell == 0) return 3;
  return mesh->nodes_per_cell == 4 ? 2 : 0;
}
static MeshSource mesh_source(const dxm_mesh* mesh, const double* u_dev) {
  MeshSource src{};
  src.coords = mesh->d_coords; src.conn = mesh->d_conn; src.u = u_dev; src.ncells = mesh->n_cells; src.point0 = 0;
  src.kind = fused_kind(mesh); src.nqp = mesh->qp.nqp;
  src.dofmap = mesh->d_dofmap; src.dphi = mesh->d_dphi; src.nd = mesh->nd; src.tdim = mesh->tdim;
  if (src.kind == 1)
    for (int q = 0; q < 8; ++q)
      for (int a = 0; a < 3; ++a) src.xi[q][a] = mesh->qp.xi[q][a];
  return src;
}
static bool fusable(const dxm_mesh* mesh) { return fused_kind(mesh) != 0; }

static int integrate_displacement_host(dxm_material* m, dxm_mesh* mesh, const double* u_host, double* flux_aos, double* isv_aos,
                                       double* ct_aos, dxm_stats* stats, const int64_t* rows) {
  if (!m || !mesh || !u_host) return fail(-1, "null argument");
  if (mesh->device != m->device) return fail(-1, "mesh and material live on different devices");
  if (dxm_mesh_npoints(mesh) != m->n) return fail(-1, "mesh has %lld Gauss points, material %lld",
                                                   (long long)dxm_mesh_npoints(mesh), (long long)m->n);
  DEVICE_GUARD(m);
  m->last_upload = DXM_UPLOAD_NONE;   // no gradient array crosses PCIe in this form
  const bool fuse = m->opt_fused_gradient && fusable(mesh);   // gradient evaluated inside the update kernel
  if (int rc = ensure_host_path_buffers(m, !fuse)) return rc;
  hipStream_t st = m->own_stream;
  if (int rc = sync_last(m)) return rc;
  if (int rc = upload_from_host(mesh->d_u, u_host, sizeof(double) * mesh->u_len, st)) return rc;
  MeshSource src{};
  if (fuse) {
    src = mesh_source(mesh, mesh->d_u);
  } else {
    const int kind = kLaws[m->law].n_grad == 9 ? 1 : 0;
    if (int rc = dxm_mesh_gradient_device(mesh, mesh->d_u, kind, m->d_grad, st)) return rc;
  }
  // the displacement upload (and the gradient array) are produced on own_stream; chunks on the second stream wait for it
  if (!mesh->grad_done) HIP_TRY(hipEventCreateWithFlags(&mesh->grad_done, hipEventDisableTiming));
  HIP_TRY(hipEventRecord(mesh->grad_done, st));
  auto upload = [&](int64_t, int64_t, hipStream_t s) -> int {
    if (s != st) HIP_TRY(hipStreamWaitEvent(s, mesh->grad_done, 0));
    return 0;
  };
  return run_and_download(m, upload, flux_aos, isv_aos, ct_aos, stats, fuse ? &src : nullptr, nullptr, rows);
}

int dxm_integrate_displacement(dxm_material* m, dxm_mesh* mesh, const double* u_host, double dt,
                               double* flux_aos, double* isv_aos, double* ct_aos, dxm_stats* stats) {
  (void)dt;
  return integrate_displacement_host(m, mesh, u_host, flux_aos, isv_aos, ct_aos, stats, nullptr);
}

int dxm_integrate_displacement_rows(dxm_material* m, dxm_mesh* mesh, const double* u_host, double dt, double* flux_rows,
                                    double* ct_rows, const int64_t* rows, dxm_stats* stats) {
  (void)dt;
  if (!m) return fail(-1, "null argument");
  if (m->n > 0 && (!flux_rows || !ct_rows || !rows)) return fail(-1, "dxm_integrate_displacement_rows needs the flux array, the tangent array and the row index");
  return integrate_displacement_host(m, mesh, u_host, flux_rows, nullptr, ct_rows, stats, rows);
}

int dxm_integrate_displacement_device(dxm_material* m, dxm_mesh* mesh, const double* u_dev, double dt,
                                      double* flux_dev, double* ct_dev, void* hip_stream) {
  (void)dt;
  if (!m || !mesh) return fail(-1, "null argument");
  if (mesh->device != m->device) return fail(-1, "mesh and material live on different devices");
  if (dxm_mesh_npoints(mesh) != m->n) return fail(-1, "mesh has %lld Gauss points, material %lld",
                                                   (long long)dxm_mesh_npoints(mesh), (long long)m->n);
  if (m->n > 0 && (!u_dev || !flux_dev || !ct_dev)) return fail(-1, "null device pointer");
  DEVICE_GUARD(m);
  hipStream_t st = (hipStream_t)hip_stream;
  const LawDesc& d = kLaws[m->law];
  if (m->opt_fused_gradient && fusable(mesh)) {   // one kernel: no gradient array at all
    const MeshSource src = mesh_source(mesh, u_dev);
    return launch(m, flux_dev /* unused, only checked for alignment */, flux_dev, ct_dev, st, &src);
  }
  // two kernels on the caller's stream through the handle's gradient scratch
  if (!m->d_grad) HIP_TRY(hipMalloc(&m->d_grad, sizeof(double) * m->n * d.n_grad));
  if (int rc = dxm_mesh_gradient_device(mesh, u_dev, d.n_grad == 9 ? 1 : 0, m->d_grad, hip_stream)) return rc;
  return launch(m, m->d_grad, flux_dev, ct_dev, st);
}

const double* dxm_state_ptr(const dxm_material* m, int which, int field, int comp) {
  if (check_field(m, which, field)) return nullptr;
  const LawDesc& d = kLaws[m->law];
  if (comp < 0 || comp >= d.isv_dim[field]) { fail(-1, "component out of range"); return nullptr; }
  return state_of(m, which) + (size_t)(d.isv_slot[field] + comp) * m->ld;
}

const char* dxm_kernel_name(const dxm_material* m) { return m ? kLaws[m->law].kernel : ""; }

int dxm_expand_tangent_device(const double* coef_dev, int64_t npoints, double* ct_dev, int device, void* hip_stream) {
  if (npoints < 0) return fail(-1, "negative point count");
  if (npoints == 0) return 0;
  if (!coef_dev || !ct_dev) return fail(-1, "null device pointer");
  if (((uintptr_t)coef_dev | (uintptr_t)ct_dev) & 15) return fail(-1, "coefficient / tangent device arrays must be 16-byte aligned");
  DeviceGuard guard(device);
  if (!guard.ok) return fail(-2, "hipSetDevice(%d) failed", device);
  const int64_t tiles = (npoints + WAVE - 1) / WAVE;
  int64_t blocks = (tiles + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(expand_tangent_kernel, dim3((unsigned)blocks), dim3(BLOCK), 0, (hipStream_t)hip_stream, npoints, coef_dev, ct_dev);
  HIP_TRY(hipGetLastError());
  return 0;
}

int dxm_expand_tangent_pack4_device(const double* flux_dev, const double* pack_dev, int64_t npoints, double* ct_dev, int device,
                                    void* hip_stream) {
  if (npoints < 0) return fail(-1, "negative point count");
  if (npoints == 0) return 0;
  if (!flux_dev || !pack_dev || !ct_dev) return fail(-1, "null device pointer");
  if (((uintptr_t)flux_dev | (uintptr_t)pack_dev | (uintptr_t)ct_dev) & 15) return fail(-1, "stress / pack / tangent device arrays must be 16-byte aligned");
  DeviceGuard guard(device);
  if (!guard.ok) return fail(-2, "hipSetDevice(%d) failed", device);
  const int64_t tiles = (npoints + WAVE - 1) / WAVE;
  int64_t blocks = (tiles + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(expand_pack4_kernel, dim3((unsigned)blocks), dim3(BLOCK), 0, (hipStream_t)hip_stream, npoints, flux_dev, pack_dev, ct_dev);
  HIP_TRY(hipGetLastError());
  return 0;
}

int dxm_notify_replay(dxm_material* m) {
  if (!m) return fail(-1, "null handle");
  m->launched = true;
  m->last_event_recorded = false;   // nothing of the replay is known here: waits fall back to the device
  m->s1_alias = false;              // the replayed kernel rewrote every slot of s1
  return 0;
}

uint64_t dxm_launch_generation(const dxm_material* m) { return m ? (m->epoch << 1) | (uint64_t)m->parity : 0; }

int dxm_set_option(dxm_material* m, const char* name, double value) {
  if (!m || !name) return fail(-1, "null argument");
  const std::string k(name);
  const bool on = value != 0.0;
  if (k == "pipeline") m->opt_pipeline = on;
  else if (k == "split_streams") m->opt_split_streams = on;
  else if (k == "packed_transfer") {
    if (!(value == 0.0 || value == 1.0 || value == 2.0)) return fail(-1, "packed_transfer must be 0, 1 or 2");
    m->opt_packed_transfer = (int)value;
  }
  else if (k == "fused_gradient") m->opt_fused_gradient = on;
  else if (k == "verbose") m->opt_verbose = on;
  else if (k == "stage_ahead") {
    if (!(value >= 1 && value <= DXM_RING - 2)) return fail(-1, "stage_ahead must be in [1, %d]", DXM_RING - 2);
    m->opt_stage_ahead = (int)value;
  }
  else if (k == "register_input") {
    if (!(value == 0.0 || value == 1.0 || value == 2.0)) return fail(-1, "register_input must be 0, 1 or 2");
    m->up.set_option((int)value);
  }
  else if (k == "keep_initial_io") m->opt_keep_initial_io = on;
  else if (k == "pageable_dma") m->opt_pageable_dma = value != 0.0;
  else if (k == "query_foreign_pointers") g_query_foreign.store(on ? 1 : 0);   // process-wide
  else if (k == "packed_min_points") {
    if (!(value >= 0 && value <= 1e12)) return fail(-1, "packed_min_points must be >= 0");
    m->opt_packed_min_points = (int64_t)value;
  }
  else if (k == "host_threads") {
    if (!(value >= 1 && value <= 256)) return fail(-1, "host_threads must be in [1, 256]");
    m->opt_host_threads = (int)value;
  } else if (k == "max_chunks") {
    if (!(value >= 1 && value <= DXM_MAX_CHUNKS)) return fail(-1, "max_chunks must be in [1, %d]", DXM_MAX_CHUNKS);
    m->opt_max_chunks = (int)value;
  } else if (k == "blocks_per_cu") {
    if (!(value >= 1 && value <= 256)) return fail(-1, "blocks_per_cu must be in [1, 256]");
    m->blocks_per_cu = (int)value;
  } else {
    return fail(-1, "unknown option '%s'", name);
  }
  ++m->epoch;
  return 0;
}

int dxm_isv_host(dxm_material* m, int which, double* isv_aos) {
  if (!m) return fail(-1, "null handle");
  if (which != DXM_S0 && which != DXM_S1) return fail(-1, "state selector must be DXM_S0 or DXM_S1");
  const LawDesc& d = kLaws[m->law];
  const int total = isv_total(d);
  if (total == 0 || m->n == 0) return 0;
  if (!isv_aos) return fail(-1, "null host pointer");
  DEVICE_GUARD(m);
  if (int rc = sync_last(m)) return rc;
  if (!m->d_isv) HIP_TRY(hipMalloc(&m->d_isv, sizeof(double) * m->n * total));
  if (int rc = pack_isv_range(m, which, 0, m->n, m->d_isv, m->own_stream)) return rc;
  return download_to_host(isv_aos, m->d_isv, sizeof(double) * m->n * total, m->own_stream);
}

int dxm_host_copy(void* dst, const void* src, uint64_t bytes, int threads) {
  if (bytes == 0) return 0;
  if (!dst || !src) return fail(-1, "null host pointer");
  dxm_host::host_copy(dst, src, bytes, threads);
  return 0;
}

// rows of `width` doubles moved through an index on several threads (utils.py:136-143 `array[index] = values`)
static int move_rows(bool scatter, double* dst, const double* src, const int64_t* rows, int64_t n, int width, int threads) {
  if (n <= 0 || width <= 0) return 0;
  if (!dst || !src || !rows) return fail(-1, "null host pointer");
  dxm_host::move_rows(scatter, dst, src, rows, n, width, threads);
  return 0;
}

int dxm_host_scatter_rows(double* dst, const double* src, const int64_t* rows, int64_t n, int width, int threads) {
  return move_rows(true, dst, src, rows, n, width, threads);
}

int dxm_host_gather_rows(double* dst, const double* src, const int64_t* rows, int64_t n, int width, int threads) {
  return move_rows(false, dst, src, rows, n, width, threads);
}

int dxm_host_index_range(const int64_t* rows, int64_t n, int threads, int64_t* lo, int64_t* hi) {
  if (!lo || !hi) return fail(-1, "null result pointer");
  if (n > 0 && !rows) return fail(-1, "null index");
  dxm_host::index_min_max(rows, n, threads, lo, hi);
  return 0;
}

int dxm_host_register(void* p, uint64_t bytes) {
  if (!p || bytes == 0) return fail(-1, "null / empty host range");
  hipError_t e = hipHostRegister(p, bytes, hipHostRegisterDefault);
  if (e != hipSuccess) {
    (void)hipGetLastError();   // (a range that is registered already, memory that cannot be pinned: the caller falls back)
    return fail(-3, "hipHostRegister(%p, %llu) failed: %s", p, (unsigned long long)bytes, hipGetErrorString(e));
  }
  note_locked(p, bytes);
  return 0;
}

int dxm_host_unregister(void* p) {
  if (p) {
    forget_locked(p);
    HIP_TRY(hipHostUnregister(p));
  }
  return 0;
}

}  // extern "C"
