// engine.hip — host side of libbpp_amd.so: device-resident loci, resident batched
// plans, and the C ABI declared in include/bpp_amd.h.  MI355X (gfx950) only.
//
// Reference boundary: the locus API of bpp v4.8.7 (locus.c:622-2631); see the
// header for the function-by-function mapping.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <chrono>
#include <algorithm>
#include <memory>
#include <mutex>
#include <atomic>

#include "bpp_amd.h"
#include "device_types.hpp"
#include "kernels.hpp"

// ------------------------------------------------------------------ errors --
static thread_local std::string g_err;
static int fail(const std::string & m) { g_err = m; return 0; }
#define HIPCHK(call)                                                              \
  do { hipError_t e_ = (call); if (e_ != hipSuccess) {                            \
    g_err = std::string(#call) + ": " + hipGetErrorString(e_); return 0; } } while (0)
#define HIPCHK_PTR(call)                                                          \
  do { hipError_t e_ = (call); if (e_ != hipSuccess) {                            \
    g_err = std::string(#call) + ": " + hipGetErrorString(e_); return nullptr; } } while (0)

// ------------------------------------------------------------------- arena --
// Bump allocator over large HBM chunks: loci created one after another are
// contiguous in memory; pointers never move, so LocusDev records stay valid.
struct Arena
{
  struct Chunk { char * base; size_t size, used; };
  std::vector<Chunk> chunks;
  size_t chunk_bytes = (size_t)256 << 20;
  size_t total = 0;

  void * alloc(size_t bytes, size_t align = 256)
  {
    if (!bytes) bytes = align;
    if (!chunks.empty())
    {
      Chunk & c = chunks.back();
      size_t off = (c.used + align - 1)/align*align;
      if (off + bytes <= c.size) { c.used = off + bytes; return c.base + off; }
    }
    size_t sz = std::max(chunk_bytes, bytes);
    char * p = nullptr;
    if (hipMalloc((void **)&p, sz) != hipSuccess) return nullptr;
    // buffers start zeroed like the reference's (locus.c:753-783)
    if (hipMemset(p, 0, sz) != hipSuccess) { (void)hipFree(p); return nullptr; }
    chunks.push_back({p, sz, bytes});
    total += sz;
    return p;
  }
  void release() { for (auto & c : chunks) (void)hipFree(c.base); chunks.clear(); total = 0; }
};

template <typename T> struct DevBuf
{
  T * p = nullptr; size_t cap = 0;
  bool reserve(size_t n)
  {
    if (n <= cap) return true;
    if (p) (void)hipFree(p);
    p = nullptr; cap = 0;
    size_t want = std::max<size_t>(n, 16);
    if (hipMalloc((void **)&p, want*sizeof(T)) != hipSuccess) return false;
    cap = want;
    return true;
  }
  void free() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf & operator=(const DevBuf &) = delete;
  ~DevBuf() { free(); }                 // (owners that must free with their device current do so explicitly first)
};

// ------------------------------------------------------------------ objects --
struct bpa_plan;

struct bpa_locus
{
  bpa_engine * eng = nullptr;
  uint32_t id = 0;
  unsigned dtype, model, tips, clv_buffers, states, sites, rate_matrices, prob_matrices,
           rate_cats, scale_buffers, attributes;
  LocusDev dev{};
  std::vector<double>   par;          // host copy of the parameter block (source of truth)
  std::vector<uint8_t>  tipcodes;     // host staging of tip codes
  std::vector<uint32_t> weights;
  std::vector<int>      eigen_valid;  // locus->eigen_decomp_valid (locus.c:735)
  bool par_dirty = true, tips_dirty = true, weights_dirty = true, queued = false, alive = true;
  bool host_par_stale = false;        // bpa_plan_set_params_device moved the device block ahead of `par`
  size_t code_bytes() const { return states == 4 ? 1 : 4; }
  bool needs_eigen() const { return !(dtype == BPA_DATA_DNA && model < BPA_DNA_MODEL_GTR); }   // locus.c:2426-2454
  std::unique_ptr<bpa_plan> scratch;  // single-locus calls reuse one small plan
  // the single-locus update API is lazy: locus_update_matrices / locus_update_partials (locus.c:2417, 2530) only
  // queue their work here; it runs — as ONE launch together with the root term — when locus_root_loglikelihood
  // (locus.c:2573) asks for the value, or before anything else touches the locus (flush)
  std::vector<unsigned> pend_pm;
  std::vector<double>   pend_len;
  std::vector<bpa_op_t> pend_ops;
  bool pending = false;
};

struct TimingSlot { hipEvent_t ev[4]; int ev_used = 4; unsigned steps = 1; double bytes = 0, bytes_codes = 0; };   // ev_used 2: only ev[1],ev[2] (kernel-attached); steps / bytes: proposal steps and algorithmic bytes the launch covers

struct bpa_engine
{
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  Arena arena;
  std::vector<bpa_locus *> loci;
  std::vector<void *> staged;           // bpa_engine_stage allocations (freed with the engine)
  std::vector<bpa_locus *> dirty;     // loci whose host-side state must be flushed
  std::vector<bpa_locus *> pending;   // loci with queued single-locus updates (bpa_locus::pend_*)
  DevBuf<LocusDev> d_loci;
  bool table_dirty = true;
  // pinned staging for the per-locus results a host-side accept/reject reads back after every step: a copy into
  // pageable memory goes through the runtime's own staging and costs several times the transfer
  void * h_stage = nullptr; size_t h_stage_bytes = 0;
  // bpa_batch_evaluate on the engine's packing: one pinned image of the step's compact records, one upload, persistent
  // device buffers (no plan object, no allocation in steady state)
  void * h_step = nullptr; size_t h_step_bytes = 0;
  // the batch between bpa_batch_begin and bpa_batch_end
  unsigned bc_T = 0, bc_units = 0, bc_nmat = 0, bc_npat = 0, bc_rmax = 1; bool bc_alljc = false;
  size_t bc_o_mat2 = 0, bc_o_len = 0, bc_o_bm = 0, bc_total = 0;
  std::vector<uint32_t> bc_pat;
  unsigned bc_wait_T = 0; bool bc_wait_zero = false;      // bpa_batch_end_async's batch, collected by bpa_batch_wait
  bool bc_fast = false;                 // sized from the packing's own bounds: no pass over the batch in begin, its checks are fill's
  std::atomic<int> bc_fallback{0};      // fill found a locus the one-image path does not take
  std::vector<uint32_t> pack_slot_pat;  // first pattern of every slot in a term array over ALL packed loci
  unsigned pack_npat = 0, pack_maxops = 0, pack_rmax = 1; int pack_homog = 0;      // 1 every packed locus JC69 / one category, 2 every one multi-category without scalers
  std::atomic<int> bc_failed{0}; std::mutex bc_mtx; std::string bc_msg;
  DevBuf<unsigned char> d_step;
  DevBuf<unsigned char> d_upload;         // flush_state's staged set-up uploads (records | payload)
  DevBuf<double> d_step_terms, d_step_lnl;
  // engine-level packing of the JC69 / one-category loci for step_jc69_v2_kernel (device_types.hpp): shared by all plans
  bool pack_dirty = true;               // a locus appeared / went away / changed its tip states or weights
  unsigned pack_epoch = 0;              // bumped when the slot numbering or a slot's shape changes: older plans fall back
  std::vector<int32_t> slot_of;         // locus id -> slot (-1: not packed)
  std::vector<uint32_t> pack_shape;     // (locus id, np, tips) per slot, to detect a change of the numbering
  DevBuf<LaneStatic> d_lane_tab;
  DevBuf<SlotStatic> d_slot_tab;
  DevBuf<uint32_t>   d_blk_slot_off;
  std::vector<uint32_t> h_blk_slot_off;
  unsigned pack_blocks = 0, pack_slots = 0;
  DevBuf<uint32_t> d_eigen_list;
  int usedata = 1;
  double bfbeta = 1.0;
  // timing
  bool timing = false;
  unsigned timing_stride = 1, timing_phase = 0;   // events on every stride-th launch only
  std::vector<TimingSlot> slots;
  size_t slots_used = 0;
  double acc_ms[3] = {0, 0, 0};
  unsigned long acc_launches = 0, acc_steps = 0;      // launches that carried events, proposal steps they covered
  double acc_bytes = 0;                               // algorithmic bytes of the kernels the events bracketed
  double acc_bytes_codes = 0;                         // ... priced as the kernels hold the data: tip children as codes, a forwarded parent not re-read
  // calls for different loci may come from different host threads (threads.c:87-200 shards loci
  // over pthreads): engine-wide state (dirty list, locus table, stream order, timing) is serialised
  std::recursive_mutex mtx;
};

struct bpa_plan
{
  bpa_engine * eng = nullptr;
  PlanDev pd{};
  unsigned states = 0, rmax = 1;
  bool has_mats = false, has_lnl = true;
  DevBuf<uint32_t> task_locus, task_pat_off, thr_task, mat_off, mat_task, mat_pmatrix, op_off, root_clv;
  DevBuf<uint32_t> blk_task_off, lane_task, task_lane0, lane_rec, task_rec;
  DevBuf<uint4>    recs;
  DevBuf<uint32_t> tile_task, tile_n0;
  DevBuf<unsigned long long> dbg;
  unsigned ntiles = 0, tile = 128;    // tiled 20-state path
  bool s20_tiledk = true;               // a workgroup = 64 patterns x R categories (pipe, mfmak)
  std::string s20_kernel;
  bool fused_klane = false;
  DevBuf<MatRec>   mat_recs;
  bool fused_jc69 = false;            // JC69, one rate category: the latency-optimised kernel
  bool jc69_v2 = false;               // ... on the compact records over the engine's packing (valid while pack_epoch is)
  bool klane_v2 = false;              // the multi-category kernel on them
  unsigned pack_epoch = 0;
  DevBuf<uint4>    recs2;
  DevBuf<MatRec2>  mat2;
  DevBuf<uint32_t> blk_mat_off;
  unsigned fused_rt = 0;              // compile-time rate-category count of the fused kernel (0 = runtime)
  unsigned fused_bs = 0;              // workgroup size of the fused single-launch path (0 = not available)
  DevBuf<int32_t>  root_scaler;
  DevBuf<double>   mat_length, site_term, lnl, lnl_sum, param_stage;
  double * sum_out = nullptr;         // where the per-launch sum of lnl[] goes (own buffer or caller's)
  unsigned sum_parts = 0;             // > 0: sum_out receives that many partial sums (bpa_plan_enable_partial_sums)
  DevBuf<OpDev>    ops;
  std::vector<uint32_t> h_locus;      // host copy of task -> locus id
  double bytes_partials = 0, bytes_pmatrix = 0, flops_partials = 0;
  double bytes_codes = 0;      // K1 + K2 bytes with a tip child priced as its codes (1 B DNA / 4 B AA per pattern) and no re-read of a parent the next update consumes
  unsigned long node_updates = 0, pattern_updates = 0;
  void free_all()
  {
    task_locus.free(); task_pat_off.free(); thr_task.free(); mat_off.free(); mat_task.free();
    mat_pmatrix.free(); op_off.free(); root_clv.free(); root_scaler.free(); mat_length.free();
    site_term.free(); lnl.free(); ops.free(); lnl_sum.free(); param_stage.free();
    blk_task_off.free(); lane_task.free(); task_lane0.free(); lane_rec.free(); task_rec.free(); recs.free(); mat_recs.free(); tile_task.free(); tile_n0.free();
    recs2.free(); mat2.free(); blk_mat_off.free();
  }
  ~bpa_plan() { free_all(); }
};

// hipSetDevice on every launch costs ~1 us; hipGetDevice is a thread-local read.  (No cache of our own: the caller —
// another engine on this thread, torch, user code — may have changed the current device since our last call.)
static int set_device(bpa_engine * e)
{
  int cur = -1;
  if (hipGetDevice(&cur) == hipSuccess && cur == e->device) return 1;
  HIPCHK(hipSetDevice(e->device));
  return 1;
}

// ------------------------------------------------------------------ engine --
extern "C" const char * bpa_version(void) { return "bpp_amd 0.1 (gfx950)"; }
extern "C" const char * bpa_last_error(void) { return g_err.c_str(); }
// 1: built with -DBPA_EXPERIMENTAL (csrc/experimental/ compiled, the A/B switches of superseded variants read from the environment)
extern "C" int bpa_experimental_build(void)
{
#ifdef BPA_EXPERIMENTAL
  return 1;
#else
  return 0;
#endif
}
extern "C" void bpa_internal_set_error(const char * msg) { g_err = msg ? msg : ""; }   // host_input.cpp

// HIP maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (4 unless set) and streams that share one run one after the
// other: a process with two engines alive put the third part-batch stream of a generic sampler on a busy queue (config 3 188 instead
// of 235 it/s, NOTES.md 12).  The runtime reads the variable when it starts — on the process's first HIP call —, so the library's
// entry points that can be that call set it to 8 first, unless the user has set it (Python hosts: bpp_amd/__init__.py does the same
// before the library loads; a host that has used HIP already sets it itself).
static void hw_queues_default()
{
  static std::once_flag once;
  std::call_once(once, [] { (void)setenv("GPU_MAX_HW_QUEUES", "8", 0); });
}

extern "C" int bpa_device_count(void)
{
  hw_queues_default();
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

extern "C" bpa_engine_t * bpa_engine_create(int device, void * stream)
{
  int n = bpa_device_count();
  if (n <= 0) { fail("bpa_engine_create: no HIP device visible (this library has no CPU fallback)"); return nullptr; }
  if (device < 0 || device >= n) { fail("bpa_engine_create: bad device index"); return nullptr; }
  bpa_engine * e = new bpa_engine();
  e->device = device;
  if (hipSetDevice(device) != hipSuccess) { fail("hipSetDevice failed"); delete e; return nullptr; }
  if (stream) e->stream = (hipStream_t)stream;
  else
  {
    if (hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess)
    { fail("hipStreamCreate failed"); delete e; return nullptr; }
    e->own_stream = true;
  }
  return e;
}

extern "C" void bpa_engine_destroy(bpa_engine_t * e)
{
  if (!e) return;
  (void)hipSetDevice(e->device);
  (void)hipStreamSynchronize(e->stream);
  for (auto * l : e->loci) delete l;
  e->arena.release();
  e->d_loci.free(); e->d_eigen_list.free();
  e->d_lane_tab.free(); e->d_slot_tab.free(); e->d_blk_slot_off.free();
  if (e->h_stage) (void)hipHostFree(e->h_stage);
  if (e->h_step) (void)hipHostFree(e->h_step);
  e->d_step.free(); e->d_step_terms.free(); e->d_step_lnl.free(); e->d_upload.free();
  for (void * q : e->staged) (void)hipFree(q);
  for (auto & s : e->slots) for (auto & ev : s.ev) (void)hipEventDestroy(ev);
  if (e->own_stream) (void)hipStreamDestroy(e->stream);
  delete e;
}

extern "C" void * bpa_engine_stage(bpa_engine_t * e, const void * host, size_t bytes)
{
  std::lock_guard<std::recursive_mutex> lock_(e->mtx);
  if (!set_device(e) || !host || !bytes) { fail("bpa_engine_stage: bad arguments"); return nullptr; }
  void * q = nullptr;
  HIPCHK_PTR(hipMalloc(&q, bytes));
  if (hipMemcpy(q, host, bytes, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(q); fail("bpa_engine_stage: copy failed"); return nullptr; }
  e->staged.push_back(q);
  return q;
}

extern "C" int bpa_engine_synchronize(bpa_engine_t * e)
{
  std::lock_guard<std::recursive_mutex> lock_(e->mtx);
  if (!set_device(e)) return 0;
  HIPCHK(hipStreamSynchronize(e->stream));
  return 1;
}

extern "C" void bpa_engine_set_options(bpa_engine_t * e, int usedata, double bfbeta)
{
  std::lock_guard<std::recursive_mutex> lock_(e->mtx);
  e->usedata = usedata; e->bfbeta = bfbeta;
}

static void mark_dirty(bpa_locus * l)
{
  if (!l->queued) { l->queued = true; l->eng->dirty.push_back(l); }
}

// ------------------------------------------------------------------- locus --
extern "C" bpa_locus_t * bpa_locus_create(bpa_engine_t * e, unsigned dtype, unsigned model,
                                          unsigned tips, unsigned clv_buffers, unsigned states,
                                          unsigned sites, unsigned rate_matrices,
                                          unsigned prob_matrices, unsigned rate_cats,
                                          unsigned scale_buffers, unsigned attributes)
{
  if (!e) { fail("bpa_locus_create: null engine"); return nullptr; }
  std::lock_guard<std::recursive_mutex> lock_(e->mtx);
  if (!((dtype == BPA_DATA_DNA && states == 4) || (dtype == BPA_DATA_AA && states == 20)))
  { fail("bpa_locus_create: only DNA (4 states) and amino-acid (20 states) data"); return nullptr; }
  if (dtype == BPA_DATA_DNA && model > BPA_DNA_MODEL_GTR)
  { fail("bpa_locus_create: unknown DNA model"); return nullptr; }
  if (dtype == BPA_DATA_AA && (model < BPA_AA_MODEL_MIN || model > BPA_AA_MODEL_MAX))
  { fail("bpa_locus_create: unknown amino-acid model"); return nullptr; }
  if (!tips || !sites || !rate_cats || !rate_matrices || !prob_matrices)
  { fail("bpa_locus_create: zero-sized locus"); return nullptr; }
  if (hipSetDevice(e->device) != hipSuccess) { fail("hipSetDevice failed"); return nullptr; }

  bpa_locus * l = new bpa_locus();
  l->eng = e; l->dtype = dtype; l->model = model; l->tips = tips; l->clv_buffers = clv_buffers;
  l->states = states; l->sites = sites; l->rate_matrices = rate_matrices;
  l->prob_matrices = prob_matrices; l->rate_cats = rate_cats; l->scale_buffers = scale_buffers;
  l->attributes = attributes;

  const size_t S = states, R = rate_cats, Np = sites;
  LocusDev & d = l->dev;
  const bool jc69 = dtype == BPA_DATA_DNA && model == BPA_DNA_MODEL_JC69;
  d.pstride = jc69 ? 2u : (uint32_t)(S*S);
  // one contiguous hot block per locus: parameters | pattern weights | tip codes (| JC69 (a,b) table),
  // so that a proposal step touches 1-3 cache lines for them instead of 4-8
  auto up16 = [](size_t x) { return (x + 15)/16*16; };
  const size_t par_bytes = up16(par_size(R, S, rate_matrices)*sizeof(double));
  const size_t w_bytes = up16(Np*sizeof(uint32_t)), tip_bytes = up16((size_t)tips*Np*l->code_bytes());
  const size_t pm_bytes = (size_t)prob_matrices*R*d.pstride*sizeof(double);
  char * hot = (char *)e->arena.alloc(par_bytes + w_bytes + tip_bytes + (jc69 ? pm_bytes : 0), 128);
  // 20-state CLVs are state-major planes: pad the plane stride to whole 128-byte lines (BPA_NO_PLANE_PAD=1: the first layout)
  static const bool no_pad = BPA_EXP_SWITCH("BPA_NO_PLANE_PAD") && atoi(BPA_EXP_SWITCH("BPA_NO_PLANE_PAD"));
  const size_t Ld = (S == 20 && !no_pad) ? up16(Np) : Np;
  d.ld = (uint32_t)Ld;
  d.clv    = (double *)e->arena.alloc(std::max<size_t>(clv_buffers, 1)*R*Ld*S*sizeof(double), 128);
  d.scaler = scale_buffers ? (uint32_t *)e->arena.alloc((size_t)scale_buffers*Np*sizeof(uint32_t), 128) : nullptr;
  if (!hot || !d.clv || (scale_buffers && !d.scaler))
  { fail("bpa_locus_create: out of device memory"); delete l; return nullptr; }
  d.par = (double *)hot;
  d.weights = (uint32_t *)(hot + par_bytes);
  d.tips = (uint8_t *)(hot + par_bytes + w_bytes);
  d.pmat = jc69 ? (double *)(hot + par_bytes + w_bytes + tip_bytes) : (double *)e->arena.alloc(pm_bytes, 128);
  if (!d.pmat) { fail("bpa_locus_create: out of device memory"); delete l; return nullptr; }
  d.dip_count = d.dip_map = d.dip_weights = nullptr;
  d.np = sites; d.tips_n = tips; d.rate_cats = rate_cats; d.states = states;
  // 0 JC69, 1..6 closed-form 4x4 models (BPA_DNA_MODEL_*), 100 = eigendecomposition (GTR, amino acids)
  d.model = (dtype == BPA_DATA_DNA && model < BPA_DNA_MODEL_GTR) ? model : 100u;
  d.dtype = dtype; d.rate_matrices = rate_matrices; d.unphased_length = 0;

  // defaults of locus_create (locus.c:727-848): param_indices 0, rates 1 (the
  // reference fills discrete-gamma means of alpha/beta defaults; callers overwrite),
  // rate_weights 1/R, pattern weights 1, frequencies 0 -> set here to 1/S so that an
  // un-parameterised JC69 locus is usable (locus_set_frequencies_and_rates, locus.c:901)
  l->par.assign(par_size(R, S, rate_matrices), 0.0);
  for (size_t k = 0; k < R; ++k)
  {
    l->par[par_rates(R) + k] = 1.0;
    l->par[par_rate_weights(R) + k] = 1.0/(double)R;
    l->par[par_param_idx(R) + k] = 0.0;
  }
  for (size_t m = 0; m < rate_matrices; ++m)
  {
    double * pm = l->par.data() + par_matrix(R, S, m);
    for (size_t s = 0; s < S; ++s) pm[pm_freqs(S) + s] = 1.0/(double)S;
    for (size_t s = 0; s < S*(S-1)/2; ++s) pm[pm_subst(S) + s] = 1.0;
  }
  l->eigen_valid.assign(rate_matrices, 0);
  l->tipcodes.assign((size_t)tips*Np*l->code_bytes(), 0);
  l->weights.assign(Np, 1u);

  l->id = (uint32_t)e->loci.size();
  e->loci.push_back(l);
  e->table_dirty = true;
  mark_dirty(l);
  return l;
}

extern "C" void bpa_locus_destroy(bpa_locus_t * l)
{
  // arena memory is released with the engine (a bump arena: BPP creates its loci once, method.c:4137, and destroys
  // them at exit, method.c:6506); the slot stays so ids remain stable
  if (!l) return;
  std::lock_guard<std::recursive_mutex> lock_(l->eng->mtx);
  (void)set_device(l->eng);
  (void)hipStreamSynchronize(l->eng->stream);        // the scratch plan's buffers may be in use
  l->alive = false; l->pending = false; l->pend_pm.clear(); l->pend_len.clear(); l->pend_ops.clear();
  l->scratch.reset(); l->eng->pack_dirty = true;
}

extern "C" int bpa_set_tip_states(bpa_locus_t * l, unsigned tip_index, const unsigned * map,
                                  const char * sequence)
{
  std::lock_guard<std::recursive_mutex> lock_(l->eng->mtx);
  if (tip_index >= l->tips) return fail("bpa_set_tip_states: tip index out of range");
  const size_t Np = l->sites;
  for (size_t n = 0; n < Np; ++n)
  {
    const unsigned c = map[(unsigned char)sequence[n]];
    if (!c) return fail(std::string("Illegal state code in tip \"") + sequence[n] + "\"");
    if (l->states == 4) l->tipcodes[tip_index*Np + n] = (uint8_t)c;
    else reinterpret_cast<uint32_t *>(l->tipcodes.data())[tip_index*Np + n] = c;
  }
  l->tips_dirty = true; mark_dirty(l);
  return 1;
}

extern "C" void bpa_set_pattern_weights(bpa_locus_t * l, const unsigned * w)
{
  std::lock_guard<std::recursive_mutex> lock_(l->eng->mtx);
  std::copy(w, w + l->sites, l->weights.begin());
  l->weights_dirty = true; mark_dirty(l);
}

// The single-locus update API is lazy (bpa_locus_update_matrices / _partials only queue): what is queued on a locus was
// asked for with the parameters it had THEN, so a setter lets the queue run before it changes them (the eager
// semantics the header documents)
static int run_queued(bpa_locus * l, bool want_lnl, unsigned root_clv, int root_scaler, double * lnl, bool persite);
static void settle_queue(bpa_locus * l)
{
  if (l->pending && l->alive) (void)run_queued(l, false, 0, BPA_SCALE_BUFFER_NONE, nullptr, false);
}

// a per-locus setter after bpa_plan_set_params_device: bring the host mirror level with the device block first
static void refresh_host_par(bpa_locus * l, bool settle = true)
{
  if (settle) settle_queue(l);
  if (!l->host_par_stale) return;
  bpa_engine * e = l->eng;
  (void)hipSetDevice(e->device);
  (void)hipStreamSynchronize(e->stream);
  const unsigned S = l->states, R = l->rate_cats;
  (void)hipMemcpy(l->par.data(), l->dev.par, 3*R*sizeof(double), hipMemcpyDeviceToHost);
  for (unsigned m = 0; m < l->rate_matrices; ++m)
  {
    const size_t off = par_matrix(R, S, m);
    (void)hipMemcpy(l->par.data() + off, l->dev.par + off, (S + S*(S-1)/2)*sizeof(double), hipMemcpyDeviceToHost);
  }
  l->host_par_stale = false;
}

extern "C" void bpa_set_frequencies(bpa_locus_t * l, unsigned index, const double * f)
{
  std::lock_guard<std::recursive_mutex> lock_(l->eng->mtx);
  if (index >= l->rate_matrices) { fail("bpa_set_frequencies: rate-matrix index out of range"); return; }
  refresh_host_par(l);
  const unsigned S = l->states, R = l->rate_cats;
  std::copy(f, f + S, l->par.begin() + par_matrix(R, S, index) + pm_freqs(S));
  l->eigen_valid[index] = 0;                    // locus.c:895
  l->par_dirty = true; mark_dirty(l);
}

extern "C" void bpa_set_subst_params(bpa_locus_t * l, unsigned index, const double * p)
{
  std::lock_guard<std::recursive_mutex> lock_(l->eng->mtx);
  if (index >= l->rate_matrices) { fail("bpa_set_subst_params: rate-matrix index out of range"); return; }
  refresh_host_par(l);
  const unsigned S = l->states, R = l->rate_cats;
  std::copy(p, p + S*(S-1)/2, l->par.begin() + par_matrix(R, S, index) + pm_subst(S));
  l->eigen_valid[index] = 0;                    // locus.c:883
  l->par_dirty = true; mark_dirty(l);
}

extern "C" void bpa_set_category_rates(bpa_locus_t * l, const double * rates)
{
  std::lock_guard<std::recursive_mutex> lock_(l->eng->mtx);
  refresh_host_par(l);
  std::copy(rates, rates + l->rate_cats, l->par.begin() + par_rates(l->rate_cats));
  l->par_dirty = true; mark_dirty(l);
}

extern "C" void bpa_set_category_weights(bpa_locus_t * l, const double * w)
{
  std::lock_guard<std::recursive_mutex> lock_(l->eng->mtx);
  refresh_host_par(l);
  std::copy(w, w + l->rate_cats, l->par.begin() + par_rate_weights(l->rate_cats));
  l->par_dirty = true; mark_dirty(l);
}

extern "C" void bpa_set_param_indices(bpa_locus_t * l, const unsigned * idx)
{
  std::lock_guard<std::recursive_mutex> lock_(l->eng->mtx);
  refresh_host_par(l);
  for (unsigned k = 0; k < l->rate_cats; ++k) l->par[par_param_idx(l->rate_cats) + k] = (double)idx[k];
  l->par_dirty = true; mark_dirty(l);
}

extern "C" int bpa_set_diploid(bpa_locus_t * l, int unphased_length,
                               const unsigned long * resolution_count,
                               const unsigned long * mapping, unsigned long mapping_len,
                               const unsigned * unphased_weights)
{
  std::lock_guard<std::recursive_mutex> lock_(l->eng->mtx);
  bpa_engine * e = l->eng;
  if (!set_device(e)) return 0;
  std::vector<uint32_t> cnt(unphased_length), map(mapping_len), w(unphased_length);
  unsigned long tot = 0;
  for (int i = 0; i < unphased_length; ++i) { cnt[i] = (uint32_t)resolution_count[i]; tot += resolution_count[i]; w[i] = unphased_weights[i]; }
  if (tot != mapping_len) return fail("bpa_set_diploid: mapping length != sum of resolution counts");
  for (unsigned long i = 0; i < mapping_len; ++i)
  {
    if (mapping[i] >= l->sites) return fail("bpa_set_diploid: mapping entry out of range");
    map[i] = (uint32_t)mapping[i];
  }
  LocusDev & d = l->dev;
  d.dip_count   = (uint32_t *)e->arena.alloc(cnt.size()*4);
  d.dip_map     = (uint32_t *)e->arena.alloc(map.size()*4);
  d.dip_weights = (uint32_t *)e->arena.alloc(w.size()*4);
  if (!d.dip_count || !d.dip_map || !d.dip_weights) return fail("bpa_set_diploid: out of device memory");
  HIPCHK(hipMemcpy(d.dip_count, cnt.data(), cnt.size()*4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d.dip_map, map.data(), map.size()*4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d.dip_weights, w.data(), w.size()*4, hipMemcpyHostToDevice));
  d.unphased_length = (uint32_t)unphased_length;
  e->table_dirty = true;
  return 1;
}

// flush host-side state of dirty loci; refresh eigensystems that were invalidated
static int run_all_queued(bpa_engine * e);
static int flush_state(bpa_engine * e);
static int flush(bpa_engine * e)
{
  if (!flush_state(e)) return 0;
  // queued single-locus updates (the lazy update API) run before anything else touches the buffers
  return e->pending.empty() ? 1 : run_all_queued(e);
}

static int flush_state(bpa_engine * e)
{
  if (!set_device(e)) return 0;
  if (e->table_dirty)
  {
    std::vector<LocusDev> tab(e->loci.size());
    for (size_t i = 0; i < tab.size(); ++i) tab[i] = e->loci[i]->dev;
    if (!e->d_loci.reserve(tab.size())) return fail("out of device memory (locus table)");
    // the table may be read by kernels still in flight on the stream
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipMemcpy(e->d_loci.p, tab.data(), tab.size()*sizeof(LocusDev), hipMemcpyHostToDevice));
    e->table_dirty = false;
    e->pack_dirty = true;
  }
  if (e->dirty.empty()) return 1;
  std::vector<uint32_t> eig;
  HIPCHK(hipStreamSynchronize(e->stream));
  // many loci at once (set-up: every locus of a data set): the pieces are staged and go up in ONE copy + a scatter kernel
  const bool staged = e->dirty.size() >= 8;
  std::vector<UpRec> up_recs;
  std::vector<unsigned char> up_data;
  auto put = [&](void * dst, const void * src, size_t bytes) -> int
  {
    if (!bytes) return 1;
    if (!staged) { HIPCHK(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice)); return 1; }
    const size_t off = (up_data.size() + 15) & ~(size_t)15;
    up_data.resize(off + bytes);
    std::memcpy(up_data.data() + off, src, bytes);
    up_recs.push_back(UpRec{(unsigned char *)dst, (uint64_t)off, (uint32_t)bytes, 0u});
    return 1;
  };
  for (bpa_locus * l : e->dirty)
  {
    l->queued = false;
    if (l->tips_dirty)
    { if (!put(l->dev.tips, l->tipcodes.data(), l->tipcodes.size())) return 0; l->tips_dirty = false; e->pack_dirty = true; }
    if (l->weights_dirty)
    { if (!put(l->dev.weights, l->weights.data(), l->weights.size()*4)) return 0; l->weights_dirty = false; e->pack_dirty = true; }
    bool need_eig = false;
    if (l->needs_eigen())
      for (int v : l->eigen_valid) if (!v) need_eig = true;
    if (l->par_dirty)
    {
      // rates / weights / freqs / exchangeabilities come from the host; the
      // eigensystem part of the block is device-owned and only overwritten
      // when it is about to be recomputed anyway.
      const unsigned S = l->states, R = l->rate_cats;
      if (!put(l->dev.par, l->par.data(), 3*R*sizeof(double))) return 0;
      for (unsigned m = 0; m < l->rate_matrices; ++m)
      {
        const size_t off = par_matrix(R, S, m);
        if (!put(l->dev.par + off, l->par.data() + off, (S + S*(S-1)/2)*sizeof(double))) return 0;
      }
      l->par_dirty = false;
      if (l->rate_cats == 1 && l->dev.model == 0) e->pack_dirty = true;       // the slot table carries a JC69 locus's rate
    }
    if (need_eig) { eig.push_back(l->id); std::fill(l->eigen_valid.begin(), l->eigen_valid.end(), 1); }
  }
  e->dirty.clear();
  if (!up_recs.empty())
  {
    const size_t rb = (up_recs.size()*sizeof(UpRec) + 15) & ~(size_t)15;
    if (!e->d_upload.reserve(rb + up_data.size())) return fail("out of device memory (set-up staging)");
    HIPCHK(hipMemcpy(e->d_upload.p, up_recs.data(), up_recs.size()*sizeof(UpRec), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(e->d_upload.p + rb, up_data.data(), up_data.size(), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(scatter_upload_kernel, dim3((unsigned)up_recs.size()), dim3(64), 0, e->stream,
                       reinterpret_cast<const UpRec *>(e->d_upload.p), (const unsigned char *)(e->d_upload.p + rb));
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(e->stream));          // (the staging buffer is reused by the next flush)
  }
  if (!eig.empty())
  {
    if (!e->d_eigen_list.reserve(eig.size())) return fail("out of device memory (eigen list)");
    HIPCHK(hipMemcpy(e->d_eigen_list.p, eig.data(), eig.size()*4, hipMemcpyHostToDevice));
    const unsigned blocks = (unsigned)((eig.size() + 63)/64);
    bool all4 = true, all20 = true;
    for (uint32_t id : eig) { all4 = all4 && e->loci[id]->states == 4; all20 = all20 && e->loci[id]->states == 20; }
    if (all4)       hipLaunchKernelGGL(eigen_kernel<4>,  dim3(blocks), dim3(64), 0, e->stream, e->d_loci.p, e->d_eigen_list.p, (uint32_t)eig.size());
    else if (all20) hipLaunchKernelGGL(eigen_kernel<20>, dim3(blocks), dim3(64), 0, e->stream, e->d_loci.p, e->d_eigen_list.p, (uint32_t)eig.size());
    else            hipLaunchKernelGGL(eigen_kernel<0>,  dim3(blocks), dim3(64), 0, e->stream, e->d_loci.p, e->d_eigen_list.p, (uint32_t)eig.size());
    HIPCHK(hipGetLastError());
  }
  return 1;
}

// -------------------------------------------------------------------- plans --
template <typename T>
static int upload(DevBuf<T> & b, const T * src, size_t n)
{
  if (!b.reserve(n)) return fail("out of device memory (plan)");
  if (n) HIPCHK(hipMemcpy(b.p, src, n*sizeof(T), hipMemcpyHostToDevice));
  return 1;
}

// The engine's packing for step_jc69_v2_kernel / step_s4_klane_v2_kernel: every 4-state locus of < 256
// lanes (patterns x rate categories) gets a slot (in locus order), whole loci fill workgroups of 256 lanes, and the per-lane and per-slot
// constants go into two device tables shared by all plans.  Rebuilt (after flush) when a locus came, went or changed
// its tip states / weights; the epoch moves only when the slot numbering or a slot's shape did.
constexpr unsigned PACK_BS = 256;       // measured on config 2: 128 lanes 7.6 us, 256 6.9 us, 512 7.8 us per launch
static int engine_pack(bpa_engine * e)
{
  if (e->table_dirty || e->slot_of.size() != e->loci.size()) e->pack_dirty = true;       // a locus was created since
  if (!e->pack_dirty) return 1;
  std::vector<uint32_t> shape, blk{0};
  std::vector<LaneStatic> lanes;
  std::vector<SlotStatic> slots;
  std::vector<int32_t> slot_of(e->loci.size(), -1);
  std::vector<uint32_t> slot_pat;
  unsigned npat_all = 0, maxops_all = 0, rmax_all = 1; bool homog_jc = true, homog_kl = true;
  const LaneStatic idle{0xffffffffu, 0, 0, 0};
  unsigned used = 0;
  for (bpa_locus * l : e->loci)
  {
    // a locus takes sites * rate_cats lanes (lane k*np + n: pattern n, category k — one category: the JC69 / generic
    // one-category layout)
    const bool ok = l->alive && l->states == 4 && l->rate_cats <= 8 && l->sites*l->rate_cats < PACK_BS &&
                    l->tips + l->clv_buffers < 256 && l->prob_matrices < 256 && l->scale_buffers < 128;
    if (!ok) continue;
    const unsigned np = l->sites, R = l->rate_cats, slot = (unsigned)slots.size();
    if (used + np*R > PACK_BS) { lanes.resize(blk.size()*PACK_BS, idle); blk.push_back(slot); used = 0; }
    SlotStatic st{};
    st.clv = l->dev.clv; st.pmat = l->dev.pmat; st.scaler = l->dev.scaler; st.par = l->dev.par;
    st.np = np; st.tips_n = l->tips; st.lane0 = (uint32_t)((blk.size() - 1)*PACK_BS + used); st.locus = l->id;
    st.unphased_length = l->dev.unphased_length; st.rate_cats = R; st.model = l->dev.model; st.pstride = l->dev.pstride; st.tips = l->dev.tips;
    refresh_host_par(l, false);                   // (bpa_plan_set_params_device may have moved the device block ahead)
    st.rate0 = l->par[par_rates(R)];
    slots.push_back(st);
    slot_of[l->id] = (int32_t)slot;
    slot_pat.push_back(npat_all); npat_all += np;
    maxops_all = std::max(maxops_all, l->tips - 1); rmax_all = std::max(rmax_all, R);
    homog_jc = homog_jc && R == 1 && l->dev.model == 0;
    homog_kl = homog_kl && R > 1 && l->scale_buffers == 0 && !l->dev.unphased_length;
    shape.push_back(l->id); shape.push_back(np*R); shape.push_back(l->tips);
    for (unsigned k = 0; k < R; ++k)
      for (unsigned n = 0; n < np; ++n)
      {
        uint32_t codes = 0;                           // <= 8 tips: their codes ride in the lane entry
        for (unsigned tip = 0; tip < l->tips && tip < 8; ++tip) codes |= (uint32_t)(l->tipcodes[(size_t)tip*np + n] & 15u) << (4*tip);
        lanes.push_back(LaneStatic{slot, l->weights[n], codes, n | np << 9 | std::min(l->tips, 31u) << 18 | k << 23 | R << 26});
      }
    used += np*R;
  }
  lanes.resize(blk.size()*PACK_BS, idle);
  blk.push_back((uint32_t)slots.size());
  // kernels in flight may still read the old tables
  HIPCHK(hipStreamSynchronize(e->stream));
  if (!upload(e->d_lane_tab, lanes.data(), lanes.size()) || !upload(e->d_slot_tab, slots.data(), slots.size()) ||
      !upload(e->d_blk_slot_off, blk.data(), blk.size()))
    return 0;
  if (shape != e->pack_shape) { e->pack_epoch++; e->pack_shape.swap(shape); }
  e->slot_of.swap(slot_of);
  e->h_blk_slot_off.swap(blk);
  e->pack_blocks = (unsigned)e->h_blk_slot_off.size() - 1;
  e->pack_slots = (unsigned)slots.size();
  e->pack_slot_pat.swap(slot_pat); e->pack_npat = npat_all; e->pack_maxops = maxops_all; e->pack_rmax = rmax_all;
  e->pack_homog = !e->pack_slots ? 0 : homog_jc ? 1 : homog_kl ? 2 : 0;
  e->pack_dirty = false;
  return 1;
}

static bool validate_op_quiet(const bpa_locus * l, const bpa_op_t & o)      // (callable from several threads: leaves the error text to the caller)
{
  const unsigned nclv = l->tips + l->clv_buffers;
  const int ns = (int)l->scale_buffers;
  return !(o.parent_clv < l->tips || o.parent_clv >= nclv || o.left_clv >= nclv || o.right_clv >= nclv ||
           o.left_pmatrix >= l->prob_matrices || o.right_pmatrix >= l->prob_matrices ||
           o.parent_scaler >= ns || o.left_scaler >= ns || o.right_scaler >= ns);
}
static int validate_op(const bpa_locus * l, const bpa_op_t & o)
{
  const unsigned nclv = l->tips + l->clv_buffers;
  if (o.parent_clv < l->tips || o.parent_clv >= nclv) return fail("op: parent clv index out of range");
  if (o.left_clv >= nclv || o.right_clv >= nclv) return fail("op: child clv index out of range");
  if (o.left_pmatrix >= l->prob_matrices || o.right_pmatrix >= l->prob_matrices) return fail("op: pmatrix index out of range");
  const int ns = (int)l->scale_buffers;
  if (o.parent_scaler >= ns || o.left_scaler >= ns || o.right_scaler >= ns) return fail("op: scaler index out of range");
  return 1;
}

static int plan_build(bpa_plan * p, bpa_engine * e, const bpa_batch_t * b)
{
  if (!set_device(e)) return 0;
  const unsigned T = b->nloci;
  if (!T) return fail("plan: empty batch");
  p->eng = e;
  p->bytes_partials = p->bytes_pmatrix = p->flops_partials = p->bytes_codes = 0; p->node_updates = p->pattern_updates = 0;     // (a plan object may be rebuilt)
  p->fused_klane = p->fused_jc69 = p->jc69_v2 = p->klane_v2 = false; p->fused_rt = 0;
  static const bool prof = BPA_EXP_SWITCH("BPA_PLAN_PROF") != nullptr;
  auto tprev = std::chrono::steady_clock::now();
  auto lap = [&](const char * what) { if (!prof) return; const auto tn = std::chrono::steady_clock::now(); fprintf(stderr, "[plan] %-22s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(tn - tprev).count()); tprev = tn; };
  std::vector<uint32_t> locus(T), pat_off(T + 1, 0), mat_task;
  p->states = b->loci[0]->states;
  p->rmax = 1;
  p->bytes_partials = p->bytes_pmatrix = p->flops_partials = 0;
  p->node_updates = p->pattern_updates = 0;
  const unsigned nmat = b->mat_off ? b->mat_off[T] : 0;
  const unsigned nops = b->op_off ? b->op_off[T] : 0;
  mat_task.resize(nmat);
  std::vector<uint8_t> seen(e->loci.size(), 0);
  for (unsigned t = 0; t < T; ++t)
  {
    const bpa_locus * l = b->loci[t];
    if (!l || l->eng != e || !l->alive) return fail("plan: locus does not belong to this engine");
    // two tasks on one locus would write the same CLV / P-matrix / scaler buffers concurrently
    if (seen[l->id]) return fail("plan: a locus is listed twice in one batch");
    seen[l->id] = 1;
    if (l->states != p->states) return fail("plan: all loci of one batch must have the same number of states");
    locus[t] = l->id;
    pat_off[t+1] = pat_off[t] + l->sites;
    p->rmax = std::max(p->rmax, l->rate_cats);
    const double S = l->states, R = l->rate_cats, Np = l->sites;
    if (b->mat_off)
      for (unsigned i = b->mat_off[t]; i < b->mat_off[t+1]; ++i)
      {
        if (b->mat_pmatrix[i] >= l->prob_matrices) return fail("plan: pmatrix index out of range");
        if (!(b->mat_length[i] >= 0)) return fail("plan: negative branch length");   // assert(t >= 0), core_pmatrix.c:723
        // two branches of one locus writing the same buffer in one step would race
        for (unsigned j = b->mat_off[t]; j < i; ++j)
          if (b->mat_pmatrix[j] == b->mat_pmatrix[i]) return fail("plan: a P-matrix buffer is listed twice for one locus");
        mat_task[i] = t;
        p->bytes_pmatrix += R*S*S*8;
      }
    if (b->op_off)
      for (unsigned i = b->op_off[t]; i < b->op_off[t+1]; ++i)
      {
        if (!validate_op(l, b->ops[i])) return 0;
        // bytes = 3*Np*R*S*8 + 2*R*S^2*8 (+12*Np with scaling); flops = Np*R*(4S^2-S)
        p->bytes_partials += 3*Np*R*S*8 + 2*R*S*S*8 + (b->ops[i].parent_scaler >= 0 ? 12*Np : 0);
        // the same update as the kernels hold the data (`frac_codes`): a tip child is its state codes (1 B DNA, 4 B AA per
        // pattern, once for all R categories), a child that is the previous update's parent is forwarded in registers
        {
          const unsigned prev = i > b->op_off[t] ? b->ops[i-1].parent_clv : ~0u;
          const double code = S == 4 ? 1.0 : 4.0;
          for (const unsigned c : {b->ops[i].left_clv, b->ops[i].right_clv})
            p->bytes_codes += c < l->tips ? Np*code : c == prev ? 0.0 : Np*R*S*8;
          p->bytes_codes += Np*R*S*8 + 2*R*S*S*8 + (b->ops[i].parent_scaler >= 0 ? 12*Np : 0);
        }
        p->flops_partials += Np*R*(4*S*S - S);
        p->node_updates += 1;
        p->pattern_updates += (unsigned long)Np;
      }
    if (b->root_clv)
    {
      if (b->root_clv[t] < l->tips || b->root_clv[t] >= l->tips + l->clv_buffers) return fail("plan: root clv index out of range");
      if (b->root_scaler && b->root_scaler[t] >= (int)l->scale_buffers) return fail("plan: root scaler index out of range");
      p->bytes_partials += Np*R*S*8 + 4*Np;            // K2
      const bool fwd = b->op_off && b->op_off[t+1] > b->op_off[t] && b->ops[b->op_off[t+1]-1].parent_clv == b->root_clv[t];
      p->bytes_codes += (fwd ? 0.0 : Np*R*S*8) + 4*Np;
    }
  }
  lap("validate");
  const unsigned P = pat_off[T];
  p->h_locus = locus;
  std::vector<int32_t> rs(T, BPA_SCALE_BUFFER_NONE);
  std::vector<uint32_t> rc(T, 0), zero_off(T + 1, 0);
  if (b->root_scaler) std::copy(b->root_scaler, b->root_scaler + T, rs.begin());
  if (b->root_clv) std::copy(b->root_clv, b->root_clv + T, rc.begin());
  else for (unsigned t = 0; t < T; ++t) rc[t] = b->loci[t]->tips;

  if (!upload(p->task_locus, locus.data(), T)) return 0;
  if (!upload(p->task_pat_off, pat_off.data(), T + 1)) return 0;
  if (!upload(p->mat_off, b->mat_off ? b->mat_off : zero_off.data(), T + 1)) return 0;
  if (!upload(p->mat_task, mat_task.data(), nmat)) return 0;
  if (!upload(p->mat_pmatrix, b->mat_pmatrix, nmat)) return 0;
  if (!upload(p->mat_length, b->mat_length, nmat)) return 0;
  if (!upload(p->op_off, b->op_off ? b->op_off : zero_off.data(), T + 1)) return 0;
  static_assert(sizeof(OpDev) == sizeof(bpa_op_t), "op layout");
  if (!upload(p->ops, reinterpret_cast<const OpDev *>(b->ops), nops)) return 0;
  if (!upload(p->root_clv, rc.data(), T)) return 0;
  if (!upload(p->root_scaler, rs.data(), T)) return 0;
  if (!p->thr_task.reserve(P) || !p->site_term.reserve(P) || !p->lnl.reserve(T))
    return fail("out of device memory (plan)");

  lap("csr uploads");
  PlanDev & d = p->pd;
  d.task_locus = p->task_locus.p; d.task_pat_off = p->task_pat_off.p; d.thr_task = p->thr_task.p;
  d.mat_off = p->mat_off.p; d.mat_task = p->mat_task.p; d.mat_pmatrix = p->mat_pmatrix.p;
  d.mat_length = p->mat_length.p; d.op_off = p->op_off.p; d.ops = p->ops.p;
  d.root_clv = p->root_clv.p; d.root_scaler = p->root_scaler.p; d.site_term = p->site_term.p;
  d.dbg = nullptr;
  d.lnl = p->lnl.p; d.ntasks = T; d.npatterns = P; d.nmat = nmat; d.pad = 0;
  p->has_mats = nmat > 0;
  p->has_lnl = b->root_clv != nullptr;

  // tiled path (20 states): one workgroup per 128-pattern tile of one locus
  p->ntiles = 0; d.tile_task = d.tile_n0 = nullptr;
  // 20-state partials kernel (kernels.hpp): the pipelined kernel (partials_lnl_pipe20_kernel: P-matrices global -> LDS
  // direct and double-buffered, one wave per rate category, CLV planes streamed, 2 waves per SIMD).  BPA_S20_KERNEL
  // selects the others, kept for comparison and as fall-backs: pipemfma (the same update on the matrix cores, on the same
  // memory pipeline) | tiled (one wave runs all categories: what loci of more than 4 categories get) | generic (no staging)
  {
    const char * v = getenv("BPA_S20_KERNEL");
    p->s20_kernel = v ? v : "wave";
    if (p->s20_kernel != "wave" && p->s20_kernel != "waverl" && p->s20_kernel != "wave2" && p->s20_kernel != "pipe" && p->s20_kernel != "pipemfma" && p->s20_kernel != "tiled" && p->s20_kernel != "generic") p->s20_kernel = "wave";
#ifndef BPA_EXPERIMENTAL
    // (the default build holds the default, north_star's matrix-core kernel and the fall-back of loci with more than 4 categories;
    //  waverl / wave2 / pipe / generic are csrc/experimental's: asking for one of them here is an error, not a silent default)
    if (p->s20_kernel != "wave" && p->s20_kernel != "pipemfma" && p->s20_kernel != "tiled")
      return fail("BPA_S20_KERNEL: this 20-state kernel is part of an experimental build only (-DBPA_EXPERIMENTAL)");
#endif
  }
  p->s20_tiledk = p->s20_kernel == "wave" || p->s20_kernel == "waverl" || p->s20_kernel == "wave2" || p->s20_kernel == "pipe" || p->s20_kernel == "pipemfma";
  if (p->s20_tiledk && p->rmax > 4) { p->s20_tiledk = false; p->s20_kernel = "tiled"; }     // 64 x R lanes must fit a 256-lane workgroup
  p->tile = (p->s20_tiledk && p->s20_kernel != "wave2") ? 64 : 128;       // (wave2: two patterns per lane)
  if (p->states == 20)
  {
    std::vector<uint32_t> tt, tn;
    for (unsigned t = 0; t < T; ++t)
      for (unsigned n0 = 0; n0 < b->loci[t]->sites; n0 += p->tile) { tt.push_back(t); tn.push_back(n0); }
    if (!upload(p->tile_task, tt.data(), tt.size()) || !upload(p->tile_n0, tn.data(), tn.size())) return 0;
    d.tile_task = p->tile_task.p; d.tile_n0 = p->tile_n0.p;
    p->ntiles = (unsigned)tt.size();
  }

  // fused single-launch path: pack whole loci into workgroups (4-state loci that fit one)
  p->fused_bs = 0; d.nblocks = 0; d.flags = 0;
  d.blk_task_off = d.lane_task = d.task_lane0 = nullptr;
  unsigned maxnp = 0;
  for (unsigned t = 0; t < T; ++t) maxnp = std::max(maxnp, b->loci[t]->sites);
  if (p->states == 4 && maxnp <= 256)
  {
    // workgroup size (lanes = patterns, whole loci per workgroup)
    unsigned BS = maxnp <= 16 ? 256 : (maxnp <= 64 ? 64 : 256);     // measured: config 2 +3 % with 256, config 3 +8 % with 64
    if (const char * ov = BPA_EXP_SWITCH("BPA_FUSED_BS")) { const unsigned v = (unsigned)atoi(ov); if ((v == 64 || v == 256) && maxnp <= v) BS = v; }
    // one lane per (pattern, rate category) when every locus has several categories, no scalers and no
    // phase averaging (step_s4_klane_kernel)
    bool klane = p->rmax > 1 && !BPA_EXP_SWITCH("BPA_NO_KLANE");
    unsigned maxlanes = 0;
    for (unsigned t = 0; t < T && klane; ++t)
    {
      const bpa_locus * l = b->loci[t];
      klane = l->rate_cats > 1 && l->scale_buffers == 0 && !l->dev.unphased_length && (!b->root_scaler || b->root_scaler[t] < 0);
      maxlanes = std::max(maxlanes, l->sites*l->rate_cats);
    }
    klane = klane && maxlanes <= 256;
    if (klane) BS = maxlanes <= 128 ? 128 : 256;
    p->fused_klane = klane;
    std::vector<uint32_t> blk_off{0}, lane_task, lane0(T);
    unsigned used = 0;
    for (unsigned t = 0; t < T; ++t)
    {
      const unsigned np = b->loci[t]->sites*(klane ? b->loci[t]->rate_cats : 1u);      // lanes of this locus
      if (used + np > BS)
      {
        lane_task.resize(blk_off.size()*BS, 0xffffffffu);
        blk_off.push_back(t); used = 0;
      }
      lane0[t] = (uint32_t)((blk_off.size() - 1)*BS + used);
      for (unsigned n = 0; n < np; ++n) lane_task.push_back(t);
      used += np;
    }
    lane_task.resize(blk_off.size()*BS, 0xffffffffu);
    blk_off.push_back(T);
    if (!upload(p->blk_task_off, blk_off.data(), blk_off.size())) return 0;
    if (!upload(p->lane_task, lane_task.data(), lane_task.size())) return 0;
    if (!upload(p->task_lane0, lane0.data(), T)) return 0;
    d.blk_task_off = p->blk_task_off.p; d.lane_task = p->lane_task.p; d.task_lane0 = p->task_lane0.p;
    d.nblocks = (uint32_t)blk_off.size() - 1;
    p->fused_bs = BS;

    // flattened records: TaskRec header (6 x 16 B) + the task's node updates (2 x 16 B each,
    // at least two slots so the eager loads of the first two stay in bounds)
    static_assert(sizeof(TaskRec) == 96 && sizeof(OpDev) == 32 && sizeof(MatRec) == 32, "record layout");
    std::vector<uint4> recs;
    std::vector<uint32_t> task_rec(T + 1), lane_rec(lane_task.size(), 0xffffffffu);      // [T]: end of the last record
    std::vector<MatRec> mrecs(nmat);
    static_assert(sizeof(OpSlot) == 48, "op slot layout");
    bool all1 = true, all4 = true, all_jc = true;
    for (unsigned t = 0; t < T; ++t)
    {
      const bpa_locus * l = b->loci[t];
      const unsigned nops_t = b->op_off ? b->op_off[t+1] - b->op_off[t] : 0;
      all1 = all1 && l->rate_cats == 1; all4 = all4 && l->rate_cats == 4;
      TaskRec r{};
      r.clv = l->dev.clv; r.pmat = l->dev.pmat; r.scaler = l->dev.scaler; r.tips = l->dev.tips;
      r.weights = l->dev.weights; r.par = l->dev.par;
      r.np = l->sites; r.tips_n = l->tips; r.rate_cats = l->rate_cats; r.lane0 = lane0[t];
      r.nops = nops_t; r.root_clv = rc[t]; r.root_scaler = rs[t]; r.task = t;
      r.unphased_length = l->dev.unphased_length; r.pat_off = pat_off[t]; r.locus = l->id; r.pstride = l->dev.pstride;
      task_rec[t] = (uint32_t)recs.size();
      // 48-byte op slots: the update + for each child the entry of this step's branch-length
      // list when that child's P-matrix is updated in this very step (-1 otherwise); at least
      // three slots so the eager loads of the first three stay in bounds
      const size_t units = 6 + 3*std::max(nops_t, 3u);
      recs.resize(recs.size() + units, uint4{0, 0, 0, 0});
      std::memcpy(&recs[task_rec[t]], &r, sizeof(r));
      for (unsigned o = 0; o < nops_t; ++o)
      {
        OpSlot sl{};
        std::memcpy(&sl.op, b->ops + b->op_off[t] + o, sizeof(OpDev));
        sl.left_e = sl.right_e = -1;
        if (b->mat_off)
          for (unsigned i = b->mat_off[t]; i < b->mat_off[t+1]; ++i)
          {
            if (b->mat_pmatrix[i] == sl.op.left_pmatrix)  sl.left_e = (int32_t)i;
            if (b->mat_pmatrix[i] == sl.op.right_pmatrix) sl.right_e = (int32_t)i;
          }
        std::memcpy(&recs[task_rec[t] + 6 + 3*o], &sl, sizeof(sl));
      }
      all_jc = all_jc && l->dev.model == 0;
      for (unsigned n = 0; n < l->sites*(klane ? l->rate_cats : 1u); ++n) lane_rec[lane0[t] + n] = task_rec[t];
      if (b->mat_off)
        for (unsigned i = b->mat_off[t]; i < b->mat_off[t+1]; ++i)
        {
          MatRec & m = mrecs[i];
          m.dst = l->dev.pmat + (size_t)b->mat_pmatrix[i]*l->rate_cats*l->dev.pstride;
          m.par = l->dev.par; m.rate_cats = l->rate_cats; m.model = l->dev.model; m.entry = i; m.pad = 0;
        }
    }
    lap("records (host)");
    if (!upload(p->recs, recs.data(), recs.size())) return 0;
    if (!upload(p->lane_rec, lane_rec.data(), lane_rec.size())) return 0;
    task_rec[T] = (uint32_t)recs.size();
    if (!upload(p->task_rec, task_rec.data(), T + 1)) return 0;
    if (!upload(p->mat_recs, mrecs.data(), nmat)) return 0;
    d.recs = p->recs.p; d.lane_rec = p->lane_rec.p; d.task_rec = p->task_rec.p; d.mat_recs = p->mat_recs.p;
    p->fused_rt = all1 ? 1 : (all4 ? 4 : 0);
    lap("records (upload)");
    p->fused_jc69 = all1 && all_jc && !BPA_EXP_SWITCH("BPA_NO_JC69_FAST");
    d.pad = p->rmax;

    // compact records over the engine's packing (step_jc69_v2_kernel): the plan's loci must be packed and come in slot order
    p->jc69_v2 = p->klane_v2 = false;
    if ((p->fused_jc69 && !BPA_EXP_SWITCH("BPA_JC69_V1")) || (p->fused_klane && !BPA_EXP_SWITCH("BPA_KLANE_V1")))
    {
      if (!engine_pack(e)) return 0;
      bool ok = e->pack_slots > 0;
      int prev = -1;
      unsigned maxops = 0;
      for (unsigned t = 0; t < T && ok; ++t)
      {
        const int sl = e->slot_of[b->loci[t]->id];
        ok = sl > prev;
        prev = sl;
        maxops = std::max(maxops, b->op_off ? b->op_off[t+1] - b->op_off[t] : 0u);
      }
      ok = ok && maxops <= 255;                      // StepRec counts a locus's updates in a byte
      if (ok)
      {
        const unsigned units = 1 + std::max(maxops, 3u);
        std::vector<uint4> r2((size_t)e->pack_slots*units, uint4{0, 0, 0, 0});
        for (unsigned sl = 0; sl < e->pack_slots; ++sl) reinterpret_cast<StepRec *>(&r2[(size_t)sl*units])->task = 0xffffffffu;
        std::vector<MatRec2> m2(nmat);
        std::vector<uint32_t> slot_mat0(e->pack_slots + 1, 0);          // first fresh matrix of the slot's locus (prefix)
        for (unsigned t = 0; t < T; ++t)
        {
          const bpa_locus * l = b->loci[t];
          const unsigned sl = (unsigned)e->slot_of[l->id];
          const unsigned nops_t = b->op_off ? b->op_off[t+1] - b->op_off[t] : 0;
          StepRec h{};
          h.task = t; h.pat_off = pat_off[t]; h.root_clv = (uint8_t)rc[t]; h.root_scaler = (int8_t)rs[t]; h.nops = (uint8_t)nops_t;
          std::memcpy(&r2[(size_t)sl*units], &h, sizeof(h));
          for (unsigned o = 0; o < nops_t; ++o)
          {
            const bpa_op_t & s = b->ops[b->op_off[t] + o];
            StepOp q{};
            q.parent_clv = (uint8_t)s.parent_clv; q.left_clv = (uint8_t)s.left_clv; q.right_clv = (uint8_t)s.right_clv;
            q.left_pmatrix = (uint8_t)s.left_pmatrix; q.right_pmatrix = (uint8_t)s.right_pmatrix;
            q.parent_scaler = (int8_t)s.parent_scaler; q.left_scaler = (int8_t)s.left_scaler; q.right_scaler = (int8_t)s.right_scaler;
            q.left_e = q.right_e = -1;
            if (b->mat_off)
              for (unsigned i = b->mat_off[t]; i < b->mat_off[t+1]; ++i)
              {
                if (b->mat_pmatrix[i] == s.left_pmatrix)  q.left_e = (int32_t)i;
                if (b->mat_pmatrix[i] == s.right_pmatrix) q.right_e = (int32_t)i;
              }
            std::memcpy(&r2[(size_t)sl*units + 1 + o], &q, sizeof(q));
          }
          if (b->mat_off)
          {
            for (unsigned i = b->mat_off[t]; i < b->mat_off[t+1]; ++i) m2[i] = MatRec2{sl, b->mat_pmatrix[i]};
            slot_mat0[sl + 1] = b->mat_off[t+1] - b->mat_off[t];
          }
        }
        // the plan's loci are in slot order, so are their fresh matrices: a workgroup's are one range
        for (unsigned sl = 0; sl < e->pack_slots; ++sl) slot_mat0[sl + 1] += slot_mat0[sl];
        std::vector<uint32_t> bm(e->pack_blocks + 1);
        for (unsigned k = 0; k <= e->pack_blocks; ++k) bm[k] = slot_mat0[e->h_blk_slot_off[k]];
        if (!upload(p->recs2, r2.data(), r2.size()) || !upload(p->mat2, m2.data(), nmat) || !upload(p->blk_mat_off, bm.data(), bm.size()))
          return 0;
        d.recs2 = p->recs2.p; d.mat2 = p->mat2.p; d.blk_mat_off = p->blk_mat_off.p; d.rec2_units = units;
        p->jc69_v2 = p->fused_jc69; p->klane_v2 = !p->fused_jc69; p->pack_epoch = e->pack_epoch;
      }
    }
  }

  lap("v2 / rest");
  hipLaunchKernelGGL(build_thr_task_kernel, dim3((P + BPA_BLOCK - 1)/BPA_BLOCK), dim3(BPA_BLOCK), 0, e->stream,
                     p->task_pat_off.p, T, P, p->thr_task.p);
  HIPCHK(hipGetLastError());
  return 1;
}

static TimingSlot * next_slot(bpa_engine * e)
{
  if (e->slots_used == e->slots.size())
  {
    if (e->slots.size() >= 8192) return nullptr;       // collect() drains them
    TimingSlot s;
    for (auto & ev : s.ev) if (hipEventCreate(&ev) != hipSuccess) return nullptr;
    e->slots.push_back(s);
  }
  return &e->slots[e->slots_used++];
}

static int timing_drain(bpa_engine * e)
{
  if (!e->slots_used) return 1;
  HIPCHK(hipStreamSynchronize(e->stream));
  for (size_t i = 0; i < e->slots_used; ++i)
  {
    for (int j = 0; j < 3; ++j)
    {
      if (e->slots[i].ev_used == 2 && j != 1) continue;
      float ms = 0;
      HIPCHK(hipEventElapsedTime(&ms, e->slots[i].ev[j], e->slots[i].ev[j+1]));
      e->acc_ms[j] += ms;
    }
    e->acc_launches++;
    e->acc_steps += e->slots[i].steps; e->acc_bytes += e->slots[i].bytes; e->acc_bytes_codes += e->slots[i].bytes_codes;
  }
  e->slots_used = 0;
  return 1;
}

// mode bits: 1 = P-matrices, 2 = partials (+ per-pattern lnL terms), 4 = per-locus reduction
// the K1 + K2 launch of the compact-record path for loci of several rate categories: step_s4_klane_v3_kernel (the step's
// records and all its P-matrices in LDS, three global round trips per step) wherever a slot record fits the 16 units a
// lane group brings in one load — loci of up to 16 tips —, step_s4_klane_v2_kernel otherwise (BPA_KLANE_V2=1: always, A/B)
template <bool FUSE_A>
static void launch_klane(const dim3 grid, hipStream_t st, hipEvent_t k0, hipEvent_t k1, const PlanDev & d)
{
  static const bool env_v2 = BPA_EXP_SWITCH("BPA_KLANE_V2") != nullptr;
  if (!env_v2 && d.rec2_units >= 2u && d.rec2_units <= 16u)
  {
    const size_t lds = (size_t)(PACK_BS/64)*BPA_KLANE_CH*1024u;            // (the wave's corner holds the matrices of BPA_KLANE_CH updates at a time)
    hipExtLaunchKernelGGL((step_s4_klane_v3_kernel<PACK_BS, FUSE_A>), grid, dim3(PACK_BS), lds, st, k0, k1, 0, d);
  }
  else
    hipExtLaunchKernelGGL((step_s4_klane_v2_kernel<PACK_BS, false, 0, FUSE_A>), grid, dim3(PACK_BS), 0, st, k0, k1, 0, d);
}

static int plan_launch_mode(bpa_plan * p, int mode)
{
  // A/B switches of DESIGN.md's appendix, read once
  static const bool env_fused_split = BPA_EXP_SWITCH("BPA_FUSED_SPLIT") != nullptr;
  bpa_engine * e = p->eng;
  if (!flush(e)) return 0;
  PlanDev d = p->pd;
  d.loci = e->d_loci.p;
  d.bfbeta = e->bfbeta;
  TimingSlot * ts = nullptr;
  if (e->timing && (e->timing_phase++ % e->timing_stride) == 0)
  {
    ts = next_slot(e);
    if (!ts) { if (!timing_drain(e)) return 0; ts = next_slot(e); }
  }
  if (p->fused_bs)
  {
    // one launch: P-matrices -> node updates + site terms -> per-locus lnL.  With timing on,
    // the dispatch carries its own start/stop events (exact kernel time, as rocprofv3 sees it).
    d.flags = (((mode & 1) && p->has_mats) ? 1u : 0u) | ((mode & 2) ? 2u : 0u) | ((mode & 4) ? 4u : 0u);
    hipEvent_t k0 = ts ? ts->ev[1] : nullptr, k1 = ts ? ts->ev[2] : nullptr;
    const dim3 grid(d.nblocks);
    bool summed = false;                 // the step kernel delivered the plan's (partial) sums itself
#define BPA_FUSED(BS_, RT_) hipExtLaunchKernelGGL((step_s4_fused_kernel<BS_, RT_>), grid, dim3(BS_), 0, e->stream, k0, k1, 0, d)
    if (p->klane_v2 && engine_pack(e) && p->pack_epoch == e->pack_epoch)
    {
      // the multi-category kernel on the compact records: P-matrix phase as its own launch, then updates + sums
      d.lane_tab = e->d_lane_tab.p; d.slot_tab = e->d_slot_tab.p; d.blk_slot_off = e->d_blk_slot_off.p; d.nblocks2 = e->pack_blocks;
      const dim3 g2(e->pack_blocks);
      if (d.flags & 1u)
      {
        PlanDev da = d; da.flags = 1u;
        static const bool pm_packed = BPA_EXP_SWITCH("BPA_PMAT_PACKED") != nullptr;        // A/B: the phase on the packing's workgroups
        if (pm_packed) hipLaunchKernelGGL((step_s4_klane_v2_kernel<PACK_BS, true>), g2, dim3(PACK_BS), 0, e->stream, da);
        else hipLaunchKernelGGL(pmatrix_s4_dense_kernel, dim3((d.nmat*std::max(da.pad, 1u) + 255u)/256u), dim3(256), 0, e->stream, da, d.nmat);
      }
      d.flags &= 6u;
      if (d.flags)
      {
        static const bool klane_direct = BPA_EXP_SWITCH("BPA_KLANE_DIRECT") != nullptr;     // A/B: no LDS staging of the P-matrices
        if (klane_direct) d.flags |= 16u;
        if ((mode & 4) && p->sum_out && p->sum_parts == e->pack_blocks) { d.flags |= 8u; d.wg_part = p->sum_out; summed = true; }
        // (register budget, measured with BPA_KLANE_OCC builds: the compiler's 126 VGPRs = 4 waves per SIMD 105 us; held to
        //  5 waves 121 us and to 6 waves 187 us (spills), to 3 waves 119 us)
        launch_klane<false>(g2, e->stream, k0, k1, d);
      }
    }
    else if (p->fused_klane)
    {
      // phase A on its own (its code is what needs the registers), then B + C; the events bracket B + C
      if (d.flags & 1u)
      {
        PlanDev da = d; da.flags = 1u;
        if (p->fused_bs == 128) hipLaunchKernelGGL((step_s4_klane_kernel<128, true>), grid, dim3(128), 0, e->stream, da);
        else                    hipLaunchKernelGGL((step_s4_klane_kernel<256, true>), grid, dim3(256), 0, e->stream, da);
      }
      d.flags &= 6u;
      if (d.flags)
      {
        if (p->fused_bs == 128) hipExtLaunchKernelGGL((step_s4_klane_kernel<128, false>), grid, dim3(128), 0, e->stream, k0, k1, 0, d);
        else                    hipExtLaunchKernelGGL((step_s4_klane_kernel<256, false>), grid, dim3(256), 0, e->stream, k0, k1, 0, d);
      }
    }
    else if (p->jc69_v2 && !d.dbg && engine_pack(e) && p->pack_epoch == e->pack_epoch)       // (the stamps of bpa_plan_probe live in the first version)
    {
      d.lane_tab = e->d_lane_tab.p; d.slot_tab = e->d_slot_tab.p; d.blk_slot_off = e->d_blk_slot_off.p; d.nblocks2 = e->pack_blocks;
      if ((mode & 4) && p->sum_out && p->sum_parts == e->pack_blocks) { d.flags |= 8u; d.wg_part = p->sum_out; summed = true; }
      hipExtLaunchKernelGGL((step_jc69_v2_kernel<PACK_BS>), dim3(e->pack_blocks), dim3(PACK_BS), 0, e->stream, k0, k1, 0, d);
    }
    else if (p->fused_jc69)
    {
      if (p->fused_bs == 64) hipExtLaunchKernelGGL((step_jc69_kernel<64, true>), grid, dim3(64), 0, e->stream, k0, k1, 0, d);
      else                        hipExtLaunchKernelGGL((step_jc69_kernel<256, true>), grid, dim3(256), 0, e->stream, k0, k1, 0, d);
    }
    else if (p->fused_bs == 64 && p->fused_rt == 4 && (d.flags & 1u) && (d.flags & 6u) && env_fused_split)
    {
      // phase A on its own (no events), then B+C with the lighter register footprint
      PlanDev da = d; da.flags = 1u;
      hipLaunchKernelGGL((step_s4_fused_kernel<64, 4, 1>), grid, dim3(64), 0, e->stream, da);
      d.flags &= 6u;
      hipExtLaunchKernelGGL((step_s4_fused_kernel<64, 4, 6>), grid, dim3(64), 0, e->stream, k0, k1, 0, d);
    }
    else if (p->fused_bs == 64)
    {
      if (p->fused_rt == 1) BPA_FUSED(64, 1); else if (p->fused_rt == 4) BPA_FUSED(64, 4); else BPA_FUSED(64, 0);
    }
    else
    {
      if (p->fused_rt == 1) BPA_FUSED(256, 1); else if (p->fused_rt == 4) BPA_FUSED(256, 4); else BPA_FUSED(256, 0);
    }
#undef BPA_FUSED
    HIPCHK(hipGetLastError());
    if ((mode & 4) && p->sum_out && !summed)
    {
      hipLaunchKernelGGL(lnl_sum_kernel, dim3(1), dim3(1024), 0, e->stream, d.lnl, d.ntasks, p->sum_out);
      HIPCHK(hipGetLastError());
      if (p->sum_parts > 1) HIPCHK(hipMemsetAsync(p->sum_out + 1, 0, (p->sum_parts - 1)*sizeof(double), e->stream));
    }
    if (ts)
    {
      // what the event pair brackets: the one fused kernel (K4 + K1 + K2), or — where the P-matrix phase is its own
      // launch (the multi-category kernels) — K1 + K2 alone
      const bool split_a = p->klane_v2 || p->fused_klane;
      ts->ev_used = 2; ts->steps = 1;
      ts->bytes = ((mode & 2) ? p->bytes_partials : 0.0) + ((!split_a && (mode & 1) && p->has_mats) ? p->bytes_pmatrix : 0.0);
      ts->bytes_codes = ((mode & 2) ? p->bytes_codes : 0.0) + ((!split_a && (mode & 1) && p->has_mats) ? p->bytes_pmatrix : 0.0);
    }
    return 1;
  }
  if (ts) { ts->ev_used = 4; ts->steps = 1; ts->bytes = (mode & 2) ? p->bytes_partials : 0.0; ts->bytes_codes = (mode & 2) ? p->bytes_codes : 0.0; }
  if (ts) HIPCHK(hipEventRecord(ts->ev[0], e->stream));
  if ((mode & 1) && p->has_mats)
  {
    if (p->states == 4)
    {
      const unsigned n = d.nmat*p->rmax;
      hipLaunchKernelGGL(pmatrix_s4_kernel, dim3((n + BPA_BLOCK - 1)/BPA_BLOCK), dim3(BPA_BLOCK), 0, e->stream, d, p->rmax);
    }
    else
    {
      hipLaunchKernelGGL(pmatrix_wg2_kernel<20>, dim3(d.nmat), dim3(256), 0, e->stream, d);
    }
    HIPCHK(hipGetLastError());
  }
  if (ts) HIPCHK(hipEventRecord(ts->ev[1], e->stream));
  if (mode & 2)
  {
    const unsigned blocks = (d.npatterns + BPA_BLOCK - 1)/BPA_BLOCK;
    if (p->states == 4)
      hipLaunchKernelGGL(partials_lnl_s4_kernel, dim3(blocks), dim3(BPA_BLOCK), 0, e->stream, d);
    else if (p->ntiles && p->s20_tiledk)
    {
      d.flags = 4u; d.pad = p->rmax;
      static const bool no_xcd = BPA_EXP_SWITCH("BPA_NO_XCD_MAP") != nullptr;       // A/B: workgroup b = tile b
      if (no_xcd) d.flags |= 32u;
      const dim3 grid(p->ntiles), block(64*p->rmax);
      if (p->s20_kernel == "pipemfma")
        hipLaunchKernelGGL((partials_lnl_pipemfma20_kernel<false, 2>), grid, block, ((size_t)4*p->rmax*400 + (size_t)p->rmax*64)*sizeof(double), e->stream, d);
#ifdef BPA_EXPERIMENTAL
      else if (p->s20_kernel == "pipe")
        hipLaunchKernelGGL((partials_lnl_pipe20_kernel<20, true, 2>), grid, block, ((size_t)4*p->rmax*400 + (size_t)p->rmax*64)*sizeof(double), e->stream, d);
      else if (p->s20_kernel == "waverl")
        hipLaunchKernelGGL((partials_lnl_wave20_kernel<20, true, 2, 1, true>), grid, block, ((size_t)4*p->rmax*400 + (size_t)p->rmax*64)*sizeof(double), e->stream, d);
      else if (p->s20_kernel == "wave2")
        hipLaunchKernelGGL((partials_lnl_wave20_kernel<20, true, 1, 2>), grid, block, ((size_t)4*p->rmax*400 + (size_t)2*p->rmax*64)*sizeof(double), e->stream, d);
#endif
      else
        hipLaunchKernelGGL((partials_lnl_wave20_kernel<20, true, 2>), grid, block, ((size_t)4*p->rmax*400 + (size_t)p->rmax*64)*sizeof(double), e->stream, d);
    }
    else if (p->ntiles && p->tile == 128 && p->s20_kernel == "tiled")
    {
      d.flags = 4u;
      const size_t lds = (size_t)2*p->rmax*400*sizeof(double);
      hipLaunchKernelGGL((partials_lnl_tiled_kernel<20, 128>), dim3(p->ntiles), dim3(128), lds, e->stream, d);
    }
    else
#ifdef BPA_EXPERIMENTAL
      hipLaunchKernelGGL(partials_lnl_sN_kernel<20>, dim3(blocks), dim3(BPA_BLOCK), 0, e->stream, d);
#else
      return fail("no 20-state kernel for this plan (BPA_S20_KERNEL=generic is an experimental build's)");
#endif
    HIPCHK(hipGetLastError());
  }
  if (ts) HIPCHK(hipEventRecord(ts->ev[2], e->stream));
  if (mode & 4)
  {
    hipLaunchKernelGGL(lnl_reduce_wave_kernel, dim3(d.ntasks), dim3(64), 0, e->stream, d);
    HIPCHK(hipGetLastError());
  }
  if ((mode & 4) && p->sum_out)
  {
    hipLaunchKernelGGL(lnl_sum_kernel, dim3(1), dim3(1024), 0, e->stream, d.lnl, d.ntasks, p->sum_out);
    HIPCHK(hipGetLastError());
    if (p->sum_parts > 1) HIPCHK(hipMemsetAsync(p->sum_out + 1, 0, (p->sum_parts - 1)*sizeof(double), e->stream));
  }
  if (ts) HIPCHK(hipEventRecord(ts->ev[3], e->stream));
  return 1;
}

extern "C" bpa_plan_t * bpa_plan_create(bpa_engine_t * e, const bpa_batch_t * b)
{
  if (!e || !b) { fail("bpa_plan_create: null argument"); return nullptr; }
  std::lock_guard<std::recursive_mutex> lock_(e->mtx);
  bpa_plan * p = new bpa_plan();
  if (!plan_build(p, e, b)) { delete p; return nullptr; }
  return p;
}

extern "C" void bpa_plan_destroy(bpa_plan_t * p)
{
  if (!p) return;
  std::lock_guard<std::recursive_mutex> lock_(p->eng->mtx);
  (void)set_device(p->eng);
  (void)hipStreamSynchronize(p->eng->stream);
  delete p;
}

extern "C" int bpa_plan_set_lengths(bpa_plan_t * p, const double * len)
{
  std::lock_guard<std::recursive_mutex> lock_(p->eng->mtx);
  if (!set_device(p->eng)) return 0;
  for (unsigned i = 0; i < p->pd.nmat; ++i)
    if (!(len[i] >= 0)) return fail("plan: negative branch length");   // assert(t >= 0), core_pmatrix.c:723
  HIPCHK(hipMemcpyAsync(p->mat_length.p, len, p->pd.nmat*sizeof(double), hipMemcpyHostToDevice, p->eng->stream));
  HIPCHK(hipStreamSynchronize(p->eng->stream));
  return 1;
}

extern "C" int bpa_plan_launch(bpa_plan_t * p)
{
  std::lock_guard<std::recursive_mutex> lock_(p->eng->mtx);
  if (!p->eng->usedata) return 1;                 // opt_usedata == 0 (locus.c:2424)
  return plan_launch_mode(p, 1 | 2 | (p->has_lnl ? 4 : 0));
}

// profiling aid (tools/): per-workgroup wall-clock stamps of the fused JC69 kernel.
// out[0..7]: mean over workgroups of (stamp i - earliest stamp 0) in microseconds; out[8]: span
extern "C" int bpa_plan_probe(bpa_plan_t * p, double * out)
{
  std::lock_guard<std::recursive_mutex> lock_(p->eng->mtx);
  bpa_engine * e = p->eng;
  if (!set_device(e) || !p->fused_bs) return fail("probe: fused plans only");
  const unsigned B = (p->klane_v2 || p->jc69_v2) ? e->pack_blocks : p->pd.nblocks;
  if (!p->dbg.reserve((size_t)B*8)) return fail("oom");
  HIPCHK(hipMemset(p->dbg.p, 0, (size_t)B*8*8));
  p->pd.dbg = p->dbg.p;
  int ok = bpa_plan_launch(p);
  p->pd.dbg = nullptr;
  if (!ok) return 0;
  HIPCHK(hipStreamSynchronize(e->stream));
  std::vector<unsigned long long> h((size_t)B*8);
  HIPCHK(hipMemcpy(h.data(), p->dbg.p, h.size()*8, hipMemcpyDeviceToHost));
  unsigned long long first = ~0ull, last = 0;
  for (unsigned b = 0; b < B; ++b) { first = std::min(first, h[b*8]); for (int i = 0; i < 8; ++i) last = std::max(last, h[b*8+i]); }
  for (int i = 0; i < 8; ++i)
  {
    double acc = 0; unsigned cnt = 0;
    for (unsigned b = 0; b < B; ++b) if (h[b*8+i]) { acc += (double)(h[b*8+i] - first)*0.01; ++cnt; }
    out[i] = cnt ? acc/cnt : -1;
  }
  out[8] = (double)(last - first)*0.01;
  // out[9..16]: mean over workgroups of (stamp i - the workgroup's OWN stamp 0): where a workgroup's life goes; out[17]: workgroups stamped
  unsigned nb = 0;
  for (int i = 0; i < 8; ++i)
  {
    double acc = 0; unsigned cnt = 0;
    for (unsigned b = 0; b < B; ++b) if (h[b*8+i] && h[b*8]) { acc += (double)(h[b*8+i] - h[b*8])*0.01; ++cnt; }
    out[9 + i] = cnt ? acc/cnt : -1;
    if (i == 0) nb = cnt;
  }
  out[17] = nb;
  return 1;
}

// Can this resident plan be a link of a chain launch (step_jc69_v2_chain_kernel)?  JC69 on the compact records of
// the current packing, a per-locus step (no plan total: an all-loci step is decided on a sum over ALL loci and is a
// launch of its own).
static bool chainable(bpa_plan * p)
{
  bpa_engine * e = p->eng;
  // a plan that leaves its total as per-workgroup partial sums (written by the step kernel itself) chains like any other;
  // one whose total is a launch of its own (lnl_sum_kernel) ends a chain
  const bool sum_in_kernel = p->sum_out && p->sum_parts == e->pack_blocks;
  return p->fused_bs && p->jc69_v2 && !p->pd.dbg && (!p->sum_out || sum_in_kernel) && p->has_lnl && p->pack_epoch == e->pack_epoch;
}

// plans[0..count) as ONE launch
static int chain_launch(bpa_plan * const * plans, unsigned count)
{
  bpa_engine * e = plans[0]->eng;
  ChainDev c{};
  c.base = plans[0]->pd;
  c.base.loci = e->d_loci.p; c.base.bfbeta = e->bfbeta;
  c.base.lane_tab = e->d_lane_tab.p; c.base.slot_tab = e->d_slot_tab.p; c.base.blk_slot_off = e->d_blk_slot_off.p; c.base.nblocks2 = e->pack_blocks;
  c.nsteps = count;
  double bytes = 0, bytes_codes = 0;
  for (unsigned i = 0; i < count; ++i)
  {
    const bpa_plan * p = plans[i];
    bytes_codes += p->bytes_codes + p->bytes_pmatrix;
    ChainStep & st = c.st[i];
    st.recs2 = p->pd.recs2; st.mat2 = p->pd.mat2; st.mat_length = p->pd.mat_length; st.blk_mat_off = p->pd.blk_mat_off;
    st.site_term = p->pd.site_term; st.lnl = p->pd.lnl; st.wg_part = p->sum_out; st.rec2_units = p->pd.rec2_units;
    st.flags = (p->has_mats ? 1u : 0u) | 2u | 4u | (p->sum_out ? 8u : 0u);
    bytes += p->bytes_partials + p->bytes_pmatrix;
  }
  TimingSlot * ts = nullptr;
  if (e->timing && (e->timing_phase++ % e->timing_stride) == 0)
  {
    ts = next_slot(e);
    if (!ts) { if (!timing_drain(e)) return 0; ts = next_slot(e); }
  }
  hipEvent_t k0 = ts ? ts->ev[1] : nullptr, k1 = ts ? ts->ev[2] : nullptr;
  hipExtLaunchKernelGGL((step_jc69_v2_chain_kernel<PACK_BS>), dim3(e->pack_blocks), dim3(PACK_BS), 0, e->stream, k0, k1, 0, c);
  HIPCHK(hipGetLastError());
  if (ts) { ts->ev_used = 2; ts->steps = count; ts->bytes = bytes; ts->bytes_codes = bytes_codes; }
  return 1;
}

extern "C" int bpa_plans_launch(bpa_plan_t * const * plans, unsigned count)
{
  if (!count) return 1;
  bpa_engine * e = plans[0]->eng;
  std::lock_guard<std::recursive_mutex> lock_(e->mtx);
  static const bool no_chain = BPA_EXP_SWITCH("BPA_NO_CHAIN") != nullptr;
  if (!e->usedata) return 1;
  unsigned i = 0;
  while (i < count)
  {
    // the longest run of consecutive chainable plans of this engine
    unsigned j = i;
    if (!no_chain && plans[i]->eng == e && flush(e) && engine_pack(e))
      while (j < count && j - i < (unsigned)BPA_CHAIN_MAX && plans[j]->eng == e && chainable(plans[j])) ++j;
    if (j - i >= 2)
    {
      if (!set_device(e) || !chain_launch(plans + i, j - i)) return 0;
      i = j;
      continue;
    }
    if (!bpa_plan_launch(plans[i])) return 0;
    ++i;
  }
  return 1;
}

extern "C" int bpa_plan_get_lnl(bpa_plan_t * p, double * lnl)
{
  std::lock_guard<std::recursive_mutex> lock_(p->eng->mtx);
  bpa_engine * e = p->eng;
  if (!set_device(e)) return 0;
  if (!e->usedata) { std::fill(lnl, lnl + p->pd.ntasks, 0.0); return 1; }
  const size_t nb = p->pd.ntasks*sizeof(double);
  if (nb > e->h_stage_bytes)
  {
    if (e->h_stage) (void)hipHostFree(e->h_stage);
    e->h_stage = nullptr; e->h_stage_bytes = 0;
    const size_t want = std::max<size_t>(nb, 1u << 16);
    if (hipHostMalloc(&e->h_stage, want, hipHostMallocDefault) == hipSuccess) e->h_stage_bytes = want;
    else e->h_stage = nullptr;
  }
  if (e->h_stage)
  {
    HIPCHK(hipMemcpyAsync(e->h_stage, p->lnl.p, nb, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    std::memcpy(lnl, e->h_stage, nb);
    return 1;
  }
  HIPCHK(hipMemcpyAsync(lnl, p->lnl.p, nb, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  return 1;
}

extern "C" void * bpa_plan_lnl_device(bpa_plan_t * p) { return p->lnl.p; }

static int plan_set_params(bpa_plan * p, int which, const double * host_values, const double * dev_values)
{
  bpa_engine * e = p->eng;
  std::lock_guard<std::recursive_mutex> lock_(e->mtx);
  if (!set_device(e)) return 0;
  if (which != 1 && which != 2 && which != 4) return fail("bpa_plan_set_params: which = 1 (frequencies), 2 (substitution parameters) or 4 (category rates)");
  const unsigned T = p->pd.ntasks;
  unsigned len = 0;
  for (unsigned t = 0; t < T; ++t)
  {
    const bpa_locus * l = e->loci[p->h_locus[t]];
    const unsigned n = which == 1 ? l->states : which == 2 ? l->states*(l->states - 1)/2 : l->rate_cats;
    if (t && n != len) return fail("bpa_plan_set_params: the loci of the plan must agree in states / rate categories");
    len = n;
  }
  if (!flush(e)) return 0;                                  // pending per-locus setters go first
  if (host_values)
  {
    // the host copy stays the mirror of the device block (flush uploads from it)
    for (unsigned t = 0; t < T; ++t)
    {
      bpa_locus * l = e->loci[p->h_locus[t]];
      const unsigned S = l->states, R = l->rate_cats;
      const size_t off = which == 4 ? par_rates(R) : par_matrix(R, S, 0) + (which == 1 ? pm_freqs(S) : pm_subst(S));
      std::copy(host_values + (size_t)t*len, host_values + (size_t)(t + 1)*len, l->par.begin() + off);
    }
    if (!p->param_stage.reserve((size_t)T*len)) return fail("out of device memory (parameter stage)");
    HIPCHK(hipMemcpyAsync(p->param_stage.p, host_values, (size_t)T*len*sizeof(double), hipMemcpyHostToDevice, e->stream));
    dev_values = p->param_stage.p;
  }
  else for (unsigned t = 0; t < T; ++t) e->loci[p->h_locus[t]]->host_par_stale = true;      // the device block is ahead of the mirror
  // the packing's slot table carries a copy of a one-category locus's rate (SlotStatic::rate0, what the JC69 step kernels
  // build fresh P-matrices from): it has to follow
  if (which == 4 && len == 1) e->pack_dirty = true;
  {
    bool all4 = true, all20 = true;
    for (unsigned t = 0; t < T; ++t) { const unsigned S = e->loci[p->h_locus[t]]->states; all4 = all4 && S == 4; all20 = all20 && S == 20; }
    const dim3 grid((T + 63)/64), block(64);
    if (all4)       hipLaunchKernelGGL(params_install_kernel<4>,  grid, block, 0, e->stream, e->d_loci.p, p->task_locus.p, T, (uint32_t)which, dev_values, len);
    else if (all20) hipLaunchKernelGGL(params_install_kernel<20>, grid, block, 0, e->stream, e->d_loci.p, p->task_locus.p, T, (uint32_t)which, dev_values, len);
    else            hipLaunchKernelGGL(params_install_kernel<0>,  grid, block, 0, e->stream, e->d_loci.p, p->task_locus.p, T, (uint32_t)which, dev_values, len);
  }
  HIPCHK(hipGetLastError());
  return 1;
}

extern "C" int bpa_plan_set_params(bpa_plan_t * p, int which, const double * values)
{ return values ? plan_set_params(p, which, values, nullptr) : fail("bpa_plan_set_params: null values"); }
extern "C" int bpa_plan_set_params_device(bpa_plan_t * p, int which, const double * device_values)
{ return device_values ? plan_set_params(p, which, nullptr, device_values) : fail("bpa_plan_set_params_device: null values"); }

extern "C" int bpa_plan_enable_sum(bpa_plan_t * p, void * device_out)
{
  std::lock_guard<std::recursive_mutex> lock_(p->eng->mtx);
  if (!set_device(p->eng)) return 0;
  p->sum_parts = 0;
  if (device_out) { p->sum_out = (double *)device_out; return 1; }
  if (!p->lnl_sum.reserve(1)) return fail("out of device memory (plan)");
  p->sum_out = p->lnl_sum.p;
  return 1;
}

extern "C" int bpa_plan_enable_partial_sums(bpa_plan_t * p, void * device_out, unsigned * count)
{
  std::lock_guard<std::recursive_mutex> lock_(p->eng->mtx);
  bpa_engine * e = p->eng;
  if (!set_device(e) || !count) return fail("bpa_plan_enable_partial_sums: null argument");
  // one value per workgroup of the engine's packing when the plan runs on it, else the total alone
  unsigned want = 1;
  if ((p->jc69_v2 || p->klane_v2) && engine_pack(e) && p->pack_epoch == e->pack_epoch) want = e->pack_blocks;
  if (device_out && *count < want) want = 1;
  if (device_out) p->sum_out = (double *)device_out;
  else
  {
    if (!p->lnl_sum.reserve(want)) return fail("out of device memory (plan)");
    p->sum_out = p->lnl_sum.p;
  }
  p->sum_parts = want;
  *count = want;
  return 1;
}

extern "C" int bpa_plan_get_sum(bpa_plan_t * p, double * sum)
{
  std::lock_guard<std::recursive_mutex> lock_(p->eng->mtx);
  bpa_engine * e = p->eng;
  if (!set_device(e)) return 0;
  if (!p->sum_out) return fail("bpa_plan_get_sum: call bpa_plan_enable_sum first");
  std::vector<double> parts(std::max(1u, p->sum_parts));
  HIPCHK(hipMemcpyAsync(parts.data(), p->sum_out, parts.size()*sizeof(double), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  double total = 0;
  for (double v : parts) total += v;
  *sum = total;
  return 1;
}

extern "C" int bpa_plan_work_codes(bpa_plan_t * p, double * bytes_codes)
{
  if (!p) return fail("plan_work_codes: null plan");
  if (bytes_codes) *bytes_codes = p->bytes_codes;
  return 1;
}

extern "C" int bpa_plan_work(bpa_plan_t * p, double * bytes_partials, double * flops_partials,
                             double * bytes_pmatrix, unsigned long * node_updates,
                             unsigned long * pattern_updates)
{
  if (bytes_partials) *bytes_partials = p->bytes_partials;
  if (flops_partials) *flops_partials = p->flops_partials;
  if (bytes_pmatrix) *bytes_pmatrix = p->bytes_pmatrix;
  if (node_updates) *node_updates = p->node_updates;
  if (pattern_updates) *pattern_updates = p->pattern_updates;
  return 1;
}

// One proposal step evaluated and forgotten (what a host-resident MCMC does after every proposal): when the batch
// runs on the engine's packing, its compact records are written straight into ONE pinned image (records | fresh-matrix
// list | branch lengths | per-workgroup matrix offsets), uploaded with ONE copy into persistent device memory and
// launched — no plan object, no device allocation, no per-array copies (a plan costs ~2.3 ms to build for 10 000
// loci, this ~0.1 ms).  handled = false: not eligible, the caller takes the general path.
// bpa_batch_evaluate's packed path in three parts (bpa_batch_begin / _fill / _end of the header): checks + sizing, the
// record image of a range of the batch's loci (disjoint ranges from several threads at once: a locus's records go to its own
// slot and its own range of the matrix list, nothing else is written), upload + launch + results.
static double batch_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static int batch_begin_packed(bpa_engine * e, const bpa_batch_t * b, bool & handled, bool fast_ok = true)
{
  handled = false;
  static const bool off = BPA_EXP_SWITCH("BPA_JC69_V1") != nullptr || BPA_EXP_SWITCH("BPA_NO_JC69_FAST") != nullptr;
  if (off || !b->root_clv || !b->nloci) return 1;
  if (!flush(e) || !engine_pack(e)) return 0;
  if (!e->pack_slots || e->timing) return 1;
  const unsigned T = b->nloci;
  // eligibility: every locus packed, in slot order, and all of one kind — JC69 with one rate category (step_jc69_v2_kernel)
  // or several categories without scalers / phase averaging (step_s4_klane_v2_kernel)
  int prev = -1;
  unsigned maxops = 0, npat = 0, rmax = 1;
  bool all_jc = true, all_kl = true;
  // a packing of one kind throughout: the image is sized from the packing's bounds and every per-locus check moves into
  // the (parallel) fill — no pass over the batch here (10 000 loci: 0.04 ms of a 0.3 ms host-driven step)
  e->bc_fast = fast_ok && e->pack_homog != 0 && e->pack_maxops <= 255 && T <= e->pack_slots;
  e->bc_fallback.store(0);
  if (e->bc_fast) { maxops = e->pack_maxops; npat = e->pack_npat; rmax = e->pack_rmax; all_jc = e->pack_homog == 1; all_kl = e->pack_homog == 2; }
  else e->bc_pat.resize((size_t)T + 1);
  for (unsigned t = 0; t < T && !e->bc_fast; ++t)
  {
    const bpa_locus * l = b->loci[t];
    if (!l || l->eng != e || !l->alive) return fail("plan: locus does not belong to this engine");
    const int sl = e->slot_of[l->id];
    if (sl <= prev) return 1;
    prev = sl;
    maxops = std::max(maxops, b->op_off ? b->op_off[t+1] - b->op_off[t] : 0u);
    e->bc_pat[t] = npat;
    npat += l->sites;
    rmax = std::max(rmax, l->rate_cats);
    all_jc = all_jc && l->rate_cats == 1 && l->dev.model == 0;
    all_kl = all_kl && l->rate_cats > 1 && l->scale_buffers == 0 && !l->dev.unphased_length && (!b->root_scaler || b->root_scaler[t] < 0);
  }
  if (!e->bc_fast) e->bc_pat[T] = npat;
  static const bool no_klane = BPA_EXP_SWITCH("BPA_KLANE_V1") != nullptr || BPA_EXP_SWITCH("BPA_NO_KLANE") != nullptr;
  if (!all_jc && (!all_kl || no_klane)) return 1;
  if (maxops > 255) return 1;                    // StepRec counts a locus's updates in a byte
  const unsigned nmat = b->mat_off ? b->mat_off[T] : 0;
  const unsigned units = 1 + std::max(maxops, 3u);
  const size_t n_recs = (size_t)e->pack_slots*units*16;
  const size_t o_mat2 = n_recs, n_mat2 = ((size_t)nmat*sizeof(MatRec2) + 15) & ~(size_t)15;
  const size_t o_len = o_mat2 + n_mat2, n_len = ((size_t)nmat*sizeof(double) + 15) & ~(size_t)15;
  const size_t o_bm = o_len + n_len, n_bm = ((size_t)(e->pack_blocks + 1)*4 + 15) & ~(size_t)15;
  const size_t total = o_bm + n_bm;
  if (total > e->h_step_bytes)
  {
    if (e->h_step) (void)hipHostFree(e->h_step);
    e->h_step = nullptr; e->h_step_bytes = 0;
    const size_t want = total + total/4;
    if (hipHostMalloc(&e->h_step, want, hipHostMallocDefault) != hipSuccess) { e->h_step = nullptr; return fail("out of pinned host memory (step image)"); }
    e->h_step_bytes = want;
  }
  if (!e->d_step.reserve(e->h_step_bytes) || !e->d_step_terms.reserve(npat) || !e->d_step_lnl.reserve(T))
    return fail("out of device memory (step image)");
  // the previous step's image may still be in flight only if the caller did not read its results: drain
  HIPCHK(hipStreamSynchronize(e->stream));
  e->bc_T = T; e->bc_units = units; e->bc_nmat = nmat; e->bc_npat = npat; e->bc_rmax = rmax; e->bc_alljc = all_jc;
  e->bc_o_mat2 = o_mat2; e->bc_o_len = o_len; e->bc_o_bm = o_bm; e->bc_total = total;
  e->bc_failed.store(0);
  handled = true;
  return 1;
}

static int batch_fill_packed(bpa_engine * e, const bpa_batch_t * b, unsigned t0, unsigned t1)
{
  const unsigned T = e->bc_T, units = e->bc_units;
  if (t1 > T) t1 = T;
  if (t0 >= t1) return 1;
  unsigned char * img = (unsigned char *)e->h_step;
  uint4 * recs = reinterpret_cast<uint4 *>(img);
  MatRec2 * m2 = reinterpret_cast<MatRec2 *>(img + e->bc_o_mat2);
  double * len = reinterpret_cast<double *>(img + e->bc_o_len);
  // the slots from this range's first locus up to the next range's first: cleared, marked "not part of the step"
  {
    const bpa_locus * la = b->loci[t0], * lb = t1 < T ? b->loci[t1] : nullptr;
    if (e->bc_fast && (!la || la->eng != e || e->slot_of[la->id] < 0 || (lb && (lb->eng != e || e->slot_of[lb->id] < 0)))) { e->bc_fallback.store(1); return 1; }
    const unsigned s_lo = t0 == 0 ? 0u : (unsigned)e->slot_of[la->id], s_hi = t1 == T ? e->pack_slots : (unsigned)e->slot_of[lb->id];
    if (s_hi < s_lo) { e->bc_fallback.store(1); return 1; }
    std::memset(recs + (size_t)s_lo*units, 0, (size_t)(s_hi - s_lo)*units*16);
    for (unsigned sl = s_lo; sl < s_hi; ++sl) reinterpret_cast<StepRec *>(recs + (size_t)sl*units)->task = 0xffffffffu;
  }
  auto bad = [&](const char * msg) { std::lock_guard<std::mutex> g(e->bc_mtx); e->bc_failed.store(1); e->bc_msg = msg; return 0; };
  // workgroups up to a locus's start their matrix range at that locus's first entry: this range writes the entries of the
  // workgroups whose first slot lies past the slot of the locus before it, up to its own last locus (the last range: all the rest)
  uint32_t * bm = reinterpret_cast<uint32_t *>(img + e->bc_o_bm);
  unsigned blk = 0;
  if (t0)
  {
    const bpa_locus * lp = b->loci[t0 - 1];
    if (!lp || lp->eng != e || lp->id >= e->slot_of.size() || e->slot_of[lp->id] < 0) { e->bc_fallback.store(1); return 1; }
    blk = (unsigned)(std::upper_bound(e->h_blk_slot_off.begin(), e->h_blk_slot_off.end(), (uint32_t)e->slot_of[lp->id]) - e->h_blk_slot_off.begin());
  }
  for (unsigned t = t0; t < t1; ++t)
  {
    const bpa_locus * l = b->loci[t];
    if (e->bc_fast)
    {
      // what begin's pass checks otherwise: the locus is this engine's, packed, and the batch runs in slot order
      if (!l || l->eng != e || !l->alive) return bad("plan: locus does not belong to this engine");
      const int sl_ = e->slot_of[l->id];
      const int prev_ = t ? (b->loci[t-1] && b->loci[t-1]->eng == e ? e->slot_of[b->loci[t-1]->id] : -2) : -1;
      if (sl_ < 0 || sl_ <= prev_ || (e->pack_homog == 2 && b->root_scaler && b->root_scaler[t] >= 0) ||
          (b->op_off && b->op_off[t+1] - b->op_off[t] > e->pack_maxops)) { e->bc_fallback.store(1); return 1; }
    }
    const unsigned sl = (unsigned)e->slot_of[l->id];
    const unsigned o0 = b->op_off ? b->op_off[t] : 0, o1 = b->op_off ? b->op_off[t+1] : 0;
    const unsigned m0 = b->mat_off ? b->mat_off[t] : 0, m1 = b->mat_off ? b->mat_off[t+1] : 0;
    while (blk <= e->pack_blocks && e->h_blk_slot_off[blk] <= sl) bm[blk++] = m0;
    if (b->root_clv[t] < l->tips || b->root_clv[t] >= l->tips + l->clv_buffers) return bad("plan: root clv index out of range");
    if (b->root_scaler && b->root_scaler[t] >= (int)l->scale_buffers) return bad("plan: root scaler index out of range");
    for (unsigned i = m0; i < m1; ++i)
    {
      if (b->mat_pmatrix[i] >= l->prob_matrices) return bad("plan: pmatrix index out of range");
      if (!(b->mat_length[i] >= 0)) return bad("plan: negative branch length");   // assert(t >= 0), core_pmatrix.c:723
      for (unsigned j = m0; j < i; ++j)
        if (b->mat_pmatrix[j] == b->mat_pmatrix[i]) return bad("plan: a P-matrix buffer is listed twice for one locus");
      m2[i] = MatRec2{sl, b->mat_pmatrix[i]};
      len[i] = b->mat_length[i];
    }
    StepRec h{};
    h.task = t; h.pat_off = e->bc_fast ? e->pack_slot_pat[sl] : e->bc_pat[t]; h.root_clv = (uint8_t)b->root_clv[t];
    h.root_scaler = (int8_t)(b->root_scaler ? b->root_scaler[t] : BPA_SCALE_BUFFER_NONE); h.nops = (uint8_t)(o1 - o0);
    std::memcpy(recs + (size_t)sl*units, &h, sizeof(h));
    for (unsigned o = o0; o < o1; ++o)
    {
      const bpa_op_t & s = b->ops[o];
      if (!validate_op_quiet(l, s)) return bad("plan: a node update's buffer index is out of range");
      StepOp q{};
      q.parent_clv = (uint8_t)s.parent_clv; q.left_clv = (uint8_t)s.left_clv; q.right_clv = (uint8_t)s.right_clv;
      q.left_pmatrix = (uint8_t)s.left_pmatrix; q.right_pmatrix = (uint8_t)s.right_pmatrix;
      q.parent_scaler = (int8_t)s.parent_scaler; q.left_scaler = (int8_t)s.left_scaler; q.right_scaler = (int8_t)s.right_scaler;
      q.left_e = q.right_e = -1;
      for (unsigned i = m0; i < m1; ++i)
      {
        if (b->mat_pmatrix[i] == s.left_pmatrix)  q.left_e = (int32_t)i;
        if (b->mat_pmatrix[i] == s.right_pmatrix) q.right_e = (int32_t)i;
      }
      std::memcpy(recs + (size_t)sl*units + 1 + (o - o0), &q, sizeof(q));
    }
  }
  if (t1 == T) while (blk <= e->pack_blocks) bm[blk++] = e->bc_nmat;
  return 1;
}

static int batch_end_packed(bpa_engine * e, const bpa_batch_t * b, double * lnl, bool async = false)
{
  if (e->bc_failed.load()) return fail(e->bc_msg.c_str());
  if (e->bc_fallback.load()) return 2;             // not the one-image path after all: the caller evaluates the batch the general way
  const unsigned T = e->bc_T, nmat = e->bc_nmat, npat = e->bc_npat, rmax = e->bc_rmax, units = e->bc_units;
  unsigned char * img = (unsigned char *)e->h_step;
  // (the workgroups' matrix ranges — blk_mat_off — were written by the fills, each for the workgroups that start in its loci's slots)
  HIPCHK(hipMemcpyAsync(e->d_step.p, img, e->bc_total, hipMemcpyHostToDevice, e->stream));
  PlanDev d{};
  d.loci = e->d_loci.p; d.bfbeta = e->bfbeta;
  d.site_term = e->d_step_terms.p; d.lnl = e->d_step_lnl.p; d.ntasks = T; d.npatterns = npat; d.nmat = nmat;
  d.recs2 = reinterpret_cast<const uint4 *>(e->d_step.p);
  d.mat2 = reinterpret_cast<const MatRec2 *>(e->d_step.p + e->bc_o_mat2);
  d.mat_length = reinterpret_cast<const double *>(e->d_step.p + e->bc_o_len);
  d.blk_mat_off = reinterpret_cast<const uint32_t *>(e->d_step.p + e->bc_o_bm);
  d.rec2_units = units;
  d.lane_tab = e->d_lane_tab.p; d.slot_tab = e->d_slot_tab.p; d.blk_slot_off = e->d_blk_slot_off.p; d.nblocks2 = e->pack_blocks;
  if (e->bc_alljc)
  {
    d.flags = (nmat ? 1u : 0u) | 2u | 4u;
    hipLaunchKernelGGL((step_jc69_v2_kernel<PACK_BS>), dim3(e->pack_blocks), dim3(PACK_BS), 0, e->stream, d);
  }
  else
  {
    d.pad = rmax;
    if (nmat)
    {
      d.flags = 1u;
      hipLaunchKernelGGL(pmatrix_s4_dense_kernel, dim3((nmat*rmax + 255u)/256u), dim3(256), 0, e->stream, d, (uint32_t)nmat);
    }
    d.flags = 2u | 4u;
    launch_klane<false>(dim3(e->pack_blocks), e->stream, nullptr, nullptr, d);
  }
  HIPCHK(hipGetLastError());
  if (async) e->bc_wait_T = 0;
  if (!lnl && !async) { HIPCHK(hipStreamSynchronize(e->stream)); return 1; }
  if (!e->usedata) { HIPCHK(hipStreamSynchronize(e->stream)); if (async) { e->bc_wait_T = T; e->bc_wait_zero = true; } else std::fill(lnl, lnl + T, 0.0); return 1; }
  e->bc_wait_zero = false;
  const size_t nb = (size_t)T*sizeof(double);
  if (nb > e->h_stage_bytes)
  {
    if (e->h_stage) (void)hipHostFree(e->h_stage);
    e->h_stage = nullptr; e->h_stage_bytes = 0;
    const size_t want = std::max<size_t>(nb, 1u << 16);
    if (hipHostMalloc(&e->h_stage, want, hipHostMallocDefault) != hipSuccess) { e->h_stage = nullptr; return fail("out of pinned host memory"); }
    e->h_stage_bytes = want;
  }
  HIPCHK(hipMemcpyAsync(e->h_stage, e->d_step_lnl.p, nb, hipMemcpyDeviceToHost, e->stream));
  if (async) { e->bc_wait_T = T; return 1; }       // bpa_batch_wait collects
  HIPCHK(hipStreamSynchronize(e->stream));
  std::memcpy(lnl, e->h_stage, nb);
  return 1;
}

static int batch_evaluate_packed(bpa_engine * e, const bpa_batch_t * b, double * lnl, bool & handled)
{
  static const bool prof = BPA_EXP_SWITCH("BPA_PLAN_PROF") != nullptr;       // section timers (stderr, every 130 calls)
  static double pt[3] = {0, 0, 0}; static unsigned pcalls = 0;
  std::lock_guard<std::recursive_mutex> lock_(e->mtx);
  const double t_0 = prof ? batch_now() : 0;
  if (!batch_begin_packed(e, b, handled)) return 0;
  if (!handled) return 1;
  const double t_1 = prof ? batch_now() : 0;
  (void)batch_fill_packed(e, b, 0, e->bc_T);
  const double t_2 = prof ? batch_now() : 0;
  {
    const int r = batch_end_packed(e, b, lnl);
    if (!r) return 0;
    if (r == 2)
    {
      // (a batch the packing's bounds did not describe: the full pass decides — and may still take the one-image path)
      if (!batch_begin_packed(e, b, handled, false)) return 0;
      if (!handled) return 1;
      (void)batch_fill_packed(e, b, 0, e->bc_T);
      if (!batch_end_packed(e, b, lnl)) return 0;
    }
  }
  if (prof)
  {
    const double t_3 = batch_now();
    pt[0] += t_1 - t_0; pt[1] += t_2 - t_1; pt[2] += t_3 - t_2;
    if (++pcalls % 130 == 0)
    {
      fprintf(stderr, "[bpa] batch_evaluate per call: checks %.3f ms, record image %.3f ms, upload + launch + lnL back %.3f ms\n",
              1e3*pt[0]/130, 1e3*pt[1]/130, 1e3*pt[2]/130);
      pt[0] = pt[1] = pt[2] = 0;
    }
  }
  return 1;
}

extern "C" int bpa_batch_begin(bpa_engine_t * e, const bpa_batch_t * b)
{
  if (!e || !b) return fail("bpa_batch_begin: null argument");
  std::lock_guard<std::recursive_mutex> lock_(e->mtx);
  bool handled = false;
  if (!batch_begin_packed(e, b, handled)) return 0;
  return handled ? 1 : 2;
}
extern "C" int bpa_batch_fill(bpa_engine_t * e, const bpa_batch_t * b, unsigned t0, unsigned t1) { return batch_fill_packed(e, b, t0, t1); }
extern "C" int bpa_batch_end(bpa_engine_t * e, const bpa_batch_t * b, double * lnl)
{
  std::lock_guard<std::recursive_mutex> lock_(e->mtx);
  return batch_end_packed(e, b, lnl);
}

extern "C" int bpa_batch_end_async(bpa_engine_t * e, const bpa_batch_t * b)
{
  std::lock_guard<std::recursive_mutex> lock_(e->mtx);
  return batch_end_packed(e, b, nullptr, true);
}
extern "C" int bpa_batch_wait(bpa_engine_t * e, double * lnl)
{
  if (!e || !lnl) return fail("bpa_batch_wait: null argument");
  std::lock_guard<std::recursive_mutex> lock_(e->mtx);
  if (!set_device(e)) return 0;
  HIPCHK(hipStreamSynchronize(e->stream));
  const unsigned T = e->bc_wait_T;
  e->bc_wait_T = 0;
  if (e->bc_wait_zero) std::fill(lnl, lnl + T, 0.0);
  else if (T) std::memcpy(lnl, e->h_stage, (size_t)T*sizeof(double));
  return 1;
}

extern "C" int bpa_batch_evaluate(bpa_engine_t * e, const bpa_batch_t * b, double * lnl)
{
  if (!e || !b) return fail("bpa_batch_evaluate: null argument");
  bool handled = false;
  if (!batch_evaluate_packed(e, b, lnl, handled)) return 0;
  if (handled) return 1;
  // general path: a transient plan.  (Tried: ONE plan object per engine rebuilt in place so that its ~25 device buffers
  // are allocated once — the blocking pageable uploads into buffers the previous step's kernels had just used then
  // stalled the following launch for 17-27 ms, ten times slower overall: the one-image scheme above is the way.)
  bpa_plan * p = bpa_plan_create(e, b);
  if (!p) return 0;
  int ok = bpa_plan_launch(p) && (lnl ? bpa_plan_get_lnl(p, lnl) : bpa_engine_synchronize(e));
  bpa_plan_destroy(p);
  return ok;
}

extern "C" void bpa_engine_enable_timing(bpa_engine_t * e, int on)
{
  std::lock_guard<std::recursive_mutex> lock_(e->mtx);
  (void)hipSetDevice(e->device);
  (void)timing_drain(e);
  e->timing = on != 0;
  e->acc_ms[0] = e->acc_ms[1] = e->acc_ms[2] = 0; e->acc_launches = 0; e->acc_steps = 0; e->acc_bytes = 0; e->acc_bytes_codes = 0;
}

extern "C" void bpa_engine_set_timing_stride(bpa_engine_t * e, unsigned stride)
{
  std::lock_guard<std::recursive_mutex> lock_(e->mtx);
  e->timing_stride = stride ? stride : 1; e->timing_phase = 0;
}

// what the timed launches covered: proposal steps (a chain launch covers several) and the algorithmic bytes (SURVEY.md
// section 8d) of the kernels whose time bpa_engine_timing reports as partials_ms — roofline.achieved = bytes / partials_ms
extern "C" int bpa_engine_timing_work(bpa_engine_t * e, unsigned long * steps, double * bytes)
{
  std::lock_guard<std::recursive_mutex> lock_(e->mtx);
  if (!set_device(e)) return 0;
  if (!timing_drain(e)) return 0;
  if (steps) *steps = e->acc_steps;
  if (bytes) *bytes = e->acc_bytes;
  return 1;
}

extern "C" int bpa_engine_timing_work_codes(bpa_engine_t * e, double * bytes_codes)
{
  std::lock_guard<std::recursive_mutex> lock_(e->mtx);
  if (!set_device(e)) return 0;
  if (!timing_drain(e)) return 0;
  if (bytes_codes) *bytes_codes = e->acc_bytes_codes;
  return 1;
}

extern "C" int bpa_engine_timing(bpa_engine_t * e, double * pmatrix_ms, double * partials_ms,
                                 double * reduce_ms, unsigned long * launches)
{
  std::lock_guard<std::recursive_mutex> lock_(e->mtx);
  if (!set_device(e)) return 0;
  if (!timing_drain(e)) return 0;
  if (pmatrix_ms) *pmatrix_ms = e->acc_ms[0];
  if (partials_ms) *partials_ms = e->acc_ms[1];
  if (reduce_ms) *reduce_ms = e->acc_ms[2];
  if (launches) *launches = e->acc_launches;
  return 1;
}

// ------------------------------------------------- single-locus update API ---
// The three calls are lazy: a proposal of the reference is locus_update_matrices -> locus_update_partials ->
// locus_root_loglikelihood on ONE locus (gtree.c:5447-5467, 7484-7566; stree.c:4727-4749), and only the last one
// returns anything.  The first two queue their work on the locus; the third runs the whole step as one launch
// (through the one-image path of bpa_batch_evaluate when the locus sits on the engine's packing, a scratch plan
// otherwise) and reads the value back — one launch and one synchronisation per proposal instead of three.
static int scratch_run(bpa_locus * l, const unsigned * pm_idx, const double * lens, unsigned nmat,
                       const bpa_op_t * ops, unsigned nops, bool lnl, unsigned root_clv, int root_scaler)
{
  unsigned mat_off[2] = {0, nmat}, op_off[2] = {0, nops};
  bpa_locus * arr[1] = {l};
  bpa_batch_t b{};
  b.nloci = 1; b.loci = arr;
  b.mat_off = mat_off; b.mat_pmatrix = pm_idx; b.mat_length = lens;
  b.op_off = op_off; b.ops = ops;
  b.root_clv = lnl ? &root_clv : nullptr; b.root_scaler = lnl ? &root_scaler : nullptr;
  if (!l->scratch) l->scratch.reset(new bpa_plan());
  // the scratch plan's buffers may still be read by the previous call's kernels
  HIPCHK(hipStreamSynchronize(l->eng->stream));
  if (!plan_build(l->scratch.get(), l->eng, &b)) return 0;
  const int mode = (nmat ? 1 : 0) | ((nops || lnl) ? 2 : 0) | (lnl ? 4 : 0);
  return plan_launch_mode(l->scratch.get(), mode);
}

static void queue_locus(bpa_locus * l)
{
  if (!l->pending) { l->pending = true; l->eng->pending.push_back(l); }
}

// a queue whose run failed goes back on the locus (in front of anything queued since): a retry then computes over updated
// buffers instead of returning a value over buffers that never saw the updates
static int requeue(bpa_locus * l, std::vector<unsigned> & pm, std::vector<double> & len, std::vector<bpa_op_t> & ops)
{
  pm.insert(pm.end(), l->pend_pm.begin(), l->pend_pm.end());    l->pend_pm.swap(pm);
  len.insert(len.end(), l->pend_len.begin(), l->pend_len.end()); l->pend_len.swap(len);
  ops.insert(ops.end(), l->pend_ops.begin(), l->pend_ops.end()); l->pend_ops.swap(ops);
  if (!l->pending) { l->pending = true; l->eng->pending.push_back(l); }
  return 0;
}

// run what is queued on l (and, with want_lnl, the root term at root_clv / root_scaler: *lnl receives it).
// persite: the caller wants the per-pattern terms, which only the scratch plan keeps.
static int run_queued(bpa_locus * l, bool want_lnl, unsigned root_clv, int root_scaler, double * lnl, bool persite)
{
  bpa_engine * e = l->eng;
  // take the queue first: the paths below flush the engine, which must not see this locus as pending again
  std::vector<unsigned> pm; std::vector<double> len; std::vector<bpa_op_t> ops;
  pm.swap(l->pend_pm); len.swap(l->pend_len); ops.swap(l->pend_ops);
  l->pending = false;
  const unsigned nmat = (unsigned)pm.size(), nops = (unsigned)ops.size();
  if (!nmat && !nops && !want_lnl) return 1;
  if (want_lnl && !persite)
  {
    unsigned mat_off[2] = {0, nmat}, op_off[2] = {0, nops};
    bpa_locus * arr[1] = {l};
    bpa_batch_t b{};
    b.nloci = 1; b.loci = arr;
    b.mat_off = mat_off; b.mat_pmatrix = pm.data(); b.mat_length = len.data();
    b.op_off = op_off; b.ops = ops.data();
    b.root_clv = &root_clv; b.root_scaler = &root_scaler;
    bool handled = false;
    if (!batch_evaluate_packed(e, &b, lnl, handled)) return requeue(l, pm, len, ops);
    if (handled) return 1;
  }
  if (!scratch_run(l, pm.data(), len.data(), nmat, ops.data(), nops, want_lnl, root_clv, root_scaler)) return requeue(l, pm, len, ops);
  if (want_lnl && !bpa_plan_get_lnl(l->scratch.get(), lnl)) return 0;
  return 1;
}

// everything queued on any locus of the engine (called by flush: before a plan launch, a buffer access, ...)
static int run_all_queued(bpa_engine * e)
{
  while (!e->pending.empty())
  {
    std::vector<bpa_locus *> todo;
    todo.swap(e->pending);
    for (bpa_locus * l : todo)
      if (l->pending && l->alive && !run_queued(l, false, 0, BPA_SCALE_BUFFER_NONE, nullptr, false)) return 0;
  }
  return 1;
}

extern "C" int bpa_locus_update_matrices(bpa_locus_t * l, const unsigned * pmatrix_indices,
                                         const double * branch_lengths, unsigned count)
{
  if (!l) return fail("bpa_locus_update_matrices: null locus");
  std::lock_guard<std::recursive_mutex> lock_(l->eng->mtx);
  if (!l->eng->usedata || !count) return 1;
  for (unsigned i = 0; i < count; ++i)
  {
    if (pmatrix_indices[i] >= l->prob_matrices) return fail("plan: pmatrix index out of range");
    if (!(branch_lengths[i] >= 0)) return fail("plan: negative branch length");     // assert(t >= 0), core_pmatrix.c:723
    for (unsigned j = 0; j < i; ++j)
      if (pmatrix_indices[j] == pmatrix_indices[i]) return fail("plan: a P-matrix buffer is listed twice for one locus");
  }
  // matrices after node updates: the queued step is complete, a new one starts
  if (!l->pend_ops.empty() && !run_queued(l, false, 0, BPA_SCALE_BUFFER_NONE, nullptr, false)) return 0;
  for (unsigned i = 0; i < count; ++i)
  {
    size_t j = 0;
    while (j < l->pend_pm.size() && l->pend_pm[j] != pmatrix_indices[i]) ++j;
    if (j < l->pend_pm.size()) l->pend_len[j] = branch_lengths[i];                 // a later call overwrites the buffer
    else { l->pend_pm.push_back(pmatrix_indices[i]); l->pend_len.push_back(branch_lengths[i]); }
  }
  queue_locus(l);
  return 1;
}

extern "C" int bpa_locus_update_partials(bpa_locus_t * l, const bpa_op_t * ops, unsigned count)
{
  if (!l) return fail("bpa_locus_update_partials: null locus");
  std::lock_guard<std::recursive_mutex> lock_(l->eng->mtx);
  if (!l->eng->usedata || !count) return 1;
  for (unsigned i = 0; i < count; ++i)
    if (!validate_op(l, ops[i])) return 0;
  if (l->pend_ops.size() + count > 255 && !run_queued(l, false, 0, BPA_SCALE_BUFFER_NONE, nullptr, false)) return 0;
  l->pend_ops.insert(l->pend_ops.end(), ops, ops + count);
  queue_locus(l);
  return 1;
}

extern "C" double bpa_locus_root_loglikelihood(bpa_locus_t * l, unsigned root_clv, int root_scaler,
                                               const unsigned * freqs_indices, double * persite_lnl)
{
  if (!l) { fail("bpa_locus_root_loglikelihood: null locus"); return NAN; }
  std::lock_guard<std::recursive_mutex> lock_(l->eng->mtx);
  bpa_engine * e = l->eng;
  if (!e->usedata) return 0.0;
  if (freqs_indices)
    for (unsigned k = 0; k < l->rate_cats; ++k)
      if ((double)freqs_indices[k] != l->par[par_param_idx(l->rate_cats) + k])
      { fail("bpa_locus_root_loglikelihood: freqs_indices must equal the locus's param_indices"); return NAN; }
  if (root_clv < l->tips || root_clv >= l->tips + l->clv_buffers) { fail("plan: root clv index out of range"); return NAN; }
  if (root_scaler >= (int)l->scale_buffers) { fail("plan: root scaler index out of range"); return NAN; }
  double v = NAN;
  if (!run_queued(l, true, root_clv, root_scaler, &v, persite_lnl != nullptr)) return NAN;
  if (persite_lnl)
  {
    if (hipMemcpy(persite_lnl, l->scratch->site_term.p, l->sites*sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
    { fail("persite copy failed"); return NAN; }
  }
  return v;
}

// locus_update_all_matrices (locus.c:1922) / locus_update_all_partials (locus.c:2523) over a flat view of the gene tree
static int check_view(const bpa_locus * l, const bpa_gtree_view_t * g)
{
  if (!g || !g->left || !g->right || !g->parent || !g->clv_index || !g->pmatrix_index || !g->scaler_index)
    return fail("gene-tree view: null array");
  if (g->nodes != 2*l->tips - 1) return fail("gene-tree view: node count must be 2*tips - 1");
  if (g->root < 0 || (unsigned)g->root >= g->nodes || g->left[g->root] < 0) return fail("gene-tree view: bad root");
  for (unsigned i = 0; i < g->nodes; ++i)
  {
    const int a = g->left[i], b = g->right[i];
    if ((a < 0) != (b < 0)) return fail("gene-tree view: a node has one child");
    if (a >= (int)g->nodes || b >= (int)g->nodes || g->parent[i] >= (int)g->nodes) return fail("gene-tree view: node index out of range");
    if (a >= 0 && (g->parent[a] != (int)i || g->parent[b] != (int)i)) return fail("gene-tree view: parent / child fields disagree");
  }
  return 1;
}

extern "C" int bpa_locus_update_all_matrices(bpa_locus_t * l, const bpa_gtree_view_t * g, double * lengths_out)
{
  if (!l) return fail("bpa_locus_update_all_matrices: null locus");
  if (!g || !g->time) return fail("gene-tree view: null array");
  if (!check_view(l, g)) return 0;
  std::vector<unsigned> idx; std::vector<double> len; std::vector<int> stack;
  // pre-order from root->left, then root->right (locus_update_all_matrices_jc69, locus.c:1905-1919)
  stack.push_back(g->right[g->root]); stack.push_back(g->left[g->root]);
  while (!stack.empty())
  {
    const int x = stack.back(); stack.pop_back();
    if (idx.size() >= g->nodes) return fail("gene-tree view: not a tree");
    const double t = (g->time[g->parent[x]] - g->time[x])*g->rate_mui;           // locus.c:1826
    idx.push_back(g->pmatrix_index[x]); len.push_back(t);
    if (lengths_out) lengths_out[x] = t;
    if (g->left[x] >= 0) { stack.push_back(g->right[x]); stack.push_back(g->left[x]); }
  }
  return bpa_locus_update_matrices(l, idx.data(), len.data(), (unsigned)idx.size());
}

extern "C" int bpa_locus_update_all_partials(bpa_locus_t * l, const bpa_gtree_view_t * g)
{
  if (!l) return fail("bpa_locus_update_all_partials: null locus");
  if (!check_view(l, g)) return 0;
  // post-order: left subtree, right subtree, node (locus_update_all_partials_recursive, locus.c:2482-2521)
  std::vector<bpa_op_t> ops;
  std::vector<std::pair<int, int>> stack{{g->root, 0}};
  while (!stack.empty())
  {
    const int x = stack.back().first, st = stack.back().second;
    stack.pop_back();
    if (g->left[x] < 0) continue;
    if (st == 0)
    {
      if (stack.size() > 4*(size_t)g->nodes) return fail("gene-tree view: not a tree");
      stack.push_back({x, 1}); stack.push_back({g->right[x], 0}); stack.push_back({g->left[x], 0});
      continue;
    }
    const int a = g->left[x], b = g->right[x];
    bpa_op_t o;
    o.parent_clv = g->clv_index[x]; o.parent_scaler = g->scaler_index[x];
    o.left_clv = g->clv_index[a];   o.left_pmatrix = g->pmatrix_index[a];  o.left_scaler = g->scaler_index[a];
    o.right_clv = g->clv_index[b];  o.right_pmatrix = g->pmatrix_index[b]; o.right_scaler = g->scaler_index[b];
    ops.push_back(o);
  }
  return bpa_locus_update_partials(l, ops.data(), (unsigned)ops.size());
}

// ---------------------------------------------- buffer access (ref. layouts) ---
static int sync_for_access(bpa_locus * l)
{
  if (!flush(l->eng)) return 0;
  HIPCHK(hipStreamSynchronize(l->eng->stream));
  return 1;
}

extern "C" int bpa_locus_get_clv(bpa_locus_t * l, unsigned idx, double * out)
{
  std::lock_guard<std::recursive_mutex> lock_(l->eng->mtx);
  if (!sync_for_access(l)) return 0;
  const size_t S = l->states, R = l->rate_cats, Np = l->sites;
  if (idx >= l->tips + l->clv_buffers) return fail("clv index out of range");
  if (idx < l->tips)
  {
    for (size_t n = 0; n < Np; ++n)
    {
      const uint32_t c = S == 4 ? l->tipcodes[idx*Np + n] : reinterpret_cast<const uint32_t *>(l->tipcodes.data())[idx*Np + n];
      for (size_t k = 0; k < R; ++k)
        for (size_t s = 0; s < S; ++s) out[(n*R + k)*S + s] = (double)((c >> s) & 1u);
    }
    return 1;
  }
  const size_t Ld = l->dev.ld;
  std::vector<double> tmp(R*Ld*S);
  HIPCHK(hipMemcpy(tmp.data(), l->dev.clv + (size_t)(idx - l->tips)*R*Ld*S, tmp.size()*8, hipMemcpyDeviceToHost));
  for (size_t n = 0; n < Np; ++n)
    for (size_t k = 0; k < R; ++k)
      for (size_t s = 0; s < S; ++s)
        out[(n*R + k)*S + s] = S == 4 ? tmp[(k*Np + n)*4 + s] : tmp[(k*S + s)*Ld + n];
  return 1;
}

extern "C" int bpa_locus_set_clv(bpa_locus_t * l, unsigned idx, const double * in)
{
  std::lock_guard<std::recursive_mutex> lock_(l->eng->mtx);
  if (!sync_for_access(l)) return 0;
  const size_t S = l->states, R = l->rate_cats, Np = l->sites;
  if (idx < l->tips || idx >= l->tips + l->clv_buffers) return fail("bpa_locus_set_clv: only inner buffers can be written (tips are state codes)");
  const size_t Ld = l->dev.ld;
  std::vector<double> tmp(R*Ld*S, 0.0);
  for (size_t n = 0; n < Np; ++n)
    for (size_t k = 0; k < R; ++k)
      for (size_t s = 0; s < S; ++s)
        (S == 4 ? tmp[(k*Np + n)*4 + s] : tmp[(k*S + s)*Ld + n]) = in[(n*R + k)*S + s];
  HIPCHK(hipMemcpy(l->dev.clv + (size_t)(idx - l->tips)*R*Ld*S, tmp.data(), tmp.size()*8, hipMemcpyHostToDevice));
  return 1;
}

extern "C" int bpa_locus_get_pmatrix(bpa_locus_t * l, unsigned idx, double * out)
{
  std::lock_guard<std::recursive_mutex> lock_(l->eng->mtx);
  if (!sync_for_access(l)) return 0;
  if (idx >= l->prob_matrices) return fail("pmatrix index out of range");
  const size_t R = l->rate_cats, S = l->states, ps = l->dev.pstride;
  std::vector<double> tmp(R*ps);
  HIPCHK(hipMemcpy(tmp.data(), l->dev.pmat + idx*R*ps, tmp.size()*8, hipMemcpyDeviceToHost));
  if (ps == 2)                                     // JC69: expand (a, b) to the reference's 4x4 layout
    for (size_t k = 0; k < R; ++k)
      for (size_t i = 0; i < 16; ++i) out[k*16 + i] = ((i >> 2) == (i & 3)) ? tmp[2*k] : tmp[2*k+1];
  else
    std::copy(tmp.begin(), tmp.end(), out);
  (void)S;
  return 1;
}

extern "C" int bpa_locus_set_pmatrix(bpa_locus_t * l, unsigned idx, const double * in)
{
  std::lock_guard<std::recursive_mutex> lock_(l->eng->mtx);
  if (!sync_for_access(l)) return 0;
  if (idx >= l->prob_matrices) return fail("pmatrix index out of range");
  const size_t R = l->rate_cats, ps = l->dev.pstride;
  std::vector<double> tmp(R*ps);
  if (ps == 2)
    for (size_t k = 0; k < R; ++k)
    {
      const double a = in[k*16], b = in[k*16 + 1];
      for (size_t i = 0; i < 16; ++i)
        if (in[k*16 + i] != (((i >> 2) == (i & 3)) ? a : b))
          return fail("bpa_locus_set_pmatrix: a JC69 locus only holds matrices of the form a on the diagonal, b elsewhere");
      tmp[2*k] = a; tmp[2*k+1] = b;
    }
  else
    std::copy(in, in + R*ps, tmp.begin());
  HIPCHK(hipMemcpy(l->dev.pmat + idx*R*ps, tmp.data(), tmp.size()*8, hipMemcpyHostToDevice));
  return 1;
}

extern "C" int bpa_locus_get_scaler(bpa_locus_t * l, unsigned idx, unsigned * out)
{
  std::lock_guard<std::recursive_mutex> lock_(l->eng->mtx);
  if (!sync_for_access(l)) return 0;
  if (idx >= l->scale_buffers) return fail("scaler index out of range");
  HIPCHK(hipMemcpy(out, l->dev.scaler + (size_t)idx*l->sites, l->sites*4, hipMemcpyDeviceToHost));
  return 1;
}

extern "C" int bpa_locus_set_scaler(bpa_locus_t * l, unsigned idx, const unsigned * in)
{
  std::lock_guard<std::recursive_mutex> lock_(l->eng->mtx);
  if (!sync_for_access(l)) return 0;
  if (idx >= l->scale_buffers) return fail("scaler index out of range");
  HIPCHK(hipMemcpy(l->dev.scaler + (size_t)idx*l->sites, in, l->sites*4, hipMemcpyHostToDevice));
  return 1;
}

extern "C" int bpa_locus_get_eigen(bpa_locus_t * l, unsigned index, double * evecs, double * ievecs, double * evals)
{
  std::lock_guard<std::recursive_mutex> lock_(l->eng->mtx);
  if (!sync_for_access(l)) return 0;
  if (index >= l->rate_matrices) return fail("rate matrix index out of range");
  const unsigned S = l->states, R = l->rate_cats;
  const double * pm = l->dev.par + par_matrix(R, S, index);
  HIPCHK(hipMemcpy(evals, pm + pm_evals(S), S*8, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(evecs, pm + pm_evecs(S), S*S*8, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(ievecs, pm + pm_ievecs(S), S*S*8, hipMemcpyDeviceToHost));
  return 1;
}

// ------------------------------------------------ library-form entry points ---
extern "C" int bpa_core_update_pmatrix(bpa_engine_t * e, double ** pmatrix, unsigned states,
                                       unsigned rate_cats, const double * rates,
                                       const double * branch_lengths,
                                       const unsigned * matrix_indices,
                                       const unsigned * param_indices,
                                       double * const * eigenvals, double * const * eigenvecs,
                                       double * const * inv_eigenvecs, unsigned count,
                                       unsigned attrib)
{
  std::lock_guard<std::recursive_mutex> lock_(e->mtx);
  (void)attrib;
  if (!set_device(e)) return 0;
  if (states != 4 && states != 20) return fail("bpa_core_update_pmatrix: 4 or 20 states");
  if (!count) return 1;
  const size_t S = states, R = rate_cats;
  unsigned nm = 0;
  for (unsigned k = 0; k < R; ++k) nm = std::max(nm, param_indices[k] + 1);
  for (unsigned i = 0; i < count; ++i)
    if (!(branch_lengths[i] >= 0)) return fail("bpa_core_update_pmatrix: negative branch length");
  std::vector<double> ev(nm*S), evec(nm*S*S), ievec(nm*S*S);
  for (unsigned m = 0; m < nm; ++m)
  {
    std::copy(eigenvals[m], eigenvals[m] + S, ev.begin() + m*S);
    std::copy(eigenvecs[m], eigenvecs[m] + S*S, evec.begin() + m*S*S);
    std::copy(inv_eigenvecs[m], inv_eigenvecs[m] + S*S, ievec.begin() + m*S*S);
  }
  DevBuf<double> d_out, d_rates, d_bl, d_ev, d_evec, d_ievec; DevBuf<uint32_t> d_pi;
  int ok = d_out.reserve(count*R*S*S) && upload(d_rates, rates, R) && upload(d_bl, branch_lengths, count)
        && upload(d_pi, param_indices, R) && upload(d_ev, ev.data(), ev.size())
        && upload(d_evec, evec.data(), evec.size()) && upload(d_ievec, ievec.data(), ievec.size());
  if (ok)
  {
    const unsigned n = (unsigned)(count*R*S);
    if (S == 4)
      hipLaunchKernelGGL(pmatrix_lib_kernel<4>, dim3((n + BPA_BLOCK - 1)/BPA_BLOCK), dim3(BPA_BLOCK), 0, e->stream,
                         d_out.p, count, (uint32_t)R, d_rates.p, d_bl.p, d_pi.p, d_ev.p, d_evec.p, d_ievec.p);
    else
      hipLaunchKernelGGL(pmatrix_lib_kernel<20>, dim3((n + BPA_BLOCK - 1)/BPA_BLOCK), dim3(BPA_BLOCK), 0, e->stream,
                         d_out.p, count, (uint32_t)R, d_rates.p, d_bl.p, d_pi.p, d_ev.p, d_evec.p, d_ievec.p);
    std::vector<double> out(count*R*S*S);
    ok = hipGetLastError() == hipSuccess
      && hipStreamSynchronize(e->stream) == hipSuccess
      && hipMemcpy(out.data(), d_out.p, out.size()*8, hipMemcpyDeviceToHost) == hipSuccess;
    if (ok)
      for (unsigned i = 0; i < count; ++i)
        std::copy(out.begin() + i*R*S*S, out.begin() + (i + 1)*R*S*S, pmatrix[matrix_indices[i]]);
    else fail("bpa_core_update_pmatrix: device error");
  }
  d_out.free(); d_rates.free(); d_bl.free(); d_ev.free(); d_evec.free(); d_ievec.free(); d_pi.free();
  return ok;
}

extern "C" int bpa_update_eigen(bpa_engine_t * e, double * eigenvecs, double * inv_eigenvecs,
                                double * eigenvals, const double * freqs,
                                const double * subst_params, unsigned states)
{
  std::lock_guard<std::recursive_mutex> lock_(e->mtx);
  if (!set_device(e)) return 0;
  if (states != 4 && states != 20) return fail("bpa_update_eigen: 4 or 20 states");
  const size_t S = states;
  DevBuf<double> f, q, ev, evec, ievec;
  int ok = upload(f, freqs, S) && upload(q, subst_params, S*(S-1)/2) && ev.reserve(S) && evec.reserve(S*S) && ievec.reserve(S*S);
  if (ok)
  {
    hipLaunchKernelGGL(eigen_lib_kernel, dim3(1), dim3(64), 0, e->stream, (uint32_t)S, f.p, q.p, ev.p, evec.p, ievec.p);
    ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(e->stream) == hipSuccess
      && hipMemcpy(eigenvals, ev.p, S*8, hipMemcpyDeviceToHost) == hipSuccess
      && hipMemcpy(eigenvecs, evec.p, S*S*8, hipMemcpyDeviceToHost) == hipSuccess
      && hipMemcpy(inv_eigenvecs, ievec.p, S*S*8, hipMemcpyDeviceToHost) == hipSuccess;
    if (!ok) fail("bpa_update_eigen: device error");
  }
  f.free(); q.free(); ev.free(); evec.free(); ievec.free();
  return ok;
}

// device-resident per-locus proposal control
#include "p2p.hpp"
#include "sampler.hpp"
