// composite.hpp — loci of several kinds behind ONE bpa_sampler_t (BPP takes a partition list with a
// model per locus and loci of any size: method.c:3320-3346, gtree.c:4585, 6531 loop over opt_locus_count).
//
// The device samplers are specialised: the LDS kernels of sampler.hpp / sweep2.hpp take JC69 loci of <= 8 tips and <= 64
// patterns, the generic path (gsampler.hpp) JC69 loci OR multi-category 4-state loci OR 20-state loci on the engine's
// packing.  Until round 4 one misfit locus decided for all of them (or the set was refused).  Here the loci are dealt to
// PARTS by kind — each part an ordinary sampler over its own loci — and the parts are stepped together exactly as the
// ranks of a sharded run are: same seed, same global stream (the same windows and acceptance numbers everywhere), the
// per-locus streams keyed by the locus's index in the WHOLE set, and the sum an all-loci step (THETA, TAU, MIX) is decided
// on exchanged through the all-reduce callback every part already has (bpa_sampler_set_allreduce).  The "collective" is a
// launch on the one engine stream that adds the parts' device sums; the parts' launch loops run as fibers of the calling
// thread (ucontext), each suspended inside its callback until every part has reached the same step.
// Several ranks (round 5; threads.c:234-353 shards ANY loci over its workers): the caller's all-reduce
// (bpa_sampler_set_allreduce on the composite) is called ONCE per step on the parts' total, between the launch that adds
// the parts and the launch that hands the result to each of them — a rank whose share is of one kind (a plain sampler:
// one collective per step) and a rank whose share is mixed therefore issue the same collectives in the same order.
// A part of LDS-kernel loci therefore runs the several-rank ("hybrid") form: its per-locus sweeps as launches of the
// persistent kernel, its all-loci steps one launch each.  The library's own proposal kernel (BPP's kernel and the program's
// moves decide inside the persistent kernel's single launch or on the generic sampler's host: homogeneous sets only).
#pragma once
#include <ucontext.h>
#include <functional>

struct bpa_composite
{
  std::vector<bpa_sampler *> parts;
  std::vector<unsigned> part_of, idx_in;            // per locus of the whole set
  // ---- fibers
  struct CbCtx { bpa_composite * c; int part; };
  std::vector<CbCtx> cb;
  ucontext_t main_ctx;
  std::vector<ucontext_t> ctx;
  std::vector<std::vector<char>> stacks;
  std::vector<int> state;                           // 0 to be resumed, 1 waiting in its callback, 2 done
  std::vector<int> result;
  std::vector<double *> pend_ptr; std::vector<unsigned> pend_n;
  std::function<int(bpa_sampler *)> job;
  int cur = -1;
  bool failed = false;
  // a stream per part (the engine's own for part 0): a part's launches go to its stream — the engine's stream field is
  // switched while the part's fiber runs —, so a small part's launch latency overlaps with a big part's kernels; events
  // order the sum of a step after every part's share and every part's continuation after the sum
  std::vector<hipStream_t> stream;
  std::vector<hipEvent_t> ev_part;
  hipEvent_t ev_sum = nullptr;
  // several ranks: the caller's collective over the ranks, the device doubles it wants the totals in (or null), and the
  // global index of this rank's first locus
  bpa_allreduce_fn ext = nullptr; void * ext_ctx = nullptr; double * ext_sum = nullptr; unsigned first = 0;
};

namespace comp {
constexpr int MAXPARTS = 8;
struct Ptrs { double * p[MAXPARTS]; };
// the parts' sums of one step, added up and handed back to every part (one thread per value); total != null: the totals go
// there only (the ranks' collective runs on them before hand_parts_kernel gives every part the result)
__global__ void sum_parts_kernel(const Ptrs P, int nparts, unsigned count, double * total)
{
  const unsigned j = blockIdx.x*blockDim.x + threadIdx.x;
  if (j >= count) return;
  double t = 0;
  for (int p = 0; p < nparts; ++p) t += P.p[p][j];
  if (total) { total[j] = t; return; }
  for (int p = 0; p < nparts; ++p) P.p[p][j] = t;
}
__global__ void hand_parts_kernel(const Ptrs P, int nparts, unsigned count, const double * total)
{
  const unsigned j = blockIdx.x*blockDim.x + threadIdx.x;
  if (j >= count) return;
  const double t = total[j];
  for (int p = 0; p < nparts; ++p) if (P.p[p] != total) P.p[p][j] = t;
}

static int callback(void * vctx, double * sums, unsigned count, void * /*stream*/)
{
  bpa_composite::CbCtx * x = static_cast<bpa_composite::CbCtx *>(vctx);
  bpa_composite * c = x->c;
  c->pend_ptr[x->part] = sums; c->pend_n[x->part] = count;
  c->state[x->part] = 1;
  swapcontext(&c->ctx[x->part], &c->main_ctx);      // back when every part has arrived and the sum is enqueued
  return c->failed ? 0 : 1;
}

static void trampoline(unsigned lo, unsigned hi)
{
  bpa_composite * c = reinterpret_cast<bpa_composite *>(((uintptr_t)hi << 32) | (uintptr_t)lo);
  const int i = c->cur;
  c->result[i] = c->job(c->parts[i]);
  c->state[i] = 2;
  swapcontext(&c->ctx[i], &c->main_ctx);
}

// run `job` on every part, all of them advancing together through their all-reduce callbacks
static int run_all(bpa_composite * c, bpa_engine * e, std::function<int(bpa_sampler *)> job)
{
  const int n = (int)c->parts.size();
  c->job = std::move(job); c->failed = false;
  // The engine's tables are uploaded synchronously on whatever stream a part's fiber has made current (flush / engine_pack
  // wait for e->stream only): when an upload is due, every part's stream is drained first, so that no kernel of ANOTHER part
  // still reads the old tables while they are replaced
  if (e->table_dirty || e->pack_dirty)
    for (int i = 0; i < n; ++i) if (c->stream[i] && hipStreamSynchronize(c->stream[i]) != hipSuccess) { c->failed = true; }
  for (int i = 0; i < n; ++i)
  {
    c->state[i] = 0; c->result[i] = 0;
    getcontext(&c->ctx[i]);
    c->ctx[i].uc_stack.ss_sp = c->stacks[i].data(); c->ctx[i].uc_stack.ss_size = c->stacks[i].size();
    c->ctx[i].uc_link = &c->main_ctx;
    const uintptr_t a = reinterpret_cast<uintptr_t>(c);
    makecontext(&c->ctx[i], reinterpret_cast<void (*)()>(trampoline), 2, (unsigned)(a & 0xffffffffu), (unsigned)(a >> 32));
  }
  for (;;)
  {
    for (int i = 0; i < n; ++i)
      if (c->state[i] == 0)
      {
        c->cur = i;
        e->stream = c->stream[i];
        swapcontext(&c->main_ctx, &c->ctx[i]);
        e->stream = c->stream[0];
      }
    int waiting = 0, done = 0;
    for (int i = 0; i < n; ++i) { waiting += c->state[i] == 1; done += c->state[i] == 2; }
    if (!waiting) break;
    bool bad = done != 0;                                   // (a part finished or failed while others wait for its sum)
    unsigned cnt = 0;
    for (int i = 0; i < n; ++i) if (c->state[i] == 1) { if (!cnt) cnt = c->pend_n[i]; bad = bad || c->pend_n[i] != cnt; }
    if (bad) c->failed = true;
    else
    {
      Ptrs P{};
      for (int i = 0; i < n; ++i) P.p[i] = c->pend_ptr[i];
      bool okq = true;
      for (int i = 1; i < n; ++i)
        okq = okq && hipEventRecord(c->ev_part[i], c->stream[i]) == hipSuccess && hipStreamWaitEvent(c->stream[0], c->ev_part[i], 0) == hipSuccess;
      // (several ranks: the totals in the caller's doubles — or part 0's — go through its collective on the same stream)
      double * total = !c->ext ? nullptr : (c->ext_sum && cnt <= (unsigned)BPA_SAMPLER_SUMS) ? c->ext_sum : c->pend_ptr[0];
      hipLaunchKernelGGL(sum_parts_kernel, dim3((cnt + 63)/64), dim3(64), 0, c->stream[0], P, n, cnt, total);
      okq = okq && hipGetLastError() == hipSuccess;
      if (okq && c->ext)
      {
        if (!c->ext(c->ext_ctx, total, cnt, (void *)c->stream[0])) { okq = false; fail("bpa_sampler (composite): the all-reduce callback failed"); }
        hipLaunchKernelGGL(hand_parts_kernel, dim3((cnt + 63)/64), dim3(64), 0, c->stream[0], P, n, cnt, total);
        okq = okq && hipGetLastError() == hipSuccess;
      }
      okq = okq && hipEventRecord(c->ev_sum, c->stream[0]) == hipSuccess;
      for (int i = 1; i < n; ++i) okq = okq && hipStreamWaitEvent(c->stream[i], c->ev_sum, 0) == hipSuccess;
      if (!okq) c->failed = true;
    }
    for (int i = 0; i < n; ++i) if (c->state[i] == 1) c->state[i] = 0;
  }
  for (int i = 1; i < n; ++i)
    if (hipEventRecord(c->ev_part[i], c->stream[i]) != hipSuccess || hipStreamWaitEvent(c->stream[0], c->ev_part[i], 0) != hipSuccess) c->failed = true;
  int ok = !c->failed;
  for (int i = 0; i < n; ++i) ok = ok && c->result[i];
  if (c->failed) fail("bpa_sampler (composite): the parts did not reach the same all-loci step together");
  return ok;
}
}  // namespace comp

static bpa_sampler * sampler_create_plain(bpa_engine_t * e, bpa_locus_t * const * loci, unsigned nloci, unsigned long seed);

// the kinds the specialised samplers take; -1: none of them (the caller reports), 4: the big-tree sampler's
static int comp_kind_of(const bpa_locus * l, bpa_engine * e)
{
  const bool counts = l && l->eng == e && l->alive && l->tips >= 2 && l->clv_buffers == 2*(l->tips - 1) && l->prob_matrices == 2*(2*l->tips - 2);
  if (!counts) return -1;
  if (l->states == 4 && (l->tips > (unsigned)gsm::NT || l->scale_buffers != 0 || l->dev.unphased_length)) return 4;
  if (l->scale_buffers != 0 || l->dev.unphased_length) return -1;
  if (l->states == 20) return (l->tips <= (unsigned)gsm::NT && l->rate_cats <= 4) ? 3 : -1;
  if (l->states != 4) return -1;
  const bool jc1 = l->rate_cats == 1 && l->dev.model == 0;
  if (jc1 && l->tips <= (unsigned)smp::MAXTIPS && l->sites <= (unsigned)smp::BS && !smp_env_generic()) return 0;      // the LDS kernels
  if (!(l->tips <= (unsigned)gsm::NT && l->rate_cats <= 8 && l->sites*l->rate_cats < PACK_BS)) return -1;
  if (jc1) return 1;                                     // generic path, JC69 records
  if (l->rate_cats > 1) return 2;                        // generic path, multi-category records
  return -1;
}

// a composite where the loci are of more than one kind (and none wants the big-tree sampler); nullptr + *plain = true: not needed
static bpa_sampler * comp_create(bpa_engine_t * e, bpa_locus_t * const * loci, unsigned nloci, unsigned long seed, bool * plain)
{
  *plain = true;
  if (getenv("BPA_SMP_NO_COMPOSITE") || smp_env_big()) return nullptr;
  std::vector<unsigned> grp[4];
  for (unsigned i = 0; i < nloci; ++i)
  {
    const int k = comp_kind_of(loci[i], e);
    if (k < 0 || k == 4) return nullptr;                   // (the plain path reports, or takes the whole set as big trees)
    grp[k].push_back(i);
  }
  // a handful of LDS-kernel loci next to generic JC69 loci: not worth a part of their own
  if (!grp[1].empty() && grp[0].size() < 64) { grp[1].insert(grp[1].end(), grp[0].begin(), grp[0].end()); std::sort(grp[1].begin(), grp[1].end()); grp[0].clear(); }
  int kinds = 0;
  for (auto & g : grp) kinds += !g.empty();
  if (kinds < 2) return nullptr;
  *plain = false;
  bpa_sampler * s = new bpa_sampler();
  s->eng = e; s->nloci = nloci; s->seed = seed;
  s->loci.assign(loci, loci + nloci);
  bpa_composite * c = new bpa_composite();
  s->comp = c;
  c->part_of.assign(nloci, 0); c->idx_in.assign(nloci, 0);
  for (int k = 0; k < 4; ++k)
  {
    if (grp[k].empty()) continue;
    std::vector<bpa_locus_t *> sub;
    for (unsigned i : grp[k]) sub.push_back(loci[i]);
    bpa_sampler * p = sampler_create_plain(e, sub.data(), (unsigned)sub.size(), seed);
    if (!p) { for (auto * q : c->parts) bpa_sampler_destroy(q); delete c; delete s; return nullptr; }
    p->stream_index = grp[k];
    for (unsigned j = 0; j < grp[k].size(); ++j)
    {
      c->part_of[grp[k][j]] = (unsigned)c->parts.size(); c->idx_in[grp[k][j]] = j;
      const a00_rng_t r = stream_seed(p, grp[k][j]);
      if (p->generic) p->g_trees[j].rng = r; else p->h_trees[j].rng = r;
    }
    c->parts.push_back(p);
    s->maxtips = std::max(s->maxtips, p->maxtips);
  }
  const size_t n = c->parts.size();
  c->cb.resize(n); c->ctx.resize(n); c->state.assign(n, 2); c->result.assign(n, 0); c->pend_ptr.assign(n, nullptr); c->pend_n.assign(n, 0);
  c->stacks.resize(n);
  c->stream.assign(n, e->stream); c->ev_part.assign(n, nullptr);
  const bool one_stream = BPA_EXP_SWITCH("BPA_COMP_ONE_STREAM") != nullptr;
  if (hipEventCreateWithFlags(&c->ev_sum, hipEventDisableTiming) != hipSuccess) c->ev_sum = nullptr;
  for (size_t i = 1; i < n && !one_stream; ++i)
  {
    hipStream_t st = nullptr;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); continue; }        // (the part then runs on the engine's stream)
    if (hipEventCreateWithFlags(&c->ev_part[i], hipEventDisableTiming) != hipSuccess)
    { (void)hipGetLastError(); (void)hipStreamDestroy(st); c->ev_part[i] = nullptr; continue; }                          // (no stream without its event: nothing leaks)
    c->stream[i] = st;
  }
  for (size_t i = 1; i < n; ++i) if (!c->ev_part[i]) (void)hipEventCreateWithFlags(&c->ev_part[i], hipEventDisableTiming);
  for (size_t i = 0; i < n; ++i)
  {
    c->stacks[i].resize((size_t)1 << 20);
    c->cb[i] = bpa_composite::CbCtx{c, (int)i};
    c->parts[i]->allreduce = comp::callback; c->parts[i]->allreduce_ctx = &c->cb[i]; c->parts[i]->sum_ext = nullptr;
  }
  return s;
}

static void comp_destroy(bpa_sampler * s)
{
  bpa_composite * c = s->comp;
  for (size_t i = 1; i < c->stream.size(); ++i)
  {
    if (c->stream[i] != s->eng->stream) { (void)hipStreamSynchronize(c->stream[i]); (void)hipStreamDestroy(c->stream[i]); }
    if (c->ev_part[i]) (void)hipEventDestroy(c->ev_part[i]);
  }
  if (c->ev_sum) (void)hipEventDestroy(c->ev_sum);
  for (auto * p : s->comp->parts) bpa_sampler_destroy(p);
  delete s->comp; s->comp = nullptr;
}
template <class F> static int comp_each(bpa_sampler * s, F f) { int ok = 1; for (auto * p : s->comp->parts) ok = f(p) && ok; return ok; }
static int comp_invalidate(bpa_sampler * s) { return comp_each(s, [](bpa_sampler * p) { return sampler_invalidate(p); }); }
// every part on the device (the upload agrees on the THETA mask through the callback: all parts together)
static int comp_upload(bpa_sampler * s)
{
  bool all = true;
  for (auto * p : s->comp->parts) all = all && p->uploaded;
  if (all) return 1;
  if (!comp_invalidate(s)) return 0;
  return comp::run_all(s->comp, s->eng, [](bpa_sampler * p) { return sampler_upload(p); });
}

// several ranks: the caller's collective over the ranks (run_all calls it once per step on the parts' total); the per-locus
// streams re-keyed by the loci's indices in the whole data set (this rank's share starts at first_locus)
static int comp_set_allreduce(bpa_sampler * s, bpa_allreduce_fn fn, void * ctx, double * device_sum, unsigned first_locus)
{
  bpa_composite * c = s->comp;
  c->ext = fn; c->ext_ctx = ctx; c->ext_sum = device_sum;
  if (first_locus == c->first) return 1;
  if (!comp_invalidate(s)) return 0;
  c->first = first_locus;
  for (unsigned i = 0; i < s->nloci; ++i)
  {
    bpa_sampler * p = c->parts[c->part_of[i]];
    const unsigned j = c->idx_in[i];
    p->stream_index[j] = first_locus + i;
    const a00_rng_t r = stream_seed(p, first_locus + i);
    if (p->generic) p->g_trees[j].rng = r; else p->h_trees[j].rng = r;
  }
  return 1;
}

static bpa_sampler * comp_part(bpa_sampler * s, unsigned i, unsigned * j)
{
  if (i >= s->nloci) return nullptr;
  *j = s->comp->idx_in[i];
  return s->comp->parts[s->comp->part_of[i]];
}
static bpa_sampler * comp_part0(bpa_sampler * s) { return s->comp->parts[0]; }
// what = 0: start-up evaluation, 1: n iterations — every part, together
static int comp_run(bpa_sampler * s, int what, unsigned n)
{
  if (!set_device(s->eng) || !comp_upload(s)) return 0;
  return comp::run_all(s->comp, s->eng, [what, n](bpa_sampler * p) { return what ? bpa_sampler_iterate(p, n) : bpa_sampler_initialize(p); });
}
static int comp_summary(bpa_sampler * s, double * total_lnl, unsigned long * proposals, unsigned long * accepted, unsigned long * launches)
{
  if (!comp_upload(s)) return 0;
  double tot = 0; unsigned long pr = 0, ac = 0, la = 0; int k = 0;
  for (auto * p : s->comp->parts)
  {
    double t; unsigned long a, b, c;
    if (!bpa_sampler_summary(p, &t, &a, &b, &c)) return 0;
    uint32_t cc[2];
    HIPCHK(hipMemcpy(cc, p->counters.p, 8, hipMemcpyDeviceToHost));
    // (every part counts the all-loci steps: once is enough)
    tot += t; pr += a - (k ? cc[0] : 0u); ac += b - (k ? cc[1] : 0u); la += c; ++k;
  }
  if (total_lnl) *total_lnl = tot;
  if (proposals) *proposals = pr;
  if (accepted) *accepted = ac;
  if (launches) *launches = la;
  return 1;
}
static int comp_timing(bpa_sampler * s, double * sweep_ms, unsigned long * sweep_launches, double * allloci_ms, unsigned long * allloci_launches)
{
  double a = 0, c = 0; unsigned long b = 0, d = 0;
  for (auto * p : s->comp->parts) { double x, z; unsigned long y, w; if (!bpa_sampler_timing(p, &x, &y, &z, &w)) return 0; a += x; b += y; c += z; d += w; }
  if (sweep_ms) *sweep_ms = a;
  if (sweep_launches) *sweep_launches = b;
  if (allloci_ms) *allloci_ms = c;
  if (allloci_launches) *allloci_launches = d;
  return 1;
}
static int comp_work(bpa_sampler * s, double * bytes, unsigned long * node_updates, unsigned long * pattern_updates, unsigned long * sweeps)
{
  if (!comp_upload(s)) return 0;
  double a = 0; unsigned long b = 0, c = 0, d = 0;
  for (auto * p : s->comp->parts) { double x; unsigned long y, z, w; if (!bpa_sampler_work(p, &x, &y, &z, &w)) return 0; a += x; b += y; c += z; d += w; }
  if (bytes) *bytes = a;
  if (node_updates) *node_updates = b;
  if (pattern_updates) *pattern_updates = c;
  if (sweeps) *sweeps = d;
  return 1;
}
