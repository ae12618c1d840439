// sweep2.hpp — the A00 iteration as ONE persistent launch (round 3).
//
// sampler.hpp's sweep kernel gives every locus one LEADER lane that runs the proposal bookkeeping of propose_ages /
// propose_spr (gtree.c:4585, 6531) alone while the other lanes of its wave wait: one wave per SIMD, <= 12 of 64 lanes
// busy, and the launch lasts as long as that lane's instruction stream.  Here
//
//  * a locus owns an ALIGNED GROUP of G = 8 (<= 4 tips) or 16 (<= 8 tips) lanes.  Every lane of the group runs the
//    proposal (the integer tree — children, parent, population of <= 15 nodes — is replicated in its registers as byte
//    arrays, so a look-up is one v_perm_b32 and costs nothing extra: a wave instruction is paid for 64 lanes anyway),
//    but every LOOP of the leader's code is gone: lane i IS node i, population i, branch i and pattern i (+G, +2G ...),
//    a loop over nodes or populations is one predicate per lane + one wave ballot whose group's byte is the node set.
//    Per-population lineage counts come from nin_p = (gene tips below p) - (coalescences strictly below p), so the
//    children-first chain of gtree_update_logprob_contrib's bookkeeping is one pass too;
//  * buffer toggles are two bit masks: clv_index of inner node i is i or i + inner, pmatrix_index i or i + edges
//    (locus.c:24-26 with the start values of gtree.c:2398, 2433) — a toggle of a node set is one XOR;
//  * the state of ALL loci (trees, CLV buffers, (a,b) tables: 17 MB for config 2) stays in LDS for the whole launch —
//    many iterations — instead of going to HBM and back ten times per iteration;
//  * the single decision of an all-loci step (TAU, MIX, THETA: stree.c:5512, prop_mixing.c:52, stree.c:3464) needs the
//    sum over every locus: each workgroup publishes its partial sum as 8-byte {epoch, half} granules with write-through
//    stores, wave 0 of every workgroup gathers all of them (relaxed agent-scope loads until every tag is this step's),
//    adds them up in workgroup order and takes the same decision bit for bit — an all-gather, no second broadcast hop.
//    Every spin is bounded (wall clock); a time-out raises the error word and every workgroup leaves.
//
// Arithmetic (proposal windows, reflections, density terms, JC69 exponentials, 4x4 mat-vecs, ordered sums, Metropolis-
// Hastings ratios) is sampler.hpp's, operation for operation: same per-locus streams, same trajectory as the host
// driver csrc/host/a00_driver.c (tests/test_gpu_sampler.py).  One GPU; with an all-reduce callback installed (several
// ranks) bpa_sampler_iterate keeps to sampler.hpp's one-launch-per-step path.
#pragma once

namespace smp2 {

using smp::Tree; using smp::Species; using smp::ByteArr; using smp::MAXPOP; using smp::MAXN;
using smp::rndu; using smp::reflect; using smp::msc_term; using smp::nth_bit; using smp::Op; using smp::make_op;

template <int NT> struct Cfg
{
  static constexpr int G     = NT <= 4 ? 8 : NT <= 8 ? 16 : 32;     // lanes per locus = node slots = population slots (32: the generic sampler's group proposals, gsampler2.hpp)
  static constexpr int LPW   = 64/G;                 // loci per wave
  static constexpr int NN    = 2*NT;                 // node slots (2 NT - 1 nodes)
  static constexpr int W     = NN/4;                 // 32-bit words of a byte array
  static constexpr int NBUF  = 2*(NT - 1);
  static constexpr int NPM   = 2*(2*NT - 2);
#ifndef SMP2_WAVES
#define SMP2_WAVES 8
#endif
  static constexpr int WAVES = NT <= 4 ? SMP2_WAVES : 4;      // per workgroup: 64 / 16 loci
  static constexpr int BS    = 64*WAVES;
  static_assert(G == NN, "lane i of a group is node i");
};

struct Loc                               // per locus, constant over the run (flattened at upload)
{
  double * clv, * pmat;                  // inner CLV buffer 0 / the (a,b) table
  double rate, rw, f0, f1, f2, f3;
  uint32_t np, tips, pat_off, pad;
  int8_t gl[16];                         // gene tips below each population
};

struct Args
{
  const uint32_t * wave_off;             // [nwaves + 1] first locus of every wave
  const Loc * loc;
  const uint2 * pat;                     // per pattern: weight, tip codes (4 bits per tip)
  Tree * trees;
  // what the one-launch-per-step path (sampler.hpp) may have left pending when this launch starts (several ranks: its
  // all-loci steps alternate with this kernel's sweeps): the decision of an all-loci step, applied here as there — a
  // rejected step's trees come from the pre-step snapshot — and thetas that moved after the trees' densities were stored
  const Tree * snap; const uint32_t * mix_flag; uint32_t epoch, refresh_logpr;
  double * taus;                         // [3 MAXPOP] tau | theta | log(2/theta): read at entry, written back by workgroup 0
  uint32_t * counters;                   // all-loci proposals / accepted, Gibbs draws of a theta / accepted
  const double * lograt;
  int8_t * pop_nc; double * pop_t2h;     // sufficient statistics of the final state (sampler.hpp's THETA kernels read them)
  uint32_t ntasks, nwaves, nwg;
  uint32_t lwaves;                       // waves with loci per workgroup (<= WAVES, PROG: <= WAVES - 1): fewer where the loci allow — one wave per SIMD has the SIMD to itself
  unsigned long long * xbuf;             // [2][XN] accumulators of the all-loci steps' sums
  int * err;                             // [0] a wait timed out: the launch left everything as it found it; [1] += the iterations it did not run; [2] += all-loci steps accepted with a term summed through the coarse companion
  a00_rng_t * grng;                      // the global stream: read at entry, written back by workgroup 0
  uint32_t niter, nsteps_gage, nsteps_gspr, theta_mask, do_allloci, dbg;
  double bfbeta;
  double * prof, * declog;
  // several GPUs (bpa_sampler_set_p2p): every rank's mailbox as mapped here (p2p.hpp: [2][world] slots of slot_bytes,
  // a sequence flag + values each), this rank's own, the sequence number before this launch's first exchange
  unsigned char * const * peers; unsigned char * mail; int32_t rank, world; unsigned long long slot_bytes, seq0, spin_limit; int * p2p_err;
  const Species * sp;                    // (device memory: by value it would sit in ~50 SGPRs for the whole launch)
  unsigned long long * pj;               // proposals / accepted by move type since the host last cleared them (bpa_sampler_adapt_finetune): gage, gspr, tau, mix, theta window
};

constexpr int XN = 128;                  // words per accumulator set: 8 shards x (15 sums + the arrival counter) = 8 x 128 bytes
constexpr int XV = 15;                   // sums per block of an exchange

template <int NT> struct Slot
{
  double time[Cfg<NT>::NN];
  double ab[Cfg<NT>::NPM][2];
  double contrib[Cfg<NT>::G], contrib_new[Cfg<NT>::G];
  double rate, rw, f[4];                 // the locus's constants (Loc), read where they are used
};
template <int NT> struct WaveLDS
{
  double clv[Cfg<NT>::NBUF][64][4];
  double term[64];
  uint2  pat[64];
  Slot<NT> slot[Cfg<NT>::LPW];
};
// what the program's moves keep about population p (wave 0's business): k_p and T_p — the sums over ALL loci of the
// coalescences in p and of T2h, from the THETA step's exchange on, carried through TAU and MIX (run_k / run_T of
// a00_driver.c) — and the inverse gamma fitted to the theta's conditional given them (a, b, c = a log b - lgamma a): the
// "current" side of every re-draw's proposal ratio, always a fit some step already made from exactly these two numbers
struct PopFit { double k, T, a, b, c; };
// what a re-draw leaves for population p: theta', log(2/theta'), the new sum T' and the fit to it
struct Redraw { double tn, l2t, T, a, b, c; };
// the part of the workgroup's LDS that the decision functions below (not inlined: their registers are their own) read by name
struct WgBase
{
  double tau[3*MAXPOP];
  double xtot[32];
  Species sp;
  // the program's moves: wave 0 takes an all-loci step's decision alone (the others wait at the barrier: a SIMD to itself) and leaves it here
  struct { unsigned long long grng; uint32_t accm, rd_mask, acc_step, run_ok; double lnacc_step, lnacc_theta; double tn[16], l2t[16], lnacc[16]; } dec;
  PopFit pf[16];
  Redraw rd[16];
  // the proposal of the species tree of the coming all-loci step, as the control wave publishes it (program-moves kernel)
  struct { int32_t q, mix; double tq_old, tq_lo, tq_hi, tq_new, minf, maxf, lminf, lmaxf, mix_c, mix_lnc; } stepp;
};
template <int NT> struct WgLDS : WgBase
{
  double lograt[(2*NT)*(2*NT)];
  unsigned long long accfx[32];                  // this workgroup's sums of an all-loci step, 2^-44 fixed point (LDS atomics)
  uint32_t anc[16];
  uint32_t abort_, bad_, xbad_, coarse_;         // coarse_: a term of this workgroup went to the coarse companion sum; xcoarse_: of any
  unsigned long long xprev[2][XN];               // wave 0: every word of either accumulator set as its previous use left it
  long long prof[24];                            // BPA_SMP_DBG & 16: cycle counters of thread 0 of workgroup 0
  long long wsweep[16];                          // BPA_SMP_DBG & 16: sweep cycles of every wave of workgroup 0
  uint32_t xcoarse_, late_;                      // late_: another workgroup gave up in this launch (read from Args::err before the store)
};

template <int G> __device__ __forceinline__ uint32_t gballot(bool p, uint32_t gbase)
{
  return (uint32_t)(__ballot(p) >> gbase) & (G >= 32 ? 0xffffffffu : ((1u << (G & 31)) - 1u));
}
// LDS traffic between the lanes of ONE wave: the hardware runs a wave's DS instructions in order; this keeps the
// compiler from moving accesses across the hand-over
__device__ __forceinline__ void wsync()
{
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// the integer part of a gene tree, replicated in every lane of the locus's group
template <int NT> struct GTree
{
  ByteArr<Cfg<NT>::W> left, right, parent, pop;
  uint32_t cf, pf;                       // toggle flags per node: CLV buffer / P-matrix buffer
  int32_t root, tips;
  __device__ __forceinline__ int cidx(int i) const { return i + (((cf >> i) & 1u) ? tips - 1 : 0); }
  __device__ __forceinline__ int pidx(int i) const { return i + (((pf >> i) & 1u) ? 2*tips - 2 : 0); }
};

__device__ __forceinline__ uint32_t splat(int v) { return (uint32_t)(v & 0xff)*0x01010101u; }
// 0xff in every byte of w that equals the byte splatted in s
__device__ __forceinline__ uint32_t byte_eq(uint32_t w, uint32_t s)
{
  const uint32_t x = w ^ s;
  const uint32_t z = ~(((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x | 0x7f7f7f7fu);      // 0x80 where the byte is zero
  return (z - (z >> 7)) | z;
}
// exchange the tree positions of node ids a and b (swap_ids of a00_driver.c): every reference to a becomes b and
// vice versa — on all bytes of a word at once —, then the two entries change places; buffer flags stay with the ids
template <int NT> __device__ __forceinline__ void swap_ids(GTree<NT> & t, double * time, int a, int b)
{
  constexpr int W = Cfg<NT>::W;
  const uint32_t sa = splat(a), sb = splat(b), x = sa ^ sb;
#pragma unroll
  for (int k = 0; k < W; ++k)
  {
    t.left.w[k]   ^= x & (byte_eq(t.left.w[k], sa)   | byte_eq(t.left.w[k], sb));
    t.right.w[k]  ^= x & (byte_eq(t.right.w[k], sa)  | byte_eq(t.right.w[k], sb));
    t.parent.w[k] ^= x & (byte_eq(t.parent.w[k], sa) | byte_eq(t.parent.w[k], sb));
  }
  const int la = t.left[a], ra = t.right[a], pa = t.parent[a], qa = t.pop[a];
  const int lb = t.left[b], rb = t.right[b], pb = t.parent[b], qb = t.pop[b];
  t.left.set(a, lb); t.right.set(a, rb); t.parent.set(a, pb); t.pop.set(a, qb);
  t.left.set(b, la); t.right.set(b, ra); t.parent.set(b, pa); t.pop.set(b, qa);
  const double ta = time[a], tb = time[b];
  time[a] = tb; time[b] = ta;
  t.root = t.root == a ? b : t.root == b ? a : t.root;
}
template <int NT> __device__ __forceinline__ uint32_t path_mask(const GTree<NT> & t, int v)
{
  uint32_t m = 0;
#pragma unroll
  for (int d = 0; d < NT; ++d) { if (v >= 0) { m |= 1u << v; v = t.parent[v]; } }
  return m;
}

// what the lanes keep about the species tree: lane i is population i
struct PopLane
{
  double tau, ptau, theta, l2t;          // this launch's current (or, inside an all-loci step, proposed) values
  int32_t parent;
  uint32_t anc, below;                   // ancestors-or-self / strict descendants of population i
};

// the proposal of one per-locus step, as left by propose_*: what to recompute
struct Prop { uint32_t chain, brm, ndm; double hast; };

// The two proposal kernels of a00_driver.c (bpp_amd_host.h: A00_KERNEL_UNIFORM / A00_KERNEL_BPP), per stream:
//   uniform  our 64-bit streams (a00_rndu), a window = finetune x (u - 1/2), the acceptance number always drawn;
//   BPP      the reference's own: legacy_rndu (z = 69069 z + 1 on 32 bits, random.c:104-122) and, for a window,
//            legacy_rnd_symmetrical = the Bactrian-Laplace variate (m = 0.90: random.c:192-238), mean 0, variance 1, two
//            draws; a proposal that cannot be made draws nothing, the acceptance number is drawn only when needed
//            (lnacc < -1e-10: gtree.c:5476, stree.c:6286).  The state lives in the low 32 bits of the stream word.
template <bool BPP> struct Stream
{
  a00_rng_t r;
  __device__ __forceinline__ double u()
  {
    if (!BPP) return rndu(&r);
    uint32_t z = (uint32_t)r*69069u + 1u;
    if (z == 0u) z = 12345671u;
    r = z;
    return (double)z*(1.0/4294967296.0);                    // ldexp(z, -32), exact
  }
  // a sliding-window step in units of the finetune
  __device__ __forceinline__ double window()
  {
    if (!BPP) return rndu(&r) - 0.5;
    const double uu = u() - 0.5;
    const double rr = log(1 - 2*fabs(uu))*0.70710678118654752440;
    const double lap = uu >= 0 ? -rr : rr;
    double v = 0.90 + lap*sqrt(1 - 0.90*0.90);
    if (u() < 0.5) v = -v;
    return v;
  }
  __device__ __forceinline__ void skip() { if (!BPP) (void)rndu(&r); }
  __device__ __forceinline__ bool accept(double lnacc)
  {
    if (BPP) return lnacc >= -1e-10 || u() < exp(lnacc);
    const double uu = rndu(&r);
    return lnacc >= 0 || uu < exp(lnacc);
  }
};

// GAGE on the k-th inner node (gage_step of a00_driver.c; propose_ages, gtree.c:4585) — all lanes of the group
template <int NT, bool BPP>
__device__ __forceinline__ bool propose_gage(GTree<NT> & t, Stream<BPP> & rng, double * time, int k, const PopLane & pl,
                                             const uint32_t * anc, const double * tau, double ft, int li, uint32_t gbase, Prop & pr)
{
  constexpr int G = Cfg<NT>::G;
  const int n = 2*t.tips - 1, v = t.tips + k;
  if (v >= n) return false;
  const double u = rng.window();
  const int l = t.left[v], r = t.right[v], p = t.parent[v];
  const double tl = time[l], tr = time[r], told = time[v], tpar = time[p < 0 ? 0 : p];
  const int pol = t.pop[l], por = t.pop[r];
  const uint32_t al = anc[pol], ar = anc[por];
  double lo = fmax(tl, tr);
  if (pol != por) lo = fmax(lo, tau[__ffs(al & ar) - 1]);          // the youngest common ancestor: the lowest common bit
  const double hi = p >= 0 ? tpar : 999.0;
  if (!(hi > lo)) { rng.skip(); return false; }
  const double tnew = reflect(told + ft*u, lo, hi);
  const int oldpop = t.pop[v];
  time[v] = tnew;
  // climb (gtree.c:4790-4797): the highest ancestor-or-self of the left child's population that has started by tnew
  const uint32_t cm = gballot<G>(li == pol || (((al >> li) & 1u) && pl.tau <= tnew), gbase);
  const int newpop = 31 - __clz(cm);
  t.pop.set(v, newpop);
  {
    const uint32_t aa = anc[oldpop], ab = anc[newpop];
    const bool a_lower = (aa >> newpop) & 1u;
    const uint32_t lw = a_lower ? aa : ab, hg = a_lower ? ab : aa;
    const int higher = a_lower ? newpop : oldpop;
    pr.chain = lw & ~(hg & ~(1u << higher));
  }
  pr.hast = 0;
  pr.brm = (1u << l) | (1u << r) | (p >= 0 ? 1u << v : 0u);
  pr.ndm = path_mask<NT>(t, v);
  return true;
}

// GSPR on the k-th non-root node (gspr_step of a00_driver.c; propose_spr, gtree.c:6531) — all lanes of the group
template <int NT, bool BPP>
__device__ __forceinline__ bool propose_gspr(GTree<NT> & t, Stream<BPP> & rng, double * time, int k, const PopLane & pl, int gl_i,
                                             const uint32_t * anc, const double * tau, const double * lograt, double ft,
                                             int li, uint32_t gbase, Prop & pr)
{
  constexpr int G = Cfg<NT>::G;
  const int n = 2*t.tips - 1;
  const int a = k < t.root ? k : k + 1;
  if (a >= n) return false;
  const double u1 = rng.window(), u2 = rng.u();
  const int root_before = t.root;
  const int p = t.parent[a], lp = t.left[p], s = lp == a ? (int)t.right[p] : lp, g = t.parent[p];
  // gene tips below a: lane i walks up from node i
  uint32_t sub;
  {
    int x = li; bool in = false;
#pragma unroll
    for (int d = 0; d < NT; ++d) { in = in || x == a; x = x >= 0 ? (int)t.parent[x] : x; }
    sub = gballot<G>(in && li < n, gbase);
  }
  const int leaves = __popc(sub & ((1u << t.tips) - 1u));
  const int popa = t.pop[a];
  const uint32_t apa = anc[popa];
  // youngest population from a's upwards that holds gene tips outside a's subtree (gtree.c:6664-6669)
  const int pop0 = __ffs(gballot<G>(((apa >> li) & 1u) && (gl_i > leaves || pl.parent < 0), gbase)) - 1;
  const double ta = time[a], tpo = time[p], troot = time[root_before];
  const double lo = fmax(ta, tau[pop0]);
  const double tnew = reflect(tpo + ft*u1, lo, 999.0);
  const int popt = 31 - __clz(gballot<G>(li == popa || (((apa >> li) & 1u) && pl.tau <= tnew), gbase));
  // targets (bit j = branch above node j; the father's own branch stands for the sibling's) and sources: lane j looks at node j
  uint32_t tmask; int nsrc;
  {
    const int pp = t.pop[p];
    const bool above_root = tnew >= troot, src_on = p != root_before;
    const int j = li, pj = t.parent[j];
    const double tj = time[j], tpj = time[pj < 0 ? 0 : pj];
    const uint32_t aj = anc[(int)t.pop[j] & 15];
    const bool in = j < n && j != a && j != root_before;
    tmask = gballot<G>(in && !above_root && tj <= tnew && tpj > tnew && ((aj >> popt) & 1u), gbase);
    nsrc = 1 + __popc(gballot<G>(in && src_on && j != s && j != p && tj <= tpo && tpj > tpo && ((aj >> pp) & 1u), gbase));
    if (above_root) tmask = 1u << root_before;
  }
  const int ntg = __popc(tmask);
  if (!ntg) { rng.skip(); return false; }
  int pick = (int)(u2*ntg);
  if (pick == ntg) pick = 0;
  int tgt = nth_bit(tmask, pick);
  if (tgt == p) tgt = s;
  // prune: the sibling takes p's place; regraft p (with a below it) above tgt at tnew in popt
  t.parent.set(s, g);
  if (g >= 0) { if (t.left[g] == p) t.left.set(g, s); else t.right.set(g, s); } else t.root = s;
  const int pc = t.parent[tgt];
  {
    const int pp = t.pop[p];
    const uint32_t aa = anc[pp], ab = anc[popt];
    const bool a_lower = (aa >> popt) & 1u;
    const uint32_t lw = a_lower ? aa : ab, hg = a_lower ? ab : aa;
    const int higher = a_lower ? popt : pp;
    pr.chain = lw & ~(hg & ~(1u << higher));
  }
  time[p] = tnew; t.pop.set(p, popt);
  t.left.set(p, a); t.right.set(p, tgt); t.parent.set(a, p); t.parent.set(tgt, p);
  t.parent.set(p, pc);
  if (pc >= 0) { if (t.left[pc] == tgt) t.left.set(pc, p); else t.right.set(pc, p); } else t.root = p;
  uint32_t ndm = path_mask<NT>(t, p);
  if (g >= 0) ndm |= path_mask<NT>(t, g);
  uint32_t bset = (1u << a) | (1u << tgt) | (1u << p) | (1u << s);
  if (t.root != root_before)
  {
    // the root node object keeps its identity (gtree.c:6129-6175): rename the two ids in the sets
    const int newtop = t.root;
    wsync();
    swap_ids<NT>(t, time, newtop, root_before);
    const uint32_t bn = 1u << newtop, br_ = 1u << root_before;
    auto ren = [&](uint32_t m) { const uint32_t hn = m & bn, hr = m & br_; m &= ~(bn | br_); if (hn) m |= br_; if (hr) m |= bn; return m; };
    ndm = ren(ndm) | path_mask<NT>(t, newtop);
    bset = ren(bset);
  }
  // branches of the set that exist (the root has none): lane j answers for node j
  pr.brm = bset & gballot<G>((int)t.parent[li] >= 0 && li < n, gbase);
  pr.ndm = ndm;
  pr.hast = lograt[ntg*(2*NT) + nsrc];
  return true;
}

// ------------------------------------------------------------------------------------------------------------------------
// The program's moves (a00_set_program_moves; THETA by the metropolized Gibbs draw, thetas re-drawn inside TAU and MIX:
// stree.c:3957, 5840; prop_mixing.c:272) — the decisions.  Wave 0 of every workgroup takes them, alone (all 64 lanes; the
// other waves wait at a barrier, so its SIMD is its own), in functions of their own: their square roots, logarithms and
// loops want ~200 registers, which inside the iteration kernel came out of the sweep's (230 spilled registers; with the
// functions apart: 7).  Lane p < 16 is population p; lane p + 16 m (m = 1, 2, 3) takes further pieces of p's arithmetic, so
// that every logarithm / quotient of a step is ONE call for all populations.  State between steps: WgBase::pf.
__device__ __forceinline__ WgBase & wg_base()
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  return *reinterpret_cast<WgBase *>(smem);
}
__device__ __forceinline__ double prog_qnan() { return __longlong_as_double(0x7ff8000000000000ll); }
// lgamma(a) from log(a) (Stirling's series: |error| < 1e-16 from 16 on), so that it shares a log call with its neighbours
__device__ __forceinline__ double lgamma_with_log(double a, double la)
{
  if (!(a >= 16.0)) return lgamma(a);
  const double r = 1.0/a, r2 = r*r;
  const double ser = r*(1.0/12 - r2*(1.0/360 - r2*(1.0/1260 - r2*(1.0/1680 - r2*(1.0/1188)))));
  return ((a - 0.5)*la - a) + (0.91893853320467274178 + ser);
}
__device__ __forceinline__ double prog_lcg(uint32_t & z)                   // a00_bpp_rndu_hd (legacy_rndu, random.c:104-122)
{
  z = z*69069u + 1u;
  if (z == 0u) z = 12345671u;
  return (double)z*(1.0/4294967296.0);
}
// gamma(shape, 1) variates for the populations of `list` (4 bits each, n entries) whose bit is set in `want`, from the global
// stream in list order, exactly the numbers a00_bpp_rndgamma (legacy_rndgamma, random.c:240-275: Marsaglia-Tsang on the
// polar normal) gives one after the other — but side by side: WHICH uniforms a variate takes is settled by cheap
// arithmetic alone (a polar pair is taken when s = u^2 + v^2 lies in (0, 1); then one uniform for the test) as long as
// every variate passes its test at the first round, so the wave walks the stream through all of them first and
// lane p then does population p's square root, logs and test.  A variate that does not pass (about one draw in a
// hundred) sends everybody back to the start of the block and through the draws one after the other.
// shape: lane p.  xarg / xlog: a number per lane >= 16 whose log rides along in the same call.  Returns the variate on lane p.
__device__ __forceinline__ double draw_gammas(uint32_t & zz, uint32_t lane, unsigned long long list, int n, uint32_t want, double shape, double xarg, double & xlog)
{
  const uint32_t z0 = zz;
  uint32_t z = z0;
  double mu = 0, ms = 0.5, m3 = 0; bool scan_ok = true;
  for (int i = 0; i < n; ++i)
  {
    const uint32_t p = (uint32_t)(list >> (4*i)) & 15u;
    if (!((want >> p) & 1u)) continue;
    double u = 0, s2 = 0; bool got = false;
    for (int rd = 0; rd < 64 && !got; ++rd)
    {
      u = 2*prog_lcg(z) - 1; const double v = 2*prog_lcg(z) - 1;
      s2 = u*u + v*v;
      got = s2 > 0 && s2 < 1;
    }
    scan_ok = scan_ok && got;
    const double u3 = prog_lcg(z);
    if (lane == p) { mu = u; ms = s2; m3 = u3; }
  }
  const bool mine = lane < 16u && ((want >> (lane & 15u)) & 1u);
  const double d = shape - 1.0/3.0, c = (1.0/3.0)/sqrt(d);
  const double L = log(lane < 16u ? ms : xarg);
  xlog = L;
  double g = prog_qnan(); bool ok = true;
  if (mine)
  {
    const double x = mu*sqrt(-2*L/ms);
    double v = 1.0 + c*x;
    ok = v > 0 && shape >= 1;
    v *= v*v;
    if (ok && !(m3 < 1 - 0.0331*x*x*x*x)) ok = log(m3) < 0.5*x*x + d*(1 - v + log(v));
    v *= d;
    if (v == 0) v = 1E-300;
    g = v;
  }
  if (!scan_ok || __any(mine && !ok))
  {
    z = z0;
    for (int i = 0; i < n; ++i)
    {
      const uint32_t p = (uint32_t)(list >> (4*i)) & 15u;
      if (!((want >> p) & 1u)) continue;
      const double gi = a00_bpp_rndgamma(&z, __shfl(shape, (int)p, 64));
      if (lane == p) g = gi;
    }
  }
  zz = z;
  return g;
}
// A re-drawn theta's part of ln(acceptance ratio) (tau_step / mix_step of a00_driver.c; stree.c:5840-5990, prop_mixing.c:272-425), lane p < 16 for population p:
//   [invgamma(theta | old fit) - invgamma(theta' | new fit)] + [gamma prior ratio] + [k (log 2/theta' - log 2/theta) - (T'/theta' - T/theta)]
// Its four logs and four quotients are taken side by side: lane p + 16 m computes piece m.  l2t_new = log(2/theta') comes out too.
__device__ __forceinline__ double redraw_ratio(const WgBase & wg, uint32_t lane, double kk, double tn, double a1, double b1, double c1, double Tn,
                                               double ao, double bo, double co, double Told, double & l2t_new)
{
  const int p = (int)(lane & 15u); const uint32_t role = lane >> 4;
  const double tn_p = __shfl(tn, p, 64), to_p = wg.tau[MAXPOP + (p < MAXPOP ? p : 0)];
  // (every shuffle by every lane: a lane that sits out a branch hands nothing over)
  const double s_b1 = __shfl(b1, p, 64), s_Tn = __shfl(Tn, p, 64), s_To = __shfl(Told, p, 64), s_bo = __shfl(bo, p, 64);
  const double num = role == 0u ? s_b1 : role == 1u ? s_Tn : role == 2u ? s_To : s_bo;
  const double q = role == 1u ? tn_p/to_p : 2.0/tn_p;
  const double L = log(role == 0u ? tn_p : role == 3u ? to_p : q);         // log theta' | log(theta'/theta) | log(2/theta') | log theta
  const double r = num/((role & 2u) ? to_p : tn_p);                        // b'/theta'  | T'/theta'         | T/theta       | b/theta
  const double L0 = __shfl(L, p, 64), L1 = __shfl(L, 16 + p, 64), L2 = __shfl(L, 32 + p, 64), L3 = __shfl(L, 48 + p, 64);
  const double r0 = __shfl(r, p, 64), r1 = __shfl(r, 16 + p, 64), r2 = __shfl(r, 32 + p, 64), r3 = __shfl(r, 48 + p, 64);
  l2t_new = L2;
  const double l2t_old = wg.tau[2*MAXPOP + (p < MAXPOP ? p : 0)];
  const double anew = (c1 + (-a1 - 1)*L0) - r0, aold = (co + (-ao - 1)*L3) - r3;
  return (aold - anew) + ((wg.sp.theta_alpha - 1)*L1 - wg.sp.theta_beta*(tn - to_p)) + (kk*(L2 - l2t_old) - (r1 - r2));
}

// THETA (theta_step_gibbs of a00_driver.c): the sums the exchange brought (xtot[0 .. 2 n): k_p, T_p of the populations of the
// mask, in order) -> pf; the fits of all thetas side by side; the Gibbs variates (populations outside slidem; a sliding
// one proposes tslide, lane p); ln of the acceptance ratios (a00_theta_lnacc + a00_theta_gibbs_hastings); the acceptance
// numbers in population order, drawn only when needed.  Leaves dec.accm / tn / l2t / lnacc; apply_now: and the accepted
// thetas in tau[] (a TAU decision follows at once).  z: the global stream.
// tau_q >= 0: the fits the TAU decision that follows at once will want (populations tau_q and its children against the sums
// xtot[tau_base + 2 .. 4]) are made in the same call, on lanes 16 + p, and left in rd[p].a / .b.
__device__ __forceinline__ uint32_t prog_theta_decide(uint32_t z, uint32_t theta_mask, uint32_t slidem, double tslide, int apply_now, int tau_q = -1, int tau_base = 0)
{
  WgBase & wg = wg_base();
  const uint32_t lane = threadIdx.x & 63u, pl16 = lane & 15u, role = lane >> 4;
  const int npop = wg.sp.npop;
  const double qnan = prog_qnan();
  const bool mine = lane < (uint32_t)npop && ((theta_mask >> pl16) & 1u);
  double runK = 0, runT = 0, fitA = qnan, fitB = qnan, fitC = qnan;
  {
    const int kx = __popc(theta_mask & ((1u << pl16) - 1u));
    runK = mine ? wg.xtot[(2*kx) & 31] : 0.0; runT = mine ? wg.xtot[(2*kx + 1) & 31] : 0.0;
  }
  const bool run_ok = !__any(mine && !(runK == runK && runT == runT));      // (an unusable term anywhere: every decision is a rejection, nothing drawn)
  double tn = mine ? ((slidem >> pl16) & 1u ? tslide : qnan) : 0.0, lnacc_p = qnan, e_p = 0, l2_p = 0;
  if (run_ok)
  {
    {
      // lane p: THETA's fit (k_p, T_p); lane 16 + p: the coming TAU's (k_p, T'_p)
      const int tcl = tau_q >= 0 ? (int)wg.sp.left[tau_q] : -1, tcr = tau_q >= 0 ? (int)wg.sp.right[tau_q] : -1;
      const bool taff = role == 1u && tau_q >= 0 && ((int)pl16 == tau_q || (int)pl16 == tcl || (int)pl16 == tcr) && ((theta_mask >> pl16) & 1u);
      const double kk = __shfl(runK, (int)pl16, 64);
      const double Cn = taff ? wg.xtot[tau_base + ((int)pl16 == tau_q ? 2 : (int)pl16 == tcl ? 3 : 4)] : qnan;
      double fa = qnan, fb = qnan;
      if (mine || (taff && Cn == Cn)) a00_theta_conditional_invgamma_fast(wg.sp.theta_alpha, wg.sp.theta_beta, (long)kk, mine ? runT : Cn, &fa, &fb);
      if (mine) { fitA = fa; fitB = fb; }
      if (role == 1u) { wg.rd[pl16].a = fa; wg.rd[pl16].b = fb; }
    }
    const uint32_t fitm = (uint32_t)__ballot(mine && fitA == fitA) & 0xffffu;
    const uint32_t gm = theta_mask & ~slidem & fitm;
    double xl;
    const double s_fa = __shfl(fitA, (int)pl16, 64), s_fb = __shfl(fitB, (int)pl16, 64);
    const double g = draw_gammas(z, lane, 0xfedcba9876543210ull, npop, gm, fitA, role == 1u ? s_fb : s_fa, xl);
    {
      const double lb = __shfl(xl, 16 + (int)pl16, 64), la = __shfl(xl, 32 + (int)pl16, 64);
      if (lane < 16u && fitA == fitA) fitC = fitA*lb - lgamma_with_log(fitA, la);
    }
    if (lane < 16u && ((gm >> pl16) & 1u)) tn = 1/(g/fitB);
    // ln of the acceptance ratio: lane p + 16 m takes piece m
    {
      const int p = (int)pl16;
      const double tn_p = __shfl(tn, p, 64), to_p = wg.tau[MAXPOP + (p < MAXPOP ? p : 0)], T_p = __shfl(runT, p, 64);
      const double q = role == 0u ? 2.0/tn_p : role == 1u ? tn_p/to_p : to_p/tn_p;
      const double L = log(q);                                             // log(2/theta') | log(theta'/theta) | log(theta/theta')
      const double r = (role < 2u ? T_p : 1.0)/((role & 1u) ? to_p : tn_p);    // T/theta' | T/theta | 1/theta' | 1/theta
      const double L2 = __shfl(L, p, 64), L1 = __shfl(L, 16 + p, 64), L3 = __shfl(L, 32 + p, 64);
      const double r0 = __shfl(r, p, 64), r1 = __shfl(r, 16 + p, 64), r2 = __shfl(r, 32 + p, 64), r3 = __shfl(r, 48 + p, 64);
      const double l2t_old = wg.tau[2*MAXPOP + (p < MAXPOP ? p : 0)];
      if (mine && tn == tn)
      {
        lnacc_p = (runK*(L2 - l2t_old) - (r0 - r1)) + ((wg.sp.theta_alpha - 1)*L1 - wg.sp.theta_beta*(tn - to_p));
        if ((gm >> pl16) & 1u) lnacc_p += (-fitA - 1)*L3 - fitB*(r3 - r2);
      }
      e_p = exp(lnacc_p);
      l2_p = L2;
    }
  }
  uint32_t accm = 0;
  for (int p = 0; p < npop; ++p)
    if ((theta_mask >> p) & 1u)
    {
      const double la_ = __shfl(lnacc_p, p, 64), tn_ = __shfl(tn, p, 64), ep_ = __shfl(e_p, p, 64);
      bool acc = la_ == la_ && tn_ > 0;
      if (acc && !(la_ >= -1e-10)) acc = prog_lcg(z) < ep_;
      accm |= acc ? 1u << p : 0u;
    }
  if (lane < 16u)
  {
    PopFit & f = wg.pf[lane];
    f.k = runK; f.T = runT; f.a = fitA; f.b = fitB; f.c = fitC;
    wg.dec.tn[lane] = tn == tn ? tn : wg.tau[MAXPOP + (lane < (uint32_t)MAXPOP ? lane : 0u)];
    wg.dec.lnacc[lane] = lnacc_p; wg.dec.l2t[lane] = l2_p;
    if (apply_now && ((accm >> lane) & 1u)) { wg.tau[MAXPOP + lane] = tn; wg.tau[2*MAXPOP + lane] = l2_p; }
  }
  if (lane == 0) { wg.dec.accm = accm; wg.dec.run_ok = run_ok ? 1u : 0u; }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  return z;
}

// TAU of population q (tau_step of a00_driver.c, the program's form): the exchange brought the loci's likelihood change
// (xtot[base], + the coarse companion) and the new T2h sums of q and its two children (xtot[base + 2 .. 4]); each of their
// thetas is re-drawn from the fit to (k, new sum) and enters the ratio against the fit to the current sums (pf).
// lnacc0 = the window's prior term.  Leaves dec.acc_step / rd_mask / lnacc_step and rd[] (installed by the caller on acceptance).
__device__ __forceinline__ uint32_t prog_tau_decide(uint32_t z, uint32_t theta_mask, int q, int base, double lnacc0, bool prefit = false)
{
  WgBase & wg = wg_base();
  const uint32_t lane = threadIdx.x & 63u, pl16 = lane & 15u, role = lane >> 4;
  const double qnan = prog_qnan();
  constexpr double FX = 1099511627776.0, FXC = 1024.0;
  double lnacc = (wg.xtot[base] + wg.xtot[base + 1]*(FX/FXC)) + lnacc0;
  const int cl = wg.sp.left[q], cr = wg.sp.right[q];
  const bool run_ok = wg.dec.run_ok != 0u;
  const PopFit f = wg.pf[pl16];
  const bool aff = lane < 16u && ((int)lane == q || (int)lane == cl || (int)lane == cr);
  const bool have = aff && ((theta_mask >> pl16) & 1u) && run_ok;
  const double Cn = have ? wg.xtot[base + ((int)lane == q ? 2 : (int)lane == cl ? 3 : 4)] : qnan;
  double fa = qnan, fb = qnan, tn = qnan, l2t = 0;
  if (prefit) { if (have && Cn == Cn) { fa = wg.rd[pl16].a; fb = wg.rd[pl16].b; } }           // (made with THETA's: prog_theta_decide)
  else if (have && Cn == Cn) a00_theta_conditional_invgamma_fast(wg.sp.theta_alpha, wg.sp.theta_beta, (long)f.k, Cn, &fa, &fb);
  const uint32_t rd_mask = (uint32_t)__ballot(have && fa == fa && f.a == f.a) & 0xffffu;
  double xl;
  const double s_fa = __shfl(fa, (int)pl16, 64), s_fb = __shfl(fb, (int)pl16, 64);
  const double g = draw_gammas(z, lane, (unsigned long long)q | ((unsigned long long)cl << 4) | ((unsigned long long)cr << 8), 3, rd_mask, fa, role == 1u ? s_fb : s_fa, xl);
  const double lb = __shfl(xl, 16 + (int)pl16, 64), la = __shfl(xl, 32 + (int)pl16, 64);
  const double c1 = fa*lb - lgamma_with_log(fa, la);
  if (lane < 16u && ((rd_mask >> pl16) & 1u)) tn = 1.0/(g/fb);
  const double x = redraw_ratio(wg, lane, f.k, tn, fa, fb, c1, Cn, f.a, f.b, f.c, f.T, l2t);
  for (int j = 0; j < 3; ++j)
  {
    const int p = j == 0 ? q : j == 1 ? cl : cr;
    if (!((theta_mask >> p) & 1u)) continue;
    const double xp = __shfl(x, p, 64);
    lnacc += run_ok && ((rd_mask >> p) & 1u) ? xp : qnan;
  }
  const bool accept = lnacc >= -1e-10 || prog_lcg(z) < exp(lnacc);        // (Stream<true>::accept)
  if (lane < 16u) { Redraw & r = wg.rd[lane]; r.tn = tn; r.l2t = l2t; r.T = Cn; r.a = fa; r.b = fb; r.c = c1; }
  if (lane == 0) { wg.dec.acc_step = accept ? 1u : 0u; wg.dec.rd_mask = rd_mask; wg.dec.lnacc_step = lnacc; }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  return z;
}

// MIX: every theta from the fit to its conditional given the SCALED trees (k, c T) (mix_step of a00_driver.c;
// prop_mixing.c: Cjstar / c) — nothing of it depends on the loci's sums, so the caller runs it between its workgroup's
// arrival at the exchange and the totals'.  Leaves rd[], dec.rd_mask and dec.lnacc_theta.
__device__ __forceinline__ uint32_t prog_mix_redraw(uint32_t z, uint32_t theta_mask, double mix_c)
{
  WgBase & wg = wg_base();
  const uint32_t lane = threadIdx.x & 63u, pl16 = lane & 15u, role = lane >> 4;
  const int npop = wg.sp.npop, pm = (int)pl16;
  const double qnan = prog_qnan();
  const bool run_ok = wg.dec.run_ok != 0u;
  const PopFit f = wg.pf[pl16];
  // lane p: the fit to the scaled trees of population p, lane 16 + p: to the current ones
  const bool have = lane < 32u && pm < npop && ((theta_mask >> pm) & 1u) && run_ok;
  const double Ts = f.T*mix_c;
  double fa = qnan, fb = qnan, tn = qnan, l2t = 0;
  if (have) a00_theta_conditional_invgamma_fast(wg.sp.theta_alpha, wg.sp.theta_beta, (long)f.k, role == 0u ? Ts : Ts/mix_c, &fa, &fb);
  const double fao = __shfl(fa, 16 + pm, 64), fbo = __shfl(fb, 16 + pm, 64);
  const uint32_t rd_mask = (uint32_t)__ballot(lane < 16u && have && fa == fa && fao == fao) & 0xffffu;
  double xl;
  const double s_fa = __shfl(fa, pm, 64), s_fb = __shfl(fb, pm, 64);
  const double g = draw_gammas(z, lane, 0xfedcba9876543210ull, npop, rd_mask, fa, role == 1u ? s_fb : role == 2u ? s_fa : fbo, xl);
  const double lb = __shfl(xl, 16 + pm, 64), la = __shfl(xl, 32 + pm, 64), lbo = __shfl(xl, 48 + pm, 64), lao = log(fao);
  const double c1 = fa*lb - lgamma_with_log(fa, la), co = fao*lbo - lgamma_with_log(fao, lao);
  if (lane < 16u && ((rd_mask >> pl16) & 1u)) tn = 1.0/(g/fb);
  const double x = redraw_ratio(wg, lane, f.k, tn, fa, fb, c1, Ts, fao, fbo, co, f.T, l2t);
  double lnacc_theta = 0;
  for (int p = 0; p < npop; ++p)
    if ((theta_mask >> p) & 1u)
    {
      const double xp = __shfl(x, p, 64);
      lnacc_theta += run_ok && ((rd_mask >> p) & 1u) ? xp : qnan;
    }
  if (lane < 16u) { Redraw & r = wg.rd[lane]; r.tn = tn; r.l2t = l2t; r.T = Ts; r.a = fa; r.b = fb; r.c = c1; }
  if (lane == 0) { wg.dec.rd_mask = rd_mask; wg.dec.lnacc_theta = lnacc_theta; }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  return z;
}

template <int NT, bool BPP, bool PROG = false>
__global__ void __launch_bounds__(Cfg<NT>::BS) iter_kernel(const Args A)
{
  using C = Cfg<NT>;
  constexpr int G = C::G, LPW = C::LPW, NN = C::NN, W = C::W, NBUF = C::NBUF, NPM = C::NPM, WAVES = C::WAVES;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  WgLDS<NT> & wg = *reinterpret_cast<WgLDS<NT> *>(smem);
  WaveLDS<NT> * wl_all = reinterpret_cast<WaveLDS<NT> *>(smem + ((sizeof(WgLDS<NT>) + 15) & ~(size_t)15));
  const uint32_t tid = threadIdx.x, wv = tid >> 6, lane = tid & 63u, b = blockIdx.x;
  const int li = (int)(lane & (uint32_t)(G - 1)); const uint32_t gbase = lane - (uint32_t)li, slot = lane/(uint32_t)G;
  // (PROG: wave 0 is the control wave — no loci, no per-wave block)
  WaveLDS<NT> & wl = wl_all[PROG ? (wv ? wv - 1u : 0u) : wv];
  Slot<NT> & S = wl.slot[slot];
  const uint32_t lw = PROG ? wv - 1u : wv;                       // this wave's place among the workgroup's waves of loci (PROG, wave 0: none)
  const uint32_t gw = lw < A.lwaves ? b*A.lwaves + lw : 0xffffffffu;       // global wave of loci
  {
    const uint32_t * src = reinterpret_cast<const uint32_t *>(A.sp);
    uint32_t * dst = reinterpret_cast<uint32_t *>(&wg.sp);
    for (uint32_t i = tid; i < sizeof(Species)/4; i += C::BS) dst[i] = src[i];
    if (tid < 24u) wg.prof[tid] = 0;
    if (tid < 16u) wg.wsweep[tid] = 0;
  }
  __syncthreads();
  const Species & SP = wg.sp;
  const int npop = SP.npop, nsp = SP.S;

  // ---- species tree: the workgroup's copy of the parameters, the topology per lane
  for (uint32_t i = tid; i < (uint32_t)(3*MAXPOP); i += C::BS) wg.tau[i] = A.taus[i];
  for (uint32_t i = tid; i < (uint32_t)((2*NT)*(2*NT)); i += C::BS) wg.lograt[i] = A.lograt[(i/(2*NT))*MAXN + i % (2*NT)];
  if (tid < 16u) wg.anc[tid] = tid < (uint32_t)MAXPOP ? (uint32_t)SP.anc[tid] : 0u;
  // an earlier launch of the stream that gave up (Args::err, cleared by the host when it has dealt with it) voids this one as
  // well: the iterations then run again IN ORDER, from the state and the random numbers the first of them started from
  if (tid == 0) { wg.abort_ = __hip_atomic_load(A.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ? 2u : 0u; wg.late_ = 0; wg.bad_ = 0; wg.xbad_ = 0; wg.coarse_ = 0; wg.xcoarse_ = 0; }
  if (tid < 32u) wg.accfx[tid] = 0ull;
  for (uint32_t i = tid; i < 2u*XN; i += C::BS) (&wg.xprev[0][0])[i] = 0ull;
  PopLane pl;
  {
    pl.parent = li < npop ? (int)SP.parent[li < MAXPOP ? li : 0] : -1;
    pl.anc = li < npop ? (uint32_t)SP.anc[li < MAXPOP ? li : 0] : 0u;
    uint32_t below = 0;
    for (int q2 = 0; q2 < npop; ++q2) if (q2 != li && (((uint32_t)SP.anc[q2] >> li) & 1u)) below |= 1u << q2;
    pl.below = li < npop ? below : 0u;
  }
  __syncthreads();
  if (wg.abort_ == 2u)
  {
    if (b == 0 && tid == 0) (void)atomicAdd(A.err + 1, (int)A.niter);
    return;
  }
  auto load_pop = [&]()
  {
    const int i = li < MAXPOP ? li : 0;
    pl.tau = wg.tau[i]; pl.theta = wg.tau[MAXPOP + i]; pl.l2t = wg.tau[2*MAXPOP + i];
    pl.ptau = pl.parent >= 0 ? wg.tau[pl.parent] : -1.0;
  };
  load_pop();
  Stream<BPP> grng{*A.grng};

  const bool prof_on = (A.dbg & 16u) && b == 0 && tid == 0;
  long long pf_t = prof_on ? clock64() : 0;
#define SMP2_TICK(i_) do { if (prof_on) { const long long t1_ = clock64(); wg.prof[i_] += t1_ - pf_t; pf_t = t1_; } } while (0)
  long long pf_s = 0;
#define SMP2_SUB0() do { if (prof_on) pf_s = clock64(); } while (0)
#define SMP2_SUB(i_) do { if (prof_on) wg.prof[i_] += clock64() - pf_s; } while (0)

  // ---- the sum over ALL loci of one all-loci step's terms.  A term enters as 2^-40 fixed point (what the host driver adds
  // up in doubles, locus by locus: the totals agree to ~1e-11), so a total does not depend on the order of the additions: the lanes add theirs to the workgroup's accumulators (LDS atomics, fx_add),
  // the workgroup adds those to the step's device accumulators (device-scope atomics) and then bumps the arrival
  // counter; wave 0 polls until every workgroup has arrived.  Device accumulators and counter only ever grow (the host
  // zeroes them before the launch) and two sets alternate, so a workgroup already in the next step never touches what
  // a slower one still reads.  Up to 7 values share one 64-byte block with the counter: the poll's ONE load brings the
  // totals with the count (they landed before the arrival was counted).  False: timed out.
  constexpr double FX = 1099511627776.0;             // 2^40: 9e-13 per term; |term| < 256 and <= 2^14 loci: the sum cannot wrap
  constexpr double FXC = 1024.0;                     // a TAU / MIX term beyond that goes to a coarse companion sum (2^-10, |term| < 2^38)
  auto fx_add = [&](int v, double x, bool coarse)
  {
    if (fabs(x) < 256.0)
      (void)__hip_atomic_fetch_add(&wg.accfx[v], (unsigned long long)__double2ll_rn(x*FX), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else if (coarse && fabs(x) < 274877906944.0)
    {
      wg.coarse_ = 1u;
      (void)__hip_atomic_fetch_add(&wg.accfx[v + 1], (unsigned long long)__double2ll_rn(x*FXC), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    else wg.bad_ = 1u;                                                       // (also NaN): the step is rejected
  };
  uint32_t nx = 0;
  unsigned long long gseq = A.seq0;                 // several GPUs: the mailboxes' sequence number
#define XT(i_) do { if (prof_on) { const long long t1_ = clock64(); wg.prof[8 + i_] += t1_ - xt0; xt0 = t1_; } } while (0)
  long long xt0 = 0;
  // one block of <= 15 values (15 sums + the counter = one 128-byte block): the workgroup's sums go to its shard, then its arrival
  // solo (the program-moves kernel): the control wave runs the exchange alone — no workgroup barrier inside
  constexpr bool solo = PROG;
  auto xpush = [&](int v0, int nv)
  {
    const uint32_t par = nx & 1u; ++nx;
    // 8 shards, a workgroup adds to shard b mod 8: atomics on one word are served one after the other
    unsigned long long * acc = A.xbuf + (size_t)par*XN + (size_t)(b & 7u)*16u;
    if (tid < (uint32_t)nv)
    {
      const unsigned long long fx = wg.accfx[v0 + (int)tid];
      wg.accfx[v0 + (int)tid] = 0ull;
      const unsigned long long old = __hip_atomic_fetch_add(acc + tid, fx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" :: "v"(old) : "memory");          // the sums have landed before the arrival is counted
    }
    XT(1);
    if (!solo) __syncthreads();
    XT(2);
    // arrival: + 1, and + 2^32 when a term of this workgroup was unusable
    if (tid == 0) (void)__hip_atomic_fetch_add(acc + XV, 1ull + ((unsigned long long)wg.bad_ << 32) + ((unsigned long long)wg.coarse_ << 48), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  // ... and the wait for everybody's: wave 0 polls, the totals go to wg.xtot[v0 ..]; last = the exchange's last block.  False: timed out
  auto xpoll = [&](int v0, int nv, bool last, bool hold) -> bool
  {
    const uint32_t par = (nx - 1u) & 1u;
    unsigned long long * set = A.xbuf + (size_t)par*XN;
    if (wv == 0)
    {
      const unsigned long long t_wait = wall_clock64();
      const unsigned long long prev0 = wg.xprev[par][lane], prev1 = wg.xprev[par][64u + lane];
      bool ok = true;
      unsigned long long cur0 = 0, cur1 = 0, d = 0, gd = 0;
      for (uint32_t rounds = 1;; ++rounds)
      {
        // TWO loads: lane 16 x + k reads word k of shards x and x + 4; the shards' growth since the set's previous use, added up
        cur0 = __hip_atomic_load(set + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        cur1 = __hip_atomic_load(set + 64u + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        d = (cur0 - prev0) + (cur1 - prev1);
        d += __shfl_xor(d, 16, 64); d += __shfl_xor(d, 32, 64);
        // (test switches, BPA_SMP_INJECT: dbg bit 1024 = every workgroup gives up at its first wait, 2048 = workgroup 0 alone — the
        //  others then run into the real time-out at the next exchange)
        if ((A.dbg & 1024u) || ((A.dbg & 2048u) && b == 0)) { ok = false; break; }
        if ((uint32_t)__shfl(d, XV, 64) >= A.nwg) break;
        if ((rounds & 63u) == 0 && wall_clock64() - t_wait > 50000000ull) { ok = false; break; }     // 0.5 s at 100 MHz
        __builtin_amdgcn_s_sleep(1);
      }
      bool anybad = ok && ((__shfl(d, XV, 64) >> 32) & 0xffffull) != 0;
      const bool anycoarse = ok && (__shfl(d, XV, 64) >> 48) != 0;     // (one rank's; several ranks: each reports its own)
      if (ok && A.world > 1)
      {
        // ---- several GPUs: this rank's sums (lanes 0..14) and its unusable-term flag (lane 15) go to slot `rank` of
        // EVERY rank's mailbox over the xGMI peer mappings — workgroup 0 publishes, values first, then the sequence
        // flag —, and every workgroup adds up the N slots of its own mailbox once their flags show this exchange.
        // Fixed point: the same total on every rank whatever the order.  Mailboxes alternate by sequence parity.
        ++gseq;
        const size_t slot = ((size_t)(gseq & 1ull)*(size_t)A.world)*A.slot_bytes;
        if (b == 0)
        {
          const unsigned long long v = lane < (uint32_t)nv ? d : (lane == (uint32_t)XV && anybad) ? 1ull : 0ull;
          if (lane < 16u)
            for (int pr_ = 0; pr_ < A.world; ++pr_)
              __hip_atomic_store(reinterpret_cast<unsigned long long *>(A.peers[pr_] + slot + (size_t)A.rank*A.slot_bytes + p2p::HDR) + lane, v,
                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          __threadfence_system();
          __builtin_amdgcn_wave_barrier();
          if (lane < (uint32_t)A.world)
            __hip_atomic_store(reinterpret_cast<unsigned long long *>(A.peers[lane] + slot + (size_t)A.rank*A.slot_bytes), gseq,
                               __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        bool here = true;
        if (lane < (uint32_t)A.world)
        {
          const unsigned long long * f = reinterpret_cast<const unsigned long long *>(A.mail + slot + (size_t)lane*A.slot_bytes);
          const unsigned long long tw = wall_clock64();
          while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != gseq)
          {
            if (wall_clock64() - tw > A.spin_limit) { here = false; break; }
            __builtin_amdgcn_s_sleep(2);
          }
        }
        if (!__all(here ? 1 : 0)) { ok = false; if (lane == 0) *A.p2p_err = 1; }
        else
        {
          unsigned long long tot = 0;
          for (int r = 0; r < A.world; ++r)
            tot += __hip_atomic_load(reinterpret_cast<const unsigned long long *>(A.mail + slot + (size_t)r*A.slot_bytes + p2p::HDR) + (lane & 15u),
                                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          gd = tot; anybad = __shfl(tot, XV, 64) != 0;
        }
      }
      XT(3);
      if (ok)
      {
        const unsigned long long dd = A.world > 1 ? gd : d;
        if (lane < (uint32_t)nv) wg.xtot[v0 + (int)lane] = anybad ? __longlong_as_double(0x7ff8000000000000ll) : (double)(long long)dd*(1.0/FX);
        wg.xprev[par][lane] = cur0; wg.xprev[par][64u + lane] = cur1;
        // (an unusable term stays flagged through every block of the exchange: its value may lie in a later one)
        if (lane == 0) { if (anybad) wg.xbad_ = 1u; wg.xcoarse_ = anycoarse ? 1u : 0u; if (last) { wg.bad_ = 0; wg.coarse_ = 0; } }
      }
      else if (lane == 0) { wg.abort_ = 1; (void)__hip_atomic_exchange(A.err, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); if (b == 0) (void)atomicAdd(A.err + 1, (int)A.niter); }
      if (hold || solo) { wsync(); return ok; }                // (wave 0 goes on to the decision; the caller's barrier publishes everything)
    }
    else if (hold || solo) return true;
    __syncthreads();
    XT(4);
    return !wg.abort_;
  };
  // an exchange of nval sums: begin = this workgroup's part (its first block is on its way when this returns), end = the wait
  // (and the blocks after the first, one after the other: two accumulator sets alternate).  What lies between the two
  // runs while the slower workgroups are still at their loci.  If any term of any block was unusable, EVERY total is NaN.
  int x_nval = 0;
  auto exchange_begin = [&](int nval)
  {
    xt0 = prof_on ? clock64() : 0;
    if (!solo) __syncthreads();                      // every lane's term is in wg.accfx
    XT(0);
    x_nval = nval;
    if (tid == 0) wg.xbad_ = 0;
    xpush(0, nval < XV ? nval : XV);
  };
  // hold: wave 0 returns from the last block's poll without the closing barrier (and is the only one that may read the
  // totals before the caller's own barrier); the other waves come straight through
  auto exchange_end = [&](int want, double & mine_tot, bool hold = false) -> bool
  {
    const int nval = x_nval;
    xt0 = prof_on ? clock64() : 0;
    if (!xpoll(0, nval < XV ? nval : XV, nval <= XV, hold && nval <= XV)) return false;
    for (int v0 = XV; v0 < nval; v0 += XV)
    {
      const int nv = nval - v0 < XV ? nval - v0 : XV;
      xpush(v0, nv);
      if (!xpoll(v0, nv, v0 + XV >= nval, hold && v0 + XV >= nval)) return false;
    }
    if (hold || solo)
    {
      if (wv == 0 && nval > XV && wg.xbad_) { if (lane < (uint32_t)nval) wg.xtot[lane] = __longlong_as_double(0x7ff8000000000000ll); wsync(); }
      return true;
    }
    if (nval > XV && wg.xbad_)
    {
      __syncthreads();
      if (tid < (uint32_t)nval) wg.xtot[tid] = __longlong_as_double(0x7ff8000000000000ll);
      __syncthreads();
    }
    mine_tot = wg.xtot[want & 31];
    return true;
  };
  auto exchange = [&](int nval, int want, double & mine_tot) -> bool
  {
    exchange_begin(nval);
    return exchange_end(want, mine_tot);
  };
#undef XT

  // =====================================================================================================================
  // The program's moves (PROG): wave 0 of every workgroup has no loci — it is the CONTROL wave.  It owns the global stream,
  // makes the proposal of the species tree of every all-loci step (window, factors -> wg.stepp), runs the exchange alone
  // (the workgroup's sums -> device accumulators -> everybody's totals), takes the decision (prog_theta_decide /
  // prog_tau_decide, MIX here) and installs its consequences for the species tree (wg.tau, wg.pf) — all between the two
  // barriers of a step: B1 "the loci's terms are in wg.accfx" and B3 "the decision is out" (+ B4 "the next proposal is out",
  // which it makes while the loci settle the decision).
  // While the loci waves work it does what does not depend on them (MIX's re-draws).  Its registers are its own: none of a
  // locus's state is alive here, none of the decisions' arithmetic in the loci waves' code (one code path cost the
  // sweep 230 spilled registers).  Stream order = a00_driver.c's: [first TAU's window] [THETA: choices + windows]
  // [THETA: variates, acceptance numbers] [TAU: variates, acceptance number] [next window] ...
  if constexpr (PROG)
  {
    if (wv == 0)
    {
      Stream<true> g{*A.grng};
      const uint32_t tm = A.theta_mask;
      const int nth = 2*__popc(tm);
      const double qnan = __longlong_as_double(0x7ff8000000000000ll);
      uint32_t cnt_prop = 0, cnt_acc = 0, cnt_gprop = 0, cnt_gacc = 0, ndec = 0;
      uint32_t pj_tau = 0, pj_tau_acc = 0, pj_mix = 0, pj_mix_acc = 0;
      const bool declog = (A.dbg & 256u) && b == 0;
      int q = -1; bool mix = false;
      double tq_old = 0, tq_new = 0, mix_c = 1, mix_lnc = 0, lnprior = 0;
      // the species-tree proposal of step `stepq` (TAU of population stepq; MIX: npop), from the current wg.tau
      auto make_step = [&](int stepq)
      {
        mix = stepq == npop; q = mix ? -1 : stepq;
        const double wprop = g.window();                         // (log c of the mixing step: finetune x the window variate, prop_mixing.c:300)
        double tq_lo = 0, tq_hi = 0, minf = 1, maxf = 1, lminf = 0, lmaxf = 0;
        tq_old = 0; tq_new = 0; mix_c = 1; mix_lnc = 0; lnprior = 0;
        if (!mix)
        {
          const int pq = SP.parent[q], cl = SP.left[q], cr = SP.right[q];
          tq_old = wg.tau[q]; tq_lo = fmax(wg.tau[cl], wg.tau[cr]); tq_hi = pq >= 0 ? wg.tau[pq] : 999.0;
          tq_new = reflect(tq_old + SP.ft_tau*wprop, tq_lo, tq_hi);
          minf = (tq_new - tq_lo)/(tq_old - tq_lo); maxf = (tq_new - tq_hi)/(tq_old - tq_hi);
          // (the three logs side by side: lane 0, 1, 2)
          const double lg = log(lane == 0 ? minf : lane == 1u ? maxf : tq_new/tq_old);
          lminf = __shfl(lg, 0, 64); lmaxf = __shfl(lg, 1, 64);
          if (pq < 0 && SP.tau_alpha > 0) lnprior = (SP.tau_alpha - 1 - (nsp - 1) + 1)*__shfl(lg, 2, 64) - SP.tau_beta*(tq_new - tq_old);
        }
        else { mix_lnc = SP.ft_mix*wprop; mix_c = exp(mix_lnc); }
        if (lane == 0)
        {
          wg.stepp.q = q; wg.stepp.mix = mix ? 1 : 0; wg.stepp.tq_old = tq_old; wg.stepp.tq_lo = tq_lo; wg.stepp.tq_hi = tq_hi; wg.stepp.tq_new = tq_new;
          wg.stepp.minf = minf; wg.stepp.maxf = maxf; wg.stepp.lminf = lminf; wg.stepp.lmaxf = lmaxf; wg.stepp.mix_c = mix_c; wg.stepp.mix_lnc = mix_lnc;
        }
      };
      // THETA's choices: which thetas slide (and where to), which take the Gibbs draw
      uint32_t slidem = 0; double tslide = 0;
      auto theta_choices = [&]()
      {
        slidem = 0; tslide = 0;
        for (int p = 0; p < npop; ++p)
          if ((tm >> p) & 1u)
          {
            if (!(g.u() < SP.theta_slide_prob)) continue;
            slidem |= 1u << p;
            const double tn = reflect(wg.tau[MAXPOP + p] + SP.ft_theta*g.window(), 0.0, 999.0);
            if (p == (int)lane) tslide = tn;
          }
      };
      bool aborted = false;
      if (A.do_allloci && A.niter) { make_step(nsp); theta_choices(); }
      __syncthreads();                                                  // B0: the first step's proposal is out
      for (uint32_t it = 0; it < A.niter && A.do_allloci && !aborted; ++it)
      {
        for (int stepq = nsp; stepq <= npop && !aborted; ++stepq)
        {
          const bool first = stepq == nsp;                           // (the THETA step's sums come with the first TAU's)
          const int base = first ? nth : 0;
          // while the loci work: what does not depend on them
          if (mix) { SMP2_SUB0(); g.r = prog_mix_redraw((uint32_t)g.r, tm, mix_c); SMP2_SUB(15); }
          SMP2_TICK(0);
          __syncthreads();                                              // B1: every locus's terms are in wg.accfx
          SMP2_TICK(1);
          double dummy = 0;
          exchange_begin(base + (mix ? 2 : 5));
          SMP2_TICK(2);
          const bool okx = exchange_end(0, dummy);
          SMP2_TICK(3);
          if (!okx) { aborted = true; __syncthreads(); __syncthreads(); break; }
          bool accept = false; double lnacc = 0; uint32_t rd_mask = 0;
          if (first)
          {
            g.r = prog_theta_decide((uint32_t)g.r, tm, slidem, tslide, 1, q, base);
            const uint32_t accm = wg.dec.accm;
            cnt_prop += (uint32_t)__popc(tm); cnt_acc += (uint32_t)__popc(accm);
            cnt_gprop += (uint32_t)__popc(tm & ~slidem); cnt_gacc += (uint32_t)__popc(accm & tm & ~slidem);
            if (declog && lane < (uint32_t)npop && ((tm >> lane) & 1u))
            {
              const uint32_t k = ndec + (uint32_t)__popc(tm & ((1u << lane) - 1u));
              if (k < 1000u) { double * r = A.declog + 4*k; r[0] = (((slidem >> lane) & 1u) ? 100 : 200) + (int)lane; r[1] = wg.dec.lnacc[lane]; r[2] = -1.0; r[3] = (accm >> lane) & 1u ? 1 : 0; }
            }
            ndec += (uint32_t)__popc(tm);
            SMP2_TICK(4);
          }
          if (!mix)
          {
            g.r = prog_tau_decide((uint32_t)g.r, tm, q, base, lnprior, first);
            accept = wg.dec.acc_step != 0u; rd_mask = wg.dec.rd_mask; lnacc = wg.dec.lnacc_step;
          }
          else
          {
            lnacc = (wg.xtot[0] + wg.xtot[1]*(FX/FXC)) + (double)(nsp - 1)*mix_lnc;
            if (SP.tau_alpha > 0)
            {
              const double troot = wg.tau[npop - 1];
              lnacc += (SP.tau_alpha - 1)*mix_lnc - SP.tau_beta*(troot*mix_c - troot) - (double)(nsp - 2)*mix_lnc;
            }
            lnacc += wg.dec.lnacc_theta; rd_mask = wg.dec.rd_mask;
            accept = g.accept(lnacc);
            if (lane == 0) wg.dec.acc_step = accept ? 1u : 0u;
          }
          if (accept && wg.xcoarse_ && b == 0 && lane == 0) (void)atomicAdd(A.err + 2, 1);      // (a run worth the name never has one)
          ++cnt_prop; cnt_acc += accept ? 1u : 0u;
          if (mix) { ++pj_mix; pj_mix_acc += accept ? 1u : 0u; } else { ++pj_tau; pj_tau_acc += accept ? 1u : 0u; }
          if (declog && lane == 0 && ndec < 1000u) { double * r = A.declog + 4*ndec; r[0] = mix ? 300 : 200 + q; r[1] = lnacc; r[2] = -1.0; r[3] = accept ? 1 : 0; }
          ++ndec;
          // the consequences for the species tree: tau(s), the re-drawn thetas, the sums and the fits the next steps start from
          if (accept)
          {
            if (!mix) { if (lane == 0) wg.tau[q] = tq_new; }
            else if (lane < (uint32_t)npop) wg.tau[lane] *= mix_c;
            if (lane < 16u)
            {
              const Redraw r = wg.rd[lane];
              if ((rd_mask >> lane) & 1u) { wg.tau[MAXPOP + lane] = r.tn; wg.tau[2*MAXPOP + lane] = r.l2t; }
              const bool moved = mix ? lane < (uint32_t)npop && ((tm >> lane) & 1u)
                                     : ((tm >> lane) & 1u) && ((int)lane == q || (int)lane == SP.left[q] || (int)lane == SP.right[q]);
              if (moved) { PopFit & f = wg.pf[lane]; f.T = r.T; f.a = r.a; f.b = r.b; f.c = r.c; }
            }
          }
          wsync();
          if (mix) SMP2_TICK(6); else SMP2_TICK(5);
          SMP2_TICK(7);
          __syncthreads();                                              // B3: the decision is out, the species tree as it now is
          // the next step's proposal, while the loci settle this one (after MIX: the next iteration's first TAU and THETA's
          // choices — unless the launch ends here: the next launch's prologue draws them, the same numbers of the stream)
          if (stepq < npop) make_step(stepq + 1);
          else if (it + 1 < A.niter) { make_step(nsp); theta_choices(); }
          __syncthreads();                                              // B4: the next proposal is out
        }
      }
      // a workgroup that became resident late can pass the launch's last exchange after the others gave up there: nobody
      // stores unless the launch's error word is still clear (the loci waves wait at the same barrier)
      if (!aborted)
      {
        if (lane == 0) wg.late_ = __hip_atomic_load(A.err, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != 0 ? 1u : 0u;      // (the giving-up workgroup's exchange is a release at the coherence point: no plain store lingering in its XCD's L2)
        __syncthreads();                                                // BF
        if (wg.late_) { aborted = true; if (b == 0 && lane == 0) (void)atomicAdd(A.err + 1, (int)A.niter); }      // (workgroup 0 counts the launch's iterations exactly once: here, or where it gave up itself)
      }
      if (b == 0)
      {
        if (lane == 0 && !aborted)
        {
          *A.grng = g.r;
          A.counters[0] += cnt_prop; A.counters[1] += cnt_acc; A.counters[2] += cnt_gprop; A.counters[3] += cnt_gacc;
          // (by move type: the theta window's = the THETA step's proposals that were not Gibbs draws)
          A.pj[4] += pj_tau; A.pj[5] += pj_tau_acc; A.pj[6] += pj_mix; A.pj[7] += pj_mix_acc;
          A.pj[8] += (cnt_prop - pj_tau - pj_mix) - cnt_gprop; A.pj[9] += (cnt_acc - pj_tau_acc - pj_mix_acc) - cnt_gacc;
        }
        if (!aborted && lane < (uint32_t)(3*MAXPOP)) A.taus[lane] = wg.tau[lane];
        if (prof_on) for (int i = 0; i < 24; ++i) A.prof[(i < 16 ? 0 : (int)A.nwg) + i] = (double)wg.prof[i];
        if (prof_on) for (int i = 0; i < 16; ++i) A.prof[(int)A.nwg + 24 + i] = (double)wg.wsweep[i];
      }
      return;
    }
  }

  // ---- load: the loci of this wave
  const uint32_t t0 = gw < A.nwaves ? A.wave_off[gw] : 0u, nt = gw < A.nwaves ? A.wave_off[gw + 1] - t0 : 0u;
  const bool act = slot < nt;
  const uint32_t task = t0 + (act ? slot : 0u);
  GTree<NT> T;
  Stream<BPP> rng{0};
  double lnl_cur = 0, logpr_cur = 0;
  uint32_t np = 0, pb = 0, nprop_done = 0, nacc = 0, w_nupd = 0, w_nbr = 0, a_nupd = 0, a_nbr = 0, a_neval = 0;
  int gl_i = 0;
  for (int k = 0; k < W; ++k) { T.left.w[k] = T.right.w[k] = T.parent.w[k] = T.pop.w[k] = 0xffffffffu; }
  T.cf = T.pf = 0; T.root = 0; T.tips = 2;
  if (act && nt)
  {
    const Loc & L = A.loc[task];
    // (an explicit load and pointer: as `restore ? A.snap[task] : A.trees[task]` bound to a reference the selection
    //  came out as "epoch != 0" alone — trees of an ACCEPTED step were taken from the snapshot)
    const uint32_t flag_now = __hip_atomic_load(A.mix_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool restore_mix = A.epoch != 0u && flag_now == A.epoch;
    const Tree * trp = A.trees + task;
    if (restore_mix) trp = A.snap + task;
    const Tree & tr = *trp;
    const Tree & cur = A.trees[task];               // (streams and counters survive a rejected all-loci step)
    np = L.np;
    double * g_pmat = L.pmat;
    if (li == 0) { S.rate = L.rate; S.rw = L.rw; S.f[0] = L.f0; S.f[1] = L.f1; S.f[2] = L.f2; S.f[3] = L.f3; }
    gl_i = L.gl[li & 15];
    for (int k = 0; k < W; ++k)
    {
      T.left.w[k] = reinterpret_cast<const uint32_t *>(tr.left)[k]; T.right.w[k] = reinterpret_cast<const uint32_t *>(tr.right)[k];
      T.parent.w[k] = reinterpret_cast<const uint32_t *>(tr.parent)[k]; T.pop.w[k] = reinterpret_cast<const uint32_t *>(tr.pop)[k];
    }
    T.root = tr.root; T.tips = tr.tips; rng.r = cur.rng; lnl_cur = tr.lnl; logpr_cur = tr.logpr;
    const int n = 2*T.tips - 1;
    T.cf = gballot<G>(li >= T.tips && li < n && tr.clv[li] != li, gbase);
    T.pf = gballot<G>(li < n && tr.pmat[li] != li, gbase);
    S.time[li] = li < n ? tr.time[li] : 0.0;
    for (uint32_t i = (uint32_t)li; i < (uint32_t)(4*(2*T.tips - 2)); i += G) (&S.ab[0][0])[i] = g_pmat[i];
  }
  // pattern slots of the wave: the loci one after the other
  {
    uint32_t acc = 0;
#pragma unroll
    for (int s2 = 0; s2 < LPW; ++s2)
    {
      const uint32_t n2 = (uint32_t)__builtin_amdgcn_readlane((int)np, s2*G);
      if ((uint32_t)s2 == slot) pb = acc;
      acc += n2;
    }
  }
  if (act)
  {
    const Loc & L = A.loc[task];
    const double * g_clv = L.clv;
    for (uint32_t q = (uint32_t)li; q < np; q += G) wl.pat[pb + q] = A.pat[L.pat_off + q];
    const uint32_t nbuf = 2u*(uint32_t)(T.tips - 1);
    for (uint32_t i = (uint32_t)li; i < nbuf*np; i += G)
    {
      const uint32_t c = i/np, q = i - c*np;
      const double2 * src = reinterpret_cast<const double2 *>(g_clv + ((size_t)c*np + q)*4);
      const double2 u = src[0], w = src[1];
      double * d = wl.clv[c][pb + q];
      d[0] = u.x; d[1] = u.y; d[2] = w.x; d[3] = w.y;
    }
  }
  const int tips = T.tips, n = 2*tips - 1;
  const bool inner_i = li >= tips && li < n;

  // lane-private density state: population li of this locus
  uint32_t mync = 0; double t2h_cur = 0, t2h_new = 0;
  uint32_t mynodes = 0, mync_new = 0; int mynin_new = 0;
  Op opw[NT - 1]; int nops = 0;
  for (int k = 0; k < NT - 1; ++k) opw[k] = 0;

  // inner nodes per population, lineages entering, coalescences — the counting part of gtree_update_logprob_contrib
  auto density_counts = [&]()
  {
    const int pop_i = T.pop[li];
    int below = 0;
#pragma unroll
    for (int q = 0; q < G - 1; ++q)
    {
      const uint32_t m = gballot<G>(inner_i && pop_i == q, gbase);
      if (li == q) mynodes = m;
      below += ((pl.below >> q) & 1u) ? __popc(m) : 0;
    }
    mync_new = (uint32_t)__popc(mynodes);
    mynin_new = gl_i - below;
  };
  // the term of population li (density_term of sampler.hpp); tk = ages of the inner nodes
  auto density_term = [&](const double * tk)
  {
    uint32_t nodes = mynodes;
    const int ncoal = (int)mync_new, nin = mynin_new;
    int steps = ncoal + (pl.ptau >= 0 ? 1 : 0);
    if (nin == steps) --steps;
    double T2h = 0, prev = pl.tau;
    int nn = nin;
#pragma unroll
    for (int k = 0; k < NT; ++k)
      if (k < steps)
      {
        double tkk = pl.ptau;
        if (k < ncoal)
        {
          int best = -1; double tb = 0;
#pragma unroll
          for (int j = 0; j < NT - 1; ++j)
            if (((nodes >> (tips + j)) & 1u) && (best < 0 || tk[j] < tb)) { best = tips + j; tb = tk[j]; }
          tkk = tb; nodes &= ~(1u << best);
        }
        T2h += nn*(nn - 1)*(tkk - prev);
        prev = tkk; --nn;
      }
    double c = 0;
    if (ncoal) c += ncoal*pl.l2t;
    if (T2h) c -= T2h/(pl.theta*1.0);
    S.contrib_new[li] = c; t2h_new = T2h;
  };
  // everything a proposal leaves to do before the decision: density terms, buffer toggles, fresh (a,b), node updates,
  // the ordered sum over the patterns.  Returns the log-likelihood; lp_new = the density.
  auto evaluate = [&](const Prop & pr, bool with_lnl, double & lp_new) -> double
  {
    double tk[NT - 1];
#pragma unroll
    for (int j = 0; j < NT - 1; ++j) tk[j] = S.time[(tips + j) & (NN - 1)];
    const double myage = S.time[li];
    density_counts();
    if ((pr.chain >> li) & 1u) density_term(tk);
    T.pf ^= pr.brm; T.cf ^= pr.ndm;
    if ((pr.brm >> li) & 1u)
    {
      const int par = T.parent[li];
      const double len = (S.time[par & (NN - 1)] - myage)*1.0;                       // rate_mui = 1 (locus.c:2350)
      double a_, b_;
      jc69_ab(len, S.rate, a_, b_);
      const int pi = T.pidx(li);
      S.ab[pi][0] = a_; S.ab[pi][1] = b_;
    }
    // node updates, children first = by age: the rank of node li among the nodes to recompute
    nops = __popc(pr.ndm);
    {
      int rank = 0;
#pragma unroll
      for (int j = 0; j < NT - 1; ++j)
        rank += (((pr.ndm >> (tips + j)) & 1u) && (tk[j] < myage || (tk[j] == myage && tips + j < li))) ? 1 : 0;
      const bool mine = (pr.ndm >> li) & 1u;
#pragma unroll
      for (int k = 0; k < NT - 1; ++k)
        if (k < nops)
        {
          const int x = __ffs(gballot<G>(mine && rank == k, gbase)) - 1;
          const int l = T.left[x], r = T.right[x];
          opw[k] = make_op(T.cidx(x), T.cidx(l), T.pidx(l), T.cidx(r), T.pidx(r));
        }
    }
    wsync();
    double lnl = 0;
    if (with_lnl)
    {
      for (uint32_t base = 0; base < np; base += G)
      {
        const uint32_t q = base + (uint32_t)li; const bool pact = q < np;
        const uint32_t ps = pb + (pact ? q : 0u);
        const uint2 pi = wl.pat[ps];
        double last[4] = {0, 0, 0, 0}; uint32_t last_c = 0xffffffffu;
#pragma unroll
        for (int k = 0; k < NT - 1; ++k)
          if (k < nops)
          {
            const Op o = opw[k];
            const uint32_t opar = (uint32_t)o & 255u, lc = (uint32_t)(o >> 8) & 255u, lp = (uint32_t)(o >> 16) & 255u,
                           rc = (uint32_t)(o >> 24) & 255u, rp = (uint32_t)(o >> 32) & 255u;
            double lv[4], rv[4], x[4], y[4];
            if (lc < (uint32_t)tips) expand_code((pi.y >> (4*lc)) & 15u, lv);
            else if (lc == last_c) { lv[0] = last[0]; lv[1] = last[1]; lv[2] = last[2]; lv[3] = last[3]; }
            else { const double * c = wl.clv[lc - tips][ps]; lv[0] = c[0]; lv[1] = c[1]; lv[2] = c[2]; lv[3] = c[3]; }
            if (rc < (uint32_t)tips) expand_code((pi.y >> (4*rc)) & 15u, rv);
            else if (rc == last_c) { rv[0] = last[0]; rv[1] = last[1]; rv[2] = last[2]; rv[3] = last[3]; }
            else { const double * c = wl.clv[rc - tips][ps]; rv[0] = c[0]; rv[1] = c[1]; rv[2] = c[2]; rv[3] = c[3]; }
            matvec4_ab(S.ab[lp][0], S.ab[lp][1], lv, x);
            matvec4_ab(S.ab[rp][0], S.ab[rp][1], rv, y);
            last[0] = x[0]*y[0]; last[1] = x[1]*y[1]; last[2] = x[2]*y[2]; last[3] = x[3]*y[3]; last_c = opar;
            if (pact) { double * out = wl.clv[opar - tips][ps]; out[0] = last[0]; out[1] = last[1]; out[2] = last[2]; out[3] = last[3]; }
          }
        // the last update is the root's (children first, the root is the oldest node of every update list)
        const double tr_ = dot4_pair(S.f[0], S.f[1], S.f[2], S.f[3], last);
        if (pact) wl.term[ps] = log(0 + tr_*S.rw)*pi.x;
      }
      wsync();
      // the terms in pattern order (core_likelihood.c:206-210): eight loads in flight, then the adds (+ 0.0 past the end)
      for (uint32_t base = 0; base < np; base += 8)
      {
        double v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = base + (uint32_t)j < np ? wl.term[pb + base + (uint32_t)j] : 0.0;
#pragma unroll
        for (int j = 0; j < 8; ++j) lnl += v[j];
      }
      lnl = A.bfbeta == 1.0 ? lnl : A.bfbeta == 0.0 ? 0.0 : A.bfbeta*lnl;
    }
    else wsync();
    double lp = 0;
    {
      // the density terms in population order, loads first (npop < G)
      double v[G];
#pragma unroll
      for (int p = 0; p < G; ++p) v[p] = p < npop ? (((pr.chain >> p) & 1u) ? S.contrib_new[p] : S.contrib[p]) : 0.0;
#pragma unroll
      for (int p = 0; p < G; ++p) lp += v[p];
    }
    lp_new = lp;
    return lnl;
  };
  auto commit_density = [&](uint32_t chain)
  {
    if ((chain >> li) & 1u) { S.contrib[li] = S.contrib_new[li]; t2h_cur = t2h_new; }
    mync = mync_new;
  };

  // ---- the current density terms (all populations): the trees arrive with their sum only
  const uint32_t allpop = (1u << npop) - 1u;
  if (act)
  {
    wsync();
    double tk[NT - 1];
#pragma unroll
    for (int j = 0; j < NT - 1; ++j) tk[j] = S.time[(tips + j) & (NN - 1)];
    density_counts();
    if (li < npop) density_term(tk);
    commit_density(allpop);
    wsync();
    if (A.refresh_logpr)
    {
      double lp = 0;
      for (int p = 0; p < npop; ++p) lp += S.contrib[p];
      logpr_cur = lp;
    }
  }

  const double qnan = __longlong_as_double(0x7ff8000000000000ll);
  uint32_t cnt_prop = 0, cnt_acc = 0;              // all-loci proposals / accepted (the same in every workgroup)
  uint32_t cnt_gprop = 0, cnt_gacc = 0;            // of those: Gibbs draws of a theta
  uint32_t pj_tau = 0, pj_tau_acc = 0, pj_mix = 0, pj_mix_acc = 0, pj_gage = 0, pj_gage_acc = 0;
  const bool declog = (A.dbg & 256u) && b == 0;    // every all-loci decision of this launch: A.declog[4 k] = what, lnacc, u, accepted
  uint32_t ndec = 0;
  const bool wgprof = (A.dbg & 32u) && tid == 0;
  long long wg_sweep = 0;
  bool aborted = false;

  // (Two waves share a SIMD — wave w and w + WAVES/2 — and the pair is bound by instruction issue: a sweep asks for ~0.68 of a
  // SIMD's issue slots, the arbiter serves the older wave first, so the older runs as if alone and the younger finishes a
  // third later: 325 M vs 441 M cycles per 3 000 sweeps.  Priorities (s_setprio) change nothing, and making the older wait
  // for the younger at three points of every proposal only moves both to 441 M: the SUM of their instructions is what the
  // SIMD takes.  10 000 four-taxon loci need five waves of loci per CU, i.e. a pair in every workgroup.)
  if constexpr (PROG) __syncthreads();             // B0: the control wave's first proposal is out
  for (uint32_t it = 0; it < A.niter && !aborted; ++it)
  {
    // ================= GAGE + GSPR of every locus
    const uint32_t nprop = A.nsteps_gage + A.nsteps_gspr;
    const long long wg_t0 = wgprof ? clock64() : 0;
    const bool wvprof = (A.dbg & 16u) && b == 0 && lane == 0;
    const long long wv_t0 = wvprof ? clock64() : 0;
    for (uint32_t step = 0; step < nprop; ++step)
    {
      // roll-back copies: registers, and the age of node li
      const GTree<NT> U = T;
      double tsave = 0, lnl = 0, lp_new = 0;
      Prop pr{0, 0, 0, 0.0};
      bool ok = false;
      if (act)
      {
        tsave = S.time[li];
        ok = step < A.nsteps_gage
          ? propose_gage<NT, BPP>(T, rng, S.time, (int)step, pl, wg.anc, wg.tau, SP.ft_gage, li, gbase, pr)
          : propose_gspr<NT, BPP>(T, rng, S.time, (int)(step - A.nsteps_gage), pl, gl_i, wg.anc, wg.tau, wg.lograt, SP.ft_gspr, li, gbase, pr);
      }
      SMP2_TICK(0);
      if (ok)
      {
        wsync();
        lnl = evaluate(pr, true, lp_new);
        SMP2_TICK(1);
      }
      if (ok)
      {
        w_nupd += (uint32_t)nops; w_nbr += (uint32_t)__popc(pr.brm);
        const double lnacc = (lp_new - logpr_cur) + (lnl - lnl_cur) + pr.hast;
        ++nprop_done;
        const bool gage_step = step < A.nsteps_gage;
        pj_gage += gage_step ? 1u : 0u;
        if (rng.accept(lnacc)) { lnl_cur = lnl; logpr_cur = lp_new; ++nacc; pj_gage_acc += gage_step ? 1u : 0u; commit_density(pr.chain); }
        else { T = U; S.time[li] = tsave; }
        wsync();
        SMP2_TICK(2);
      }
      else if (act) { T = U; S.time[li] = tsave; wsync(); }
    }
    if (wgprof) wg_sweep += clock64() - wg_t0;
    if (wvprof) wg.wsweep[wv] += clock64() - wv_t0;
    if (!A.do_allloci) continue;

    // ================= TAU per species divergence, then MIX: one decision each for all loci.  A step in four parts — the
    // proposal of the species tree (step_begin), every locus's share (step_locus: its terms go to the exchange's slots
    // base ...), the decision (step_decide: whoever decides, from wg.xtot[base ...]) and its consequences (step_apply) —
    // so that the program's first TAU can ride on the THETA step's exchange (below).
    // ---- the program's TAU and MIX re-draw thetas inside the proposal (a00_set_program_moves: opt_rb_theta_update,
    // opt_mix_theta_update): the densities' change over all loci then follows from k_p and the T2h sums, the loci contribute
    // their likelihood change (and, in TAU, the new T2h of the three populations around the divergence)
    constexpr bool program = PROG;
    bool mix = false; int q = -1;
    double wprop = 0, uacc = -1.0;
    double tq_old = 0, tq_lo = 0, tq_hi = 0, minf = 1, maxf = 1, lminf = 0, lmaxf = 0, tq_new = 0, mix_c = 1, mix_lnc = 0;
    double lnacc_theta = 0;
    // what a re-draw leaves on lane p < 16 for population p: theta', log(2/theta'), the new sum T' and the fit to it
    double rd_tn = qnan, rd_l2t = 0, rd_T = 0, rd_a = qnan, rd_b = qnan, rd_c = qnan;
    uint32_t rd_mask = 0;                                   // populations whose theta the step re-draws
    uint32_t cf0 = 0, pf0 = 0;
    double tsave = 0, lnl_new = 0, lp_new = 0, lnprior = 0, dl_tot = 0, lnacc = 0;
    bool evaluated = false, accept = false;
    auto step_begin = [&](int stepq)
    {
      rd_mask = 0; lnprior = 0; accept = false; lnacc_theta = 0;
      if constexpr (PROG)
      {
        // the control wave's proposal of the species tree (wg.stepp, out since the last barrier)
        (void)stepq;
        mix = wg.stepp.mix != 0; q = wg.stepp.q;
        tq_old = wg.stepp.tq_old; tq_lo = wg.stepp.tq_lo; tq_hi = wg.stepp.tq_hi; tq_new = wg.stepp.tq_new;
        minf = wg.stepp.minf; maxf = wg.stepp.maxf; lminf = wg.stepp.lminf; lmaxf = wg.stepp.lmaxf; mix_c = wg.stepp.mix_c; mix_lnc = wg.stepp.mix_lnc;
      }
      else
      {
        mix = stepq == npop;
        q = mix ? -1 : stepq;
        // (log c of the mixing step: finetune x BPP's window variate with its kernel, prop_mixing.c:300; uniform with ours)
        wprop = mix && !BPP ? grng.u() - 0.5 : grng.window(); uacc = BPP ? -1.0 : grng.u();
        tq_old = 0; tq_lo = 0; tq_hi = 0; minf = 1; maxf = 1; lminf = 0; lmaxf = 0; tq_new = 0; mix_c = 1; mix_lnc = 0;
        if (!mix)
        {
          const int pq = SP.parent[q], cl = SP.left[q], cr = SP.right[q];
          tq_old = wg.tau[q]; tq_lo = fmax(wg.tau[cl], wg.tau[cr]); tq_hi = pq >= 0 ? wg.tau[pq] : 999.0;
          tq_new = reflect(tq_old + SP.ft_tau*wprop, tq_lo, tq_hi);
          minf = (tq_new - tq_lo)/(tq_old - tq_lo); maxf = (tq_new - tq_hi)/(tq_old - tq_hi);
          lminf = log(minf); lmaxf = log(maxf);
        }
        else { mix_lnc = SP.ft_mix*wprop; mix_c = exp(mix_lnc); }
      }
      // the proposed species tree: in the lanes' registers only
      if (!mix)
      {
        if (li == q) pl.tau = tq_new;
        if (pl.parent == q) pl.ptau = tq_new;
      }
      else
      {
        pl.tau *= mix_c;
        if (pl.parent >= 0) pl.ptau *= mix_c;
      }
    };
    auto step_locus = [&](int base)
    {
      cf0 = T.cf; pf0 = T.pf;
      tsave = act ? S.time[li] : 0.0;
      lnl_new = lnl_cur; lp_new = logpr_cur;
      evaluated = false;
      if (act)
      {
        Prop pr{allpop, 0, 0, 0.0};
        double hast = 0, hast2 = 0;
        if (!mix)
        {
          // the gene nodes of q and its children between the bounds ride the rubber band (stree.c:4338-4479)
          const int cl = SP.left[q], cr = SP.right[q];
          const int pk = T.pop[li];
          const double tk_ = tsave;
          const bool moved = inner_i && (pk == q || pk == cl || pk == cr) && !(tk_ < tq_lo || tk_ > tq_hi);
          const bool up = moved && tk_ >= tq_old;
          if (moved) S.time[li] = up ? tq_hi + maxf*(tk_ - tq_hi) : tq_lo + minf*(tk_ - tq_lo);
          const uint32_t mm = gballot<G>(moved, gbase);
          const int above = __popc(gballot<G>(up, gbase)), below = __popc(mm) - above;
          const int par = T.parent[li];
          pr.brm = gballot<G>(li < n && par >= 0 && (((mm >> li) & 1u) || ((mm >> (par & 31)) & 1u)), gbase);
          uint32_t m = mm;
          const int l = T.left[li], r = T.right[li];
#pragma unroll
          for (int d = 0; d < NT - 2; ++d) m |= gballot<G>(inner_i && (((m >> (l & 31)) & 1u) || ((m >> (r & 31)) & 1u)), gbase);
          pr.ndm = m;
          hast = below*lminf; hast2 = above*lmaxf;
        }
        else
        {
          if (inner_i) S.time[li] = tsave*mix_c;
          pr.ndm = gballot<G>(inner_i, gbase);
          pr.brm = gballot<G>(li < n && (int)T.parent[li] >= 0, gbase);
          hast = (double)(tips - 1)*mix_lnc;
        }
        wsync();
        evaluated = pr.ndm != 0;
        const double lnl = evaluate(pr, evaluated, lp_new);
        if (evaluated) { lnl_new = lnl; a_nupd += (uint32_t)nops; a_nbr += (uint32_t)__popc(pr.brm); ++a_neval; }
        const double dpr = program ? 0.0 : lp_new - logpr_cur;
        const double h = mix ? dpr + hast : (dpr + hast) + hast2;
        const double dl = evaluated ? (lnl_new - lnl_cur) + h : h;
        if (li == 0) fx_add(base, dl, true);
        if (program && !mix && li < npop && ((A.theta_mask >> li) & 1u))
        {
          const int slot = li == q ? 2 : li == SP.left[q] ? 3 : li == SP.right[q] ? 4 : -1;
          if (slot >= 0) fx_add(base + slot, t2h_new, false);
        }
      }
    };
    // the decision (decide of sampler.hpp; stree.c:6280, prop_mixing.c:203-205)
    // the decision where every wave takes it for itself (decide of sampler.hpp; stree.c:6280, prop_mixing.c:203-205)
    auto step_decide = [&](int base)
    {
      dl_tot = wg.xtot[base] + wg.xtot[base + 1]*(FX/FXC);      // (+ the coarse sum: terms of 256 and more — none in any run worth the name)
      lnacc = dl_tot;
      if (!mix) lnacc += lnprior;
      else
      {
        lnacc += (double)(nsp - 1)*mix_lnc;
        if (SP.tau_alpha > 0)
        {
          const double troot = wg.tau[npop - 1];
          lnacc += (SP.tau_alpha - 1)*mix_lnc - SP.tau_beta*(troot*mix_c - troot) - (double)(nsp - 2)*mix_lnc;
        }
      }
      accept = BPP ? grng.accept(lnacc) : (lnacc >= 0 || uacc < exp(lnacc));
    };
    // the consequences: every wave's copy of the species tree (non-PROG: workgroup's LDS copy by its first lanes; PROG: the
    // control wave has done that), the loci's trees — commit or roll back —, the densities where thetas moved
    auto step_apply = [&](bool refresh)
    {
      if constexpr (!PROG)
      {
        if (accept && wg.xcoarse_ && b == 0 && tid == 0) (void)atomicAdd(A.err + 2, 1);      // (a run worth the name never has one)
        ++cnt_prop; cnt_acc += accept ? 1u : 0u;
        if (mix) { ++pj_mix; pj_mix_acc += accept ? 1u : 0u; } else { ++pj_tau; pj_tau_acc += accept ? 1u : 0u; }
        if (declog && tid == 0 && ndec < 1000u) { double * r = A.declog + 4*ndec; r[0] = mix ? 300 : 200 + q; r[1] = lnacc; r[2] = uacc; r[3] = accept ? 1 : 0; }
        ++ndec;
        __syncthreads();                                      // everyone has read the old taus
        if (accept)
        {
          if (!mix) { if (tid == 0) wg.tau[q] = tq_new; }
          else if (tid < (uint32_t)npop) wg.tau[tid] *= mix_c;
        }
      }
      if (accept) { if (act) { lnl_cur = lnl_new; logpr_cur = lp_new; commit_density(allpop); } }
      else if (act) { T.cf = cf0; T.pf = pf0; S.time[li] = tsave; }
      if constexpr (!PROG) __syncthreads();
      load_pop();
      wsync();
      if (PROG && (refresh || accept) && act)
      {
        // the densities with the thetas as they are now (THETA's decisions, a TAU's or MIX's re-draws), from the statistics of the trees as settled
        if (li < npop) S.contrib[li] = msc_term((int)mync, t2h_cur, pl.theta, pl.l2t);
        wsync();
        double lp = 0;
        for (int p = 0; p < npop; ++p) lp += S.contrib[p];
        logpr_cur = lp;
        wsync();
      }
    };
    if constexpr (PROG)
    {
      // ================= the program's moves, a loci wave's part: its loci's terms of every all-loci step; the control wave
      // (above) does the rest between the step's two barriers.  THETA's sums ride with the first TAU's.
      const bool on = li < npop && ((A.theta_mask >> li) & 1u);
      const int kidx = __popc(A.theta_mask & ((1u << li) - 1u)), nth = 2*__popc(A.theta_mask);
      for (int stepq = nsp; stepq <= npop && !aborted; ++stepq)
      {
        const bool first = stepq == nsp;
        step_begin(stepq);
        if (first && act && on) { fx_add(2*kidx, (double)mync, false); fx_add(2*kidx + 1, t2h_cur, false); }
        step_locus(first ? nth : 0);
        __syncthreads();                                                // B1: the terms are in
        __syncthreads();                                                // B3: the decision is out (and the species tree as it now is, and the next proposal)
        const bool ab = wg.abort_ != 0u;
        accept = wg.dec.acc_step != 0u;
        if (!ab) step_apply(first);
        __syncthreads();                                                // B4: the next proposal is out
        if (ab) { aborted = true; break; }
      }
    }
    else
    {
    // ================= THETA: every population that can hold a coalescence, decided independently (theta_step_all)
    if (SP.theta_alpha > 0 && A.theta_mask)
    {
      const bool on = li < npop && ((A.theta_mask >> li) & 1u);
      const double told = pl.theta, l2t_old = pl.l2t;
      double tnew = told, uacc = -1.0, my_lnacc = 0;
      bool accept = false, gibbs_me = false;
      {
        // the windows of all populations first (theta_step_all of a00_driver.c: the uniform kernel draws the acceptance
        // number right behind each window, BPP's kernel only when a decision needs it)
        double win = 0;
        for (int p = 0; p < npop; ++p)
          if ((A.theta_mask >> p) & 1u)
          {
            const double w_ = grng.window(), a_ = BPP ? -1.0 : grng.u();
            if (p == li) { win = w_; uacc = a_; }
          }
        tnew = reflect(told + SP.ft_theta*win, 0.0, 999.0);
        const double l2t_new = log(2.0/(1.0*tnew));
        if (act && on) fx_add(li, msc_term((int)mync, t2h_cur, tnew, l2t_new) - msc_term((int)mync, t2h_cur, told, l2t_old), false);
        SMP2_TICK(3);
        double th_tot = 0;
        if (!exchange(npop, li, th_tot)) { aborted = true; break; }
        SMP2_TICK(6);
        // every wave takes the (same) decisions for itself: lane li decides population li; BPP's kernel draws its
        // acceptance numbers now, in population order, so every lane walks through all of them
        if (BPP)
        {
          if (on) { wl.term[li] = tnew; wl.term[16 + li] = told; }
          wsync();
          for (int p = 0; p < npop; ++p)
            if ((A.theta_mask >> p) & 1u)
            {
              const double tn = wl.term[p], to = wl.term[16 + p];
              const double lnacc = wg.xtot[p] + ((SP.theta_alpha - 1)*log(tn/to) - SP.theta_beta*(tn - to));
              const bool acc = tn > 0 && grng.accept(lnacc);
              if (p == li) { accept = acc; my_lnacc = lnacc; }
            }
          wsync();
        }
        else if (on)
        {
          my_lnacc = th_tot + ((SP.theta_alpha - 1)*log(tnew/told) - SP.theta_beta*(tnew - told));
          accept = tnew > 0 && (my_lnacc >= 0 || uacc < exp(my_lnacc));
        }
      }
      const double l2t_new = log(2.0/(1.0*tnew));
      if (accept) { pl.theta = tnew; pl.l2t = l2t_new; }
      if (declog && tid < (uint32_t)G && on)
      {
        const uint32_t k = ndec + (uint32_t)__popc(A.theta_mask & ((1u << li) - 1u));
        if (k < 2048u) { double * r = A.declog + 4*k; r[0] = (gibbs_me ? 200 : 100) + li; r[1] = my_lnacc; r[2] = uacc; r[3] = accept ? 1 : 0; }
      }
      ndec += (uint32_t)__popc(A.theta_mask);
      {
        const uint32_t onm = gballot<G>(on, gbase), accm = gballot<G>(accept, gbase), gm = gballot<G>(on && gibbs_me, gbase);
        cnt_prop += (uint32_t)__popc(onm); cnt_acc += (uint32_t)__popc(accm);
        cnt_gprop += (uint32_t)__popc(gm); cnt_gacc += (uint32_t)__popc(gm & accm);
      }
      __syncthreads();                                        // everyone has read the totals and the old thetas
      if (tid < (uint32_t)G && accept) { wg.tau[MAXPOP + li] = tnew; wg.tau[2*MAXPOP + li] = l2t_new; }
      __syncthreads();
      // every tree's density with the new thetas, from its statistics, in population order
      if (act)
      {
        if (li < npop) S.contrib[li] = msc_term((int)mync, t2h_cur, pl.theta, pl.l2t);
        wsync();
        double lp = 0;
        for (int p = 0; p < npop; ++p) lp += S.contrib[p];
        logpr_cur = lp;
        wsync();
      }
    }
    SMP2_TICK(3);

    // ================= TAU per species divergence, then MIX
    for (int stepq = nsp; stepq <= npop && !aborted; ++stepq)
    {
      step_begin(stepq);
      step_locus(0);
      if (mix) SMP2_TICK(5); else SMP2_TICK(4);
      if (!mix && SP.parent[q] < 0 && SP.tau_alpha > 0)
        lnprior = (SP.tau_alpha - 1 - (nsp - 1) + 1)*log(tq_new/tq_old) - SP.tau_beta*(tq_new - tq_old);
      if (!exchange(2, 0, dl_tot)) { aborted = true; break; }
      if (A.dbg & 64u) { SMP2_TICK(7); double dummy; if (!exchange(2, 0, dummy)) { aborted = true; break; } SMP2_TICK(6); }     // (the protocol alone: nobody is late)
      SMP2_TICK(7);
      SMP2_SUB0();
      step_decide(0);
      SMP2_SUB(14);
      step_apply(false);
      if (mix) SMP2_TICK(5); else SMP2_TICK(4);
    }
    }
  }
#undef SMP2_TICK
#undef SMP2_SUB0
#undef SMP2_SUB
  if (aborted || wg.abort_) return;                 // (HBM still holds the state the launch started from)
  if constexpr (!PROG) { if (tid == 0) wg.late_ = __hip_atomic_load(A.err, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != 0 ? 1u : 0u; }
  __syncthreads();                                  // BF: the error word as it stands after the last exchange (PROG: read by the control wave)
  if (wg.late_)
  {
    // (workgroup 0 counts the launch's iterations exactly once: where it gave up itself, or here)
    if constexpr (!PROG) { if (b == 0 && tid == 0) (void)atomicAdd(A.err + 1, (int)A.niter); }
    return;
  }

  // ---- store
  if (act)
  {
    Tree & tr = A.trees[task];
    double * g_clv = A.loc[task].clv, * g_pmat = A.loc[task].pmat;
    {
      // (word li of each byte array: selected, not indexed — a run-time index would put the whole tree into scratch memory)
      uint32_t wl_ = 0, wr_ = 0, wp_ = 0, wq_ = 0;
#pragma unroll
      for (int k = 0; k < W; ++k) if (li == k) { wl_ = T.left.w[k]; wr_ = T.right.w[k]; wp_ = T.parent.w[k]; wq_ = T.pop.w[k]; }
      if (li < W)
      {
        reinterpret_cast<uint32_t *>(tr.left)[li] = wl_; reinterpret_cast<uint32_t *>(tr.right)[li] = wr_;
        reinterpret_cast<uint32_t *>(tr.parent)[li] = wp_; reinterpret_cast<uint32_t *>(tr.pop)[li] = wq_;
      }
    }
    if (li < n) { tr.time[li] = S.time[li]; tr.clv[li] = (int8_t)T.cidx(li); tr.pmat[li] = (int8_t)T.pidx(li); }
    if (li == 0)
    {
      tr.lnl = lnl_cur; tr.logpr = logpr_cur; tr.rng = rng.r; tr.root = T.root;
      tr.proposals += nprop_done; tr.accepted += nacc; tr.sw_nupd += w_nupd; tr.sw_nbr += w_nbr;
      tr.al_nupd += a_nupd; tr.al_nbr += a_nbr; tr.al_neval += a_neval;
      // by move type, over all loci (what the burn-in's step-length rule reads: bpa_sampler_adapt_finetune)
      (void)atomicAdd(A.pj + 0, (unsigned long long)pj_gage); (void)atomicAdd(A.pj + 1, (unsigned long long)pj_gage_acc);
      (void)atomicAdd(A.pj + 2, (unsigned long long)(nprop_done - pj_gage)); (void)atomicAdd(A.pj + 3, (unsigned long long)(nacc - pj_gage_acc));
    }
    if (li < npop) { A.pop_nc[(size_t)li*A.ntasks + task] = (int8_t)mync; A.pop_t2h[(size_t)li*A.ntasks + task] = t2h_cur; }
    for (uint32_t i = (uint32_t)li; i < (uint32_t)(4*(2*tips - 2)); i += G) g_pmat[i] = (&S.ab[0][0])[i];
    const uint32_t nbuf = 2u*(uint32_t)(tips - 1);
    for (uint32_t i = (uint32_t)li; i < nbuf*np; i += G)
    {
      const uint32_t c = i/np, q = i - c*np;
      const double * d = wl.clv[c][pb + q];
      double2 u, w; u.x = d[0]; u.y = d[1]; w.x = d[2]; w.y = d[3];
      double2 * dst = reinterpret_cast<double2 *>(g_clv + ((size_t)c*np + q)*4);
      dst[0] = u; dst[1] = w;
    }
  }
  if (b == 0 && tid == 0)
  {
    *A.grng = grng.r;
    A.counters[0] += cnt_prop; A.counters[1] += cnt_acc; A.counters[2] += cnt_gprop; A.counters[3] += cnt_gacc;
    if constexpr (!PROG)
    {
      A.pj[4] += pj_tau; A.pj[5] += pj_tau_acc; A.pj[6] += pj_mix; A.pj[7] += pj_mix_acc;
      A.pj[8] += (cnt_prop - pj_tau - pj_mix) - cnt_gprop; A.pj[9] += (cnt_acc - pj_tau_acc - pj_mix_acc) - cnt_gacc;
    }
  }
  if (b == 0 && tid < (uint32_t)(3*MAXPOP)) A.taus[tid] = wg.tau[tid];
  if (prof_on) for (int i = 0; i < 24; ++i) A.prof[(i < 16 ? 0 : (int)A.nwg) + i] = (double)wg.prof[i];
  if (prof_on) for (int i = 0; i < 16; ++i) A.prof[(int)A.nwg + 24 + i] = (double)wg.wsweep[i];
  if (wgprof) A.prof[16 + b] = (double)wg_sweep;
}

} // namespace smp2
