/*
 * bpp_amd_host.h — host-side MCMC control in C over the likelihood boundary.
 *
 * BASELINE.json north_star: "host MCMC control flow stays in C and calls HIP through a thin
 * C-ABI shim".  This is that host side, reduced to what drives the likelihood path: a
 * lock-step sampler of the gene trees of all loci under the multispecies coalescent on a fixed
 * species tree, with the moves of BPP's A00 iteration (method.c:5490-5602) that call the
 * likelihood:
 *
 *   GAGE  gene-node age, sliding window reflected into (max(children, tau of the children's
 *         common population), parent)                              propose_ages, gtree.c:4585
 *   GSPR  subtree prune and regraft: new age of the father, target drawn among the branches
 *         crossing it in the same population, Hastings ratio targets/sources
 *                                                                   propose_spr, gtree.c:6531
 *   TAU   species divergence time with the rubber band on the gene nodes of the three
 *         populations around it, ONE decision for all loci
 *                                   propose_tau / propose_tau_update_gtrees, stree.c:5512/4338
 *   MIX   all ages and taus times c, ONE decision              proposal_mixing, prop_mixing.c:52
 *
 * It keeps the gnode_t fields the reference keeps (left/right/parent/time/pop/clv_index/
 * scaler_index/pmatrix_index), toggles the double buffers before every evaluation exactly as the
 * reference does (SWAP_CLV_INDEX / SWAP_PMAT_INDEX / SWAP_SCALER_INDEX, locus.c:24-26) and toggles
 * back on rejection.  "Step j of every locus" is handed to a likelihood back-end as one batch.
 * Back-ends: libbpp_amd.so (the product, a00_backend_hip); for tests the real reference's locus
 * API (oracle/ref_shim.c: ref_backend_eval) — same driver, same seeds, same trajectory on both is
 * the drop-in check — and a00_backend_prior (lnL = 0, BPP's usedata = 0).
 *
 * Acceptance: Metropolis-Hastings on  MSC density x likelihood.  The MSC density of a gene tree is
 * gtree_logprob (gtree.c:3957) = the sum over populations of gtree_update_logprob_contrib
 * (gtree.c:3859), a00_msc_logpr below, bit-equal to the reference's (tests/test_msc_density.py).
 * The taus carry BPP's gamma/Dirichlet prior (a00_set_tau_prior) and the thetas BPP's gamma prior with a
 * THETA step per population (a00_set_theta_prior): with both on, the sampler targets the A00 posterior
 * of BPP (species tree fixed, JC69 or whatever the back-end's loci use).
 */
#ifndef BPP_AMD_HOST_H
#define BPP_AMD_HOST_H

#include <math.h>
#include "bpp_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Random numbers: every locus owns a stream (so that loci can be advanced independently, on the
   host or on the device, with identical results) and the all-loci mixing step a global one.
   64-bit LCG (Knuth MMIX constants); a00_rndu gives a uniform in (0,1).  A GAGE proposal always
   consumes 2 numbers of its locus's stream (proposal, acceptance), a GSPR proposal 3.           */
typedef unsigned long long a00_rng_t;
static inline a00_rng_t a00_rng_seed(unsigned long seed, unsigned stream)
{
  a00_rng_t z = 0x9E3779B97F4A7C15ULL*(a00_rng_t)(stream + 1) ^ (a00_rng_t)seed*0xD1B54A32D192ED03ULL;
  z ^= z >> 31; z *= 0xBF58476D1CE4E5B9ULL; z ^= z >> 29;
  return z | 1ULL;
}
static inline double a00_rndu(a00_rng_t * r)
{
  *r = *r*6364136223846793005ULL + 1442695040888963407ULL;
  return (double)((*r >> 11) + 0.5)*(1.0/9007199254740992.0);
}
#define A00_GLOBAL_STREAM 0xFFFFFFFFu

/* BPP's OWN generator and window kernel, restated (a00_set_proposal_kernel(d, A00_KERNEL_BPP) makes the driver use
   them): legacy_rndu (random.c:104-122: z = z*69069 + 1 on 32 bits, 0 replaced by 12345671, value z*2^-32) and
   legacy_rnd_symmetrical (random.c:230-238) = the Bactrian-Laplace variate rndBactrianLaplace (random.c:201-213;
   rndLaplace :192-199; mBactrian = 0.90, random.c:24-25): mean 0, variance 1, two draws of the generator.  Bit-equal
   to the reference's functions on the same state (tests/test_bpp_kernel.py).                                      */
static inline double a00_bpp_rndu(unsigned int * z)
{
  *z = *z*69069u + 1u;
  if (*z == 0) *z = 12345671u;
  return ldexp((double)(*z), -32);
}
static inline double a00_bpp_rnd_laplace(unsigned int * z)
{
  const double u = a00_bpp_rndu(z) - 0.5;
  const double r = log(1 - 2*fabs(u))*0.70710678118654752440;
  return u >= 0 ? -r : r;
}
static inline double a00_bpp_rnd_symmetrical(unsigned int * z)
{
  double v = 0.90 + a00_bpp_rnd_laplace(z)*sqrt(1 - 0.90*0.90);
  if (a00_bpp_rndu(z) < 0.5) v = -v;
  return v;
}

/* BPP's THETA move is, nine times out of ten, not the sliding window but a METROPOLIZED GIBBS draw
   (stree_propose_theta, stree.c:3957-4060: sliding window with probability opt_theta_slide_prob = 0.1, bpp.c:650;
   propose_theta_gibbs, stree.c:3645-3826).  Given the gene trees, theta_p's conditional depends on two sums over the
   loci only: k = the coalescences in p and T = sum of C2ji/heredity (the T2h of a00_msc_term).  Under the gamma(a, b)
   prior the program fits an INVERSE-GAMMA(a1, b1) to that conditional (get_gamma_conditional_approx, stree.c:3384-3459,
   the branch of opt_theta_prop = MG_INVG, which bpp.c:975 makes the default for a gamma prior): its mode m and
   curvature give mmv = m^2/v, a1 is the root of x^3 - (4 + mmv) x^2 + (5 - 2 mmv) x - (2 + mmv) between (mmv + 2)/2
   and 2 (mmv + 2) found by bisection to 1e-6 (cubic_root, stree.c:3360), b1 = m (a1 + 1).  The proposal
   theta' = b1 / gamma(a1, 1) (legacy_rndgamma, random.c:240-275: Marsaglia-Tsang with the Box-Muller-Marsaglia normal
   of rndNormal, random.c:215-232) is accepted on
       k log(theta/theta') - T (1/theta' - 1/theta)  +  (a - 1) log(theta'/theta) - b (theta' - theta)
                                                      +  (-a1 - 1) log(theta/theta') - b1 (1/theta - 1/theta').
   The functions below restate the three pieces; they are bit-equal to the reference's on the same state and the same
   arguments (tests/test_bpp_kernel.py) and compile for the host driver and the device kernel alike (A00_HD).  The two
   rejection loops are bounded (64 rounds each: a round fails with probability < 0.22 resp. < 0.05); a draw that runs
   out returns NaN and the step is rejected.                                                                         */
#ifdef __HIPCC__
#define A00_HD __host__ __device__
#else
#define A00_HD
#endif
static inline A00_HD double a00_bpp_rndu_hd(unsigned int * z)
{
  *z = *z*69069u + 1u;
  if (*z == 0) *z = 12345671u;
  return (double)(*z)*(1.0/4294967296.0);                       /* = ldexp(z, -32), exact */
}
static inline A00_HD double a00_bpp_rndnormal(unsigned int * z)
{
  int round_;
  for (round_ = 0; round_ < 64; ++round_)
  {
    const double u = 2*a00_bpp_rndu_hd(z) - 1, v = 2*a00_bpp_rndu_hd(z) - 1;
    const double s = u*u + v*v;
    if (s > 0 && s < 1) return u*sqrt(-2*log(s)/s);             /* (the second variate of the pair is not used) */
  }
  return NAN;
}
static inline A00_HD double a00_bpp_rndgamma(unsigned int * z, double shape)
{
  const double a = shape < 1 ? shape + 1 : shape;
  const double d = a - 1.0/3.0, c = (1.0/3.0)/sqrt(d);
  double x, v = NAN, u;
  int round_, inner;
  for (round_ = 0; round_ < 64; ++round_)
  {
    for (inner = 0; inner < 64; ++inner)
    {
      x = a00_bpp_rndnormal(z);
      v = 1.0 + c*x;
      if (!(v <= 0)) break;
    }
    if (!(v > 0)) return NAN;
    v *= v*v;
    u = a00_bpp_rndu_hd(z);
    if (u < 1 - 0.0331*x*x*x*x) break;
    if (log(u) < 0.5*x*x + d*(1 - v + log(v))) break;
  }
  if (round_ == 64) return NAN;
  v *= d;
  if (shape < 1) v *= pow(a00_bpp_rndu_hd(z), 1/shape);
  if (v == 0) v = 1E-300;
  return v;
}
static inline A00_HD double a00_cubic_value(const double * c, double x) { return c[0]*x*x*x + c[1]*x*x + c[2]*x + c[3]; }
static inline A00_HD void a00_theta_conditional_invgamma(double a, double b, long k, double T, double * a1, double * b1)
{
  double c[4], lo, hi, flo, x = 0, f;
  int i;
  if (T == 0) { *a1 = a + 2; *b1 = a*(a + 1)/b; return; }
  {
    const double a1k = a - 1 - k;
    const double m = (a1k + sqrt(a1k*a1k + 4*b*T))/(2*b);       /* the conditional's mode */
    const double ddl = -(a1k + 2*T/m)/(m*m);
    const double v = -1/ddl, mmv = m*m/v;
    c[0] = 1; c[1] = -(4 + mmv); c[2] = 5 - 2*mmv; c[3] = -(2 + mmv);
    lo = (mmv + 2)/2; hi = (mmv + 2)*2;
    flo = a00_cubic_value(c, lo);
    if (!(flo*a00_cubic_value(c, hi) <= 0)) { *a1 = NAN; *b1 = NAN; return; }        /* (the program stops here: 'bounds error') */
    for (i = 0; i < 100; ++i)
    {
      x = (lo + hi)/2;
      if (fabs(lo - hi) < 1e-6) break;
      f = a00_cubic_value(c, x);
      if (flo*f > 0) { lo = x; flo = f; } else hi = x;
    }
    *a1 = x; *b1 = m*(x + 1);
  }
}
/* The same fit — the same midpoints, the same decisions, the same bits — with the bisection's 35 cubic evaluations replaced by
   comparisons: between the bounds the cubic falls to one minimum and rises through its only root r there (for m^2/v > 2.5:
   its local maximum then lies left of 0), so "the cubic at x has the sign it has at lo" is "x < r".  r comes from a few
   Newton steps started right of it (r = M + 6 - 16/M + ...; convex there: monotone convergence); a midpoint closer to r than
   2^-44 r — a hundred times the rounding error of either r or the cubic — sends the whole search back to the plain loop.  Anything outside that picture (small m^2/v, other signs at the bounds, no convergence) takes the plain
   loop.  The device samplers' form of a00_theta_conditional_invgamma (tests/test_bpp_kernel.py: bit-equal on 10^5 cases). */
static inline A00_HD void a00_theta_conditional_invgamma_fast(double a, double b, long k, double T, double * a1, double * b1)
{
  double c[4], lo, hi, flo, fhi, x = 0, f, r, s = 0, guard, m, mmv;
  int i, fast;
  if (T == 0) { *a1 = a + 2; *b1 = a*(a + 1)/b; return; }
  {
    const double a1k = a - 1 - k;
    double ddl, v;
    m = (a1k + sqrt(a1k*a1k + 4*b*T))/(2*b);
    ddl = -(a1k + 2*T/m)/(m*m);
    v = -1/ddl; mmv = m*m/v;
  }
  c[0] = 1; c[1] = -(4 + mmv); c[2] = 5 - 2*mmv; c[3] = -(2 + mmv);
  lo = (mmv + 2)/2; hi = (mmv + 2)*2;
  flo = a00_cubic_value(c, lo); fhi = a00_cubic_value(c, hi);
  if (!(flo*fhi <= 0)) { *a1 = NAN; *b1 = NAN; return; }
  fast = mmv > 2.5 && flo < 0 && fhi > 0;
  r = mmv + 6;
  if (fast)
  {
    for (i = 0; i < 40; ++i)
    {
      const double fr = ((r + c[1])*r + c[2])*r + c[3], dr = (3*r + 2*c[1])*r + c[2];
      s = fr/dr;
      r -= s;
      if (!(fabs(s) > 1e-13*r)) break;
    }
    fast = fabs(s) <= 1e-13*r && r > lo && r < hi;
  }
  if (fast)
  {
    /* the number of halvings until the bounds are closer than 1e-6 follows from their distance (a width differs from
       w0 / 2^k by a few roundings of the midpoints: ~1e-11): n = the smallest k with w0 / 2^k < 1e-6.  Then n steps of
       add, halve, compare, select — nothing else in the loop.  Checked afterwards: the widths fall on either side of 1e-6
       with room to spare (1e-9), and no midpoint came closer to r than 2^-44 r (a hundred times either rounding error:
       such a midpoint is one of the final bounds) — else (one case in a few hundred) the plain loop below. */
    const double lo0 = lo, hi0 = hi, w0 = hi - lo;
    int n = 0;
    if (w0 >= 1e-6)
    {
      n = ilogb(w0) + 20;                                       /* (ilogb(1e-6) = -20) */
      if (n < 1) n = 1;
      if (ldexp(w0, -(n - 1)) < 1e-6) --n; else if (!(ldexp(w0, -n) < 1e-6)) ++n;
    }
    for (i = 0; i < n; ++i)
    {
      x = (lo + hi)/2;
      if (x < r) lo = x; else hi = x;
    }
    guard = r*5.684341886080802e-14;                            /* 2^-44 r */
    {
      const double w = hi - lo;
      if (n <= 90 && w < 1e-6 - 1e-9 && (n == 0 || 2*w > 1e-6 + 1e-9) && r - lo > guard && hi - r > guard)
      { x = (lo + hi)/2; *a1 = x; *b1 = m*(x + 1); return; }
    }
    lo = lo0; hi = hi0;
  }
  for (i = 0; i < 100; ++i)
  {
    x = (lo + hi)/2;
    if (fabs(lo - hi) < 1e-6) break;
    f = a00_cubic_value(c, x);
    if (flo*f > 0) { lo = x; flo = f; } else hi = x;
  }
  *a1 = x; *b1 = m*(x + 1);
}
/* ln of the acceptance ratio of a theta proposal from the two sums (both moves: the density's change is the first line) */
static inline A00_HD double a00_theta_lnacc(long k, double T, double told, double tnew, double a, double b)
{
  return (k*(log(2.0/tnew) - log(2.0/told)) - (T/tnew - T/told)) + ((a - 1)*log(tnew/told) - b*(tnew - told));
}
static inline A00_HD double a00_theta_gibbs_hastings(double a1, double b1, double told, double tnew)
{
  return (-a1 - 1)*log(told/tnew) - b1*(1/told - 1/tnew);
}
/* The program's MIXING step moves the thetas too (proposal_mixing, prop_mixing.c:272-425; opt_mix_theta_update = 1,
   bpp.c:581): with all ages and taus times c, every theta is drawn from the inverse-gamma fitted to its conditional given
   the SCALED trees (k, c T); the proposal ratio is  invgamma(theta | fit(k, T)) / invgamma(theta' | fit(k, c T))
   (logPDFInvG, prop_mixing.c:254-263), the prior ratio the gamma prior's; log c is finetune x the Bactrian-Laplace variate. */
static inline A00_HD double a00_invgamma_logpdf(double x, double a, double b)
{
  return a*log(b) - lgamma(a) + (-a - 1)*log(x) - b/x;
}

/* gene tree of one locus: tips 0..tips-1, inner nodes after; the root node object stays
   the root (gtree.c:6129-6175), so pmatrix indices never collide */
typedef struct a00_tree
{
  int      tips, n, root;
  int *    left, * right, * parent;      /* [n], -1 = none          */
  double * time;                         /* [n] node ages           */
  int *    clv, * pmat, * scaler;        /* [n] current buffer indices (gnode_t fields) */
  double   rate_mui;                     /* gtree_t.rate_mui        */
  double   lnl;                          /* current log-likelihood  */
  int *    pop;                          /* [n] gnode_t.pop: species of a tip, population an inner node coalesces in */
  double   logpr;                        /* current MSC density, gtree_t.logpr */
} a00_tree_t;

/* one proposal step for a set of loci, in node terms */
typedef struct a00_step
{
  unsigned             nloci;
  const unsigned *     locus;            /* [nloci] index of the locus            */
  a00_tree_t * const * tree;             /* [nloci] its tree, proposal installed  */
  const unsigned *     br_off;           /* [nloci+1] */
  const int *          branches;         /* child node ids whose P-matrix changes */
  const unsigned *     nd_off;           /* [nloci+1] */
  const int *          nodes;            /* inner node ids to recompute, children first */
} a00_step_t;

typedef int (*a00_eval_fn)(void * ctx, const a00_step_t * step, double * lnl /* [nloci] */);

typedef struct a00_driver a00_driver_t;

a00_driver_t * a00_create(unsigned nloci, a00_eval_fn eval, void * ctx, unsigned long seed);
void           a00_destroy(a00_driver_t *);
/* install the start tree of locus i (arrays are copied); scaling != 0 gives inner nodes scalers */
int            a00_set_tree(a00_driver_t *, unsigned i, int tips, const int * left, const int * right,
                            const double * times, int root, int scaling);
const a00_tree_t * a00_tree(const a00_driver_t *, unsigned i);
/* The species tree: `species` tips (populations 0..species-1) and species-1 inner populations
   after them, children before parents (the root last); parent[] (-1 for the root), tau[] (0 for
   tips) and theta[] have 2*species-1 entries, stree->nodes order (tips first).  species <= 8.
   Must be set before a00_initialize.                                                             */
#define A00_MAXPOP 15
int            a00_set_species_tree(a00_driver_t *, int species, const int * parent, const double * tau,
                                    const double * theta);
/* species of the tips of locus i (default: tip k belongs to species k) */
int            a00_set_tip_species(a00_driver_t *, unsigned i, const int * species);
/* Which generator and window kernel the moves draw from.
   A00_KERNEL_UNIFORM (default): a00_rndu streams, a step = finetune x (u - 1/2), acceptance "lnacc >= 0 or u < exp(lnacc)"
     with u always drawn — the kernel the device-resident sampler (bpa_sampler_t) also runs: same streams, same trajectory.
   A00_KERNEL_BPP: the reference's — legacy_rndu streams (one 32-bit state per locus + one global), a step =
     finetune x legacy_rnd_symmetrical() for the ages, the taus and the thetas (gtree.c:4722, 6666; stree.c:5632, 3848),
     finetune x (rndu - 1/2) for log c of the mixing step (prop_mixing.c), acceptance "lnacc >= -1e-10 or rndu < exp(lnacc)"
     with the uniform drawn only when needed (gtree.c:5476, stree.c:6286).  The finetunes then mean what they mean in a
     BPP control file.  Call before a00_initialize.                                                                   */
#define A00_KERNEL_UNIFORM 0
#define A00_KERNEL_BPP     1
void           a00_set_proposal_kernel(a00_driver_t *, int kind);
/* THETA, TAU and MIX as the program runs them (needs A00_KERNEL_BPP and a theta prior; default off):
     THETA  each theta gets the sliding window with probability slide_prob (BPP: 0.1) and the metropolized Gibbs draw
            above otherwise; both are decided from the two sums over loci (k exactly, T as a sum of 2^-40 fixed-point
            terms, so that the device kernel's order-free sum is the same number);
     TAU    the rubber band also re-draws the thetas of the population and its two children (opt_rb_theta_update = 1,
            bpp.c:618; propose_tau, stree.c:5840-5990), each from the inverse-gamma fitted to (k, sum of the T2h after
            the move), proposal ratio by a00_invgamma_logpdf;
     MIX    likewise every theta, from (k, c T) (opt_mix_theta_update = 1, bpp.c:581; prop_mixing.c:272-425).
   In TAU and MIX the change of the gene-tree densities over all loci is taken from the sums:
   k (log 2/theta' - log 2/theta) - (T'/theta' - T/theta) per re-drawn theta; the loci contribute their likelihood change
   (and, in TAU, their three new T2h).  k and T are carried from the THETA step's sums through the iteration.          */
void           a00_set_program_moves(a00_driver_t *, int on, double slide_prob);
void           a00_gibbs_counters(const a00_driver_t *, unsigned long * proposals, unsigned long * accepted);
/* worker threads of the per-locus loops (proposal, MSC density, bookkeeping, roll-back; OpenMP).  Every draw of a per-locus
   proposal comes from that locus's own stream and sums over loci are taken in locus order afterwards, so the trajectory
   does not depend on the count.  Default 1, or the environment's A00_THREADS; threads.c:87-200 is the reference's form */
void           a00_set_threads(a00_driver_t *, int threads);
/* the first n values of the restated generator (what = 0) / window variate (1) / gamma(shape, 1) variate (2) from
   state `seed` (tests) */
void           a00_bpp_kernel_sequence(unsigned int seed, int what, int n, double * out);
void           a00_bpp_gamma_sequence(unsigned int seed, double shape, int n, double * out);
void           a00_theta_conditional(double a, double b, long k, double T, double * a1b1);
void           a00_theta_conditional_fast(double a, double b, long k, double T, double * a1b1);
/* window widths of the four moves (defaults 0.004, 0.004, 0.001, 0.3) */
void           a00_set_finetune(a00_driver_t *, double gage, double gspr, double tau, double mix);
/* prior on the divergence times as BPP's 'tauprior = gamma a b': gamma(alpha, beta) on the root tau, the
   others uniform below it (stree.c:5655-5657); alpha = 0 (default): flat */
void           a00_set_tau_prior(a00_driver_t *, double alpha, double beta);
/* thetas: BPP's 'thetaprior = gamma a b' with a sliding-window THETA step per population that can hold a
   coalescence, after the gene-tree moves of every iteration; alpha = 0 (default): thetas stay fixed */
void           a00_set_theta_prior(a00_driver_t *, double alpha, double beta, double finetune);
unsigned       a00_get_thetas(const a00_driver_t *, double * theta);
/* current tau[] (2*species-1 entries); returns the number of populations */
unsigned       a00_get_taus(const a00_driver_t *, double * tau);
/* MSC density of the current gene tree of locus i, recomputed from scratch */
double         a00_locus_logpr(const a00_driver_t *, unsigned i);

/* reflection of x into (a,b) (reflect, gtree.c:3983) */
static inline double a00_reflect(double x, double a, double b)
{
  const double w = b - a; double e; long n;
  if (!(w > 0)) return a;
  if (x >= a && x <= b) return x;
  e = x < a ? a - x : x - b;
  n = (long)(e/w);
  e -= (double)n*w;
  /* an even number of whole widths keeps the side the walk left from */
  if ((x < a) == ((n & 1) == 0)) return a + e;
  return b - e;
}
/* gtree_update_logprob_contrib (gtree.c:3859-3955) for one population: tau = its start, ptau = its
   parent's tau (< 0 for the root), nin = lineages entering, times[ncoal] the coalescent times in it
   SORTED ascending; heredity multiplies theta as in the reference.  Same operations, same order. */
static inline double a00_msc_t2h(double tau, double ptau, int nin, const double * times, int ncoal)
{
  double T2h = 0, prev = tau; int k, n = nin;
  /* intervals end at every coalescence and at the parent's tau; the one after the last possible
     coalescence (n == 1) is not visited */
  int steps = ncoal + (ptau >= 0 ? 1 : 0);
  if (nin == steps) --steps;
  for (k = 0; k < steps; ++k, --n)
  {
    const double t = k < ncoal ? times[k] : ptau;
    T2h += n*(n - 1)*(t - prev);
    prev = t;
  }
  return T2h;
}
/* the population's term from its sufficient statistics (coalescences, total 2h coalescent waiting time) */
static inline double a00_msc_term(int ncoal, double T2h, double theta, double heredity)
{
  double logpr = 0;
  if (ncoal) logpr += ncoal*log(2.0/(heredity*theta));
  if (T2h) logpr -= T2h/(theta*heredity);
  return logpr;
}
static inline double a00_msc_contrib(double tau, double ptau, double theta, double heredity, int nin,
                                     const double * times, int ncoal)
{
  return a00_msc_term(ncoal, a00_msc_t2h(tau, ptau, nin, times, ncoal), theta, heredity);
}
/* ---- the per-locus substitution-parameter moves of a GTR(+Gamma) analysis (locus.c:2782-3419, prop_gamma.c:52-224;
   cmd_run runs them after the mixing step, method.c:5699-5735).  Every move is a full recomputation of the locus
   (every P-matrix, every partial) with the proposed value and a per-locus decision:
     FREQS   each base frequency j != T in turn: sliding window on log pi_j, reflected into (log 1e-5, log(pi_j + pi_T)),
             pi_T takes up the difference; lnacc = delta log pi_j + delta lnL                       (propose_freqs, locus.c:2782)
     QRATES  each exchangeability j != reference (AG for GTR) in turn, the same way               (propose_qrates, locus.c:3168)
     ALPHA   the shape of the discrete-gamma site rates: sliding window on log alpha, category rates by
             pll_compute_gamma_cats, gamma(a, b) prior: lnacc = delta log alpha + prior ratio + delta lnL   (prop_gamma.c:52)
   The parameters live in the driver (a00_set_subst_model); the likelihood back-end learns new values through the
   a00_param_fn it registers (which = 1 frequencies [4], 2 exchangeabilities [6], 4 category rates [ncat]).          */
typedef int (*a00_param_fn)(void * ctx, unsigned locus, int which, const double * values, unsigned n);
void           a00_set_param_backend(a00_driver_t *, a00_param_fn);
int            a00_set_subst_model(a00_driver_t *, unsigned i, const double * freqs, const double * qrates, double alpha, int ncat);
int            a00_get_subst_model(const a00_driver_t *, unsigned i, double * freqs, double * qrates, double * alpha);
/* window widths (0 = that move is off; all off by default) and the gamma(a, b) prior on alpha ('alphaprior = a b ncat') */
void           a00_set_subst_moves(a00_driver_t *, double ft_freqs, double ft_qrates, double ft_alpha, double alpha_a, double alpha_b);
int            a00_backend_hip_params(void * ctx /* a00_hip_ctx_t* */, unsigned locus, int which, const double * values, unsigned n);

/* start-up evaluation: all matrices, all partials, lnL (method.c:4285-4297) */
int            a00_initialize(a00_driver_t *);
/* one iteration: GAGE over inner nodes, GSPR over non-root nodes, THETA per population (if a prior is set),
   one TAU step per inner population, one MIX step */
int            a00_iterate(a00_driver_t *);
double         a00_total_lnl(const a00_driver_t *);
void           a00_counters(const a00_driver_t *, unsigned long * proposals, unsigned long * accepted,
                            unsigned long * steps);

/* likelihood back-end on libbpp_amd.so: loci[i] must have been created with the buffer
   counts of method.c:4110-4146 (2*inner CLVs, 2*edges P-matrices, 2*inner scalers)       */
typedef struct a00_hip_ctx { bpa_engine_t * engine; bpa_locus_t ** loci; } a00_hip_ctx_t;
int a00_backend_hip(void * ctx /* a00_hip_ctx_t* */, const a00_step_t * step, double * lnl);

/* Two cohorts of loci on two engines (one GPU, two streams): loci [0, split) are evaluated through ctx0, the others
   through ctx1 — the loci's handles in a context's array sit at their DRIVER index, whichever engine made them.  A
   per-locus step (GAGE k, GSPR k) of one cohort is proposed, marshalled and sent off while the other cohort's launch
   runs; an all-loci step sends each cohort's share to its engine and waits for both.  Every draw of a per-locus step
   comes from that locus's own stream and the all-loci sums run in locus order, so the trajectory is the one of the
   plain driver (tests/test_gpu_host_driver.py).
     submit   1: in flight (wait will deliver the per-locus lnL), 2: evaluated already, lnl filled, 0: error
     wait     the lnL of the context's batch in flight                                                                  */
typedef int (*a00_submit_fn)(void * ctx, const a00_step_t * step, double * lnl /* [nloci] */);
typedef int (*a00_wait_fn)(void * ctx, double * lnl /* [n] */, unsigned n);
int a00_set_cohorts(a00_driver_t *, unsigned split, a00_submit_fn, a00_wait_fn, void * ctx0, void * ctx1);
int a00_backend_hip_submit(void * ctx /* a00_hip_ctx_t* */, const a00_step_t * step, double * lnl);
int a00_backend_hip_wait(void * ctx /* a00_hip_ctx_t* */, double * lnl, unsigned n);
/* for a synchronous backend used as its own submit (an a00_eval_fn has submit's signature): nothing to wait for */
int a00_backend_wait_none(void * ctx, double * lnl, unsigned n);
/* lnL = 0 for every locus: the sampler then draws gene trees from the MSC prior (BPP's usedata = 0) */
int a00_backend_prior(void * ctx, const a00_step_t * step, double * lnl);

#ifdef __cplusplus
}
#endif
#endif
