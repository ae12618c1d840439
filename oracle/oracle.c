/*
 * oracle/oracle.c — TEST INFRASTRUCTURE ONLY (checker, never product).
 *
 * CPU restatement, in plain C99, of the reference's per-locus likelihood path
 * (bpp v4.8.7): Felsenstein pruning, root log-likelihood, P-matrix update,
 * eigendecomposition, discrete-Gamma rates, site-pattern compression.  Written
 * from the arithmetic in SURVEY.md App. A — not from the reference's text —
 * and pinned against the real reference by tests/test_oracle_pin.py
 * (oracle/_ref/libbppref.so, built in place by oracle/Makefile) and against the
 * golden vectors under tests/golden/.
 *
 * Build: gcc -O2 -std=c99 -ffp-contract=off -mfma (the only FMAs are the
 * explicit fma() calls that mirror the reference's AVX2+FMA back-end).
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

/* ------------------------------------------------------------------ maps ---
 * State codes: nucleotides are 4-bit sets A=1,C=2,G=4,T/U=8 with the IUPAC
 * ambiguity unions, gap/N/?/X/O = 15 (maps.c:26-44); amino acids are 20-bit
 * one-hot in the order ARNDCQEGHILKMFPSTWYV, B=N|D, Z=Q|E, X/-/?/'*' = all
 * (maps.c:126-144).  The tables are filled once by a constructor.           */
unsigned int orc_map_nt_tab[256];
unsigned int orc_map_aa_tab[256];
const unsigned int * orc_get_map_nt(void) { return orc_map_nt_tab; }
const unsigned int * orc_get_map_aa(void) { return orc_map_aa_tab; }

static void put(unsigned int * tab, char c, unsigned int code)
{
  tab[(unsigned char)c] = code;
  if (c >= 'A' && c <= 'Z') tab[(unsigned char)(c - 'A' + 'a')] = code;
}

__attribute__((constructor)) static void fill_maps(void)
{
  const char * nt = "ACGT";
  const char * aa = "ARNDCQEGHILKMFPSTWYV";
  unsigned int i;
  for (i = 0; i < 4; ++i) put(orc_map_nt_tab, nt[i], 1u << i);
  put(orc_map_nt_tab, 'U', 8);
  put(orc_map_nt_tab, 'R', 1|4);  put(orc_map_nt_tab, 'Y', 2|8);
  put(orc_map_nt_tab, 'S', 2|4);  put(orc_map_nt_tab, 'W', 1|8);
  put(orc_map_nt_tab, 'K', 4|8);  put(orc_map_nt_tab, 'M', 1|2);
  put(orc_map_nt_tab, 'B', 2|4|8); put(orc_map_nt_tab, 'D', 1|4|8);
  put(orc_map_nt_tab, 'H', 1|2|8); put(orc_map_nt_tab, 'V', 1|2|4);
  put(orc_map_nt_tab, 'N', 15); put(orc_map_nt_tab, 'X', 15); put(orc_map_nt_tab, 'O', 15);
  put(orc_map_nt_tab, '-', 15); put(orc_map_nt_tab, '?', 15);

  for (i = 0; i < 20; ++i) put(orc_map_aa_tab, aa[i], 1u << i);
  put(orc_map_aa_tab, 'B', (1u << 2) | (1u << 3));
  put(orc_map_aa_tab, 'Z', (1u << 5) | (1u << 6));
  put(orc_map_aa_tab, 'X', 0xFFFFF); put(orc_map_aa_tab, '*', 0xFFFFF);
  put(orc_map_aa_tab, '-', 0xFFFFF); put(orc_map_aa_tab, '?', 0xFFFFF);
}

/* -------------------------------------------------------------- tip CLVs ---
 * locus.c:525-559 (set_tipclv): entry [n][k][s] = bit s of map[seq[n]] as
 * 0.0/1.0, replicated over the rate categories.                             */
void orc_set_tipclv(unsigned states, unsigned sites, unsigned rate_cats,
                    const unsigned * map, const char * seq, double * clv)
{
  unsigned n, k, s;
  for (n = 0; n < sites; ++n)
  {
    unsigned code = map[(unsigned char)seq[n]];
    for (k = 0; k < rate_cats; ++k)
      for (s = 0; s < states; ++s)
        clv[((size_t)n*rate_cats + k)*states + s] = (double)((code >> s) & 1u);
  }
}

/* ------------------------------------------------------------ dot orders ---
 * One row-times-vector dot product in each of the reference's three
 * roundings.                                    */
static double dot_seq(const double * a, const double * b, unsigned n)
{
  double s = 0; unsigned j;
  for (j = 0; j < n; ++j) s += a[j]*b[j];
  return s;
}

static double dot_pair4(const double * a, const double * b)
{
  double p0 = a[0]*b[0], p1 = a[1]*b[1], p2 = a[2]*b[2], p3 = a[3]*b[3];
  return (p0 + p1) + (p2 + p3);
}

/* four lane accumulators, lane l takes columns l, l+4, l+8, ... with fused
   multiply-add; then (acc0+acc1)+(acc2+acc3).  n must be a multiple of 4.   */
static double dot_fma4(const double * a, const double * b, unsigned n)
{
  double acc[4] = {0,0,0,0}; unsigned j, l;
  for (j = 0; j < n; j += 4)
    for (l = 0; l < 4; ++l)
      acc[l] = fma(a[j+l], b[j+l], acc[l]);
  return (acc[0] + acc[1]) + (acc[2] + acc[3]);
}

static double dot_order(const double * a, const double * b, unsigned n, int order)
{
  if (order == ORC_ORDER_PAIR && n == 4) return dot_pair4(a, b);
  if (order == ORC_ORDER_FMA4 && (n & 3) == 0) return dot_fma4(a, b, n);
  return dot_seq(a, b, n);
}

/* --------------------------------------------------------------- K1 --------
 * pll_core_update_partial_ii (core_partials.c:585-756; 4-state AVX order
 * core_partials_avx.c:414-530; generic AVX2 order core_partials_avx2.c:626-797).
 * parent[n][k][i] = (P_left[k][i][.] . left[n][k][.]) * (P_right[k][i][.] . right[n][k][.]).
 * Per-pattern scaling (only with a parent scaler): scaler = left + right; if
 * every entry of the pattern is < 2^-256 (strict) multiply all by 2^256, ++scaler. */
int orc_update_partial_ii(unsigned states, unsigned sites, unsigned rate_cats,
                          double * parent_clv, unsigned * parent_scaler,
                          const double * left_clv, const double * right_clv,
                          const double * left_matrix, const double * right_matrix,
                          const unsigned * left_scaler, const unsigned * right_scaler,
                          int order)
{
  unsigned n, k, i;
  size_t span = (size_t)rate_cats*states;

  for (n = 0; n < sites; ++n)
  {
    double * out = parent_clv + n*span;
    int all_small = 1;
    for (k = 0; k < rate_cats; ++k)
    {
      const double * lv = left_clv  + n*span + (size_t)k*states;
      const double * rv = right_clv + n*span + (size_t)k*states;
      const double * lm = left_matrix  + (size_t)k*states*states;
      const double * rm = right_matrix + (size_t)k*states*states;
      for (i = 0; i < states; ++i)
      {
        double x = dot_order(lm + (size_t)i*states, lv, states, order);
        double y = dot_order(rm + (size_t)i*states, rv, states, order);
        double v = x*y;
        out[(size_t)k*states + i] = v;
        all_small &= (v < ORC_SCALE_THRESHOLD);
      }
    }
    if (parent_scaler)
    {
      unsigned s = (left_scaler ? left_scaler[n] : 0) + (right_scaler ? right_scaler[n] : 0);
      if (all_small)
      {
        for (i = 0; i < span; ++i) out[i] *= ORC_SCALE_FACTOR;
        s += 1;
      }
      parent_scaler[n] = s;
    }
  }
  return 1;
}

/* --------------------------------------------------------------- K2 / K3 ---
 * Per-pattern site likelihood  sum_k rw[k] * (pi . clv[n][k][.]) :
 * core_likelihood.c:179-195 (scalar), core_likelihood_avx.c:117-139 (4-state:
 * products then (x0+x1)+(x2+x3)), core_likelihood_avx2.c:45-72 (generic: FMA
 * lanes then (a0+a1)+(a2+a3); that file is built with -mfma so gcc also fuses
 * `term += term_r*rw` — reproduced here for ORC_ORDER_FMA4).                 */
static double site_lh(unsigned states, unsigned rate_cats, const double * clv_n,
                      const double * freqs, const double * rw, int order)
{
  double term = 0; unsigned k;
  for (k = 0; k < rate_cats; ++k)
  {
    double tr = dot_order(freqs, clv_n + (size_t)k*states, states, order);
    if (order == ORC_ORDER_FMA4) term = fma(tr, rw[k], term);
    else                         term += tr*rw[k];
  }
  return term;
}

double orc_root_loglikelihood(unsigned states, unsigned sites, unsigned rate_cats,
                              const double * clv, const unsigned * scaler,
                              const double * freqs, const double * rate_weights,
                              const unsigned * pattern_weights, double * persite_lnl,
                              int order)
{
  double logl = 0; unsigned n;
  for (n = 0; n < sites; ++n)
  {
    double t = log(site_lh(states, rate_cats, clv + (size_t)n*rate_cats*states,
                           freqs, rate_weights, order));
    if (scaler && scaler[n])
    {
      /* core_likelihood_avx2.c is compiled with -mfma: the scaler correction fuses too */
      if (order == ORC_ORDER_FMA4) t = fma((double)scaler[n], log(ORC_SCALE_THRESHOLD), t);
      else                         t += scaler[n]*log(ORC_SCALE_THRESHOLD);
    }
    t *= pattern_weights[n];
    if (persite_lnl) persite_lnl[n] = t;
    logl += t;
  }
  return logl;
}

/* pll_core_root_likelihood_vector (core_likelihood.c:214-408): the site
   likelihood itself, no log, scalers ignored (core_likelihood_avx.c:277-290) */
void orc_root_likelihood_vector(unsigned states, unsigned sites, unsigned rate_cats,
                                const double * clv, const double * freqs,
                                const double * rate_weights, double * persite_lh,
                                int order)
{
  unsigned n;
  for (n = 0; n < sites; ++n)
    persite_lh[n] = site_lh(states, rate_cats, clv + (size_t)n*rate_cats*states,
                            freqs, rate_weights, order);
}

/* diploid branch of locus_root_loglikelihood (locus.c:2586-2615): mean over
   the phase resolutions of each unphased pattern, log, times weight          */
double orc_diploid_loglikelihood(const double * persite_lh, int unphased_length,
                                 const unsigned long * resolution_count,
                                 const unsigned long * mapping,
                                 const unsigned * unphased_weights)
{
  double logl = 0; int u; unsigned long r, k = 0;
  for (u = 0; u < unphased_length; ++u)
  {
    double m = 0;
    for (r = 0; r < resolution_count[u]; ++r) m += persite_lh[mapping[k++]];
    m /= resolution_count[u];
    logl += log(m)*unphased_weights[u];
  }
  return logl;
}

/* --------------------------------------------------------------- K4 --------
 * locus_update_matrices_jc69 (locus.c:2342-2414): bl = t*rate; identity when
 * bl < 1e-100, else diagonal a=(1+3exp(-4bl/3))/4, off-diagonal b=(1-a)/3.   */
void orc_pmatrix_jc69(unsigned rate_cats, const double * rates, double t, double * pmat)
{
  unsigned k, i, j;
  for (k = 0; k < rate_cats; ++k)
  {
    double bl = t*rates[k], a = 1, b = 0;
    if (!(bl < 1e-100))
    {
      a = (1 + 3*exp(-4*bl/3))/4;
      b = (1 - a)/3;
    }
    for (i = 0; i < 4; ++i)
      for (j = 0; j < 4; ++j)
        pmat[k*16 + i*4 + j] = (i == j) ? a : b;
  }
}

/* --------------------------------------------------------------- K7 --------
 * Closed-form 4x4 P(t) of the other nucleotide models, state order A,C,G,T:
 * K80 (locus.c:2240-2323), F81 (2172-2238), HKY / F84 / TN93 (2060-2170), T92 (1981-2058).
 * Written with expm1 as the reference does (P = I + ...), entry by entry in the same
 * evaluation order so the values are bit-identical.  model: 1 K80, 2 F81, 3 HKY, 4 T92,
 * 5 TN93, 6 F84 (bpp.h:216-221); q = locus->subst_params, f = locus->frequencies.       */
void orc_pmatrix_dna(unsigned model, const double * f, const double * q, unsigned rate_cats,
                     const double * rates, double t, double * pmat)
{
  unsigned k, i, j;
  for (k = 0; k < rate_cats; ++k)
  {
    double * p = pmat + k*16;
    const double bl = t*rates[k];
    if (model == 1)                                   /* K80: kappa = q[1]/q[0] */
    {
      const double kappa = q[1]/q[0];
      const double e1 = expm1(-4*bl/(kappa + 2));
      if (fabs(kappa - 1) < 1e-20)
        for (i = 0; i < 4; ++i) for (j = 0; j < 4; ++j) p[4*i + j] = (i == j) ? 1. + 3/4.*e1 : -e1/4;
      else
      {
        const double e2 = expm1(-2*bl*(kappa + 1)/(kappa + 2));
        for (i = 0; i < 4; ++i)
          for (j = 0; j < 4; ++j)
            p[4*i + j] = (i == j) ? 1 + (e1 + 2*e2)/4            /* same base          */
                       : ((i ^ j) == 2) ? (e1 - 2*e2)/4           /* transition A<->G, C<->T */
                       : -e1/4;                                   /* transversion       */
      }
    }
    else if (model == 2)                              /* F81 */
    {
      double beta = 1;
      for (j = 0; j < 4; ++j) beta -= f[j]*f[j];
      beta = 1./beta;
      {
        const double e = exp(-beta*bl), em1 = expm1(-beta*bl);
        for (i = 0; i < 4; ++i) for (j = 0; j < 4; ++j) p[4*i + j] = (i == j) ? e - f[j]*em1 : -f[j]*em1;
      }
    }
    else if (model == 4)                              /* T92: GC content + kappa */
    {
      const double GC = f[3] + f[2];
      const double e1 = expm1(-bl);
      const double e2 = expm1(-(q[0]/q[1] + 1)*bl/2);
      const double a = -(1 - GC)/2*e1, g = -GC/2*e1;
      p[0]  = a;                       p[1]  = GC/2*e1 - GC*e2;          p[2]  = g;                     p[3]  = 1 + 0.5*(1 - GC)*e1 + GC*e2;
      p[4]  = a;                       p[5]  = 1 + GC/2*e1 + (1 - GC)*e2; p[6]  = g;                     p[7]  = (1 - GC)/2*e1 - (1 - GC)*e2;
      p[8]  = 1 + 0.5*(1 - GC)*e1 + GC*e2; p[9]  = g;                     p[10] = GC/2*e1 - GC*e2;       p[11] = a;
      p[12] = (1 - GC)/2*e1 - (1 - GC)*e2; p[13] = g;                     p[14] = 1 + GC/2*e1 + (1 - GC)*e2; p[15] = a;
    }
    else                                              /* HKY (3), TN93 (5), F84 (6) */
    {
      const double A = f[0], C = f[1], G = f[2], T = f[3], Y = T + C, R = A + G;
      double bt, a1t, a2t, e1, e2, e3;
      if (model == 3)
      {
        const double kappa = q[1]/q[0];
        const double mr = 1/(2*T*C*kappa + 2*A*G*kappa + 2*Y*R);
        bt = bl*mr; a1t = a2t = kappa*bt;
      }
      else if (model == 6)
      {
        const double kappa = q[0]/q[1];
        const double mr = 1/(2*T*C*kappa + 2*A*G*kappa + 2*Y*R);
        bt = bl*mr; a1t = (1 + kappa/Y)*bt; a2t = (1 + kappa/R)*bt;
      }
      else
      {
        const double mr = 1/(2*T*C*q[0] + 2*A*G*q[1] + 2*Y*R);
        bt = bl*mr; a1t = (q[0]/q[2])*bt; a2t = (q[1]/q[2])*bt;
      }
      e1 = expm1(-bt); e2 = expm1(-(R*a2t + Y*bt)); e3 = expm1(-(Y*a1t + R*bt));
      p[0]  = 1 + Y*A/R*e1 + G/R*e2;  p[1]  = -C*e1;             p[2]  = Y*G/R*e1 - G/R*e2;     p[3]  = -T*e1;
      p[4]  = -A*e1;                  p[5]  = 1 + (R*C*e1 + T*e3)/Y; p[6]  = -G*e1;               p[7]  = (R*e1 - e3)*T/Y;
      p[8]  = Y*A/R*e1 - A/R*e2;      p[9]  = -C*e1;             p[10] = 1 + Y*G/R*e1 + A/R*e2; p[11] = -T*e1;
      p[12] = -A*e1;                  p[13] = (R*e1 - e3)*C/Y;   p[14] = -G*e1;                 p[15] = 1 + (R*T*e1 + C*e3)/Y;
    }
  }
}

/* --------------------------------------------------------------- K6 --------
 * pll_update_eigen (core_pmatrix.c:186-297).  Symmetrised rate matrix
 * A_ij = r_ij sqrt(pi_i pi_j), A_ii = -sum_j r_ij pi_j, scaled to mean rate 1;
 * Householder tridiagonalisation then implicit-shift QL (the classical
 * EISPACK tred2/tql2 pair in the column-oriented form the reference uses,
 * core_pmatrix.c:28-182), same operation order so the result is bit-identical. */
static void householder_tridiag(double ** a, unsigned n, double * d, double * e)
{
  unsigned i, j, k;
  for (i = n - 1; i >= 1; --i)
  {
    unsigned l = i;              /* number of leading elements in column i */
    double h = 0, scale = 0;
    if (l > 1)
    {
      for (k = 0; k < l; ++k) scale += fabs(a[k][i]);
      if (scale == 0.0)
        e[i] = a[l-1][i];
      else
      {
        double f, g, hh;
        for (k = 0; k < l; ++k) { a[k][i] /= scale; h += a[k][i]*a[k][i]; }
        f = a[l-1][i];
        g = (f > 0) ? -sqrt(h) : sqrt(h);
        e[i] = scale*g;
        h -= f*g;
        a[l-1][i] = f - g;
        f = 0.0;
        for (j = 0; j < l; ++j)
        {
          a[i][j] = a[j][i]/h;
          g = 0.0;
          for (k = 0; k <= j; ++k)    g += a[k][j]*a[k][i];
          for (k = j + 1; k < l; ++k) g += a[j][k]*a[k][i];
          e[j] = g/h;
          f += e[j]*a[j][i];
        }
        hh = f/(h + h);
        for (j = 0; j < l; ++j)
        {
          f = a[j][i];
          g = e[j] - hh*f;
          e[j] = g;
          for (k = 0; k <= j; ++k) a[k][j] -= (f*e[k] + g*a[k][i]);
        }
      }
    }
    else
      e[i] = a[l-1][i];
    d[i] = h;
  }
  d[0] = 0.0; e[0] = 0.0;

  for (i = 0; i < n; ++i)
  {
    unsigned l = i;
    if (d[i] != 0.0)
      for (j = 0; j < l; ++j)
      {
        double g = 0.0;
        for (k = 0; k < l; ++k) g += a[k][i]*a[j][k];
        for (k = 0; k < l; ++k) a[j][k] -= g*a[i][k];
      }
    d[i] = a[i][i];
    a[i][i] = 1.0;
    for (j = 0; j < l; ++j) a[i][j] = a[j][i] = 0.0;
  }
}

static void ql_implicit(double * d, double * e, unsigned n, double ** z)
{
  unsigned l, m, i, k;
  for (i = 1; i < n; ++i) e[i-1] = e[i];
  e[n-1] = 0.0;

  for (l = 0; l < n; ++l)
  {
    for (;;)
    {
      double g, r, s, c, p, f, b;
      for (m = l; m + 1 < n; ++m)
      {
        double dd = fabs(d[m]) + fabs(d[m+1]);
        if (fabs(e[m]) + dd == dd) break;
      }
      if (m == l) break;

      g = (d[l+1] - d[l])/(2.0*e[l]);
      r = sqrt(g*g + 1.0);
      g = d[m] - d[l] + e[l]/(g + ((g < 0) ? -fabs(r) : fabs(r)));
      s = c = 1.0;
      p = 0.0;
      for (i = m; i-- > l; )           /* i = m-1 down to l */
      {
        f = s*e[i];
        b = c*e[i];
        if (fabs(f) >= fabs(g))
        {
          c = g/f;
          r = sqrt(c*c + 1.0);
          e[i+1] = f*r;
          c *= (s = 1.0/r);
        }
        else
        {
          s = f/g;
          r = sqrt(s*s + 1.0);
          e[i+1] = g*r;
          s *= (c = 1.0/r);
        }
        g = d[i+1] - p;
        r = (d[i] - g)*s + 2.0*c*b;
        p = s*r;
        d[i+1] = g + p;
        g = c*r - b;
        for (k = 0; k < n; ++k)
        {
          f = z[i+1][k];
          z[i+1][k] = s*z[i][k] + c*f;
          z[i][k]   = c*z[i][k] - s*f;
        }
      }
      d[l] = d[l] - p;
      e[l] = g;
      e[m] = 0.0;
    }
  }
}

void orc_update_eigen(unsigned states, const double * freqs, const double * subst_params,
                      double * eigenvecs, double * inv_eigenvecs, double * eigenvals)
{
  unsigned n = states, i, j, k, np = n*(n-1)/2;
  double * r = (double *)malloc(np*sizeof(double));
  double ** a = (double **)malloc(n*sizeof(double *));
  double * d = (double *)malloc(n*sizeof(double));
  double * e = (double *)malloc(n*sizeof(double));
  double mean = 0;

  /* exchangeabilities relative to the last one (core_pmatrix.c:198-202) */
  memcpy(r, subst_params, np*sizeof(double));
  if (r[np-1] > 0.0)
    for (i = 0; i < np; ++i) r[i] /= r[np-1];

  for (i = 0; i < n; ++i) a[i] = (double *)calloc(n, sizeof(double));
  for (k = 0, i = 0; i < n; ++i)
    for (j = i + 1; j < n; ++j)
    {
      double x = r[k++];
      a[i][j] = a[j][i] = x*sqrt(freqs[i]*freqs[j]);
      a[i][i] -= x*freqs[j];
      a[j][j] -= x*freqs[i];
    }
  for (i = 0; i < n; ++i) mean += freqs[i]*(-a[i][i]);
  for (i = 0; i < n; ++i)
    for (j = 0; j < n; ++j) a[i][j] /= mean;

  householder_tridiag(a, n, d, e);
  ql_implicit(d, e, n, a);

  /* rows of `a` are eigenvectors u_m; eigenvecs[m][k] = u_m[k] sqrt(pi_k),
     inv_eigenvecs[j][m] = u_m[j]/sqrt(pi_j)  (core_pmatrix.c:262-290)        */
  for (i = 0; i < n; ++i)
  {
    eigenvals[i] = d[i];
    for (j = 0; j < n; ++j)
    {
      inv_eigenvecs[j*n + i] = a[i][j]/sqrt(freqs[j]);
      eigenvecs[i*n + j]     = a[i][j]*sqrt(freqs[j]);
    }
  }
  for (i = 0; i < n; ++i) free(a[i]);
  free(a); free(d); free(e); free(r);
}

/* --------------------------------------------------------------- K5 --------
 * bpp_core_update_pmatrix (core_pmatrix.c:728-772, inference form:
 * expm1(lambda*(t*rate)), identity when t*rate < 1e-100) and
 * pll_core_update_pmatrix (core_pmatrix.c:814-862, library/simulator form:
 * expm1(lambda*rate*t), identity when t == 0).  P = I + (V^-1 diag(e)) V,
 * accumulated from the identity entry with m ascending.                      */
void orc_pmatrix_eigen(unsigned states, unsigned rate_cats, const double * rates, double t,
                       const double * eigenvals, const double * eigenvecs,
                       const double * inv_eigenvecs, double * pmat, int library_form)
{
  unsigned n = states, k, j, c, m;
  double * ex = (double *)malloc(n*sizeof(double));
  double * tmp = (double *)malloc((size_t)n*n*sizeof(double));
  for (k = 0; k < rate_cats; ++k)
  {
    double * p = pmat + (size_t)k*n*n;
    double bl = t*rates[k];
    int ident = library_form ? (t == 0.0) : (bl < 1e-100);
    if (ident)
    {
      for (j = 0; j < n; ++j)
        for (c = 0; c < n; ++c) p[j*n + c] = (j == c) ? 1.0 : 0.0;
      continue;
    }
    for (j = 0; j < n; ++j)
      ex[j] = library_form ? expm1(eigenvals[j]*rates[k]*t) : expm1(eigenvals[j]*bl);
    for (j = 0; j < n; ++j)
      for (c = 0; c < n; ++c) tmp[j*n + c] = inv_eigenvecs[j*n + c]*ex[c];
    for (j = 0; j < n; ++j)
      for (c = 0; c < n; ++c)
      {
        double acc = (j == c) ? 1.0 : 0.0;
        for (m = 0; m < n; ++m) acc += tmp[j*n + m]*eigenvecs[m*n + c];
        p[j*n + c] = acc;
      }
  }
  free(ex); free(tmp);
}

/* --------------------------------------------------------------- K8 --------
 * pll_compute_gamma_cats, mean-rate mode (gamma.c:221-284), built from the
 * published routines it uses: log-gamma by Pike & Hill (1966, Alg. 291),
 * incomplete gamma ratio by Bhattacharjee (1970, AS 32), normal quantile by
 * Odeh & Evans (1974, AS 70), chi-square quantile by Best & Roberts (1975,
 * AS 91).  Constants are those algorithms' published constants.              */
static double ln_gamma_ph(double alpha)                       /* gamma.c:97-132 */
{
  double x = alpha, f = 0.0, z;
  if (x < 7.0)
  {
    f = 1.0;
    z = alpha - 1.0;              /* (alpha-1)+1 need not round back to alpha */
    while ((z = z + 1.0) < 7.0) f *= z;
    x = z;
    f = -log(f);
  }
  z = 1/(x*x);
  return f + (x - 0.5)*log(x) - x + .918938533204673
       + (((-.000595238095238*z + .000793650793651)*z - .002777777777778)*z
          + .083333333333333)/x;
}

static double inc_gamma_as32(double x, double alpha, double ln_gamma_alpha)  /* gamma.c:28-96 */
{
  const double accurate = 1e-8, overflow = 1e30;
  double p = alpha, factor, gin, term, rn;
  int i;
  if (x == 0) return 0;
  if (x < 0 || p <= 0) return -1;
  factor = exp(p*log(x) - x - ln_gamma_alpha);
  if (!(x > 1 && x >= p))
  {
    /* series expansion */
    gin = 1; term = 1; rn = p;
    do { rn++; term *= x/rn; gin += term; } while (term > accurate);
    return gin*(factor/p);
  }
  else
  {
    /* continued fraction */
    double a = 1 - p, b = a + x + 1, an, dif, pn[6];
    term = 0;
    pn[0] = 1; pn[1] = x; pn[2] = x + 1; pn[3] = x*b;
    gin = pn[2]/pn[3];
    for (;;)
    {
      a++; b += 2; term++;
      an = a*term;
      for (i = 0; i < 2; ++i) pn[i+4] = b*pn[i+2] - an*pn[i];
      if (pn[5] != 0)
      {
        rn = pn[4]/pn[5];
        dif = fabs(gin - rn);
        if (dif <= accurate && dif <= accurate*rn) break;
        gin = rn;
      }
      for (i = 0; i < 4; ++i) pn[i] = pn[i+2];
      if (fabs(pn[4]) >= overflow)
        for (i = 0; i < 4; ++i) pn[i] /= overflow;
    }
    return 1 - factor*gin;
  }
}

static double point_normal_as70(double prob)                 /* gamma.c:134-159 */
{
  const double a0 = -.322232431088, a1 = -1, a2 = -.342242088547, a3 = -.0204231210245,
               a4 = -.453642210148e-4, b0 = .0993484626060, b1 = .588581570495,
               b2 = .531103462366, b3 = .103537752850, b4 = .0038560700634;
  double p1 = (prob < 0.5) ? prob : 1 - prob, y, z;
  if (p1 < 1e-20) return -9999;
  y = sqrt(log(1/(p1*p1)));
  z = y + ((((y*a4 + a3)*y + a2)*y + a1)*y + a0)/((((y*b4 + b3)*y + b2)*y + b1)*y + b0);
  return (prob < 0.5) ? -z : z;
}

static double point_chi2_as91(double prob, double v)          /* gamma.c:161-219 */
{
  const double e = .5e-6, aa = .6931471805;
  double p = prob, g, xx, c, ch, a, q, p1, p2, t, x, b, s1, s2, s3, s4, s5, s6;
  if (p < .000002 || p > .999998 || v <= 0) return -1;
  g = ln_gamma_ph(v/2);
  xx = v/2; c = xx - 1;
  if (v < -1.24*log(p))
  {
    ch = pow(p*xx*exp(g + xx*aa), 1/xx);
    if (ch - e < 0) return ch;
  }
  else if (v > .32)
  {
    x = point_normal_as70(p);
    p1 = 0.222222/v;
    ch = v*pow(x*sqrt(p1) + 1 - p1, 3.0);
    if (ch > 2.2*v + 6) ch = -2*(log(1 - p) - c*log(.5*ch) + g);
  }
  else
  {
    ch = 0.4; a = log(1 - p);
    do
    {
      q = ch; p1 = 1 + ch*(4.67 + ch); p2 = ch*(6.73 + ch*(6.66 + ch));
      t = -0.5 + (4.67 + 2*ch)/p1 - (6.73 + ch*(13.32 + 3*ch))/p2;
      ch -= (1 - exp(a + g + .5*ch + c*aa)*p2/p1)/t;
    } while (fabs(q/ch - 1) - .01 > 0);
  }
  do
  {
    q = ch; p1 = .5*ch;
    if ((t = inc_gamma_as32(p1, xx, g)) < 0.0) return -1;
    p2 = p - t;
    t = p2*exp(xx*aa + g + p1 - c*log(ch));
    b = t/ch; a = 0.5*t - b*c;
    s1 = (210 + a*(140 + a*(105 + a*(84 + a*(70 + 60*a)))))/420;
    s2 = (420 + a*(735 + a*(966 + a*(1141 + 1278*a))))/2520;
    s3 = (210 + a*(462 + a*(707 + 932*a)))/2520;
    s4 = (252 + a*(672 + 1182*a) + c*(294 + a*(889 + 1740*a)))/5040;
    s5 = (84 + 264*a + c*(175 + 606*a))/2520;
    s6 = (120 + c*(346 + 127*c))/5040;
    ch += t*(1 + 0.5*t*s1 - b*c*(s1 - b*(s2 - b*(s3 - b*(s4 - b*(s5 - b*s6))))));
  } while (fabs(q/ch - 1) > e);
  return ch;
}

int orc_gamma_cats(double alpha, double beta, unsigned categories, double * rates)
{
  unsigned i;
  double mean = alpha/beta, lnga1;
  double * cut;
  if (categories == 1) { rates[0] = 1.0; return 1; }
  cut = (double *)malloc(categories*sizeof(double));
  lnga1 = ln_gamma_ph(alpha + 1);
  for (i = 0; i + 1 < categories; ++i)
    cut[i] = point_chi2_as91((i + 1.0)/categories, 2.0*alpha)/(2.0*beta);
  for (i = 0; i + 1 < categories; ++i)
    cut[i] = inc_gamma_as32(cut[i]*beta, alpha + 1, lnga1);
  rates[0] = cut[0]*mean*categories;
  rates[categories-1] = (1 - cut[categories-2])*mean*categories;
  for (i = 1; i + 1 < categories; ++i)
    rates[i] = (cut[i] - cut[i-1])*mean*categories;
  free(cut);
  return 1;
}

/* --------------------------------------------------------------- a15 -------
 * compress_site_patterns (compress.c:218-376): alignment columns -> unique
 * patterns + integer weights.  With jc69 != 0, columns made only of
 * unambiguous nucleotides and gaps (codes 1,2,4,8,15) are first relabelled in
 * order of first appearance so that columns equal up to a permutation of the
 * four nucleotides merge (compress.c:161-216).  The representative of a merged
 * class is the class's first member in the reference's (randomly pivoted)
 * sort order and is therefore not a contract; here it is the member with the
 * lowest original column index, and patterns come out in lexicographic order
 * of their (relabelled) code strings.  Compare as multisets of canonical
 * columns.  Sequences are compacted in place to *length patterns.
 * Reference quirk kept on purpose (pattern counts are the contract): relabelled
 * codes 1,2,3,4 share the code space of un-relabelled columns, so a column
 * holding the ambiguity code 3 (M) — e.g. A,C,M — compares equal to a
 * relabelled A,G,T = 1,2,3 and the two merge (compress.c:293-337).            */
typedef struct { const unsigned char * key; int orig; } col_t;
static int g_collen;
static int col_cmp(const void * a, const void * b)
{
  const col_t * x = (const col_t *)a, * y = (const col_t *)b;
  int c = memcmp(x->key, y->key, (size_t)g_collen);
  return c ? c : (x->orig - y->orig);
}

int orc_compress(char ** seqs, int count, int * length, const unsigned * map,
                 int jc69, unsigned * weights)
{
  int len = *length, i, j, np = 0;
  unsigned char * keys;
  col_t * cols;
  char * out;
  if (!count || !map || map[0]) return 0;
  keys = (unsigned char *)malloc((size_t)len*count*4);
  cols = (col_t *)malloc((size_t)len*sizeof(col_t));
  out  = (char *)malloc((size_t)len*count);

  /* key of a column = its state codes, 4 bytes each (AA codes need 20 bits),
     big-endian so memcmp orders numerically */
  for (i = 0; i < len; ++i)
  {
    unsigned char * key = keys + (size_t)i*count*4;
    unsigned relabel[16]; unsigned next = 1; int simple = jc69;
    memset(relabel, 0, sizeof(relabel));
    relabel[15] = 15;
    if (jc69)
      for (j = 0; j < count; ++j)
      {
        unsigned c = map[(unsigned char)seqs[j][i]];
        if (!(c == 1 || c == 2 || c == 4 || c == 8 || c == 15)) { simple = 0; break; }
      }
    for (j = 0; j < count; ++j)
    {
      unsigned c = map[(unsigned char)seqs[j][i]];
      if (simple)
      {
        if (!relabel[c]) relabel[c] = next++;
        c = relabel[c];
      }
      key[4*j] = (unsigned char)(c >> 24); key[4*j+1] = (unsigned char)(c >> 16);
      key[4*j+2] = (unsigned char)(c >> 8); key[4*j+3] = (unsigned char)c;
    }
    cols[i].key = key; cols[i].orig = i;
  }
  g_collen = count*4;
  qsort(cols, (size_t)len, sizeof(col_t), col_cmp);

  for (i = 0; i < len; ++i)
  {
    if (i && !memcmp(cols[i].key, cols[i-1].key, (size_t)g_collen))
      weights[np-1]++;
    else
    {
      for (j = 0; j < count; ++j) out[(size_t)j*len + np] = seqs[j][cols[i].orig];
      weights[np++] = 1;
    }
  }
  for (j = 0; j < count; ++j)
  {
    memcpy(seqs[j], out + (size_t)j*len, (size_t)np);
    seqs[j][np] = 0;
  }
  *length = np;
  free(keys); free(cols); free(out);
  return 1;
}

/* --------------------------------------------------------------- tape ------
 * Replay of a proposal tape in the explicit-index form the batched engine
 * consumes (include/bpp_amd.h: bpa_batch_t) for ONE locus, with this file's
 * kernels: per step the listed P-matrices (JC69 closed form or eigen form),
 * the listed node updates, then the root log-likelihood — the body of every
 * proposal of the reference (gtree.c:5447-5467).  CPU-baseline "port" leg and
 * parity checker.  clv[]: tips first then inner buffers; pmat[]/scaler[] by
 * buffer index.  Returns elapsed seconds for `repeats` passes.               */
#include <time.h>
typedef struct { unsigned parent_clv; int parent_scaler; unsigned left_clv, left_pmatrix;
                 int left_scaler; unsigned right_clv, right_pmatrix; int right_scaler; } orc_op_t;

double orc_run_tape(unsigned states, unsigned sites, unsigned rate_cats, int jc69, int order,
                    double ** clv, double ** pmat, unsigned ** scaler,
                    const double * rates, const double * rate_weights, const double * freqs,
                    const double * eigenvals, const double * eigenvecs, const double * inv_eigenvecs,
                    const unsigned * weights, unsigned nsteps,
                    const unsigned * mat_off, const unsigned * mat_pmatrix, const double * mat_length,
                    const unsigned * op_off, const orc_op_t * ops,
                    const unsigned * root_clv, const int * root_scaler,
                    double * lnl_out, unsigned repeats)
{
  struct timespec t0, t1;
  unsigned r, s, i;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (r = 0; r < repeats; ++r)
    for (s = 0; s < nsteps; ++s)
    {
      for (i = mat_off[s]; i < mat_off[s+1]; ++i)
      {
        if (jc69) orc_pmatrix_jc69(rate_cats, rates, mat_length[i], pmat[mat_pmatrix[i]]);
        else orc_pmatrix_eigen(states, rate_cats, rates, mat_length[i], eigenvals, eigenvecs,
                               inv_eigenvecs, pmat[mat_pmatrix[i]], 0);
      }
      for (i = op_off[s]; i < op_off[s+1]; ++i)
      {
        const orc_op_t * o = ops + i;
        orc_update_partial_ii(states, sites, rate_cats, clv[o->parent_clv],
                              o->parent_scaler >= 0 ? scaler[o->parent_scaler] : NULL,
                              clv[o->left_clv], clv[o->right_clv],
                              pmat[o->left_pmatrix], pmat[o->right_pmatrix],
                              o->left_scaler >= 0 ? scaler[o->left_scaler] : NULL,
                              o->right_scaler >= 0 ? scaler[o->right_scaler] : NULL, order);
      }
      lnl_out[s] = orc_root_loglikelihood(states, sites, rate_cats, clv[root_clv[s]],
                                          root_scaler[s] >= 0 ? scaler[root_scaler[s]] : NULL,
                                          freqs, rate_weights, weights, NULL, order);
    }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  return (t1.tv_sec - t0.tv_sec) + 1e-9*(t1.tv_nsec - t0.tv_nsec);
}
