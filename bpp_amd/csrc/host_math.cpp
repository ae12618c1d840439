// host_math.cpp — the parts of the path that the reference also runs as host
// scalar / integer code at start-up or once per proposal, kept on the host here
// too (SURVEY.md §8a rows a14, a15 and the state tables of maps.c):
//
//   bpa_compute_gamma_cats       pll_compute_gamma_cats       gamma.c:221-284
//   bpa_compress_site_patterns   compress_site_patterns       compress.c:218-376
//   bpa_map_nt / bpa_map_aa      pll_map_nt / pll_map_aa      maps.c:26,126
//
// Build with -ffp-contract=off: the discrete-gamma rates feed every P-matrix, so
// they are kept bit-identical to the reference's (gamma.c is built without FMA).
#include <cmath>
#include <cstring>
#include <cstdint>
#include <vector>
#include <array>
#include <algorithm>
#include <numeric>
#include "bpp_amd.h"

// ------------------------------------------------------------- state tables --
namespace {

struct Maps
{
  unsigned nt[256];
  unsigned aa[256];
  static void both_cases(unsigned * tab, char c, unsigned code)
  {
    tab[(unsigned char)c] = code;
    if (c >= 'A' && c <= 'Z') tab[(unsigned char)(c + ('a' - 'A'))] = code;
  }
  Maps()
  {
    std::memset(nt, 0, sizeof(nt));
    std::memset(aa, 0, sizeof(aa));
    // nucleotides: bit set over {A,C,G,T}; IUPAC ambiguity = union of its members
    struct { char c; const char * members; } iupac[] = {
      {'A',"A"},{'C',"C"},{'G',"G"},{'T',"T"},{'U',"T"},{'R',"AG"},{'Y',"CT"},{'S',"CG"},
      {'W',"AT"},{'K',"GT"},{'M',"AC"},{'B',"CGT"},{'D',"AGT"},{'H',"ACT"},{'V',"ACG"},
      {'N',"ACGT"},{'X',"ACGT"},{'O',"ACGT"},{'-',"ACGT"},{'?',"ACGT"}};
    for (auto & e : iupac)
    {
      unsigned code = 0;
      for (const char * m = e.members; *m; ++m) code |= 1u << (std::strchr("ACGT", *m) - "ACGT");
      both_cases(nt, e.c, code);
    }
    // amino acids: one-hot in the order ARNDCQEGHILKMFPSTWYV; B = N|D, Z = Q|E
    const char * order = "ARNDCQEGHILKMFPSTWYV";
    for (int i = 0; i < 20; ++i) both_cases(aa, order[i], 1u << i);
    both_cases(aa, 'B', aa[(unsigned char)'N'] | aa[(unsigned char)'D']);
    both_cases(aa, 'Z', aa[(unsigned char)'Q'] | aa[(unsigned char)'E']);
    for (char c : {'X', '*', '-', '?'}) both_cases(aa, c, (1u << 20) - 1);
  }
};
const Maps & maps() { static Maps m; return m; }

} // namespace

extern "C" const unsigned * bpa_map_nt(void) { return maps().nt; }
extern "C" const unsigned * bpa_map_aa(void) { return maps().aa; }

// ------------------------------------------------------ discrete-gamma rates --
// Mean rate of each of `categories` equal-probability classes of Gamma(alpha,beta):
// class boundaries from the chi-square quantile (Best & Roberts 1975, AS 91, with
// the normal quantile of Odeh & Evans 1974, AS 70), class means from the
// incomplete gamma ratio at shape alpha+1 (Bhattacharjee 1970, AS 32), log-gamma
// by Pike & Hill (1966, Alg. 291) — the same published routines, in the same
// evaluation order, as gamma.c:28-219.
namespace {

double lngamma(double alpha)
{
  double x = alpha, f = 0.0;
  if (x < 7.0)
  {
    f = 1.0;
    double z = alpha - 1.0;
    for (z = z + 1.0; z < 7.0; z = z + 1.0) f *= z;
    x = z;
    f = -std::log(f);
  }
  const double z = 1/(x*x);
  return f + (x - 0.5)*std::log(x) - x + .918938533204673
       + (((-.000595238095238*z + .000793650793651)*z - .002777777777778)*z + .083333333333333)/x;
}

double incomplete_gamma(double x, double alpha, double ln_gamma_alpha)
{
  const double accurate = 1e-8, overflow = 1e30;
  if (x == 0) return 0;
  if (x < 0 || alpha <= 0) return -1;
  const double factor = std::exp(alpha*std::log(x) - x - ln_gamma_alpha);
  if (!(x > 1 && x >= alpha))
  {
    double gin = 1, term = 1, rn = alpha;
    do { rn++; term *= x/rn; gin += term; } while (term > accurate);
    gin *= factor/alpha;
    return gin;
  }
  double a = 1 - alpha, b = a + x + 1, term = 0;
  std::array<double, 6> pn{1, x, x + 1, x*b, 0, 0};
  double gin = pn[2]/pn[3];
  for (;;)
  {
    a++; b += 2; term++;
    const double an = a*term;
    pn[4] = b*pn[2] - an*pn[0];
    pn[5] = b*pn[3] - an*pn[1];
    if (pn[5] != 0)
    {
      const double rn = pn[4]/pn[5];
      const double dif = std::fabs(gin - rn);
      if (dif <= accurate && dif <= accurate*rn) break;
      gin = rn;
    }
    for (int i = 0; i < 4; ++i) pn[i] = pn[i+2];
    if (std::fabs(pn[4]) >= overflow)
      for (int i = 0; i < 4; ++i) pn[i] /= overflow;
  }
  return 1 - factor*gin;
}

double normal_quantile(double prob)
{
  const double a0 = -.322232431088, a1 = -1, a2 = -.342242088547, a3 = -.0204231210245,
               a4 = -.453642210148e-4, b0 = .0993484626060, b1 = .588581570495,
               b2 = .531103462366, b3 = .103537752850, b4 = .0038560700634;
  const double p1 = prob < 0.5 ? prob : 1 - prob;
  if (p1 < 1e-20) return -9999;
  const double y = std::sqrt(std::log(1/(p1*p1)));
  const double z = y + ((((y*a4 + a3)*y + a2)*y + a1)*y + a0)/((((y*b4 + b3)*y + b2)*y + b1)*y + b0);
  return prob < 0.5 ? -z : z;
}

double chi2_quantile(double p, double v)
{
  const double e = .5e-6, aa = .6931471805;
  if (p < .000002 || p > .999998 || v <= 0) return -1;
  const double g = lngamma(v/2), xx = v/2, c = xx - 1;
  double ch;
  if (v < -1.24*std::log(p))
  {
    ch = std::pow(p*xx*std::exp(g + xx*aa), 1/xx);
    if (ch - e < 0) return ch;
  }
  else if (v > .32)
  {
    const double x = normal_quantile(p), p1 = 0.222222/v;
    ch = v*std::pow(x*std::sqrt(p1) + 1 - p1, 3.0);
    if (ch > 2.2*v + 6) ch = -2*(std::log(1 - p) - c*std::log(.5*ch) + g);
  }
  else
  {
    ch = 0.4;
    const double a = std::log(1 - p);
    double q;
    do
    {
      q = ch;
      const double p1 = 1 + ch*(4.67 + ch), p2 = ch*(6.73 + ch*(6.66 + ch));
      const double t = -0.5 + (4.67 + 2*ch)/p1 - (6.73 + ch*(13.32 + 3*ch))/p2;
      ch -= (1 - std::exp(a + g + .5*ch + c*aa)*p2/p1)/t;
    } while (std::fabs(q/ch - 1) - .01 > 0);
  }
  double q;
  do
  {
    q = ch;
    const double p1 = .5*ch;
    double t = incomplete_gamma(p1, xx, g);
    if (t < 0.0) return -1;
    const double p2 = p - t;
    t = p2*std::exp(xx*aa + g + p1 - c*std::log(ch));
    const double b = t/ch, a = 0.5*t - b*c;
    const double s1 = (210 + a*(140 + a*(105 + a*(84 + a*(70 + 60*a)))))/420;
    const double s2 = (420 + a*(735 + a*(966 + a*(1141 + 1278*a))))/2520;
    const double s3 = (210 + a*(462 + a*(707 + 932*a)))/2520;
    const double s4 = (252 + a*(672 + 1182*a) + c*(294 + a*(889 + 1740*a)))/5040;
    const double s5 = (84 + 264*a + c*(175 + 606*a))/2520;
    const double s6 = (120 + c*(346 + 127*c))/5040;
    ch += t*(1 + 0.5*t*s1 - b*c*(s1 - b*(s2 - b*(s3 - b*(s4 - b*(s5 - b*s6))))));
  } while (std::fabs(q/ch - 1) > e);
  return ch;
}

} // namespace

extern "C" int bpa_compute_gamma_cats(double alpha, double beta, unsigned categories, double * rates)
{
  if (!categories || !rates) return 0;
  if (categories == 1) { rates[0] = 1.0; return 1; }
  const double mean = alpha/beta;
  const double lnga1 = lngamma(alpha + 1);
  std::vector<double> cut(categories - 1);
  for (unsigned i = 0; i + 1 < categories; ++i)
    cut[i] = chi2_quantile((i + 1.0)/categories, 2.0*alpha)/(2.0*beta);
  for (unsigned i = 0; i + 1 < categories; ++i)
    cut[i] = incomplete_gamma(cut[i]*beta, alpha + 1, lnga1);
  rates[0] = cut[0]*mean*categories;
  rates[categories - 1] = (1 - cut[categories - 2])*mean*categories;
  for (unsigned i = 1; i + 1 < categories; ++i) rates[i] = (cut[i] - cut[i-1])*mean*categories;
  return 1;
}

// --------------------------------------------------- site-pattern compression --
// Unique alignment columns + integer weights.  A column's key is the vector of
// its state codes; with jc69 set, columns holding only unambiguous nucleotides and
// gaps (codes 1,2,4,8,15) are first renamed in order of first appearance so that
// columns equal up to a permutation of the nucleotides merge (compress.c:161-216).
// Renamed codes 1..4 share the code space of un-renamed columns exactly as in the
// reference (a column A,C,M = 1,2,3 merges with a renamed A,G,T): pattern counts are
// the contract.  The class representative is its lowest-index member; patterns are
// emitted in lexicographic key order, which is the order the reference's multikey quicksort
// (compress.c:35-101) ends in; its rand() pivots only decide which member represents a class.
extern "C" int bpa_compress_site_patterns(char ** sequences, const unsigned * map, int count,
                                          int * length, int jc69, unsigned * weights)
{
  if (!sequences || !map || count <= 0 || !length || *length <= 0 || map[0]) return 0;
  const int len = *length;
  // sort key of a character: its state code; when the codes do not fit a byte (amino acids) the rank
  // of the code by the first character carrying it (remap_range, compress.c:104-128) — the order of
  // the emitted patterns is then the reference's
  uint32_t code[256];
  {
    uint32_t mx = 0;
    for (int c = 0; c < 256; ++c) mx = std::max<uint32_t>(mx, map[c]);
    uint32_t k = 1;
    for (int c = 0; c < 256; ++c) code[c] = mx < 256 ? map[c] : 0;
    if (mx >= 256)
      for (int c = 0; c < 256; ++c)
        if (map[c] && !code[c])
        {
          for (int e = c; e < 256; ++e) if (map[e] == map[c]) code[e] = k;
          ++k;
        }
  }
  std::vector<uint32_t> keys((size_t)len*count);
  for (int i = 0; i < len; ++i)
  {
    uint32_t * key = &keys[(size_t)i*count];
    bool simple = jc69 != 0;
    for (int j = 0; j < count; ++j)
    {
      key[j] = code[(unsigned char)sequences[j][i]];
      if (!key[j]) return 0;
      if (!(key[j] == 1 || key[j] == 2 || key[j] == 4 || key[j] == 8 || key[j] == 15)) simple = false;
    }
    if (simple)
    {
      uint32_t rename[16] = {0}; rename[15] = 15;
      uint32_t next = 1;
      for (int j = 0; j < count; ++j)
      {
        if (!rename[key[j]]) rename[key[j]] = next++;
        key[j] = rename[key[j]];
      }
    }
  }
  std::vector<int> order(len);
  std::iota(order.begin(), order.end(), 0);
  auto less = [&](int a, int b)
  {
    const uint32_t * x = &keys[(size_t)a*count], * y = &keys[(size_t)b*count];
    for (int j = 0; j < count; ++j) if (x[j] != y[j]) return x[j] < y[j];
    return a < b;
  };
  std::sort(order.begin(), order.end(), less);
  std::vector<char> out((size_t)count*len);
  int np = 0;
  for (int r = 0; r < len; ++r)
  {
    const int i = order[r];
    const bool same = r && std::equal(&keys[(size_t)i*count], &keys[(size_t)i*count] + count,
                                      &keys[(size_t)order[r-1]*count]);
    if (same) weights[np-1]++;
    else
    {
      for (int j = 0; j < count; ++j) out[(size_t)j*len + np] = sequences[j][i];
      weights[np++] = 1;
    }
  }
  for (int j = 0; j < count; ++j)
  {
    std::memcpy(sequences[j], &out[(size_t)j*len], (size_t)np);
    sequences[j][np] = 0;
  }
  *length = np;
  return np;
}
