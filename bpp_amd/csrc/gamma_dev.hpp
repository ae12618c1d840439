// gamma_dev.hpp — pll_compute_gamma_cats (gamma.c:221-284, mean mode) on the DEVICE: the generic sampler's alpha move
// (gsampler.hpp; prop_gamma.c:93) turns a proposed shape into category rates without a host round trip.  The same
// published routines in the same evaluation order as csrc/host_math.cpp (Pike & Hill 1966 Alg. 291; Bhattacharjee 1970
// AS 32; Best & Roberts 1975 AS 91; Odeh & Evans 1974 AS 70) — device exp / log / pow differ from glibc's in the last
// place, so the rates agree with bpa_compute_gamma_cats to ~1e-15 relative, not to the bit.
#pragma once

namespace gdev {

__device__ inline double lngamma(double alpha)
{
  double x = alpha, f = 0.0;
  if (x < 7.0)
  {
    f = 1.0;
    double z = alpha - 1.0;
    for (z = z + 1.0; z < 7.0; z = z + 1.0) f *= z;
    x = z;
    f = -log(f);
  }
  const double z = 1/(x*x);
  return f + (x - 0.5)*log(x) - x + .918938533204673
       + (((-.000595238095238*z + .000793650793651)*z - .002777777777778)*z + .083333333333333)/x;
}

__device__ inline double incomplete_gamma(double x, double alpha, double ln_gamma_alpha)
{
  const double accurate = 1e-8, overflow = 1e30;
  if (x == 0) return 0;
  if (x < 0 || alpha <= 0) return -1;
  const double factor = exp(alpha*log(x) - x - ln_gamma_alpha);
  if (!(x > 1 && x >= alpha))
  {
    double gin = 1, term = 1, rn = alpha;
    do { rn++; term *= x/rn; gin += term; } while (term > accurate);
    gin *= factor/alpha;
    return gin;
  }
  double a = 1 - alpha, b = a + x + 1, term = 0;
  double p0 = 1, p1 = x, p2 = x + 1, p3 = x*b, p4 = 0, p5 = 0;
  double gin = p2/p3;
  for (int guard = 0; guard < 100000; ++guard)
  {
    a++; b += 2; term++;
    const double an = a*term;
    p4 = b*p2 - an*p0;
    p5 = b*p3 - an*p1;
    if (p5 != 0)
    {
      const double rn = p4/p5;
      const double dif = fabs(gin - rn);
      if (dif <= accurate && dif <= accurate*rn) break;
      gin = rn;
    }
    p0 = p2; p1 = p3; p2 = p4; p3 = p5;
    if (fabs(p4) >= overflow) { p0 /= overflow; p1 /= overflow; p2 /= overflow; p3 /= overflow; }
  }
  return 1 - factor*gin;
}

__device__ inline double normal_quantile(double prob)
{
  const double a0 = -.322232431088, a1 = -1, a2 = -.342242088547, a3 = -.0204231210245,
               a4 = -.453642210148e-4, b0 = .0993484626060, b1 = .588581570495,
               b2 = .531103462366, b3 = .103537752850, b4 = .0038560700634;
  const double p1 = prob < 0.5 ? prob : 1 - prob;
  if (p1 < 1e-20) return -9999;
  const double y = sqrt(log(1/(p1*p1)));
  const double z = y + ((((y*a4 + a3)*y + a2)*y + a1)*y + a0)/((((y*b4 + b3)*y + b2)*y + b1)*y + b0);
  return prob < 0.5 ? -z : z;
}

__device__ inline double chi2_quantile(double p, double v)
{
  const double e = .5e-6, aa = .6931471805;
  if (p < .000002 || p > .999998 || v <= 0) return -1;
  const double g = lngamma(v/2), xx = v/2, c = xx - 1;
  double ch;
  if (v < -1.24*log(p))
  {
    ch = pow(p*xx*exp(g + xx*aa), 1/xx);
    if (ch - e < 0) return ch;
  }
  else if (v > .32)
  {
    const double x = normal_quantile(p), p1 = 0.222222/v;
    ch = v*pow(x*sqrt(p1) + 1 - p1, 3.0);
    if (ch > 2.2*v + 6) ch = -2*(log(1 - p) - c*log(.5*ch) + g);
  }
  else
  {
    ch = 0.4;
    const double a = log(1 - p);
    double q;
    int guard = 0;
    do
    {
      q = ch;
      const double p1 = 1 + ch*(4.67 + ch), p2 = ch*(6.73 + ch*(6.66 + ch));
      const double t = -0.5 + (4.67 + 2*ch)/p1 - (6.73 + ch*(13.32 + 3*ch))/p2;
      ch -= (1 - exp(a + g + .5*ch + c*aa)*p2/p1)/t;
    } while (fabs(q/ch - 1) - .01 > 0 && ++guard < 1000);
  }
  double q;
  int guard = 0;
  do
  {
    q = ch;
    const double p1 = .5*ch;
    double t = incomplete_gamma(p1, xx, g);
    if (t < 0.0) return -1;
    const double p2 = p - t;
    t = p2*exp(xx*aa + g + p1 - c*log(ch));
    const double b = t/ch, a = 0.5*t - b*c;
    const double s1 = (210 + a*(140 + a*(105 + a*(84 + a*(70 + 60*a)))))/420;
    const double s2 = (420 + a*(735 + a*(966 + a*(1141 + 1278*a))))/2520;
    const double s3 = (210 + a*(462 + a*(707 + 932*a)))/2520;
    const double s4 = (252 + a*(672 + 1182*a) + c*(294 + a*(889 + 1740*a)))/5040;
    const double s5 = (84 + 264*a + c*(175 + 606*a))/2520;
    const double s6 = (120 + c*(346 + 127*c))/5040;
    ch += t*(1 + 0.5*t*s1 - b*c*(s1 - b*(s2 - b*(s3 - b*(s4 - b*(s5 - b*s6))))));
  } while (fabs(q/ch - 1) > e && ++guard < 1000);
  return ch;
}

// rates[0 .. categories): the class means of Gamma(alpha, alpha); categories <= 8
__device__ inline void gamma_cats(double alpha, unsigned categories, double * rates)
{
  if (categories <= 1) { rates[0] = 1.0; return; }
  const double beta = alpha, mean = alpha/beta;
  const double lnga1 = lngamma(alpha + 1);
  double cut[8];
  for (unsigned i = 0; i + 1 < categories; ++i)
    cut[i] = chi2_quantile((i + 1.0)/categories, 2.0*alpha)/(2.0*beta);
  for (unsigned i = 0; i + 1 < categories; ++i)
    cut[i] = incomplete_gamma(cut[i]*beta, alpha + 1, lnga1);
  rates[0] = cut[0]*mean*categories;
  rates[categories - 1] = (1 - cut[categories - 2])*mean*categories;
  for (unsigned i = 1; i + 1 < categories; ++i) rates[i] = (cut[i] - cut[i-1])*mean*categories;
}

} // namespace gdev
