// Input side of the likelihood path (include/bpp_amd_input.h).
// Host-only.  Each function states the behaviour of the reference routine it replaces:
//
//   bpa_phylip_read                  phylip_parse_multisequential   phylip.c:467-681
//   bpa_msa_remove_missing_sequences msa_remove_missing_sequences   msa.c:245-307
//   bpa_msa_count_ambiguous_sites    msa_count_ambiguous_sites      msa.c:137-156
//   bpa_msa_remove_ambiguous         msa_remove_ambiguous           msa.c:158-243
//   bpa_imap_read                    parse_mapfile                  parsemap.c:89-276
//   bpa_msa_diploid_resolve          diploid_resolve_locus          diploid.c:307-647
//   bpa_msa_compress_diploid         compress_site_patterns_diploid compress.c:378-547
//   bpa_msa_write_phylip             msa_print_phylip               msa.c:45-135
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <string>
#include <vector>
#include <algorithm>
#include <numeric>
#include "bpp_amd.h"
#include "bpp_amd_input.h"

extern "C" void bpa_internal_set_error(const char * msg);      // engine.hip

namespace {

int fail(const std::string & m) { bpa_internal_set_error(m.c_str()); return 0; }

struct Tables
{
  unsigned fasta[256], amb[256], nt_missing[256], aa_missing[256];
  unsigned char nt_print[256];
  Tables()
  {
    // sequence-reader classes: control characters are fatal except TAB..CR (silently dropped);
    // letters are legal except lower-case j and o; digits, '-' and '?' are legal; '.' is fatal;
    // every other byte is dropped and counted
    for (int c = 0; c < 256; ++c)
    {
      unsigned s = 0;
      if (c < 32) s = (c >= 9 && c <= 13) ? 3 : 2;
      else if (c == '.') s = 2;
      else if (c == '-' || c == '?' || (c >= '0' && c <= '9')) s = 1;
      else if (c >= 'A' && c <= 'Z') s = 1;
      else if (c >= 'a' && c <= 'z') s = (c == 'j' || c == 'o') ? 0 : 1;
      fasta[c] = s;
      amb[c] = nt_missing[c] = aa_missing[c] = 0;
      nt_print[c] = 0;
    }
    for (const char * p = "-?BDHKMNORSVWXY"; *p; ++p)
    {
      amb[(unsigned char)*p] = 1;
      if (*p >= 'A') amb[(unsigned char)(*p + 32)] = 1;
    }
    for (const char * p = "-?NXnx"; *p; ++p) nt_missing[(unsigned char)*p] = 1;
    aa_missing[(unsigned char)'-'] = aa_missing[(unsigned char)'?'] = 1;
    // characters the DNA writer prints (bpp_nt_normal, msa.c:24-42): upper case, U as T
    for (const char * p = "ABCDGHKMNORSTVWXY"; *p; ++p)
      nt_print[(unsigned char)*p] = nt_print[(unsigned char)(*p + 32)] = (unsigned char)*p;
    nt_print[(unsigned char)'U'] = nt_print[(unsigned char)'u'] = 'T';
    nt_print[(unsigned char)'-'] = '-';
    nt_print[(unsigned char)'?'] = '?';
  }
};
const Tables & tables() { static const Tables t; return t; }

bool is_ws(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r'; }
bool blank(const std::string & s) { return s.find_first_not_of(" \t\r\n") == std::string::npos; }

// one line without its '\n' (a '\r' stays and is white space to every consumer)
struct LineReader
{
  FILE * fp;
  long lineno = 0;
  explicit LineReader(FILE * f) : fp(f) {}
  bool next(std::string & out)
  {
    out.clear();
    char buf[4096];
    bool any = false;
    while (fgets(buf, sizeof buf, fp))
    {
      any = true;
      size_t n = strlen(buf);
      if (n && buf[n-1] == '\n') { out.append(buf, n - 1); ++lineno; return true; }
      out.append(buf, n);
    }
    if (any) ++lineno;
    return any;
  }
};

} // namespace

struct bpa_msa
{
  std::vector<std::string> label, seq;
  int length = 0;
};

struct bpa_imap
{
  std::vector<std::string> individual, species;
};

extern "C" const unsigned * bpa_map_fasta(void)      { return tables().fasta; }
extern "C" const unsigned * bpa_map_amb(void)        { return tables().amb; }
extern "C" const unsigned * bpa_map_nt_missing(void) { return tables().nt_missing; }
extern "C" const unsigned * bpa_map_aa_missing(void) { return tables().aa_missing; }

// ------------------------------------------------------------- alignment object --
extern "C" bpa_msa_t * bpa_msa_create(int count, int length, const char * const * labels,
                                      const char * const * sequences)
{
  if (count <= 0 || length <= 0 || !labels || !sequences) { fail("bpa_msa_create: bad arguments"); return nullptr; }
  bpa_msa * m = new bpa_msa;
  m->length = length;
  for (int i = 0; i < count; ++i)
  {
    if (!labels[i] || !sequences[i] || (int)strlen(sequences[i]) != length)
    { delete m; fail("bpa_msa_create: sequence " + std::to_string(i + 1) + " does not have the alignment length"); return nullptr; }
    m->label.emplace_back(labels[i]);
    m->seq.emplace_back(sequences[i]);
  }
  return m;
}
extern "C" void bpa_msa_destroy(bpa_msa_t * m) { delete m; }
extern "C" int  bpa_msa_count(const bpa_msa_t * m) { return (int)m->seq.size(); }
extern "C" int  bpa_msa_length(const bpa_msa_t * m) { return m->length; }
extern "C" const char * bpa_msa_label(const bpa_msa_t * m, int i)
{ return i >= 0 && i < (int)m->label.size() ? m->label[i].c_str() : nullptr; }
extern "C" const char * bpa_msa_sequence(const bpa_msa_t * m, int i)
{ return i >= 0 && i < (int)m->seq.size() ? m->seq[i].c_str() : nullptr; }
extern "C" void bpa_msa_list_free(bpa_msa_t ** list, long count)
{
  if (!list) return;
  for (long i = 0; i < count; ++i) delete list[i];
  free(list);
}

// ----------------------------------------------------------------- PHYLIP reader --
namespace {

// appends the legal characters of `text` to seq (at most `length`); -1 on error
int take_sites(const std::string & text, size_t from, std::string & seq, int length, int seqno,
               const std::string & label, long lineno)
{
  const unsigned * cls = tables().fasta;
  int n = 0;
  for (size_t i = from; i < text.size(); ++i)
  {
    const unsigned char c = (unsigned char)text[i];
    if (!c) break;
    switch (cls[c])
    {
      case 1:
        if ((int)seq.size() >= length)
        { fail("Sequence " + std::to_string(seqno + 1) + " (" + label.substr(0, 100) + ") longer than expected"); return -1; }
        seq.push_back((char)c); ++n;
        break;
      case 2:
      {
        char msg[200];
        if (c >= 32) snprintf(msg, sizeof msg, "illegal character '%c' on line %ld in the fasta file", c, lineno);
        else snprintf(msg, sizeof msg, "illegal unprintable character %#.2x (hexadecimal) on line %ld in the fasta file", c, lineno);
        fail(msg);
        return -1;
      }
      default: break;                          // 0 (stripped) and 3 (silently stripped)
    }
  }
  return n;
}

// "count length" and nothing else on the line (sequential format, phylip.c:167-219)
bool parse_header(const std::string & line, int & count, int & length)
{
  int used = 0, v = 0;
  const char * p = line.c_str();
  if (sscanf(p, "%d%n", &v, &used) < 1 || !used || !v) { fail("Invalid number of sequences in header"); return false; }
  count = v; p += used; used = 0; v = 0;
  if (sscanf(p, "%d%n", &v, &used) < 1 || !used || !v) { fail("Invalid sequence length in header"); return false; }
  length = v; p += used;
  while (*p && is_ws(*p)) ++p;
  if (*p) { fail("Invalid PHYLIP header '" + line.substr(0, 100) + "'"); return false; }
  if (count < 0 || length < 0) { fail("Invalid PHYLIP header '" + line.substr(0, 100) + "'"); return false; }
  return true;
}

// one alignment whose header is `line` (already read); phylip_parse_sequential, phylip.c:467-620
bpa_msa * read_alignment(LineReader & rd, const std::string & header)
{
  int count = 0, length = 0;
  if (!parse_header(header, count, length)) return nullptr;
  bpa_msa * m = new bpa_msa;
  m->length = length;
  std::string line;
  for (int seqno = 0; seqno < count; ++seqno)
  {
    // next line that is not blank
    size_t p = 0;
    for (;;)
    {
      if (!rd.next(line))
      { fail("Found " + std::to_string(seqno) + " sequence(s) but expected " + std::to_string(count)); delete m; return nullptr; }
      p = 0;
      while (p < line.size() && is_ws(line[p])) ++p;
      if (p < line.size()) break;
    }
    // the label ends at the first blank of the line; failing that at the first tab, then CR
    size_t end = line.find(' ', p);
    if (end == std::string::npos) end = line.find('\t', p);
    if (end == std::string::npos) end = line.find('\r', p);
    if (end == std::string::npos) end = line.size();
    std::string label = line.substr(p, end - p);
    std::string seq;
    seq.reserve((size_t)length);
    size_t from = end;
    for (;;)
    {
      if (take_sites(line, from, seq, length, seqno, label, rd.lineno) < 0) { delete m; return nullptr; }
      if ((int)seq.size() == length) break;
      if (!rd.next(line))
      {
        fail("Sequence " + std::to_string(seqno + 1) + " (" + label.substr(0, 100) + ") has " + std::to_string(seq.size()) +
             " characters but expected " + std::to_string(length));
        delete m; return nullptr;
      }
      from = 0;
    }
    m->label.push_back(std::move(label));
    m->seq.push_back(std::move(seq));
  }
  return m;
}

} // namespace

extern "C" int bpa_phylip_read(const char * path, long max_loci, bpa_msa_t *** out, long * count)
{
  if (!path || !out || !count) return fail("bpa_phylip_read: null argument");
  FILE * fp = fopen(path, "r");
  if (!fp) return fail(std::string("Unable to open file (") + path + ")");
  LineReader rd(fp);
  std::vector<bpa_msa *> list;
  std::string line;
  bool have = rd.next(line);
  while (have && blank(line)) have = rd.next(line);
  if (!have) { fclose(fp); return fail(std::string("No alignment in file (") + path + ")"); }
  for (;;)
  {
    bpa_msa * m = read_alignment(rd, line);
    if (!m) { for (bpa_msa * x : list) delete x; fclose(fp); return 0; }
    list.push_back(m);
    if (max_loci > 0 && (long)list.size() == max_loci) break;
    have = rd.next(line);
    while (have && blank(line)) have = rd.next(line);
    if (!have) break;
  }
  fclose(fp);
  bpa_msa_t ** arr = (bpa_msa_t **)malloc(list.size()*sizeof(bpa_msa_t *));
  if (!arr) { for (bpa_msa * x : list) delete x; return fail("bpa_phylip_read: out of memory"); }
  for (size_t i = 0; i < list.size(); ++i) arr[i] = list[i];
  *out = arr;
  *count = (long)list.size();
  return 1;
}

// ------------------------------------------------------- cleaning the alignment --
extern "C" int bpa_msa_remove_missing_sequences(bpa_msa_t * m, int dtype)
{
  const unsigned * miss = dtype == BPA_DATA_DNA ? tables().nt_missing : tables().aa_missing;
  std::vector<std::string> label, seq;
  int deleted = 0;
  for (size_t i = 0; i < m->seq.size(); ++i)
  {
    bool all = true;
    for (char c : m->seq[i]) if (!miss[(unsigned char)c]) { all = false; break; }
    if (all) ++deleted;
    else { label.push_back(m->label[i]); seq.push_back(m->seq[i]); }
  }
  if (deleted == (int)m->seq.size()) return -1;
  m->label.swap(label); m->seq.swap(seq);
  return deleted;
}

static std::vector<int> ambiguous_sites(const bpa_msa * m)
{
  const unsigned * amb = tables().amb;
  std::vector<int> v((size_t)m->length, 0);
  for (int i = 0; i < m->length; ++i)
    for (const std::string & s : m->seq) if (amb[(unsigned char)s[i]]) { v[i] = 1; break; }
  return v;
}

extern "C" int bpa_msa_count_ambiguous_sites(const bpa_msa_t * m, int dtype)
{
  if (dtype == BPA_DATA_AA) return 0;
  const std::vector<int> v = ambiguous_sites(m);
  return (int)std::count(v.begin(), v.end(), 1);
}

extern "C" int bpa_msa_remove_ambiguous(bpa_msa_t * m)
{
  std::vector<int> amb = ambiguous_sites(m);
  const int namb = (int)std::count(amb.begin(), amb.end(), 1);
  if (namb == m->length) return 0;
  // every ambiguous site in the kept prefix trades places with the right-most clean site of the tail
  int i = 0, j = m->length - 1;
  for (;;)
  {
    while (i < m->length && !amb[i]) ++i;
    while (j >= 0 && amb[j]) --j;
    if (j < i) break;
    for (std::string & s : m->seq) std::swap(s[i], s[j]);
    std::swap(amb[i], amb[j]);
    ++i; --j;
  }
  m->length -= namb;
  for (std::string & s : m->seq) s.resize((size_t)m->length);
  return 1;
}

extern "C" int bpa_msa_compress(bpa_msa_t * m, int dtype, int jc69, unsigned * weights)
{
  if (!m || !weights || m->length <= 0) return fail("bpa_msa_compress: bad arguments");
  std::vector<char *> rows;
  for (std::string & s : m->seq) rows.push_back(&s[0]);
  int len = m->length;
  const int np = bpa_compress_site_patterns(rows.data(), dtype == BPA_DATA_DNA ? bpa_map_nt() : bpa_map_aa(),
                                            (int)rows.size(), &len, jc69, weights);
  if (!np) return fail("bpa_msa_compress: a character of the alignment is not in the state map");
  m->length = np;
  for (std::string & s : m->seq) s.resize((size_t)np);
  return np;
}

// ------------------------------------------------------------------------ Imap --
extern "C" bpa_imap_t * bpa_imap_read(const char * path)
{
  FILE * fp = path ? fopen(path, "r") : nullptr;
  if (!fp) { fail(std::string("Unable to open file (") + (path ? path : "") + ")"); return nullptr; }
  LineReader rd(fp);
  bpa_imap * im = new bpa_imap;
  std::string line;
  auto comment_or_end = [](const std::string & s, size_t p)
  {
    p = s.find_first_not_of(" \t\r\n", p);
    return p == std::string::npos || s[p] == '*' || s[p] == '#';
  };
  while (rd.next(line))
  {
    if (comment_or_end(line, 0)) continue;
    std::string tok[2];
    size_t p = 0;
    bool ok = true;
    for (int k = 0; k < 2 && ok; ++k)
    {
      p = line.find_first_not_of(" \t\r\n", p);
      if (p == std::string::npos || line[p] == '*' || line[p] == '#') { ok = false; break; }
      size_t e = line.find_first_of(" \t\r\n", p);
      if (e == std::string::npos) e = line.size();
      tok[k] = line.substr(p, e - p);
      p = e;
    }
    if (!ok || !comment_or_end(line, p))
    {
      fail(std::string("Invalid entry in ") + path + " (line " + std::to_string(rd.lineno) + ")");
      delete im; fclose(fp); return nullptr;
    }
    im->individual.push_back(tok[0]);
    im->species.push_back(tok[1]);
  }
  fclose(fp);
  return im;
}
extern "C" void bpa_imap_destroy(bpa_imap_t * im) { delete im; }
extern "C" long bpa_imap_count(const bpa_imap_t * im) { return (long)im->individual.size(); }
extern "C" const char * bpa_imap_individual(const bpa_imap_t * im, long i)
{ return i >= 0 && i < (long)im->individual.size() ? im->individual[i].c_str() : nullptr; }
extern "C" const char * bpa_imap_species(const bpa_imap_t * im, long i)
{ return i >= 0 && i < (long)im->species.size() ? im->species[i].c_str() : nullptr; }

extern "C" int bpa_imap_lookup(const bpa_imap_t * im, const char * label, const char * const * species, int nspecies)
{
  const char * tag = label ? strchr(label, '^') : nullptr;
  if (!tag) { fail(std::string("Cannot find species tag on sequence ") + (label ? label : "")); return -1; }
  ++tag;
  if (!*tag) { fail(std::string("Sequence ") + label + " contains no label"); return -1; }
  for (size_t i = 0; i < im->individual.size(); ++i)
    if (im->individual[i] == tag)
    {
      for (int s = 0; s < nspecies; ++s) if (im->species[i] == species[s]) return s;
      fail("Cannot find node with population label " + im->species[i]);
      return -1;
    }
  fail(std::string("Cannot find species mapping for sequence ") + tag);
  return -1;
}

// ------------------------------------------------------------- diploid phasing --
namespace {
const long * g_sort_key;
// sites with more heterozygotes first; ties are left to the C library's qsort exactly as the
// reference leaves them (diploid.c:29-36, 434)
int cmp_more_hets_first(const void * a, const void * b)
{
  const long x = g_sort_key[*(const long *)a], y = g_sort_key[*(const long *)b];
  return (x < y) - (x > y);
}
int popcount(unsigned x) { return __builtin_popcount(x); }
}

extern "C" long bpa_msa_diploid_resolve(bpa_msa_t * m, const unsigned * diploid, const unsigned * weights,
                                        unsigned long * resolution_count)
{
  if (!m || !diploid || !weights || !resolution_count) return fail("bpa_msa_diploid_resolve: null argument");
  const unsigned * map = bpa_map_nt();
  const long count = (long)m->seq.size(), len = m->length;
  unsigned char letter[256] = {0};                         // state code -> character (last one wins)
  for (int c = 0; c < 256; ++c) if (map[c]) letter[map[c] & 255u] = (unsigned char)c;

  // h[i][j]: 1 = heterozygote still to enumerate, -1 = heterozygote with its phase pinned, 0 otherwise
  std::vector<int> h((size_t)count*len, 0);
  std::vector<long> resolved((size_t)count, 1), singletons((size_t)count, 0), sitehets((size_t)len, 0);
  long unresolved = 0;
  for (long i = 0; i < count; ++i)
  {
    if (!diploid[i]) continue;
    for (long j = 0; j < len; ++j)
    {
      const unsigned code = map[(unsigned char)m->seq[i][j]];
      if (!code) return fail("bpa_msa_diploid_resolve: a character of the alignment is not in the state map");
      if (popcount(code) == 2)
      {
        h[(size_t)i*len + j] = 1;
        sitehets[j]++;
        if (resolved[i]) { /* first heterozygote of this sequence */ }
        resolved[i] = 0;
        unresolved++;
        if (weights[j] == 1) singletons[i]++;
      }
    }
  }
  // singleton patterns (weight 1) holding at least one heterozygote
  std::vector<long> single;
  for (long j = 0; j < len; ++j) if (weights[j] == 1 && sitehets[j]) single.push_back(j);

  // one sequence per round gets the phase of one heterozygote pinned: at the singleton site with most
  // heterozygotes (first in sorted order that has an unresolved heterozygote), the unresolved sequence
  // with the fewest singleton heterozygotes
  for (long round = 0; round < count && unresolved; ++round)
  {
    g_sort_key = sitehets.data();
    qsort(single.data(), single.size(), sizeof(long), cmp_more_hets_first);
    long chosen = -1;
    for (size_t s = 0; s < single.size(); ++s)
    {
      const long site = single[s];
      long best = len + 1;
      for (long j = 0; j < count; ++j)
      {
        if (resolved[j] || h[(size_t)j*len + site] == 0) continue;
        if (singletons[j] < best) { best = singletons[j]; chosen = j; }
      }
      if (chosen >= 0)
      {
        h[(size_t)chosen*len + site] = -1;
        sitehets[site]--;
        resolved[chosen] = 1;
        unresolved--;
        if (sitehets[site] == 0) single.erase(single.begin() + (long)s);
        break;
      }
    }
    if (chosen == -1) break;
  }

  size_t patterns = 0;
  for (long j = 0; j < len; ++j)
  {
    if (sitehets[j] >= 40) return fail("bpa_msa_diploid_resolve: too many heterozygotes at one site");
    resolution_count[j] = 1ul << sitehets[j];
    patterns += resolution_count[j];
  }
  std::vector<long> slot((size_t)count);
  long nseq = 0;
  for (long i = 0; i < count; ++i) { slot[i] = nseq; nseq += diploid[i] ? 2 : 1; }
  std::vector<std::string> seq((size_t)nseq, std::string(patterns, '\0')), label((size_t)nseq);
  for (long i = 0; i < count; ++i)
  {
    if (diploid[i]) { label[slot[i]] = m->label[i] + ".1"; label[slot[i] + 1] = m->label[i] + ".2"; }
    else label[slot[i]] = m->label[i];
  }
  std::vector<char> site((size_t)nseq);
  std::vector<long> hets;
  size_t q = 0;
  for (long j = 0; j < len; ++j)
  {
    hets.clear();
    for (long i = 0; i < count; ++i)
    {
      const int hv = h[(size_t)i*len + j];
      const long k = slot[i];
      if (hv == 0)
      {
        site[k] = m->seq[i][j];
        if (diploid[i]) site[k + 1] = m->seq[i][j];
      }
      else if (hv == -1)
      {
        const unsigned code = map[(unsigned char)m->seq[i][j]];
        const unsigned lo = code & (~code + 1u);
        site[k] = (char)letter[lo];
        site[k + 1] = (char)letter[code & ~lo];
      }
      else hets.push_back(i);
    }
    const long n = (long)hets.size();
    for (unsigned long r = 0; r < (1ul << n); ++r)
    {
      // bit k of r orders the alleles of the k-th heterozygote counted from the LAST sequence
      unsigned long bits = r;
      for (long k = 0; k < n; ++k, bits >>= 1)
      {
        const long i1 = hets[n - 1 - k];
        const unsigned code = map[(unsigned char)m->seq[i1][j]];
        unsigned a = code & (~code + 1u), b = code & ~a;
        if (bits & 1) std::swap(a, b);
        site[slot[i1]] = (char)letter[a];
        site[slot[i1] + 1] = (char)letter[b];
      }
      for (long k = 0; k < nseq; ++k) seq[k][q] = site[k];
      ++q;
    }
  }
  m->seq.swap(seq);
  m->label.swap(label);
  m->length = (int)patterns;
  return (long)patterns;
}

extern "C" int bpa_msa_compress_diploid(bpa_msa_t * m, int jc69, unsigned * weights, unsigned long * mapping)
{
  if (!m || !weights || !mapping || m->length <= 0) return fail("bpa_msa_compress_diploid: bad arguments");
  // compress a copy with every column tagged by its index, then read the mapping off the class keys:
  // the A2 column i maps to the A3 pattern holding its class
  const int count = (int)m->seq.size(), len = m->length;
  const unsigned * map = bpa_map_nt();
  std::vector<uint32_t> keys((size_t)len*count);
  for (int i = 0; i < len; ++i)
  {
    uint32_t * key = &keys[(size_t)i*count];
    bool simple = jc69 != 0;
    for (int j = 0; j < count; ++j)
    {
      key[j] = map[(unsigned char)m->seq[j][i]];
      if (!key[j]) return fail("bpa_msa_compress_diploid: a character of the alignment is not in the state map");
      if (!(key[j] == 1 || key[j] == 2 || key[j] == 4 || key[j] == 8 || key[j] == 15)) simple = false;
    }
    if (simple)
    {
      uint32_t rename[16] = {0}; rename[15] = 15;
      uint32_t next = 1;
      for (int j = 0; j < count; ++j) { if (!rename[key[j]]) rename[key[j]] = next++; key[j] = rename[key[j]]; }
    }
  }
  std::vector<int> order(len);
  std::iota(order.begin(), order.end(), 0);
  std::sort(order.begin(), order.end(), [&](int a, int b)
  {
    const uint32_t * x = &keys[(size_t)a*count], * y = &keys[(size_t)b*count];
    for (int j = 0; j < count; ++j) if (x[j] != y[j]) return x[j] < y[j];
    return a < b;
  });
  std::vector<std::string> out((size_t)count);
  int np = 0;
  for (int r = 0; r < len; ++r)
  {
    const int i = order[r];
    const bool same = r && std::equal(&keys[(size_t)i*count], &keys[(size_t)i*count] + count, &keys[(size_t)order[r-1]*count]);
    if (same) weights[np-1]++;
    else
    {
      for (int j = 0; j < count; ++j) out[j].push_back(m->seq[j][i]);
      weights[np++] = 1;
    }
    mapping[i] = (unsigned long)(np - 1);
  }
  m->seq.swap(out);
  m->length = np;
  return np;
}

// ---------------------------------------------------------------------- writer --
extern "C" int bpa_msa_write_phylip(const char * path, bpa_msa_t * const * list, long count,
                                    const unsigned * const * weights, const int * dtypes)
{
  if (!path || !list || !weights || count <= 0) return fail("bpa_msa_write_phylip: bad arguments");
  FILE * fp = fopen(path, "w");
  if (!fp) return fail(std::string("Unable to open file (") + path + ")");
  size_t pad = 0;
  for (long i = 0; i < count; ++i) for (const std::string & l : list[i]->label) pad = std::max(pad, l.size());
  pad += 4;
  const unsigned char * prn = tables().nt_print;
  for (long i = 0; i < count; ++i)
  {
    const bpa_msa * m = list[i];
    const bool dna = !dtypes || dtypes[i] == BPA_DATA_DNA;
    fprintf(fp, "%d %d P\n", (int)m->seq.size(), m->length);
    for (size_t s = 0; s < m->seq.size(); ++s)
    {
      fprintf(fp, "%-*s", (int)pad, m->label[s].c_str());
      for (int j = 0; j < m->length; ++j)
      {
        if (j % 10 == 0) fputc(' ', fp);
        fputc(dna ? prn[(unsigned char)m->seq[s][j]] : m->seq[s][j], fp);
      }
      fputc('\n', fp);
    }
    for (int j = 0; j < m->length; ++j) fprintf(fp, j ? " %u" : "%u", weights[i][j]);
    fprintf(fp, "\n\n");
  }
  fclose(fp);
  return 1;
}
