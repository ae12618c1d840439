/*
 * oracle/oracle.h — TEST INFRASTRUCTURE ONLY (checker, never product).
 *
 * Plain-C restatement of the reference's (bpp v4.8.7) per-locus likelihood hot
 * path, written from the arithmetic described in SURVEY.md App. A.  Every
 * function cites the reference file:line it follows.  Pinned by
 * tests/test_oracle_pin.py against (a) the real reference compiled in place
 * (oracle/_ref/libbppref.so) and (b) golden vectors committed under
 * tests/golden/ that were generated from that reference (tests/golden/make_golden.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 */
#ifndef BPP_ORACLE_H
#define BPP_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* summation orders of the reference's back-ends */
#define ORC_ORDER_SEQ   0   /* scalar: left-to-right (core_partials.c:713-717)            */
#define ORC_ORDER_PAIR  1   /* 4-state AVX: (p0+p1)+(p2+p3), no FMA (core_partials_avx.c:461-487) */
#define ORC_ORDER_FMA4  2   /* generic AVX2: 4 FMA lane accumulators (core_partials_avx2.c:666-745) */

#define ORC_SCALE_FACTOR    0x1p+256
#define ORC_SCALE_THRESHOLD 0x1p-256

const unsigned int * orc_get_map_nt(void);   /* maps.c:26  (pll_map_nt) */
const unsigned int * orc_get_map_aa(void);   /* maps.c:126 (pll_map_aa) */

void   orc_set_tipclv(unsigned states, unsigned sites, unsigned rate_cats,
                      const unsigned * map, const char * seq, double * clv);
int    orc_update_partial_ii(unsigned states, unsigned sites, unsigned rate_cats,
                             double * parent_clv, unsigned * parent_scaler,
                             const double * left_clv, const double * right_clv,
                             const double * left_matrix, const double * right_matrix,
                             const unsigned * left_scaler, const unsigned * right_scaler,
                             int order);
double orc_root_loglikelihood(unsigned states, unsigned sites, unsigned rate_cats,
                              const double * clv, const unsigned * scaler,
                              const double * freqs, const double * rate_weights,
                              const unsigned * pattern_weights, double * persite_lnl,
                              int order);
void   orc_root_likelihood_vector(unsigned states, unsigned sites, unsigned rate_cats,
                                  const double * clv, const double * freqs,
                                  const double * rate_weights, double * persite_lh,
                                  int order);
double orc_diploid_loglikelihood(const double * persite_lh, int unphased_length,
                                 const unsigned long * resolution_count,
                                 const unsigned long * mapping,
                                 const unsigned * unphased_weights);
void   orc_pmatrix_jc69(unsigned rate_cats, const double * rates, double t, double * pmat);
void   orc_pmatrix_dna(unsigned model, const double * f, const double * q, unsigned rate_cats,
                       const double * rates, double t, double * pmat);
void   orc_update_eigen(unsigned states, const double * freqs, const double * subst_params,
                        double * eigenvecs, double * inv_eigenvecs, double * eigenvals);
void   orc_pmatrix_eigen(unsigned states, unsigned rate_cats, const double * rates, double t,
                         const double * eigenvals, const double * eigenvecs,
                         const double * inv_eigenvecs, double * pmat, int library_form);
int    orc_gamma_cats(double alpha, double beta, unsigned categories, double * rates);
int    orc_compress(char ** seqs, int count, int * length, const unsigned * map,
                    int jc69, unsigned * weights);

#ifdef __cplusplus
}
#endif
#endif
