/* bpp_amd_input.h — the input side of the likelihood path: what
 * BPP's init() does between "open the sequence file" and "locus_create + set tip states"
 * (method.c:3299-3470, 3573-3672): multi-locus sequential-PHYLIP reader, Imap reader,
 * removal of all-missing sequences / ambiguous sites, site-pattern compression,
 * analytic phasing of unphased diploid sequences with the second (mapping) compression,
 * and the writer of <jobname>.compressed-aln.phy.
 *
 * Host-only code exported by libbpp_amd.so (bpp_amd/csrc/host_input.cpp); plain C ABI.
 * Functions return 1 / a handle on success and 0 / NULL on failure with the message in
 * bpa_last_error() (bpp_amd.h) — where the reference calls fatal() (util.c) the caller
 * of this library decides.
 */
#ifndef BPP_AMD_INPUT_H
#define BPP_AMD_INPUT_H

#ifdef __cplusplus
extern "C" {
#endif

/* one alignment: msa_t (bpp.h:821-838) */
typedef struct bpa_msa bpa_msa_t;
/* the Imap list: list_t of mapping_t (bpp.h:954-959) */
typedef struct bpa_imap bpa_imap_t;

/* character classes of the sequence reader, pll_map_fasta (maps.c:173-198):
   0 stripped, 1 legal, 2 fatal, 3 silently stripped */
const unsigned * bpa_map_fasta(void);
/* pll_map_amb (maps.c:66-84), pll_map_nt_missing (maps.c:86-104), pll_map_aa_missing (maps.c:106-124) */
const unsigned * bpa_map_amb(void);
const unsigned * bpa_map_nt_missing(void);
const unsigned * bpa_map_aa_missing(void);

/* phylip_open + phylip_parse_multisequential (phylip.c:270-318, 622-681): read the
   alignments of a multi-locus sequential PHYLIP file; max_loci > 0 stops after that many
   (the 'nloci' option, phylip.c:650).  *out is malloc'ed (free with bpa_msa_list_free).  */
int  bpa_phylip_read(const char * path, long max_loci, bpa_msa_t *** out, long * count);
void bpa_msa_list_free(bpa_msa_t ** list, long count);

bpa_msa_t * bpa_msa_create(int count, int length, const char * const * labels,
                           const char * const * sequences);
void bpa_msa_destroy(bpa_msa_t *);                                   /* msa_destroy, msa.c:309 */
int  bpa_msa_count(const bpa_msa_t *);
int  bpa_msa_length(const bpa_msa_t *);
const char * bpa_msa_label(const bpa_msa_t *, int i);
const char * bpa_msa_sequence(const bpa_msa_t *, int i);

/* msa_remove_missing_sequences (msa.c:245-307): drops sequences made of missing data only;
   returns the number dropped, -1 when nothing is left.  dtype: BPA_DATA_DNA / BPA_DATA_AA  */
int  bpa_msa_remove_missing_sequences(bpa_msa_t *, int dtype);
/* msa_count_ambiguous_sites (msa.c:137-156); 0 for amino-acid data                         */
int  bpa_msa_count_ambiguous_sites(const bpa_msa_t *, int dtype);
/* msa_remove_ambiguous (msa.c:229-243), the 'cleandata = 1' path: the ambiguous sites are
   swapped to the right end and cut off (so the order of the kept sites is the reference's);
   0 when every site is ambiguous                                                            */
int  bpa_msa_remove_ambiguous(bpa_msa_t *);
/* compress_site_patterns on the alignment (method.c:3425-3459): in place; weights must hold
   bpa_msa_length() entries; returns the number of patterns (0 on failure)                   */
int  bpa_msa_compress(bpa_msa_t *, int dtype, int jc69, unsigned * weights);

/* parse_mapfile (parsemap.c:227-276): "individual species" per line, '*' and '#' comments   */
bpa_imap_t * bpa_imap_read(const char * path);
void bpa_imap_destroy(bpa_imap_t *);
long bpa_imap_count(const bpa_imap_t *);
const char * bpa_imap_individual(const bpa_imap_t *, long i);
const char * bpa_imap_species(const bpa_imap_t *, long i);
/* species of a sequence label "name^individual" (the lookup of diploid.c:66-90 /
   gtree.c population assignment): index into species[] or -1 with the error set             */
int  bpa_imap_lookup(const bpa_imap_t *, const char * label, const char * const * species,
                     int nspecies);

/* diploid_resolve_locus (diploid.c:307-647): expands the compressed alignment A1 (with its
   pattern weights) into the alignment A2 of all phase resolutions; diploid[i] != 0 marks
   sequence i as an unphased diploid.  resolution_count must hold the A1 length.
   Labels become "x.1"/"x.2".  Returns the A2 length (0 on failure).                         */
long bpa_msa_diploid_resolve(bpa_msa_t *, const unsigned * diploid, const unsigned * weights,
                             unsigned long * resolution_count);
/* compress_site_patterns_diploid (compress.c:378-547): compresses A2 to A3 in place and
   returns the A2 -> A3 pattern mapping (mapping must hold the A2 length) and A3's weights   */
int  bpa_msa_compress_diploid(bpa_msa_t *, int jc69, unsigned * weights, unsigned long * mapping);

/* msa_print_phylip (msa.c:109-135): the format of <jobname>.compressed-aln.phy            */
int  bpa_msa_write_phylip(const char * path, bpa_msa_t * const * list, long count,
                          const unsigned * const * weights, const int * dtypes);

#ifdef __cplusplus
}
#endif
#endif
