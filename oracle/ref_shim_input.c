/*
 * oracle/ref_shim_input.c — TEST INFRASTRUCTURE ONLY.
 *
 * Flat-array shim over the REAL reference's input side (bpp v4.8.7), compiled against
 * the reference's bpp.h where it lies and linked into oracle/_ref/libbppref.so next to
 * ref_shim.c.  Nothing of the reference is copied; this file only *calls*
 *
 *   phylip_open / phylip_parse_multisequential   phylip.c:270, 622
 *   msa_remove_missing_sequences                 msa.c:245
 *   msa_count_ambiguous_sites / msa_remove_ambiguous   msa.c:137, 229
 *   compress_site_patterns                       compress.c:218
 *   parse_mapfile                                parsemap.c:227
 *   diploid_resolve                              diploid.c:649
 *   compress_site_patterns_diploid               compress.c:378
 *   msa_print_phylip                             msa.c:109
 *
 * in the order method.c:3299-3672 does.  Used by tests/test_input_pin.py and by
 * tests/golden/make_golden_input.py.  Never imported by the product.
 */
#include "bpp.h"

const unsigned int * ref_map_fasta(void)      { return pll_map_fasta; }
const unsigned int * ref_map_amb(void)        { return pll_map_amb; }
const unsigned int * ref_map_nt_missing(void) { return pll_map_nt_missing; }
const unsigned int * ref_map_aa_missing(void) { return pll_map_aa_missing; }

msa_t ** ref_phylip_read(const char * path, long max_loci, long * count)
{
  msa_t ** list;
  phylip_t * fd;
  opt_locus_count = max_loci;
  fd = phylip_open(path, pll_map_fasta);
  if (!fd) return NULL;
  list = phylip_parse_multisequential(fd, count);
  phylip_close(fd);
  opt_locus_count = 0;
  return list;
}

int          ref_msa_count(msa_t ** l, long k)           { return l[k]->count; }
int          ref_msa_length(msa_t ** l, long k)          { return l[k]->length; }
const char * ref_msa_label(msa_t ** l, long k, int i)    { return l[k]->label[i]; }
const char * ref_msa_sequence(msa_t ** l, long k, int i) { return l[k]->sequence[i]; }

void ref_msa_set_type(msa_t ** l, long k, int dtype, int model)
{
  l[k]->dtype = dtype;
  l[k]->model = model;
}

int ref_msa_remove_missing(msa_t ** l, long k)  { return msa_remove_missing_sequences(l[k]); }
int ref_msa_remove_ambiguous(msa_t ** l, long k) { return msa_remove_ambiguous(l[k]); }
int ref_msa_count_ambiguous(msa_t ** l, long k)
{
  msa_count_ambiguous_sites(l[k], pll_map_amb);
  return l[k]->amb_sites_count;
}

/* method.c:3425-3459; the weights stay with the shim (weights[k]) for the later steps */
static unsigned int ** g_weights;
static long g_weights_n;

int ref_msa_compress(msa_t ** l, long nloci, long k, int jc69, unsigned int * w_out)
{
  int i;
  if (!g_weights || g_weights_n != nloci)
  {
    g_weights = (unsigned int **)calloc((size_t)nloci, sizeof(unsigned int *));
    g_weights_n = nloci;
  }
  l[k]->original_length = l[k]->length;
  g_weights[k] = compress_site_patterns(l[k]->sequence,
                                        l[k]->dtype == BPP_DATA_DNA ? pll_map_nt : pll_map_aa,
                                        l[k]->count, &(l[k]->length),
                                        jc69 ? COMPRESS_JC69 : COMPRESS_GENERAL);
  if (!g_weights[k]) return 0;
  for (i = 0; i < l[k]->length; ++i) w_out[i] = g_weights[k][i];
  return l[k]->length;
}

list_t * ref_imap_read(const char * path) { return parse_mapfile(path); }
long ref_imap_count(list_t * m) { return m->count; }
static mapping_t * imap_at(list_t * m, long i)
{
  list_item_t * li = m->head;
  while (i-- && li) li = li->next;
  return li ? (mapping_t *)(li->data) : NULL;
}
const char * ref_imap_individual(list_t * m, long i) { return imap_at(m, i)->individual; }
const char * ref_imap_species(list_t * m, long i)    { return imap_at(m, i)->species; }

/* diploid_resolve (diploid.c:649) on all loci: a species tree of tips only (label + diploid flag is
   all it reads).  resolution counts are kept by the shim; read them with ref_resolution_count.     */
static unsigned long ** g_rescount;

int ref_diploid_resolve(msa_t ** l, long nloci, list_t * maplist, int nspecies,
                        const char ** species, const unsigned int * phase)
{
  int i;
  stree_t * st = (stree_t *)calloc(1, sizeof(stree_t));
  st->tip_count = (unsigned int)nspecies;
  st->nodes = (snode_t **)calloc((size_t)nspecies, sizeof(snode_t *));
  for (i = 0; i < nspecies; ++i)
  {
    st->nodes[i] = (snode_t *)calloc(1, sizeof(snode_t));
    st->nodes[i]->label = xstrdup(species[i]);
    st->nodes[i]->diploid = phase[i];
    st->nodes[i]->node_index = (unsigned int)i;
  }
  opt_datefile = NULL;
  g_rescount = diploid_resolve(st, l, maplist, NULL, g_weights, (int)nloci);
  return g_rescount != NULL;
}

void ref_resolution_count(long k, long n, unsigned long * out)
{
  long i;
  for (i = 0; i < n; ++i) out[i] = g_rescount[k][i];
}

/* method.c:3657: A2 -> A3 */
int ref_msa_compress_diploid(msa_t ** l, long k, int jc69, unsigned int * w_out, unsigned long * mapping_out)
{
  int i, n2 = l[k]->length;
  unsigned int * w = NULL;
  unsigned long * mp = compress_site_patterns_diploid(l[k]->sequence, pll_map_nt, l[k]->count,
                                                      &(l[k]->length), &w,
                                                      jc69 ? COMPRESS_JC69 : COMPRESS_GENERAL);
  if (!mp) return 0;
  for (i = 0; i < n2; ++i) mapping_out[i] = mp[i];
  for (i = 0; i < l[k]->length; ++i) w_out[i] = w[i];
  free(mp); free(w);
  return l[k]->length;
}

/* msa_print_phylip with the shim-held first-compression weights */
int ref_msa_print_phylip(const char * path, msa_t ** l, long nloci)
{
  FILE * fp = fopen(path, "w");
  if (!fp) return 0;
  msa_print_phylip(fp, l, nloci, g_weights);
  fclose(fp);
  return 1;
}

/* ---------------------------------------------------------------------------
 * MSC density: gtree_logprob (gtree.c:3957) = sum over populations of
 * gtree_update_logprob_contrib (gtree.c:3859) on a hand-built species tree: the
 * populations in stree->nodes order with, for locus 0, their coalescent-event
 * lists, seqin_count and coal_count as gtree.c keeps them.  pop[] gives the
 * population of every gene node (tips: their species).
 * ------------------------------------------------------------------------- */
double ref_msc_logpr(int species, const int * parent, const double * tau, const double * theta,
                     int tips, const int * left, const int * right, const double * time, const int * pop,
                     double * contrib_out)
{
  int np = 2*species - 1, n = 2*tips - 1, p, k;
  double logpr;
  stree_t * st = (stree_t *)calloc(1, sizeof(stree_t));
  gnode_t * gn = (gnode_t *)calloc((size_t)n, sizeof(gnode_t));
  st->tip_count = (unsigned int)species; st->inner_count = (unsigned int)(species - 1); st->hybrid_count = 0;
  st->nodes = (snode_t **)calloc((size_t)np, sizeof(snode_t *));
  for (p = 0; p < np; ++p)
  {
    snode_t * s = st->nodes[p] = (snode_t *)calloc(1, sizeof(snode_t));
    s->tau = tau[p]; s->theta = theta[p]; s->node_index = (unsigned int)p;
    s->coalevent = (dlist_t **)calloc(1, sizeof(dlist_t *)); s->coalevent[0] = dlist_create();
    s->seqin_count = (int *)calloc(1, sizeof(int)); s->coal_count = (int *)calloc(1, sizeof(int));
    s->C2ji = (double *)calloc(1, sizeof(double)); s->old_C2ji = (double *)calloc(1, sizeof(double));
    s->logpr_contrib = (double *)calloc(1, sizeof(double)); s->old_logpr_contrib = (double *)calloc(1, sizeof(double));
  }
  for (p = 0; p < np; ++p)
    if (parent[p] >= 0)
    {
      snode_t * q = st->nodes[parent[p]];
      st->nodes[p]->parent = q;
      if (!q->left) q->left = st->nodes[p]; else q->right = st->nodes[p];
    }
  st->root = st->nodes[np - 1];
  for (k = 0; k < n; ++k)
  {
    gn[k].time = time[k];
    if (left[k] >= 0) { dlist_append(st->nodes[pop[k]]->coalevent[0], (void *)(gn + k)); st->nodes[pop[k]]->coal_count[0]++; }
    else st->nodes[pop[k]]->seqin_count[0]++;
  }
  for (p = species; p < np; ++p)      /* children precede parents in this ordering */
  {
    snode_t * s = st->nodes[p];
    s->seqin_count[0] = (s->left->seqin_count[0] - s->left->coal_count[0]) + (s->right->seqin_count[0] - s->right->coal_count[0]);
  }
  opt_est_theta = 1; opt_msci = 0; opt_migration = 0; opt_datefile = NULL;
  if (!global_sortbuffer_r)
  {
    global_sortbuffer_r = (double **)calloc(1, sizeof(double *));
    global_sortbuffer_r[0] = (double *)calloc(4096, sizeof(double));
  }
  logpr = gtree_logprob(st, 1.0, 0, 0);
  if (contrib_out) for (p = 0; p < np; ++p) contrib_out[p] = st->nodes[p]->logpr_contrib[0];
  for (p = 0; p < np; ++p)
  {
    snode_t * s = st->nodes[p];
    dlist_clear(s->coalevent[0], NULL); dlist_destroy(s->coalevent[0]); free(s->coalevent);
    free(s->seqin_count); free(s->coal_count); free(s->C2ji); free(s->old_C2ji); free(s->logpr_contrib); free(s->old_logpr_contrib);
    free(s);
  }
  free(st->nodes); free(st); free(gn);
  return logpr;
}
