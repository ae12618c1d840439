/* rccl_exchange.c — the sum all-reduce of several GPUs' all-loci steps as native code (include/bpp_amd_rccl.h):
 * ncclAllReduce on the engine's stream, no Python inside bpa_sampler_iterate.  Analogue in the reference: the
 * reduction over the worker threads' partial sums after every all-loci proposal (threads.c:525-591). */
#define __HIP_PLATFORM_AMD__ 1
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include "bpp_amd_rccl.h"

struct bpa_rccl { ncclComm_t comm; int nranks, rank, device; unsigned long calls; };

static __thread char errbuf[256];
const char * bpa_rccl_last_error(void) { return errbuf; }
static int fail(const char * what, const char * detail)
{
  snprintf(errbuf, sizeof errbuf, "%s: %s", what, detail ? detail : "");
  return 0;
}

int bpa_rccl_unique_id(char * id)
{
  ncclUniqueId u;
  ncclResult_t r;
  if (!id) return fail("bpa_rccl_unique_id", "null id");
  if (sizeof u.internal != BPA_RCCL_ID_BYTES) return fail("bpa_rccl_unique_id", "unexpected NCCL_UNIQUE_ID_BYTES");
  r = ncclGetUniqueId(&u);
  if (r != ncclSuccess) return fail("ncclGetUniqueId", ncclGetErrorString(r));
  memcpy(id, u.internal, BPA_RCCL_ID_BYTES);
  return 1;
}

bpa_rccl_t * bpa_rccl_create(const char * id, int nranks, int rank, int device)
{
  ncclUniqueId u;
  ncclResult_t r;
  bpa_rccl_t * x;
  if (!id || nranks < 1 || rank < 0 || rank >= nranks) { fail("bpa_rccl_create", "bad argument"); return NULL; }
  if (hipSetDevice(device) != hipSuccess) { fail("bpa_rccl_create", "hipSetDevice failed"); return NULL; }
  x = (bpa_rccl_t *)calloc(1, sizeof *x);
  if (!x) { fail("bpa_rccl_create", "out of memory"); return NULL; }
  memcpy(u.internal, id, BPA_RCCL_ID_BYTES);
  r = ncclCommInitRank(&x->comm, nranks, u, rank);
  if (r != ncclSuccess) { fail("ncclCommInitRank", ncclGetErrorString(r)); free(x); return NULL; }
  x->nranks = nranks; x->rank = rank; x->device = device;
  return x;
}

void bpa_rccl_destroy(bpa_rccl_t * x)
{
  if (!x) return;
  (void)hipSetDevice(x->device);
  (void)ncclCommDestroy(x->comm);
  free(x);
}

int bpa_rccl_allreduce_sum(bpa_rccl_t * x, double * p, unsigned count, void * stream)
{
  ncclResult_t r;
  if (!x || !p) return fail("bpa_rccl_allreduce", "null argument");
  r = ncclAllReduce(p, p, (size_t)count, ncclDouble, ncclSum, x->comm, (hipStream_t)stream);
  if (r != ncclSuccess) return fail("ncclAllReduce", ncclGetErrorString(r));
  x->calls++;
  return 1;
}

int bpa_rccl_allreduce(void * ctx, double * device_sums, unsigned count, void * stream)
{
  return bpa_rccl_allreduce_sum((bpa_rccl_t *)ctx, device_sums, count, stream);
}

unsigned long bpa_rccl_calls(const bpa_rccl_t * x) { return x ? x->calls : 0; }
