/*
 * bpp_amd_rccl.h — the exchange of several GPUs as native code: an RCCL sum all-reduce on the engine's stream, behind
 * the callback type of bpp_amd.h (bpa_allreduce_fn).  SURVEY.md section 8e: loci are sharded over the ranks, the only
 * traffic is the sum an all-loci step (TAU, MIX, THETA) is decided on — the reference's reduction inside its worker
 * loop (threads.c:525-591: td.td[i] summed after every all-loci proposal) is the analogue.  With this callback installed
 * nothing but C runs inside bpa_sampler_iterate: no Python, no framework.
 *
 * One communicator per process (one process per GPU).  Rank 0 makes the id (bpa_rccl_unique_id), every rank receives
 * its 128 bytes by whatever out-of-band means the launcher has (a file, MPI, torch.distributed.broadcast ...) and calls
 * bpa_rccl_create, which is collective.  libbpp_amd_rccl.so links librccl.so.1; it is a library of its own so that
 * libbpp_amd.so does not depend on RCCL.
 */
#ifndef BPP_AMD_RCCL_H
#define BPP_AMD_RCCL_H

#ifdef __cplusplus
extern "C" {
#endif

#define BPA_RCCL_ID_BYTES 128
typedef struct bpa_rccl bpa_rccl_t;

/* rank 0: a fresh communicator id (BPA_RCCL_ID_BYTES bytes); returns 0 on failure */
int          bpa_rccl_unique_id(char * id);
/* collective over the nranks processes; device = this rank's GPU (hipSetDevice) */
bpa_rccl_t * bpa_rccl_create(const char * id, int nranks, int rank, int device);
void         bpa_rccl_destroy(bpa_rccl_t *);
/* THE callback: pass it as fn and the bpa_rccl_t* as ctx to bpa_sampler_set_allreduce (bpp_amd.h).  Enqueues
   ncclAllReduce(device_sums, device_sums, count, ncclDouble, ncclSum, comm, stream); non-zero on success */
int          bpa_rccl_allreduce(void * ctx, double * device_sums, unsigned count, void * stream);
/* the same as a plain call (tests): in place on the device doubles at p, on hipStream_t stream */
int          bpa_rccl_allreduce_sum(bpa_rccl_t *, double * p, unsigned count, void * stream);
unsigned long bpa_rccl_calls(const bpa_rccl_t *);       /* all-reduces enqueued so far */
const char * bpa_rccl_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
