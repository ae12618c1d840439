// p2p.hpp — one-shot sum all-reduce of a few hundred doubles between the GPUs of one node.
//
// The only exchange of a sharded run is the sum an all-loci proposal is decided on: 8 bytes to ~2 KB, four to five
// times per MCMC iteration, on the critical path of every step that follows.  A ring collective pays 2(N-1) hops of
// latency for it; here every rank stores its values straight into a mailbox of EVERY rank through xGMI peer mappings
// (hipIpc handles of uncached / fine-grained device memory), raises a sequence flag with system-scope release, waits for the N
// flags in its own mailbox and adds the N vectors up in rank order — one hop, one kernel, the same bits on every rank.
// Mailboxes are double-buffered by sequence parity: a rank can only be one exchange ahead of the slowest.  Waits are
// bounded: a flag that does not arrive sets an error the host reads back (bench.py then repeats the run over RCCL,
// which is also what its start-up self-test compares this path with).
//
// Included at the end of engine.hip.
#pragma once

namespace p2p {
constexpr int MAXW = 16;
constexpr unsigned HDR = 16;                         // bytes before a slot's values: the sequence flag (+ padding)

__global__ void __launch_bounds__(512) allreduce_kernel(double * data, unsigned n, unsigned char * const * peers,
                                                        unsigned char * mine, int rank, int world, unsigned long long seq,
                                                        size_t slot_bytes, int * err, unsigned long long spin_limit)
{
  // after a time-out the exchange is void on this rank: later calls return at once (the peers time out once, too),
  // so a broken link costs seconds, not seconds per step
  if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
  const unsigned tid = threadIdx.x, par = (unsigned)(seq & 1ull);
  const size_t my_slot = ((size_t)par*world + rank)*slot_bytes;          // this rank's slot inside every mailbox
  if (tid < n)
  {
    const double v = data[tid];
    for (int p = 0; p < world; ++p)
      __hip_atomic_store(reinterpret_cast<double *>(peers[p] + my_slot + HDR) + tid, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __threadfence_system();
  __syncthreads();
  if (tid < (unsigned)world)
  {
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(peers[tid] + my_slot), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned long long * f = reinterpret_cast<const unsigned long long *>(mine + ((size_t)par*world + tid)*slot_bytes);
    const unsigned long long t0 = wall_clock64();             // constant-rate counter (100 MHz)
    while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != seq)
    {
      if (wall_clock64() - t0 > spin_limit) { *err = 1; break; }
      __builtin_amdgcn_s_sleep(2);
    }
  }
  __syncthreads();
  if (tid < n)
  {
    double acc = 0;
    for (int r = 0; r < world; ++r)
      acc += __hip_atomic_load(reinterpret_cast<const double *>(mine + ((size_t)par*world + r)*slot_bytes + HDR) + tid,
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    data[tid] = acc;
  }
}
}  // namespace p2p

struct bpa_p2p
{
  bpa_engine * eng = nullptr;
  int rank = 0, world = 1;
  unsigned nmax = 0;
  size_t slot_bytes = 0;
  unsigned char * mail = nullptr;                  // this rank's mailbox: [2][world] slots, fine-grained device memory
  unsigned char * peer[p2p::MAXW] = {};            // every rank's mailbox as mapped here (peer[rank] == mail)
  DevBuf<unsigned char *> d_peer;
  DevBuf<int> d_err;
  unsigned long long seq = 0;
  unsigned long long spin_limit = 300000000ull;    // bound of a wait in ticks of the 100 MHz counter: 3 s (bpa_p2p_set_timeout)
  bool connected = false;
};

extern "C" bpa_p2p_t * bpa_p2p_create(bpa_engine_t * e, int rank, int world, unsigned max_doubles, void * handle_out)
{
  if (!e || !handle_out || world < 1 || world > p2p::MAXW || rank < 0 || rank >= world || !max_doubles || max_doubles > 512)
  { fail("bpa_p2p_create: 1..16 ranks, 1..512 doubles"); return nullptr; }
  std::lock_guard<std::recursive_mutex> lock_(e->mtx);
  if (!set_device(e)) return nullptr;
  bpa_p2p * p = new bpa_p2p();
  p->eng = e; p->rank = rank; p->world = world; p->nmax = max_doubles;
  p->slot_bytes = (p2p::HDR + (size_t)max_doubles*sizeof(double) + 127) & ~(size_t)127;
  const size_t bytes = 2*(size_t)world*p->slot_bytes;
  int zero = 0;
  hipIpcMemHandle_t h;
  // uncached device memory (never held in an L2: what a kernel polls must be what a peer wrote over xGMI), else fine-grained
  if (hipExtMallocWithFlags((void **)&p->mail, bytes, hipDeviceMallocUncached) != hipSuccess)
  {
    (void)hipGetLastError();
    p->mail = nullptr;
    if (hipExtMallocWithFlags((void **)&p->mail, bytes, hipDeviceMallocFinegrained) != hipSuccess) p->mail = nullptr;
  }
  if (!p->mail || hipMemset(p->mail, 0, bytes) != hipSuccess ||
      hipDeviceSynchronize() != hipSuccess || hipIpcGetMemHandle(&h, p->mail) != hipSuccess || !upload(p->d_err, &zero, 1))
  {
    fail(std::string("bpa_p2p_create: ") + hipGetErrorString(hipGetLastError()));
    if (p->mail) (void)hipFree(p->mail);
    delete p; return nullptr;
  }
  static_assert(sizeof(hipIpcMemHandle_t) == BPA_P2P_HANDLE_BYTES, "handle size");
  std::memcpy(handle_out, &h, sizeof(h));
  p->peer[rank] = p->mail;
  return p;
}

extern "C" int bpa_p2p_connect(bpa_p2p_t * p, const void * handles)
{
  std::lock_guard<std::recursive_mutex> lock_(p->eng->mtx);
  if (!set_device(p->eng)) return 0;
  for (int r = 0; r < p->world; ++r)
  {
    if (r == p->rank) continue;
    hipIpcMemHandle_t h;
    std::memcpy(&h, (const unsigned char *)handles + (size_t)r*sizeof(h), sizeof(h));
    if (hipIpcOpenMemHandle((void **)&p->peer[r], h, hipIpcMemLazyEnablePeerAccess) != hipSuccess)
      return fail(std::string("bpa_p2p_connect: cannot map the mailbox of rank ") + std::to_string(r) + ": " + hipGetErrorString(hipGetLastError()));
  }
  if (!upload(p->d_peer, p->peer, (size_t)p->world)) return 0;
  p->connected = true;
  return 1;
}

extern "C" int bpa_p2p_allreduce(bpa_p2p_t * p, double * device_values, unsigned n)
{
  std::lock_guard<std::recursive_mutex> lock_(p->eng->mtx);
  if (!p->connected) return fail("bpa_p2p_allreduce: call bpa_p2p_connect first");
  if (!n || n > p->nmax) return fail("bpa_p2p_allreduce: count out of range");
  if (!set_device(p->eng)) return 0;
  p->seq++;
  hipLaunchKernelGGL(p2p::allreduce_kernel, dim3(1), dim3(512), 0, p->eng->stream, device_values, n, p->d_peer.p, p->mail, p->rank, p->world,
                     p->seq, p->slot_bytes, p->d_err.p, p->spin_limit);
  HIPCHK(hipGetLastError());
  return 1;
}

extern "C" void bpa_p2p_set_timeout(bpa_p2p_t * p, unsigned milliseconds)
{
  std::lock_guard<std::recursive_mutex> lock_(p->eng->mtx);
  p->spin_limit = (unsigned long long)(milliseconds ? milliseconds : 1u)*100000ull;
}

extern "C" int bpa_plans_launch_exchange(bpa_plan_t * const * plans, unsigned count, bpa_p2p_t * p, double * device_values, unsigned n)
{
  return bpa_plans_launch(plans, count) && bpa_p2p_allreduce(p, device_values, n);
}

extern "C" int bpa_p2p_status(bpa_p2p_t * p)
{
  std::lock_guard<std::recursive_mutex> lock_(p->eng->mtx);
  if (!set_device(p->eng)) return -1;
  int err = 0;
  if (hipStreamSynchronize(p->eng->stream) != hipSuccess || hipMemcpy(&err, p->d_err.p, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return err;
}

extern "C" void bpa_p2p_destroy(bpa_p2p_t * p)
{
  if (!p) return;
  std::lock_guard<std::recursive_mutex> lock_(p->eng->mtx);
  (void)hipSetDevice(p->eng->device);
  (void)hipStreamSynchronize(p->eng->stream);
  for (int r = 0; r < p->world; ++r) if (r != p->rank && p->peer[r]) (void)hipIpcCloseMemHandle(p->peer[r]);
  if (p->mail) (void)hipFree(p->mail);
  p->d_peer.free(); p->d_err.free();
  delete p;
}
