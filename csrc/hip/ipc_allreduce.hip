// All-reduce over peer-mapped memory for decode-sized messages (tensor-parallel serving): a one-shot kernel for small
// messages and a two-shot (reduce-scatter + all-gather) kernel for larger ones.
//
// Every rank owns one exported region
//   [ data slot 0 | data slot 1 | result slot 0 | result slot 1 | flags1[2][8][MAX_BLOCKS] | flags2[2][8][MAX_BLOCKS] | state ]
// and maps the
// regions of its peers (hipIpc handles: xGMI loads between GPUs of a node, plain local memory when two processes share a
// GPU). A call with sequence number s uses slot s & 1. Workgroup b of every rank
//   1. copies chunk b of its input into its own slot,
//   2. publishes s in flags[slot][my rank][b] of EVERY peer (release, system scope),
//   3. waits until all peers have published s in its own flags[slot][*][b] (acquire, bounded spin),
//   4. sums chunk b of all ranks' slots in rank order in fp32 - the same order everywhere, so all ranks get the same bits.
// No workgroup needs another workgroup of its own grid, so there is no grid-wide barrier; double buffering replaces the
// closing barrier: slot s & 1 is rewritten by call s + 2, which starts after this rank's call s + 1 has seen every peer's
// flag s + 1, i.e. after every peer has finished call s (stream order on the peer). The sequence number lives in device
// memory and is advanced by the last workgroup to finish, so a captured graph replays correctly.
// A peer that has not arrived within the time limit (60 s by default) makes the wait give up and raises the region's error
// word instead of hanging the device; the host checks that word with every plan.
//
// Two-shot (ipc_allreduce_two_shot_kernel): after step 3 workgroup b of rank r sums only the r-th of the nranks parts of
// chunk b, writes the rounded sums into the RESULT slot of every rank (16-byte stores over xGMI), publishes s in flags2, waits
// for the peers' flags2 and copies chunk b of its own result slot to the output. Per rank that is (n-1)/n of the message read
// from and written to peers instead of (n-1) times the message read: 4x less link traffic at 8 ranks, at the price of a second
// flag round trip - it wins from a few hundred KB upwards and from four ranks on. Every element is summed by exactly one rank,
// so all ranks still end up with the same bits. The slot-reuse argument carries over: a result slot is rewritten by a peer's
// call s + 2 only after that peer has seen this rank's flags1 of call s + 2, i.e. after this rank has finished call s. Role in the reference: the custom all-reduce of its serving engines (`disable_custom_all_reduce`,
// areal/api/cli_args.py:398-454).
#include "common.hpp"

namespace {
constexpr int AR_MAX_RANKS = 8;
constexpr int AR_MAX_BLOCKS = 128;
constexpr int AR_THREADS = 512;

struct ArPeers {
  const bf16_t* data[AR_MAX_RANKS];  // base of every rank's two data slots (own rank included)
  int* flags[AR_MAX_RANKS];          // base of every rank's flag table
};


// Wait until `*f` has reached `seq` (a peer's release store), at most `timeout_ticks` of the 100 MHz wall clock: a peer
// that never arrives raises the region's error word instead of hanging the device. The caller's result is garbage then;
// the host reads the error word with every plan and fails the engine (ops/ipc_allreduce.py).
DEVINL void wait_flag(const int* f, unsigned seq, unsigned* state, long timeout_ticks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  // once a call has given up, the calls queued behind it (the rest of a decode plan) do not wait again: the peer
  // is gone, and every one of them waiting out the limit would hold the device for hours
  if (__hip_atomic_load(&state[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) timeout_ticks = 0;
  while ((int)(__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - (int)seq) < 0) {
    __builtin_amdgcn_s_sleep(16);
    if ((long)(__builtin_amdgcn_s_memrealtime() - t0) > timeout_ticks) {
      atomicExch(&state[2], 1u);
      break;
    }
  }
}

// state: [0] sequence number of the last completed call, [1] finished-workgroup ticket, [2] error word
__global__ __launch_bounds__(AR_THREADS) void ipc_allreduce_kernel(ArPeers peers, bf16_t* my_data, int* my_flags,
                                                                  unsigned* state, const bf16_t* __restrict__ in,
                                                                  bf16_t* __restrict__ out, long n, long cap, int nranks,
                                                                  int rank, long timeout_ticks) {
  const int tid = threadIdx.x, b = blockIdx.x;
  const unsigned seq = __hip_atomic_load(&state[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
  const int slot = (int)(seq & 1u);
  // chunk of this workgroup, in 8-element (16-byte) vectors
  const long nvec = (n + 7) >> 3;
  const long per = (nvec + gridDim.x - 1) / gridDim.x;
  const long v0 = (long)b * per, v1 = v0 + per < nvec ? v0 + per : nvec;
  bf16_t* mine = my_data + (long)slot * cap;
  for (long v = v0 + tid; v < v1; v += AR_THREADS) {
    const long e = v << 3;
    bf16x8 x;
    if (e + 8 <= n) {
      x = ld8(in + e);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = e + j < n ? in[e + j] : (bf16_t)0.f;
    }
    *reinterpret_cast<bf16x8*>(mine + e) = x;  // cap is a multiple of 8: the padded tail stays inside the slot
  }
  __threadfence_system();
  __syncthreads();
  const int fidx = (slot * AR_MAX_RANKS) * AR_MAX_BLOCKS + b;
  if (tid < nranks)
    __hip_atomic_store(peers.flags[tid] + fidx + rank * AR_MAX_BLOCKS, (int)seq, __ATOMIC_RELEASE,
                       __HIP_MEMORY_SCOPE_SYSTEM);
  if (tid < nranks) {
    wait_flag(my_flags + fidx + tid * AR_MAX_BLOCKS, seq, state, timeout_ticks);
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  for (long v = v0 + tid; v < v1; v += AR_THREADS) {
    const long e = v << 3;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    for (int p = 0; p < nranks; ++p) {
      const bf16x8 x = *reinterpret_cast<const bf16x8*>(peers.data[p] + (long)slot * cap + e);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += bf2f(x[j]);
    }
    if (e + 8 <= n) {
      bf16x8 y;
#pragma unroll
      for (int j = 0; j < 8; ++j) y[j] = f2bf(acc[j]);
      *reinterpret_cast<bf16x8*>(out + e) = y;
    } else {
      for (int j = 0; j < 8 && e + j < n; ++j) out[e + j] = f2bf(acc[j]);
    }
  }
  // the last workgroup to finish advances the sequence number (every workgroup has read it by then)
  __syncthreads();
  if (tid == 0) {
    __threadfence();
    const unsigned t = atomicAdd(&state[1], 1u);
    if (t == gridDim.x - 1) {
      atomicExch(&state[1], 0u);
      __hip_atomic_store(&state[0], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}
// Lanes 0 .. nranks-1 of the workgroup: publish `seq` in every peer's flag table, then wait for every peer's flag in mine.
DEVINL void signal_and_wait(int* const* peer_flags, int* my_flags, int fidx, int rank, int nranks, unsigned seq,
                            unsigned* state, long timeout_ticks) {
  const int tid = threadIdx.x;
  if (tid < nranks)
    __hip_atomic_store(peer_flags[tid] + fidx + rank * AR_MAX_BLOCKS, (int)seq, __ATOMIC_RELEASE,
                       __HIP_MEMORY_SCOPE_SYSTEM);
  if (tid < nranks) {
    wait_flag(my_flags + fidx + tid * AR_MAX_BLOCKS, seq, state, timeout_ticks);
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
}

struct ArPeers2 {
  const bf16_t* data[AR_MAX_RANKS];
  bf16_t* result[AR_MAX_RANKS];
  int* flags1[AR_MAX_RANKS];
  int* flags2[AR_MAX_RANKS];
};

__global__ __launch_bounds__(AR_THREADS) void ipc_allreduce_two_shot_kernel(ArPeers2 peers, unsigned* state,
                                                                           const bf16_t* __restrict__ in,
                                                                           bf16_t* __restrict__ out, long n, long cap,
                                                                           int nranks, int rank, long timeout_ticks) {
  const int tid = threadIdx.x, b = blockIdx.x;
  const unsigned seq = __hip_atomic_load(&state[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
  const int slot = (int)(seq & 1u);
  const long nvec = (n + 7) >> 3;
  const long per = (nvec + gridDim.x - 1) / gridDim.x;
  const long v0 = (long)b * per, v1 = v0 + per < nvec ? v0 + per : nvec;
  bf16_t* mine = const_cast<bf16_t*>(peers.data[rank]) + (long)slot * cap;
  for (long v = v0 + tid; v < v1; v += AR_THREADS) {
    const long e = v << 3;
    bf16x8 x;
    if (e + 8 <= n) {
      x = ld8(in + e);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = e + j < n ? in[e + j] : (bf16_t)0.f;
    }
    *reinterpret_cast<bf16x8*>(mine + e) = x;
  }
  __threadfence_system();
  __syncthreads();
  const int fidx = (slot * AR_MAX_RANKS) * AR_MAX_BLOCKS + b;
  signal_and_wait(peers.flags1, peers.flags1[rank], fidx, rank, nranks, seq, state, timeout_ticks);
  // reduce-scatter: this rank's part of the chunk, summed in rank order, pushed into every rank's result slot
  const long part = (v1 - v0 + nranks - 1) / nranks;
  const long p0 = v0 + rank * part < v1 ? v0 + rank * part : v1;
  const long p1 = p0 + part < v1 ? p0 + part : v1;
  for (long v = p0 + tid; v < p1; v += AR_THREADS) {
    const long e = v << 3;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    for (int p = 0; p < nranks; ++p) {
      const bf16x8 x = *reinterpret_cast<const bf16x8*>(peers.data[p] + (long)slot * cap + e);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += bf2f(x[j]);
    }
    bf16x8 y;
#pragma unroll
    for (int j = 0; j < 8; ++j) y[j] = f2bf(acc[j]);
    for (int p = 0; p < nranks; ++p) *reinterpret_cast<bf16x8*>(peers.result[p] + (long)slot * cap + e) = y;
  }
  __threadfence_system();
  __syncthreads();
  signal_and_wait(peers.flags2, peers.flags2[rank], fidx, rank, nranks, seq, state, timeout_ticks);
  // all-gather: every part of the chunk has landed in this rank's result slot
  const bf16_t* res = peers.result[rank] + (long)slot * cap;
  for (long v = v0 + tid; v < v1; v += AR_THREADS) {
    const long e = v << 3;
    const bf16x8 y = *reinterpret_cast<const bf16x8*>(res + e);
    if (e + 8 <= n) {
      *reinterpret_cast<bf16x8*>(out + e) = y;
    } else {
      for (int j = 0; j < 8 && e + j < n; ++j) out[e + j] = y[j];
    }
  }
  __syncthreads();
  if (tid == 0) {
    __threadfence();
    const unsigned t = atomicAdd(&state[1], 1u);
    if (t == gridDim.x - 1) {
      atomicExch(&state[1], 0u);
      __hip_atomic_store(&state[0], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// byte offsets of the parts of a region with capacity `cap` elements
struct ArLayout {
  long result, flags1, flags2, state, total;
  explicit ArLayout(long cap) {
    const long slots = 2 * cap * (long)sizeof(bf16_t), flags = 2L * AR_MAX_RANKS * AR_MAX_BLOCKS * (long)sizeof(int);
    result = slots;
    flags1 = 2 * slots;
    flags2 = flags1 + flags;
    state = flags2 + flags;
    total = state + 64;
  }
};
}  // namespace

// how long a workgroup waits for a peer before it raises the error word, in ticks of the 100 MHz wall clock
static long g_ar_timeout_ticks = 60L * 100000000L;
extern "C" void areal_ipc_allreduce_set_timeout(double seconds) {
  g_ar_timeout_ticks = (long)(seconds * 1e8);
  if (g_ar_timeout_ticks < 1000) g_ar_timeout_ticks = 1000;
}

extern "C" long areal_ipc_allreduce_region_bytes(long cap_elems) { return ArLayout(cap_elems).total; }
// offset of the state words (sequence number, ticket, error word) inside a region
extern "C" long areal_ipc_allreduce_state_offset(long cap_elems) { return ArLayout(cap_elems).state; }

// `peer_bases[r]`: address at which rank r's region is mapped in this process (own region: its own address). `in` and
// `out` may be the same tensor. in / out must be 16-byte aligned; cap_elems a multiple of 8 and >= n.
extern "C" int areal_ipc_allreduce(const void* in, void* out, long n, void* const* peer_bases, int nranks, int rank,
                                   long cap_elems, hipStream_t st) {
  if (n <= 0) return 0;
  if (nranks < 1 || nranks > AR_MAX_RANKS || rank < 0 || rank >= nranks) return -1;
  if (cap_elems % 8 != 0 || n > cap_elems) return -2;
  if (((uintptr_t)in | (uintptr_t)out) % 16 != 0) return -3;
  ArPeers peers;
  const ArLayout lay(cap_elems);
  for (int r = 0; r < AR_MAX_RANKS; ++r) {
    char* base = (char*)peer_bases[r < nranks ? r : rank];
    peers.data[r] = (const bf16_t*)base;
    peers.flags[r] = (int*)(base + lay.flags1);
  }
  char* mine = (char*)peer_bases[rank];
  int* my_flags = (int*)(mine + lay.flags1);
  unsigned* state = (unsigned*)(mine + lay.state);
  const long nvec = (n + 7) / 8;
  long blocks = (nvec + AR_THREADS * 2 - 1) / (AR_THREADS * 2);  // >= 2 vectors per thread before a second workgroup
  if (blocks < 1) blocks = 1;
  if (blocks > AR_MAX_BLOCKS) blocks = AR_MAX_BLOCKS;
  hipLaunchKernelGGL(ipc_allreduce_kernel, dim3((unsigned)blocks), dim3(AR_THREADS), 0, st, peers, (bf16_t*)mine, my_flags,
                     state, (const bf16_t*)in, (bf16_t*)out, n, cap_elems, nranks, rank, g_ar_timeout_ticks);
  return (int)hipGetLastError();
}

// Two-shot variant of areal_ipc_allreduce (same arguments and region): reduce-scatter + all-gather, for messages from
// a few hundred KB and groups of four or more ranks.
extern "C" int areal_ipc_allreduce_two_shot(const void* in, void* out, long n, void* const* peer_bases, int nranks,
                                            int rank, long cap_elems, hipStream_t st) {
  if (n <= 0) return 0;
  if (nranks < 1 || nranks > AR_MAX_RANKS || rank < 0 || rank >= nranks) return -1;
  if (cap_elems % 8 != 0 || n > cap_elems) return -2;
  if (((uintptr_t)in | (uintptr_t)out) % 16 != 0) return -3;
  ArPeers2 peers;
  const ArLayout lay(cap_elems);
  for (int r = 0; r < AR_MAX_RANKS; ++r) {
    char* base = (char*)peer_bases[r < nranks ? r : rank];
    peers.data[r] = (const bf16_t*)base;
    peers.result[r] = (bf16_t*)(base + lay.result);
    peers.flags1[r] = (int*)(base + lay.flags1);
    peers.flags2[r] = (int*)(base + lay.flags2);
  }
  unsigned* state = (unsigned*)((char*)peer_bases[rank] + lay.state);
  const long nvec = (n + 7) / 8;
  long blocks = (nvec + AR_THREADS * 2 - 1) / (AR_THREADS * 2);
  if (blocks < 1) blocks = 1;
  if (blocks > AR_MAX_BLOCKS) blocks = AR_MAX_BLOCKS;
  hipLaunchKernelGGL(ipc_allreduce_two_shot_kernel, dim3((unsigned)blocks), dim3(AR_THREADS), 0, st, peers, state,
                     (const bf16_t*)in, (bf16_t*)out, n, cap_elems, nranks, rank, g_ar_timeout_ticks);
  return (int)hipGetLastError();
}
