// Trainer collectives over peer-mapped memory: one kernel that is an all-to-all of equal pieces with strided source and
// destination rows, and from it the sequence<->head exchange of Ulysses attention (the transposes are the kernel's address
// arithmetic, not extra copies), all-gather, and reduce-scatter (fp32 or bf16 sums in rank order).
//
// Every rank owns one exported region
//   [ slot 0 | slot 1 | flags[2][8][MAX_BLOCKS] | state ]
// and maps the regions of its peers (hipIpc: stores over xGMI between the GPUs of a node, local memory when processes share a
// GPU). A slot holds one piece from every rank, piece s at s * piece_bytes. A call with sequence number q uses slot q & 1.
// Workgroup b of every rank owns the same range of 16-byte vectors of every piece and
//   1. PUSHES that range of piece d of its input into slot[my rank] of rank d, for every d (posted stores: a link carries
//      them at full rate, where loads from a peer would wait a round trip each), starting with rank + 1 so that at any time
//      the ranks target different links,
//   2. publishes q in flags[slot][my rank][b] of every peer (release, system scope) and waits for theirs in its own table,
//   3. moves the range of every received piece from its slot to the output: scattered to strided rows (all-to-all /
//      all-gather), or summed over the pieces in rank order (reduce-scatter; the same bits whatever the arrival order).
// There is no grid-wide barrier (a workgroup only needs workgroup b of the peers), and no closing barrier: slot q & 1 is
// written again by a peer's call q + 2, which starts after that peer's call q + 1 has seen this rank's flag q + 1, and this
// rank set that flag after its call q had finished (stream order). All calls on one region must therefore be issued on ONE
// stream per rank, in the same order on every rank. The sequence number lives in device memory and is advanced by the last
// workgroup to finish, so captured graphs replay. A peer that does not arrive within the time limit raises the error word
// instead of hanging the device.
//
// Role in the reference: the collectives its trainers issue through NCCL - the Ulysses all-to-all pair
// (areal/utils/ulysses.py:125-149), sequence-parallel all-gather / reduce-scatter around tensor-parallel linears
// (realhf/impl/model/parallelism/tensor_parallel/modules.py:242-402) and the sharded optimizer's gradient reduce-scatter and
// parameter all-gather (areal/utils/fsdp.py:253-285).
#include "common.hpp"

namespace {
constexpr int PC_MAX_RANKS = 8;
constexpr int PC_MAX_BLOCKS = 128;
constexpr int PC_THREADS = 512;

struct PcArgs {
  char* slot[PC_MAX_RANKS];  // base of every rank's region (the two slots come first)
  int* flags[PC_MAX_RANKS];  // every rank's flag table
  unsigned* state;           // this rank's state words: [0] sequence number, [1] ticket, [2] error word
  const char* in;
  char* out;
  long slot_bytes;
  unsigned piece_vecs;  // 16-byte vectors per piece
  unsigned row_vecs;    // ... per row of a piece (piece_vecs when the piece is one contiguous run)
  long in_piece_stride, in_row_stride;    // bytes from piece d to d + 1 of the input / between rows of a piece
  long out_piece_stride, out_row_stride;  // the same for the output (pieces by SOURCE rank)
  int nranks, rank;
  long timeout_ticks;
};

DEVINL void pc_wait(const int* f, unsigned seq, unsigned* state, long timeout_ticks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  if (__hip_atomic_load(&state[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) timeout_ticks = 0;
  while ((int)(__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - (int)seq) < 0) {
    __builtin_amdgcn_s_sleep(16);
    if ((long)(__builtin_amdgcn_s_memrealtime() - t0) > timeout_ticks) {
      atomicExch(&state[2], 1u);
      break;
    }
  }
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

DEVINL long pc_off(unsigned v, unsigned row_vecs, long row_stride) {
  const unsigned row = v / row_vecs;
  return (long)row * row_stride + (long)(v - row * row_vecs) * 16;
}

// MODE 0: scatter the received pieces; 1: sum them as fp32; 2: sum them as bf16 (fp32 accumulation, one rounding)
template <int MODE>
__global__ __launch_bounds__(PC_THREADS) void peer_exchange_kernel(PcArgs a) {
  const int tid = threadIdx.x, b = blockIdx.x;
  const unsigned seq = __hip_atomic_load(&a.state[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
  const long slot_off = (long)(seq & 1u) * a.slot_bytes;
  const unsigned per = (a.piece_vecs + gridDim.x - 1) / gridDim.x;
  const unsigned v0 = (unsigned)b * per < a.piece_vecs ? (unsigned)b * per : a.piece_vecs;
  const unsigned v1 = v0 + per < a.piece_vecs ? v0 + per : a.piece_vecs;
  const long piece_bytes = (long)a.piece_vecs * 16;
  const bool flat_in = a.row_vecs == a.piece_vecs, flat_out = flat_in;
  // 1. push
  for (int i = 1; i <= a.nranks; ++i) {
    const int d = (a.rank + i) % a.nranks;
    char* dst = a.slot[d] + slot_off + (long)a.rank * piece_bytes;
    const char* src = a.in + (long)d * a.in_piece_stride;
    unsigned v = v0 + tid;
    for (; v + 3 * PC_THREADS < v1; v += 4 * PC_THREADS) {  // four loads in flight per thread
      u32x4 x[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned w = v + j * PC_THREADS;
        x[j] = *reinterpret_cast<const u32x4*>(src + (flat_in ? (long)w * 16 : pc_off(w, a.row_vecs, a.in_row_stride)));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) *reinterpret_cast<u32x4*>(dst + (long)(v + j * PC_THREADS) * 16) = x[j];
    }
    for (; v < v1; v += PC_THREADS)
      *reinterpret_cast<u32x4*>(dst + (long)v * 16) =
          *reinterpret_cast<const u32x4*>(src + (flat_in ? (long)v * 16 : pc_off(v, a.row_vecs, a.in_row_stride)));
  }
  __threadfence_system();
  __syncthreads();
  // 2. signal every peer, wait for every peer
  const int fidx = (int)(seq & 1u) * PC_MAX_RANKS * PC_MAX_BLOCKS + b;
  if (tid < a.nranks)
    __hip_atomic_store(a.flags[tid] + fidx + a.rank * PC_MAX_BLOCKS, (int)seq, __ATOMIC_RELEASE,
                       __HIP_MEMORY_SCOPE_SYSTEM);
  if (tid < a.nranks) pc_wait(a.flags[a.rank] + fidx + tid * PC_MAX_BLOCKS, seq, a.state, a.timeout_ticks);
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  // 3. out of the slot
  const char* mine = a.slot[a.rank] + slot_off;
  if (MODE == 0) {
    for (int s = 0; s < a.nranks; ++s) {
      const char* src = mine + (long)s * piece_bytes;
      char* dst = a.out + (long)s * a.out_piece_stride;
      unsigned v = v0 + tid;
      for (; v + 3 * PC_THREADS < v1; v += 4 * PC_THREADS) {
        u32x4 x[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) x[j] = *reinterpret_cast<const u32x4*>(src + (long)(v + j * PC_THREADS) * 16);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const unsigned w = v + j * PC_THREADS;
          *reinterpret_cast<u32x4*>(dst + (flat_out ? (long)w * 16 : pc_off(w, a.row_vecs, a.out_row_stride))) = x[j];
        }
      }
      for (; v < v1; v += PC_THREADS)
        *reinterpret_cast<u32x4*>(dst + (flat_out ? (long)v * 16 : pc_off(v, a.row_vecs, a.out_row_stride))) =
            *reinterpret_cast<const u32x4*>(src + (long)v * 16);
    }
  } else {
    for (unsigned v = v0 + tid; v < v1; v += PC_THREADS) {
      char* dst = a.out + (flat_out ? (long)v * 16 : pc_off(v, a.row_vecs, a.out_row_stride));
      if (MODE == 1) {
        float4 acc = *reinterpret_cast<const float4*>(mine + (long)v * 16);
        for (int s = 1; s < a.nranks; ++s) {
          const float4 x = *reinterpret_cast<const float4*>(mine + (long)s * piece_bytes + (long)v * 16);
          acc.x += x.x, acc.y += x.y, acc.z += x.z, acc.w += x.w;
        }
        *reinterpret_cast<float4*>(dst) = acc;
      } else {
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        for (int s = 0; s < a.nranks; ++s) {
          const bf16x8 x = *reinterpret_cast<const bf16x8*>(mine + (long)s * piece_bytes + (long)v * 16);
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] += bf2f(x[j]);
        }
        bf16x8 y;
#pragma unroll
        for (int j = 0; j < 8; ++j) y[j] = f2bf(acc[j]);
        *reinterpret_cast<bf16x8*>(dst) = y;
      }
    }
  }
  // the last workgroup to finish advances the sequence number (every workgroup has read it by then)
  __syncthreads();
  if (tid == 0) {
    __threadfence();
    const unsigned t = atomicAdd(&a.state[1], 1u);
    if (t == gridDim.x - 1) {
      atomicExch(&a.state[1], 0u);
      __hip_atomic_store(&a.state[0], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

struct PcLayout {
  long flags, state, total;
  explicit PcLayout(long slot_bytes) {
    flags = 2 * slot_bytes;
    state = flags + 2L * PC_MAX_RANKS * PC_MAX_BLOCKS * (long)sizeof(int);
    total = state + 64;
  }
};
}  // namespace

static long g_pc_timeout_ticks = 120L * 100000000L;
extern "C" void areal_peer_set_timeout(double seconds) {
  g_pc_timeout_ticks = (long)(seconds * 1e8);
  if (g_pc_timeout_ticks < 1000) g_pc_timeout_ticks = 1000;
}

extern "C" long areal_peer_region_bytes(long slot_bytes) { return PcLayout(slot_bytes).total; }
extern "C" long areal_peer_state_offset(long slot_bytes) { return PcLayout(slot_bytes).state; }

// One exchange of `nranks` equal pieces of `piece_bytes` (a multiple of 16) made of rows of `row_bytes` (a multiple of 16
// that divides piece_bytes). Input piece d (sent to rank d) starts at in + d * in_piece_stride, its rows in_row_stride apart;
// the piece received from rank s goes to out + s * out_piece_stride with rows out_row_stride apart (mode 0), or all received
// pieces are summed into the rows of out (mode 1: fp32, mode 2: bf16). `peer_bases[r]`: where rank r's region is mapped in
// this process. `max_blocks` bounds the workgroups (0: default), so that the exchange leaves compute units to kernels of
// other streams.
extern "C" int areal_peer_exchange(const void* in, void* out, long piece_bytes, long row_bytes, long in_piece_stride,
                                   long in_row_stride, long out_piece_stride, long out_row_stride, int mode,
                                   void* const* peer_bases, int nranks, int rank, long slot_bytes, int max_blocks,
                                   hipStream_t st) {
  if (piece_bytes <= 0) return 0;
  if (nranks < 1 || nranks > PC_MAX_RANKS || rank < 0 || rank >= nranks) return -1;
  if (piece_bytes % 16 != 0 || row_bytes <= 0 || row_bytes % 16 != 0 || piece_bytes % row_bytes != 0) return -2;
  if ((long)nranks * piece_bytes > slot_bytes || piece_bytes / 16 >= (1L << 31)) return -3;
  if (((uintptr_t)in | (uintptr_t)out) % 16 != 0) return -4;
  if ((in_piece_stride | in_row_stride | out_piece_stride | out_row_stride) % 16 != 0) return -5;
  if (mode < 0 || mode > 2) return -6;
  PcArgs a;
  const PcLayout lay(slot_bytes);
  for (int r = 0; r < PC_MAX_RANKS; ++r) {
    char* base = (char*)peer_bases[r < nranks ? r : rank];
    a.slot[r] = base;
    a.flags[r] = (int*)(base + lay.flags);
  }
  a.state = (unsigned*)((char*)peer_bases[rank] + lay.state);
  a.in = (const char*)in;
  a.out = (char*)out;
  a.slot_bytes = slot_bytes;
  a.piece_vecs = (unsigned)(piece_bytes / 16);
  a.row_vecs = (unsigned)(row_bytes / 16);
  a.in_piece_stride = in_piece_stride, a.in_row_stride = in_row_stride;
  a.out_piece_stride = out_piece_stride, a.out_row_stride = out_row_stride;
  a.nranks = nranks, a.rank = rank;
  a.timeout_ticks = g_pc_timeout_ticks;
  // >= 4 vectors per thread and destination before another workgroup is worth its flag traffic
  long blocks = ((long)a.piece_vecs + PC_THREADS * 4 - 1) / (PC_THREADS * 4);
  const long cap = max_blocks > 0 && max_blocks < PC_MAX_BLOCKS ? max_blocks : PC_MAX_BLOCKS;
  if (blocks < 1) blocks = 1;
  if (blocks > cap) blocks = cap;
  if (mode == 0)
    hipLaunchKernelGGL(peer_exchange_kernel<0>, dim3((unsigned)blocks), dim3(PC_THREADS), 0, st, a);
  else if (mode == 1)
    hipLaunchKernelGGL(peer_exchange_kernel<1>, dim3((unsigned)blocks), dim3(PC_THREADS), 0, st, a);
  else
    hipLaunchKernelGGL(peer_exchange_kernel<2>, dim3((unsigned)blocks), dim3(PC_THREADS), 0, st, a);
  return (int)hipGetLastError();
}
