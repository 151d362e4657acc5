// Token routing glue of mixture-of-experts layers on gfx950: the permutation of token rows into expert order and
// the weighted recombination of the expert outputs, both as GATHERS (every output row is written once by one
// workgroup, no atomics, deterministic) - the scatter-add formulations (index_add / indexing backward) serialise on
// the k copies of every token. The reference does this inside Megatron-style token dispatchers
// (realhf/impl/model/modules/moe/token_dispatcher.py).
//
//   gather_rows : out[p, :] = scale[p] * src[idx[p], :]                  (permute forward; combine backward)
//   combine_rows: out[t, :] = sum_j w[t, j] * src[inv[t, j], :]         (combine forward; permute backward, w = 1)
//   combine_dots: dw[t, j]  = < g[t, :], src[inv[t, j], :] >             (gradient of the combine weights)
// Rows are H bf16 values, H a multiple of 8; one workgroup of 256 threads walks a row in 16-byte pieces.
#include "common.hpp"

__global__ __launch_bounds__(256) void moe_gather_rows_kernel(const bf16_t* __restrict__ src,
                                                              const int64_t* __restrict__ idx,
                                                              const bf16_t* __restrict__ scale,
                                                              bf16_t* __restrict__ out, int64_t n_rows, int H) {
  const int64_t p = blockIdx.x;
  if (p >= n_rows) return;
  const bf16_t* s = src + idx[p] * (int64_t)H;
  bf16_t* o = out + p * (int64_t)H;
  const float sc = scale ? bf2f(scale[p]) : 1.f;
  for (int c = threadIdx.x * 8; c < H; c += 256 * 8) {
    bf16x8 v = ld8(s + c);
    if (scale) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = f2bf(bf2f(v[j]) * sc);
    }
    st8(o + c, v);
  }
}

template <int KMAX>
__global__ __launch_bounds__(256) void moe_combine_rows_kernel(const bf16_t* __restrict__ src,
                                                               const int64_t* __restrict__ inv,
                                                               const bf16_t* __restrict__ w,
                                                               bf16_t* __restrict__ out, int64_t T, int k, int H) {
  const int64_t t = blockIdx.x;
  if (t >= T) return;
  int64_t rows[KMAX];
  float wt[KMAX];
#pragma unroll
  for (int j = 0; j < KMAX; ++j) {
    rows[j] = j < k ? inv[t * k + j] : 0;
    wt[j] = j < k ? (w ? bf2f(w[t * k + j]) : 1.f) : 0.f;
  }
  bf16_t* o = out + t * (int64_t)H;
  for (int c = threadIdx.x * 8; c < H; c += 256 * 8) {
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
#pragma unroll
    for (int j = 0; j < KMAX; ++j) {
      if (j < k) {
        const bf16x8 v = ld8(src + rows[j] * (int64_t)H + c);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] += wt[j] * bf2f(v[i]);
      }
    }
    bf16x8 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = f2bf(acc[i]);
    st8(o + c, r);
  }
}

__global__ __launch_bounds__(256) void moe_combine_dots_kernel(const bf16_t* __restrict__ g,
                                                               const bf16_t* __restrict__ src,
                                                               const int64_t* __restrict__ inv,
                                                               bf16_t* __restrict__ dw, int64_t T, int k, int H) {
  __shared__ float red[16];
  const int64_t t = blockIdx.x;
  if (t >= T) return;
  const bf16_t* gr = g + t * (int64_t)H;
  for (int j = 0; j < k; ++j) {
    const bf16_t* s = src + inv[t * k + j] * (int64_t)H;
    float acc = 0.f;
    for (int c = threadIdx.x * 8; c < H; c += 256 * 8) {
      const bf16x8 a = ld8(gr + c), b = ld8(s + c);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc += bf2f(a[i]) * bf2f(b[i]);
    }
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) dw[t * k + j] = f2bf(acc);
  }
}

static int check_rows(const void* a, const void* b, int H) {
  if (H % 8 != 0) return -1;
  if (((uintptr_t)a | (uintptr_t)b) % 16 != 0) return -2;
  return 0;
}

extern "C" int areal_moe_gather_rows(const void* src, const int64_t* idx, const void* scale, void* out,
                                     int64_t n_rows, int H, hipStream_t st) {
  if (n_rows == 0) return 0;
  if (int e = check_rows(src, out, H)) return e;
  hipLaunchKernelGGL(moe_gather_rows_kernel, dim3((unsigned)n_rows), dim3(256), 0, st, (const bf16_t*)src, idx,
                     (const bf16_t*)scale, (bf16_t*)out, n_rows, H);
  return (int)hipGetLastError();
}

extern "C" int areal_moe_combine_rows(const void* src, const int64_t* inv, const void* w, void* out, int64_t T,
                                      int k, int H, hipStream_t st) {
  if (T == 0) return 0;
  if (int e = check_rows(src, out, H)) return e;
  if (k < 1 || k > 16) return -3;
  dim3 grid((unsigned)T), block(256);
  if (k <= 2)
    hipLaunchKernelGGL(moe_combine_rows_kernel<2>, grid, block, 0, st, (const bf16_t*)src, inv, (const bf16_t*)w,
                       (bf16_t*)out, T, k, H);
  else if (k <= 8)
    hipLaunchKernelGGL(moe_combine_rows_kernel<8>, grid, block, 0, st, (const bf16_t*)src, inv, (const bf16_t*)w,
                       (bf16_t*)out, T, k, H);
  else
    hipLaunchKernelGGL(moe_combine_rows_kernel<16>, grid, block, 0, st, (const bf16_t*)src, inv, (const bf16_t*)w,
                       (bf16_t*)out, T, k, H);
  return (int)hipGetLastError();
}

extern "C" int areal_moe_combine_dots(const void* g, const void* src, const int64_t* inv, void* dw, int64_t T,
                                      int k, int H, hipStream_t st) {
  if (T == 0) return 0;
  if (int e = check_rows(g, src, H)) return e;
  hipLaunchKernelGGL(moe_combine_dots_kernel, dim3((unsigned)T), dim3(256), 0, st, (const bf16_t*)g,
                     (const bf16_t*)src, inv, (bf16_t*)dw, T, k, H);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// Routing: softmax + top-k of the router logits, and the expert-sorted order of the (token, choice) pairs, as a few
// fixed launches. Done with library primitives the same work is a dozen dependent small kernels per layer (softmax,
// top-k, sort, index-put, bincount, cumsum, floor-divide ...), about a third of a decode step of a 128-expert model.
//
//   topk : one wave per token. Lane l holds logits l, l + 64, ...; the wave takes max and sum-exp, then extracts the
//          k largest logits one by one (value descending, index ascending on ties); weight = softmax probability of
//          the choice, divided by the sum over the k choices when `norm`.
//   sort : counting sort by expert, STABLE in the pair index (the order torch.argsort(stable=True) gives). The pairs
//          are cut in chunks; `count` builds per-chunk histograms, `scan` turns them into chunk bases and the experts'
//          row ranges, `place` ranks every pair among the same-expert pairs before it in its chunk. A list that fits
//          one chunk takes one launch (histogram, scan and placement by one workgroup).
// Expert ids may be 0 .. E: id E is the bin of dropped pairs, sorted behind every expert.
constexpr int RT_MAX_PER_LANE = 8;  // experts <= 512

template <typename TIN, int PER_LANE>
__global__ __launch_bounds__(256) void moe_topk_kernel(const TIN* __restrict__ logits, int64_t T, int E, int k,
                                                       int norm, int64_t* __restrict__ topi,
                                                       float* __restrict__ topv, bf16_t* __restrict__ topv_bf) {
  const int lane = threadIdx.x & 63;
  const int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= T) return;
  const TIN* row = logits + t * (int64_t)E;
  float v[PER_LANE];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < PER_LANE; ++i) {
    const int e = lane + 64 * i;
    v[i] = e < E ? (float)row[e] : -INFINITY;
    mx = fmaxf(mx, v[i]);
  }
  mx = wave_max(mx);
  float se = 0.f;
#pragma unroll
  for (int i = 0; i < PER_LANE; ++i) se += v[i] > -INFINITY ? __expf(v[i] - mx) : 0.f;
  se = wave_sum(se);
  float mine_v = 0.f;  // lane j keeps choice j
  int mine_i = 0;
  float picked = 0.f;
  for (int j = 0; j < k; ++j) {
    float bv = -INFINITY;
    int bi = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < PER_LANE; ++i) {  // ascending expert index within the lane: first maximum wins
      if (v[i] > bv) {
        bv = v[i];
        bi = lane + 64 * i;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ov > bv || (ov == bv && oi < bi)) {
        bv = ov;
        bi = oi;
      }
    }
    if (bi >= E) bi = 0;  // fewer finite logits than k: cannot happen for finite router outputs
    const float p = __expf(bv - mx) / se;
    picked += p;
    if (lane == j) {
      mine_v = p;
      mine_i = bi;
    }
    if ((bi & 63) == lane) {
#pragma unroll
      for (int i = 0; i < PER_LANE; ++i)
        if (i == (bi >> 6)) v[i] = -INFINITY;
    }
  }
  if (lane < k) {
    topi[t * k + lane] = mine_i;
    const float w = norm ? mine_v / picked : mine_v;
    if (topv) topv[t * k + lane] = w;
    if (topv_bf) topv_bf[t * k + lane] = f2bf(w);
  }
}

constexpr int SORT_BINS_MAX = 513;

// ranks of one 256-pair tile and their placement; `run[e]` = next free position of expert e, advanced per tile
DEVINL void sort_place_tile(const int64_t* __restrict__ flat_e, int64_t n, int64_t base, int k, int* run, int* tile,
                            int64_t* __restrict__ order, int64_t* __restrict__ inv, int64_t* __restrict__ src_row) {
  const int i = threadIdx.x;
  const int64_t p = base + i;
  const int e = p < n ? (int)flat_e[p] : -1;
  tile[i] = e;
  __syncthreads();
  int pos = -1;
  if (e >= 0) {
    int r = 0;
    for (int j = 0; j < i; ++j) r += tile[j] == e ? 1 : 0;
    pos = run[e] + r;
  }
  __syncthreads();
  if (e >= 0) {
    atomicAdd(&run[e], 1);  // counts only: the positions above were taken from the value before this tile
    order[pos] = p;
    inv[p] = pos;
    src_row[pos] = p / k;
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void moe_sort_count_kernel(const int64_t* __restrict__ flat_e, int64_t n, int bins,
                                                             int items, int* __restrict__ counts) {
  __shared__ int hist[SORT_BINS_MAX];
  for (int e = threadIdx.x; e < bins; e += 256) hist[e] = 0;
  __syncthreads();
  const int64_t lo = (int64_t)blockIdx.x * items * 256;
  for (int s = 0; s < items; ++s) {
    const int64_t p = lo + (int64_t)s * 256 + threadIdx.x;
    if (p < n) atomicAdd(&hist[(int)flat_e[p]], 1);
  }
  __syncthreads();
  for (int e = threadIdx.x; e < bins; e += 256) counts[(size_t)blockIdx.x * bins + e] = hist[e];
}

// counts [C, bins] -> exclusive chunk bases in place, offsets [bins + 1]
__global__ __launch_bounds__(1024) void moe_sort_scan_kernel(int* __restrict__ counts, int C, int bins,
                                                             int* __restrict__ offsets) {
  __shared__ int total[SORT_BINS_MAX];
  for (int e = threadIdx.x; e < bins; e += 1024) {
    int acc = 0;
    int c = 0;
    for (; c + 4 <= C; c += 4) {  // four independent loads per round
      const int a0 = counts[(size_t)c * bins + e], a1 = counts[(size_t)(c + 1) * bins + e];
      const int a2 = counts[(size_t)(c + 2) * bins + e], a3 = counts[(size_t)(c + 3) * bins + e];
      counts[(size_t)c * bins + e] = acc;
      counts[(size_t)(c + 1) * bins + e] = acc + a0;
      counts[(size_t)(c + 2) * bins + e] = acc + a0 + a1;
      counts[(size_t)(c + 3) * bins + e] = acc + a0 + a1 + a2;
      acc += a0 + a1 + a2 + a3;
    }
    for (; c < C; ++c) {
      const int a = counts[(size_t)c * bins + e];
      counts[(size_t)c * bins + e] = acc;
      acc += a;
    }
    total[e] = acc;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int acc = 0;
    for (int e = 0; e < bins; ++e) {
      offsets[e] = acc;
      acc += total[e];
    }
    offsets[bins] = acc;
  }
}

__global__ __launch_bounds__(256) void moe_sort_place_kernel(const int64_t* __restrict__ flat_e, int64_t n, int bins,
                                                             int items, int k, const int* __restrict__ bases,
                                                             const int* __restrict__ offsets,
                                                             int64_t* __restrict__ order, int64_t* __restrict__ inv,
                                                             int64_t* __restrict__ src_row) {
  __shared__ int run[SORT_BINS_MAX];
  __shared__ int tile[256];
  for (int e = threadIdx.x; e < bins; e += 256) run[e] = offsets[e] + bases[(size_t)blockIdx.x * bins + e];
  __syncthreads();
  const int64_t lo = (int64_t)blockIdx.x * items * 256;
  for (int s = 0; s < items; ++s) sort_place_tile(flat_e, n, lo + (int64_t)s * 256, k, run, tile, order, inv, src_row);
}

// rank of every lane among the lanes BEFORE it that hold the same value (64 scalar broadcasts, no LDS)
DEVINL int wave_rank_of_equal(int e, int lane) {
  int r = 0;
#pragma unroll
  for (int j = 0; j < 64; ++j) {
    const int ej = __builtin_amdgcn_readlane(e, j);
    r += (ej == e && j < lane) ? 1 : 0;
  }
  return r;
}

// The whole list in one workgroup of 16 waves: the list is cut in 64-pair slices; every wave histograms its slices
// into LDS (slice x bin), one thread per bin turns the slice counts into slice bases, the bins are scanned, and
// every wave places its slices (rank inside a slice by scalar broadcasts). `table` holds slices x bins ints.
__global__ __launch_bounds__(1024) void moe_sort_single_kernel(const int64_t* __restrict__ flat_e, int64_t n,
                                                               int bins, int k, int* __restrict__ offsets,
                                                               int64_t* __restrict__ order, int64_t* __restrict__ inv,
                                                               int64_t* __restrict__ src_row) {
  extern __shared__ int table[];  // [slices][bins], then totals [bins]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int slices = (int)((n + 63) >> 6);
  int* total = table + slices * bins;
  for (int i = threadIdx.x; i < slices * bins + bins; i += 1024) table[i] = 0;
  __syncthreads();
  for (int s = wave; s < slices; s += 16) {
    const int64_t p = (int64_t)s * 64 + lane;
    if (p < n) atomicAdd(&table[s * bins + (int)flat_e[p]], 1);
  }
  __syncthreads();
  for (int e = threadIdx.x; e < bins; e += 1024) {
    int acc = 0;
    for (int s = 0; s < slices; ++s) {
      const int c = table[s * bins + e];
      table[s * bins + e] = acc;
      acc += c;
    }
    total[e] = acc;
  }
  __syncthreads();
  if (wave == 0) {  // exclusive scan of the bin totals, 64 bins per round
    int carry = 0;
    for (int b0 = 0; b0 < bins; b0 += 64) {
      const int e = b0 + lane;
      const int c = e < bins ? total[e] : 0;
      int x = c;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
      }
      if (e < bins) {
        total[e] = carry + x - c;
        offsets[e] = carry + x - c;
      }
      carry += __shfl(x, 63, 64);
    }
    if (lane == 0) offsets[bins] = carry;
  }
  __syncthreads();
  for (int s = wave; s < slices; s += 16) {
    const int64_t p = (int64_t)s * 64 + lane;
    const int e = p < n ? (int)flat_e[p] : -1 - lane;  // idle lanes: distinct values nobody matches
    const int r = wave_rank_of_equal(e, lane);
    if (p < n) {
      const int pos = total[e] + table[s * bins + e] + r;
      order[pos] = p;
      inv[p] = pos;
      src_row[pos] = p / k;
    }
  }
}

template <typename TIN>
static int topk_launch(const TIN* logits, int64_t T, int E, int k, int norm, int64_t* topi, float* topv,
                       bf16_t* topv_bf, hipStream_t st) {
  const dim3 grid((unsigned)((T + 3) / 4)), block(256);
#define TOPK_CASE(P) \
  if (E <= 64 * P) { \
    hipLaunchKernelGGL((moe_topk_kernel<TIN, P>), grid, block, 0, st, logits, T, E, k, norm, topi, topv, topv_bf); \
    return (int)hipGetLastError(); \
  }
  TOPK_CASE(1) TOPK_CASE(2) TOPK_CASE(4) TOPK_CASE(8)
#undef TOPK_CASE
  return -1;
}

// logits [T, E] fp32 (dtype 1) or bf16 (dtype 0); weights are written as fp32 (topv), bf16 (topv_bf) or both.
extern "C" int areal_moe_topk(const void* logits, int dtype, int64_t T, int E, int k, int norm, int64_t* topi,
                              float* topv, void* topv_bf, hipStream_t st) {
  if (T == 0) return 0;
  if (E < 1 || E > 64 * RT_MAX_PER_LANE || k < 1 || k > 64 || k > E) return -1;
  if (dtype == 0) return topk_launch((const bf16_t*)logits, T, E, k, norm, topi, topv, (bf16_t*)topv_bf, st);
  return topk_launch((const float*)logits, T, E, k, norm, topi, topv, (bf16_t*)topv_bf, st);
}

// Largest list the one-workgroup sort takes (its slice x bin table has to fit 64 KiB of LDS for every bin count up
// to the maximum); chunks of the three-launch form hold `items` tiles of 256 pairs.
static bool sort_single_fits(int64_t n, int bins) {
  const int64_t slices = (n + 63) / 64;
  return n <= 8192 && (slices * bins + bins) * 4 <= 64 * 1024;
}

// 1 when a list of n pairs over `bins` bins is sorted by the one-workgroup kernel (no workspace needed)
extern "C" int areal_moe_sort_single(int64_t n, int bins) { return sort_single_fits(n, bins) ? 1 : 0; }

// flat_e [n] expert id per pair (0 .. E; E = dropped), bins = E + 1. offsets [bins + 1]. ws: C * bins ints with
// C = ceil(n / (256 * items)) (unused when areal_moe_sort_single(n, bins)).
extern "C" int areal_moe_sort(const int64_t* flat_e, int64_t n, int bins, int k, int64_t* order, int64_t* inv,
                              int64_t* src_row, int* offsets, int* ws, int items, hipStream_t st) {
  if (bins < 1 || bins > SORT_BINS_MAX || k < 1) return -1;
  if (sort_single_fits(n, bins)) {
    const size_t lds = (size_t)(((n + 63) / 64) * bins + bins) * 4;
    hipLaunchKernelGGL(moe_sort_single_kernel, dim3(1), dim3(1024), lds, st, flat_e, n, bins, k, offsets, order, inv,
                       src_row);
    return (int)hipGetLastError();
  }
  if (items < 1 || !ws) return -2;
  const int64_t per = (int64_t)items * 256;
  const int C = (int)((n + per - 1) / per);
  hipLaunchKernelGGL(moe_sort_count_kernel, dim3(C), dim3(256), 0, st, flat_e, n, bins, items, ws);
  hipLaunchKernelGGL(moe_sort_scan_kernel, dim3(1), dim3(1024), 0, st, ws, C, bins, offsets);
  hipLaunchKernelGGL(moe_sort_place_kernel, dim3(C), dim3(256), 0, st, flat_e, n, bins, items, k, ws, offsets, order,
                     inv, src_row);
  return (int)hipGetLastError();
}
