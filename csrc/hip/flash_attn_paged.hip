// Prefill attention against the PAGED KV cache (head size 128): the queries of a prefill chunk attend to the whole
// context of their request - a reused prefix (prefix cache hit, multi-turn continuation), earlier chunks of a chunked
// prefill, and the chunk itself - read straight out of the pages through the block table, in the cache's own
// layouts: K [P, Hkv, 32, D] (a key's D values contiguous) and V [P, Hkv, D, 32] (a page's 32 tokens of one channel
// contiguous). No gather of the context into contiguous tensors, no un-transposing of V.
//
// Structure of flash_fwd_kernel (flash_attn.hip): 4 waves x 32 query rows, 64-key tiles (= two pages), S^T = K.Q^T
// with the query on the lane, online softmax lane-local, O^T += V^T.P^T with P taken from the accumulator
// registers. Differences:
//   * K rows are addressed page by page;
//   * the V tile lives in LDS as V^T [128 channels][64 keys] - exactly what the cache holds - so the A operand of
//     O^T += V^T.P^T is read with plain 8-byte row reads instead of transposed reads. 128-byte rows, the 8-byte
//     chunks XOR-swizzled with (row >> 1) & 15: the 32 lanes of a half-wave read 32 different bank pairs;
//   * cache slots past the end of a context (the rest of its last page) hold whatever an earlier request left
//     there: their V values are zeroed on the way into LDS (their probabilities are zero, but 0 x NaN is not).
// Causal masking is aligned bottom-right: query i of a chunk of q_len rows sees keys 0 .. k_len - q_len + i.
//
// Role in the reference: prefill against the radix cache inside its serving engine, and
// flash_attn_with_kvcache at realhf/impl/model/modules/attn.py:272.
#include "flash_tiles.hpp"

// FP8: the caches hold e4m3 bytes (same layouts): a thread fetches 8 bytes per fragment and widens them to bf16 on the
// way into LDS; everything behind the staging is the bf16 kernel.
template <bool FP8>
__global__ __launch_bounds__(256, 2) void flash_fwd_paged_kernel(
    const bf16_t* __restrict__ q, const void* __restrict__ kcache, const void* __restrict__ vcache,
    const int* __restrict__ block_tables, int max_pages, bf16_t* __restrict__ o, float* __restrict__ lse,
    const int* __restrict__ cu_q, const int* __restrict__ cu_k, int64_t q_stride, int64_t o_stride, int Hq, int Hkv,
    int nbatch, int total_q, float scale_log2, int causal) {
  constexpr int D = 128;
  __shared__ __attribute__((aligned(16))) char smem[2 * 64 * 256];
  char* ks = smem;             // K tile  [64 keys][256 B], 16-byte chunks swizzled (flash_tiles.hpp)
  char* vs = smem + 64 * 256;  // V^T tile [128 channels][128 B = 64 keys], 8-byte chunks swizzled

  // grid order: see flash_fwd_kernel (heaviest query tiles first, query heads of one kv group on one XCD)
  const int G = Hq / Hkv;
  const int n_g = Hkv * nbatch, n_g_pad = (n_g + 7) & ~7;
  const int per_rank = G * n_g_pad;
  const int rank = blockIdx.x / per_rank, rem = blockIdx.x - rank * per_rank;
  const int hj = rem / n_g_pad, slot = rem - hj * n_g_pad;
  int g = slot - rank % n_g_pad;
  g += g < 0 ? n_g_pad : 0;
  if (g >= n_g) return;
  const int b = g / Hkv, kvh = g - b * Hkv, head = kvh * G + hj;
  const int q_beg = cu_q[b], q_len = cu_q[b + 1] - q_beg;
  const int k_len = cu_k[b + 1] - cu_k[b];
  const int ntile_q = (q_len + 127) / 128;
  if (rank >= ntile_q) return;
  const int qt = ntile_q - 1 - rank;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = lane & 31, hf = lane >> 5;
  const int shift = k_len - q_len;
  const int* bt = block_tables + (size_t)b * max_pages;
  const int last_page = (k_len - 1) >> 5;

  const int q0 = qt * 128 + wid * 32;
  const int my_q = q0 + r;
  const int my_q_c = my_q < q_len ? my_q : q_len - 1;
  bf16x8 qf[8];
  {
    const bf16_t* qp = q + (size_t)(q_beg + my_q_c) * q_stride + (size_t)head * D + hf * 8;
#pragma unroll
    for (int s = 0; s < 8; ++s) qf[s] = ld8(qp + s * 16);
  }
  f32x16 oacc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) oacc[i][j] = 0.f;
  float m = -INFINITY, l = 0.f;

  int k_hi = k_len;
  if (causal) {
    const int last_q = min(q_len, qt * 128 + 128) - 1;
    k_hi = min(k_len, last_q + shift + 1);
  }
  const int ntile_k = k_hi > 0 ? (k_hi + 63) / 64 : 0;

  // K: thread -> (key row = tid>>4 + 16 i, 16-byte chunk tid&15). V^T: piece id = tid + 256 i ->
  // (page half = id>>9, channel = (id>>2)&127, 8-token chunk = id&3): a wave reads 1 KiB of contiguous cache.
  const int lrow = tid >> 4, lch = tid & 15;
  bf16x8 kreg[4], vreg[4];
  auto load_tile = [&](int t) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int key = t * 64 + lrow + 16 * i;
      key = key < k_len ? key : k_len - 1;
      const int page = bt[key >> 5];
      kreg[i] = ld8_cache<FP8>(kcache, (((size_t)page * Hkv + kvh) * 32 + (key & 31)) * D + lch * 8);
      const int id = tid + 256 * i;
      const int half = id >> 9, ch = (id >> 2) & 127, c16 = id & 3;
      int pg = t * 2 + half;
      pg = pg < last_page ? pg : last_page;
      vreg[i] = ld8_cache<FP8>(vcache, (((size_t)bt[pg] * Hkv + kvh) * D + ch) * 32 + c16 * 8);
    }
  };
  auto store_tile = [&](int t) {
    const bool ragged = t * 64 + 64 > k_len;  // uniform over the block
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      lds_row_write(ks, lrow + 16 * i, lch, kreg[i]);
      const int id = tid + 256 * i;
      const int half = id >> 9, ch = (id >> 2) & 127, c16 = id & 3;
      bf16x8 v = vreg[i];
      if (ragged) {
        const int key0 = t * 64 + half * 32 + c16 * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (key0 + e >= k_len) v[e] = (bf16_t)0.f;
      }
      const int c8 = half * 8 + c16 * 2, sw = (ch >> 1) & 15;
      char* row = vs + ch * 128;
      bf16x4 lo = {v[0], v[1], v[2], v[3]}, hi = {v[4], v[5], v[6], v[7]};
      *reinterpret_cast<bf16x4*>(row + ((c8 ^ sw) << 3)) = lo;
      *reinterpret_cast<bf16x4*>(row + (((c8 + 1) ^ sw) << 3)) = hi;
    }
  };
  if (ntile_k > 0) {
    load_tile(0);
    store_tile(0);
  }
  __syncthreads();
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): see flash_fwd_kernel

  const int vsw = (r >> 1) & 15;  // swizzle key of this lane's V^T rows (32 db + r)
  for (int t = 0; t < ntile_k; ++t) {
    if (t + 1 < ntile_k) load_tile(t + 1);
    f32x16 sacc[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int j = 0; j < 16; ++j) sacc[kb][j] = 0.f;
#pragma unroll
      for (int s0 = 0; s0 < 8; s0 += 4) {
        bf16x8 kf[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) kf[s] = lds_row_read(ks, 32 * kb + r, 2 * (s0 + s) + hf);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 4; ++s) sacc[kb] = mfma32(kf[s], qf[s0 + s], sacc[kb]);
      }
    }
    const int kbase = t * 64;
    const bool need_mask = (kbase + 64 > k_len) || (causal && (kbase + 63 > q0 + shift));
    float mx = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        float sv = sacc[kb][j] * scale_log2;
        if (need_mask) {
          const int key = kbase + 32 * kb + (j & 3) + 8 * (j >> 2) + 4 * hf;
          if (key >= k_len || (causal && key > my_q + shift)) sv = -INFINITY;
        }
        sacc[kb][j] = sv;
        mx = fmaxf(mx, sv);
      }
    mx = max_xor32(mx);
    const float mn = fmaxf(m, mx);
    const float alpha = (mn == -INFINITY) ? 1.f : fast_exp2(m - mn);
    m = mn;
    float ls = 0.f;
    bf16x8 pf[4];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float pr = (m == -INFINITY) ? 0.f : fast_exp2(sacc[kb][j] - m);
        ls += pr;
        pf[2 * kb + (j >> 3)][j & 7] = f2bf(pr);
      }
    l = l * alpha + ls;
    if (__any(alpha != 1.f)) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) oacc[i][j] *= alpha;
    }
    // O^T[d, q] += V^T[d, keys] . P^T[keys, q]: k-step s covers keys 16 s .. 16 s + 15, and element j of lane half hf
    // of P is key 16 s + 4 hf + (j & 3) + 8 (j >> 2): the A operand takes the same keys, two 8-byte reads per row
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      bf16x4 a0[4], a1[4];
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        const char* row = vs + (32 * db + r) * 128;
        a0[db] = *reinterpret_cast<const bf16x4*>(row + (((4 * s + hf) ^ vsw) << 3));
        a1[db] = *reinterpret_cast<const bf16x4*>(row + (((4 * s + 2 + hf) ^ vsw) << 3));
      }
#pragma unroll
      for (int db = 0; db < 4; ++db) oacc[db] = mfma32(cat4(a0[db], a1[db]), pf[s], oacc[db]);
    }
    __syncthreads();
    if (t + 1 < ntile_k) store_tile(t + 1);
    __syncthreads();
  }
  l = sum_xor32(l);
  const float inv_l = l > 0.f ? 1.f / l : 0.f;
  if (my_q < q_len) {
    bf16_t* op = o + (size_t)(q_beg + my_q) * o_stride + (size_t)head * D;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        bf16x4 w;
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = f2bf(oacc[db][4 * gq + j] * inv_l);
        *reinterpret_cast<bf16x4*>(op + 32 * db + 8 * gq + 4 * hf) = w;
      }
    if (hf == 0 && lse)
      lse[(size_t)head * total_q + q_beg + my_q] = (l > 0.f) ? (m * 0.6931471805599453f + __logf(l)) : -INFINITY;
  }
}

// q [total_q, Hq, 128] (row stride q_stride), K cache [P, Hkv, 32, 128], V cache [P, Hkv, 128, 32], block_tables
// [B, max_pages] int32 (page of tokens 32 i .. 32 i + 31 of request b), cu_q / cu_k [B + 1]: query rows and context
// lengths (the context of request b INCLUDES its query rows: they are in the cache already). Every request needs
// q_len >= 1 and k_len >= q_len. lse may be null.
extern "C" int areal_flash_fwd_paged(const void* q, const void* kcache, const void* vcache, const int* block_tables,
                                     int max_pages, void* o, float* lse, const int* cu_q, const int* cu_k, int B,
                                     int max_q, int total_q, int Hq, int Hkv, int D, int64_t q_stride,
                                     int64_t o_stride, float softmax_scale, int causal, int fp8, hipStream_t st) {
  if (B == 0 || max_q == 0) return 0;
  if (D != 128) return -1;
  if (Hq % Hkv != 0) return -2;
  if ((q_stride | o_stride) % 8 != 0) return -3;
  if (((uintptr_t)q | (uintptr_t)kcache | (uintptr_t)vcache | (uintptr_t)o) % 16 != 0) return -4;
  const float scale_log2 = softmax_scale * 1.4426950408889634f;
  const int n_g_pad = (Hkv * B + 7) & ~7;
  dim3 grid((unsigned)(((max_q + 127) / 128) * (Hq / Hkv) * n_g_pad)), block(256);
  if (fp8)
    hipLaunchKernelGGL(flash_fwd_paged_kernel<true>, grid, block, 0, st, (const bf16_t*)q, kcache, vcache, block_tables,
                       max_pages, (bf16_t*)o, lse, cu_q, cu_k, q_stride, o_stride, Hq, Hkv, B, total_q, scale_log2, causal);
  else
    hipLaunchKernelGGL(flash_fwd_paged_kernel<false>, grid, block, 0, st, (const bf16_t*)q, kcache, vcache, block_tables,
                       max_pages, (bf16_t*)o, lse, cu_q, cu_k, q_stride, o_stride, Hq, Hkv, B, total_q, scale_log2, causal);
  return (int)hipGetLastError();
}
