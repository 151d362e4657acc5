// Paged-KV decode attention, fused RoPE + KV-cache write and cache gather for gfx950.
//
// This is the rollout engine's replacement for flash_attn_with_kvcache
// (realhf/impl/model/modules/attn.py:272) and for the SGLang/vLLM paged attention the
// reference delegates to. The cache layout is chosen for the MFMA operand maps so the
// decode kernel needs no LDS staging and no transposes:
//
//   K cache  [num_pages, Hkv, 32, D]   (key-major rows of D)
//   V cache  [num_pages, Hkv, D, 32]   (stored TRANSPOSED: for one d, 32 keys contiguous)
//
// A page holds 32 tokens. For one page and one kv head:
//   S^T[key, q]  = K[key, :] . Q[q, :]        A = K rows (16 B per lane straight from HBM)
//   O^T[d, q]   += V^T[d, key] * P^T[key, q]  A = V^T rows (16 B per lane straight from HBM)
// with v_mfma_f32_16x16x32_bf16. The 16 "q" columns are the query heads that share the kv
// head (GQA group, zero padded). The two 16-key S^T tiles of a page take the keys
// {8c+r} and {8c+4+r} (c = lane>>4, r = 0..3) so that each lane ends up holding P for keys
// 8c..8c+7, which is exactly the B-operand order of the P.V product.
#include "common.hpp"

#define PAGE 32

DEVINL f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// grid: (B, Hkv, nsplit); block: 256 threads = 4 waves. D = 64, 128 or 256 (template; 256 holds a page's 32 KiB of
// K and V fragments in registers and runs one workgroup per SIMD set instead of two).
// q: [B, Hq, D]; out: [B, Hq, D] when nsplit == 1 else partials.
// part_o: [B, Hq, nsplit, D] fp32, part_ml: [B, Hq, nsplit, 2] (m in log2 units, l).
// FP8: the caches hold e4m3 bytes (same layouts); a lane fetches 8 bytes per fragment and widens them to bf16 in
// registers - half the HBM traffic of the kernel, which is what bounds it at long contexts.
template <int D, bool FP8>
__global__ __launch_bounds__(256, (D > 128 ? 1 : 2)) void paged_decode_kernel(
    const bf16_t* __restrict__ q, const void* __restrict__ kcache,
    const void* __restrict__ vcache, const int* __restrict__ block_tables,
    const int* __restrict__ context_lens, bf16_t* __restrict__ out, float* __restrict__ part_o,
    float* __restrict__ part_ml, int Hq, int Hkv, int max_pages, float scale_log2, int q_stride) {
  constexpr int KS = D / 32;  // k-steps of the QK product
  constexpr int DB = D / 16;  // 16-wide d blocks of the PV product
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int b = blockIdx.x, h = blockIdx.y, split = blockIdx.z, nsplit = gridDim.z;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int G = Hq / Hkv;
  const int col = lane & 15, c = lane >> 4;
  const int ctx = context_lens[b];
  const int npages = (ctx + PAGE - 1) / PAGE;
  const int per = (npages + nsplit - 1) / nsplit;
  const int p_beg = split * per;
  const int p_end = min(npages, p_beg + per);

  // Q as the B operand of S^T = K.Q^T: lane (q = col, c) holds Q[q][32ks + 8c + j]
  bf16x8 qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    if (col < G) qf[ks] = ld8(q + (size_t)b * q_stride + (size_t)(h * G + col) * D + ks * 32 + c * 8);
    else
#pragma unroll
      for (int j = 0; j < 8; ++j) qf[ks][j] = (bf16_t)0.f;
  }
  f32x4 o[DB];
#pragma unroll
  for (int i = 0; i < DB; ++i) o[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m = -INFINITY, l = 0.f;  // running max (log2 units) is shared by the 4 lanes of a q

  const int* bt = block_tables + (size_t)b * max_pages;
  for (int p = p_beg + wid; p < p_end; p += 4) {
    const int page = bt[p];
    const size_t kp = ((size_t)page * Hkv + h) * PAGE * D;  // element offsets of this page's K / V block
    const size_t vp = ((size_t)page * Hkv + h) * D * PAGE;
    // issue all loads of the page up front (16 KiB per wave in flight)
    bf16x8 ka[KS], kb[KS], vt[DB];
    const int key_a = 8 * (col >> 2) + (col & 3);  // row `col` of tile a
    const int key_b = key_a + 4;                   // row `col` of tile b
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      ka[ks] = ld8_cache<FP8>(kcache, kp + key_a * D + ks * 32 + c * 8);
      kb[ks] = ld8_cache<FP8>(kcache, kp + key_b * D + ks * 32 + c * 8);
    }
#pragma unroll
    for (int i = 0; i < DB; ++i) vt[i] = ld8_cache<FP8>(vcache, vp + (i * 16 + col) * PAGE + c * 8);

    f32x4 sa = {0.f, 0.f, 0.f, 0.f}, sb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      sa = mfma16(ka[ks], qf[ks], sa);
      sb = mfma16(kb[ks], qf[ks], sb);
    }
    // lane (q = col, c): sa[r] = S[key 8c + r], sb[r] = S[key 8c + 4 + r]
    float s[8];
    const int key0 = p * PAGE + 8 * c;
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const float v = (r < 4 ? sa[r] : sb[r - 4]) * scale_log2;
      s[r] = (key0 + r < ctx) ? v : -INFINITY;
      mx = fmaxf(mx, s[r]);
    }
    mx = max_xor16(mx);
    mx = max_xor32(mx);
    const float mn = fmaxf(m, mx);
    if (__any(mn > m)) {
      const float alpha = (m == -INFINITY) ? 0.f : fast_exp2(m - mn);
      l *= alpha;
#pragma unroll
      for (int i = 0; i < DB; ++i) o[i] *= alpha;
      m = mn;
    }
    bf16x8 pf;
    float ls = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const float pr = (m == -INFINITY) ? 0.f : fast_exp2(s[r] - m);
      ls += pr;
      pf[r] = f2bf(pr);
    }
    l += ls;
    // O^T[d, q] += V^T[d, keys] . P^T[keys, q]
#pragma unroll
    for (int i = 0; i < DB; ++i) o[i] = mfma16(vt[i], pf, o[i]);
  }
  // l is a per-lane partial over this lane's keys: sum the 4 lanes that share a q
  l = sum_xor16(l);
  l = sum_xor32(l);

  // combine the 4 waves through LDS: per wave [16 q][D] fp32 O + [16] m + [16] l
  // Rows are DP = D + 4 floats apart: with a stride of D floats the 16 q rows of a store fall on the same banks
  // (a 16-way conflict); 4 floats of padding move each row by one 16-byte slot, and each lane stores its 4
  // consecutive values as one 16-byte write. Only the G real q rows are stored.
  constexpr int DP = D + 4;
  float* so = smem;                 // [4][16][DP]
  float* sm = smem + 4 * 16 * DP;   // [4][16]
  float* sl = sm + 4 * 16;          // [4][16]
  if (col < G) {
#pragma unroll
    for (int i = 0; i < DB; ++i) *reinterpret_cast<f32x4*>(so + (wid * 16 + col) * DP + i * 16 + 4 * c) = o[i];
  }
  if (c == 0) { sm[wid * 16 + col] = m; sl[wid * 16 + col] = l; }
  __syncthreads();
  // thread t handles (q = t / (D/8)... ) : 256 threads cover G*D elements in a strided loop
  for (int idx = threadIdx.x; idx < G * D; idx += 256) {
    const int qq = idx / D, d = idx % D;
    float mm = -INFINITY;
#pragma unroll
    for (int w = 0; w < 4; ++w) mm = fmaxf(mm, sm[w * 16 + qq]);
    float acc = 0.f, ll = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float mw = sm[w * 16 + qq];
      const float f = (mw == -INFINITY) ? 0.f : fast_exp2(mw - mm);
      acc += so[(w * 16 + qq) * DP + d] * f;
      ll += sl[w * 16 + qq] * f;
    }
    const int head = h * G + qq;
    if (nsplit == 1) {
      out[(size_t)b * Hq * D + (size_t)head * D + d] = f2bf(ll > 0.f ? acc / ll : 0.f);
    } else {
      part_o[(((size_t)b * Hq + head) * nsplit + split) * D + d] = acc;
      if (d == 0) {
        part_ml[(((size_t)b * Hq + head) * nsplit + split) * 2 + 0] = mm;
        part_ml[(((size_t)b * Hq + head) * nsplit + split) * 2 + 1] = ll;
      }
    }
  }
}

// Combine split partials: grid (B * Hq), block D threads. The (max, sum) pairs of the splits are fetched by one
// parallel load into LDS and the partial outputs four splits at a time: a loop of one dependent load per split is a
// chain of memory round trips (16 splits at a batch of 8 rows: 7.5 us for a few KB).
__global__ void paged_decode_reduce_kernel(const float* __restrict__ part_o,
                                           const float* __restrict__ part_ml,
                                           bf16_t* __restrict__ out, int nsplit, int D) {
  constexpr int MAXS = 256;
  __shared__ float sm[MAXS], sl[MAXS];
  const int bh = blockIdx.x, d = threadIdx.x;
  const bool staged = nsplit <= MAXS;
  if (staged) {
    for (int s = d; s < nsplit; s += blockDim.x) {
      const float2 ml = *reinterpret_cast<const float2*>(part_ml + ((size_t)bh * nsplit + s) * 2);
      sm[s] = ml.x;
      sl[s] = ml.y;
    }
    __syncthreads();
  }
  auto M = [&](int s) { return staged ? sm[s] : part_ml[((size_t)bh * nsplit + s) * 2]; };
  auto Lsum = [&](int s) { return staged ? sl[s] : part_ml[((size_t)bh * nsplit + s) * 2 + 1]; };
  float mm = -INFINITY;
  for (int s = 0; s < nsplit; ++s) mm = fmaxf(mm, M(s));
  float acc = 0.f, ll = 0.f;
  const float* po = part_o + (size_t)bh * nsplit * D + d;
  int s = 0;
  for (; s + 4 <= nsplit; s += 4) {
    const float o0 = po[(size_t)s * D], o1 = po[(size_t)(s + 1) * D];
    const float o2 = po[(size_t)(s + 2) * D], o3 = po[(size_t)(s + 3) * D];
    const float os[4] = {o0, o1, o2, o3};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float ms = M(s + j);
      const float f = (ms == -INFINITY) ? 0.f : fast_exp2(ms - mm);
      acc += os[j] * f;
      ll += Lsum(s + j) * f;
    }
  }
  for (; s < nsplit; ++s) {
    const float ms = M(s);
    const float f = (ms == -INFINITY) ? 0.f : fast_exp2(ms - mm);
    acc += po[(size_t)s * D] * f;
    ll += Lsum(s) * f;
  }
  out[(size_t)bh * D + d] = f2bf(ll > 0.f ? acc / ll : 0.f);
}

template <int D, bool FP8>
static int launch_paged_decode(const void* q, const void* kcache, const void* vcache, const int* block_tables,
                               const int* context_lens, void* out, float* part_o, float* part_ml, int B, int Hq, int Hkv,
                               int max_pages, int nsplit, float scale_log2, int q_stride, hipStream_t st) {
  const size_t lds = (size_t)(4 * 16 * (D + 4) + 2 * 4 * 16) * sizeof(float);
  if (D == 256) {
    static bool attr_set = false;  // 67 KiB of dynamic LDS: above the default limit
    if (!attr_set) {
      HIP_RET(hipFuncSetAttribute((const void*)paged_decode_kernel<D, FP8>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds));
      attr_set = true;
    }
  }
  hipLaunchKernelGGL((paged_decode_kernel<D, FP8>), dim3(B, Hkv, nsplit), dim3(256), lds, st, (const bf16_t*)q, kcache,
                     vcache, block_tables, context_lens, (bf16_t*)out, part_o, part_ml, Hq, Hkv, max_pages, scale_log2,
                     q_stride);
  return 0;
}

// fp8: the caches hold e4m3 bytes instead of bf16 values
extern "C" int areal_paged_decode(const void* q, const void* kcache, const void* vcache,
                                  const int* block_tables, const int* context_lens, void* out,
                                  float* part_o, float* part_ml, int B, int Hq, int Hkv, int D,
                                  int max_pages, int nsplit, float softmax_scale, int q_stride, int fp8,
                                  hipStream_t st) {
  if (B == 0) return 0;
  if (Hq % Hkv != 0 || Hq / Hkv > 16) return -1;
  if (D != 128 && D != 64 && D != 256) return -2;
  if (nsplit < 1 || (nsplit > 1 && (!part_o || !part_ml))) return -3;
  const float scale_log2 = softmax_scale * 1.4426950408889634f;
  int rc = 0;
#define PD(DD, F8)                                                                                                    \
  rc = launch_paged_decode<DD, F8>(q, kcache, vcache, block_tables, context_lens, out, part_o, part_ml, B, Hq, Hkv,   \
                                   max_pages, nsplit, scale_log2, q_stride, st)
  if (D == 256) { if (fp8) PD(256, true); else PD(256, false); }
  else if (D == 128) { if (fp8) PD(128, true); else PD(128, false); }
  else { if (fp8) PD(64, true); else PD(64, false); }
#undef PD
  if (rc) return rc;
  if (nsplit > 1)
    hipLaunchKernelGGL(paged_decode_reduce_kernel, dim3(B * Hq), dim3(D), 0, st, part_o, part_ml,
                       (bf16_t*)out, nsplit, D);
  return (int)hipGetLastError();
}

// ----------------------------------------------------------------------------------
// Fused RoPE + KV-cache write. qkv: [T, Hq + 2*Hkv, D] (q heads, k heads, v heads).
// Rotates q and k in place (half-split layout) and scatters k / v of every token into
// the paged caches at slot_mapping[t] (= page * 32 + offset; negative = skip the write).
// grid: T blocks; block: 256 threads.
// ----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rope_cache_write_kernel(
    bf16_t* __restrict__ qkv, const int* __restrict__ pos, const int64_t* __restrict__ slots,
    const float* __restrict__ cos_t, const float* __restrict__ sin_t, void* __restrict__ kcache,
    void* __restrict__ vcache, int Hq, int Hkv, int D, int max_pos, int do_rope, int fp8) {
  const int t = blockIdx.x;
  const int half = D >> 1;
  const int lph = half >> 3;  // lanes per head for the rotation
  int p = pos[t];
  p = p < 0 ? 0 : (p >= max_pos ? max_pos - 1 : p);
  bf16_t* row = qkv + (size_t)t * (Hq + 2 * Hkv) * D;
  const int64_t slot = slots ? slots[t] : -1;
  const int64_t page = slot >= 0 ? slot / PAGE : 0;
  const int off = slot >= 0 ? (int)(slot % PAGE) : 0;
  // phase 1: rotate q and k heads; k results go to the cache too
  const int nrot = (Hq + Hkv) * lph;
  for (int i = threadIdx.x; i < nrot; i += blockDim.x) {
    const int head = i / lph, sub = i % lph;
    bf16_t* base = row + (size_t)head * D + sub * 8;
    bf16x8 lo = ld8(base), hi = ld8(base + half);
    if (do_rope) {
      const float* cp = cos_t + (size_t)p * half + sub * 8;
      const float* sp = sin_t + (size_t)p * half + sub * 8;
      bf16x8 olo, ohi;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float a = bf2f(lo[j]), bb = bf2f(hi[j]);
        olo[j] = f2bf(a * cp[j] - bb * sp[j]);
        ohi[j] = f2bf(bb * cp[j] + a * sp[j]);
      }
      lo = olo; hi = ohi;
      st8(base, lo);
      st8(base + half, hi);
    }
    if (head >= Hq && slot >= 0) {
      const int kh = head - Hq;
      const size_t kd = (((size_t)page * Hkv + kh) * PAGE + off) * D + sub * 8;
      st8_cache(kcache, kd, lo, fp8);
      st8_cache(kcache, kd + half, hi, fp8);
    }
  }
  // phase 2: v heads -> transposed cache (scattered element stores, one page row apart)
  if (slot >= 0) {
    const bf16_t* vrow = row + (size_t)(Hq + Hkv) * D;
    for (int i = threadIdx.x; i < Hkv * D; i += blockDim.x) {
      const int vh = i / D, d = i % D;
      st1_cache(vcache, (((size_t)page * Hkv + vh) * D + d) * PAGE + off, bf2f(vrow[i]), fp8);
    }
  }
}

extern "C" int areal_rope_cache_write(void* qkv, const int* pos, const int64_t* slots,
                                      const float* cos_t, const float* sin_t, void* kcache,
                                      void* vcache, int T, int Hq, int Hkv, int D, int max_pos,
                                      int do_rope, int fp8, hipStream_t st) {
  if (T == 0) return 0;
  if (D % 16 != 0) return -1;
  hipLaunchKernelGGL(rope_cache_write_kernel, dim3(T), dim3(256), 0, st, (bf16_t*)qkv, pos, slots,
                     cos_t, sin_t, kcache, vcache, Hq, Hkv, D, max_pos, do_rope, fp8);
  return (int)hipGetLastError();
}

// Gather cached K/V of token slots back into contiguous [N, Hkv, D] tensors (used when a
// request extends a cached prefix: the prefix K/V are read from pages, the new tokens are
// appended, and the varlen flash kernel runs over the concatenation).
__global__ __launch_bounds__(256) void cache_gather_kernel(const void* __restrict__ kcache,
                                                           const void* __restrict__ vcache,
                                                           const int64_t* __restrict__ slots,
                                                           bf16_t* __restrict__ k_out,
                                                           bf16_t* __restrict__ v_out, int Hkv, int D,
                                                           int64_t out_stride, int fp8) {
  const int t = blockIdx.x;
  const int64_t slot = slots[t];
  const int64_t page = slot / PAGE;
  const int off = (int)(slot % PAGE);
  for (int i = threadIdx.x; i < Hkv * D; i += blockDim.x) {
    const int hh = i / D, d = i % D;
    const size_t ki = (((size_t)page * Hkv + hh) * PAGE + off) * D + d;
    const size_t vi = (((size_t)page * Hkv + hh) * D + d) * PAGE + off;
    if (fp8) {
      k_out[(size_t)t * out_stride + i] = f2bf(fp82f(reinterpret_cast<const fp8_t*>(kcache)[ki]));
      v_out[(size_t)t * out_stride + i] = f2bf(fp82f(reinterpret_cast<const fp8_t*>(vcache)[vi]));
    } else {
      k_out[(size_t)t * out_stride + i] = reinterpret_cast<const bf16_t*>(kcache)[ki];
      v_out[(size_t)t * out_stride + i] = reinterpret_cast<const bf16_t*>(vcache)[vi];
    }
  }
}

extern "C" int areal_cache_gather(const void* kcache, const void* vcache, const int64_t* slots,
                                  void* k_out, void* v_out, int N, int Hkv, int D, int64_t out_stride,
                                  int fp8, hipStream_t st) {
  if (N == 0) return 0;
  hipLaunchKernelGGL(cache_gather_kernel, dim3(N), dim3(256), 0, st, kcache, vcache, slots, (bf16_t*)k_out,
                     (bf16_t*)v_out, Hkv, D, out_stride, fp8);
  return (int)hipGetLastError();
}

// Copy whole pages (all layers share the same page ids): used for copy-on-write of a
// partially filled shared prefix page. kcache/vcache: [num_pages, page_elems].
__global__ void page_copy_kernel(bf16_t* __restrict__ cache, const int64_t* __restrict__ src,
                                 const int64_t* __restrict__ dst, int64_t page_elems) {
  const int64_t s = src[blockIdx.x], d = dst[blockIdx.x];
  for (int64_t i = threadIdx.x * 8; i < page_elems; i += (int64_t)blockDim.x * 8)
    st8(cache + d * page_elems + i, ld8(cache + s * page_elems + i));
}

extern "C" int areal_page_copy(void* cache, const int64_t* src, const int64_t* dst, int n,
                               int64_t page_elems, hipStream_t st) {
  if (n == 0) return 0;
  if (page_elems % 8 != 0) return -1;
  hipLaunchKernelGGL(page_copy_kernel, dim3(n), dim3(256), 0, st, (bf16_t*)cache, src, dst,
                     page_elems);
  return (int)hipGetLastError();
}

// Device-side bookkeeping between consecutive decode steps of one captured graph:
// feed the sampled token back, advance position / context length and derive the slot
// of the next token from the block table. Rows with active[b] == 0 are left untouched.
__global__ void advance_step_kernel(const int64_t* __restrict__ sampled, int64_t* __restrict__ input_ids,
                                    int* __restrict__ positions, int* __restrict__ context_lens,
                                    int64_t* __restrict__ slots, const int* __restrict__ block_tables,
                                    const int* __restrict__ active, int B, int max_pages) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  if (active && !active[b]) return;
  input_ids[b] = sampled[b];
  const int np = positions[b] + 1;
  positions[b] = np;
  context_lens[b] = np + 1;
  const int pg = np / PAGE;
  slots[b] = pg < max_pages ? (int64_t)block_tables[(size_t)b * max_pages + pg] * PAGE + np % PAGE : -1;
}

extern "C" int areal_advance_step(const int64_t* sampled, int64_t* input_ids, int* positions,
                                  int* context_lens, int64_t* slots, const int* block_tables,
                                  const int* active, int B, int max_pages, hipStream_t st) {
  if (B == 0) return 0;
  hipLaunchKernelGGL(advance_step_kernel, dim3((B + 255) / 256), dim3(256), 0, st, sampled, input_ids,
                     positions, context_lens, slots, block_tables, active, B, max_pages);
  return (int)hipGetLastError();
}


// ---------------------------------------------------------------------------------------------------
// Device-side bookkeeping of a multi-step decode plan with early exit.
//
// Per row: `active` (1 while the request still generates), `remaining` (tokens it may still produce:
// max_new_tokens and the context cap), `min_remaining` (tokens before stop ids count), `stop_ids`
// [B, n_stop] (-1 padded). After every sampled token the row either advances (next input token, position,
// context length, KV slot from the page table) or retires: no further KV writes (slot -1), attention over
// a single token (context length 1), sampler skipped. The host applies the same rules to the tokens it
// gets back, so both sides agree on where a request ends.
__global__ void decode_advance_kernel(const int64_t* __restrict__ sampled, int64_t* __restrict__ input_ids,
                                      int* __restrict__ positions, int* __restrict__ context_lens,
                                      int64_t* __restrict__ slots, const int* __restrict__ block_tables,
                                      int* __restrict__ active, int* __restrict__ remaining,
                                      int* __restrict__ min_remaining, const int* __restrict__ stop_ids,
                                      int n_stop, int B, int max_pages) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B || !active[b]) return;
  const int64_t tok = sampled[b];
  const int rem = remaining[b] - 1;
  const int mrem = min_remaining[b] > 0 ? min_remaining[b] - 1 : 0;
  remaining[b] = rem;
  min_remaining[b] = mrem;
  bool stop = rem <= 0;
  if (!stop && mrem == 0) {
    // stop ids count once min_new_tokens have been produced INCLUDING this token
    for (int j = 0; j < n_stop; ++j) stop |= stop_ids[(size_t)b * n_stop + j] == (int)tok;
  }
  if (stop) {
    active[b] = 0;
    slots[b] = -1;
    context_lens[b] = 1;
    return;
  }
  input_ids[b] = tok;
  const int np = positions[b] + 1;
  positions[b] = np;
  context_lens[b] = np + 1;
  const int pg = np / PAGE;
  slots[b] = pg < max_pages ? (int64_t)block_tables[(size_t)b * max_pages + pg] * PAGE + np % PAGE : -1;
}

// Start of a decode plan that continues the previous one on the same rows: rows whose request ended on
// the host (`kill`) retire, and live rows re-derive their KV slot from the refreshed page table.
__global__ void decode_begin_kernel(const int* __restrict__ kill, int* __restrict__ active,
                                    const int* __restrict__ positions, int* __restrict__ context_lens,
                                    int64_t* __restrict__ slots, const int* __restrict__ block_tables, int B,
                                    int max_pages) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  if (kill[b]) active[b] = 0;
  if (!active[b]) {
    slots[b] = -1;
    context_lens[b] = 1;
    return;
  }
  const int np = positions[b];
  const int pg = np / PAGE;
  slots[b] = pg < max_pages ? (int64_t)block_tables[(size_t)b * max_pages + pg] * PAGE + np % PAGE : -1;
}

extern "C" int areal_decode_advance(const int64_t* sampled, int64_t* input_ids, int* positions,
                                    int* context_lens, int64_t* slots, const int* block_tables, int* active,
                                    int* remaining, int* min_remaining, const int* stop_ids, int n_stop, int B,
                                    int max_pages, hipStream_t st) {
  if (B == 0) return 0;
  hipLaunchKernelGGL(decode_advance_kernel, dim3((B + 255) / 256), dim3(256), 0, st, sampled, input_ids,
                     positions, context_lens, slots, block_tables, active, remaining, min_remaining, stop_ids,
                     n_stop, B, max_pages);
  return (int)hipGetLastError();
}

extern "C" int areal_decode_begin(const int* kill, int* active, const int* positions, int* context_lens,
                                  int64_t* slots, const int* block_tables, int B, int max_pages,
                                  hipStream_t st) {
  if (B == 0) return 0;
  hipLaunchKernelGGL(decode_begin_kernel, dim3((B + 255) / 256), dim3(256), 0, st, kill, active, positions,
                     context_lens, slots, block_tables, B, max_pages);
  return (int)hipGetLastError();
}
