// Variable-length causal flash attention, forward and backward, for gfx950 (CDNA4).
//
// Owns what the reference borrows from flash-attn 2 (flash_attn_varlen_func at
// realhf/impl/model/modules/attn.py:289 and HF attn_implementation=flash_attention_2 via
// areal/engine/base_hf_engine.py:291-406): packed [T, H, D] q/k/v with cu_seqlens, GQA,
// causal masking aligned to the bottom-right corner when q and k lengths differ.
//
// Both kernels are built on v_mfma_f32_32x32x16_bf16 and arrange every product so that an
// accumulator tile feeds the next MFMA as an operand without leaving registers:
//   forward : S^T = K.Q^T (key in the accumulator rows, query on the lane) so the softmax
//             row statistics are lane-local; O^T += V^T.P^T takes P from the accumulator
//             registers as the B operand; V^T fragments come from ds_read_b64_tr_b16.
//   backward: S = Q.K^T and dP = dO.V^T keep the key on the lane, so P and dS are already
//             the B operands of dV^T += dO^T.P and dK^T += Q^T.dS; only dS crosses LDS once
//             (transposed) for dQ = dS.K, which is accumulated with fp32 atomics. From 2048 keys per sequence on
//             dQ is computed by a second, query-major kernel instead (forward-shaped, recomputes S and dP).
// LDS tiles use 256-byte rows with the 16-byte-chunk XOR swizzle
//   off(row, ch) = 256*row + 16*(ch ^ (((row&3)<<2) | ((row>>2)&3)))
// which is conflict-free for both ds_read_b128 row reads and transposed reads.
#include "flash_tiles.hpp"

// =====================================================================================
// Forward. 1-D grid over (q-tile rank, sequence, head). block: 256 threads = 4 waves x 32 q rows.
// KV tile = 64 keys. D = 128.
// =====================================================================================
__global__ __launch_bounds__(256, 2) void flash_fwd_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
    bf16_t* __restrict__ o, float* __restrict__ lse, const int* __restrict__ cu_q,
    const int* __restrict__ cu_k, int64_t q_stride, int64_t k_stride, int64_t v_stride,
    int64_t o_stride, int Hq, int Hkv, int nbatch, int total_q, float scale_log2, int causal) {
  constexpr int D = 128;
  __shared__ __attribute__((aligned(16))) char smem[2 * 64 * 256];
  char* ks = smem;
  char* vs = smem + 64 * 256;

  // 1-D grid, query-tile rank first: rank 0 is the LAST (heaviest under causal masking) tile of every
  // (sequence, head), so heavy workgroups of all sequences start first and light ones fill the tail.
  // XCD placement: consecutive workgroup ids go round-robin over the 8 XCDs, each with its own L2. Inside
  // a rank the slot of the (sequence, kv head) group runs fastest over an extent padded to a multiple of 8,
  // and group g sits in slot (g + rank) mod extent, so id % 8 == (g + rank) % 8: all query heads that read
  // the K/V of one group for one query tile run on the same XCD and share those tiles in its L2, while the
  // rotation by rank spreads a group's query tiles - and with them the padding slots and any remainder
  // of the group count modulo 8 - evenly over the XCDs (without it a single long sequence with two kv
  // heads would keep six of the eight XCDs idle).
  const int G = Hq / Hkv;
  const int n_g = Hkv * nbatch, n_g_pad = (n_g + 7) & ~7;
  const int per_rank = G * n_g_pad;
  const int rank = blockIdx.x / per_rank, rem = blockIdx.x - rank * per_rank;
  const int hj = rem / n_g_pad, slot = rem - hj * n_g_pad;
  int g = slot - rank % n_g_pad;
  g += g < 0 ? n_g_pad : 0;
  if (g >= n_g) return;
  const int b = g / Hkv, head = (g - b * Hkv) * G + hj;
  const int q_beg = cu_q[b], q_len = cu_q[b + 1] - q_beg;
  const int k_beg = cu_k[b], k_len = cu_k[b + 1] - k_beg;
  const int ntile_q = (q_len + 127) / 128;
  if (rank >= ntile_q) return;
  const int qt = ntile_q - 1 - rank;
  const int kvh = head / (Hq / Hkv);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = lane & 31, hf = lane >> 5, gl = lane & 15, grp = lane >> 4;
  const int shift = k_len - q_len;  // bottom-right aligned causal mask

  const int q0 = qt * 128 + wid * 32;  // first q row of this wave (within the sequence)
  const int my_q = q0 + r;
  const int my_q_c = my_q < q_len ? my_q : q_len - 1;
  // Q^T as the B operand: lane (r, hf) holds Q[r][16ks + 8hf + j]
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

  // number of kv tiles this block needs (the whole block walks the same range)
  int k_hi = k_len;
  if (causal) {
    const int last_q = min(q_len, qt * 128 + 128) - 1;
    k_hi = min(k_len, last_q + shift + 1);
  }
  const int ntile_k = k_hi > 0 ? (k_hi + 63) / 64 : 0;

  // cooperative tile load: thread -> (row = tid>>4 + 16*i, chunk = tid&15), i = 0..3
  const int lrow = tid >> 4, lch = tid & 15;
  bf16x8 kreg[4], vreg[4];
  auto load_tile = [&](int t) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int key = t * 64 + lrow + 16 * i;
      key = key < k_len ? key : k_len - 1;
      kreg[i] = ld8(k + (size_t)(k_beg + key) * k_stride + (size_t)kvh * D + lch * 8);
      vreg[i] = ld8(v + (size_t)(k_beg + key) * v_stride + (size_t)kvh * D + lch * 8);
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      lds_row_write(ks, lrow + 16 * i, lch, kreg[i]);
      lds_row_write(vs, lrow + 16 * i, lch, vreg[i]);
    }
  };
  if (ntile_k > 0) {
    load_tile(0);
    store_tile();
  }
  __syncthreads();
  // Every load issued so far (the Q fragments above all) has landed before the loop starts. Without this
  // the compiler carries "Q may still be in flight" around the loop and, because the counter is in-order,
  // makes each S^T MFMA wait for the NEXT tile's prefetch loads issued just before it: the global-load
  // latency it was meant to hide then sits in every iteration.
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)

  for (int t = 0; t < ntile_k; ++t) {
    if (t + 1 < ntile_k) load_tile(t + 1);
    // ---- S^T = K . Q^T : two 32-key blocks
    f32x16 sacc[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int j = 0; j < 16; ++j) sacc[kb][j] = 0.f;
#pragma unroll
      // fragments are read four at a time ahead of the MFMAs that use them, so that an MFMA waits for a
      // read issued four instructions earlier instead of for the one just before it
      for (int s0 = 0; s0 < 8; s0 += 4) {
        bf16x8 kf[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) kf[s] = lds_row_read(ks, 32 * kb + r, 2 * (s0 + s) + hf);
        __builtin_amdgcn_sched_barrier(0);  // keep the scheduler from sinking the reads back to their uses
#pragma unroll
        for (int s = 0; s < 4; ++s) sacc[kb] = mfma32(kf[s], qf[s0 + s], sacc[kb]);
      }
    }
    // ---- online softmax; lane holds q row `my_q`, keys (reg&3)+8(reg>>2)+4hf of each block
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
    const float alpha = (mn == -INFINITY) ? 1.f : fast_exp2(m - mn);  // m=-inf -> 0
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
    // ---- O^T[d, q] += V^T[d, keys] . P^T[keys, q]
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      // group `grp`: rows = keys 16s + 4hf (+8), cols = d 32db + 16(grp&1); all eight transposed reads of
      // a key slab are issued before its four MFMAs
      bf16x4 a0[4], a1[4];
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        const int col0 = 32 * db + 16 * (grp & 1);
        a0[db] = tr_read(vs, 16 * s + 4 * hf, col0, gl);
        a1[db] = tr_read(vs, 16 * s + 8 + 4 * hf, col0, gl);
      }
#pragma unroll
      for (int db = 0; db < 4; ++db) oacc[db] = mfma32(cat4(a0[db], a1[db]), pf[s], oacc[db]);
    }
    __syncthreads();
    if (t + 1 < ntile_k) store_tile();
    __syncthreads();
  }
  // ---- epilogue
  l = sum_xor32(l);
  const float inv_l = l > 0.f ? 1.f / l : 0.f;
  if (my_q < q_len) {
    bf16_t* op = o + (size_t)(q_beg + my_q) * o_stride + (size_t)head * D;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bf16x4 w;
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = f2bf(oacc[db][4 * g + j] * inv_l);
        *reinterpret_cast<bf16x4*>(op + 32 * db + 8 * g + 4 * hf) = w;
      }
    if (hf == 0 && lse)
      lse[(size_t)head * total_q + q_beg + my_q] =
          (l > 0.f) ? (m * 0.6931471805599453f + __logf(l)) : -INFINITY;
  }
}

// head size 256: flash_attn_d256.hip
extern "C" int areal_flash_fwd_d256(const void* q, const void* k, const void* v, void* o, float* lse,
                                    const int* cu_q, const int* cu_k, int B, int max_q, int total_q, int Hq, int Hkv,
                                    int64_t q_stride, int64_t k_stride, int64_t v_stride, int64_t o_stride,
                                    float softmax_scale, int causal, hipStream_t st);
extern "C" int areal_flash_bwd_d256(const void* q, const void* k, const void* v, const void* o, const void* d_o,
                                    const float* lse, float* delta, void* dq, void* dk, void* dv, const int* cu_q,
                                    const int* cu_k, int B, int max_k, int total_q, int Hq, int Hkv, int64_t q_stride,
                                    int64_t k_stride, int64_t v_stride, int64_t o_stride, int64_t do_stride,
                                    int64_t dq_stride, int64_t dk_stride, int64_t dv_stride, float softmax_scale,
                                    int causal, hipStream_t st);

extern "C" int areal_flash_fwd(const void* q, const void* k, const void* v, void* o, float* lse,
                               const int* cu_q, const int* cu_k, int B, int max_q, int total_q,
                               int Hq, int Hkv, int D, int64_t q_stride, int64_t k_stride,
                               int64_t v_stride, int64_t o_stride, float softmax_scale, int causal,
                               hipStream_t st) {
  if (B == 0 || max_q == 0) return 0;
  if (D != 128 && D != 256) return -1;
  if (Hq % Hkv != 0) return -2;
  if ((q_stride | k_stride | v_stride | o_stride) % 8 != 0) return -3;
  if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o) % 16 != 0) return -4;
  if (D == 256)
    return areal_flash_fwd_d256(q, k, v, o, lse, cu_q, cu_k, B, max_q, total_q, Hq, Hkv, q_stride, k_stride, v_stride,
                                o_stride, softmax_scale, causal, st);
  const int n_g_pad = (Hkv * B + 7) & ~7;  // see the XCD placement note in the kernel
  dim3 grid((unsigned)(((max_q + 127) / 128) * (Hq / Hkv) * n_g_pad)), block(256);
  hipLaunchKernelGGL(flash_fwd_kernel, grid, block, 0, st, (const bf16_t*)q, (const bf16_t*)k,
                     (const bf16_t*)v, (bf16_t*)o, lse, cu_q, cu_k, q_stride, k_stride, v_stride,
                     o_stride, Hq, Hkv, B, total_q, softmax_scale * 1.4426950408889634f, causal);
  return (int)hipGetLastError();
}

// =====================================================================================
// Backward.
//  pre : delta[h, t] = sum_d dO[t, h, d] * O[t, h, d]
//  main: 1-D grid over (key block of 128, sequence, kv head); block 256 threads = 4 waves x 32 keys. Each wave
//        keeps its K / V fragments and the dK^T / dV^T accumulators in registers and sweeps
//        the query heads of the GQA group x 32-row query slices. dQ goes out through fp32
//        atomics into dq_acc [Tq, Hq, D].
// =====================================================================================
__global__ __launch_bounds__(256) void flash_bwd_delta_kernel(
    const bf16_t* __restrict__ o, const bf16_t* __restrict__ d_o, float* __restrict__ delta,
    int64_t o_stride, int64_t do_stride, int Hq, int total_q) {
  constexpr int D = 128;
  // 16 lanes per (token, head)
  const int64_t idx = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const int sub = threadIdx.x & 15;
  if (idx >= (int64_t)total_q * Hq) return;
  const int64_t t = idx / Hq;
  const int h = (int)(idx % Hq);
  bf16x8 a = ld8(o + t * o_stride + (size_t)h * D + sub * 8);
  bf16x8 bb = ld8(d_o + t * do_stride + (size_t)h * D + sub * 8);
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) s += bf2f(a[j]) * bf2f(bb[j]);
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if (sub == 0) delta[(size_t)h * total_q + t] = s;
}

// NS = number of 32-row query slices per loop iteration. One slice per iteration keeps a wave's MFMA chain
// (~1.3k cycles) shorter than the fixed costs around it (two barriers, the LDS round trip of dS, the wait for the
// prefetched Q / dO rows); with two slices the chain doubles while those costs stay, and the scheduler can overlap
// the LDS reads of one slice with the MFMAs of the other. The kernel runs one workgroup per CU (dK / dV
// accumulators + V fragments hold 192 registers per lane), so the extra temporaries are free.
// WITH_DQ = false leaves dQ to flash_bwd_dq_kernel (long sequences, see areal_flash_bwd): no dS exchange through
// LDS, no third barrier, no atomics.
template <int NS, bool WITH_DQ>
__global__ __launch_bounds__(256, 1) void flash_bwd_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
    const bf16_t* __restrict__ d_o, const float* __restrict__ lse, const float* __restrict__ delta,
    float* __restrict__ dq_acc, bf16_t* __restrict__ dk, bf16_t* __restrict__ dv,
    const int* __restrict__ cu_q, const int* __restrict__ cu_k, int64_t q_stride, int64_t k_stride,
    int64_t v_stride, int64_t do_stride, int64_t dk_stride, int64_t dv_stride, int Hq, int Hkv,
    int nbatch, int total_q, float softmax_scale, int causal) {
  constexpr int D = 128;
  constexpr int QR = 32 * NS;  // query rows per iteration
  // LDS: Q tile [QR][256B] | dO tile [QR][256B] | K tile [128][256B] | dS^T [128 keys][QR q] bf16
  //      | lse[QR] f32 | delta[QR] f32. dS^T rows are padded by 8 bytes: with a power-of-two stride the
  //      8-byte writes of lanes r and r+4 land in the same banks (8-way conflict), the padding spreads the
  //      32 rows of a wave over all 64 banks.
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* qs = smem;
  char* dos = qs + QR * 256;
  char* kts = dos + QR * 256;
  constexpr int DSS = 2 * QR + 8;        // dS^T row stride in bytes
  char* dss = kts + 128 * 256;           // [128][DSS]
  float* lse_s = reinterpret_cast<float*>(dss + 128 * DSS);
  float* dl_s = lse_s + QR;

  // 1-D grid ordered by key block first: under causal masking the work of a workgroup falls with its
  // key-block index (later keys are seen by fewer queries), so the heaviest workgroups of ALL sequences
  // are dispatched first and the light ones fill the tail (longest-processing-time-first scheduling;
  // a (block, head, sequence) grid leaves CUs idle behind each sequence's heavy blocks).
  // (No XCD grouping here, unlike the forward: pinning all key blocks of a group to one XCD cut HBM reads by
  // 30 % but cost 5 % run time on 50 groups, because 50 groups do not spread evenly over 8 XCDs and this
  // kernel runs one workgroup per CU.)
  const int n_bh = Hkv * nbatch;
  const int bid = blockIdx.x;
  const int kblk = bid / n_bh, bh = bid - kblk * n_bh;
  const int b = bh / Hkv, kvh = bh - b * Hkv;
  const int q_beg = cu_q[b], q_len = cu_q[b + 1] - q_beg;
  const int k_beg = cu_k[b], k_len = cu_k[b + 1] - k_beg;
  const int kb0 = kblk * 128;
  if (kb0 >= k_len) return;
  const int G = Hq / Hkv;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = lane & 31, hf = lane >> 5, gl = lane & 15, grp = lane >> 4;
  const int shift = k_len - q_len;
  const float scale_log2 = softmax_scale * 1.4426950408889634f;

  // this wave's keys: kb0 + 32*wid + r
  const int my_key = kb0 + 32 * wid + r;
  const int my_key_c = my_key < k_len ? my_key : k_len - 1;
  bf16x8 kf[8], vf[8];
  {
    const bf16_t* kp = k + (size_t)(k_beg + my_key_c) * k_stride + (size_t)kvh * D + hf * 8;
    const bf16_t* vp = v + (size_t)(k_beg + my_key_c) * v_stride + (size_t)kvh * D + hf * 8;
#pragma unroll
    for (int s = 0; s < 8; ++s) { kf[s] = ld8(kp + s * 16); vf[s] = ld8(vp + s * 16); }
  }
  // K tile of the block into LDS (for the transposed reads of dQ = dS.K)
  if (WITH_DQ) {
    const int lrow = tid >> 4, lch = tid & 15;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      int key = kb0 + lrow + 16 * i;
      key = key < k_len ? key : k_len - 1;
      lds_row_write(kts, lrow + 16 * i, lch,
                    ld8(k + (size_t)(k_beg + key) * k_stride + (size_t)kvh * D + lch * 8));
    }
  }
  f32x16 dkacc[4], dvacc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) { dkacc[i][j] = 0.f; dvacc[i][j] = 0.f; }

  // first query row that can see this key block
  int qs_first = 0;
  if (causal) {
    const int fq = kb0 - shift;  // q index with q + shift >= kb0
    qs_first = fq > 0 ? (fq / QR) * QR : 0;
  }
  const int lrow = tid >> 4, lch = tid & 15;  // 16 rows x 16 chunks per pass, 2 * NS passes for QR rows

  // Work list: (query head of the GQA group) x (QR-row query slices that can see this key block),
  // walked as one flat loop so that the loads of the next slice - also across a head boundary - are
  // in flight while the current slice is computed (a global load takes ~2-4k cycles: without the prefetch
  // every slice exposes the full latency).
  const int nslice = q_len > qs_first ? (q_len - qs_first + QR - 1) / QR : 0;
  const int nwork = G * nslice;
  bf16x8 qreg[2 * NS], doreg[2 * NS];
  float lse_r = INFINITY, dl_r = 0.f;
  auto prefetch = [&](int it) {
    const int head = kvh * G + it / nslice;
    const int qs0 = qs_first + QR * (it % nslice);
#pragma unroll
    for (int i = 0; i < 2 * NS; ++i) {
      int qi = qs0 + lrow + 16 * i;
      qi = qi < q_len ? qi : q_len - 1;
      qreg[i] = ld8(q + (size_t)(q_beg + qi) * q_stride + (size_t)head * D + lch * 8);
      doreg[i] = ld8(d_o + (size_t)(q_beg + qi) * do_stride + (size_t)head * D + lch * 8);
    }
    if (tid < QR) {
      int qi = qs0 + tid;
      const bool ok = qi < q_len;
      qi = ok ? qi : q_len - 1;
      // rows past the end get lse = +inf so that p = exp2(-inf) = 0
      // (scaled to base 2 when staged: arithmetic here would wait for the load just issued)
      lse_r = ok ? lse[(size_t)head * total_q + q_beg + qi] : INFINITY;
      dl_r = ok ? delta[(size_t)head * total_q + q_beg + qi] : 0.f;
    }
  };
  if (nwork > 0) prefetch(0);
  // All loads issued so far (the K / V fragments of this wave) have landed before the loop: otherwise the
  // compiler keeps them "possibly in flight" around the loop and, the counter being in-order, makes the
  // S / dP MFMAs that read them wait for the prefetch of the NEXT slice issued just before.
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)

  for (int it = 0; it < nwork; ++it) {
    const int head = kvh * G + it / nslice;
    const int qs0 = qs_first + QR * (it % nslice);
    {
      __syncthreads();  // previous slice fully consumed (Q/dO/dS tiles)
      // ---- stage the prefetched Q and dO slices + row constants, then start the next loads
#pragma unroll
      for (int i = 0; i < 2 * NS; ++i) {
        lds_row_write(qs, lrow + 16 * i, lch, qreg[i]);
        lds_row_write(dos, lrow + 16 * i, lch, doreg[i]);
      }
      if (tid < QR) {
        lse_s[tid] = lse_r * 1.4426950408889634f;
        dl_s[tid] = dl_r;
      }
      if (it + 1 < nwork) prefetch(it + 1);
      __syncthreads();
      bf16x8 pfr[NS][2], dsfr[NS][2];
#pragma unroll
      for (int u = 0; u < NS; ++u) {
        const int qu = qs0 + 32 * u;  // first query row of this sub-slice
        // ---- S = Q.K^T, dP = dO.V^T (rows = q in registers, col = key on the lane)
        f32x16 sacc, pacc;
#pragma unroll
        for (int j = 0; j < 16; ++j) { sacc[j] = 0.f; pacc[j] = 0.f; }
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          bf16x8 qa = lds_row_read(qs, 32 * u + r, 2 * s + hf);
          bf16x8 da = lds_row_read(dos, 32 * u + r, 2 * s + hf);
          sacc = mfma32(qa, kf[s], sacc);
          pacc = mfma32(da, vf[s], pacc);
        }
        // ---- P = exp2(S*c - lse), dS = P * (dP - delta). Only slices that touch the end of the key range or
        //      the causal diagonal need per-element masking (wave-uniform test).
        const bool need_mask = (kb0 + 128 > k_len) || (causal && (kb0 + 127 > qu + shift));
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse_s + 32 * u + 8 * g4 + 4 * hf);
          const f32x4 d4 = *reinterpret_cast<const f32x4*>(dl_s + 32 * u + 8 * g4 + 4 * hf);
          bf16x4 dsw;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int reg = 4 * g4 + j;
            const int qrow = qu + 8 * g4 + 4 * hf + j;
            float sv = sacc[reg] * scale_log2 - l4[j];  // lse = +inf on rows past the end -> p = 0
            if (need_mask && (my_key >= k_len || (causal && my_key > qrow + shift))) sv = -INFINITY;
            const float p = fast_exp2(sv);
            const float ds = p * (pacc[reg] - d4[j]);
            pfr[u][reg >> 3][reg & 7] = f2bf(p);
            dsfr[u][reg >> 3][reg & 7] = f2bf(ds);
            dsw[j] = f2bf(ds);
          }
          // dS^T[key][q]: this lane's key row, 4 consecutive q -> 8 bytes
          if (WITH_DQ) *reinterpret_cast<bf16x4*>(dss + (32 * wid + r) * DSS + (32 * u + 8 * g4 + 4 * hf) * 2) = dsw;
        }
      }
      // ---- dV^T += dO^T.P ; dK^T += Q^T.dS   (A operand from transposed reads)
#pragma unroll
      for (int u = 0; u < NS; ++u) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
          for (int db = 0; db < 4; ++db) {
            const int col0 = 32 * db + 16 * (grp & 1);
            const int row0 = 32 * u + 16 * s;
            bf16x8 da = cat4(tr_read(dos, row0 + 4 * hf, col0, gl), tr_read(dos, row0 + 8 + 4 * hf, col0, gl));
            dvacc[db] = mfma32(da, pfr[u][s], dvacc[db]);
            bf16x8 qa = cat4(tr_read(qs, row0 + 4 * hf, col0, gl), tr_read(qs, row0 + 8 + 4 * hf, col0, gl));
            dkacc[db] = mfma32(qa, dsfr[u][s], dkacc[db]);
          }
        }
      }
      if (WITH_DQ) __syncthreads();  // dS^T of all four waves visible
#pragma unroll
      for (int u = 0; u < (WITH_DQ ? NS : 0); ++u) {
        const int qu = qs0 + 32 * u;
        if (qu >= q_len) break;  // uniform: the whole sub-slice lies past the sequence
        // ---- dQ[32 q x 32 d (block wid)] = dS[32 q x 128 keys] . K[128 keys x 32 d]
        f32x16 dqacc;
#pragma unroll
        for (int j = 0; j < 16; ++j) dqacc[j] = 0.f;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          // A = dS: lane (q = r, hf) needs keys 16s + 8hf + j  -> transposed reads of dS^T
          //   group rows = keys, cols = q block 16(grp&1)
          const char* a0p = dss + (16 * s + 8 * hf + (gl >> 2)) * DSS + (32 * u + 16 * (grp & 1) + 4 * (gl & 3)) * 2;
          bf16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(a0p));
          bf16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(a0p + 4 * DSS));
          // B = K: lane (d = r, hf) needs K[key 16s + 8hf + j][32 wid + r]
          const int col0 = 32 * wid + 16 * (grp & 1);
          bf16x8 kb = cat4(tr_read(kts, 16 * s + 8 * hf, col0, gl),
                           tr_read(kts, 16 * s + 8 * hf + 4, col0, gl));
          dqacc = mfma32(cat4(a0, a1), kb, dqacc);
        }
        // atomics: lane (d col = r, hf), reg -> q row
        float* dq_row = dq_acc + ((size_t)(q_beg + qu + 4 * hf) * Hq + head) * D + 32 * wid + r;
        if (qu + 32 <= q_len) {  // whole slice inside the sequence (wave-uniform): no per-row guards
#pragma unroll
          for (int reg = 0; reg < 16; ++reg)
            atomicAdd(dq_row + (size_t)((reg & 3) + 8 * (reg >> 2)) * Hq * D, dqacc[reg] * softmax_scale);
        } else {
#pragma unroll
          for (int reg = 0; reg < 16; ++reg) {
            const int dr = (reg & 3) + 8 * (reg >> 2);
            if (qu + 4 * hf + dr < q_len) atomicAdd(dq_row + (size_t)dr * Hq * D, dqacc[reg] * softmax_scale);
          }
        }
      }
    }
  }
  // ---- epilogue: dK^T / dV^T accumulators: lane (key col = r, hf), reg -> d row
  if (my_key < k_len) {
    bf16_t* dkp = dk + (size_t)(k_beg + my_key) * dk_stride + (size_t)kvh * D;
    bf16_t* dvp = dv + (size_t)(k_beg + my_key) * dv_stride + (size_t)kvh * D;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        bf16x4 wk, wv;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          wk[j] = f2bf(dkacc[db][4 * g4 + j] * softmax_scale);
          wv[j] = f2bf(dvacc[db][4 * g4 + j]);
        }
        *reinterpret_cast<bf16x4*>(dkp + 32 * db + 8 * g4 + 4 * hf) = wk;
        *reinterpret_cast<bf16x4*>(dvp + 32 * db + 8 * g4 + 4 * hf) = wv;
      }
  }
}

// dQ on its own, one workgroup per 128-row query tile (the grid and XCD placement of the forward): for long
// sequences the fp32 atomics of the fused kernel - every dQ element receives one add per 128-key block, and the
// accumulator no longer fits any cache - cost more than recomputing S and dP here. Same operand arrangement as
// the forward: S^T = K.Q^T and dP^T = V.dO^T keep the query on the lane (lse and delta are per-lane scalars), dS^T
// comes out of the accumulators as the B operand of dQ^T += K^T.dS^T, K^T fragments are transposed LDS reads.
// dQ is written once, in bf16, with the softmax scale applied.
__global__ __launch_bounds__(256, 1) void flash_bwd_dq_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
    const bf16_t* __restrict__ d_o, const float* __restrict__ lse, const float* __restrict__ delta,
    bf16_t* __restrict__ dq, const int* __restrict__ cu_q, const int* __restrict__ cu_k, int64_t q_stride,
    int64_t k_stride, int64_t v_stride, int64_t do_stride, int64_t dq_stride, int Hq, int Hkv, int nbatch,
    int total_q, float softmax_scale, int causal) {
  constexpr int D = 128;
  __shared__ __attribute__((aligned(16))) char smem[2 * 64 * 256];
  char* ks = smem;
  char* vs = smem + 64 * 256;
  const int G = Hq / Hkv;
  const int n_g = Hkv * nbatch, n_g_pad = (n_g + 7) & ~7;
  const int per_rank = G * n_g_pad;
  const int rank = blockIdx.x / per_rank, rem = blockIdx.x - rank * per_rank;
  const int hj = rem / n_g_pad, slot = rem - hj * n_g_pad;
  int g = slot - rank % n_g_pad;
  g += g < 0 ? n_g_pad : 0;
  if (g >= n_g) return;
  const int b = g / Hkv, head = (g - b * Hkv) * G + hj;
  const int q_beg = cu_q[b], q_len = cu_q[b + 1] - q_beg;
  const int k_beg = cu_k[b], k_len = cu_k[b + 1] - k_beg;
  const int ntile_q = (q_len + 127) / 128;
  if (rank >= ntile_q) return;
  const int qt = ntile_q - 1 - rank;
  const int kvh = head / G;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = lane & 31, hf = lane >> 5, gl = lane & 15, grp = lane >> 4;
  const int shift = k_len - q_len;
  const float scale_log2 = softmax_scale * 1.4426950408889634f;

  const int q0 = qt * 128 + wid * 32;
  const int my_q = q0 + r;
  const bool q_ok = my_q < q_len;
  const int my_q_c = q_ok ? my_q : q_len - 1;
  bf16x8 qf[8], dof[8];
  {
    const bf16_t* qp = q + (size_t)(q_beg + my_q_c) * q_stride + (size_t)head * D + hf * 8;
    const bf16_t* dp = d_o + (size_t)(q_beg + my_q_c) * do_stride + (size_t)head * D + hf * 8;
#pragma unroll
    for (int s = 0; s < 8; ++s) { qf[s] = ld8(qp + s * 16); dof[s] = ld8(dp + s * 16); }
  }
  // rows past the end: lse = +inf makes every probability zero
  const float lse2 = q_ok ? lse[(size_t)head * total_q + q_beg + my_q] * 1.4426950408889634f : INFINITY;
  const float dlt = q_ok ? delta[(size_t)head * total_q + q_beg + my_q] : 0.f;
  f32x16 dqacc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) dqacc[i][j] = 0.f;

  int k_hi = k_len;
  if (causal) {
    const int last_q = min(q_len, qt * 128 + 128) - 1;
    k_hi = min(k_len, last_q + shift + 1);
  }
  const int ntile_k = k_hi > 0 ? (k_hi + 63) / 64 : 0;
  const int lrow = tid >> 4, lch = tid & 15;
  bf16x8 kreg[4], vreg[4];
  auto load_tile = [&](int t) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int key = t * 64 + lrow + 16 * i;
      key = key < k_len ? key : k_len - 1;
      kreg[i] = ld8(k + (size_t)(k_beg + key) * k_stride + (size_t)kvh * D + lch * 8);
      vreg[i] = ld8(v + (size_t)(k_beg + key) * v_stride + (size_t)kvh * D + lch * 8);
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      lds_row_write(ks, lrow + 16 * i, lch, kreg[i]);
      lds_row_write(vs, lrow + 16 * i, lch, vreg[i]);
    }
  };
  if (ntile_k > 0) {
    load_tile(0);
    store_tile();
  }
  __syncthreads();
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): see the note in the forward kernel

  for (int t = 0; t < ntile_k; ++t) {
    if (t + 1 < ntile_k) load_tile(t + 1);
    const int kbase = t * 64;
    const bool need_mask = (kbase + 64 > k_len) || (causal && (kbase + 63 > q0 + shift));
    bf16x8 dsf[4];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      // ---- S^T = K.Q^T and dP^T = V.dO^T for one block of 32 keys
      f32x16 sacc, pacc;
#pragma unroll
      for (int j = 0; j < 16; ++j) { sacc[j] = 0.f; pacc[j] = 0.f; }
#pragma unroll
      for (int s0 = 0; s0 < 8; s0 += 4) {
        bf16x8 kf[4], vf[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          kf[s] = lds_row_read(ks, 32 * kb + r, 2 * (s0 + s) + hf);
          vf[s] = lds_row_read(vs, 32 * kb + r, 2 * (s0 + s) + hf);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          sacc = mfma32(kf[s], qf[s0 + s], sacc);
          pacc = mfma32(vf[s], dof[s0 + s], pacc);
        }
      }
      // ---- P^T = exp2(S^T c - lse), dS^T = P^T (dP^T - delta); lane = q row, reg -> key
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        float sv = sacc[j] * scale_log2 - lse2;
        if (need_mask) {
          const int key = kbase + 32 * kb + (j & 3) + 8 * (j >> 2) + 4 * hf;
          if (key >= k_len || (causal && key > my_q + shift)) sv = -INFINITY;
        }
        const float pr = fast_exp2(sv);
        dsf[2 * kb + (j >> 3)][j & 7] = f2bf(pr * (pacc[j] - dlt));
      }
    }
    // ---- dQ^T[d, q] += K^T[d, keys] . dS^T[keys, q]
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      bf16x4 a0[4], a1[4];
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        const int col0 = 32 * db + 16 * (grp & 1);
        a0[db] = tr_read(ks, 16 * s + 4 * hf, col0, gl);
        a1[db] = tr_read(ks, 16 * s + 8 + 4 * hf, col0, gl);
      }
#pragma unroll
      for (int db = 0; db < 4; ++db) dqacc[db] = mfma32(cat4(a0[db], a1[db]), dsf[s], dqacc[db]);
    }
    __syncthreads();
    if (t + 1 < ntile_k) store_tile();
    __syncthreads();
  }
  if (q_ok) {
    bf16_t* op = dq + (size_t)(q_beg + my_q) * dq_stride + (size_t)head * D;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        bf16x4 w;
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = f2bf(dqacc[db][4 * g4 + j] * softmax_scale);
        *reinterpret_cast<bf16x4*>(op + 32 * db + 8 * g4 + 4 * hf) = w;
      }
  }
}

// dq (bf16, strided) = dq_acc (fp32, [T, Hq, D])
__global__ __launch_bounds__(256) void flash_bwd_dq_cast_kernel(const float* __restrict__ acc,
                                                                bf16_t* __restrict__ dq,
                                                                int64_t dq_stride, int Hq,
                                                                int64_t total_q) {
  constexpr int D = 128;
  const int64_t nvec = total_q * Hq * (D / 8);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % (D / 8));
    const int64_t th = i / (D / 8);
    const int h = (int)(th % Hq);
    const int64_t t = th / Hq;
    const float* p = acc + (th * D) + c * 8;
    f32x4 a = *reinterpret_cast<const f32x4*>(p), bq = *reinterpret_cast<const f32x4*>(p + 4);
    bf16x8 w;
#pragma unroll
    for (int j = 0; j < 4; ++j) { w[j] = f2bf(a[j]); w[j + 4] = f2bf(bq[j]); }
    st8(dq + t * dq_stride + (size_t)h * D + c * 8, w);
  }
}

// dq_acc must be zeroed by the caller ([total_q, Hq, D] fp32); delta: [Hq, total_q] fp32.
// Does a backward over sequences of up to `max_k` keys give dQ a kernel of its own (no atomics, no fp32 dq_acc)?
// Measured on MI355X with tools/attn_probe.py (28 q / 4 kv heads, fused -> split backward): 8 x 2048 tokens 1.66 -> 1.45 ms,
// 4 x 4096 3.08 -> 2.57 ms, 24576 28.1 -> 22.7 ms; 25 sequences of ~1150 tokens (12 / 2 heads) 0.86 -> 0.95 ms, so short
// sequences stay on the fused kernel. AREAL_FLASH_BWD_SPLIT_MIN_K overrides the threshold (0 = never split).
extern "C" int areal_flash_bwd_split(int max_k) {
  static const int split_min_k = [] {
    const char* e = getenv("AREAL_FLASH_BWD_SPLIT_MIN_K");
    return e ? atoi(e) : 2048;
  }();
  return split_min_k > 0 && max_k >= split_min_k;
}

extern "C" int areal_flash_bwd(const void* q, const void* k, const void* v, const void* o,
                               const void* d_o, const float* lse, float* delta, float* dq_acc,
                               void* dq, void* dk, void* dv, const int* cu_q, const int* cu_k,
                               int B, int max_k, int total_q, int Hq, int Hkv, int D,
                               int64_t q_stride, int64_t k_stride, int64_t v_stride,
                               int64_t o_stride, int64_t do_stride, int64_t dq_stride,
                               int64_t dk_stride, int64_t dv_stride, float softmax_scale, int causal,
                               hipStream_t st) {
  if (B == 0 || total_q == 0) return 0;
  if (D != 128 && D != 256) return -1;
  if (Hq % Hkv != 0) return -2;
  if ((q_stride | k_stride | v_stride | o_stride | do_stride | dq_stride | dk_stride | dv_stride) % 8)
    return -3;
  if (D == 256)  // always the two-kernel form (dq_acc is not used)
    return areal_flash_bwd_d256(q, k, v, o, d_o, lse, delta, dq, dk, dv, cu_q, cu_k, B, max_k, total_q, Hq, Hkv,
                                q_stride, k_stride, v_stride, o_stride, do_stride, dq_stride, dk_stride, dv_stride,
                                softmax_scale, causal, st);
  {
    const int64_t n = (int64_t)total_q * Hq * 16;
    hipLaunchKernelGGL(flash_bwd_delta_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                       (const bf16_t*)o, (const bf16_t*)d_o, delta, o_stride, do_stride, Hq, total_q);
  }
  // query slices per loop iteration of the main kernel (AREAL_FLASH_BWD_SLICES=1 selects the one-slice form)
  static const int ns = [] {
    const char* e = getenv("AREAL_FLASH_BWD_SLICES");
    return (e && e[0] == '1') ? 1 : 2;
  }();
  const bool split = areal_flash_bwd_split(max_k) != 0;
  if (!split && dq_acc == nullptr) return -5;
  const int qr = 32 * ns;
  const size_t lds = (size_t)qr * 256 * 2 + 128 * 256 + 128 * (2 * qr + 8) + 2 * qr * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    const int l1 = (int)(32 * 256 * 2 + 128 * 256 + 128 * 72 + 64 * sizeof(float));
    const int l2 = (int)(64 * 256 * 2 + 128 * 256 + 128 * 136 + 128 * sizeof(float));
    HIP_RET(hipFuncSetAttribute((const void*)flash_bwd_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, l1));
    HIP_RET(hipFuncSetAttribute((const void*)flash_bwd_kernel<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, l2));
    HIP_RET(hipFuncSetAttribute((const void*)flash_bwd_kernel<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, l1));
    HIP_RET(hipFuncSetAttribute((const void*)flash_bwd_kernel<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, l2));
    attr_set = true;
  }
  dim3 grid((unsigned)(((max_k + 127) / 128) * Hkv * B)), block(256);
#define AREAL_LAUNCH_BWD(NS_, DQ_)                                                                              \
  hipLaunchKernelGGL((flash_bwd_kernel<NS_, DQ_>), grid, block, lds, st, (const bf16_t*)q, (const bf16_t*)k,   \
                     (const bf16_t*)v, (const bf16_t*)d_o, lse, delta, dq_acc, (bf16_t*)dk, (bf16_t*)dv, cu_q, \
                     cu_k, q_stride, k_stride, v_stride, do_stride, dk_stride, dv_stride, Hq, Hkv, B, total_q, \
                     softmax_scale, causal)
  if (split) {
    if (ns == 1) AREAL_LAUNCH_BWD(1, false); else AREAL_LAUNCH_BWD(2, false);
    const int n_g_pad = (Hkv * B + 7) & ~7;
    // query-tile ranks: no sequence has more than total_q rows (workgroups past a sequence's last tile exit at once;
    // the split only runs for long sequences, so there are few sequences and few such workgroups)
    dim3 qgrid((unsigned)(((total_q + 127) / 128) * (Hq / Hkv) * n_g_pad));
    hipLaunchKernelGGL(flash_bwd_dq_kernel, qgrid, block, 0, st, (const bf16_t*)q, (const bf16_t*)k,
                       (const bf16_t*)v, (const bf16_t*)d_o, lse, delta, (bf16_t*)dq, cu_q, cu_k, q_stride, k_stride,
                       v_stride, do_stride, dq_stride, Hq, Hkv, B, total_q, softmax_scale, causal);
    return (int)hipGetLastError();
  }
  if (ns == 1) AREAL_LAUNCH_BWD(1, true); else AREAL_LAUNCH_BWD(2, true);
#undef AREAL_LAUNCH_BWD
  {
    const int64_t nvec = (int64_t)total_q * Hq * (D / 8);
    int64_t g = (nvec + 255) / 256;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(flash_bwd_dq_cast_kernel, dim3((unsigned)g), dim3(256), 0, st, dq_acc,
                       (bf16_t*)dq, dq_stride, Hq, (int64_t)total_q);
  }
  return (int)hipGetLastError();
}
