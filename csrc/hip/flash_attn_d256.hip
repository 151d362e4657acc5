// Variable-length causal flash attention for head size 256 (Gemma), forward and backward, gfx950.
//
// Same operand arrangement as the head-size-128 kernels of flash_attn.hip (v_mfma_f32_32x32x16_bf16, an
// accumulator tile feeds the next MFMA as an operand without leaving registers, transposed operands through
// ds_read_b64_tr_b16). A 256-wide row is staged as TWO swizzled sub-tiles of 128 columns (256-byte rows) so the
// conflict-free layout of flash_tiles.hpp carries over unchanged.
//   forward : a wave holds 32 query rows: 16 Q fragments and a [256 x 32] fp32 output accumulator (128 VGPRs),
//             one workgroup per CU.
//   backward: always the two-kernel form. dK / dV: a wave holds K and V of its 32 keys as MFMA operands over the
//             whole head (128 VGPRs) - S = Q.K^T and dP = dO.V^T need all 256 columns - but accumulates only ONE
//             128-column half of dK^T / dV^T; the grid carries the half as its fastest index, so the pair of
//             workgroups that share a key block run side by side and share the Q / dO rows in L2. S and dP are
//             computed twice; the accumulators of both halves (256 VGPRs) next to the operands would not fit.
//             dQ: query-major like the forward (recomputes S and dP, no atomics).
#include "flash_tiles.hpp"

namespace {
constexpr int D = 256;
constexpr int KV_HALF = 64 * 256;  // bytes of one 64-row x 128-column sub-tile
constexpr int Q_HALF = 32 * 256;   // bytes of one 32-row x 128-column sub-tile

// =====================================================================================
// Forward. Grid and XCD placement as in flash_fwd_kernel: 1-D over (q-tile rank, query head of the group,
// rotated (sequence, kv head) slot); block = 4 waves x 32 q rows; KV tile = 64 keys.
// =====================================================================================
__global__ __launch_bounds__(256, 1) void flash_fwd_d256_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
    bf16_t* __restrict__ o, float* __restrict__ lse, const int* __restrict__ cu_q,
    const int* __restrict__ cu_k, int64_t q_stride, int64_t k_stride, int64_t v_stride,
    int64_t o_stride, int Hq, int Hkv, int nbatch, int total_q, float scale_log2, int causal) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // K [2][64][256 B] | V [2][64][256 B]
  char* ks = smem;
  char* vs = smem + 2 * KV_HALF;

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
  const int shift = k_len - q_len;  // bottom-right aligned causal mask

  const int q0 = qt * 128 + wid * 32;
  const int my_q = q0 + r;
  const int my_q_c = my_q < q_len ? my_q : q_len - 1;
  // Q^T as the B operand: lane (r, hf) holds Q[r][16s + 8hf + j]
  bf16x8 qf[16];
  {
    const bf16_t* qp = q + (size_t)(q_beg + my_q_c) * q_stride + (size_t)head * D + hf * 8;
#pragma unroll
    for (int s = 0; s < 16; ++s) qf[s] = ld8(qp + s * 16);
  }
  f32x16 oacc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) oacc[i][j] = 0.f;
  float m = -INFINITY, l = 0.f;

  int k_hi = k_len;
  if (causal) {
    const int last_q = min(q_len, qt * 128 + 128) - 1;
    k_hi = min(k_len, last_q + shift + 1);
  }
  const int ntile_k = k_hi > 0 ? (k_hi + 63) / 64 : 0;

  // cooperative tile load: thread -> (row = tid>>4 + 16*i, column half h, chunk = tid&15)
  const int lrow = tid >> 4, lch = tid & 15;
  bf16x8 kreg[8], vreg[8];
  auto load_tile = [&](int t) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int key = t * 64 + lrow + 16 * i;
      key = key < k_len ? key : k_len - 1;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        kreg[2 * i + h] = ld8(k + (size_t)(k_beg + key) * k_stride + (size_t)kvh * D + h * 128 + lch * 8);
        vreg[2 * i + h] = ld8(v + (size_t)(k_beg + key) * v_stride + (size_t)kvh * D + h * 128 + lch * 8);
      }
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        lds_row_write(ks + h * KV_HALF, lrow + 16 * i, lch, kreg[2 * i + h]);
        lds_row_write(vs + h * KV_HALF, lrow + 16 * i, lch, vreg[2 * i + h]);
      }
  };
  if (ntile_k > 0) {
    load_tile(0);
    store_tile();
  }
  __syncthreads();
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the Q fragments have landed before the loop starts

  for (int t = 0; t < ntile_k; ++t) {
    if (t + 1 < ntile_k) load_tile(t + 1);
    // ---- S^T = K . Q^T : two 32-key blocks
    f32x16 sacc[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int j = 0; j < 16; ++j) sacc[kb][j] = 0.f;
#pragma unroll
      for (int s0 = 0; s0 < 16; s0 += 4) {
        bf16x8 kf[4];
#pragma unroll
        for (int s = 0; s < 4; ++s)
          kf[s] = lds_row_read(ks + ((s0 + s) >> 3) * KV_HALF, 32 * kb + r, 2 * ((s0 + s) & 7) + hf);
        __builtin_amdgcn_sched_barrier(0);
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
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) oacc[i][j] *= alpha;
    }
    // ---- O^T[d, q] += V^T[d, keys] . P^T[keys, q], four 32-column blocks of d at a time
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int d0 = 0; d0 < 8; d0 += 4) {
        bf16x4 a0[4], a1[4];
        const char* tile = vs + (d0 >> 2) * KV_HALF;
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          const int col0 = 32 * db + 16 * (grp & 1);
          a0[db] = tr_read(tile, 16 * s + 4 * hf, col0, gl);
          a1[db] = tr_read(tile, 16 * s + 8 + 4 * hf, col0, gl);
        }
#pragma unroll
        for (int db = 0; db < 4; ++db) oacc[d0 + db] = mfma32(cat4(a0[db], a1[db]), pf[s], oacc[d0 + db]);
      }
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
    for (int db = 0; db < 8; ++db)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        bf16x4 w;
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = f2bf(oacc[db][4 * g4 + j] * inv_l);
        *reinterpret_cast<bf16x4*>(op + 32 * db + 8 * g4 + 4 * hf) = w;
      }
    if (hf == 0 && lse)
      lse[(size_t)head * total_q + q_beg + my_q] = (l > 0.f) ? (m * 0.6931471805599453f + __logf(l)) : -INFINITY;
  }
}

// =====================================================================================
// Backward.
//  pre : delta[h, t] = sum_d dO[t, h, d] * O[t, h, d], 32 lanes per (token, head)
// =====================================================================================
__global__ __launch_bounds__(256) void flash_bwd_delta_d256_kernel(
    const bf16_t* __restrict__ o, const bf16_t* __restrict__ d_o, float* __restrict__ delta,
    int64_t o_stride, int64_t do_stride, int Hq, int total_q) {
  const int64_t idx = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 5;
  const int sub = threadIdx.x & 31;
  if (idx >= (int64_t)total_q * Hq) return;
  const int64_t t = idx / Hq;
  const int h = (int)(idx % Hq);
  bf16x8 a = ld8(o + t * o_stride + (size_t)h * D + sub * 8);
  bf16x8 bb = ld8(d_o + t * do_stride + (size_t)h * D + sub * 8);
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) s += bf2f(a[j]) * bf2f(bb[j]);
#pragma unroll
  for (int off = 16; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if (sub == 0) delta[(size_t)h * total_q + t] = s;
}

// dK / dV. 1-D grid over (key block of 128, (sequence, kv head), column half); block = 4 waves x 32 keys; one
// 32-row query slice per iteration.
__global__ __launch_bounds__(256, 1) void flash_bwd_dkdv_d256_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
    const bf16_t* __restrict__ d_o, const float* __restrict__ lse, const float* __restrict__ delta,
    bf16_t* __restrict__ dk, bf16_t* __restrict__ dv, const int* __restrict__ cu_q, const int* __restrict__ cu_k,
    int64_t q_stride, int64_t k_stride, int64_t v_stride, int64_t do_stride, int64_t dk_stride, int64_t dv_stride,
    int Hq, int Hkv, int nbatch, int total_q, float softmax_scale, int causal) {
  constexpr int QR = 32;
  // LDS: Q [2][32][256 B] | dO [2][32][256 B] | lse[32] f32 | delta[32] f32
  __shared__ __attribute__((aligned(16))) char smem[4 * Q_HALF + 2 * QR * sizeof(float)];
  char* qs = smem;
  char* dos = smem + 2 * Q_HALF;
  float* lse_s = reinterpret_cast<float*>(smem + 4 * Q_HALF);
  float* dl_s = lse_s + QR;

  const int n_bh = Hkv * nbatch;
  const int dhalf = blockIdx.x & 1;
  const int bid = blockIdx.x >> 1;
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

  const int my_key = kb0 + 32 * wid + r;
  const int my_key_c = my_key < k_len ? my_key : k_len - 1;
  bf16x8 kf[16], vf[16];
  {
    const bf16_t* kp = k + (size_t)(k_beg + my_key_c) * k_stride + (size_t)kvh * D + hf * 8;
    const bf16_t* vp = v + (size_t)(k_beg + my_key_c) * v_stride + (size_t)kvh * D + hf * 8;
#pragma unroll
    for (int s = 0; s < 16; ++s) { kf[s] = ld8(kp + s * 16); vf[s] = ld8(vp + s * 16); }
  }
  f32x16 dkacc[4], dvacc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) { dkacc[i][j] = 0.f; dvacc[i][j] = 0.f; }

  int qs_first = 0;  // first query row that can see this key block
  if (causal) {
    const int fq = kb0 - shift;
    qs_first = fq > 0 ? (fq / QR) * QR : 0;
  }
  const int lrow = tid >> 4, lch = tid & 15;
  const int nslice = q_len > qs_first ? (q_len - qs_first + QR - 1) / QR : 0;
  const int nwork = G * nslice;
  bf16x8 qreg[4], doreg[4];
  float lse_r = INFINITY, dl_r = 0.f;
  auto prefetch = [&](int it) {
    const int head = kvh * G + it / nslice;
    const int qs0 = qs_first + QR * (it % nslice);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int qi = qs0 + lrow + 16 * i;
      qi = qi < q_len ? qi : q_len - 1;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        qreg[2 * i + h] = ld8(q + (size_t)(q_beg + qi) * q_stride + (size_t)head * D + h * 128 + lch * 8);
        doreg[2 * i + h] = ld8(d_o + (size_t)(q_beg + qi) * do_stride + (size_t)head * D + h * 128 + lch * 8);
      }
    }
    if (tid < QR) {
      int qi = qs0 + tid;
      const bool ok = qi < q_len;
      qi = ok ? qi : q_len - 1;
      lse_r = ok ? lse[(size_t)head * total_q + q_beg + qi] : INFINITY;  // +inf: p = 0 on rows past the end
      dl_r = ok ? delta[(size_t)head * total_q + q_beg + qi] : 0.f;
    }
  };
  if (nwork > 0) prefetch(0);
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): K / V fragments have landed before the loop

  for (int it = 0; it < nwork; ++it) {
    const int qs0 = qs_first + QR * (it % nslice);
    __syncthreads();  // previous slice fully consumed
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        lds_row_write(qs + h * Q_HALF, lrow + 16 * i, lch, qreg[2 * i + h]);
        lds_row_write(dos + h * Q_HALF, lrow + 16 * i, lch, doreg[2 * i + h]);
      }
    if (tid < QR) {
      lse_s[tid] = lse_r * 1.4426950408889634f;
      dl_s[tid] = dl_r;
    }
    if (it + 1 < nwork) prefetch(it + 1);
    __syncthreads();
    // ---- S = Q.K^T, dP = dO.V^T over all 256 columns (rows = q in registers, col = key on the lane)
    f32x16 sacc, pacc;
#pragma unroll
    for (int j = 0; j < 16; ++j) { sacc[j] = 0.f; pacc[j] = 0.f; }
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      bf16x8 qa = lds_row_read(qs + (s >> 3) * Q_HALF, r, 2 * (s & 7) + hf);
      bf16x8 da = lds_row_read(dos + (s >> 3) * Q_HALF, r, 2 * (s & 7) + hf);
      sacc = mfma32(qa, kf[s], sacc);
      pacc = mfma32(da, vf[s], pacc);
    }
    // ---- P = exp2(S*c - lse), dS = P * (dP - delta)
    bf16x8 pfr[2], dsfr[2];
    const bool need_mask = (kb0 + 128 > k_len) || (causal && (kb0 + 127 > qs0 + shift));
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse_s + 8 * g4 + 4 * hf);
      const f32x4 d4 = *reinterpret_cast<const f32x4*>(dl_s + 8 * g4 + 4 * hf);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int reg = 4 * g4 + j;
        const int qrow = qs0 + 8 * g4 + 4 * hf + j;
        float sv = sacc[reg] * scale_log2 - l4[j];
        if (need_mask && (my_key >= k_len || (causal && my_key > qrow + shift))) sv = -INFINITY;
        const float p = fast_exp2(sv);
        pfr[reg >> 3][reg & 7] = f2bf(p);
        dsfr[reg >> 3][reg & 7] = f2bf(p * (pacc[reg] - d4[j]));
      }
    }
    // ---- this workgroup's column half: dV^T += dO^T.P ; dK^T += Q^T.dS  (A operand from transposed reads)
    const char* do_h = dos + dhalf * Q_HALF;
    const char* q_h = qs + dhalf * Q_HALF;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        const int col0 = 32 * db + 16 * (grp & 1);
        const int row0 = 16 * s;
        bf16x8 da = cat4(tr_read(do_h, row0 + 4 * hf, col0, gl), tr_read(do_h, row0 + 8 + 4 * hf, col0, gl));
        dvacc[db] = mfma32(da, pfr[s], dvacc[db]);
        bf16x8 qa = cat4(tr_read(q_h, row0 + 4 * hf, col0, gl), tr_read(q_h, row0 + 8 + 4 * hf, col0, gl));
        dkacc[db] = mfma32(qa, dsfr[s], dkacc[db]);
      }
    }
  }
  // ---- epilogue: lane (key col = r, hf), reg -> d row of this half
  if (my_key < k_len) {
    bf16_t* dkp = dk + (size_t)(k_beg + my_key) * dk_stride + (size_t)kvh * D + dhalf * 128;
    bf16_t* dvp = dv + (size_t)(k_beg + my_key) * dv_stride + (size_t)kvh * D + dhalf * 128;
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

// dQ, one workgroup per 128-row query tile (grid of the forward): S^T = K.Q^T and dP^T = V.dO^T keep the query on
// the lane, dS^T comes out of the accumulators as the B operand of dQ^T += K^T.dS^T.
__global__ __launch_bounds__(256, 1) void flash_bwd_dq_d256_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
    const bf16_t* __restrict__ d_o, const float* __restrict__ lse, const float* __restrict__ delta,
    bf16_t* __restrict__ dq, const int* __restrict__ cu_q, const int* __restrict__ cu_k, int64_t q_stride,
    int64_t k_stride, int64_t v_stride, int64_t do_stride, int64_t dq_stride, int Hq, int Hkv, int nbatch,
    int total_q, float softmax_scale, int causal) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // K [2][64][256 B] | V [2][64][256 B]
  char* ks = smem;
  char* vs = smem + 2 * KV_HALF;
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
  bf16x8 qf[16], dof[16];
  {
    const bf16_t* qp = q + (size_t)(q_beg + my_q_c) * q_stride + (size_t)head * D + hf * 8;
    const bf16_t* dp = d_o + (size_t)(q_beg + my_q_c) * do_stride + (size_t)head * D + hf * 8;
#pragma unroll
    for (int s = 0; s < 16; ++s) { qf[s] = ld8(qp + s * 16); dof[s] = ld8(dp + s * 16); }
  }
  const float lse2 = q_ok ? lse[(size_t)head * total_q + q_beg + my_q] * 1.4426950408889634f : INFINITY;
  const float dlt = q_ok ? delta[(size_t)head * total_q + q_beg + my_q] : 0.f;
  f32x16 dqacc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) dqacc[i][j] = 0.f;

  int k_hi = k_len;
  if (causal) {
    const int last_q = min(q_len, qt * 128 + 128) - 1;
    k_hi = min(k_len, last_q + shift + 1);
  }
  const int ntile_k = k_hi > 0 ? (k_hi + 63) / 64 : 0;
  const int lrow = tid >> 4, lch = tid & 15;
  // K / V tiles go from global memory to LDS through registers right before the barrier pair (no register
  // prefetch across the MFMAs: Q, dO and the dQ accumulator already hold 256 VGPRs per lane)
  auto stage_tile = [&](int t) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int key = t * 64 + lrow + 16 * i;
      key = key < k_len ? key : k_len - 1;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        lds_row_write(ks + h * KV_HALF, lrow + 16 * i, lch,
                      ld8(k + (size_t)(k_beg + key) * k_stride + (size_t)kvh * D + h * 128 + lch * 8));
        lds_row_write(vs + h * KV_HALF, lrow + 16 * i, lch,
                      ld8(v + (size_t)(k_beg + key) * v_stride + (size_t)kvh * D + h * 128 + lch * 8));
      }
    }
  };
  if (ntile_k > 0) stage_tile(0);
  __syncthreads();

  for (int t = 0; t < ntile_k; ++t) {
    const int kbase = t * 64;
    const bool need_mask = (kbase + 64 > k_len) || (causal && (kbase + 63 > q0 + shift));
    bf16x8 dsf[4];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      f32x16 sacc, pacc;
#pragma unroll
      for (int j = 0; j < 16; ++j) { sacc[j] = 0.f; pacc[j] = 0.f; }
#pragma unroll
      for (int s0 = 0; s0 < 16; s0 += 4) {
        bf16x8 kf[4], vf[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          kf[s] = lds_row_read(ks + ((s0 + s) >> 3) * KV_HALF, 32 * kb + r, 2 * ((s0 + s) & 7) + hf);
          vf[s] = lds_row_read(vs + ((s0 + s) >> 3) * KV_HALF, 32 * kb + r, 2 * ((s0 + s) & 7) + hf);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          sacc = mfma32(kf[s], qf[s0 + s], sacc);
          pacc = mfma32(vf[s], dof[s0 + s], pacc);
        }
      }
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
#pragma unroll
      for (int d0 = 0; d0 < 8; d0 += 4) {
        bf16x4 a0[4], a1[4];
        const char* tile = ks + (d0 >> 2) * KV_HALF;
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          const int col0 = 32 * db + 16 * (grp & 1);
          a0[db] = tr_read(tile, 16 * s + 4 * hf, col0, gl);
          a1[db] = tr_read(tile, 16 * s + 8 + 4 * hf, col0, gl);
        }
#pragma unroll
        for (int db = 0; db < 4; ++db) dqacc[d0 + db] = mfma32(cat4(a0[db], a1[db]), dsf[s], dqacc[d0 + db]);
      }
    }
    __syncthreads();
    if (t + 1 < ntile_k) stage_tile(t + 1);
    __syncthreads();
  }
  if (q_ok) {
    bf16_t* op = dq + (size_t)(q_beg + my_q) * dq_stride + (size_t)head * D;
#pragma unroll
    for (int db = 0; db < 8; ++db)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        bf16x4 w;
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = f2bf(dqacc[db][4 * g4 + j] * softmax_scale);
        *reinterpret_cast<bf16x4*>(op + 32 * db + 8 * g4 + 4 * hf) = w;
      }
  }
}

constexpr int KV_LDS = 4 * KV_HALF;  // 64 KiB: K and V tiles of the forward and the dQ kernel
}  // namespace

extern "C" int areal_flash_fwd_d256(const void* q, const void* k, const void* v, void* o, float* lse,
                                    const int* cu_q, const int* cu_k, int B, int max_q, int total_q, int Hq, int Hkv,
                                    int64_t q_stride, int64_t k_stride, int64_t v_stride, int64_t o_stride,
                                    float softmax_scale, int causal, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    HIP_RET(hipFuncSetAttribute((const void*)flash_fwd_d256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, KV_LDS));
    attr_set = true;
  }
  const int n_g_pad = (Hkv * B + 7) & ~7;
  dim3 grid((unsigned)(((max_q + 127) / 128) * (Hq / Hkv) * n_g_pad)), block(256);
  hipLaunchKernelGGL(flash_fwd_d256_kernel, grid, block, KV_LDS, st, (const bf16_t*)q, (const bf16_t*)k,
                     (const bf16_t*)v, (bf16_t*)o, lse, cu_q, cu_k, q_stride, k_stride, v_stride, o_stride, Hq, Hkv, B,
                     total_q, softmax_scale * 1.4426950408889634f, causal);
  return (int)hipGetLastError();
}

extern "C" int areal_flash_bwd_d256(const void* q, const void* k, const void* v, const void* o, const void* d_o,
                                    const float* lse, float* delta, void* dq, void* dk, void* dv, const int* cu_q,
                                    const int* cu_k, int B, int max_k, int total_q, int Hq, int Hkv, int64_t q_stride,
                                    int64_t k_stride, int64_t v_stride, int64_t o_stride, int64_t do_stride,
                                    int64_t dq_stride, int64_t dk_stride, int64_t dv_stride, float softmax_scale,
                                    int causal, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    HIP_RET(hipFuncSetAttribute((const void*)flash_bwd_dq_d256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, KV_LDS));
    attr_set = true;
  }
  {
    const int64_t n = (int64_t)total_q * Hq * 32;
    hipLaunchKernelGGL(flash_bwd_delta_d256_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                       (const bf16_t*)o, (const bf16_t*)d_o, delta, o_stride, do_stride, Hq, total_q);
  }
  dim3 block(256);
  dim3 kgrid((unsigned)(((max_k + 127) / 128) * Hkv * B * 2));
  hipLaunchKernelGGL(flash_bwd_dkdv_d256_kernel, kgrid, block, 0, st, (const bf16_t*)q, (const bf16_t*)k,
                     (const bf16_t*)v, (const bf16_t*)d_o, lse, delta, (bf16_t*)dk, (bf16_t*)dv, cu_q, cu_k, q_stride,
                     k_stride, v_stride, do_stride, dk_stride, dv_stride, Hq, Hkv, B, total_q, softmax_scale, causal);
  const int n_g_pad = (Hkv * B + 7) & ~7;
  dim3 qgrid((unsigned)(((total_q + 127) / 128) * (Hq / Hkv) * n_g_pad));
  hipLaunchKernelGGL(flash_bwd_dq_d256_kernel, qgrid, block, KV_LDS, st, (const bf16_t*)q, (const bf16_t*)k,
                     (const bf16_t*)v, (const bf16_t*)d_o, lse, delta, (bf16_t*)dq, cu_q, cu_k, q_stride, k_stride,
                     v_stride, do_stride, dq_stride, Hq, Hkv, B, total_q, softmax_scale, causal);
  return (int)hipGetLastError();
}
