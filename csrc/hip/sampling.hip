// Fused temperature / top-k / top-p sampling with log-prob output for gfx950.
//
// One 1024-thread block per batch row. Replaces the torch sort/cumsum/topk logits
// warpers + Categorical sampling of the reference's native generator
// (realhf/impl/model/utils/logits_warper.py:52-125, 207-225;
// realhf/impl/model/nn/real_llm_generate.py:30-146) and SGLang's sampler.
//
//  * greedy (temperature == 0): arg-max.
//  * unfiltered (top_k >= V and top_p >= 1): Gumbel-max over z = logit / T, in the same single pass
//    over the row (16-byte loads) that accumulates the running max / sum-exp for the log-prob.
//  * filtered: thresholds for top-k (k-th largest z) and top-p (smallest set whose mass
//    reaches top_p of the top-k mass) are found by bisection on z - over the whole row only until
//    the bracket holds a few thousand entries, then among those entries in LDS (fixed-point sums:
//    order-independent) -, then Gumbel-max runs over the surviving entries. No sort, no [B, V]
//    temporaries.
//  * frequency / presence penalties: a request that asks for them owns a row of a token-count table
//    (`counts`, int16 [slots, count_ld]); its logits become logit - freq * count - pres * (count > 0)
//    before the temperature, in every pass, and the kernel bumps the count of the token it draws.
//  * rows flagged inactive (`active[row] == 0`: finished inside a multi-step decode plan) return at
//    once with token 0.
// The reported log-prob is log_softmax(adjusted logit / T)[token] over the FULL vocabulary (before
// top-k / top-p filtering); without penalties this is what the trainer recomputes.
// Randomness: Philox keyed by (seed, row, step) where `step` is read from device memory and
// can be bumped by the kernel itself, so a captured graph replays with fresh noise.
#include "common.hpp"

typedef short i16x8 __attribute__((ext_vector_type(8)));

struct ArgMax {
  float v;
  int i;
};
DEVINL ArgMax amax(ArgMax a, ArgMax b) {
  return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a;
}
DEVINL ArgMax block_argmax(ArgMax x, float* redv, int* redi) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    ArgMax y;
    y.v = __shfl_xor(x.v, o, 64);
    y.i = __shfl_xor(x.i, o, 64);
    x = amax(x, y);
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if (lane == 0) { redv[wid] = x.v; redi[wid] = x.i; }
  __syncthreads();
  ArgMax r;
  r.v = lane < nw ? redv[lane] : -INFINITY;
  r.i = lane < nw ? redi[lane] : 0x7fffffff;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    ArgMax y;
    y.v = __shfl_xor(r.v, o, 64);
    y.i = __shfl_xor(r.i, o, 64);
    r = amax(r, y);
  }
  return r;
}

template <typename T>
DEVINL float ldz(const T* p, int i) { return (float)p[i]; }

// 8 consecutive entries of a row as floats (one 16-byte load for bf16, two for fp32)
DEVINL void load8(const bf16_t* p, float* z) {
  const bf16x8 v = ld8(p);
#pragma unroll
  for (int j = 0; j < 8; ++j) z[j] = (float)v[j];
}
DEVINL void load8(const float* p, float* z) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(p);
  const f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
  for (int j = 0; j < 4; ++j) { z[j] = a[j]; z[4 + j] = b[j]; }
}

// One row of logits with its penalties: z_i = (logit_i - freq * count_i - pres * [count_i > 0]) / T.
template <typename T>
struct Row {
  const T* p;
  const int16_t* cnt;  // this request's token counts, or nullptr
  float freq, pres, inv_t;
  DEVINL float adjust(float x, int c) const { return x - freq * (float)c - (c > 0 ? pres : 0.f); }
  DEVINL float z1(int i) const {
    float x = ldz(p, i);
    if (cnt) x = adjust(x, cnt[i]);
    return x * inv_t;
  }
  // entries [i8, i8 + 8), i8 a multiple of 8 (16-byte loads of the logits and of the counts)
  DEVINL void z8(int i8, float* z) const {
    load8(p + i8, z);
    if (cnt) {
      const i16x8 c = *reinterpret_cast<const i16x8*>(cnt + i8);
#pragma unroll
      for (int j = 0; j < 8; ++j) z[j] = adjust(z[j], c[j]);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) z[j] *= inv_t;
  }
};

// f(i, z_i) over one row: vector loads over [0, V8), scalar loads over the tail [V8, V)
// (a thread's chunks are fetched four at a time: with one load in flight a pass over a 152k-entry row is a chain of
// ~19 memory round trips per thread, and the filtered path makes dozens of passes)
template <typename T, typename F>
DEVINL void for_each_z(const Row<T>& r, int V, int V8, F f) {
  constexpr int PF = 4;
  const int stride = blockDim.x * 8;
  for (int base = threadIdx.x * 8; base < V8; base += stride * PF) {
    float zz[PF][8];
#pragma unroll
    for (int u = 0; u < PF; ++u)
      if (base + u * stride < V8) r.z8(base + u * stride, zz[u]);
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int i8 = base + u * stride;
      if (i8 >= V8) break;
#pragma unroll
      for (int j = 0; j < 8; ++j) f(i8 + j, zz[u][j]);
    }
  }
  for (int i = V8 + threadIdx.x; i < V; i += blockDim.x) f(i, r.z1(i));
}

DEVINL float gumbel(uint32_t bits) { return -__logf(-__logf(u32_to_unit(bits) * 0.99999994f)); }

DEVINL unsigned long long block_sum_u64(unsigned long long v, unsigned long long* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) red[wid] = v;
  __syncthreads();
  unsigned long long t = lane < nw ? red[lane] : 0ull;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
  return t;
}

constexpr int THR_CAP = 8192;  // candidates kept in LDS once a threshold's bracket holds no more than that
constexpr float THR_FIX = 1099511627776.f;  // 2^40: fixed-point scale of exp(z - max) in the LDS phase

// Largest t in [lo, hi] with F(t) >= target, F(t) = number (MASS = false) or softmax mass relative to exp(-mx)
// (MASS = true) of the entries z >= t, to the resolution of 32 halvings of [lo, hi] - the threshold of top-k
// (target k) and of top-p (target p x mass of the allowed entries; lo = the top-k threshold, so that only allowed
// entries count). The halvings run over the whole row only until the bracket [lo, hi) holds at most THR_CAP
// entries; those are copied into LDS and the remaining halvings run there. Sums of the LDS phase are taken in fixed
// point: they do not depend on the order in which the candidates were appended, so replays and the ranks of a
// tensor-parallel group agree bit for bit.
template <bool MASS, typename T>
DEVINL float find_threshold(const Row<T>& R, int V, int V8, float lo, float hi, float target, float mx, float* red,
                            unsigned long long* red64, float* buf, int* buf_n, float n_ge_lo) {
  // n_ge_lo: number of entries >= lo, or any upper bound of it (V when unknown)
  float c_lo = n_ge_lo, c_hi = 0.f, F_hi = 0.f;
  bool hi_moved = false;
  int it = 0;
  for (; it < 32 && c_lo - c_hi > (float)THR_CAP; ++it) {
    const float mid = 0.5f * (lo + hi);
    float cm = 0.f, fm = 0.f;
    for_each_z(R, V, V8, [&](int, float z) {
      if (z >= mid) {
        cm += 1.f;
        if (MASS) fm += __expf(z - mx);
      }
    });
    cm = block_sum(cm, red);
    const float Fm = MASS ? block_sum(fm, red) : cm;
    if (Fm >= target) {
      lo = mid;
      c_lo = cm;
    } else {
      hi = mid;
      c_hi = cm;
      F_hi = Fm;
      hi_moved = true;
    }
  }
  if (it == 32) return lo;
  if (threadIdx.x == 0) *buf_n = 0;
  __syncthreads();
  for_each_z(R, V, V8, [&](int, float z) {
    if (z >= lo && (!hi_moved || z < hi)) {
      const int pos = atomicAdd(buf_n, 1);
      if (pos < THR_CAP) buf[pos] = z;
    }
  });
  __syncthreads();
  const int n = min(*buf_n, THR_CAP);
  for (; it < 32; ++it) {
    const float mid = 0.5f * (lo + hi);
    unsigned long long q = 0ull;
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
      const float z = buf[j];
      if (z >= mid) q += MASS ? (unsigned long long)(__expf(z - mx) * THR_FIX) : 1ull;
    }
    q = block_sum_u64(q, red64);
    const float Fm = F_hi + (MASS ? (float)q * (1.f / THR_FIX) : (float)q);
    if (Fm >= target) lo = mid; else hi = mid;
  }
  return lo;
}

template <typename T>
__global__ __launch_bounds__(1024) void sample_kernel(
    const T* __restrict__ logits, int64_t ld, int V, const float* __restrict__ temperature,
    const int* __restrict__ top_k, const float* __restrict__ top_p, uint64_t seed,
    const int64_t* __restrict__ step_ptr, int64_t* __restrict__ out_tok,
    float* __restrict__ out_logp, const int* __restrict__ active, int16_t* __restrict__ counts,
    int64_t count_ld, const int* __restrict__ count_slot, const float* __restrict__ freq_pen,
    const float* __restrict__ pres_pen) {
  __shared__ float red[16];
  __shared__ float redv[16];
  __shared__ int redi[16];
  __shared__ unsigned long long red64[16];
  __shared__ float cand[THR_CAP];
  __shared__ int cand_n;
  const int row = blockIdx.x;
  if (active && !active[row]) {  // uniform over the block
    if (threadIdx.x == 0) {
      out_tok[row] = 0;
      if (out_logp) out_logp[row] = 0.f;
    }
    return;
  }
  const T* p = logits + (size_t)row * ld;
  const float temp = temperature ? temperature[row] : 1.f;
  const bool greedy = temp <= 0.f;
  const float inv_t = greedy ? 1.f : 1.f / temp;
  const int k = top_k ? top_k[row] : V;
  const float tp = top_p ? top_p[row] : 1.f;
  const uint64_t step = step_ptr ? (uint64_t)step_ptr[0] : 0;
  const bool filtered = !greedy && ((k < V && k >= 1) || tp < 1.f);
  // noise word of entry i: word i & 3 of the Philox block with counter ctr0 + i / 4
  const uint64_t ctr0 = step * (uint64_t)((V + 3) / 4);
  const int V8 = (reinterpret_cast<uintptr_t>(p) & 15) == 0 ? (V & ~7) : 0;
  const int slot = (counts && count_slot) ? count_slot[row] : -1;
  int16_t* cnt = slot >= 0 ? counts + (size_t)slot * count_ld : nullptr;  // count_ld is a multiple of 8
  Row<T> R;
  R.p = p;
  R.cnt = cnt;
  R.freq = cnt && freq_pen ? freq_pen[row] : 0.f;
  R.pres = cnt && pres_pen ? pres_pen[row] : 0.f;
  R.inv_t = inv_t;

  // pass 1: running max / sum-exp (rescaled when the max moves) and min of z. Without filtering the
  // arg-max of z (+ Gumbel noise unless greedy) is taken in the same pass, so the row is read once.
  const bool pick_now = !filtered;
  const bool noise = !greedy;
  float mx = -INFINITY, mn = INFINITY, se = 0.f;
  ArgMax best;
  best.v = -INFINITY;
  best.i = 0x7fffffff;
  // The row is walked in groups of PF chunks per thread whose loads are all issued before the first one is used: a
  // thread sees ~19 chunks of a 152k-entry row, and with one load in flight at a time the pass is a chain of that
  // many memory round trips (the per-entry noise and exponentials are cheap next to them).
  constexpr int PF = 4;
  for (int base = threadIdx.x * 8; base < V8; base += blockDim.x * 8 * PF) {
    float zz[PF][8];
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int i8 = base + u * (int)blockDim.x * 8;
      if (i8 < V8) R.z8(i8, zz[u]);
    }
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int i8 = base + u * (int)blockDim.x * 8;
      if (i8 >= V8) break;
      float* z = zz[u];
      float m8 = -INFINITY;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        m8 = fmaxf(m8, z[j]);
        mn = fminf(mn, z[j]);
      }
      if (m8 > mx) {
        se *= __expf(mx - m8);
        mx = m8;
      }
      if (mx > -INFINITY) {
#pragma unroll
        for (int j = 0; j < 8; ++j) se += __expf(z[j] - mx);
      }
      if (pick_now) {
        u32x4 r0, r1;
        if (noise) {
          r0 = Philox::gen(seed, (uint64_t)row, ctr0 + (uint64_t)(i8 >> 2));
          r1 = Philox::gen(seed, (uint64_t)row, ctr0 + (uint64_t)(i8 >> 2) + 1);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          ArgMax c;
          c.v = noise ? z[j] + gumbel(j < 4 ? r0[j] : r1[j - 4]) : z[j];
          c.i = i8 + j;
          best = amax(best, c);
        }
      }
    }
  }
  for (int i = V8 + threadIdx.x; i < V; i += blockDim.x) {
    const float z = R.z1(i);
    mn = fminf(mn, z);
    if (z > mx) {
      se *= __expf(mx - z);
      mx = z;
    }
    if (mx > -INFINITY) se += __expf(z - mx);
    if (pick_now) {
      ArgMax c;
      c.v = z;
      c.i = i;
      if (noise) c.v += gumbel(Philox::gen(seed, (uint64_t)row, ctr0 + (uint64_t)(i >> 2))[i & 3]);
      best = amax(best, c);
    }
  }
  const float mx_all = block_max(mx, red);
  mn = -block_max(-mn, red);
  se = block_sum(mx > -INFINITY ? se * __expf(mx - mx_all) : 0.f, red);
  mx = mx_all;
  const float lse = mx + __logf(se);

  if (filtered) {
    float thr = -INFINITY;
    float mass_k = se;  // mass (relative to exp(-mx)) of the top-k set
    float n_allowed = (float)V;  // entries in it (ties at the k-th value included)
    if (k < V && k >= 1) {
      // largest t with count(z >= t) >= k
      thr = find_threshold<false>(R, V, V8, mn, mx, (float)k, mx, red, red64, cand, &cand_n, (float)V);
      if (tp < 1.f) {  // the nucleus is taken inside the top-k set: its mass is the reference
        float mk = 0.f, nk = 0.f;
        for_each_z(R, V, V8, [&](int, float z) {
          if (z >= thr) {
            mk += __expf(z - mx);
            nk += 1.f;
          }
        });
        mass_k = block_sum(mk, red);
        n_allowed = block_sum(nk, red);
      }
    }
    if (tp < 1.f) {
      // largest t >= thr with mass(z >= t) >= tp * mass_k
      const float t2 = find_threshold<true>(R, V, V8, fmaxf(thr, mn), mx, tp * mass_k, mx, red, red64, cand, &cand_n,
                                            n_allowed);
      thr = fmaxf(thr, t2);
    }
    // arg-max of z + Gumbel noise over the entries that survive the threshold
    // (few entries survive: the noise block of four entries is generated only where one of them does)
    for_each_z(R, V, V8, [&](int i, float z) {
      if (z >= thr) {
        ArgMax c;
        c.v = z + gumbel(Philox::gen(seed, (uint64_t)row, ctr0 + (uint64_t)(i >> 2))[i & 3]);
        c.i = i;
        best = amax(best, c);
      }
    });
  }
  best = block_argmax(best, redv, redi);
  if (threadIdx.x == 0) {
    int tok = best.i;
    if (tok < 0 || tok >= V) tok = 0;
    out_tok[row] = tok;
    if (out_logp) out_logp[row] = R.z1(tok) - lse;
    if (cnt && cnt[tok] < 32767) cnt[tok] += 1;
  }
}

// ---------------------------------------------------------------------------------------------
// Unfiltered sampling (no top-k / top-p anywhere in the batch) with every row split over `S` workgroups: one
// 1024-thread workgroup per row leaves a 384-row decode batch with 1.5 workgroups per CU, each walking 152k entries
// as a chain of dependent passes. Here S x B workgroups of 256 threads each reduce a slice of a row to
// (max, sum-exp relative to it, best perturbed value, its index); a one-thread-per-row kernel merges the slices,
// writes token and log-prob, bumps the token count and the step. The noise of entry i is the same Philox word as
// in the one-workgroup kernel, so both draw the SAME token; the log-prob differs by summation order only.
template <typename T>
__global__ __launch_bounds__(256) void sample_partial_kernel(
    const T* __restrict__ logits, int64_t ld, int V, const float* __restrict__ temperature, uint64_t seed,
    const int64_t* __restrict__ step_ptr, const int* __restrict__ active, const int16_t* __restrict__ counts,
    int64_t count_ld, const int* __restrict__ count_slot, const float* __restrict__ freq_pen,
    const float* __restrict__ pres_pen, float* __restrict__ ws, int S) {
  __shared__ float red[16];
  __shared__ float redv[16];
  __shared__ int redi[16];
  const int row = blockIdx.y, s = blockIdx.x;
  if (active && !active[row]) return;
  const T* p = logits + (size_t)row * ld;
  const float temp = temperature ? temperature[row] : 1.f;
  const bool greedy = temp <= 0.f;
  const uint64_t step = step_ptr ? (uint64_t)step_ptr[0] : 0;
  const uint64_t ctr0 = step * (uint64_t)((V + 3) / 4);
  const int V8 = (reinterpret_cast<uintptr_t>(p) & 15) == 0 ? (V & ~7) : 0;
  const int slot = (counts && count_slot) ? count_slot[row] : -1;
  Row<T> R;
  R.p = p;
  R.cnt = slot >= 0 ? counts + (size_t)slot * count_ld : nullptr;
  R.freq = R.cnt && freq_pen ? freq_pen[row] : 0.f;
  R.pres = R.cnt && pres_pen ? pres_pen[row] : 0.f;
  R.inv_t = greedy ? 1.f : 1.f / temp;
  const bool noise = !greedy;
  const int chunks = V8 >> 3;
  const int per = (chunks + S - 1) / S;
  const int lo = min(chunks, s * per) * 8, hi = min(chunks, (s + 1) * per) * 8;
  float mx = -INFINITY, se = 0.f;
  ArgMax best;
  best.v = -INFINITY;
  best.i = 0x7fffffff;
  constexpr int PF = 4;
  for (int base = lo + threadIdx.x * 8; base < hi; base += 256 * 8 * PF) {
    float zz[PF][8];
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int i8 = base + u * 256 * 8;
      if (i8 < hi) R.z8(i8, zz[u]);
    }
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int i8 = base + u * 256 * 8;
      if (i8 >= hi) break;
      float* z = zz[u];
      float m8 = -INFINITY;
#pragma unroll
      for (int j = 0; j < 8; ++j) m8 = fmaxf(m8, z[j]);
      if (m8 > mx) {
        se *= __expf(mx - m8);
        mx = m8;
      }
      if (mx > -INFINITY) {
#pragma unroll
        for (int j = 0; j < 8; ++j) se += __expf(z[j] - mx);
      }
      u32x4 r0, r1;
      if (noise) {
        r0 = Philox::gen(seed, (uint64_t)row, ctr0 + (uint64_t)(i8 >> 2));
        r1 = Philox::gen(seed, (uint64_t)row, ctr0 + (uint64_t)(i8 >> 2) + 1);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        ArgMax c;
        c.v = noise ? z[j] + gumbel(j < 4 ? r0[j] : r1[j - 4]) : z[j];
        c.i = i8 + j;
        best = amax(best, c);
      }
    }
  }
  if (s == S - 1) {  // the entries behind the last whole group of 8 (or the whole row when it is not 16-byte aligned)
    for (int i = V8 + threadIdx.x; i < V; i += 256) {
      const float z = R.z1(i);
      if (z > mx) {
        se *= __expf(mx - z);
        mx = z;
      }
      if (mx > -INFINITY) se += __expf(z - mx);
      ArgMax c;
      c.v = z;
      c.i = i;
      if (noise) c.v += gumbel(Philox::gen(seed, (uint64_t)row, ctr0 + (uint64_t)(i >> 2))[i & 3]);
      best = amax(best, c);
    }
  }
  const float mx_all = block_max(mx, red);
  se = block_sum(mx > -INFINITY ? se * __expf(mx - mx_all) : 0.f, red);
  best = block_argmax(best, redv, redi);
  if (threadIdx.x == 0) {
    float* o = ws + ((size_t)row * S + s) * 4;
    o[0] = mx_all;
    o[1] = se;
    o[2] = best.v;
    o[3] = __int_as_float(best.i);
  }
}

template <typename T>
__global__ void sample_combine_kernel(const T* __restrict__ logits, int64_t ld, int V, int B,
                                      const float* __restrict__ temperature, int64_t* __restrict__ step_ptr,
                                      int bump_step, int64_t* __restrict__ out_tok, float* __restrict__ out_logp,
                                      const int* __restrict__ active, int16_t* __restrict__ counts, int64_t count_ld,
                                      const int* __restrict__ count_slot, const float* __restrict__ freq_pen,
                                      const float* __restrict__ pres_pen, const float* __restrict__ ws, int S) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row == 0 && bump_step && step_ptr) step_ptr[0] += 1;  // every reader of the step ran in the previous kernel
  if (row >= B) return;
  if (active && !active[row]) {
    out_tok[row] = 0;
    if (out_logp) out_logp[row] = 0.f;
    return;
  }
  float mx = -INFINITY;
  for (int s = 0; s < S; ++s) mx = fmaxf(mx, ws[((size_t)row * S + s) * 4]);
  float se = 0.f;
  ArgMax best;
  best.v = -INFINITY;
  best.i = 0x7fffffff;
  for (int s = 0; s < S; ++s) {
    const float* o = ws + ((size_t)row * S + s) * 4;
    if (o[0] > -INFINITY) se += o[1] * __expf(o[0] - mx);
    ArgMax c;
    c.v = o[2];
    c.i = __float_as_int(o[3]);
    best = amax(best, c);
  }
  const float temp = temperature ? temperature[row] : 1.f;
  const int slot = (counts && count_slot) ? count_slot[row] : -1;
  Row<T> R;
  R.p = logits + (size_t)row * ld;
  R.cnt = slot >= 0 ? counts + (size_t)slot * count_ld : nullptr;
  R.freq = R.cnt && freq_pen ? freq_pen[row] : 0.f;
  R.pres = R.cnt && pres_pen ? pres_pen[row] : 0.f;
  R.inv_t = temp <= 0.f ? 1.f : 1.f / temp;
  int tok = best.i;
  if (tok < 0 || tok >= V) tok = 0;
  out_tok[row] = tok;
  if (out_logp) out_logp[row] = R.z1(tok) - (mx + __logf(se));
  if (slot >= 0) {
    int16_t* cnt = counts + (size_t)slot * count_ld;
    if (cnt[tok] < 32767) cnt[tok] += 1;
  }
}

// Workspace floats the split path needs for B rows, and the number of slices per row it will use.
extern "C" int areal_sample_splits(int B) {
  if (B >= 1024) return 1;
  const int s = (1024 + B - 1) / B;
  return s > 16 ? 16 : s;
}

extern "C" int areal_sample_unfiltered(const void* logits, int dtype, int B, int V, int64_t ld,
                                       const float* temperature, uint64_t seed, int64_t* step_ptr, int bump_step,
                                       int64_t* out_tok, float* out_logp, const int* active, int16_t* counts,
                                       int64_t count_ld, const int* count_slot, const float* freq_pen,
                                       const float* pres_pen, float* ws, int S, hipStream_t st) {
  if (B == 0) return 0;
  if (S < 1 || S > 64 || !ws) return (int)hipErrorInvalidValue;
  if (counts && (count_ld % 8 != 0 || count_ld < V)) return (int)hipErrorInvalidValue;
  const dim3 grid(S, B);
  const int cb = (B + 63) / 64;
  if (dtype == 0) {
    hipLaunchKernelGGL(sample_partial_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)logits, ld, V,
                       temperature, seed, step_ptr, active, counts, count_ld, count_slot, freq_pen, pres_pen, ws, S);
    hipLaunchKernelGGL(sample_combine_kernel<bf16_t>, dim3(cb), dim3(64), 0, st, (const bf16_t*)logits, ld, V, B,
                       temperature, step_ptr, bump_step, out_tok, out_logp, active, counts, count_ld, count_slot,
                       freq_pen, pres_pen, ws, S);
  } else {
    hipLaunchKernelGGL(sample_partial_kernel<float>, grid, dim3(256), 0, st, (const float*)logits, ld, V,
                       temperature, seed, step_ptr, active, counts, count_ld, count_slot, freq_pen, pres_pen, ws, S);
    hipLaunchKernelGGL(sample_combine_kernel<float>, dim3(cb), dim3(64), 0, st, (const float*)logits, ld, V, B,
                       temperature, step_ptr, bump_step, out_tok, out_logp, active, counts, count_ld, count_slot,
                       freq_pen, pres_pen, ws, S);
  }
  return (int)hipGetLastError();
}

__global__ void bump_step_kernel(int64_t* step_ptr) {
  if (threadIdx.x == 0 && blockIdx.x == 0) step_ptr[0] += 1;
}

extern "C" int areal_sample(const void* logits, int dtype, int B, int V, int64_t ld,
                            const float* temperature, const int* top_k, const float* top_p,
                            uint64_t seed, int64_t* step_ptr, int bump_step, int64_t* out_tok,
                            float* out_logp, const int* active, int16_t* counts, int64_t count_ld,
                            const int* count_slot, const float* freq_pen, const float* pres_pen,
                            hipStream_t st) {
  if (B == 0) return 0;
  if (counts && (count_ld % 8 != 0 || count_ld < V)) return (int)hipErrorInvalidValue;
  if (dtype == 0)
    hipLaunchKernelGGL(sample_kernel<bf16_t>, dim3(B), dim3(1024), 0, st, (const bf16_t*)logits, ld, V,
                       temperature, top_k, top_p, seed, step_ptr, out_tok, out_logp, active, counts, count_ld,
                       count_slot, freq_pen, pres_pen);
  else
    hipLaunchKernelGGL(sample_kernel<float>, dim3(B), dim3(1024), 0, st, (const float*)logits, ld, V,
                       temperature, top_k, top_p, seed, step_ptr, out_tok, out_logp, active, counts, count_ld,
                       count_slot, freq_pen, pres_pen);
  if (bump_step && step_ptr)
    hipLaunchKernelGGL(bump_step_kernel, dim3(1), dim3(64), 0, st, step_ptr);
  return (int)hipGetLastError();
}
