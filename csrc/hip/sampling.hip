// Fused temperature / top-k / top-p sampling with log-prob output for gfx950.
//
// One 1024-thread block per batch row. Replaces the torch sort/cumsum/topk logits
// warpers + Categorical sampling of the reference's native generator
// (realhf/impl/model/utils/logits_warper.py:52-125, 207-225;
// realhf/impl/model/nn/real_llm_generate.py:30-146) and SGLang's sampler.
//
//  * greedy (temperature == 0): arg-max.
//  * unfiltered (top_k >= V and top_p >= 1): Gumbel-max over z = logit / T. One pass, no sort.
//  * filtered: thresholds for top-k (k-th largest z) and top-p (smallest set whose mass
//    reaches top_p of the top-k mass) are found by bisection on z, then Gumbel-max runs over
//    the surviving entries. No sort, no [B, V] temporaries.
// The reported log-prob is log_softmax(logit / T)[token] over the FULL vocabulary (before
// filtering), which is what the trainer recomputes.
// Randomness: Philox keyed by (seed, row, step) where `step` is read from device memory and
// can be bumped by the kernel itself, so a captured graph replays with fresh noise.
#include "common.hpp"

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

template <typename T>
__global__ __launch_bounds__(1024) void sample_kernel(
    const T* __restrict__ logits, int64_t ld, int V, const float* __restrict__ temperature,
    const int* __restrict__ top_k, const float* __restrict__ top_p, uint64_t seed,
    const int64_t* __restrict__ step_ptr, int64_t* __restrict__ out_tok,
    float* __restrict__ out_logp) {
  __shared__ float red[16];
  __shared__ float redv[16];
  __shared__ int redi[16];
  const int row = blockIdx.x;
  const T* p = logits + (size_t)row * ld;
  const float temp = temperature ? temperature[row] : 1.f;
  const bool greedy = temp <= 0.f;
  const float inv_t = greedy ? 1.f : 1.f / temp;
  const int k = top_k ? top_k[row] : V;
  const float tp = top_p ? top_p[row] : 1.f;
  const uint64_t step = step_ptr ? (uint64_t)step_ptr[0] : 0;

  // pass 1: max / min of z
  float mx = -INFINITY, mn = INFINITY;
  for (int i = threadIdx.x; i < V; i += blockDim.x) {
    const float z = ldz(p, i) * inv_t;
    mx = fmaxf(mx, z);
    mn = fminf(mn, z);
  }
  mx = block_max(mx, red);
  mn = -block_max(-mn, red);
  // pass 2: sum exp
  float se = 0.f;
  for (int i = threadIdx.x; i < V; i += blockDim.x) se += __expf(ldz(p, i) * inv_t - mx);
  se = block_sum(se, red);
  const float lse = mx + __logf(se);

  float thr = -INFINITY;
  if (!greedy && (k < V || tp < 1.f)) {
    float lo, hi;
    float mass_k = se;  // mass (relative to exp(-mx)) of the top-k set
    if (k < V && k >= 1) {
      // largest t with count(z >= t) >= k
      lo = mn; hi = mx;
      for (int it = 0; it < 32; ++it) {
        const float mid = 0.5f * (lo + hi);
        float cnt = 0.f;
        for (int i = threadIdx.x; i < V; i += blockDim.x) cnt += (ldz(p, i) * inv_t >= mid) ? 1.f : 0.f;
        cnt = block_sum(cnt, red);
        if (cnt >= (float)k) lo = mid; else hi = mid;
      }
      thr = lo;
      float mk = 0.f;
      for (int i = threadIdx.x; i < V; i += blockDim.x) {
        const float z = ldz(p, i) * inv_t;
        mk += z >= thr ? __expf(z - mx) : 0.f;
      }
      mass_k = block_sum(mk, red);
    }
    if (tp < 1.f) {
      // largest t >= thr with mass(z >= t) >= tp * mass_k
      lo = fmaxf(thr, mn); hi = mx;
      const float target = tp * mass_k;
      for (int it = 0; it < 32; ++it) {
        const float mid = 0.5f * (lo + hi);
        float ms = 0.f;
        for (int i = threadIdx.x; i < V; i += blockDim.x) {
          const float z = ldz(p, i) * inv_t;
          ms += z >= mid ? __expf(z - mx) : 0.f;
        }
        ms = block_sum(ms, red);
        if (ms >= target) lo = mid; else hi = mid;
      }
      thr = fmaxf(thr, lo);
    }
  }

  // final pass: arg-max of z (+ Gumbel noise unless greedy) over entries with z >= thr
  ArgMax best;
  best.v = -INFINITY;
  best.i = 0x7fffffff;
  for (int i4 = threadIdx.x * 4; i4 < V; i4 += blockDim.x * 4) {
    u32x4 r;
    if (!greedy) r = Philox::gen(seed, (uint64_t)row, step * (uint64_t)((V + 3) / 4) + (uint64_t)(i4 >> 2));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = i4 + j;
      if (i < V) {
        float z = ldz(p, i) * inv_t;
        if (z >= thr) {
          if (!greedy) z += -__logf(-__logf(u32_to_unit(r[j]) * 0.99999994f));
          ArgMax c;
          c.v = z;
          c.i = i;
          best = amax(best, c);
        }
      }
    }
  }
  best = block_argmax(best, redv, redi);
  if (threadIdx.x == 0) {
    int tok = best.i;
    if (tok < 0 || tok >= V) tok = 0;
    out_tok[row] = tok;
    if (out_logp) out_logp[row] = ldz(p, tok) * inv_t - lse;
  }
}

__global__ void bump_step_kernel(int64_t* step_ptr) {
  if (threadIdx.x == 0 && blockIdx.x == 0) step_ptr[0] += 1;
}

extern "C" int areal_sample(const void* logits, int dtype, int B, int V, int64_t ld,
                            const float* temperature, const int* top_k, const float* top_p,
                            uint64_t seed, int64_t* step_ptr, int bump_step, int64_t* out_tok,
                            float* out_logp, hipStream_t st) {
  if (B == 0) return 0;
  if (dtype == 0)
    hipLaunchKernelGGL(sample_kernel<bf16_t>, dim3(B), dim3(1024), 0, st, (const bf16_t*)logits, ld, V,
                       temperature, top_k, top_p, seed, step_ptr, out_tok, out_logp);
  else
    hipLaunchKernelGGL(sample_kernel<float>, dim3(B), dim3(1024), 0, st, (const float*)logits, ld, V,
                       temperature, top_k, top_p, seed, step_ptr, out_tok, out_logp);
  if (bump_step && step_ptr)
    hipLaunchKernelGGL(bump_step_kernel, dim3(1), dim3(64), 0, st, step_ptr);
  return (int)hipGetLastError();
}
