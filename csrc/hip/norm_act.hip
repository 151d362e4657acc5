// RMSNorm (with fused residual add), SwiGLU and rotary embedding kernels for gfx950.
//
// All three are HBM-bound, so every access is a 16-byte (8 x bf16) vector per lane
// and each row is handled by one 64-lane wave with in-wave shuffles for reductions.
// These replace the TransformerEngine/apex/flash-attn-Triton ops the reference leans
// on (realhf/impl/model/modules/mlp.py:319-341, realhf/impl/model/modules/rotary.py).
#include "common.hpp"

// ----------------------------------------------------------------------------------
// RMSNorm forward.  y = (x [+ res]) * rsqrt(mean((x+res)^2) + eps) * (w [+ 1 if gemma])
// When `res` is given, the sum x+res is written to `res_out` (the new residual stream).
// One wave per row; H % 8 == 0.
// ----------------------------------------------------------------------------------
template <int MAXV>  // number of 8-element vectors each lane keeps in registers
__global__ __launch_bounds__(256) void rmsnorm_fwd_kernel(
    const bf16_t* __restrict__ x, const bf16_t* __restrict__ res, const bf16_t* __restrict__ w,
    bf16_t* __restrict__ y, bf16_t* __restrict__ res_out, float* __restrict__ rstd_out,
    int T, int H, float eps, int gemma) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= T) return;
  const int nvec = H >> 3;
  const bf16_t* xr = x + (size_t)row * H;
  const bf16_t* rr = res ? res + (size_t)row * H : nullptr;
  float v[MAXV][8];
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + i * 64;
    if (c < nvec) {
      bf16x8 a = ld8(xr + c * 8);
      if (rr) {
        bf16x8 b = ld8(rr + c * 8);
        bf16x8 s;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          // the residual stream is stored in bf16, so round before normalising: the
          // backward pass re-reads the rounded value.
          s[j] = f2bf(bf2f(a[j]) + bf2f(b[j]));
          v[i][j] = bf2f(s[j]);
        }
        st8(res_out + (size_t)row * H + c * 8, s);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[i][j] = bf2f(a[j]);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) ss += v[i][j] * v[i][j];
    }
  }
  ss = wave_sum(ss);
  const float rstd = rsqrtf(ss / (float)H + eps);
  if (lane == 0 && rstd_out) rstd_out[row] = rstd;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + i * 64;
    if (c < nvec) {
      bf16x8 ww = ld8(w + c * 8);
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float wf = bf2f(ww[j]) + (gemma ? 1.f : 0.f);
        o[j] = f2bf(v[i][j] * rstd * wf);
      }
      st8(y + (size_t)row * H + c * 8, o);
    }
  }
}

// RMSNorm backward. dx = rstd * (g - xhat * mean(g * xhat)) [+ dres], g = dy * w.
// dw partials: each block accumulates its rows' dy*xhat per column in registers,
// reduces its 4 waves through LDS and writes one partial row; a second kernel sums
// the partial rows (no atomics: every block would hit the same H floats).
template <int MAXV>
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(
    const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
    const float* __restrict__ rstd, const bf16_t* __restrict__ dres, bf16_t* __restrict__ dx,
    float* __restrict__ dw_partial, int T, int H, int gemma) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int nw = blockDim.x >> 6;
  const int nvec = H >> 3;
  float dwacc[MAXV][8];
#pragma unroll
  for (int i = 0; i < MAXV; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) dwacc[i][j] = 0.f;

  for (int row = blockIdx.x * nw + wid; row < T; row += gridDim.x * nw) {
    const float rs = rstd[row];
    float g[MAXV][8], xh[MAXV][8];
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = lane + i * 64;
      if (c < nvec) {
        bf16x8 a = ld8(dy + (size_t)row * H + c * 8);
        bf16x8 b = ld8(x + (size_t)row * H + c * 8);
        bf16x8 ww = ld8(w + c * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float dyf = bf2f(a[j]);
          xh[i][j] = bf2f(b[j]) * rs;
          g[i][j] = dyf * (bf2f(ww[j]) + (gemma ? 1.f : 0.f));
          dot += g[i][j] * xh[i][j];
          dwacc[i][j] += dyf * xh[i][j];
        }
      }
    }
    dot = wave_sum(dot) / (float)H;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = lane + i * 64;
      if (c < nvec) {
        bf16x8 o;
        bf16x8 dr;
        if (dres) dr = ld8(dres + (size_t)row * H + c * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float d = rs * (g[i][j] - xh[i][j] * dot);
          if (dres) d += bf2f(dr[j]);
          o[j] = f2bf(d);
        }
        st8(dx + (size_t)row * H + c * 8, o);
      }
    }
  }
  // cross-wave reduction of dw partials through LDS: smem[nw][H]
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + i * 64;
    if (c < nvec) {
#pragma unroll
      for (int j = 0; j < 8; ++j) smem[(size_t)wid * H + c * 8 + j] = dwacc[i][j];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < H; c += blockDim.x) {
    float s = 0.f;
    for (int k = 0; k < nw; ++k) s += smem[(size_t)k * H + c];
    dw_partial[(size_t)blockIdx.x * H + c] = s;
  }
}

// Second stage of the weight-gradient reduction: column sums of the G partial rows. A block covers 32
// columns x 8 row slices, so H / 32 blocks are in flight and a thread issues only G / 8 loads, four of them
// independent at a time (one thread per column over all G rows is a chain of G dependent-latency loads on
// a handful of blocks: 124 us for G = 512, H = 1536).
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ part,
                                                            bf16_t* __restrict__ out_bf16,
                                                            float* __restrict__ out_f32, int G, int H) {
  __shared__ float red[8][33];
  const int cx = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cx;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < H) {
    int g = sl;
    for (; g + 24 < G; g += 32) {
      s0 += part[(size_t)g * H + c];
      s1 += part[(size_t)(g + 8) * H + c];
      s2 += part[(size_t)(g + 16) * H + c];
      s3 += part[(size_t)(g + 24) * H + c];
    }
    for (; g < G; g += 8) s0 += part[(size_t)g * H + c];
  }
  red[sl][cx] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (sl == 0 && c < H) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += red[i][cx];
    if (out_bf16) out_bf16[c] = f2bf(s);
    if (out_f32) out_f32[c] = s;
  }
}

extern "C" int areal_rmsnorm_fwd(const void* x, const void* res, const void* w, void* y, void* res_out,
                                 float* rstd, int T, int H, float eps, int gemma, hipStream_t st) {
  if (H % 8 != 0 || H > 8 * 64 * 16) return -1;
  if (T == 0) return 0;
  const int rows_per_block = 4;
  dim3 grid((T + rows_per_block - 1) / rows_per_block), block(256);
  const int nvec = H / 8;
  const int maxv = (nvec + 63) / 64;
#define LAUNCH(MV)                                                                              \
  hipLaunchKernelGGL(rmsnorm_fwd_kernel<MV>, grid, block, 0, st, (const bf16_t*)x,              \
                     (const bf16_t*)res, (const bf16_t*)w, (bf16_t*)y, (bf16_t*)res_out, rstd,  \
                     T, H, eps, gemma)
  if (maxv <= 2) LAUNCH(2);
  else if (maxv <= 4) LAUNCH(4);
  else if (maxv <= 8) LAUNCH(8);
  else LAUNCH(16);
#undef LAUNCH
  return (int)hipGetLastError();
}

// workspace: G*H floats where G = areal_rmsnorm_bwd_nblocks(T)
extern "C" int areal_rmsnorm_bwd_nblocks(int T) {
  int g = (T + 3) / 4;
  return g < 1 ? 1 : (g > 512 ? 512 : g);
}

extern "C" int areal_rmsnorm_bwd(const void* dy, const void* x, const void* w, const float* rstd,
                                 const void* dres, void* dx, void* dw_bf16, float* dw_f32,
                                 float* workspace, int T, int H, int gemma, hipStream_t st) {
  if (H % 8 != 0 || H > 8 * 64 * 16) return -1;
  if (T == 0) return 0;
  const int G = areal_rmsnorm_bwd_nblocks(T);
  const size_t lds = (size_t)4 * H * sizeof(float);
  if (lds > 160 * 1024) return -2;
  dim3 grid(G), block(256);
  const int maxv = (H / 8 + 63) / 64;
#define LAUNCH(MV)                                                                             \
  do {                                                                                         \
    if (lds > 64 * 1024)                                                                       \
      HIP_RET(hipFuncSetAttribute((const void*)rmsnorm_bwd_kernel<MV>,                         \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));      \
    hipLaunchKernelGGL(rmsnorm_bwd_kernel<MV>, grid, block, lds, st, (const bf16_t*)dy,        \
                       (const bf16_t*)x, (const bf16_t*)w, rstd, (const bf16_t*)dres,          \
                       (bf16_t*)dx, workspace, T, H, gemma);                                   \
  } while (0)
  if (maxv <= 2) LAUNCH(2);
  else if (maxv <= 4) LAUNCH(4);
  else if (maxv <= 8) LAUNCH(8);
  else LAUNCH(16);
#undef LAUNCH
  hipLaunchKernelGGL(colsum_partial_kernel, dim3((H + 31) / 32), dim3(256), 0, st, workspace,
                     (bf16_t*)dw_bf16, dw_f32, G, H);
  return (int)hipGetLastError();
}

// ----------------------------------------------------------------------------------
// Gated linear unit on a fused [T, 2F] gate|up projection: out[t, f] = act(gu[t, f]) * gu[t, F + f]
// ACT 0: SiLU (SwiGLU - Llama / Qwen / Mistral); ACT 1: tanh-approximated GELU (GeGLU - Gemma).
// ----------------------------------------------------------------------------------
template <int ACT>
DEVINL void glu_act(float g, float& a, float& da) {
  if (ACT == 0) {
    const float sig = 1.f / (1.f + __expf(-g));
    a = g * sig;
    da = sig * (1.f + g * (1.f - sig));
  } else {
    const float k = 0.7978845608028654f, c = 0.044715f;
    const float inner = k * (g + c * g * g * g);
    const float e = __expf(-2.f * fabsf(inner));       // tanh via exp, stable for large |inner|
    const float th = copysignf((1.f - e) / (1.f + e), inner);
    a = 0.5f * g * (1.f + th);
    da = 0.5f * (1.f + th) + 0.5f * g * (1.f - th * th) * k * (1.f + 3.f * c * g * g);
  }
}

template <int ACT>
__global__ __launch_bounds__(256) void swiglu_fwd_kernel(const bf16_t* __restrict__ gu,
                                                         bf16_t* __restrict__ out, int64_t T, int F) {
  const int nvec_row = F >> 3;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nvec_row) return;
  for (int64_t t = blockIdx.y; t < T; t += gridDim.y) {
    bf16x8 g = ld8(gu + t * 2 * F + c * 8);
    bf16x8 u = ld8(gu + t * 2 * F + F + c * 8);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float a, da;
      glu_act<ACT>(bf2f(g[j]), a, da);
      o[j] = f2bf(a * bf2f(u[j]));
    }
    st8(out + t * F + c * 8, o);
  }
}

template <int ACT>
__global__ __launch_bounds__(256) void swiglu_bwd_kernel(const bf16_t* __restrict__ dy,
                                                         const bf16_t* __restrict__ gu,
                                                         bf16_t* __restrict__ dgu, int64_t T, int F) {
  const int nvec_row = F >> 3;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nvec_row) return;
  for (int64_t t = blockIdx.y; t < T; t += gridDim.y) {
    bf16x8 g = ld8(gu + t * 2 * F + c * 8);
    bf16x8 u = ld8(gu + t * 2 * F + F + c * 8);
    bf16x8 d = ld8(dy + t * F + c * 8);
    bf16x8 dg, du;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float uf = bf2f(u[j]), df = bf2f(d[j]);
      float a, da;
      glu_act<ACT>(bf2f(g[j]), a, da);
      dg[j] = f2bf(df * uf * da);
      du[j] = f2bf(df * a);
    }
    st8(dgu + t * 2 * F + c * 8, dg);
    st8(dgu + t * 2 * F + F + c * 8, du);
  }
}

static inline int ew_grid(int64_t total, int block) {
  int64_t g = (total + block - 1) / block;
  if (g > 256 * 8) g = 256 * 8;
  if (g < 1) g = 1;
  return (int)g;
}

extern "C" int areal_glu_fwd(const void* gu, void* out, int64_t T, int F, int act, hipStream_t st) {
  if (F % 8 != 0) return -1;
  if (act != 0 && act != 1) return -2;
  if (T == 0) return 0;
  const dim3 grid((F / 8 + 255) / 256, (int)(T < 8192 ? T : 8192));
  if (act == 0)
    hipLaunchKernelGGL(swiglu_fwd_kernel<0>, grid, dim3(256), 0, st, (const bf16_t*)gu, (bf16_t*)out, T, F);
  else
    hipLaunchKernelGGL(swiglu_fwd_kernel<1>, grid, dim3(256), 0, st, (const bf16_t*)gu, (bf16_t*)out, T, F);
  return (int)hipGetLastError();
}
extern "C" int areal_glu_bwd(const void* dy, const void* gu, void* dgu, int64_t T, int F, int act,
                             hipStream_t st) {
  if (F % 8 != 0) return -1;
  if (act != 0 && act != 1) return -2;
  if (T == 0) return 0;
  const dim3 grid((F / 8 + 255) / 256, (int)(T < 8192 ? T : 8192));
  if (act == 0)
    hipLaunchKernelGGL(swiglu_bwd_kernel<0>, grid, dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)gu,
                       (bf16_t*)dgu, T, F);
  else
    hipLaunchKernelGGL(swiglu_bwd_kernel<1>, grid, dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)gu,
                       (bf16_t*)dgu, T, F);
  return (int)hipGetLastError();
}
extern "C" int areal_swiglu_fwd(const void* gu, void* out, int64_t T, int F, hipStream_t st) {
  return areal_glu_fwd(gu, out, T, F, 0, st);
}
extern "C" int areal_swiglu_bwd(const void* dy, const void* gu, void* dgu, int64_t T, int F,
                                hipStream_t st) {
  return areal_glu_bwd(dy, gu, dgu, T, F, 0, st);
}

// ----------------------------------------------------------------------------------
// Rotary embedding (half-split / "NeoX" layout, as HF Llama/Qwen use), in place on the
// first `n_rot_heads` heads of a [T, n_heads_total, D] tensor (the fused qkv projection:
// q heads then k heads are rotated, v heads are left alone). cos/sin come from a host-
// precomputed fp32 table [max_pos, D/2] so no transcendental runs on device.
// `sign` = +1 forward, -1 backward (the inverse rotation is the gradient).
// Each group of D/16 lanes handles one head: a lane rotates 8 (lo, hi) pairs.
// ----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rope_kernel(bf16_t* __restrict__ x,
                                                   const int* __restrict__ pos,
                                                   const float* __restrict__ cos_t,
                                                   const float* __restrict__ sin_t, int64_t T,
                                                   int n_heads_total, int n_rot_heads, int D,
                                                   float sign, int max_pos) {
  const int half = D >> 1;
  const int lanes_per_head = half >> 3;  // 8 for D=128
  const int per_tok = n_rot_heads * lanes_per_head;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;  // (head, sub) of one token
  if (i >= per_tok) return;
  const int head = i / lanes_per_head, sub = i - head * lanes_per_head;
  for (int64_t t = blockIdx.y; t < T; t += gridDim.y) {
    int p = pos[t];
    p = p < 0 ? 0 : (p >= max_pos ? max_pos - 1 : p);
    bf16_t* base = x + (t * n_heads_total + head) * D + sub * 8;
    bf16x8 lo = ld8(base), hi = ld8(base + half);
    const float* c = cos_t + (size_t)p * half + sub * 8;
    const float* s = sin_t + (size_t)p * half + sub * 8;
    f32x4 c0 = *reinterpret_cast<const f32x4*>(c), c1 = *reinterpret_cast<const f32x4*>(c + 4);
    f32x4 s0 = *reinterpret_cast<const f32x4*>(s), s1 = *reinterpret_cast<const f32x4*>(s + 4);
    bf16x8 olo, ohi;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float cf = j < 4 ? c0[j] : c1[j - 4];
      const float sf = sign * (j < 4 ? s0[j] : s1[j - 4]);
      const float a = bf2f(lo[j]), b = bf2f(hi[j]);
      olo[j] = f2bf(a * cf - b * sf);
      ohi[j] = f2bf(b * cf + a * sf);
    }
    st8(base, olo);
    st8(base + half, ohi);
  }
}

extern "C" int areal_rope_inplace(void* x, const int* pos, const float* cos_t, const float* sin_t,
                                  int64_t T, int n_heads_total, int n_rot_heads, int D, float sign,
                                  int max_pos, hipStream_t st) {
  if (D % 16 != 0) return -1;
  if (T == 0) return 0;
  const int per_tok = n_rot_heads * (D / 16);
  const int bs = per_tok >= 256 ? 256 : ((per_tok + 63) / 64) * 64;
  const int gx = (per_tok + bs - 1) / bs;
  const int gy = (int)(T < 16384 ? T : 16384);
  hipLaunchKernelGGL(rope_kernel, dim3(gx, gy), dim3(bs), 0, st, (bf16_t*)x, pos, cos_t, sin_t, T,
                     n_heads_total, n_rot_heads, D, sign, max_pos);
  return (int)hipGetLastError();
}

// ----------------------------------------------------------------------------------
// Per-head RMSNorm of the q and k heads inside a packed projection output qkv [T, Hq + 2 Hkv, D] (Qwen3): one pass
// that normalises the q rows with one weight vector, the k rows with another and copies the v rows, instead of
// slicing the heads out (two strided copies), normalising them and concatenating the result (a third pass), forward
// and backward. A row is D = 64 / 128 / 256 values: D / 8 lanes hold it in 16-byte pieces, a wave takes 64 / (D / 8)
// rows at a time. rstd [T, Hq + Hkv] is kept for the backward pass.
//   y = x * rstd * w,  rstd = rsqrt(mean(x^2) + eps)
//   dx = rstd * (g - xh * mean(g * xh)),  g = dy * w,  xh = x * rstd;   dw = sum over rows of dy * xh
// ----------------------------------------------------------------------------------
template <int LPR>  // lanes per row = D / 8
DEVINL float row_sum(float v) {
#pragma unroll
  for (int o = LPR >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <int LPR>
__global__ __launch_bounds__(256) void qk_norm_fwd_kernel(const bf16_t* __restrict__ qkv,
                                                          const bf16_t* __restrict__ qw,
                                                          const bf16_t* __restrict__ kw, bf16_t* __restrict__ out,
                                                          float* __restrict__ rstd, int64_t rows, int Hq, int Hkv,
                                                          float eps) {
  constexpr int D = LPR * 8, RPW = 64 / LPR;
  const int lane = threadIdx.x & 63, sub = lane % LPR;
  const int Ht = Hq + 2 * Hkv, Hn = Hq + Hkv;
  const int64_t r = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + lane / LPR;
  if (r >= rows) return;
  const int h = (int)(r % Ht);
  const int64_t t = r / Ht;
  const bf16x8 v = ld8(qkv + r * D + sub * 8);
  if (h >= Hn) {  // v head: unchanged
    st8(out + r * D + sub * 8, v);
    return;
  }
  float x[8], ss = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    x[j] = bf2f(v[j]);
    ss += x[j] * x[j];
  }
  // rows of one wave may be q, k or v rows: the shuffles stay inside the LPR lanes of a row, and the v rows' lanes
  // above have left - reduce with the lanes that remain (xor partners inside a row are all of the same kind)
  ss = row_sum<LPR>(ss);
  const float rs = rsqrtf(ss / (float)D + eps);
  const bf16x8 w = ld8((h < Hq ? qw : kw) + sub * 8);
  bf16x8 y;
#pragma unroll
  for (int j = 0; j < 8; ++j) y[j] = f2bf(x[j] * rs * bf2f(w[j]));
  st8(out + r * D + sub * 8, y);
  if (sub == 0) rstd[t * Hn + h] = rs;
}

// dw partial sums: part [gridDim.x][2][D] (q weight, k weight)
template <int LPR>
__global__ __launch_bounds__(256) void qk_norm_bwd_kernel(const bf16_t* __restrict__ dy,
                                                          const bf16_t* __restrict__ qkv,
                                                          const bf16_t* __restrict__ qw,
                                                          const bf16_t* __restrict__ kw,
                                                          const float* __restrict__ rstd, bf16_t* __restrict__ dx,
                                                          float* __restrict__ part, int64_t rows, int Hq, int Hkv) {
  constexpr int D = LPR * 8, RPW = 64 / LPR;
  __shared__ float red[4][2][D];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, sub = lane % LPR;
  const int Ht = Hq + 2 * Hkv, Hn = Hq + Hkv;
  float wq[8], wk[8], aq[8], ak[8];
  {
    const bf16x8 a = ld8(qw + sub * 8), b = ld8(kw + sub * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      wq[j] = bf2f(a[j]);
      wk[j] = bf2f(b[j]);
      aq[j] = ak[j] = 0.f;
    }
  }
  const int64_t per_block = 4 * RPW;
  for (int64_t r0 = (int64_t)blockIdx.x * per_block; r0 < rows; r0 += (int64_t)gridDim.x * per_block) {
    const int64_t r = r0 + wid * RPW + lane / LPR;
    const bool live = r < rows;
    const int h = live ? (int)(r % Ht) : Ht;
    const int64_t t = live ? r / Ht : 0;
    bf16x8 g8, x8;
    if (live) {
      g8 = ld8(dy + r * D + sub * 8);
      if (h < Hn) x8 = ld8(qkv + r * D + sub * 8);
    }
    if (live && h >= Hn) st8(dx + r * D + sub * 8, g8);  // v head: gradient passes through
    const bool normed = live && h < Hn;
    const float rs = normed ? rstd[t * Hn + h] : 0.f;
    float g[8], xh[8], dot = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float d = normed ? bf2f(g8[j]) : 0.f;
      xh[j] = normed ? bf2f(x8[j]) * rs : 0.f;
      g[j] = d * (h < Hq ? wq[j] : wk[j]);
      dot += g[j] * xh[j];
      if (normed) {
        if (h < Hq) aq[j] += d * xh[j]; else ak[j] += d * xh[j];
      }
    }
    dot = row_sum<LPR>(dot) / (float)D;  // every lane of the wave takes part (zeros for rows that are not normed)
    if (normed) {
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = f2bf(rs * (g[j] - xh[j] * dot));
      st8(dx + r * D + sub * 8, o);
    }
  }
  // lanes with the same `sub` hold the same columns: sum them inside the wave, then over the four waves
#pragma unroll
  for (int j = 0; j < 8; ++j) {
#pragma unroll
    for (int o = LPR; o < 64; o <<= 1) {
      aq[j] += __shfl_xor(aq[j], o, 64);
      ak[j] += __shfl_xor(ak[j], o, 64);
    }
  }
  if (lane < LPR) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      red[wid][0][sub * 8 + j] = aq[j];
      red[wid][1][sub * 8 + j] = ak[j];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * D; c += 256) {
    const int which = c / D, col = c % D;
    part[(size_t)blockIdx.x * 2 * D + c] = red[0][which][col] + red[1][which][col] + red[2][which][col] + red[3][which][col];
  }
}

extern "C" int areal_qk_norm_bwd_nblocks(int64_t rows) {
  const int64_t g = (rows + 63) / 64;
  return (int)(g < 1 ? 1 : (g > 1024 ? 1024 : g));
}

extern "C" int areal_qk_norm_fwd(const void* qkv, const void* qw, const void* kw, void* out, float* rstd, int64_t T,
                                 int Hq, int Hkv, int D, float eps, hipStream_t st) {
  if (T == 0) return 0;
  if (D != 64 && D != 128 && D != 256) return -1;
  const int64_t rows = T * (Hq + 2 * Hkv);
  const int rpw = 64 / (D / 8);
  const dim3 grid((unsigned)((rows + 4 * rpw - 1) / (4 * rpw))), block(256);
#define QKN(L_) hipLaunchKernelGGL(qk_norm_fwd_kernel<L_>, grid, block, 0, st, (const bf16_t*)qkv, (const bf16_t*)qw, \
                                   (const bf16_t*)kw, (bf16_t*)out, rstd, rows, Hq, Hkv, eps)
  if (D == 64) QKN(8); else if (D == 128) QKN(16); else QKN(32);
#undef QKN
  return (int)hipGetLastError();
}

// workspace: areal_qk_norm_bwd_nblocks(rows) * 2 * D floats. dw_q / dw_k: bf16 [D] each (one buffer [2, D]).
extern "C" int areal_qk_norm_bwd(const void* dy, const void* qkv, const void* qw, const void* kw, const float* rstd,
                                 void* dx, void* dw_bf16, float* workspace, int64_t T, int Hq, int Hkv, int D,
                                 hipStream_t st) {
  if (T == 0) return 0;
  if (D != 64 && D != 128 && D != 256) return -1;
  const int64_t rows = T * (Hq + 2 * Hkv);
  const int G = areal_qk_norm_bwd_nblocks(rows);
  const dim3 grid(G), block(256);
#define QKN(L_) hipLaunchKernelGGL(qk_norm_bwd_kernel<L_>, grid, block, 0, st, (const bf16_t*)dy, (const bf16_t*)qkv, \
                                   (const bf16_t*)qw, (const bf16_t*)kw, rstd, (bf16_t*)dx, workspace, rows, Hq, Hkv)
  if (D == 64) QKN(8); else if (D == 128) QKN(16); else QKN(32);
#undef QKN
  hipLaunchKernelGGL(colsum_partial_kernel, dim3((2 * D + 31) / 32), dim3(256), 0, st, workspace, (bf16_t*)dw_bf16,
                     (float*)nullptr, G, 2 * D);
  return (int)hipGetLastError();
}
