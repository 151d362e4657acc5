// Skinny-M GEMM for decode: Y[M, N] = A[M, K] * W[N, K]^T with M <= a few hundred rows.
//
// At decode batch sizes the weight matrix is read exactly once and everything else is noise, so the
// kernel is built around streaming W from HBM: every wave owns 16*NT weight rows, loads its W and A
// fragments straight from global memory into MFMA operand registers (both operands are K-contiguous,
// so a lane's 16-byte load IS its mfma_f32_16x16x32_bf16 fragment; a wave instruction covers 16 rows
// x 64 contiguous bytes), keeps a DEPTH-deep register prefetch pipeline and never touches LDS or a
// barrier. A is shared by all waves and comes from L2 / L1.
//
// The product is computed transposed (D = W_tile * A_tile^T) so that a lane ends up with 4 consecutive
// output columns of one row: 8-byte bf16 stores.
//
// Fusions (each removes an elementwise kernel and its launch from the decode step):
//   prologue NORM      A = rmsnorm(x) = x * rsqrt(rowsumsq/K + eps) * w   (rowsumsq comes from the
//                      producer's epilogue, so no pass over the row is needed here)
//   epilogue RESIDUAL  y = bf16(acc + residual); stored as the new residual stream; per-row sum of
//                      squares of y accumulated with fp32 atomics for the next NORM prologue
//   epilogue SWIGLU    wave owns gate tile j and up tile j: out = silu(gate) * up, [M, N/2]
//   split-K            partial sums go to an fp32 workspace with atomics; the last wave to arrive on a
//                      tile (ticket counter) reloads the sums, runs the epilogue and re-zeroes the tile
#include "common.hpp"

namespace {

enum { PRO_NONE = 0, PRO_NORM = 1 };
enum { EPI_STORE = 0, EPI_RESIDUAL = 1, EPI_SWIGLU = 2 };

struct SkinnyArgs {
  const bf16_t* X;
  long xs;
  const bf16_t* W;
  const bf16_t* bias;
  bf16_t* Y;
  long ys;
  int M, N, K;
  const float* rss_in;
  const bf16_t* norm_w;
  float eps;
  int gemma;
  const bf16_t* R;
  long rs;
  float* rss_out;
  int ksplit;
  int k_per_split;
  float* ws;
  int* counters;
  int n_groups;
};

DEVINL float silu(float x) { return x / (1.f + __expf(-x)); }

template <int MT, int NT, int DEPTH, int PRO, int EPI>
__global__ __launch_bounds__(256) void skinny_gemm_kernel(const SkinnyArgs a) {
  const int lane = threadIdx.x & 63;
  const int ng = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ng >= a.n_groups) return;
  const int split = blockIdx.y;
  const int mb = blockIdx.z;  // block of 16*MT rows
  const int r = lane & 15, q = lane >> 4;
  const int K = a.K;
  const int k_begin = split * a.k_per_split;
  const int k_end = min(K, k_begin + a.k_per_split);
  const int nsteps = (k_end - k_begin) >> 5;

  // weight-row base of every n-tile of this wave
  int ncol[NT];
  if (EPI == EPI_SWIGLU) {
    ncol[0] = ng * 16;
    ncol[NT - 1] = a.N / 2 + ng * 16;
  } else {
#pragma unroll
    for (int t = 0; t < NT; ++t) ncol[t] = (ng * NT + t) * 16;
  }
  const bf16_t* wp[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) wp[t] = a.W + (size_t)(ncol[t] + r) * K + k_begin + 8 * q;
  const int m0 = mb * 16 * MT;
  const bf16_t* xp[MT];
  float rstd[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    int row = m0 + t * 16 + r;
    row = row < a.M ? row : a.M - 1;
    xp[t] = a.X + (size_t)row * a.xs + k_begin + 8 * q;
    if (PRO == PRO_NORM) rstd[t] = rsqrtf(a.rss_in[row] / (float)K + a.eps);
  }
  const bf16_t* nwp = (PRO == PRO_NORM) ? a.norm_w + k_begin + 8 * q : nullptr;

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  bf16x8 wf[DEPTH][NT], af[DEPTH][MT], nf[DEPTH];
  auto issue = [&](int slot, int step) {
    const int ko = step << 5;
#pragma unroll
    for (int t = 0; t < NT; ++t) wf[slot][t] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wp[t] + ko));
#pragma unroll
    for (int i = 0; i < MT; ++i) af[slot][i] = ld8(xp[i] + ko);
    if (PRO == PRO_NORM) nf[slot] = ld8(nwp + ko);
  };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
    if (d < nsteps) issue(d, d);

  for (int base = 0; base < nsteps; base += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int step = base + d;
      if (step < nsteps) {
        bf16x8 av[MT];
        if (PRO == PRO_NORM) {
          float wv[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) wv[j] = bf2f(nf[d][j]) + (a.gemma ? 1.f : 0.f);
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) av[i][j] = f2bf(bf2f(af[d][i][j]) * rstd[i] * wv[j]);
        } else {
#pragma unroll
          for (int i = 0; i < MT; ++i) av[i] = af[d][i];
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int t = 0; t < NT; ++t)
            acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[d][t], av[i], acc[i][t], 0, 0, 0);
        if (step + DEPTH < nsteps) issue(d, step + DEPTH);
      }
    }
  }

  // acc[i][t][j] = sum for row m0 + 16 i + r, column ncol[t] + 4 q + j
  if (a.ksplit > 1) {
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int row = m0 + i * 16 + r;
      if (row < a.M) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            unsafeAtomicAdd(a.ws + (size_t)row * a.N + ncol[t] + 4 * q + j, acc[i][t][j]);
      }
    }
    __threadfence();
    int ticket = 0;
    int* ctr = a.counters + mb * a.n_groups + ng;
    if (lane == 0) ticket = atomicAdd(ctr, 1);
    ticket = __shfl(ticket, 0, 64);
    if (ticket != a.ksplit - 1) return;
    __threadfence();
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int row = m0 + i * 16 + r;
      if (row < a.M) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float* p = a.ws + (size_t)row * a.N + ncol[t] + 4 * q + j;
            acc[i][t][j] = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *p = 0.f;
          }
      }
    }
    if (lane == 0) *ctr = 0;
  }

#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int row = m0 + i * 16 + r;
    float ss = 0.f;
    if (row < a.M) {
      if (EPI == EPI_SWIGLU) {
        bf16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          // round gate / up to bf16 first: same values the unfused bf16 GEMM output would carry
          const float g = bf2f(f2bf(acc[i][0][j])), u = bf2f(f2bf(acc[i][NT - 1][j]));
          o[j] = f2bf(silu(g) * u);
        }
        *reinterpret_cast<bf16x4*>(a.Y + (size_t)row * a.ys + ncol[0] + 4 * q) = o;
      } else {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int n = ncol[t] + 4 * q;
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = acc[i][t][j];
          if (a.bias) {
            const bf16x4 b = *reinterpret_cast<const bf16x4*>(a.bias + n);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += bf2f(b[j]);
          }
          bf16x4 o;
          if (EPI == EPI_RESIDUAL) {
            const bf16x4 rr = *reinterpret_cast<const bf16x4*>(a.R + (size_t)row * a.rs + n);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              // the GEMM output is a bf16 tensor in the unfused path: round before the residual add
              o[j] = f2bf(bf2f(f2bf(v[j])) + bf2f(rr[j]));
              const float y = bf2f(o[j]);
              ss += y * y;
            }
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = f2bf(v[j]);
          }
          *reinterpret_cast<bf16x4*>(a.Y + (size_t)row * a.ys + n) = o;
        }
      }
    }
    if (EPI == EPI_RESIDUAL) {
      // lanes r, r+16, r+32, r+48 hold the same row
      ss += __shfl_xor(ss, 16, 64);
      ss += __shfl_xor(ss, 32, 64);
      if (q == 0 && row < a.M) unsafeAtomicAdd(a.rss_out + row, ss);
    }
  }
}

// Embedding gather + per-row sum of squares; also clears the other layers' sum-of-squares slots.
__global__ __launch_bounds__(256) void embed_rss_kernel(const long* __restrict__ ids, const bf16_t* __restrict__ table,
                                                        bf16_t* __restrict__ out, float* __restrict__ rss0,
                                                        float* __restrict__ rss_clear, long n_clear, int M, int H,
                                                        float scale) {
  const long gtid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  for (long i = gtid; i < n_clear; i += (long)gridDim.x * blockDim.x) rss_clear[i] = 0.f;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const bf16_t* src = table + (size_t)ids[row] * H;
  float ss = 0.f;
  for (int c = lane * 8; c < H; c += 64 * 8) {
    bf16x8 v = ld8(src + c);
    if (scale != 1.f) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = f2bf(bf2f(v[j]) * scale);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) ss += bf2f(v[j]) * bf2f(v[j]);
    st8(out + (size_t)row * H + c, v);
  }
  ss = wave_sum(ss);
  if (lane == 0) rss0[row] = ss;
}

template <int MT, int NT, int DEPTH>
int launch_mt(const SkinnyArgs& a, int pro, int epi, dim3 grid, hipStream_t st) {
#define GO(P, E) hipLaunchKernelGGL((skinny_gemm_kernel<MT, NT, DEPTH, P, E>), grid, dim3(256), 0, st, a)
  if (pro == PRO_NONE && epi == EPI_STORE) GO(PRO_NONE, EPI_STORE);
  else if (pro == PRO_NORM && epi == EPI_STORE) GO(PRO_NORM, EPI_STORE);
  else if (pro == PRO_NONE && epi == EPI_RESIDUAL) GO(PRO_NONE, EPI_RESIDUAL);
  else if (pro == PRO_NORM && epi == EPI_SWIGLU) GO(PRO_NORM, EPI_SWIGLU);
  else if (pro == PRO_NONE && epi == EPI_SWIGLU) GO(PRO_NONE, EPI_SWIGLU);
  else return -10;
#undef GO
  return (int)hipGetLastError();
}

}  // namespace

// M rows (any), N % (16*nt) == 0, K % 32 == 0, k split so that every split is a multiple of 32.
// `nt`: weight tiles per wave (1 or 2; SWIGLU forces 2). ws: [M, N] fp32 zeroed, counters zeroed.
extern "C" int areal_skinny_gemm(const void* X, long xs, const void* W, const void* bias, void* Y, long ys, int M,
                                 int N, int K, int prologue, const float* rss_in, const void* norm_w, float eps,
                                 int gemma, int epilogue, const void* R, long rs, float* rss_out, int ksplit,
                                 float* ws, int* counters, int nt, hipStream_t st) {
  if (M <= 0) return 0;
  if (K % 32 != 0 || xs % 8 != 0 || ys % 4 != 0) return -1;
  if (epilogue == EPI_SWIGLU) nt = 2;
  if (nt != 1 && nt != 2) return -2;
  const int cols_per_wave = (epilogue == EPI_SWIGLU) ? 16 : 16 * nt;
  const int n_eff = (epilogue == EPI_SWIGLU) ? N / 2 : N;
  if (n_eff % cols_per_wave != 0) return -3;
  if (ksplit < 1) ksplit = 1;
  int kps = ((K / 32 + ksplit - 1) / ksplit) * 32;
  ksplit = (K + kps - 1) / kps;
  if (ksplit > 1 && (!ws || !counters)) return -4;
  if (prologue == PRO_NORM && (!rss_in || !norm_w)) return -5;
  if (epilogue == EPI_RESIDUAL && (!R || !rss_out)) return -6;
  SkinnyArgs a;
  a.X = (const bf16_t*)X; a.xs = xs; a.W = (const bf16_t*)W; a.bias = (const bf16_t*)bias;
  a.Y = (bf16_t*)Y; a.ys = ys; a.M = M; a.N = N; a.K = K;
  a.rss_in = rss_in; a.norm_w = (const bf16_t*)norm_w; a.eps = eps; a.gemma = gemma;
  a.R = (const bf16_t*)R; a.rs = rs; a.rss_out = rss_out;
  a.ksplit = ksplit; a.k_per_split = kps; a.ws = ws; a.counters = counters;
  a.n_groups = n_eff / cols_per_wave;
  const int mt = M <= 16 ? 1 : (M <= 32 ? 2 : (M <= 64 ? 4 : 8));
  const int mblocks = (M + 16 * mt - 1) / (16 * mt);
  dim3 grid((a.n_groups + 3) / 4, ksplit, mblocks);
  if (nt == 1) {
    switch (mt) {
      case 1: return launch_mt<1, 1, 8>(a, prologue, epilogue, grid, st);
      case 2: return launch_mt<2, 1, 8>(a, prologue, epilogue, grid, st);
      case 4: return launch_mt<4, 1, 6>(a, prologue, epilogue, grid, st);
      default: return launch_mt<8, 1, 4>(a, prologue, epilogue, grid, st);
    }
  }
  switch (mt) {
    case 1: return launch_mt<1, 2, 8>(a, prologue, epilogue, grid, st);
    case 2: return launch_mt<2, 2, 8>(a, prologue, epilogue, grid, st);
    case 4: return launch_mt<4, 2, 6>(a, prologue, epilogue, grid, st);
    default: return launch_mt<8, 2, 3>(a, prologue, epilogue, grid, st);
  }
}

extern "C" int areal_embed_rss(const void* ids, const void* table, void* out, float* rss0, float* rss_clear,
                               long n_clear, int M, int H, float scale, hipStream_t st) {
  if (M <= 0) return 0;
  if (H % 8 != 0) return -1;
  hipLaunchKernelGGL(embed_rss_kernel, dim3((M + 3) / 4), dim3(256), 0, st, (const long*)ids, (const bf16_t*)table,
                     (bf16_t*)out, rss0, rss_clear, n_clear, M, H, scale);
  return (int)hipGetLastError();
}
