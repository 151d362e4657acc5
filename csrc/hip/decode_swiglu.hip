// gate|up projection of a decode step with the SwiGLU activation in its epilogue, decomposed so that EVERY CU of the
// chip gets one workgroup:   Y[M, F] = silu(X W_gate^T) * (X W_up^T),  W = [gate; up] [2F, K],  M = 64 ... 512 rows.
//
// What a workgroup of such a GEMM waits for is its CU's load path (every workgroup stages all of its activation rows
// through L2 besides its share of the weights), so the time of the launch is the time of the busiest CU. Tiles of
// 32-column multiples leave CUs idle for the common layer widths (Qwen2-1.5B: F = 8960 = 140 x 64 -> 140 workgroups of
// the 256-row x 128-column kernel in decode_gemm.hip on 256 CUs). Here a workgroup owns a column range of ANY width
// w (a multiple of 4, up to 16 TW): F / w ranges, chosen by the caller so that ranges x row blocks ~ number of CUs
// (F = 8960: w = 36 -> 249 workgroups). The MFMA tiles are 16 columns wide (v_mfma_f32_16x16x32_bf16), so a range is
// computed as ceil(w / 16) tiles with the surplus columns fed from clamped (repeated) weight rows and never stored:
// matrix-pipe time is not what this shape waits for.
//
//   * 4 consumer waves (one per SIMD): wave v owns activation rows [16 RT v, 16 RT (v + 1)) of the block and all 2 TW
//     column tiles (TW gate tiles and the matching TW up tiles, so SwiGLU stays inside a lane);
//   * 4 loader waves issue the LDS-DMA pieces (global_load_lds_dwordx4, 8 rows x 128 B each) of STAGES-deep staging,
//     one raw s_barrier per 64-deep K step, counted vmcnt so that STAGES - 2 tiles stay in flight across it; the XOR
//     swizzle that makes the ds_read_b128 fragment reads conflict-free sits on the per-lane SOURCE address;
//   * product transposed (A = weight rows, B = activation rows): a lane ends up with 4 consecutive output columns of
//     one row -> 8-byte bf16 stores;
//   * with two row blocks (M > 256) the two workgroups that stream the same weight columns get block ids 8 apart, i.e.
//     the same XCD under round-robin placement: the second reader is served by that XCD's L2 (speed only).
//
// Role in the reference: the MLP GEMM + activation kernels of its serving engine (areal/api/cli_args.py:398-454).
#include "common.hpp"

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef float f32x4_t __attribute__((ext_vector_type(4)));

struct SGArgs {
  const bf16_t* X;
  long xs;
  const bf16_t* W;
  bf16_t* Y;
  long ys;
  int M, F, K;
  int w;        // columns per range (multiple of 4)
  int ntiles;   // ceil(F / w)
  int mblocks;  // row blocks (1 or 2)
  int nt;       // non-temporal weight loads
  int act;      // gate activation: 0 SiLU (SwiGLU), 1 tanh-approximated GELU (GeGLU)
};

template <int N>
DEVINL void sg_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int IPW>
DEVINL void sg_wait_tiles(int keep) {
  constexpr int C = 63;
  switch (keep) {
    case 0: sg_wait_vmcnt<0>(); break;
    case 1: sg_wait_vmcnt<(IPW < C ? IPW : C)>(); break;
    default: sg_wait_vmcnt<(2 * IPW < C ? 2 * IPW : C)>(); break;
  }
}

DEVINL float sg_silu(float x) { return x / (1.f + __expf(-x)); }
DEVINL float sg_gate(float x, int act) {
  if (act == 0) return sg_silu(x);
  const float inner = 0.7978845608028654f * (x + 0.044715f * x * x * x);
  const float e = __expf(-2.f * fabsf(inner));
  return 0.5f * x * (1.f + copysignf((1.f - e) / (1.f + e), inner));
}

// RT: 16-row activation tiles per consumer wave (block rows BM = 64 RT); TW: 16-column tiles per half (gate / up).
template <int RT, int TW, int STAGES>
__global__ __launch_bounds__(512) void dswiglu_kernel(const SGArgs a) {
  constexpr int BM = 64 * RT, HW = 16 * TW, WR = 2 * HW;  // weight rows of a stage: HW gate + HW up
  constexpr int NP = (WR + BM) / 8, IPW = NP / 4;          // 1-KiB pieces per stage, per loader wave
  static_assert(NP % 4 == 0, "pieces divide evenly over the four loader waves");
  constexpr int STAGE_BYTES = (WR + BM) * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool is_loader = wave >= 4;
  const int lwave = wave - 4;

  // block id -> (column range, row block); the row blocks of one range are 8 ids apart
  int tile, mb;
  {
    const int id = blockIdx.x;
    if (a.mblocks == 1) {
      tile = id;
      mb = 0;
    } else {
      const int x = id & 7, slot = id >> 3;
      mb = slot % a.mblocks;
      tile = (slot / a.mblocks) * 8 + x;
    }
  }
  if (tile >= a.ntiles) return;
  const int c0 = tile * a.w;
  const int c1 = min(a.F, c0 + a.w);
  const int m0 = mb * BM;
  const int nsteps = a.K >> 6;

  if (is_loader) {
    // ---- staging: piece i of a stage = LDS rows 8i .. 8i+7 (gate rows, up rows, activation rows)
    const bf16_t* src[IPW];
#pragma unroll
    for (int j = 0; j < IPW; ++j) {
      const int i = lwave * IPW + j;
      const int row = 8 * i + (lane >> 3);
      const int c = (lane & 7) ^ ((row >> 1) & 7);  // logical 16-byte chunk stored at physical chunk lane&7
      if (8 * i < WR) {
        const int half = row >= HW ? 1 : 0;
        int col = c0 + (row - half * HW);
        col = col < c1 ? col : c1 - 1;  // surplus columns of the last 16-wide tile repeat the last real one
        src[j] = a.W + ((size_t)half * a.F + col) * a.K + 8 * c;
      } else {
        int m = m0 + row - WR;
        m = m < a.M ? m : a.M - 1;
        src[j] = a.X + (size_t)m * a.xs + 8 * c;
      }
    }
    auto issue = [&](int buf, int step) {
      char* base = smem + buf * STAGE_BYTES + lwave * (IPW * 1024);
#pragma unroll
      for (int j = 0; j < IPW; ++j) {
        if (a.nt && 8 * (lwave * IPW + j) < WR)
          __builtin_amdgcn_global_load_lds((gptr_t)(src[j] + step * 64), (lptr_t)(base + j * 1024), 16, 0, 2);
        else
          __builtin_amdgcn_global_load_lds((gptr_t)(src[j] + step * 64), (lptr_t)(base + j * 1024), 16, 0, 0);
      }
    };
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
      if (s < nsteps) issue(s, s);
    for (int t = 0; t < nsteps; ++t) {
      const int ahead = nsteps - 1 - t;
      constexpr int KEEP = STAGES - 2 < 2 ? STAGES - 2 : 2;
      sg_wait_tiles<IPW>(ahead < KEEP ? ahead : KEEP);
      __builtin_amdgcn_s_barrier();
      if (t + STAGES - 1 < nsteps) issue((t + STAGES - 1) % STAGES, t + STAGES - 1);
    }
    return;
  }

  // ---- consumers
  const int r16 = lane & 15, q = lane >> 4;
  const int key = (r16 >> 1) & 7;  // tile bases are multiples of 16 rows: the row's swizzle key is the lane's
  int coff[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) coff[s] = ((4 * s + q) ^ key) << 4;
  const int wbase = r16 * 128;                                  // + tile * 2048
  const int xbase = (WR + wave * 16 * RT + r16) * 128;           // + rt * 2048

  f32x4_t acc[2 * TW][RT];
#pragma unroll
  for (int t = 0; t < 2 * TW; ++t)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[t][rt] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  for (int t = 0; t < nsteps; ++t) {
    __builtin_amdgcn_s_barrier();
    const char* sb = smem + (t % STAGES) * STAGE_BYTES;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 wf[2 * TW], xf[RT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) xf[rt] = *reinterpret_cast<const bf16x8*>(sb + xbase + rt * 2048 + coff[s]);
#pragma unroll
      for (int ct = 0; ct < 2 * TW; ++ct) wf[ct] = *reinterpret_cast<const bf16x8*>(sb + wbase + ct * 2048 + coff[s]);
#pragma unroll
      for (int ct = 0; ct < 2 * TW; ++ct)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
          acc[ct][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ct], xf[rt], acc[ct][rt], 0, 0, 0);
    }
    // every fragment read of this buffer has returned before any wave may restage it
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }

  // ---- epilogue. acc[ct][rt][i]: activation row m0 + (wave RT + rt) 16 + r16, column c0 + 16 (ct % TW) + 4 q + i
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int m = m0 + (wave * RT + rt) * 16 + r16;
    if (m >= a.M) continue;
    bf16_t* yrow = a.Y + (size_t)m * a.ys;
#pragma unroll
    for (int ct = 0; ct < TW; ++ct) {
      const int n = c0 + 16 * ct + 4 * q;
      if (n < c1) {  // ranges are multiples of 4 wide: the four columns are inside together
        bf16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          // gate / up rounded to bf16 first: the values an unfused bf16 GEMM output would carry
          const float gg = bf2f(f2bf(acc[ct][rt][j])), uu = bf2f(f2bf(acc[ct + TW][rt][j]));
          o[j] = f2bf(sg_gate(gg, a.act) * uu);
        }
        *reinterpret_cast<bf16x4*>(yrow + n) = o;
      }
    }
  }
}

template <int RT, int TW, int STAGES>
int sg_launch(const SGArgs& a, hipStream_t st) {
  constexpr size_t lds = (size_t)STAGES * (32 * TW + 64 * RT) * 128;
  static_assert(lds <= 160 * 1024, "staging exceeds the LDS of a CU");
  auto kern = dswiglu_kernel<RT, TW, STAGES>;
  static bool attr_set = false;  // per instantiation
  if (!attr_set) {
    HIP_RET(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  const int groups = a.mblocks == 1 ? a.ntiles : ((a.ntiles + 7) / 8) * 8 * a.mblocks;
  hipLaunchKernelGGL(kern, dim3(groups), dim3(512), lds, st, a);
  return (int)hipGetLastError();
}

template <int RT, int TW>
int sg_stages(const SGArgs& a, int stages, hipStream_t st) {
  constexpr int stage_bytes = (32 * TW + 64 * RT) * 128;
  if (stages >= 4 && 4 * stage_bytes <= 160 * 1024) {
    if constexpr (4 * stage_bytes <= 160 * 1024) return sg_launch<RT, TW, 4>(a, st);
  }
  if (stages >= 3 && 3 * stage_bytes <= 160 * 1024) {
    if constexpr (3 * stage_bytes <= 160 * 1024) return sg_launch<RT, TW, 3>(a, st);
  }
  return sg_launch<RT, TW, 2>(a, st);
}

}  // namespace

// X [M, K] bf16 (row stride xs), W = [gate; up] [2F, K] bf16 contiguous, Y [M, F] bf16 (row stride ys).
// K % 64 == 0, F % 4 == 0, w % 4 == 0, 4 <= w <= 80, M <= 256 mblocks, mblocks in {1, 2}. Returns -10 for a shape
// no instantiation covers (the caller then takes another kernel).
extern "C" int areal_dswiglu(const void* X, long xs, const void* W, void* Y, long ys, int M, int F, int K, int w,
                             int mblocks, int stages, int nt, int act, hipStream_t st) {
  if (M <= 0) return 0;
  if (K % 64 != 0 || xs % 8 != 0 || F % 4 != 0 || w % 4 != 0 || w < 4 || ys % 4 != 0) return -1;
  if (mblocks < 1 || mblocks > 2) return -2;
  if (((uintptr_t)X | (uintptr_t)W) % 16 != 0 || ((uintptr_t)Y) % 8 != 0) return -3;
  SGArgs a;
  a.X = (const bf16_t*)X; a.xs = xs; a.W = (const bf16_t*)W; a.Y = (bf16_t*)Y; a.ys = ys;
  a.M = M; a.F = F; a.K = K; a.w = w; a.ntiles = (F + w - 1) / w; a.mblocks = mblocks; a.nt = nt; a.act = act;
  const int rows = (M + mblocks - 1) / mblocks;  // rows per block
  const int rt = (rows + 63) / 64;
  const int tw = (w + 15) / 16;
  if (rt < 1 || rt > 4) return -10;
#define SG_CASE(R, T) \
  if (rt == R && tw <= T) return sg_stages<R, T>(a, stages, st)
  SG_CASE(1, 3); SG_CASE(2, 3); SG_CASE(3, 3); SG_CASE(4, 3);
  SG_CASE(1, 5); SG_CASE(2, 5); SG_CASE(3, 5); SG_CASE(4, 5);
#undef SG_CASE
  return -10;
}
