// Decode-batch GEMM for gfx950: Y[M, N] = X[M, K] * W[N, K]^T for M = 32 ... 512 rows (one decode step of a
// continuous-batching engine), plus the row kernels that consume its split-K partial sums.
//
// At these sizes neither operand is "free": the weights are read once from HBM, but every column tile
// re-reads the activations from L2, and a CU's load path (not its MFMA pipe, not HBM) is what a block waits
// for. The kernel is built around that:
//
//   * both operands go global -> LDS with 16-byte LDS-DMA loads (`global_load_lds_dwordx4`), one 1 KiB piece
//     (8 rows x 128 B) per wave instruction, no VGPR staging; the LDS image is linear, the XOR swizzle that
//     makes the `ds_read_b128` fragment reads conflict-free is applied to the per-lane SOURCE address;
//   * STAGES LDS buffers, one raw `s_barrier` per 64-deep K step, counted `s_waitcnt vmcnt(N)` so that the
//     next STAGES-2 tiles stay in flight across the barrier;
//   * v_mfma_f32_32x32x16_bf16 with the product transposed (A = weight rows, B = activation rows), so a lane
//     owns 4 consecutive output columns of one row: 8-byte bf16 / 16-byte fp32 stores;
//   * split-K over blockIdx.x (the fastest-varying grid dimension, so blocks that read the same K slice of X
//     share an XCD's L2 under round-robin placement): partial sums go to an fp32 slab [split][M][N] with plain
//     stores and are summed IN A FIXED ORDER by the next kernel of the layer, which exists anyway (residual
//     add + RMSNorm, or bias + RoPE + KV-cache write). No atomics, no in-launch hand-off, graph-safe.
//   * epilogues: bf16 store (+bias), fp32 slab, SwiGLU on a [gate; up] weight (the block's column tile is
//     BN/2 gate columns and the matching BN/2 up columns).
//
// This is the role SGLang's decode GEMMs + elementwise kernels play for the reference
// (areal/api/cli_args.py:398-454 configures that server).
#include <algorithm>

#include "common.hpp"

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

enum { DG_BF16 = 0, DG_SLAB = 1, DG_SWIGLU = 2, DG_ROPE = 3 };

struct DGArgs {
  const bf16_t* X;
  long xs;
  const bf16_t* W;
  const bf16_t* bias;
  void* Y;
  long ys;
  int M, N, K, k_per_split;
  int nt;  // non-temporal weight loads (each weight byte is read once per launch)
  int act;  // gate activation of the gated-MLP epilogue (gate_act_f)
  // DG_ROPE: W = the packed q|k|v projection, heads of D columns. Rotary tables [max_pos, D/2] fp32, position and
  // KV slot per row; q goes to q_out [M, Hq, D], k / v into the paged caches (K [P, Hkv, 32, D], V [P, Hkv, D, 32])
  const int* pos;
  const int64_t* slots;
  const float* cos_t;
  const float* sin_t;
  bf16_t* q_out;
  bf16_t* kcache;
  bf16_t* vcache;
  int Hq, Hkv, D, max_pos, do_rope;
};

template <int N>
DEVINL void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Wait until at most `keep` staged tiles (IPW loads each, issued after the one needed now) are still in flight.
template <int IPW>
DEVINL void wait_tiles(int keep) {
  constexpr int C = 63;  // largest count the instruction encodes; a smaller count only waits for more
  switch (keep) {
    case 0: wait_vmcnt<0>(); break;
    case 1: wait_vmcnt<(IPW < C ? IPW : C)>(); break;
    case 2: wait_vmcnt<(2 * IPW < C ? 2 * IPW : C)>(); break;
    case 3: wait_vmcnt<(3 * IPW < C ? 3 * IPW : C)>(); break;
    case 4: wait_vmcnt<(4 * IPW < C ? 4 * IPW : C)>(); break;
    case 5: wait_vmcnt<(5 * IPW < C ? 5 * IPW : C)>(); break;
    default: wait_vmcnt<(6 * IPW < C ? 6 * IPW : C)>(); break;
  }
}

DEVINL float silu_f(float x) { return x / (1.f + __expf(-x)); }
// gate activation of the gated-MLP epilogue: 0 = SiLU (SwiGLU), 1 = tanh-approximated GELU (GeGLU, Gemma)
DEVINL float gate_act_f(float x, int act) {
  if (act == 0) return silu_f(x);
  const float inner = 0.7978845608028654f * (x + 0.044715f * x * x * x);
  const float e = __expf(-2.f * fabsf(inner));
  return 0.5f * x * (1.f + copysignf((1.f - e) / (1.f + e), inner));
}

// TM / TN: 32-row activation / weight tiles per wave; WM x WN MFMA waves per block. BK = 64.
// LW = 0: every wave stages its share of each tile and computes. LW > 0: LW extra waves do nothing but issue the
// LDS-DMA pieces (a piece costs a dedicated wave ~25 issue cycles, but 100-185 in a wave that is also issuing
// ds_reads and MFMAs), the WM x WN waves only compute; both kinds meet at the one barrier per K step.
template <int TM, int TN, int WM, int WN, int STAGES, int EPI, int LW>
__global__ __launch_bounds__(64 * (WM * WN + LW)) void dgemm_kernel(const DGArgs a) {
  constexpr int NW = WM * WN, BM = 32 * TM * WM, BN = 32 * TN * WN;
  constexpr int NLOAD = LW > 0 ? LW : NW;  // waves that stage
  constexpr int NINST = (BM + BN) / 8, IPW = NINST / NLOAD;
  static_assert(NINST % NLOAD == 0, "staging pieces must divide evenly over the loading waves");
  static_assert((EPI != DG_SWIGLU && EPI != DG_ROPE) || TN % 2 == 0, "paired column tiles need both halves in one wave");
  constexpr bool PAIRED = EPI == DG_SWIGLU || EPI == DG_ROPE;
  constexpr int STAGE_BYTES = (BM + BN) * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool is_loader = LW > 0 ? wave >= NW : true;
  const bool is_consumer = LW > 0 ? wave < NW : true;
  const int lwave = LW > 0 ? wave - NW : wave;  // index among the loading waves
  const int wm = wave % WM, wn = wave / WM;
  const int split = blockIdx.x, ct = blockIdx.y, mb = blockIdx.z;
  const int m0 = mb * BM;
  const int k_begin = split * a.k_per_split;
  const int k_end = min(a.K, k_begin + a.k_per_split);
  const int nsteps = (k_end - k_begin) >> 6;
  const int F = a.N >> 1;  // SwiGLU: number of gate columns
  const int half = a.D >> 1;  // RoPE: a head's rotation partner of column d is column d + D/2

  // global column of LDS weight-tile row `wr`. Paired epilogues put the two columns an output element needs
  // into the same lane: tile rows [0, BN/2) are gate columns (first halves of heads), rows [BN/2, BN) the
  // matching up columns (second halves).
  auto wcol = [&](int wr) -> int {
    if (EPI == DG_SWIGLU) return wr < BN / 2 ? ct * (BN / 2) + wr : F + ct * (BN / 2) + (wr - BN / 2);
    if (EPI == DG_ROPE) {
      const int i = ct * (BN / 2) + (wr < BN / 2 ? wr : wr - BN / 2);
      return (i / half) * a.D + (i % half) + (wr < BN / 2 ? 0 : half);
    }
    return ct * BN + wr;
  };

  // ---- staging: piece i of a stage = LDS rows 8i .. 8i+7 (weight rows first, then activation rows)
  const bf16_t* src[IPW];
#pragma unroll
  for (int j = 0; j < IPW; ++j) {
    const int i = (is_loader ? lwave : 0) * IPW + j;
    const int row = 8 * i + (lane >> 3);
    const int c = (lane & 7) ^ ((row >> 1) & 7);  // logical 16-byte chunk stored at physical chunk lane&7
    if (8 * i < BN) {
      int n = wcol(row);
      n = n < a.N ? n : a.N - 1;
      src[j] = a.W + (size_t)n * a.K + k_begin + 8 * c;
    } else {
      int m = m0 + row - BN;
      m = m < a.M ? m : a.M - 1;
      src[j] = a.X + (size_t)m * a.xs + k_begin + 8 * c;
    }
  }
  auto issue = [&](int buf, int step) {
    char* base = smem + buf * STAGE_BYTES + lwave * (IPW * 1024);
#pragma unroll
    for (int j = 0; j < IPW; ++j) {
      if (a.nt && 8 * (lwave * IPW + j) < BN)
        __builtin_amdgcn_global_load_lds((gptr_t)(src[j] + step * 64), (lptr_t)(base + j * 1024), 16, 0, 2);
      else
        __builtin_amdgcn_global_load_lds((gptr_t)(src[j] + step * 64), (lptr_t)(base + j * 1024), 16, 0, 0);
    }
  };

  // ---- fragment addressing
  const int r = lane & 31, h = lane >> 5;
  const int key = (r >> 1) & 7;
  int coff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) coff[ks] = ((2 * ks + h) ^ key) << 4;
  int aoff[TN], boff[TM];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    int base;
    if (PAIRED) {
      constexpr int HT = TN / 2 > 0 ? TN / 2 : 1;
      base = tn < HT ? (wn * HT + tn) * 32 : BN / 2 + (wn * HT + (tn - HT)) * 32;
    } else {
      base = (wn * TN + tn) * 32;
    }
    aoff[tn] = (base + r) * 128;
  }
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) boff[tm] = (BN + (wm * TM + tm) * 32 + r) * 128;

  f32x16 acc[TN][TM];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn)
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[tn][tm][i] = 0.f;

  if (is_loader) {
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
      if (s < nsteps) issue(s, s);
  }

  for (int t = 0; t < nsteps; ++t) {
    if (is_loader) {
      const int ahead = nsteps - 1 - t;  // tiles issued after tile t that may stay in flight
      constexpr int KEEP = STAGES - 2 < 6 ? STAGES - 2 : 6;
      wait_tiles<IPW>(ahead < KEEP ? ahead : KEEP);
    }
    __builtin_amdgcn_s_barrier();
    if (is_loader && t + STAGES - 1 < nsteps) issue((t + STAGES - 1) % STAGES, t + STAGES - 1);
    if (is_consumer) {
      const char* sb = smem + (t % STAGES) * STAGE_BYTES;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        bf16x8 af[TN], bfr[TM];
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) af[tn] = *reinterpret_cast<const bf16x8*>(sb + aoff[tn] + coff[ks]);
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) bfr[tm] = *reinterpret_cast<const bf16x8*>(sb + boff[tm] + coff[ks]);
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
          for (int tm = 0; tm < TM; ++tm)
            acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[tn], bfr[tm], acc[tn][tm], 0, 0, 0);
      }
      // every fragment read of this buffer has returned before any wave may restage it
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  if (!is_consumer) return;

  // ---- epilogue. acc[tn][tm][i]: row m = m0 + (wm*TM+tm)*32 + r, column = tile base + (i&3) + 8*(i>>2) + 4*h
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const int m = m0 + (wm * TM + tm) * 32 + r;
    if (m >= a.M) continue;
    if (EPI == DG_ROPE) {
      constexpr int HT = TN / 2 > 0 ? TN / 2 : 1;
      int p = a.pos[m];
      p = p < 0 ? 0 : (p >= a.max_pos ? a.max_pos - 1 : p);
      const int64_t slot = a.slots ? a.slots[m] : -1;
      const int64_t page = slot >= 0 ? slot / 32 : 0;
      const int off = slot >= 0 ? (int)(slot % 32) : 0;
#pragma unroll
      for (int tn = 0; tn < HT; ++tn) {
        const int i0 = ct * (BN / 2) + (wn * HT + tn) * 32;  // index among the first-half columns of all heads
        const int head = i0 / half, dbase = i0 % half;       // uniform over the wave: 32 divides D/2
        if (head >= a.Hq + 2 * a.Hkv) continue;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int d0 = dbase + 8 * g + 4 * h;
          float lo[4], hi[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            lo[j] = acc[tn][tm][4 * g + j];
            hi[j] = acc[tn + HT][tm][4 * g + j];
          }
          if (a.bias) {
            const bf16x4 bl = *reinterpret_cast<const bf16x4*>(a.bias + head * a.D + d0);
            const bf16x4 bh = *reinterpret_cast<const bf16x4*>(a.bias + head * a.D + half + d0);
#pragma unroll
            for (int j = 0; j < 4; ++j) { lo[j] += bf2f(bl[j]); hi[j] += bf2f(bh[j]); }
          }
          // the unfused path carries the projection as a bf16 tensor before rotating it
#pragma unroll
          for (int j = 0; j < 4; ++j) { lo[j] = bf2f(f2bf(lo[j])); hi[j] = bf2f(f2bf(hi[j])); }
          bf16x4 olo, ohi;
          if (a.do_rope && head < a.Hq + a.Hkv) {
            const f32x4 c = *reinterpret_cast<const f32x4*>(a.cos_t + (size_t)p * half + d0);
            const f32x4 sn = *reinterpret_cast<const f32x4*>(a.sin_t + (size_t)p * half + d0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              olo[j] = f2bf(lo[j] * c[j] - hi[j] * sn[j]);
              ohi[j] = f2bf(hi[j] * c[j] + lo[j] * sn[j]);
            }
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) { olo[j] = f2bf(lo[j]); ohi[j] = f2bf(hi[j]); }
          }
          if (head < a.Hq) {
            bf16_t* qd = a.q_out + ((size_t)m * a.Hq + head) * a.D + d0;
            *reinterpret_cast<bf16x4*>(qd) = olo;
            *reinterpret_cast<bf16x4*>(qd + half) = ohi;
          } else if (slot >= 0) {
            if (head < a.Hq + a.Hkv) {
              bf16_t* kd = a.kcache + (((size_t)page * a.Hkv + (head - a.Hq)) * 32 + off) * a.D + d0;
              *reinterpret_cast<bf16x4*>(kd) = olo;
              *reinterpret_cast<bf16x4*>(kd + half) = ohi;
            } else {
              bf16_t* vd = a.vcache + (((size_t)page * a.Hkv + (head - a.Hq - a.Hkv)) * a.D + d0) * 32 + off;
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                vd[(size_t)j * 32] = olo[j];
                vd[(size_t)(half + j) * 32] = ohi[j];
              }
            }
          }
        }
      }
    } else if (EPI == DG_SWIGLU) {
      constexpr int HT = TN / 2 > 0 ? TN / 2 : 1;
      bf16_t* yrow = (bf16_t*)a.Y + (size_t)m * a.ys;
#pragma unroll
      for (int tn = 0; tn < HT; ++tn) {
        const int nb = ct * (BN / 2) + (wn * HT + tn) * 32;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = nb + 8 * g + 4 * h;
          if (n < F) {
            bf16x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              // gate / up rounded to bf16 first: the values an unfused bf16 GEMM output would carry
              const float gg = bf2f(f2bf(acc[tn][tm][4 * g + j])), uu = bf2f(f2bf(acc[tn + HT][tm][4 * g + j]));
              o[j] = f2bf(gate_act_f(gg, a.act) * uu);
            }
            *reinterpret_cast<bf16x4*>(yrow + n) = o;
          }
        }
      }
    } else {
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        const int nb = ct * BN + (wn * TN + tn) * 32;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = nb + 8 * g + 4 * h;
          if (n < a.N) {
            if (EPI == DG_SLAB) {
              float* yrow = (float*)a.Y + ((size_t)split * a.M + m) * a.ys;
              f32x4 o = {acc[tn][tm][4 * g], acc[tn][tm][4 * g + 1], acc[tn][tm][4 * g + 2], acc[tn][tm][4 * g + 3]};
              *reinterpret_cast<f32x4*>(yrow + n) = o;
            } else {
              bf16_t* yrow = (bf16_t*)a.Y + (size_t)m * a.ys;
              bf16x4 o;
              if (a.bias) {
                const bf16x4 b = *reinterpret_cast<const bf16x4*>(a.bias + n);
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = f2bf(acc[tn][tm][4 * g + j] + bf2f(b[j]));
              } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = f2bf(acc[tn][tm][4 * g + j]);
              }
              *reinterpret_cast<bf16x4*>(yrow + n) = o;
            }
          }
        }
      }
    }
  }
}

template <int TM, int TN, int WM, int WN, int STAGES, int EPI, int LW>
int launch_cfg(const DGArgs& a, int ksplit, hipStream_t st) {
  constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
  constexpr size_t lds = (size_t)STAGES * (BM + BN) * 128;
  if (lds > 160 * 1024) return -20;
  if (LW > 0 && ((BM + BN) / 8) % LW != 0) return -23;
  auto kern = dgemm_kernel<TM, TN, WM, WN, STAGES, EPI, ((BM + BN) / 8) % (LW > 0 ? LW : 1) == 0 ? LW : 0>;
  if (lds > 64 * 1024) {
    static bool attr_set = false;  // per instantiation
    if (!attr_set) {
      HIP_RET(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      attr_set = true;
    }
  }
  const int ncols = (EPI == DG_SWIGLU || EPI == DG_ROPE) ? a.N / 2 : a.N;
  constexpr int cols_per_tile = (EPI == DG_SWIGLU || EPI == DG_ROPE) ? BN / 2 : BN;
  dim3 grid(ksplit, (ncols + cols_per_tile - 1) / cols_per_tile, (a.M + BM - 1) / BM);
  hipLaunchKernelGGL(kern, grid, dim3(64 * (WM * WN + LW)), lds, st, a);
  return (int)hipGetLastError();
}

template <int TM, int TN, int WM, int WN, int STAGES, int LW>
int launch_epi(const DGArgs& a, int epi, int ksplit, hipStream_t st) {
  switch (epi) {
    case DG_BF16: return launch_cfg<TM, TN, WM, WN, STAGES, DG_BF16, LW>(a, ksplit, st);
    case DG_SLAB: return launch_cfg<TM, TN, WM, WN, STAGES, DG_SLAB, LW>(a, ksplit, st);
    case DG_SWIGLU: return launch_cfg<TM, TN, WM, WN, STAGES, DG_SWIGLU, LW>(a, ksplit, st);
  }
  return -21;
}

template <int TM, int TN, int WM, int WN>
int launch_stages(const DGArgs& a, int stages, int epi, int ksplit, int lw, hipStream_t st) {
  if (lw == 0) {
    switch (stages) {
      case 2: return launch_epi<TM, TN, WM, WN, 2, 0>(a, epi, ksplit, st);
      case 3: return launch_epi<TM, TN, WM, WN, 3, 0>(a, epi, ksplit, st);
      case 4: return launch_epi<TM, TN, WM, WN, 4, 0>(a, epi, ksplit, st);
    }
  } else if (lw == 4) {
    switch (stages) {
      case 2: return launch_epi<TM, TN, WM, WN, 2, 4>(a, epi, ksplit, st);
      case 3: return launch_epi<TM, TN, WM, WN, 3, 4>(a, epi, ksplit, st);
      case 4: return launch_epi<TM, TN, WM, WN, 4, 4>(a, epi, ksplit, st);
    }
  }
  return -22;
}

template <int TM, int TN, int WM, int WN>
int launch_rope(const DGArgs& a, int stages, int lw, hipStream_t st) {
  if (lw == 0) {
    switch (stages) {
      case 2: return launch_cfg<TM, TN, WM, WN, 2, DG_ROPE, 0>(a, 1, st);
      case 3: return launch_cfg<TM, TN, WM, WN, 3, DG_ROPE, 0>(a, 1, st);
      case 4: return launch_cfg<TM, TN, WM, WN, 4, DG_ROPE, 0>(a, 1, st);
      case 6: return launch_cfg<TM, TN, WM, WN, 6, DG_ROPE, 0>(a, 1, st);
      case 8: return launch_cfg<TM, TN, WM, WN, 8, DG_ROPE, 0>(a, 1, st);
    }
  } else if (lw == 4) {
    switch (stages) {
      case 2: return launch_cfg<TM, TN, WM, WN, 2, DG_ROPE, 4>(a, 1, st);
      case 3: return launch_cfg<TM, TN, WM, WN, 3, DG_ROPE, 4>(a, 1, st);
      case 4: return launch_cfg<TM, TN, WM, WN, 4, DG_ROPE, 4>(a, 1, st);
      case 6: return launch_cfg<TM, TN, WM, WN, 6, DG_ROPE, 4>(a, 1, st);
      case 8: return launch_cfg<TM, TN, WM, WN, 8, DG_ROPE, 4>(a, 1, st);
    }
  }
  return -22;
}

}  // namespace

// Tile configurations (rows x columns per block, waves): see DG_CONFIGS in ops/decode_gemm.py.
extern "C" int areal_dgemm_block_shape(int cfg, int* bm, int* bn, int* threads) {
  static const int tab[][4] = {{1, 2, 8, 1}, {2, 2, 4, 2}, {2, 2, 2, 2}, {1, 2, 4, 1}, {1, 2, 2, 2},
                               {2, 2, 4, 1}, {1, 4, 8, 1}, {2, 4, 4, 1}, {1, 2, 2, 1}, {1, 2, 1, 1}};
  if (cfg < 0 || cfg >= (int)(sizeof(tab) / sizeof(tab[0]))) return -1;
  *bm = 32 * tab[cfg][0] * tab[cfg][2];
  *bn = 32 * tab[cfg][1] * tab[cfg][3];
  *threads = 64 * tab[cfg][2] * tab[cfg][3];
  return 0;
}

// X [M, K] bf16 (row stride xs), W [N, K] bf16 contiguous, K % 64 == 0, N % 32 == 0 (SwiGLU: N/2 % 32 == 0).
// epilogue 0: Y bf16 [M, N] (+bias); 1: Y fp32 slab [ksplit, M, N] (ys = N; bias ignored);
// 2: W = [gate; up], Y bf16 [M, N/2] = silu(X gate^T) * (X up^T); 3: the same with the tanh-approximated GELU on the
// gate (GeGLU). ksplit > 1 only with epilogue 1.
extern "C" int areal_dgemm(const void* X, long xs, const void* W, const void* bias, void* Y, long ys, int M, int N,
                           int K, int cfg, int stages, int epilogue, int ksplit, int loader_waves, int nt,
                           hipStream_t st) {
  if (M <= 0) return 0;
  int act = 0;
  if (epilogue == 3) {
    act = 1;
    epilogue = DG_SWIGLU;
  }
  if (K % 64 != 0 || xs % 8 != 0 || N % 32 != 0) return -1;
  if (epilogue == DG_SWIGLU && (N / 2) % 32 != 0) return -2;
  if (ksplit < 1) ksplit = 1;
  if (ksplit > 1 && epilogue != DG_SLAB) return -3;
  int kps = ((K / 64 + ksplit - 1) / ksplit) * 64;
  if ((K + kps - 1) / kps != ksplit) return -4;  // caller must pass a split count that leaves no empty split
  if (epilogue == DG_SLAB && ys != N) return -5;
  DGArgs a;
  a.X = (const bf16_t*)X; a.xs = xs; a.W = (const bf16_t*)W; a.bias = (const bf16_t*)bias; a.Y = Y; a.ys = ys;
  a.M = M; a.N = N; a.K = K; a.k_per_split = kps; a.nt = nt; a.act = act;
  a.pos = nullptr; a.slots = nullptr; a.cos_t = a.sin_t = nullptr; a.q_out = a.kcache = a.vcache = nullptr;
  a.Hq = a.Hkv = 0; a.D = 2; a.max_pos = 0; a.do_rope = 0;
  switch (cfg) {
    case 0: return launch_stages<1, 2, 8, 1>(a, stages, epilogue, ksplit, loader_waves, st);
    case 1: return launch_stages<2, 2, 4, 2>(a, stages, epilogue, ksplit, loader_waves, st);
    case 2: return launch_stages<2, 2, 2, 2>(a, stages, epilogue, ksplit, loader_waves, st);
    case 3: return launch_stages<1, 2, 4, 1>(a, stages, epilogue, ksplit, loader_waves, st);
    case 4: return launch_stages<1, 2, 2, 2>(a, stages, epilogue, ksplit, loader_waves, st);
    case 5: return launch_stages<2, 2, 4, 1>(a, stages, epilogue, ksplit, loader_waves, st);
    case 6: return launch_stages<1, 4, 8, 1>(a, stages, epilogue, ksplit, loader_waves, st);
    case 7: return launch_stages<2, 4, 4, 1>(a, stages, epilogue, ksplit, loader_waves, st);
    case 8: return launch_stages<1, 2, 2, 1>(a, stages, epilogue, ksplit, loader_waves, st);
    case 9: return launch_stages<1, 2, 1, 1>(a, stages, epilogue, ksplit, loader_waves, st);
  }
  return -6;
}

// The packed q|k|v projection of a decode step in one launch: Y = X W^T (+bias) rounded to bf16, q and k rotated
// (half-split rotary layout), q written to q_out [M, Hq, D], k / v scattered into the paged caches at slots[m]
// (rows with slot < 0 write no K/V). A block's column tile is BN/2 first-half columns of a head and their BN/2
// rotation partners, so the rotation stays inside a lane. Configurations 3 (128x64), 4 (64x128), 8 (64x64),
// 9 (32x64); D/2 must be a multiple of the half tile (32, for configuration 4: 64).
extern "C" int areal_dgemm_qkv_rope(const void* X, long xs, const void* W, const void* bias, const int* pos,
                                    const int64_t* slots, const float* cos_t, const float* sin_t, void* q_out,
                                    void* kcache, void* vcache, int M, int K, int Hq, int Hkv, int D, int max_pos,
                                    int do_rope, int cfg, int stages, int loader_waves, int nt, hipStream_t st) {
  if (M <= 0) return 0;
  const int N = (Hq + 2 * Hkv) * D;
  if (K % 64 != 0 || xs % 8 != 0 || D % 64 != 0) return -1;
  if (cfg == 4 && (D / 2) % 64 != 0) return -2;
  DGArgs a;
  a.X = (const bf16_t*)X; a.xs = xs; a.W = (const bf16_t*)W; a.bias = (const bf16_t*)bias; a.Y = nullptr; a.ys = 0;
  a.M = M; a.N = N; a.K = K; a.k_per_split = K; a.nt = nt; a.act = 0;
  a.pos = pos; a.slots = slots; a.cos_t = cos_t; a.sin_t = sin_t; a.q_out = (bf16_t*)q_out;
  a.kcache = (bf16_t*)kcache; a.vcache = (bf16_t*)vcache; a.Hq = Hq; a.Hkv = Hkv; a.D = D; a.max_pos = max_pos;
  a.do_rope = do_rope;
  switch (cfg) {
    case 3: return launch_rope<1, 2, 4, 1>(a, stages, loader_waves, st);
    case 4: return launch_rope<1, 2, 2, 2>(a, stages, loader_waves, st);
    case 8: return launch_rope<1, 2, 2, 1>(a, stages, loader_waves, st);
    case 9: return launch_rope<1, 2, 1, 1>(a, stages, loader_waves, st);
  }
  return -6;
}

// ----------------------------------------------------------------------------------
// Row kernels that finish a split-K GEMM (sum of the fp32 slabs in split order) and do the layer's
// elementwise work in the same pass.
// ----------------------------------------------------------------------------------

// Sum of the S slabs for one float4 column chunk of one row, four independent loads in flight per step (a
// runtime-trip-count loop of dependent load -> add pairs would pay one memory latency per slab).
DEVINL f32x4 slab_sum4(const float* __restrict__ p, size_t stride, int S) {
  f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
  int k = 0;
  for (; k + 4 <= S; k += 4) {
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(p + (size_t)k * stride);
    const f32x4 v1 = *reinterpret_cast<const f32x4*>(p + (size_t)(k + 1) * stride);
    const f32x4 v2 = *reinterpret_cast<const f32x4*>(p + (size_t)(k + 2) * stride);
    const f32x4 v3 = *reinterpret_cast<const f32x4*>(p + (size_t)(k + 3) * stride);
    a0 += v0; a1 += v1; a2 += v2; a3 += v3;
  }
  for (; k < S; ++k) a0 += *reinterpret_cast<const f32x4*>(p + (size_t)k * stride);
  return (a0 + a1) + (a2 + a3);
}

// s = bf16(sum_k slab[k] [+ bias] [+ res]); res_out = s; y = rmsnorm(s) * w (when w != null, else y = s).
// One block of THREADS threads per row (M blocks in flight, every thread owns up to MAXC float4 chunks of the row:
// rows of up to 2048 elements get one chunk per thread, so that all of a thread's loads are in flight at once);
// N % 4 == 0, N <= 4 * THREADS * MAXC.
template <int MAXC, int THREADS>
__global__ __launch_bounds__(THREADS) void splitk_residual_rmsnorm_kernel(
    const float* __restrict__ slab, int S, const bf16_t* __restrict__ bias, const bf16_t* __restrict__ res,
    const bf16_t* __restrict__ w, bf16_t* __restrict__ res_out, bf16_t* __restrict__ y, int M, int N, float eps,
    int gemma) {
  __shared__ float red[16];
  const int row = blockIdx.x;
  const int nchunk = N >> 2;
  const size_t stride = (size_t)M * N;
  float v[MAXC][4];
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = threadIdx.x + i * THREADS;
    if (c < nchunk) {
      f32x4 s = slab_sum4(slab + (size_t)row * N + c * 4, stride, S);
      if (bias) {
        const bf16x4 b = *reinterpret_cast<const bf16x4*>(bias + c * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) s[j] += bf2f(b[j]);
      }
      bf16x4 o;
      if (res) {
        const bf16x4 rr = *reinterpret_cast<const bf16x4*>(res + (size_t)row * N + c * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = f2bf(bf2f(f2bf(s[j])) + bf2f(rr[j]));
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = f2bf(s[j]);
      }
      if (res_out) *reinterpret_cast<bf16x4*>(res_out + (size_t)row * N + c * 4) = o;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[i][j] = bf2f(o[j]);
        ss += v[i][j] * v[i][j];
      }
    }
  }
  if (!w) {
    if (y && y != res_out) {
#pragma unroll
      for (int i = 0; i < MAXC; ++i) {
        const int c = threadIdx.x + i * THREADS;
        if (c < nchunk) {
          bf16x4 o;
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = f2bf(v[i][j]);
          *reinterpret_cast<bf16x4*>(y + (size_t)row * N + c * 4) = o;
        }
      }
    }
    return;
  }
  bf16x4 wv[MAXC];  // loaded before the reduction's barriers, not after them
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = threadIdx.x + i * THREADS;
    if (c < nchunk) wv[i] = *reinterpret_cast<const bf16x4*>(w + c * 4);
  }
  ss = block_sum(ss, red);
  const float rstd = rsqrtf(ss / (float)N + eps);
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = threadIdx.x + i * THREADS;
    if (c < nchunk) {
      const bf16x4 ww = wv[i];
      bf16x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = f2bf(v[i][j] * rstd * (bf2f(ww[j]) + (gemma ? 1.f : 0.f)));
      *reinterpret_cast<bf16x4*>(y + (size_t)row * N + c * 4) = o;
    }
  }
}

extern "C" int areal_splitk_residual_rmsnorm(const float* slab, int S, const void* bias, const void* res,
                                             const void* w, void* res_out, void* y, int M, int N, float eps,
                                             int gemma, hipStream_t st) {
  if (M <= 0) return 0;
  if (N % 4 != 0 || N > 4 * 256 * 16 || S < 1) return -1;
  const int nchunk = N / 4;
  dim3 grid(M);
#define LAUNCH(MC, TH)                                                                                          \
  hipLaunchKernelGGL((splitk_residual_rmsnorm_kernel<MC, TH>), grid, dim3(TH), 0, st, slab, S, (const bf16_t*)bias, \
                     (const bf16_t*)res, (const bf16_t*)w, (bf16_t*)res_out, (bf16_t*)y, M, N, eps, gemma)
  if (nchunk <= 256) LAUNCH(1, 256);
  else if (nchunk <= 384) LAUNCH(1, 384);
  else if (nchunk <= 512) LAUNCH(1, 512);
  else if (nchunk <= 1024) LAUNCH(2, 512);
  else if (nchunk <= 2048) LAUNCH(4, 512);
  else LAUNCH(16, 256);
#undef LAUNCH
  return (int)hipGetLastError();
}

// qkv = bf16(sum_k slab[k] + bias), laid out [T, Hq + 2 Hkv, D]: rotate q and k (half-split layout), write q to
// q_out [T, Hq, D] (bf16), k / v into the paged caches at slots[t] (K [P, Hkv, 32, D], V [P, Hkv, D, 32]).
// One block per token (256 .. 1024 threads by row length): the summed row is staged in LDS (fp32, Nq floats), then
// rotated / scattered from there.
__global__ __launch_bounds__(1024) void splitk_rope_cache_write_kernel(
    const float* __restrict__ slab, int S, const bf16_t* __restrict__ bias, const int* __restrict__ pos,
    const int64_t* __restrict__ slots, const float* __restrict__ cos_t, const float* __restrict__ sin_t,
    bf16_t* __restrict__ q_out, void* __restrict__ kcache, void* __restrict__ vcache, int T, int Hq, int Hkv,
    int D, int max_pos, int do_rope, const bf16_t* __restrict__ q_norm, const bf16_t* __restrict__ k_norm,
    float norm_eps, int fp8) {
  extern __shared__ __attribute__((aligned(16))) float rowbuf[];
  const int t = blockIdx.x;
  const int half = D >> 1;
  const int lph = half >> 3;
  const int Nq = (Hq + 2 * Hkv) * D;
  const size_t stride = (size_t)T * Nq;
  // Position, slot and this thread's first rotary-table rows are requested before the slab sum, so that their
  // latency (a dependent pos -> table chain) runs under the slab loads instead of after the barrier.
  int p = pos[t];
  p = p < 0 ? 0 : (p >= max_pos ? max_pos - 1 : p);
  const int64_t slot = slots ? slots[t] : -1;
  const int nrot = (Hq + Hkv) * lph;
  f32x4 c_pre[2], s_pre[2];
  if (do_rope && (int)threadIdx.x < nrot) {
    const int sub0 = threadIdx.x % lph;
    const f32x4* cp4 = reinterpret_cast<const f32x4*>(cos_t + (size_t)p * half + sub0 * 8);
    const f32x4* sp4 = reinterpret_cast<const f32x4*>(sin_t + (size_t)p * half + sub0 * 8);
    c_pre[0] = cp4[0]; c_pre[1] = cp4[1]; s_pre[0] = sp4[0]; s_pre[1] = sp4[1];
  }
  for (int c = threadIdx.x; c < (Nq >> 2); c += blockDim.x) {
    f32x4 s = slab_sum4(slab + (size_t)t * Nq + c * 4, stride, S);
    if (bias) {
      const bf16x4 b = *reinterpret_cast<const bf16x4*>(bias + c * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) s[j] += bf2f(b[j]);
    }
    // the unfused path carries the projection as a bf16 tensor
#pragma unroll
    for (int j = 0; j < 4; ++j) s[j] = bf2f(f2bf(s[j]));
    *reinterpret_cast<f32x4*>(rowbuf + c * 4) = s;
  }
  __syncthreads();
  const int64_t page = slot >= 0 ? slot / 32 : 0;
  const int off = slot >= 0 ? (int)(slot % 32) : 0;
  for (int i = threadIdx.x; i < nrot; i += blockDim.x) {
    const int head = i / lph, sub = i % lph;
    // 16-byte LDS / global reads: dword reads at this 32-byte lane stride would be 8-way bank conflicts
    const f32x4* lo4 = reinterpret_cast<const f32x4*>(rowbuf + head * D + sub * 8);
    const f32x4* hi4 = reinterpret_cast<const f32x4*>(rowbuf + head * D + sub * 8 + half);
    // A lane's two 16-byte halves sit 32 bytes from its neighbour's: read in the same order by all lanes, the 16
    // lanes that ds_read_b128 serves together ({0-3, 12-15, 20-27}, ...) would use only every other 16-byte slot of
    // the 256-byte bank row, twice each. Lanes 16-31 and 48-63 take their halves in the opposite order instead.
    const int sel = (threadIdx.x >> 4) & 1;
    float lo[8], hi[8];
    {
      const f32x4 a0 = lo4[sel], a1 = lo4[sel ^ 1], b0 = hi4[sel], b1 = hi4[sel ^ 1];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        lo[j] = sel ? a1[j] : a0[j];
        lo[4 + j] = sel ? a0[j] : a1[j];
        hi[j] = sel ? b1[j] : b0[j];
        hi[4 + j] = sel ? b0[j] : b1[j];
      }
    }
    if (q_norm) {
      // per-head RMSNorm of q and k ahead of the rotation (Qwen3): the lph lanes that share a head are neighbours
      // in one wave; normalised values are rounded to bf16 as the unfused norm kernel's output is
      float ss = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) ss += lo[j] * lo[j] + hi[j] * hi[j];
      for (int o = 1; o < lph; o <<= 1) ss += __shfl_xor(ss, o, 64);
      const float rstd = rsqrtf(ss / (float)D + norm_eps);
      const bf16_t* wn = (head < Hq ? q_norm : k_norm) + sub * 8;
      const bf16x8 wl = ld8(wn), wh = ld8(wn + half);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        lo[j] = bf2f(f2bf(lo[j] * rstd * bf2f(wl[j])));
        hi[j] = bf2f(f2bf(hi[j] * rstd * bf2f(wh[j])));
      }
    }
    bf16x8 olo, ohi;
    if (do_rope) {
      const bool first = i == (int)threadIdx.x;  // the rows requested up front
      const f32x4* cp4 = reinterpret_cast<const f32x4*>(cos_t + (size_t)p * half + sub * 8);
      const f32x4* sp4 = reinterpret_cast<const f32x4*>(sin_t + (size_t)p * half + sub * 8);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        f32x4 c, sn;
        if (first) { c = c_pre[q]; sn = s_pre[q]; } else { c = cp4[q]; sn = sp4[q]; }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          olo[4 * q + j] = f2bf(lo[4 * q + j] * c[j] - hi[4 * q + j] * sn[j]);
          ohi[4 * q + j] = f2bf(hi[4 * q + j] * c[j] + lo[4 * q + j] * sn[j]);
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) { olo[j] = f2bf(lo[j]); ohi[j] = f2bf(hi[j]); }
    }
    if (head < Hq) {
      bf16_t* qd = q_out + ((size_t)t * Hq + head) * D + sub * 8;
      st8(qd, olo);
      st8(qd + half, ohi);
    } else if (slot >= 0) {
      const int kh = head - Hq;
      const size_t kd = (((size_t)page * Hkv + kh) * 32 + off) * D + sub * 8;
      st8_cache(kcache, kd, olo, fp8);
      st8_cache(kcache, kd + half, ohi, fp8);
    }
  }
  if (slot >= 0) {
    const float* vrow = rowbuf + (Hq + Hkv) * D;
    for (int i = threadIdx.x; i < Hkv * D; i += blockDim.x) {
      const int vh = i / D, d = i % D;
      st1_cache(vcache, (((size_t)page * Hkv + vh) * D + d) * 32 + off, vrow[i], fp8);
    }
  }
}

extern "C" int areal_splitk_rope_cache_write(const float* slab, int S, const void* bias, const int* pos,
                                             const int64_t* slots, const float* cos_t, const float* sin_t,
                                             void* q_out, void* kcache, void* vcache, int T, int Hq, int Hkv, int D,
                                             int max_pos, int do_rope, const void* q_norm, const void* k_norm,
                                             float norm_eps, int fp8, hipStream_t st) {
  if (T <= 0) return 0;
  const size_t lds = (size_t)(Hq + 2 * Hkv) * D * sizeof(float);
  if (D % 16 != 0 || S < 1 || lds > 64 * 1024) return -1;
  if ((q_norm == nullptr) != (k_norm == nullptr)) return -2;
  if (q_norm && (D / 16 > 64 || ((D / 16) & (D / 16 - 1)) != 0)) return -3;  // lanes of a head: a power of two in one wave
  // one float4 chunk of the summed row per thread where the row allows it: all of a thread's slab loads are in
  // flight at once (a second pass over the row would wait for the first one's loads)
  const int threads = std::min(1024, std::max(256, (((Hq + 2 * Hkv) * D / 4 + 63) / 64) * 64));
  hipLaunchKernelGGL(splitk_rope_cache_write_kernel, dim3(T), dim3(threads), lds, st, slab, S, (const bf16_t*)bias, pos,
                     slots, cos_t, sin_t, (bf16_t*)q_out, kcache, vcache, T, Hq, Hkv, D, max_pos,
                     do_rope, (const bf16_t*)q_norm, (const bf16_t*)k_norm, norm_eps, fp8);
  return (int)hipGetLastError();
}
