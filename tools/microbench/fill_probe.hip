// Per-CU rate of staging tiles into LDS with LDS-DMA loads, by source (HBM-streamed weight panel, L2-resident
// activation matrix, both) and by the number of waves that issue the loads.
//
//   hipcc --offload-arch=gfx950 -O3 -o fill_probe tools/microbench/fill_probe.hip && ./fill_probe
//
// A workgroup of NW waves stages, per K step of 64 columns, WROWS rows of its own weight panel (streamed from HBM,
// several matrix copies are cycled) and XROWS rows of a shared activation matrix X[XROWS][K] (L2-resident), in
// STAGES buffers with counted vmcnt and one barrier per step - the staging loop of the decode GEMM without its
// MFMAs. Reported: time per launch and the fill rate per CU (bytes landed in LDS / time / workgroups).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                               \
  do {                                                                         \
    hipError_t e = (x);                                                        \
    if (e != hipSuccess) {                                                     \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); \
      exit(1);                                                                 \
    }                                                                          \
  } while (0)

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int N>
__device__ inline void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int WROWS, int XROWS, int NW, int STAGES>
__global__ __launch_bounds__(64 * NW) void fill_kernel(const unsigned short* __restrict__ W,
                                                        const unsigned short* __restrict__ X, int K, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NPIECE = (WROWS + XROWS) / 8;  // 1 KiB pieces per step
  constexpr int IPW = NPIECE / NW;
  static_assert(NPIECE % NW == 0, "pieces must divide over the waves");
  constexpr int STAGE_BYTES = (WROWS + XROWS) * 128;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nsteps = K / 64;
  const unsigned short* src[IPW];
#pragma unroll
  for (int j = 0; j < IPW; ++j) {
    const int row = 8 * (wave * IPW + j) + (lane >> 3);
    const int c = lane & 7;
    src[j] = row < WROWS ? W + ((size_t)blockIdx.x * WROWS + row) * K + 8 * c : X + (size_t)(row - WROWS) * K + 8 * c;
  }
  auto issue = [&](int buf, int step) {
    char* base = smem + buf * STAGE_BYTES + wave * (IPW * 1024);
#pragma unroll
    for (int j = 0; j < IPW; ++j)
      __builtin_amdgcn_global_load_lds((gptr_t)(src[j] + step * 64), (lptr_t)(base + j * 1024), 16, 0, 0);
  };
  for (int s = 0; s < STAGES - 1 && s < nsteps; ++s) issue(s, s);
  float acc = 0.f;
  for (int t = 0; t < nsteps; ++t) {
    const int ahead = nsteps - 1 - t;
    if (STAGES >= 4 && ahead >= 2) wait_vmcnt<2 * IPW>();
    else if (STAGES >= 3 && ahead >= 1) wait_vmcnt<IPW>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (t + STAGES - 1 < nsteps) issue((t + STAGES - 1) % STAGES, t + STAGES - 1);
    acc += *reinterpret_cast<const float*>(smem + (t % STAGES) * STAGE_BYTES + (threadIdx.x & 255) * 16);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  if (acc == 12345.678f) sink[blockIdx.x] = acc;
}

template <int WROWS, int XROWS, int NW, int STAGES>
void run(const std::vector<unsigned short*>& copies, const unsigned short* X, int nblocks, int K, float* sink) {
  const size_t lds = (size_t)STAGES * (WROWS + XROWS) * 128;
  if (lds > 160 * 1024) return;
  auto kern = fill_kernel<WROWS, XROWS, NW, STAGES>;
  CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int iters = 20;
  for (auto* w : copies) hipLaunchKernelGGL(kern, dim3(nblocks), dim3(64 * NW), lds, 0, w, X, K, sink);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i)
    for (auto* w : copies) hipLaunchKernelGGL(kern, dim3(nblocks), dim3(64 * NW), lds, 0, w, X, K, sink);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / (iters * copies.size());
  const double per_block = (double)(WROWS + XROWS) * K * 2;
  printf("  W rows %3d (HBM) + X rows %3d (L2), %2d waves, %d stages, %3d workgroups: %6.1f us, %5.1f GB/s per workgroup, "
         "%5.2f TB/s total (W alone %5.2f TB/s)\n", WROWS, XROWS, NW, STAGES, nblocks, us, per_block / us / 1e3,
         per_block * nblocks / us / 1e6, (double)WROWS * K * 2 * nblocks / us / 1e6);
}

int main() {
  const int K = 1536, N = 17920;
  float* sink;
  CHECK(hipMalloc(&sink, 1 << 20));
  const size_t bytes = (size_t)N * K * 2;
  std::vector<unsigned short*> copies(9);
  for (auto& c : copies) {
    CHECK(hipMalloc(&c, bytes));
    CHECK(hipMemset(c, 1, bytes));
  }
  unsigned short* X;
  CHECK(hipMalloc(&X, (size_t)512 * K * 2));
  CHECK(hipMemset(X, 1, (size_t)512 * K * 2));
  printf("gate|up shape: N=%d K=%d\n", N, K);
  // weights only / activations only / both, 4 .. 16 issuing waves
  run<128, 0, 4, 4>(copies, X, 140, K, sink);
  run<128, 0, 8, 4>(copies, X, 140, K, sink);
  run<128, 0, 16, 4>(copies, X, 140, K, sink);
  run<0, 256, 4, 3>(copies, X, 140, K, sink);
  run<0, 256, 8, 3>(copies, X, 140, K, sink);
  run<0, 256, 16, 3>(copies, X, 140, K, sink);
  run<0, 256, 16, 4>(copies, X, 140, K, sink);
  run<0, 256, 8, 3>(copies, X, 256, K, sink);
  run<128, 256, 4, 3>(copies, X, 140, K, sink);
  run<128, 256, 8, 3>(copies, X, 140, K, sink);
  run<128, 256, 12, 3>(copies, X, 140, K, sink);
  run<128, 256, 16, 3>(copies, X, 140, K, sink);
  run<128, 128, 8, 4>(copies, X, 140, K, sink);
  run<128, 128, 16, 4>(copies, X, 140, K, sink);
  run<64, 256, 8, 3>(copies, X, 280, K, sink);
  run<64, 256, 8, 4>(copies, X, 280, K, sink);
  return 0;
}
