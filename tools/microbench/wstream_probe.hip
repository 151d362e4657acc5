// How fast can a decode GEMM's weight panels be streamed into LDS, as a function of their layout in memory?
//
//   hipcc --offload-arch=gfx950 -O3 -o wstream_probe tools/microbench/wstream_probe.hip && ./wstream_probe
//
// Every workgroup streams one panel of ROWS weight rows x K columns (bf16) through LDS in K steps of 64 columns,
// exactly as the decode GEMM stages its weight tiles (16-byte LDS-DMA loads, 8 rows x 128 B per wave instruction,
// STAGES buffers, counted vmcnt), and does nothing else. Layouts:
//   row-major  W[n][k]                       - a K step touches ROWS separate 128-byte pieces, K*2 bytes apart
//   tiled      W[n / ROWS][k / 64][ROWS][64] - a K step is one contiguous ROWS * 128-byte block
// Several copies of the matrix are cycled so that every launch reads from HBM (not the 256 MB Infinity Cache).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                      \
  do {                                                                                \
    hipError_t e = (x);                                                               \
    if (e != hipSuccess) {                                                            \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e));        \
      exit(1);                                                                        \
    }                                                                                 \
  } while (0)

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int N>
__device__ inline void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// ROWS rows per panel, 256 threads (4 waves), every wave stages ROWS / 32 pieces of 8 rows per step.
template <int ROWS, int STAGES, bool TILED>
__global__ __launch_bounds__(256) void stream_kernel(const unsigned short* __restrict__ W, int K, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int IPW = ROWS / 32;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int panel = blockIdx.x;
  const int nsteps = K / 64;
  const unsigned short* src[IPW];
#pragma unroll
  for (int j = 0; j < IPW; ++j) {
    const int row = 8 * (wave * IPW + j) + (lane >> 3);
    const int c = lane & 7;
    if (TILED)
      src[j] = W + ((size_t)panel * nsteps * ROWS + row) * 64 + 8 * c;  // + step * ROWS * 64
    else
      src[j] = W + ((size_t)panel * ROWS + row) * K + 8 * c;  // + step * 64
  }
  auto issue = [&](int buf, int step) {
    char* base = smem + buf * (ROWS * 128) + wave * (IPW * 1024);
#pragma unroll
    for (int j = 0; j < IPW; ++j) {
      const unsigned short* p = src[j] + (TILED ? (size_t)step * ROWS * 64 : (size_t)step * 64);
      __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(base + j * 1024), 16, 0, 0);
    }
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
    // touch the staged tile so that the loads cannot be dropped
    acc += *reinterpret_cast<const float*>(smem + (t % STAGES) * (ROWS * 128) + threadIdx.x * 16);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  if (acc == 12345.678f) sink[blockIdx.x] = acc;
}

template <int ROWS, int STAGES, bool TILED>
float run(const std::vector<unsigned short*>& copies, int N, int K, float* sink, int iters) {
  const size_t lds = (size_t)STAGES * ROWS * 128;
  CHECK(hipFuncSetAttribute((const void*)stream_kernel<ROWS, STAGES, TILED>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (auto* w : copies) hipLaunchKernelGGL((stream_kernel<ROWS, STAGES, TILED>), dim3(N / ROWS), dim3(256), lds, 0, w, K, sink);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i)
    for (auto* w : copies) hipLaunchKernelGGL((stream_kernel<ROWS, STAGES, TILED>), dim3(N / ROWS), dim3(256), lds, 0, w, K, sink);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3f / (iters * copies.size());  // us per launch
}

int main() {
  const int shapes[][2] = {{17920, 1536}, {1536, 8960}, {2048, 1536}, {151936, 1536}};
  float* sink;
  CHECK(hipMalloc(&sink, 1 << 20));
  for (auto& sh : shapes) {
    const int N = sh[0] / 128 * 128, K = sh[1];
    const size_t bytes = (size_t)N * K * 2;
    const int ncopy = (int)(400e6 / bytes) + 2;
    std::vector<unsigned short*> copies(ncopy);
    for (auto& c : copies) {
      CHECK(hipMalloc(&c, bytes));
      CHECK(hipMemset(c, 1, bytes));
    }
    printf("N=%d K=%d (%.1f MB, %d copies)\n", N, K, bytes / 1e6, ncopy);
#define ROW(R, S)                                                                                              \
  {                                                                                                            \
    const float a = run<R, S, false>(copies, N, K, sink, 20), b = run<R, S, true>(copies, N, K, sink, 20);     \
    printf("  panels of %3d rows (%4d workgroups), %d stages: row-major %7.1f us %5.2f TB/s | tiled %7.1f us " \
           "%5.2f TB/s\n", R, N / R, S, a, bytes / a / 1e6, b, bytes / b / 1e6);                               \
  }
    ROW(128, 2) ROW(128, 3) ROW(128, 4) ROW(64, 4) ROW(256, 3) ROW(32, 4)
    for (auto c : copies) CHECK(hipFree(c));
  }
  return 0;
}
