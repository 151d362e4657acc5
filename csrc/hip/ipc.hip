// Peer-memory weight shipping: a trainer process exports the device allocation that holds its flat bf16
// parameter buffer, a rollout process maps it and pulls the bytes with an asynchronous device-to-device copy
// on a side stream (over xGMI when the two processes own different GPUs of a node, a local copy when they
// share one). Host-side wrappers over the HIP IPC API with a C ABI for ctypes.
//
// Replaces the reference's weight funnel (every parameter all-gathered on the trainer, then broadcast from
// trainer rank 0 into every server rank: areal/engine/fsdp_engine.py:399-433) for layouts whose flat buffers
// match: no collective, no process group between trainer and servers, one copy per server.
#include <hip/hip_runtime.h>
#include <string.h>

#define IPC_RET(expr)                      \
  do {                                     \
    hipError_t _e = (expr);                \
    if (_e != hipSuccess) return (int)_e;  \
  } while (0)

// handle_out: 64 bytes. offset_out: byte offset of `ptr` inside its allocation (the handle names the
// allocation's base). alloc_size_out: size of that allocation.
extern "C" int areal_ipc_export(const void* ptr, unsigned char* handle_out, long* offset_out, long* alloc_size_out) {
  hipDeviceptr_t base = nullptr;
  size_t size = 0;
  IPC_RET(hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)ptr));
  hipIpcMemHandle_t h;
  IPC_RET(hipIpcGetMemHandle(&h, (void*)base));
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "handle size");
  memcpy(handle_out, &h, sizeof(h));
  *offset_out = (long)((const char*)ptr - (const char*)base);
  *alloc_size_out = (long)size;
  return 0;
}

// A device allocation of its own (not carved out of a framework's caching allocator), so that the handle
// names exactly `bytes` bytes: mapping an exported allocation of 2 GiB or more does not return on this
// platform (ROCm 7.2, dmabuf IPC), so shipping buffers are exported in pieces below that size.
extern "C" int areal_ipc_alloc(long bytes, void** out) {
  IPC_RET(hipMalloc(out, (size_t)bytes));
  return 0;
}

// The same, but uncached / fine-grained: memory that kernels of SEVERAL devices write while a kernel of this device
// polls or reads it (flag tables and result slots of the peer-memory all-reduce). Lines of an ordinary allocation
// may sit in this device's L2, which a peer's stores over xGMI do not pass through; uncached memory has no such copy.
extern "C" int areal_ipc_alloc_uncached(long bytes, void** out) {
  IPC_RET(hipExtMallocWithFlags(out, (size_t)bytes, hipDeviceMallocUncached));
  return 0;
}

extern "C" int areal_ipc_free(void* p) {
  IPC_RET(hipFree(p));
  return 0;
}

extern "C" int areal_ipc_open(const unsigned char* handle, void** base_out) {
  hipIpcMemHandle_t h;
  memcpy(&h, handle, sizeof(h));
  IPC_RET(hipIpcOpenMemHandle(base_out, h, hipIpcMemLazyEnablePeerAccess));
  return 0;
}

extern "C" int areal_ipc_close(void* base) {
  IPC_RET(hipIpcCloseMemHandle(base));
  return 0;
}

// Copy between any two addresses of the unified address space (device, peer-mapped or pinned host memory): the
// runtime works out the direction.
extern "C" int areal_memcpy_async(void* dst, const void* src, long bytes, hipStream_t st) {
  if (bytes <= 0) return 0;
  IPC_RET(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDefault, st));
  return 0;
}

extern "C" int areal_memcpy_d2d_async(void* dst, const void* src, long bytes, hipStream_t st) {
  if (bytes <= 0) return 0;
  IPC_RET(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToDevice, st));
  return 0;
}

// ----------------------------------------------------------------------------------------------------------
// Many (source offset, destination offset, length) byte ranges from one mapped peer region into a local buffer
// in ONE launch: the copy of a weight pull between DIFFERENT parallel layouts (a tensor- / pipeline-parallel
// trainer shard holds row blocks and row slices of the parameters a server rank needs: up to one range per
// matrix row). A workgroup walks ranges; 16-byte vectors where both ends are 16-byte aligned, bytes otherwise.
// Reads cross xGMI when the region lives on another GPU, writes are local.
// ----------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void interval_move_kernel(const char* __restrict__ src, char* __restrict__ dst,
                                                            const long* __restrict__ triples, int n) {
  for (int i = blockIdx.x; i < n; i += gridDim.x) {
    const long so = triples[3 * i], d_o = triples[3 * i + 1], len = triples[3 * i + 2];
    const char* s = src + so;
    char* d = dst + d_o;
    if ((((unsigned long)s | (unsigned long)d) & 15) == 0) {
      const long nv = len >> 4;
      for (long k = (long)blockIdx.y * blockDim.x + threadIdx.x; k < nv; k += (long)gridDim.y * blockDim.x)
        reinterpret_cast<uint4*>(d)[k] = reinterpret_cast<const uint4*>(s)[k];
      for (long k = (nv << 4) + (long)blockIdx.y * blockDim.x + threadIdx.x; k < len; k += (long)gridDim.y * blockDim.x)
        d[k] = s[k];
    } else {
      for (long k = (long)blockIdx.y * blockDim.x + threadIdx.x; k < len; k += (long)gridDim.y * blockDim.x) d[k] = s[k];
    }
  }
}

// triples: device array of n x (src byte offset, dst byte offset, byte length); max_len: the longest range.
extern "C" int areal_interval_move(const void* src, void* dst, const long* triples, int n, long max_len, hipStream_t st) {
  if (n <= 0) return 0;
  long gy = (max_len / 16 + 256 * 4 - 1) / (256 * 4);  // ~4 vectors per thread before a second block row
  gy = gy < 1 ? 1 : (gy > 128 ? 128 : gy);
  const int gx = n > 2048 ? 2048 : n;
  hipLaunchKernelGGL(interval_move_kernel, dim3(gx, (unsigned)gy), dim3(256), 0, st, (const char*)src, (char*)dst,
                     triples, n);
  return (int)hipGetLastError();
}
