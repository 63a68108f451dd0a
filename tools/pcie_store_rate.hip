// How fast a kernel's stores reach pinned host memory over PCIe, for transfers of the size the
// cascade-hash compaction writes per batch (a few megabytes) and for large ones:
//     hipcc -O3 --offload-arch=gfx950 -o tools/pcie_store_rate tools/pcie_store_rate.hip && tools/pcie_store_rate
// Prints GB/s per (bytes, bytes per lane and store, workgroups).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CHECK(x)                                                                    \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) {                                                         \
      std::printf("%s: %s\n", #x, hipGetErrorString(e_));                           \
      return 1;                                                                     \
    }                                                                               \
  } while (0)

template <class T>
__global__ __launch_bounds__(256) void k_store(T* __restrict__ dst, size_t n, T v) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = v;
}

template <class T>
int run(const char* what, void* host_dev, size_t bytes, int blocks, T v) {
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  const size_t n = bytes / sizeof(T);
  for (int k = 0; k < 3; k++) hipLaunchKernelGGL(k_store<T>, dim3(blocks), dim3(256), 0, 0, (T*)host_dev, n, v);
  CHECK(hipDeviceSynchronize());
  const int reps = bytes > (32u << 20) ? 5 : 40;
  CHECK(hipEventRecord(a, 0));
  for (int k = 0; k < reps; k++) hipLaunchKernelGGL(k_store<T>, dim3(blocks), dim3(256), 0, 0, (T*)host_dev, n, v);
  CHECK(hipEventRecord(b, 0));
  CHECK(hipEventSynchronize(b));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, a, b));
  std::printf("%9zu bytes, %2zu B per lane and store (%s), %4d workgroups: %6.1f us per launch, %5.1f GB/s\n", bytes, sizeof(T),
              what, blocks, ms * 1e3 / reps, (double)bytes * reps / (ms * 1e-3) / 1e9);
  return 0;
}

int main() {
  const size_t cap = (size_t)256 << 20;
  void* host = nullptr;
  CHECK(hipHostMalloc(&host, cap, hipHostMallocDefault));
  void* dev = nullptr;
  CHECK(hipHostGetDevicePointer(&dev, host, 0));
  for (size_t bytes : {(size_t)3718700 / 16 * 16, (size_t)16 << 20, (size_t)256 << 20})
    for (int blocks : {64, 496, 2048}) {
      if (run<uint32_t>("dword", dev, bytes, blocks, 7u)) return 1;
      if (run<uint4>("dwordx4", dev, bytes, blocks, make_uint4(1, 2, 3, 4))) return 1;
    }
  // device-to-host copy engine for comparison
  void* d = nullptr;
  CHECK(hipMalloc(&d, cap));
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  for (size_t bytes : {(size_t)3718700, (size_t)256 << 20}) {
    CHECK(hipMemcpy(host, d, bytes, hipMemcpyDeviceToHost));
    CHECK(hipEventRecord(a, 0));
    for (int k = 0; k < 5; k++) CHECK(hipMemcpyAsync(host, d, bytes, hipMemcpyDeviceToHost, 0));
    CHECK(hipEventRecord(b, 0));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    std::printf("%9zu bytes, hipMemcpyAsync device to host: %6.1f us per copy, %5.1f GB/s\n", bytes, ms * 1e3 / 5,
                (double)bytes * 5 / (ms * 1e-3) / 1e9);
  }
  return 0;
}
