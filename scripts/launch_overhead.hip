// Dev microbenchmark: per-launch cost of back-to-back kernels as a function of kernarg size, static
// LDS and scratch (private segment) use.  hipcc --offload-arch=gfx950 -O3 -o launch_overhead launch_overhead.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

struct Big { float v[300]; };

__global__ void k_plain(float* out) { if (out && threadIdx.x == 9999) out[0] = 1; }
__global__ void k_args(Big b, float* out, int idx) { if (out && threadIdx.x == 9999) out[0] = b.v[idx]; }
__global__ void k_lds(float* out) {
  __shared__ float s[9800];
  s[threadIdx.x] = threadIdx.x;
  __syncthreads();
  if (out && threadIdx.x == 9999) out[0] = s[(threadIdx.x * 7) % 9800];
}
template <int N>
__global__ void k_scratch(float* out, int idx) {
  float a[N];  // dynamically indexed -> private segment; only ONE store and one load execute
  a[(idx * 3 + threadIdx.x) % N] = 1.f;
  a[(idx * 5 + threadIdx.x) % N] = threadIdx.x;
  float r = a[(idx + threadIdx.x) % N];
  if (out && r == -1.f) out[0] = r;
}

template <class F>
static void timeit(const char* name, F launch, hipStream_t st) {
  for (int i = 0; i < 200; ++i) launch();
  hipStreamSynchronize(st);
  auto t0 = std::chrono::steady_clock::now();
  const int n = 3000;
  for (int i = 0; i < n; ++i) launch();
  hipStreamSynchronize(st);
  double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
  printf("%-28s %.2f us/launch\n", name, us);
}

int main() {
  hipStream_t st;
  hipStreamCreate(&st);
  float* d;
  hipMalloc(&d, 1024);
  Big b{};
  for (int blocks : {1, 750}) {
    printf("blocks = %d x 256 threads\n", blocks);
    timeit("plain", [&] { hipLaunchKernelGGL(k_plain, dim3(blocks), dim3(256), 0, st, d); }, st);
    timeit("kernarg 1.2 KB", [&] { hipLaunchKernelGGL(k_args, dim3(blocks), dim3(256), 0, st, b, d, 3); }, st);
    timeit("LDS 39 KB", [&] { hipLaunchKernelGGL(k_lds, dim3(blocks), dim3(256), 0, st, d); }, st);
    timeit("scratch 64 B", [&] { hipLaunchKernelGGL(k_scratch<16>, dim3(blocks), dim3(256), 0, st, d, 3); }, st);
    timeit("scratch 512 B", [&] { hipLaunchKernelGGL(k_scratch<128>, dim3(blocks), dim3(256), 0, st, d, 3); }, st);
    timeit("scratch 2 KB", [&] { hipLaunchKernelGGL(k_scratch<512>, dim3(blocks), dim3(256), 0, st, d, 3); }, st);
  }
  return 0;
}
