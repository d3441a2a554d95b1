// Dev microbenchmark: per-kernel cost of back-to-back launches - stream launches vs a captured HIP graph of the
// same chain - for an empty kernel and for a ~20 us kernel (spin), 752 x 256 threads.
// hipcc --offload-arch=gfx950 -O3 -o launch_overhead launch_overhead.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

__global__ void k_empty(float* out) { if (out && threadIdx.x == 9999) out[0] = 1; }
__global__ void k_spin(float* out, long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
  if (out && threadIdx.x == 9999) out[0] = 1;
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  hipStream_t st;
  hipStreamCreate(&st);
  float* d;
  hipMalloc(&d, 1024);
  const int chain = 200, reps = 20;
  for (long long ticks : {0LL, 2000LL}) {  // 0 -> empty kernel, 2000 x 10 ns = 20 us
    auto launch = [&] {
      if (ticks == 0) hipLaunchKernelGGL(k_empty, dim3(752), dim3(256), 0, st, d);
      else hipLaunchKernelGGL(k_spin, dim3(752), dim3(256), 0, st, d, ticks);
    };
    for (int i = 0; i < chain; ++i) launch();
    hipStreamSynchronize(st);
    double t0 = now_us();
    for (int r = 0; r < reps; ++r) for (int i = 0; i < chain; ++i) launch();
    hipStreamSynchronize(st);
    const double per_stream = (now_us() - t0) / (chain * reps);
    hipGraph_t g;
    hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    for (int i = 0; i < chain; ++i) launch();
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, st);
    hipStreamSynchronize(st);
    t0 = now_us();
    for (int r = 0; r < reps; ++r) hipGraphLaunch(ge, st);
    hipStreamSynchronize(st);
    const double per_graph = (now_us() - t0) / (chain * reps);
    printf("%s kernel: stream %.2f us/kernel, graph %.2f us/kernel\n", ticks ? "20 us" : "empty", per_stream, per_graph);
  }
  return 0;
}
