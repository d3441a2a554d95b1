// round 4, VERDICT item 4(c): does the size of the kernel-argument segment show in the boundary between two dependent
// launches?  750 workgroups of 256 threads that each spin for ~8 us (s_memrealtime, 100 MHz), launched back to back on one
// stream: the period per launch minus the spin is the boundary.  Variants: 1 pointer argument; the md_step_kernel's
// ~35 arguments with four by-value structs (~470 B); the same through hipExtLaunchKernelGGL as the integrator does.
//   hipcc --offload-arch=gfx950 -O2 scripts/exp_kernarg_r04.hip -o gpurun_out/exp_kernarg && gpurun_out/exp_kernarg
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#include <algorithm>

struct Box { float l[3], il[3]; int on; };
struct LC { float v[12]; };
struct Cut { float v[8]; unsigned m; };
struct Frame { float4* p[8]; };
struct Pseq { const float* a; const int* b; const float* c; int t; double* g; double* h; int r; };

__device__ __forceinline__ void spin(int ticks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(1);
}

__global__ __launch_bounds__(256) void k_small(int* out, int ticks) {
  spin(ticks);
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] += 1;
}

__global__ __launch_bounds__(256) void k_big(const float* Pg, const Box box, const LC K, const Cut cut, int n, const Frame in, const Frame outf,
                                             const int* rows, const int* row_len, const int* row_close, int row_stride, int extra, float kick,
                                             int do_step, unsigned long long seed, unsigned long long step, const float4* r0, const float4* r1,
                                             const float4* r2, int* flags, float* tc, float* tq, double* e_part, const int* order,
                                             const int* ovf, int k_index, int ablate, int prio, const Pseq ps, int* out, int ticks) {
  spin(ticks);
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] += (int)(box.on + K.v[0] + cut.m + n + row_stride + extra + do_step + (int)seed + (int)step + k_index + ablate + prio + ps.t) * 0 + 1;
}

int main() {
  int* d;
  hipMalloc(&d, 64);
  hipMemset(d, 0, 64);
  hipStream_t st;
  hipStreamCreate(&st);
  hipEvent_t a, b;
  hipEventCreate(&a), hipEventCreate(&b);
  const int grid = 752, n_launch = 4000;
  Box box{}; LC K{}; Cut cut{}; Frame f{}; Pseq ps{};
  for (int ticks : {0, 400, 800}) {  // 0, 4, 8 us of spin
    for (int variant = 0; variant < 3; ++variant) {
      std::vector<float> ms;
      for (int rep = 0; rep < 5; ++rep) {
        hipStreamSynchronize(st);
        hipEventRecord(a, st);
        for (int k = 0; k < n_launch; ++k) {
          if (variant == 0) hipLaunchKernelGGL(k_small, dim3(grid), dim3(256), 0, st, d, ticks);
          else if (variant == 1)
            hipLaunchKernelGGL(k_big, dim3(grid), dim3(256), 0, st, nullptr, box, K, cut, 24000, f, f, nullptr, nullptr, nullptr, 64, 0, 0.5f, 1, 1ull,
                               (unsigned long long)k, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, k, 0, 1, ps, d, ticks);
          else
            hipExtLaunchKernelGGL(k_big, dim3(grid), dim3(256), 0, st, nullptr, nullptr, 0, nullptr, box, K, cut, 24000, f, f, nullptr, nullptr, nullptr, 64, 0,
                                  0.5f, 1, 1ull, (unsigned long long)k, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, k, 0,
                                  1, ps, d, ticks);
        }
        hipEventRecord(b, st);
        hipEventSynchronize(b);
        float t;
        hipEventElapsedTime(&t, a, b);
        ms.push_back(t);
      }
      std::sort(ms.begin(), ms.end());
      printf("spin %4.1f us  %-28s period %.3f us per launch (median of 5 x %d)\n", ticks / 100.0,
             variant == 0 ? "1 pointer + 1 int" : (variant == 1 ? "md_step_kernel's arguments" : "same, hipExtLaunchKernelGGL"), 1e3 * ms[2] / n_launch, n_launch);
    }
  }
  return 0;
}
