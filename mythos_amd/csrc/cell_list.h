// Hashed cell list shared by the neighbour-row builders (oxDNA: neighbors.hip, MARTINI: martini_md.hip):
// count -> two-pass coalesced scan -> fill -> per-bucket sort.  Buckets are sorted by particle index so the
// rows built from them are deterministic.  Kernels are file-local (static) because this header is compiled
// into more than one translation unit.
#ifndef MYTHOS_CELL_LIST_H
#define MYTHOS_CELL_LIST_H

#include <hip/hip_runtime.h>

namespace mythos {

template <typename R>
struct CellGrid {
  R inv[3];   // 1 / cell edge
  R ibox[3];  // 1 / box edge (periodic)
  int nc[3];  // cells per box edge (periodic) or 0 (free space)
  int hmask;
  int direct = 0;  // 1: slots are linear cell indices (periodic grid), no hashing
};

template <typename R>
__device__ __forceinline__ void cell_of(const CellGrid<R>& g, R x, R y, R z, int& cx, int& cy, int& cz) {
  const R p[3] = {x, y, z};
  int c[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    if (g.nc[k] > 0) {
      R f = p[k] * g.ibox[k];
      f -= floor(f);
      int v = (int)(f * R(g.nc[k]));
      c[k] = v >= g.nc[k] ? g.nc[k] - 1 : v;
    } else {
      c[k] = (int)floor(p[k] * g.inv[k]);
    }
  }
  cx = c[0], cy = c[1], cz = c[2];
}

__device__ __forceinline__ int cell_hash(int cx, int cy, int cz, int hmask) {
  return (int)(((unsigned)cx * 73856093u) ^ ((unsigned)cy * 19349663u) ^ ((unsigned)cz * 83492791u)) & hmask;
}

// table slot of a cell: hashed (free space, or a periodic grid larger than the table) or, when the periodic
// grid fits, the cell's own linear index - then a bucket holds exactly one cell and candidates need no check
template <typename R>
__device__ __forceinline__ int cell_slot(const CellGrid<R>& g, int cx, int cy, int cz) {
  if (g.direct) return (cz * g.nc[1] + cy) * g.nc[0] + cx;
  return cell_hash(cx, cy, cz, g.hmask);
}

template <typename R, bool VEC4>
__global__ void cell_count_kernel(int n, const R* __restrict__ pos, const CellGrid<R> g, int* __restrict__ slot_of,
                                  int* __restrict__ cnt) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  constexpr int S = VEC4 ? 4 : 3;
  int cx, cy, cz;
  cell_of(g, pos[S * i], pos[S * i + 1], pos[S * i + 2], cx, cy, cz);
  const int h = cell_slot(g, cx, cy, cz);
  slot_of[i] = h;
  atomicAdd(&cnt[h], 1);
}

// exclusive scan of cnt[0..m) into start[0..m] in two coalesced passes; cnt is cleared for reuse as a
// cursor.  Pass 1: each 1024-thread workgroup scans 4096 counters (int4 per lane, wave shuffles + one LDS
// hop) and publishes its total.  Pass 2 adds the totals of the preceding workgroups.
constexpr int kScanBlock = 1024;
constexpr int kScanPerBlock = 4 * kScanBlock;

static __global__ __launch_bounds__(kScanBlock) void cell_scan_local_kernel(int m, int* __restrict__ cnt,
                                                                      int* __restrict__ start,
                                                                      int* __restrict__ block_sum) {
  __shared__ int wave_tot[kScanBlock / 64];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int base = (blockIdx.x * kScanBlock + t) * 4;
  int4 v = make_int4(0, 0, 0, 0);
  if (base + 3 < m) {
    v = *reinterpret_cast<const int4*>(cnt + base);
    *reinterpret_cast<int4*>(cnt + base) = make_int4(0, 0, 0, 0);
  } else {
    int* pv = &v.x;
    for (int k = 0; k < 4; ++k)
      if (base + k < m) {
        pv[k] = cnt[base + k];
        cnt[base + k] = 0;
      }
  }
  const int s = v.x + v.y + v.z + v.w;
  int inc = s;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int u = __shfl_up(inc, o, 64);
    if (lane >= o) inc += u;
  }
  if (lane == 63) wave_tot[w] = inc;
  __syncthreads();
  if (t == 0) {
    int run = 0;
    for (int k = 0; k < kScanBlock / 64; ++k) {
      const int x = wave_tot[k];
      wave_tot[k] = run;
      run += x;
    }
    block_sum[blockIdx.x] = run;
  }
  __syncthreads();
  const int pre = wave_tot[w] + inc - s;
  const int o4[4] = {pre, pre + v.x, pre + v.x + v.y, pre + v.x + v.y + v.z};
  for (int k = 0; k < 4; ++k)
    if (base + k < m) start[base + k] = o4[k];
}

static __global__ void cell_scan_fix_kernel(int m, int n_blocks, const int* __restrict__ block_sum, int* __restrict__ start) {
  const int h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h > m) return;
  const int b = (h < m ? h : m - 1) / kScanPerBlock;
  int off = 0;
  for (int k = 0; k < b; ++k) off += block_sum[k];
  if (h < m) {
    start[h] += off;
  } else {
    int tot = 0;
    for (int k = 0; k < n_blocks; ++k) tot += block_sum[k];
    start[m] = tot;
  }
}

static __global__ void cell_fill_kernel(int n, const int* __restrict__ slot_of, const int* __restrict__ start,
                                 int* __restrict__ cursor, int* __restrict__ bucket) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int h = slot_of[i];
  bucket[start[h] + atomicAdd(&cursor[h], 1)] = i;
}

// One wavefront per bucket: buckets of up to 64 entries (all of them in practice: a cell holds a few tens of
// particles) are sorted by a 64-lane bitonic network on registers, longer ones by one lane serially.
static __global__ __launch_bounds__(256) void cell_sort_kernel(int m, const int* __restrict__ start,
                                                               int* __restrict__ bucket, int* __restrict__ cursor) {
  const int h = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (h >= m) return;
  if (lane == 0) cursor[h] = 0;  // the fill cursor: left clean for the next build (no memset between builds)
  const int lo = start[h], cnt = start[h + 1] - lo;
  if (cnt <= 1) return;
  if (cnt <= 64) {
    int v = (lane < cnt) ? bucket[lo + lane] : 0x7fffffff;
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
      for (int j = k >> 1; j > 0; j >>= 1) {
        const int o = __shfl_xor(v, j, 64);
        const bool up = ((lane & k) == 0);        // ascending block
        const bool lower = ((lane & j) == 0);     // this lane keeps the smaller value of the pair
        v = (lower == up) ? min(v, o) : max(v, o);
      }
    }
    if (lane < cnt) bucket[lo + lane] = v;
  } else if (lane == 0) {
    for (int a = lo + 1; a < lo + cnt; ++a) {
      const int v = bucket[a];
      int b2 = a - 1;
      while (b2 >= lo && bucket[b2] > v) {
        bucket[b2 + 1] = bucket[b2];
        --b2;
      }
      bucket[b2 + 1] = v;
    }
  }
}

static inline int next_pow2(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

// Scratch layout inside one int allocation: cnt[H] start[H+1] (pad to keep int4 alignment) slot_of[n]
// bucket[n] block_sum[...].  Returns the number of ints needed.
struct CellScratch {
  int *cnt, *start, *slot_of, *bucket, *block_sum;
  static size_t ints(int H, int n) { return (size_t)2 * H + 4 + (size_t)2 * n + 1024; }
  CellScratch(int* base, int H, int n)
      : cnt(base), start(base + H), slot_of(base + 2 * (size_t)H + 4), bucket(slot_of + n), block_sum(bucket + n) {}
};

// count, scan, fill and sort for n particles at pos (stride 3 or 4 reals); H = table size (power of two)
// clean: the counters are known to be zero (left so by the previous build on the same scratch and table size)
template <typename R, bool VEC4>
static inline int cell_list_build(int n, const R* pos, CellGrid<R>& g, int H, const CellScratch& cs, bool clean,
                                  hipStream_t st) {
  if (!g.direct) g.hmask = H - 1;
  if (!clean && hipMemsetAsync(cs.cnt, 0, (size_t)H * sizeof(int), st) != hipSuccess) return -1;
  const int tb = (n + 255) / 256;
  hipLaunchKernelGGL((cell_count_kernel<R, VEC4>), dim3(tb), dim3(256), 0, st, n, pos, g, cs.slot_of, cs.cnt);
  const int nsb = (H + kScanPerBlock - 1) / kScanPerBlock;
  hipLaunchKernelGGL(cell_scan_local_kernel, dim3(nsb), dim3(kScanBlock), 0, st, H, cs.cnt, cs.start, cs.block_sum);
  hipLaunchKernelGGL(cell_scan_fix_kernel, dim3((H + 1 + 255) / 256), dim3(256), 0, st, H, nsb, cs.block_sum, cs.start);
  hipLaunchKernelGGL(cell_fill_kernel, dim3(tb), dim3(256), 0, st, n, cs.slot_of, cs.start, cs.cnt, cs.bucket);
  hipLaunchKernelGGL(cell_sort_kernel, dim3((H + 3) / 4), dim3(256), 0, st, H, cs.start, cs.bucket, cs.cnt);
  return 0;
}

}  // namespace mythos

#endif  // MYTHOS_CELL_LIST_H
