// Cell list shared by the neighbour-row builders (oxDNA: neighbors.hip, MARTINI: martini_md.hip): a hashed (free
// space) or direct-mapped (periodic) table of fixed-capacity buckets filled by one kernel.  Kernels are file-local
// (static) because this header is compiled into more than one translation unit.
#ifndef MYTHOS_CELL_LIST_H
#define MYTHOS_CELL_LIST_H

#include <hip/hip_runtime.h>

#include <cstdlib>

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

// ------------------------------------------------------------------------------------------------
// Fixed-capacity buckets: table slot h owns bucket[h * cap .. h * cap + cap).  Binning is then ONE kernel - a
// particle takes the next free place of its slot with one atomic - instead of count, scan (two passes), fill and
// a counter memset, five dependent launches of ~5 us each plus their gaps for a few microseconds of work.  The
// counters are double-buffered: a build counts into one half and clears the other half for the build after it.
// ------------------------------------------------------------------------------------------------
constexpr int kCellSpill = 4096;  // places of the spill list

// A bucket place holds the particle's position next to its index (x, y, z, index): the row builders read the
// candidates of a cell as one contiguous stream instead of an index list followed by a gather of positions - one
// dependent round trip less and 16-byte loads that coalesce (the gathers kept the texture addressers busy for half
// of the MARTINI builder's time).
template <typename R> struct CellPlace;
template <> struct CellPlace<float> { using type = float4; };
template <> struct CellPlace<double> { using type = double4; };
__device__ __forceinline__ float cell_index_as_real(int i, float) { return __int_as_float(i); }  // bits, never computed with
__device__ __forceinline__ double cell_index_as_real(int i, double) { return (double)i; }
__device__ __forceinline__ int cell_index_of(float w) { return __float_as_int(w); }
__device__ __forceinline__ int cell_index_of(double w) { return (int)w; }

// With sites (oxDNA rows built from MD frames): two more arrays of the same shape behind the first, the particle's
// backbone offset and base vector - what the row builder's site criteria need of a candidate.  They used to be gathered
// by index once the candidate's place had arrived (a second dependent round trip per sweep, 16-byte gathers that do not
// coalesce); now the three records of a candidate sit at the same offset of three streams and are requested together.
struct CellBins {
  void* place = nullptr;    // [H * cap] CellPlace<R>: (x, y, z, index) of the particles of a slot; sites: [3][H * cap]
  int* cnt_cur = nullptr;   // [H + 1] this build's counters (zero on entry); [H] counts the spill list
  int* cnt_next = nullptr;  // [H + 1] cleared by this build
  int* bucket = nullptr;    // [H * cap] the indices alone (sorting, home lookups)
  int* spill = nullptr;     // [kCellSpill] particles that found their bucket full: candidates for everybody
  int H = 0, cap = 0;
  bool sites = false;
  static size_t half(int H) { return (size_t)4 * ((H + 1 + 3) / 4); }
  static size_t place_ints(int H, int cap, size_t real_bytes, bool sites = false) {  // 4 reals / 4 B
    return (size_t)H * cap * real_bytes * (sites ? 3 : 1);
  }
  static size_t ints(int H, int cap, size_t real_bytes, bool sites = false) {
    return place_ints(H, cap, real_bytes, sites) + 2 * half(H) + (size_t)H * cap + kCellSpill;
  }
  // view on one allocation of ints(H, cap, sizeof(R)) ints whose counter halves were zeroed when it was made (see
  // zero_offset / zero_ints); phase flips per build
  CellBins(int* base, int H_, int cap_, size_t real_bytes, int phase, bool sites_ = false)
      : place(base), H(H_), cap(cap_), sites(sites_) {
    int* c = base + place_ints(H_, cap_, real_bytes, sites_);
    cnt_cur = c + (phase & 1) * half(H_);
    cnt_next = c + ((phase & 1) ^ 1) * half(H_);
    bucket = c + 2 * half(H_);
    spill = bucket + (size_t)H_ * cap_;
  }
  static size_t zero_offset(int H, int cap, size_t real_bytes, bool sites = false) { return place_ints(H, cap, real_bytes, sites); }
  static size_t zero_ints(int H) { return 2 * half(H); }
};

// A full bucket does not lose particles: they go to the spill list, which every row builder sweeps after its 27
// cells (hash collisions come and go as molecules move, so a bucket can fill up in the middle of a run; the list
// is empty otherwise).  Only a spill list that itself overflows is an error (overflow[1]).
// Particles that follow each other in memory mostly follow each other in space, so the 64 lanes of a wavefront name
// only a handful of slots, in runs.  One atomic per run (by its first lane, for the whole run) instead of one per
// lane: same-address atomics serialise in L2 (15 beads per cell: 11 us of binning for 20 480 MARTINI beads).
struct SlotRun {
  int head;  // first lane of the run of equal slots this lane belongs to
  int len;   // length of that run (meaningful in the head lane)
};
__device__ __forceinline__ SlotRun slot_run(int slot, int lane) {
  const int prev = __builtin_amdgcn_update_dpp(-2, slot, 0x138, 0xF, 0xF, false);  // wave_shr:1, lane 0 sees -2
  const unsigned long long heads = __ballot(slot != prev);
  const unsigned long long upto = heads & (~0ull >> (63 - lane));
  const unsigned long long above = (lane == 63) ? 0ull : (heads >> (lane + 1));
  SlotRun r;
  r.head = 63 - __clzll((long long)upto);
  r.len = above ? __ffsll((long long)above) : (64 - lane);
  return r;
}

template <typename R, bool VEC4>
static __global__ __launch_bounds__(256) void cell_bin_kernel(int n, const R* __restrict__ pos, const CellGrid<R> g,
                                                             int H, int* __restrict__ cnt_cur,
                                                             int* __restrict__ cnt_next, int* __restrict__ bucket,
                                                             int cap, int* __restrict__ spill,
                                                             typename CellPlace<R>::type* __restrict__ place,
                                                             int* __restrict__ overflow,
                                                             const typename CellPlace<R>::type* __restrict__ off,
                                                             const typename CellPlace<R>::type* __restrict__ a1) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  if (i <= H) cnt_next[i] = 0;
  constexpr int S = VEC4 ? 4 : 3;
  int h = -1;  // lanes past the last particle stay in the wavefront for the run detection
  R x = R(0), y = R(0), z = R(0);
  if (i < n) {
    x = pos[S * i], y = pos[S * i + 1], z = pos[S * i + 2];
    int cx, cy, cz;
    cell_of(g, x, y, z, cx, cy, cz);
    h = cell_slot(g, cx, cy, cz);
  }
  const SlotRun r = slot_run(h, lane);
  int base = 0;
  if (r.head == lane && h >= 0) base = atomicAdd(&cnt_cur[h], r.len);
  base = __shfl(base, r.head, 64);
  if (h < 0) return;
  const int p = base + lane - r.head;
  if (p < cap) {
    bucket[(size_t)h * cap + p] = i;
    typename CellPlace<R>::type pl;
    pl.x = x, pl.y = y, pl.z = z, pl.w = cell_index_as_real(i, R(0));
    place[(size_t)h * cap + p] = pl;
    if (off) {  // (real4 per particle, as the MD frames hold them)
      const size_t HC = (size_t)H * cap;
      place[HC + (size_t)h * cap + p] = off[i];
      place[2 * HC + (size_t)h * cap + p] = a1[i];
    }
  } else {
    const int q = atomicAdd(&cnt_cur[H], 1);
    if (q < kCellSpill)
      spill[q] = i;
    else
      atomicMax(&overflow[1], q + 1);
  }
  if (2 * (p + 1) > cap) atomicMax(&overflow[2], p + 1);  // more than half full: headroom hint, rare by design
}

// One wavefront per slot: orders a bucket by particle index (the atomics above land in any order), for row
// builders that copy candidates in bucket order.  Up to 64 entries by a bitonic network on registers; up to kCellSortLds
// by rank through an LDS copy (indices are distinct: an entry's rank is the number of smaller ones - count^2 / 64 LDS
// reads per lane; until round 4 these buckets were insertion-sorted by ONE lane in global memory, 220 us per build of
// a MARTINI bilayer at a 0.6 nm skin, where cells hold ~75 beads, against 5 us); longer ones by one lane serially.
constexpr int kCellSortLds = 512;
template <typename R, bool VEC4>
static __global__ __launch_bounds__(256) void cell_sort_bins_kernel(int H, const int* __restrict__ cnt,
                                                                    int* __restrict__ bucket, int cap,
                                                                    int* __restrict__ spill,
                                                                    const R* __restrict__ pos,
                                                                    typename CellPlace<R>::type* __restrict__ place) {
  const int h = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (h > H) return;
  const int count = min(cnt[h], h < H ? cap : kCellSpill);  // h == H: the spill list
  if (count <= 1) return;
  int* b = h < H ? bucket + (size_t)h * cap : spill;
  constexpr int S = VEC4 ? 4 : 3;
  if (count <= 64) {
    int v = (lane < count) ? b[lane] : 0x7fffffff;
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
    if (lane < count) b[lane] = v;
  } else if (count <= kCellSortLds) {
    __shared__ int s_sort[4][kCellSortLds];
    int* sb = s_sort[threadIdx.x >> 6];
    for (int k = lane; k < count; k += 64) sb[k] = b[k];
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    for (int k = lane; k < count; k += 64) {
      const int v = sb[k];
      int r = 0;
      for (int q = 0; q < count; ++q) r += (sb[q] < v) ? 1 : 0;
      b[r] = v;
    }
  } else if (lane == 0) {
    for (int a = 1; a < count; ++a) {
      const int v = b[a];
      int q = a - 1;
      while (q >= 0 && b[q] > v) {
        b[q + 1] = b[q];
        --q;
      }
      b[q + 1] = v;
    }
  }
  if (h == H) return;  // the spill list holds indices only
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  for (int k = lane; k < count; k += 64) {  // the places follow the new order
    const int j = b[k];
    typename CellPlace<R>::type pl;
    pl.x = pos[S * j], pl.y = pos[S * j + 1], pl.z = pos[S * j + 2], pl.w = cell_index_as_real(j, R(0));
    place[(size_t)h * cap + k] = pl;
  }
}

// mythos_debug_set(MYTHOS_DEBUG_CELL_BUCKET_CAP, places): fixes the bucket capacity (no growth), e.g. to a handful of
// places to exercise the spill path in tests.  0: managed by the callers.
long long debug_value(int key);
static inline int cell_cap_override() {
  const long long v = debug_value(0 /* MYTHOS_DEBUG_CELL_BUCKET_CAP */);
  return v > 0 ? (int)v : 0;
}

static inline int next_pow2(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

// bins n particles at pos (stride 3 or 4 reals) into b; overflow: int[3] of the caller ([0] rows, [1] spill list over
// capacity, [2] fullest bucket if over half its capacity), not cleared here.  sort_buckets: order every bucket by particle index afterwards (one more launch).
// off, a1 (with b.sites; real4 per particle): copied into the two site streams; the bucket sort moves the first stream only
template <typename R, bool VEC4>
static inline void cell_bins_build(int n, const R* pos, CellGrid<R>& g, const CellBins& b, int* overflow,
                                   bool sort_buckets, hipStream_t st, const R* off = nullptr, const R* a1 = nullptr) {
  if (!g.direct) g.hmask = b.H - 1;
  const int threads = n > b.H + 1 ? n : b.H + 1;
  using PL = typename CellPlace<R>::type;
  const bool sites = b.sites && off && a1 && !sort_buckets;
  hipLaunchKernelGGL((cell_bin_kernel<R, VEC4>), dim3((threads + 255) / 256), dim3(256), 0, st, n, pos, g, b.H, b.cnt_cur,
                     b.cnt_next, b.bucket, b.cap, b.spill, (PL*)b.place, overflow, sites ? (const PL*)off : nullptr,
                     sites ? (const PL*)a1 : nullptr);
  if (sort_buckets)
    hipLaunchKernelGGL((cell_sort_bins_kernel<R, VEC4>), dim3((b.H + 1 + 3) / 4), dim3(256), 0, st, b.H, b.cnt_cur, b.bucket, b.cap,
                       b.spill, pos, (typename CellPlace<R>::type*)b.place);
}

}  // namespace mythos

#endif  // MYTHOS_CELL_LIST_H
