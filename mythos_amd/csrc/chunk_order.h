// Spatial order of the MD kernels' workgroup chunks, computed on the device (no host round trip, no synchronisation).
//
// The step kernels give XCD x the x-th contiguous eighth of a list of chunks (32 particles each), so that the
// neighbours a workgroup reads are mostly in its own XCD's L2.  In index order that holds only when index
// neighbours are space neighbours; the two strands of a duplex are stored one after the other and run antiparallel,
// so the partners of a chunk are at the other end of the index range.  Listing the chunks in the order of the Morton
// code of their position makes the eighths contiguous in space whatever the storage order (12 kbp duplex: +2.6 %).
// Not used by the MARTINI kernel: the tiled bilayer is stored tile by tile, molecule by molecule, already compact,
// and ordering its 32-bead chunks by their first bead made it 9 % slower.
//
// Two small kernels on the caller's stream: a Morton key per chunk (from the position of its first particle, cells
// of the list range counted from the lower corner of the bounding box), then a rank sort -
// chunk c goes to place #{c' : (key, c') < (key, c)}.  The rank sort is O(chunks^2) key comparisons: 0.6 M for a
// 12 kbp duplex, 39 M for 100 kbp (a few tens of microseconds, once per load and every few dozen list rebuilds).
#ifndef MYTHOS_CHUNK_ORDER_H
#define MYTHOS_CHUNK_ORDER_H

#include <hip/hip_runtime.h>

#include <cstdint>

namespace mythos {

__device__ __forceinline__ unsigned long long morton_spread21(unsigned long long x) {  // 21 bits -> every third bit
  x &= 0x1fffffull;
  x = (x | x << 32) & 0x1f00000000ffffull;
  x = (x | x << 16) & 0x1f0000ff0000ffull;
  x = (x | x << 8) & 0x100f00f00f00f00full;
  x = (x | x << 4) & 0x10c30c30c30c30c3ull;
  x = (x | x << 2) & 0x1249249249249249ull;
  return x;
}

// ONE workgroup: the lower corner of the representatives' bounding box first (cells are counted from it: counted
// from a fixed origin, a molecule that straddles a coordinate plane - a duplex along z through the origin does, in
// x and y - has its chunks on either side of the largest power-of-two boundary of the Morton curve, as far apart in
// the order as two chunks can be), then the keys.
template <typename V4>
__global__ __launch_bounds__(256) void chunk_keys_kernel(const V4* __restrict__ pos, int blocks, int per_chunk, double inv_cell,
                                                         unsigned long long* __restrict__ keys) {
  __shared__ double lo_s[3][256];
  double lo[3] = {1e300, 1e300, 1e300};
  for (int c = threadIdx.x; c < blocks; c += 256) {
    const V4 v = pos[(size_t)c * per_chunk];
    lo[0] = fmin(lo[0], (double)v.x), lo[1] = fmin(lo[1], (double)v.y), lo[2] = fmin(lo[2], (double)v.z);
  }
  for (int k = 0; k < 3; ++k) lo_s[k][threadIdx.x] = lo[k];
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o)
      for (int k = 0; k < 3; ++k) lo_s[k][threadIdx.x] = fmin(lo_s[k][threadIdx.x], lo_s[k][threadIdx.x + o]);
    __syncthreads();
  }
  const double l0 = lo_s[0][0], l1 = lo_s[1][0], l2 = lo_s[2][0];
  auto cell = [&](double x, double l) {
    const double f = floor((x - l) * inv_cell);
    return (unsigned long long)(f < 0.0 ? 0.0 : (f > 2097151.0 ? 2097151.0 : f));
  };
  for (int c = threadIdx.x; c < blocks; c += 256) {
    const V4 v = pos[(size_t)c * per_chunk];
    keys[c] = morton_spread21(cell((double)v.x, l0)) | morton_spread21(cell((double)v.y, l1)) << 1 |
              morton_spread21(cell((double)v.z, l2)) << 2;
  }
}

static __global__ void chunk_rank_kernel(const unsigned long long* __restrict__ keys, int blocks, int* __restrict__ order) {
  __shared__ unsigned long long tile[256];
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned long long mine = c < blocks ? keys[c] : 0ull;
  int rank = 0;
  for (int base = 0; base < blocks; base += 256) {
    const int t = base + (int)threadIdx.x;
    tile[threadIdx.x] = t < blocks ? keys[t] : ~0ull;
    __syncthreads();
    const int lim = min(256, blocks - base);
    for (int k = 0; k < lim; ++k) {
      const unsigned long long o = tile[k];
      rank += (o < mine || (o == mine && base + k < c)) ? 1 : 0;
    }
    __syncthreads();
  }
  if (c < blocks) order[rank] = c;
}

// pos: device array of V4 (x, y, z, *), one per particle; chunk c is represented by particle c * per_chunk.
// d_keys [blocks] and d_order [blocks] are the caller's device buffers.  Asynchronous on st.
template <typename V4>
static inline hipError_t chunk_order_device(const V4* pos, int blocks, int per_chunk, double cell,
                                            unsigned long long* d_keys, int* d_order, hipStream_t st) {
  const int g = (blocks + 255) / 256;
  hipLaunchKernelGGL((chunk_keys_kernel<V4>), dim3(1), dim3(256), 0, st, pos, blocks, per_chunk, 1.0 / cell, d_keys);
  hipLaunchKernelGGL(chunk_rank_kernel, dim3(g), dim3(256), 0, st, (const unsigned long long*)d_keys, blocks, d_order);
  return hipGetLastError();
}

}  // namespace mythos

#endif  // MYTHOS_CHUNK_ORDER_H
