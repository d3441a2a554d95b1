// Spatial order of the MD kernels' workgroup chunks (host side).
//
// The step kernels give XCD x the x-th contiguous eighth of a list of chunks (32 particles each), so that the
// neighbours a workgroup reads are mostly in its own XCD's L2.  In index order that holds only when index
// neighbours are space neighbours; the two strands of a duplex are stored one after the other and run antiparallel,
// so the partners of a chunk are at the other end of the index range.  Listing the chunks in the order of the Morton
// code of their position makes the eighths contiguous in space whatever the storage order (12 kbp duplex: +2.6 %).
// Not used by the MARTINI kernel: the tiled bilayer is stored tile by tile, molecule by molecule, already compact,
// and ordering its 32-bead chunks by their first bead made it 9 % slower.
#ifndef MYTHOS_CHUNK_ORDER_H
#define MYTHOS_CHUNK_ORDER_H

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <utility>
#include <vector>

namespace mythos {

// pos: device array of V4 (x, y, z, *), one per particle; chunk c is represented by particle c * per_chunk.
// *d_order is allocated on first use ([blocks] ints).  Synchronises the stream.  Returns a hipError_t.
template <typename V4>
static inline hipError_t chunk_order_update(const V4* pos, int blocks, int per_chunk, double cell, int** d_order,
                                            hipStream_t st) {
  std::vector<V4> rep((size_t)blocks);
  hipError_t e = hipMemcpy2DAsync(rep.data(), sizeof(V4), pos, (size_t)per_chunk * sizeof(V4), sizeof(V4), (size_t)blocks,
                                  hipMemcpyDeviceToHost, st);
  if (e != hipSuccess) return e;
  if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;
  double lo[3] = {1e300, 1e300, 1e300};
  for (const V4& v : rep)
    lo[0] = std::min(lo[0], (double)v.x), lo[1] = std::min(lo[1], (double)v.y), lo[2] = std::min(lo[2], (double)v.z);
  auto spread = [](uint64_t x) {  // 21 bits -> every third bit
    x &= 0x1fffff;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
  };
  std::vector<std::pair<uint64_t, int>> key((size_t)blocks);
  for (int c = 0; c < blocks; ++c) {
    const V4& v = rep[(size_t)c];
    const uint64_t ix = (uint64_t)(((double)v.x - lo[0]) / cell), iy = (uint64_t)(((double)v.y - lo[1]) / cell),
                   iz = (uint64_t)(((double)v.z - lo[2]) / cell);
    key[(size_t)c] = {spread(ix) | spread(iy) << 1 | spread(iz) << 2, c};
  }
  std::sort(key.begin(), key.end());
  std::vector<int> order((size_t)blocks);
  for (int c = 0; c < blocks; ++c) order[(size_t)c] = key[(size_t)c].second;
  if (!*d_order && (e = hipMalloc((void**)d_order, (size_t)blocks * sizeof(int))) != hipSuccess) return e;
  if ((e = hipMemcpyAsync(*d_order, order.data(), (size_t)blocks * sizeof(int), hipMemcpyHostToDevice, st)) != hipSuccess) return e;
  return hipStreamSynchronize(st);  // order[] leaves scope
}

}  // namespace mythos

#endif  // MYTHOS_CHUNK_ORDER_H
