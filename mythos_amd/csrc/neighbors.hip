// Neighbour rows: conversion from a reference-style pair list and GPU Verlet-list builds.
//
// Replaces the reference's O(N^2) pair arrays (mythos/input/topology.py:186-190) and its jax_md
// neighbour-list factory (mythos/utils/neighbors.py:12-59, simulators/jax_md/utils.py:70-126)
// with per-nucleotide rows (see mythos_internal.h for the slot encoding).  Rows are written in
// ascending neighbour order so every downstream sum is reproducible.
#include <algorithm>

#include "oxdna_gather.h"

namespace mythos {

int rows_reserve(mythos_system* sys, int stride) {
  const size_t need = (size_t)sys->n * stride;
  if (need > sys->rows_cap) {
    if (sys->d_rows) (void)hipFree(sys->d_rows);
    sys->d_rows = nullptr;
    sys->rows_cap = 0;
    MYTHOS_HIP_TRY(hipMalloc((void**)&sys->d_rows, need * sizeof(int)));
    sys->rows_cap = need;
  }
  sys->row_stride = stride;
  return 0;
}

int rows_from_pairs(mythos_system* sys, const int32_t* pairs, int n_pairs) {
  const int n = sys->n;
  std::vector<int> count(n, 0);
  for (int k = 0; k < n_pairs; ++k) {
    const int i = pairs[2 * k], j = pairs[2 * k + 1];
    if (i < 0 || j < 0 || i >= n || j >= n || i == j) {
      set_error("set_neighbors: pair index out of range or i == j");
      return MYTHOS_ERR_INVALID_ARGUMENT;
    }
    ++count[i];
    ++count[j];
  }
  int mx = 0;
  for (int c : count) mx = std::max(mx, c);
  const int stride = ((mx + ROW_BONDED_SLOTS + 15) / 16) * 16;
  std::vector<int> rows((size_t)n * stride, -1), len(n, ROW_BONDED_SLOTS);
  for (int i = 0; i < n; ++i) {
    rows[(size_t)i * stride + 0] = sys->h_partners[2 * i + 0];
    rows[(size_t)i * stride + 1] = sys->h_partners[2 * i + 1];
  }
  for (int k = 0; k < n_pairs; ++k) {
    const int i = pairs[2 * k], j = pairs[2 * k + 1];
    rows[(size_t)i * stride + len[i]++] = j;               // i plays op_i
    rows[(size_t)j * stride + len[j]++] = i | ROW_ROLE_Q;  // j plays op_j
  }
  if (int rc = rows_reserve(sys, stride)) return rc;
  MYTHOS_HIP_TRY(hipMemcpy(sys->d_rows, rows.data(), rows.size() * sizeof(int), hipMemcpyHostToDevice));
  MYTHOS_HIP_TRY(hipMemcpy(sys->d_row_len, len.data(), n * sizeof(int), hipMemcpyHostToDevice));
  sys->nbrs_set = true;
  return 0;
}

// ------------------------------------------------------------------------------------------------
// all-pairs Verlet build: one wavefront per nucleotide sweeps every other nucleotide in strides
// of 64 (coalesced position reads served from L2) and appends hits by ballot compaction, which
// keeps each row in ascending index order.  O(N^2) work: the exact reference for the cell build.
// ------------------------------------------------------------------------------------------------
template <typename R, bool VEC4>
__global__ __launch_bounds__(256) void build_rows_allpairs_kernel(int n, const R* __restrict__ pos,
                                                                   const BoxT<R> box, R rc2,
                                                                   const int* __restrict__ partners_rows_in,
                                                                   int* __restrict__ rows, int* __restrict__ row_len,
                                                                   int row_stride, int* __restrict__ overflow) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (wave >= n) return;
  const int i = wave;
  constexpr int S = VEC4 ? 4 : 3;
  const V3<R> ci{pos[S * i], pos[S * i + 1], pos[S * i + 2]};
  int* row = rows + (size_t)i * row_stride;
  const int b0 = partners_rows_in[2 * i], b1 = partners_rows_in[2 * i + 1];
  int cnt = ROW_BONDED_SLOTS;
  for (int j0 = 0; j0 < n; j0 += 64) {
    const int j = j0 + lane;
    bool hit = false;
    if (j < n && j != i && j != b0 && j != b1) {
      V3<R> d{pos[S * j] - ci.x, pos[S * j + 1] - ci.y, pos[S * j + 2] - ci.z};
      d = min_image(d, box);
      hit = dot(d, d) < rc2;
    }
    const unsigned long long m = __ballot(hit);
    if (hit) {
      const int slot = cnt + __popcll(m & ((1ull << lane) - 1ull));
      if (slot < row_stride) row[slot] = (j < i) ? (j | ROW_ROLE_Q) : j;
    }
    cnt += __popcll(m);
  }
  if (lane == 0) {
    row[0] = b0;
    row[1] = b1;
    if (cnt > row_stride) {
      atomicMax(overflow, cnt);
      cnt = row_stride;
    }
    row_len[i] = cnt;
  }
}

int rows_build_device(mythos_system* sys, const void* center, bool center_is_vec4, double r_cut, double skin,
                      hipStream_t stream) {
  const int n = sys->n;
  if (sys->row_stride == 0)
    if (int rc = rows_reserve(sys, 64)) return rc;
  const double rl = r_cut + skin;
  int* d_partners = sys->d_row_len + n;  // [n][2], uploaded at creation
  const int blocks = (n * 64 + 255) / 256;
  MYTHOS_HIP_TRY(hipMemsetAsync(sys->d_overflow, 0, sizeof(int), stream));
  if (sys->dtype == MYTHOS_F32) {
    const BoxT<float> box = make_box<float>(sys);
    if (center_is_vec4)
      hipLaunchKernelGGL((build_rows_allpairs_kernel<float, true>), dim3(blocks), dim3(256), 0, stream, n,
                         (const float*)center, box, float(rl * rl), d_partners, sys->d_rows, sys->d_row_len,
                         sys->row_stride, sys->d_overflow);
    else
      hipLaunchKernelGGL((build_rows_allpairs_kernel<float, false>), dim3(blocks), dim3(256), 0, stream, n,
                         (const float*)center, box, float(rl * rl), d_partners, sys->d_rows, sys->d_row_len,
                         sys->row_stride, sys->d_overflow);
  } else {
    const BoxT<double> box = make_box<double>(sys);
    if (center_is_vec4)
      hipLaunchKernelGGL((build_rows_allpairs_kernel<double, true>), dim3(blocks), dim3(256), 0, stream, n,
                         (const double*)center, box, rl * rl, d_partners, sys->d_rows, sys->d_row_len,
                         sys->row_stride, sys->d_overflow);
    else
      hipLaunchKernelGGL((build_rows_allpairs_kernel<double, false>), dim3(blocks), dim3(256), 0, stream, n,
                         (const double*)center, box, rl * rl, d_partners, sys->d_rows, sys->d_row_len,
                         sys->row_stride, sys->d_overflow);
  }
  MYTHOS_HIP_TRY(hipGetLastError());
  sys->nbrs_set = true;
  return 0;
}

}  // namespace mythos
