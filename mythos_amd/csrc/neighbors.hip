// Neighbour rows: conversion from a reference-style pair list and GPU Verlet-list builds.
//
// Replaces the reference's O(N^2) pair arrays (mythos/input/topology.py:186-190) and its jax_md
// neighbour-list factory (mythos/utils/neighbors.py:12-59, simulators/jax_md/utils.py:70-126)
// with per-nucleotide rows (see mythos_internal.h for the slot encoding).  Rows are written in
// ascending neighbour order so every downstream sum is reproducible.
#include <algorithm>
#include <cmath>

#include "cell_list.h"
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
    for (int k = 0; k < ROW_BONDED_SLOTS; ++k) rows[(size_t)i * stride + k] = sys->h_partners[(size_t)ROW_BONDED_SLOTS * i + k];
  }
  for (int k = 0; k < n_pairs; ++k) {
    const int i = pairs[2 * k], j = pairs[2 * k + 1];
    rows[(size_t)i * stride + len[i]++] = j;               // i plays op_i
    rows[(size_t)j * stride + len[j]++] = i | ROW_ROLE_Q;  // j plays op_j
  }
  if (int rc = rows_reserve(sys, stride)) return rc;
  MYTHOS_HIP_TRY(hipMemcpy(sys->d_rows, rows.data(), rows.size() * sizeof(int), hipMemcpyHostToDevice));
  MYTHOS_HIP_TRY(hipMemcpy(sys->d_row_len, len.data(), n * sizeof(int), hipMemcpyHostToDevice));
  // a user-supplied pair list carries no distance classes: every entry is in the "close" segment
  MYTHOS_HIP_TRY(hipMemcpy(row_close_of(sys), len.data(), n * sizeof(int), hipMemcpyHostToDevice));
  sys->nbrs_set = true;
  return 0;
}

// ------------------------------------------------------------------------------------------------
// all-pairs Verlet build: one wavefront per nucleotide sweeps every other nucleotide in strides
// of 64 (coalesced position reads served from L2) and appends hits by ballot compaction, which
// keeps each row in ascending index order.  O(N^2) work: the exact reference for the cell build.
// ------------------------------------------------------------------------------------------------
// Far-segment refinement: beyond the close range only the backbone-backbone terms act, so when the backbone
// offsets are known (MD frames) a pair is listed only if its BACKBONE sites are within their range + skin - in a
// duplex that keeps ~4 of the ~18 pairs a centre-distance criterion would list.  off = real4 per nucleotide
// (backbone offset k1 a1 + k2 a2) or null (centre criterion only).
template <typename R>
__device__ __forceinline__ bool far_in_range(const R* __restrict__ off, int i, int j, const V3<R>& d, R rbb2) {
  if (!off) return true;
  const V3<R> e{d.x + off[4 * j] - off[4 * i], d.y + off[4 * j + 1] - off[4 * i + 1], d.z + off[4 * j + 2] - off[4 * i + 2]};
  return dot(e, e) < rbb2;
}

template <typename R, bool VEC4>
__global__ __launch_bounds__(256) void build_rows_allpairs_kernel(int n, const R* __restrict__ pos,
                                                                   const BoxT<R> box, R rc2, R rcl2,
                                                                   const R* __restrict__ off, R rbb2,
                                                                   const int* __restrict__ partners_rows_in,
                                                                   int* __restrict__ rows, int* __restrict__ row_len,
                                                                   int* __restrict__ row_close, int row_stride,
                                                                   int* __restrict__ overflow) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (wave >= n) return;
  const int i = wave;
  constexpr int S = VEC4 ? 4 : 3;
  const V3<R> ci{pos[S * i], pos[S * i + 1], pos[S * i + 2]};
  int* row = rows + (size_t)i * row_stride;
  const int* bq = partners_rows_in + (size_t)ROW_BONDED_SLOTS * i;
  const int4 bp = make_int4(bq[0], bq[1], bq[2], bq[3]);
  int cnt = ROW_BONDED_SLOTS, n_close = 0;
  for (int pass = 0; pass < 2; ++pass) {  // pass 0: "close" segment (r2 < rcl2), pass 1: the rest of the list
    for (int j0 = 0; j0 < n; j0 += 64) {
      const int j = j0 + lane;
      bool hit = false;
      if (j < n && j != i && j != bp.x && j != bp.y && j != bp.z && j != bp.w) {
        V3<R> d{pos[S * j] - ci.x, pos[S * j + 1] - ci.y, pos[S * j + 2] - ci.z};
        d = min_image(d, box);
        const R r2 = dot(d, d);
        hit = (pass == 0) ? (r2 < rcl2) : (r2 >= rcl2 && r2 < rc2 && far_in_range<R>(off, i, j, d, rbb2));
      }
      const unsigned long long m = __ballot(hit);
      if (hit) {
        const int slot = cnt + __popcll(m & ((1ull << lane) - 1ull));
        if (slot < row_stride) row[slot] = (j < i) ? (j | ROW_ROLE_Q) : j;
      }
      cnt += __popcll(m);
    }
    if (pass == 0) n_close = cnt;
  }
  if (lane == 0) {
    row[0] = bp.x, row[1] = bp.y, row[2] = bp.z, row[3] = bp.w;
    if (cnt > row_stride) {
      atomicMax(overflow, cnt);
      cnt = row_stride;
    }
    row_len[i] = cnt;
    row_close[i] = min(n_close, cnt);
  }
}

// ------------------------------------------------------------------------------------------------
// Hashed cell list (O(N)): cells of edge >= r_list are hashed into a power-of-two table, so the
// grid never has to cover a bounding box (a 12 kbp duplex is ~5000 length units long and one
// cell thin).  Buckets may mix cells that collide in the hash; a candidate is accepted only for
// the neighbour cell it really lies in, which also removes duplicates.  Buckets are sorted by
// index after the atomic fill and rows are filled in (cell order, bucket order), so the list is
// reproducible run to run.
// ------------------------------------------------------------------------------------------------
// one wavefront per nucleotide: lanes 0..26 look up the 27 neighbour cells, the candidate lists
// are concatenated by a wave prefix sum and swept 64 at a time with ballot compaction
template <typename R, bool VEC4>
__global__ __launch_bounds__(256) void build_rows_cells_kernel(int n, const R* __restrict__ pos, const BoxT<R> box,
                                                                const CellGrid<R> g, R rc2, R rcl2,
                                                                const R* __restrict__ off, R rbb2,
                                                                const int* __restrict__ partners,
                                                                const int* __restrict__ start,
                                                                const int* __restrict__ bucket, int* __restrict__ rows,
                                                                int* __restrict__ row_len, int* __restrict__ row_close,
                                                                int row_stride, int* __restrict__ overflow) {
  __shared__ int s_pre[4][28], s_st[4][27], s_c[4][27][3];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + w;
  if (i >= n) return;
  constexpr int S = VEC4 ? 4 : 3;
  const V3<R> ci{pos[S * i], pos[S * i + 1], pos[S * i + 2]};
  int cx, cy, cz;
  cell_of(g, ci.x, ci.y, ci.z, cx, cy, cz);
  int cnt = 0;
  if (lane < 27) {
    int c[3] = {cx + lane % 3 - 1, cy + (lane / 3) % 3 - 1, cz + lane / 9 - 1};
#pragma unroll
    for (int k = 0; k < 3; ++k)
      if (g.nc[k] > 0) c[k] = (c[k] + g.nc[k]) % g.nc[k];
    const int h = cell_slot(g, c[0], c[1], c[2]);
    const int st = start[h];
    cnt = start[h + 1] - st;
    s_st[w][lane] = st;
    s_c[w][lane][0] = c[0], s_c[w][lane][1] = c[1], s_c[w][lane][2] = c[2];
  }
  int inc = cnt;  // inclusive wave prefix sum
#pragma unroll
  for (int o = 1; o < 32; o <<= 1) {
    const int v = __shfl_up(inc, o, 64);
    if (lane >= o) inc += v;
  }
  if (lane < 27) s_pre[w][lane + 1] = inc;
  if (lane == 0) s_pre[w][0] = 0;
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  const int total = __shfl(inc, 26, 64);
  int* row = rows + (size_t)i * row_stride;
  const int* bq = partners + (size_t)ROW_BONDED_SLOTS * i;
  const int4 bp = make_int4(bq[0], bq[1], bq[2], bq[3]);
  int out = ROW_BONDED_SLOTS, n_close = 0;
  for (int pass = 0; pass < 2; ++pass) {  // pass 0: "close" segment, pass 1: the rest
  for (int t0 = 0; t0 < total; t0 += 64) {
    const int t = t0 + lane;
    bool hit = false;
    int j = -1;
    if (t < total) {
      int lo = 0, hi = 27;  // largest cell index with s_pre[cell] <= t
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (s_pre[w][mid] <= t) lo = mid; else hi = mid;
      }
      j = bucket[s_st[w][lo] + (t - s_pre[w][lo])];
      if (j != i && j != bp.x && j != bp.y && j != bp.z && j != bp.w) {
        const R xj = pos[S * j], yj = pos[S * j + 1], zj = pos[S * j + 2];
        int jx, jy, jz;
        cell_of(g, xj, yj, zj, jx, jy, jz);
        if (jx == s_c[w][lo][0] && jy == s_c[w][lo][1] && jz == s_c[w][lo][2]) {
          V3<R> d{xj - ci.x, yj - ci.y, zj - ci.z};
          d = min_image(d, box);
          const R r2 = dot(d, d);
          hit = (pass == 0) ? (r2 < rcl2) : (r2 >= rcl2 && r2 < rc2 && far_in_range<R>(off, i, j, d, rbb2));
        }
      }
    }
    const unsigned long long m = __ballot(hit);
    if (hit) {
      const int slot = out + __popcll(m & ((1ull << lane) - 1ull));
      if (slot < row_stride) row[slot] = (j < i) ? (j | ROW_ROLE_Q) : j;
    }
    out += __popcll(m);
  }
  if (pass == 0) n_close = out;
  }
  if (lane == 0) {
    row[0] = bp.x, row[1] = bp.y, row[2] = bp.z, row[3] = bp.w;
    if (out > row_stride) {
      atomicMax(overflow, out);
      out = row_stride;
    }
    row_len[i] = out;
    row_close[i] = min(n_close, out);
  }
}

template <typename R>
static int build_cells_typed(mythos_system* sys, const R* pos, bool vec4, double rl, double skin, const R* off,
                             hipStream_t st) {
  const int n = sys->n;
  // classification radius of the leading "close" segment (everything is close until parameters exist)
  const double rcl = sys->params_set ? std::min(rl, oxdna_close_range(sys) + skin) : rl;
  int* d_close = row_close_of(sys);
  // range of the backbone-backbone terms (excluded volume, and Debye-Hueckel in oxDNA2) for the far segment
  double rbb = rl;
  if (off && sys->params_set) {
    rbb = sys->pd[NEXC_BACKBONE_RC];
    if (sys->model == 2) rbb = std::max(rbb, (double)sys->pd[DH_RCUT]);
    rbb = std::min(rl, rbb + skin);
  }
  if (!sys->params_set) off = nullptr;
  const R rbb2 = R(rbb * rbb);
  CellGrid<R> g;
  bool ok = true;
  for (int k = 0; k < 3; ++k) {
    if (sys->has_box) {
      const int nc = (int)std::floor(sys->box[k] / rl);
      if (nc < 3) ok = false;
      g.nc[k] = std::max(nc, 1);
      g.ibox[k] = R(1.0 / sys->box[k]);
      g.inv[k] = R(g.nc[k] / sys->box[k]);
    } else {
      g.nc[k] = 0;
      g.ibox[k] = R(1);
      g.inv[k] = R(1.0 / rl);
    }
  }
  const int blocks_ap = (n * 64 + 255) / 256;
  int* d_partners = sys->d_row_len + n;
  const BoxT<R> box = make_box<R>(sys);
  if (!ok || n < 512) {  // tiny systems / boxes under three cells: the all-pairs sweep is exact and cheap
    if (vec4)
      hipLaunchKernelGGL((build_rows_allpairs_kernel<R, true>), dim3(blocks_ap), dim3(256), 0, st, n, pos, box,
                         R(rl * rl), R(rcl * rcl), off, rbb2, d_partners, sys->d_rows, sys->d_row_len, d_close, sys->row_stride, sys->d_overflow);
    else
      hipLaunchKernelGGL((build_rows_allpairs_kernel<R, false>), dim3(blocks_ap), dim3(256), 0, st, n, pos, box,
                         R(rl * rl), R(rcl * rcl), off, rbb2, d_partners, sys->d_rows, sys->d_row_len, d_close, sys->row_stride, sys->d_overflow);
    return 0;
  }
  const int H = next_pow2(2 * n);
  const size_t need = CellScratch::ints(H, n);
  if (need > sys->cell_cap) {
    if (sys->d_cell) (void)hipFree(sys->d_cell);
    sys->d_cell = nullptr;
    sys->cell_cap = 0;
    MYTHOS_HIP_TRY(hipMalloc((void**)&sys->d_cell, need * sizeof(int)));
    sys->cell_cap = need;
  }
  const CellScratch cs(sys->d_cell, H, n);
  const int* start = cs.start;
  const int* bucket = cs.bucket;
  if ((vec4 ? cell_list_build<R, true>(n, pos, g, H, cs, st) : cell_list_build<R, false>(n, pos, g, H, cs, st)) != 0) {
    set_error("neighbour build: cell-list scratch memset failed");
    return MYTHOS_ERR_HIP;
  }
  const int wb = (n + 3) / 4;
  if (vec4)
    hipLaunchKernelGGL((build_rows_cells_kernel<R, true>), dim3(wb), dim3(256), 0, st, n, pos, box, g, R(rl * rl),
                       R(rcl * rcl), off, rbb2, d_partners, start, bucket, sys->d_rows, sys->d_row_len, d_close, sys->row_stride,
                       sys->d_overflow);
  else
    hipLaunchKernelGGL((build_rows_cells_kernel<R, false>), dim3(wb), dim3(256), 0, st, n, pos, box, g, R(rl * rl),
                       R(rcl * rcl), off, rbb2, d_partners, start, bucket, sys->d_rows, sys->d_row_len, d_close, sys->row_stride,
                       sys->d_overflow);
  return 0;
}

int rows_build_device(mythos_system* sys, const void* center, bool center_is_vec4, double r_cut, double skin,
                      const void* backbone_offsets, hipStream_t stream) {
  if (sys->row_stride == 0)
    if (int rc = rows_reserve(sys, 64)) return rc;
  const double rl = r_cut + skin;
  MYTHOS_HIP_TRY(hipMemsetAsync(sys->d_overflow, 0, sizeof(int), stream));
  int rc;
  if (sys->dtype == MYTHOS_F32)
    rc = build_cells_typed<float>(sys, (const float*)center, center_is_vec4, rl, skin, (const float*)backbone_offsets, stream);
  else
    rc = build_cells_typed<double>(sys, (const double*)center, center_is_vec4, rl, skin, (const double*)backbone_offsets, stream);
  if (rc) return rc;
  MYTHOS_HIP_TRY(hipGetLastError());
  sys->nbrs_set = true;
  return 0;
}

}  // namespace mythos
