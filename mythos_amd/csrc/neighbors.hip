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
// The same idea for the close segment: with the base vectors a1 known, a pair belongs to it only if one of its
// base-base (H-bond, cross-stacking, excluded volume), stack-stack (coaxial) or backbone-base site distances is
// within range + skin - ~10 entries in a duplex instead of the ~17 inside the orientation-independent centre range.
template <typename R>
struct SiteCrit {
  const R* off;  // backbone offsets, real4 per nucleotide; null: centre criteria only
  const R* a1;   // base vectors, real4 per nucleotide
  R rbb2;        // (backbone-backbone range + skin)^2
  R rbase2, rstack2, rkb2;  // (range + skin)^2 of the base-base, stack-stack and backbone-base site pairs
  R g_ba, g_st;  // site positions along a1
};

// (the _v forms take the four vectors; the index forms fetch them)
template <typename R, class A>
__device__ __forceinline__ bool far_in_range_v(const SiteCrit<R>& sc, const A& oi, const A& oj, const V3<R>& d) {
  const V3<R> e{d.x + oj[0] - oi[0], d.y + oj[1] - oi[1], d.z + oj[2] - oi[2]};
  return dot(e, e) < sc.rbb2;
}
template <typename R>
__device__ __forceinline__ bool far_in_range(const SiteCrit<R>& sc, int i, int j, const V3<R>& d) {
  if (!sc.off) return true;
  return far_in_range_v(sc, sc.off + 4 * i, sc.off + 4 * j, d);
}

// d = centre_j - centre_i (minimum image), already known to be inside the centre range of the close segment
template <typename R, class A>
__device__ __forceinline__ bool site_close_v(const SiteCrit<R>& sc, const A& ai, const A& oi, const A& aj, const A& oj, const V3<R>& d) {
  const V3<R> da{aj[0] - ai[0], aj[1] - ai[1], aj[2] - ai[2]};
  V3<R> e = d;
  axpy(e, sc.g_ba, da);
  if (dot(e, e) < sc.rbase2) return true;
  e = d;
  axpy(e, sc.g_st, da);
  if (dot(e, e) < sc.rstack2) return true;
  e = V3<R>{d.x + sc.g_ba * aj[0] - oi[0], d.y + sc.g_ba * aj[1] - oi[1], d.z + sc.g_ba * aj[2] - oi[2]};  // base_j - back_i
  if (dot(e, e) < sc.rkb2) return true;
  e = V3<R>{d.x + oj[0] - sc.g_ba * ai[0], d.y + oj[1] - sc.g_ba * ai[1], d.z + oj[2] - sc.g_ba * ai[2]};  // back_j - base_i
  return dot(e, e) < sc.rkb2;
}
template <typename R>
__device__ __forceinline__ bool site_close(const SiteCrit<R>& sc, int i, int j, const V3<R>& d) {
  if (!sc.off) return true;
  return site_close_v(sc, sc.a1 + 4 * i, sc.off + 4 * i, sc.a1 + 4 * j, sc.off + 4 * j, d);
}

template <typename R, bool VEC4>
__global__ __launch_bounds__(256) void build_rows_allpairs_kernel(int n, const R* __restrict__ pos,
                                                                   const BoxT<R> box, R rc2, R rcl2,
                                                                   const SiteCrit<R> sc,
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
        if (r2 < rc2) {
          const bool cl = r2 < rcl2 && site_close(sc, i, j, d);
          hit = (pass == 0) ? cl : (!cl && far_in_range(sc, i, j, d));
        }
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
// cnt row entries in LDS (buf, any order) -> dst[0 .. min(cnt, room)) ordered by neighbour index, by the G lanes of
// one group.  Rank sort: an entry goes to the slot numbered by how many entries have a smaller index (indices are
// distinct); the comparisons read the segment as LDS broadcasts, ~cnt cycles for the ~10 entries of a segment.
template <int G>
__device__ __forceinline__ void group_sort_to_row(const int* buf, int cnt, int* __restrict__ dst, int room, int l) {
  for (int b = 0; b < cnt; b += G) {
    const int k = b + l;
    const int e = (k < cnt) ? buf[k] : 0;
    const int key = e & ROW_INDEX_MASK;
    int rank = 0;
    for (int q = 0; q < cnt; ++q) rank += ((buf[q] & ROW_INDEX_MASK) < key) ? 1 : 0;
    if (k < cnt && rank < room) dst[rank] = e;
  }
}

// ------------------------------------------------------------------------------------------------
// Cell-list builder: G lanes per nucleotide (kRowG = 16: four nucleotides per wavefront).  A row is the end of a
// chain of dependent reads (own position -> 27 cell counters -> bucket entries: position, index and - from the site
// streams of the cell table, cell_list.h - backbone offset and base vector of the candidate, requested together) and a
// wavefront spends its life waiting on them; a 12 kbp duplex has ~57 candidates per nucleotide, so a
// full wavefront per nucleotide filled the machine three times over with waves that each wait the whole chain.
// Four nucleotides per wavefront fit all of them on the chip at once.  The group's lanes look up the 27 neighbour
// cells and the spill list, the candidate lists are concatenated by a prefix sum and swept G at a time with ballot
// compaction into LDS; close and far segments are then ordered by neighbour index on their way to the row (the
// buckets are in the order the binning atomics landed), so rows are reproducible run to run.
// ------------------------------------------------------------------------------------------------
#ifndef MYTHOS_ROW_G
#define MYTHOS_ROW_G 16
#endif
constexpr int kRowG = MYTHOS_ROW_G;

template <typename R, bool VEC4, int G>
__global__ __launch_bounds__(256) void build_rows_cells_kernel(int n, const R* __restrict__ pos, const BoxT<R> box,
                                                                const CellGrid<R> g, R rc2, R rcl2,
                                                                const SiteCrit<R> sc,
                                                                const int* __restrict__ partners,
                                                                const int* __restrict__ cell_cnt,
                                                                const typename CellPlace<R>::type* __restrict__ place,
                                                                int cell_cap, const int* __restrict__ spill, int cell_H,
                                                                size_t site_stride,  // H * cap when the table carries the site streams, else 0
                                                                int* __restrict__ rows,
                                                                int* __restrict__ row_len, int* __restrict__ row_close,
                                                                int row_stride, int* __restrict__ overflow,
                                                                R* __restrict__ ref_pos, R* __restrict__ ref_off,
                                                                R* __restrict__ ref_a1) {
  constexpr int NG = 256 / G;  // nucleotides per workgroup
  __shared__ int s_pre[NG][29], s_st[NG][28], s_c[NG][28][3];
  extern __shared__ int s_seg_all[];  // [NG][2][row_stride]: the close and the far entries of a row, in candidate order
  const int kSegCap = row_stride;
  const int grp = threadIdx.x / G, l = threadIdx.x % G;
  const int gshift = (threadIdx.x & 63) & ~(G - 1);  // first lane of this group inside its wavefront
  const int i = blockIdx.x * NG + grp;
  if (i >= n) return;
  int* s_close = s_seg_all + (size_t)grp * 2 * row_stride;
  int* s_far = s_close + row_stride;
  constexpr int S = VEC4 ? 4 : 3;
  const V3<R> ci{pos[S * i], pos[S * i + 1], pos[S * i + 2]};
  using PL = typename CellPlace<R>::type;
  R oi[3] = {R(0), R(0), R(0)}, ai[3] = {R(0), R(0), R(0)};  // own backbone offset and base vector
  if (sc.off) {
#pragma unroll
    for (int a = 0; a < 3; ++a) oi[a] = sc.off[4 * (size_t)i + a], ai[a] = sc.a1[4 * (size_t)i + a];
  }
  int cx, cy, cz;
  cell_of(g, ci.x, ci.y, ci.z, cx, cy, cz);
  // the 27 cells' counters and the spill list's, G at a time, with their running sum (an inclusive scan over the group's
  // lanes through the cross-lane network; one lane adding up 28 LDS words was 1.5 us of every row's chain)
  int carry = 0;
  if (l == 0) s_pre[grp][0] = 0;
  for (int k0 = 0; k0 < 28; k0 += G) {
    const int k = k0 + l;
    int cnt = 0;
    if (k < 27) {
      int c[3] = {cx + k % 3 - 1, cy + (k / 3) % 3 - 1, cz + k / 9 - 1};
#pragma unroll
      for (int a = 0; a < 3; ++a)
        if (g.nc[a] > 0) c[a] = (c[a] + g.nc[a]) % g.nc[a];
      const int h = cell_slot(g, c[0], c[1], c[2]);
      cnt = min(cell_cnt[h], cell_cap);
      s_st[grp][k] = h * cell_cap;
      s_c[grp][k][0] = c[0], s_c[grp][k][1] = c[1], s_c[grp][k][2] = c[2];
    } else if (k == 27) {  // the spill list: particles whose bucket was full, candidates for every row
      cnt = min(cell_cnt[cell_H], kCellSpill);
    }
    int run = cnt;
#pragma unroll
    for (int d = 1; d < G; d <<= 1) {
      const int o = __shfl_up(run, d, G);
      if (l >= d) run += o;
    }
    if (k < 28) s_pre[grp][k + 1] = carry + run;
    carry += __shfl(run, G - 1, G);
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  const int total = s_pre[grp][28];
  int* row = rows + (size_t)i * row_stride;
  const int* bq = partners + (size_t)ROW_BONDED_SLOTS * i;
  const int4 bp = make_int4(bq[0], bq[1], bq[2], bq[3]);
  int out_c = 0, out_f = 0;
  int lo = 0;  // cell of this lane's candidate: t grows by G per sweep, so it only ever advances
  constexpr unsigned int kGroupMask = (G == 32) ? 0xffffffffu : ((1u << G) - 1u);
  for (int t0 = 0; t0 < total; t0 += G) {
    const int t = t0 + l;
    bool hit_c = false, hit_f = false;
    int j = -1;
    if (t < total) {
      while (s_pre[grp][lo + 1] <= t) ++lo;
      R xj, yj, zj;
      R oj[3] = {R(0), R(0), R(0)}, aj[3] = {R(0), R(0), R(0)};
      if (lo < 27) {  // everything about the candidate arrives together, as contiguous streams per cell
        const size_t at = (size_t)s_st[grp][lo] + (t - s_pre[grp][lo]);
        const PL pl = place[at];
        if (site_stride) {
          const PL po = place[site_stride + at], pa = place[2 * site_stride + at];
          oj[0] = po.x, oj[1] = po.y, oj[2] = po.z, aj[0] = pa.x, aj[1] = pa.y, aj[2] = pa.z;
        }
        xj = pl.x, yj = pl.y, zj = pl.z;
        j = cell_index_of(pl.w);
      } else {
        j = spill[t - s_pre[grp][27]];
        xj = pos[S * j], yj = pos[S * j + 1], zj = pos[S * j + 2];
        if (site_stride) {
#pragma unroll
          for (int a = 0; a < 3; ++a) oj[a] = sc.off[4 * (size_t)j + a], aj[a] = sc.a1[4 * (size_t)j + a];
        }
      }
      if (j != i && j != bp.x && j != bp.y && j != bp.z && j != bp.w) {
        bool mine = true;  // hashed table: a bucket may mix cells, a candidate counts for the cell it lies in
        if (!g.direct && lo < 27) {
          int jx, jy, jz;
          cell_of(g, xj, yj, zj, jx, jy, jz);
          mine = jx == s_c[grp][lo][0] && jy == s_c[grp][lo][1] && jz == s_c[grp][lo][2];
        }
        if (mine) {
          V3<R> d{xj - ci.x, yj - ci.y, zj - ci.z};
          d = min_image(d, box);
          const R r2 = dot(d, d);
          if (r2 < rc2) {
            if (site_stride) {
              hit_c = r2 < rcl2 && site_close_v(sc, ai, oi, aj, oj, d);
              hit_f = !hit_c && far_in_range_v(sc, oi, oj, d);
            } else {
              hit_c = r2 < rcl2 && site_close(sc, i, j, d);
              hit_f = !hit_c && far_in_range(sc, i, j, d);
            }
          }
        }
      }
    }
    const int e = (j < i) ? (j | ROW_ROLE_Q) : j;
    const unsigned int mc = (unsigned int)(__ballot(hit_c) >> gshift) & kGroupMask;
    const unsigned int mf = (unsigned int)(__ballot(hit_f) >> gshift) & kGroupMask;
    const unsigned int below = (1u << l) - 1u;
    if (hit_c) {
      const int k = out_c + __popc(mc & below);
      if (k < kSegCap) s_close[k] = e;
    }
    if (hit_f) {
      const int k = out_f + __popc(mf & below);
      if (k < kSegCap) s_far[k] = e;
    }
    out_c += __popc(mc);
    out_f += __popc(mf);
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  const int n_close = ROW_BONDED_SLOTS + out_c;
  int out = n_close + out_f;
  group_sort_to_row<G>(s_close, min(out_c, kSegCap), row + ROW_BONDED_SLOTS, row_stride - ROW_BONDED_SLOTS, l);
  if (n_close < row_stride) group_sort_to_row<G>(s_far, min(out_f, kSegCap), row + n_close, row_stride - n_close, l);
  if (l == 0) {
    row[0] = bp.x, row[1] = bp.y, row[2] = bp.z, row[3] = bp.w;
    if (out > row_stride) {
      atomicMax(overflow, out);
      out = row_stride;
    }
    row_len[i] = out;
    row_close[i] = min(n_close, out);
    if (ref_pos) {  // what the MD displacement check compares against: the state this list was built from
      R* rp = ref_pos + 4 * (size_t)i;
      rp[0] = ci.x, rp[1] = ci.y, rp[2] = ci.z, rp[3] = VEC4 ? pos[S * i + 3] : R(0);
      if (sc.off) {
        R* ro = ref_off + 4 * (size_t)i;
        R* ra = ref_a1 + 4 * (size_t)i;
#pragma unroll
        for (int k = 0; k < 4; ++k) ro[k] = sc.off[4 * (size_t)i + k], ra[k] = sc.a1[4 * (size_t)i + k];
      }
    }
  }
}

template <typename R>
static int build_cells_typed(mythos_system* sys, const R* pos, bool vec4, double rl, double skin, const R* off,
                             const R* a1, bool write_refs, hipStream_t st) {
  R* ref_pos = write_refs ? (R*)sys->d_ref_pos : nullptr;
  R* ref_off = write_refs ? (R*)sys->d_ref_off : nullptr;
  R* ref_a1 = write_refs ? (R*)sys->d_ref_a1 : nullptr;
  const int n = sys->n;
  // classification radius of the leading "close" segment (everything is close until parameters exist)
  const double rcl = sys->params_set ? std::min(rl, oxdna_close_range(sys) + skin) : rl;
  int* d_close = row_close_of(sys);
  // range of the backbone-backbone terms (excluded volume, and Debye-Hueckel in oxDNA2) for the far segment
  double rbb = rl;
  if (off && sys->params_set) {
    rbb = oxdna_param_max(sys, NEXC_BACKBONE_RC);
    if (sys->model >= 2) rbb = std::max(rbb, oxdna_param_max(sys, DH_RCUT));
    rbb = std::min(rl, rbb + skin);
  }
  if (!sys->params_set || !a1) off = nullptr;
  SiteCrit<R> sc{off, a1, R(rbb * rbb), R(0), R(0), R(0), R(0), R(0)};
  if (off) {
    // oxNA: the largest range over the three vectors; the base / stack sites of the two geometries sit at the mean
    // offset along a1, and the ranges grow by the distance either real site can be from there
    auto mx = [&](int idx) { return oxdna_param_max(sys, idx); };
    const double* Pd = oxdna_param_set(sys, 0);
    const double* Pr = oxdna_param_set(sys, sys->param_sets() == 1 ? 0 : 1);
    const double slack_ba = std::fabs(Pd[GEO_BASE] - Pr[GEO_BASE]), slack_st = std::fabs(Pd[GEO_STACK] - Pr[GEO_STACK]);
    auto sq = [&](double r) { return R((r + skin) * (r + skin)); };
    sc.rbase2 = sq(std::max({mx(HYDR_RCHIGH), mx(CRST_RCHIGH), mx(NEXC_BASE_RC)}) + slack_ba);
    sc.rstack2 = sq(mx(CXST_RCHIGH) + slack_st);
    sc.rkb2 = sq(std::max(mx(NEXC_BACK_BASE_RC), mx(NEXC_BASE_BACK_RC)) + 0.5 * slack_ba);
    sc.g_ba = R(0.5 * (Pd[GEO_BASE] + Pr[GEO_BASE]));
    sc.g_st = R(0.5 * (Pd[GEO_STACK] + Pr[GEO_STACK]));
  }
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
                         R(rl * rl), R(rcl * rcl), sc, d_partners, sys->d_rows, sys->d_row_len, d_close, sys->row_stride, sys->d_overflow);
    else
      hipLaunchKernelGGL((build_rows_allpairs_kernel<R, false>), dim3(blocks_ap), dim3(256), 0, st, n, pos, box,
                         R(rl * rl), R(rcl * rcl), sc, d_partners, sys->d_rows, sys->d_row_len, d_close, sys->row_stride, sys->d_overflow);
    if (write_refs && vec4) {  // the cell-list kernel writes these itself
      const size_t bytes = (size_t)n * 4 * sizeof(R);
      MYTHOS_HIP_TRY(hipMemcpyAsync(ref_pos, pos, bytes, hipMemcpyDeviceToDevice, st));
      if (sc.off) {
        MYTHOS_HIP_TRY(hipMemcpyAsync(ref_off, sc.off, bytes, hipMemcpyDeviceToDevice, st));
        MYTHOS_HIP_TRY(hipMemcpyAsync(ref_a1, sc.a1, bytes, hipMemcpyDeviceToDevice, st));
      }
    }
    return 0;
  }
  const int H = next_pow2(2 * n);
  if (cell_cap_override()) sys->cell_bucket_cap = cell_cap_override();
  const int cap = sys->cell_bucket_cap;
  const bool sites = sc.off != nullptr && vec4;  // MD frames: offsets and base vectors ride in the cell table
  const size_t need = CellBins::ints(H, cap, sizeof(R), sites);
  if (need > sys->cell_cap || H != sys->cell_H || cap != sys->cell_alloc_bucket_cap || sites != sys->cell_sites) {
    if (sys->d_cell) (void)hipFree(sys->d_cell);
    sys->d_cell = nullptr;
    sys->cell_cap = 0;
    MYTHOS_HIP_TRY(hipMalloc((void**)&sys->d_cell, need * sizeof(int)));
    MYTHOS_HIP_TRY(hipMemsetAsync(sys->d_cell + CellBins::zero_offset(H, cap, sizeof(R), sites), 0, CellBins::zero_ints(H) * sizeof(int), st));
    sys->cell_cap = need;
    sys->cell_sites = sites;
    sys->cell_H = H;
    sys->cell_alloc_bucket_cap = cap;
    sys->cell_phase = 0;
  }
  const CellBins bins(sys->d_cell, H, cap, sizeof(R), sys->cell_phase, sites);
  sys->cell_phase ^= 1;
  const size_t site_stride = sites ? (size_t)H * cap : 0;
  // the row builder orders its rows itself: no bucket sort
  if (vec4)
    cell_bins_build<R, true>(n, pos, g, bins, sys->d_overflow, false, st, sc.off, sc.a1);
  else
    cell_bins_build<R, false>(n, pos, g, bins, sys->d_overflow, false, st);
  constexpr int kPerBlock = 256 / kRowG;
  const int wb = (n + kPerBlock - 1) / kPerBlock;
  const size_t far_lds = (size_t)kPerBlock * 2 * sys->row_stride * sizeof(int);
  if (vec4)
    hipLaunchKernelGGL((build_rows_cells_kernel<R, true, kRowG>), dim3(wb), dim3(256), far_lds, st, n, pos, box, g, R(rl * rl),
                       R(rcl * rcl), sc, d_partners, bins.cnt_cur, (const typename CellPlace<R>::type*)bins.place, bins.cap, bins.spill, bins.H, site_stride, sys->d_rows, sys->d_row_len, d_close, sys->row_stride,
                       sys->d_overflow, ref_pos, ref_off, ref_a1);
  else
    hipLaunchKernelGGL((build_rows_cells_kernel<R, false, kRowG>), dim3(wb), dim3(256), far_lds, st, n, pos, box, g, R(rl * rl),
                       R(rcl * rcl), sc, d_partners, bins.cnt_cur, (const typename CellPlace<R>::type*)bins.place, bins.cap, bins.spill, bins.H, site_stride, sys->d_rows, sys->d_row_len, d_close, sys->row_stride,
                       sys->d_overflow, ref_pos, ref_off, ref_a1);
  return 0;
}

int rows_build_device(mythos_system* sys, const void* center, bool center_is_vec4, double r_cut, double skin,
                      const void* backbone_offsets, const void* base_vectors, bool write_refs, hipStream_t stream) {
  if (sys->row_stride == 0)
    if (int rc = rows_reserve(sys, 64)) return rc;
  const double rl = r_cut + skin;
  int rc;
  if (sys->dtype == MYTHOS_F32)
    rc = build_cells_typed<float>(sys, (const float*)center, center_is_vec4, rl, skin, (const float*)backbone_offsets, (const float*)base_vectors, write_refs, stream);
  else
    rc = build_cells_typed<double>(sys, (const double*)center, center_is_vec4, rl, skin, (const double*)backbone_offsets, (const double*)base_vectors, write_refs, stream);
  if (rc) return rc;
  MYTHOS_HIP_TRY(hipGetLastError());
  sys->nbrs_set = true;
  return 0;
}

// Builds the rows, growing the row stride and the bucket capacity until the build fits (synchronises the stream).
// Buckets end up at most half full (fuller ones work, through the spill list, but slowly).  headroom: leave a
// quarter of spare row length for builds that follow without a chance to grow (inside an MD run, where a row
// overflow ends the run with an error).
int rows_build_until_fit(mythos_system* sys, const void* center, bool center_is_vec4, double r_cut, double skin,
                         const void* backbone_offsets, const void* base_vectors, bool write_refs, bool headroom,
                         hipStream_t stream) {
  for (int attempt = 0; attempt < 6; ++attempt) {
    MYTHOS_HIP_TRY(hipMemsetAsync(sys->d_overflow, 0, kOverflowWords * sizeof(int), stream));
    if (int rc = rows_build_device(sys, center, center_is_vec4, r_cut, skin, backbone_offsets, base_vectors, write_refs, stream))
      return rc;
    int ov[kOverflowWords] = {0, 0, 0};
    MYTHOS_HIP_TRY(hipMemcpyAsync(ov, sys->d_overflow, sizeof(ov), hipMemcpyDeviceToHost, stream));
    MYTHOS_HIP_TRY(hipStreamSynchronize(stream));
    if (ov[1] > 0) {
      set_error("neighbour build: more than " + std::to_string(kCellSpill) + " particles did not fit the buckets of their cells");
      return MYTHOS_ERR_OVERFLOW;
    }
    const int bucket_demand = cell_cap_override() ? 0 : ov[2];  // a bucket more than half full: double the places
    if (ov[0] == 0 && bucket_demand == 0) {
      if (ov[2] > 0) MYTHOS_HIP_TRY(hipMemsetAsync(sys->d_overflow + 2, 0, sizeof(int), stream));
      return MYTHOS_OK;
    }
    if (ov[0] > 0) {
      const int want = headroom ? ((ov[0] + ov[0] / 4 + 15) / 16) * 16 : ((ov[0] + 15) / 16) * 16 + 16;
      if (int rc = rows_reserve(sys, want)) return rc;
    }
    if (bucket_demand > 0) sys->cell_bucket_cap = ((2 * bucket_demand + 15) / 16) * 16;  // reallocated by the next build
  }
  set_error("neighbour build: rows or cell buckets keep overflowing");
  return MYTHOS_ERR_OVERFLOW;
}

}  // namespace mythos
