// kernels_fuse.hip — K5/K6: depth-point propagation, per-pixel probabilistic fusion, clean and
// regularisation on a dense DepthMap (gfx950).
//
// Replaces DepthFusion::update / propagate_one_point / fusion
// (esvo_core/src/core/DepthFusion.cpp:18-192), DepthPoint::update_studentT
// (esvo_core/src/container/DepthPoint.cpp:167-188), SmartGrid::clean
// (esvo_core/include/esvo_core/container/SmartGrid.h:222-243) and
// DepthRegularization::apply (esvo_core/src/core/DepthRegularization.cpp:19-110).
//
// The reference fuses sequentially: frames newest -> oldest, points in stored order, and for
// each point its 2x2 (or 3x3) cells in (dy,dx) order; the result of a cell depends on the
// order of the observations that hit it.  On the GPU every (point, cell) pair gets the record
// id  q*K + k  (q = position of the point in fusion order) which IS that sequential order.
// Records are bucketed by cell with a counting sort (atomic histogram -> scan -> scatter), each
// cell's short list is sorted by id, and one thread walks it applying the reference's state
// machine.  Cells are independent, so the result equals the sequential one exactly.
//
// DepthMap layout: one 104-byte MapCell per pixel (dense; the reference's list + pointer grid
// is re-created every tick anyway, esvo_Mapping.cpp:268-272).  MapCell::row/col are the
// coordinates the element believes it has; they differ from the true cell only after the
// replace branch (DepthFusion.cpp:186), whose side effects on
// clean/regularisation are reproduced through the ALIVE/GRID flag pair.
#include <algorithm>
#include <type_traits>
#define DEV_HOOKS_FUSE_TU
#include "common.hpp"
#include "dev_hooks.hpp"
#include "fdiv.hpp"
#include "scan.hpp"

namespace esvo {

// Register budget of the back-stage kernels: they run beside the LM kernel of the next tick, whose two waves per SIMD
// leave 144 of 512 VGPRs.  At <= 80 VGPRs a second back-stage wave fits where only one did (propagate 114 -> 72,
// reg_apply 124 -> 78, no spills): +1.5 % per tick.  fuse_cells keeps its 112 (capping it spills; dropping its software
// prefetch gives 80 and was measured neutral).
// (Raising the issue priority of the back stage's waves inside their SIMD -- s_setprio 1..3 -- was measured in round 4: slower.)
#ifndef BACK_WAVES
#define BACK_WAVES 6
#endif
#ifndef FUSE_WAVES
#define FUSE_WAVES 1
#endif

__device__ inline bool boundaryCheck(double x, double y, int W, int H) {  // DepthFusion.cpp:194-205
  return !(x < 0 || x >= (double)W || y < 0 || y >= (double)H);
}

// cell k of a propagated point (DepthFusion.cpp:98-117); returns false when outside the image
__device__ inline bool fusion_cell(u32 prow, u32 pcol, int k, int radius, int W, int H, int& row, int& col) {
  int dy, dx;
  if (radius == 0) { dy = k >> 1; dx = k & 1; }
  else { dy = k / 3 - 1; dx = k % 3 - 1; }
  row = (int)prow + dy;
  col = (int)pcol + dx;
  return row >= 0 && row < H && col >= 0 && col < W;
}

// ---- the three observation models of DepthFusion -------------------------------------------------------------------
//   FUSE_TDIST  DepthFusion::update with LSnorm "Tdist" (every shipped configuration): Student-t propagation and fusion
//   FUSE_L2     the same with LSnorm "l2": Gaussian propagation (DepthFusion.cpp:49-53), chiSquareTest (:207-218) and
//               DepthPoint::update (DepthPoint.cpp:146-164) in the compatible branch
//   FUSE_NAIVE  DepthFusion::naive_propagation (:234-288): Gaussian propagation, always 2 x 2 cells, an occupied cell is
//               replaced by a propagated point that is not farther and has the smaller residual -- esvo_MVStereo's
//               PURE_BLOCK_MATCHING mode (esvo_MVStereo.cpp:416-428)
enum { FUSE_TDIST = 0, FUSE_L2 = 1, FUSE_NAIVE = 2 };

// ---- the fusion front: tiles (round 4) ------------------------------------------------------------------------------------
// Until round 3 every (point, cell) record went through a per-CELL counting sort in global memory: 9 device-scope atomics per
// point on a 300 k-entry histogram, a 3-pass scan of it, 9 more atomics + a scattered 4-byte store per point, a sort kernel for
// the long lists, then the walk -- 2 x 1.44 M atomics per tick, each a 32-byte memory-side write on this eight-XCD part
// (profiles/r03_v3_hbm_traffic.csv: 40 MB written for 5.8 MB of ids), in 8 launches.  Now the sort is two-level and only the
// coarse level uses global atomics, ONE per point:
//   A  propagate    one thread per window point: the propagated point (as before) and its id appended to the list of the
//                   8 x 8-cell TILE its centre cell lies in (fixed-capacity lists; a point that finds its list full goes
//                   to one shared overflow list -- dense scenes only, every tile then looks through it)
//   T  tile_lists   one wave per tile, one lane per cell: the points of the tile and of its eight neighbours (a 3 x 3
//                   footprint reaches at most one tile further) become each cell's record list, in id order, IN LDS (see the
//                   kernel); the lists leave the chip once (one reservation per tile), with each cell's (offset, count), and
//                   the touched cells are appended to sixteen length classes (one reservation per tile and class)
//   W  fuse_cells   one thread per touched cell, longest lists first (waves of uniform length): the reference's state machine
// Order: a cell's records are applied in increasing id q K + k exactly as before (the lists are sorted, whatever order the
// atomics produced), so every map element keeps its bits.
// Capacity: a tile's records are ordered in LDS up to `cap` ids at a time; a denser tile is handled in runs of consecutive
// cells that fit, and a single cell with more records than `cap` (a degenerate scene) is ordered in global memory by one
// thread.  ESVO_FUSE_LDS_CAP / ESVO_FUSE_TILE_CAP / ESVO_FUSE_PMAX (read at esvo_create) force small capacities in the tests.
#define FT FUSE_TILE               // tile edge in cells (common.hpp)
#define FT_CELLS (FT * FT)
#define FUSE_LDS_CAP_MAX 3072      // record ids per run in LDS (dense path)
#define FUSE_SORT_TMP 512          // per-wave scratch of the cooperative sort of long lists
#define FUSE_NB 16                 // length classes of the touched cells
#define FUSE_SLICES 64              // each class's list is kept per slice of the tiles (tile % 64): 64x fewer hits per counter

__host__ __device__ inline int fuse_tiles_x(int W) { return (W + FT - 1) / FT; }
__host__ __device__ inline int fuse_tiles_y(int H) { return (H + FT - 1) / FT; }
// class = ceil(log2(n)): lists in a wave differ by at most 2x while cells stay in tile order inside a class, which keeps the
// propagated points they share in L1 / L2
__device__ inline u32 fuse_bucket(u32 n) {  // n >= 1: 1->0, 2->1, 3..4->2, 5..8->3, ...
  const u32 b = (n <= 1u) ? 0u : 32u - (u32)__builtin_clz(n - 1u);
  return b < FUSE_NB - 1u ? b : FUSE_NB - 1u;
}

template <int MODEL>
__global__ void __launch_bounds__(256, BACK_WAVES) propagate_kernel(FuseArgs a, DevParams p, int K) {
  const u32 q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= a.n_pts) return;
  u32 f = 0;  // frame of point q (binary search in the cumulative counts)
  {
    u32 lo = 0, hi = a.n_frames;
    while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (a.fr_cum[mid] <= q) lo = mid; else hi = mid; }
    f = lo;
  }
  const DevPoint prior = a.win[a.fr_off[f] + (q - a.fr_cum[f])];
  const double* pose = a.frame_pose_T + ((size_t)a.fr_slot[f] * a.max_poses + prior.pose_idx) * 16;
  double T[16];
  mat4_mul(a.T_frame_world, pose, T);  // T_frame_obs, DepthFusion.cpp:80
  DevPoint prop;
  double pp[3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
    pp[r] = ((T[r * 4 + 0] * prior.p_cam[0] + T[r * 4 + 1] * prior.p_cam[1]) + T[r * 4 + 2] * prior.p_cam[2]) + T[r * 4 + 3];
  double u, v;
  world2Cam(p.camL, pp, u, v);
  // boundaryCheck; NaN coordinates pass the reference's test (all comparisons false) and then floor(NaN) -> size_t is
  // undefined: treated as rejected
  if (!boundaryCheck(u, v, p.W, p.H) || !(u == u) || !(v == v)) {
    prop.row = 0xffffffffu;
    prop.col = 0;
    a.prop[q] = prop;
    return;
  }
  prop.row = (u32)(size_t)floor(v);
  prop.col = (u32)(size_t)floor(u);
  prop.x[0] = u;
  prop.x[1] = v;
  const double invDepth = 1.0 / pp[2];
  double denominator = (T[8] * prior.p_cam[0] + T[9] * prior.p_cam[1]) + T[11];
  denominator /= prior.p_cam[2];
  denominator += T[10];
  const double J = T[10] / (denominator * denominator);
  prop.inv_depth = invDepth;  // update_studentT / update on a fresh DepthPoint: plain assignment
  if constexpr (MODEL == FUSE_TDIST) {
    const double scale2 = J * J * prior.scale2;
    const double nu = prior.nu;
    prop.scale2 = scale2;
    prop.variance = nu / (nu - 2) * scale2;
    prop.nu = nu;
  } else {  // DepthPoint::update(invDepth, J^2 variance) on a new point + boundVariance (DepthPoint.cpp:140-164)
    double variance = J * J * prior.variance;
    if (variance < 1e-6) variance = 1e-6;
    prop.variance = variance;
    prop.scale2 = 0;  // the Gaussian update leaves scaleSquared_ / nu_ as constructed (zero here and in the oracle, Appendix A-8)
    prop.nu = 0;
  }
  prop.p_cam[0] = pp[0]; prop.p_cam[1] = pp[1]; prop.p_cam[2] = pp[2];
  prop.residual = prior.residual;
  prop.age = prior.age;
  prop.pose_idx = 0;
  prop.seq = q;
  a.prop[q] = prop;
  (void)K;
  const u32 tile = (prop.row / FT) * (u32)fuse_tiles_x(p.W) + prop.col / FT;
  // the list entry carries the point's cell with its id: the tile kernel decides what reaches it from the lists alone (reading
  // row / col out of the 104-byte records cost a cache line per candidate, 9 x 160 k lines per tick)
  const uint2 entry = make_uint2(q, (prop.row << 16) | prop.col);
  const u32 pos = atomicAdd(&a.tile_count[tile], 1u);
  if (pos < a.tile_cap) a.tile_pts[(size_t)tile * a.tile_cap + pos] = entry;
  else a.over_pts[atomicAdd(a.over_count, 1u)] = entry;  // the tile's list is full (tile_count keeps counting: the reader clamps)
}

// DepthPoint::update_studentT, DepthPoint.cpp:167-188
__device__ inline void update_studentT(MapCell& c, double invD, double s2, double var, double nu_in) {
  if (c.inv_depth > -1e-6) {
    const double nu_update = (c.nu < nu_in) ? c.nu : nu_in;  // std::min(nu, nu_)
    const Recip rsum = make_recip(c.scale2 + s2);  // three quotients share this divisor (fdiv.hpp)
    const double n1 = s2 * c.inv_depth + c.scale2 * invD;
    const double dd = c.inv_depth - invD;
    const double n2 = dd * dd;
    const double invDepth_update = div_by(n1, rsum);
    const double scale2_update = div_by((nu_update + div_by(n2, rsum)) / (nu_update + 1) * (c.scale2 * s2), rsum);
    c.inv_depth = invDepth_update;
    c.scale2 = scale2_update;
    c.nu = nu_update + 1;
    c.variance = c.nu / (c.nu - 2) * c.scale2;
    c.age++;
  } else {
    c.inv_depth = invD; c.scale2 = s2; c.variance = var; c.nu = nu_in;
  }
}

// one record applied to a cell: DepthFusion::fusion's body (DepthFusion.cpp:119-190; naive_propagation: :262-282)
template <int MODEL>
__device__ inline void fuse_record(const DevParams& p, MapCell& c, bool& exists, u32& numFusion, const DevPoint& prop, u32 id, int crow,
                                   int ccol) {
  if (!exists) {  // case 1: DepthFusion.cpp:127-146 (naive_propagation: :262-272)
    c.row = (u32)crow; c.col = (u32)ccol;
    c.x[0] = (double)ccol + 0.5; c.x[1] = (double)crow + 0.5;
    c.inv_depth = prop.inv_depth; c.scale2 = prop.scale2; c.variance = prop.variance; c.nu = prop.nu;
    if (MODEL != FUSE_TDIST && c.variance < 1e-6) c.variance = 1e-6;  // DepthPoint::update -> boundVariance
    c.residual = prop.residual;
    c.age = prop.age;
    cam2World(p.camL, c.x[0], c.x[1], prop.inv_depth, c.p_cam);
    c.seq = id;
    exists = true;
    return;
  }
  if constexpr (MODEL == FUSE_NAIVE) {  // case 2 of naive_propagation, :273-282
    if (c.inv_depth > prop.inv_depth) return;  // the propagated point is farther
    if (prop.residual < c.residual) {
      c.row = prop.row; c.col = prop.col;        // `get(row, col) = dp_prop`: its row / col / x travel with it (Appendix A-7)
      c.x[0] = prop.x[0]; c.x[1] = prop.x[1];
      c.inv_depth = prop.inv_depth; c.scale2 = prop.scale2; c.nu = prop.nu; c.variance = prop.variance;
      c.residual = prop.residual; c.age = prop.age;
      c.p_cam[0] = prop.p_cam[0]; c.p_cam[1] = prop.p_cam[1]; c.p_cam[2] = prop.p_cam[2];
    }
    return;
  } else {
    bool compatible;
    if constexpr (MODEL == FUSE_L2) {  // chiSquareTest, :207-218
      const double d = prop.inv_depth - c.inv_depth, dd = d * d;
      compatible = dd / prop.variance + dd / c.variance < 5.99;
    } else {  // studentTCompatibleTest, :220-231
      const double s1 = sqrt(prop.variance), s2 = sqrt(c.variance), diff = fabs(prop.inv_depth - c.inv_depth);
      compatible = diff < 2 * s1 || diff < 2 * s2;
    }
    if (compatible) {  // case 2.1
      if constexpr (MODEL == FUSE_L2) {  // DepthPoint::update, DepthPoint.cpp:146-164
        if (c.inv_depth > -1e-6) {
          const double temp = c.inv_depth, tv = c.variance;
          c.inv_depth = (tv * prop.inv_depth + prop.variance * temp) / (tv + prop.variance);
          c.variance = (tv * prop.variance) / (tv + prop.variance);
        } else {
          c.inv_depth = prop.inv_depth;
          c.variance = prop.variance;
        }
        if (c.variance < 1e-6) c.variance = 1e-6;
      } else {
        update_studentT(c, prop.inv_depth, prop.scale2, prop.variance, prop.nu);
      }
      c.age++;                                                        // :171
      c.residual = (prop.residual < c.residual) ? prop.residual : c.residual;  // std::min
      cam2World(p.camL, c.x[0], c.x[1], prop.inv_depth, c.p_cam);     // :173-175
      numFusion++;
    } else {  // case 2.2
      if (c.inv_depth - 2 * sqrt(c.variance) > prop.inv_depth) return;  // occluded
      if (prop.variance < c.variance && prop.residual < c.residual) {
        // operator=: the propagated point's row/col/x travel with it (Appendix A-7)
        c.row = prop.row; c.col = prop.col;
        c.x[0] = prop.x[0]; c.x[1] = prop.x[1];
        c.inv_depth = prop.inv_depth; c.scale2 = prop.scale2; c.nu = prop.nu; c.variance = prop.variance;
        c.residual = prop.residual; c.age = prop.age;
        c.p_cam[0] = prop.p_cam[0]; c.p_cam[1] = prop.p_cam[1]; c.p_cam[2] = prop.p_cam[2];
      }
    }
  }
}

// T: one wave per 8 x 8-cell tile, one lane per cell -- the cells' sorted record lists.
// Candidates: the points of the tile's own list and of its eight neighbours' (+ the shared overflow list) whose footprint
// reaches the tile -- its own points and a one-cell rim of the neighbours'.
//   fast path (at most FUSE_PMAX candidates, i.e. every tile of an ordinary scene): the candidates are RANKED by id in LDS
//     (id_j < id_i counted over all j: the ids are unique), every (candidate, cell) record sets bit `rank` of the cell's bit
//     row, and a cell's list is its row read lowest bit first -- born sorted, no per-cell sort, no fill pass
//   dense path: per-cell histogram / scan / fill with LDS atomics in runs of cells that fit `cap` ids, each list ordered in
//     LDS (insertion for short ones, the wave's rank sort for long ones); a single cell beyond `cap` is ordered in global
//     memory by its lane
#define FUSE_PMAX 1024
#define FUSE_BITW (FUSE_PMAX / 32 + 1)   // words per cell row (+1: no bank conflicts between the lanes' rows)
#define FUSE_BUF (FUSE_LDS_CAP_MAX > 2 * FUSE_PMAX + FT_CELLS * FUSE_BITW ? FUSE_LDS_CAP_MAX : 2 * FUSE_PMAX + FT_CELLS * FUSE_BITW)
__global__ void __launch_bounds__(FT_CELLS, 8) tile_lists_kernel(FuseArgs a, DevParams p, int K, int radius, u32 cap, u32 pmax) {
  __shared__ u32 s_buf[FUSE_BUF];        // fast path: ids | (row, col) | bit rows; dense path: record ids
  __shared__ u32 s_cnt[FT_CELLS];        // dense path: records per cell of the tile
  __shared__ u32 s_off[FT_CELLS + 1];    //             their exclusive scan
  __shared__ u32 s_fill[FT_CELLS];
  __shared__ u32 s_tmp[FUSE_SORT_TMP];
  __shared__ u32 s_scan[1];
  __shared__ u32 s_hist[FUSE_NB], s_hbase[FUSE_NB];
  const int lane = threadIdx.x;
  const int tiles_x = fuse_tiles_x(p.W), tiles_y = fuse_tiles_y(p.H);
  const int tx = (int)(blockIdx.x % tiles_x), ty = (int)(blockIdx.x / tiles_x);
  const int r0 = ty * FT, c0 = tx * FT;
  const int crow = r0 + lane / FT, ccol = c0 + lane % FT;
  const int ncell = p.W * p.H;
  const bool in_img = crow < p.H && ccol < p.W;
  const bool in_band = in_img && crow >= p.cband_y0 && crow < p.cband_y1;
  const int cell = crow * p.W + ccol;
  const u64 lt_mask = (1ull << lane) - 1ull;
  if (lane < FUSE_NB) s_hist[lane] = 0;
  // what fuse_reset cleared per cell: the regulariser's owner marks (reg_view re-creates them by atomics)
  if (in_img && a.owner_max) { a.owner_max[cell] = 0; a.owner_min[cell] = 0xffffffffu; }
  // footprint of a point (DepthFusion.cpp:98-117): rows prow + [lo, hi], the same for the columns
  const int f_lo = radius == 0 ? 0 : -1, f_hi = 1;
  auto reaches = [&](u32 prow, u32 pcol) {
    return (int)prow + f_hi >= r0 && (int)prow + f_lo < r0 + FT && (int)pcol + f_hi >= c0 && (int)pcol + f_lo < c0 + FT;
  };
  // every candidate list in turn: fn(q) for the points this lane is dealt (call sites are wave-uniform)
  const u32 n_over = *a.over_count;
  auto for_lists = [&](auto&& fn) {
    for (int dy = -1; dy <= 1; ++dy) {
      const int ny = ty + dy;
      if (ny < 0 || ny >= tiles_y) continue;
      for (int dx = -1; dx <= 1; ++dx) {
        const int nx = tx + dx;
        if (nx < 0 || nx >= tiles_x) continue;
        const int nt = ny * tiles_x + nx;
        u32 cnt = a.tile_count[nt];
        if (cnt > a.tile_cap) cnt = a.tile_cap;
        fn(a.tile_pts + (size_t)nt * a.tile_cap, cnt);
      }
    }
    fn(a.over_pts, n_over);
  };
  // ---- gather the candidates that reach the tile, compacted: ids and packed (row, col) ----
  u32* s_q = s_buf;
  u32* s_rc = s_buf + FUSE_PMAX;
  u32 P = 0;
  // the ten lists as one flat index range (wave-uniform): list l covers [lpref[l], lpref[l + 1])
  const uint2* lptr[10];
  u32 lpref[11];
  {
    int l = 0;
    lpref[0] = 0;
    for (int dy = -1; dy <= 1; ++dy)
      for (int dx = -1; dx <= 1; ++dx) {
        const int ny = ty + dy, nx = tx + dx;
        u32 cnt = 0;
        const uint2* ptr = a.tile_pts;
        if (ny >= 0 && ny < tiles_y && nx >= 0 && nx < tiles_x) {
          const int nt = ny * tiles_x + nx;
          cnt = a.tile_count[nt];
          if (cnt > a.tile_cap) cnt = a.tile_cap;
          ptr = a.tile_pts + (size_t)nt * a.tile_cap;
        }
        lptr[l] = ptr;
        lpref[l + 1] = lpref[l] + cnt;
        ++l;
      }
    lptr[9] = a.over_pts;
    lpref[10] = lpref[9] + n_over;
  }
  const u32 C = lpref[10];
  constexpr int GQ = 16;  // candidates per lane held in registers: all loads of a phase are issued back to back
  for (u32 cbase = 0; cbase < C; cbase += (u32)GQ * ESVO_WAVE) {  // (one trip for all but the densest neighbourhoods)
    u32 qv[GQ], rcv[GQ];
#pragma unroll
    for (int j = 0; j < GQ; ++j) {
      const u32 i = cbase + (u32)j * ESVO_WAVE + (u32)lane;
      qv[j] = 0;
      rcv[j] = 0xffffffffu;
      if (i < C) {
        int l = 0;
#pragma unroll
        for (int t = 1; t < 10; ++t) l += (i >= lpref[t]) ? 1 : 0;
        const uint2 e = lptr[l][i - lpref[l]];
        qv[j] = e.x;
        rcv[j] = e.y;
      }
    }
#pragma unroll
    for (int j = 0; j < GQ; ++j) {
      const u32 i = cbase + (u32)j * ESVO_WAVE + (u32)lane;
      const bool keep = i < C && reaches(rcv[j] >> 16, rcv[j] & 0xffffu);
      const u64 km = __ballot(keep);
      if (keep) {
        const u32 pos = P + (u32)__popcll(km & lt_mask);
        if (pos < FUSE_PMAX) { s_q[pos] = qv[j]; s_rc[pos] = rcv[j]; }  // (beyond: the dense path re-reads the lists)
      }
      P += (u32)__popcll(km);
    }
  }
  __syncthreads();
  u32 n = 0, ex = 0, total = 0;
  u32* s_bits = s_buf + 2 * FUSE_PMAX;
  const bool fast = P <= pmax;   // pmax <= FUSE_PMAX (smaller: tests)
  if (fast) {
    u32* s_q2 = s_q;     // ranked IN PLACE: every lane reads its candidates and counts before anyone writes
    u32* s_rc2 = s_rc;
    for (u32 w = (u32)lane; w < FT_CELLS * FUSE_BITW; w += ESVO_WAVE) s_bits[w] = 0;
    // (as many candidates per lane as P needs: 1, 4 or 16 -- the counting loop costs P x that many compares)
    auto rank_in_place = [&](auto rq_tag) {
      constexpr int RQ = decltype(rq_tag)::value;
      u32 myq[RQ], myrc[RQ], myr[RQ];
#pragma unroll
      for (int j = 0; j < RQ; ++j) {
        const u32 i = (u32)j * ESVO_WAVE + (u32)lane;
        myq[j] = i < P ? s_q[i] : 0u;
        myrc[j] = i < P ? s_rc[i] : 0u;
        myr[j] = 0;
      }
      for (u32 jj = 0; jj < P; ++jj) {  // the ids below the lane's own, counted over all candidates (LDS broadcast)
        const u32 qj = s_q[jj];
#pragma unroll
        for (int j = 0; j < RQ; ++j) myr[j] += (qj < myq[j]) ? 1u : 0u;
      }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < RQ; ++j) {
        const u32 i = (u32)j * ESVO_WAVE + (u32)lane;
        if (i < P) { s_q2[myr[j]] = myq[j]; s_rc2[myr[j]] = myrc[j]; }
      }
      __syncthreads();
    };
    if (P <= ESVO_WAVE) {
      rank_in_place(std::integral_constant<int, 1>{});
    } else if (P <= 2 * ESVO_WAVE) {
      rank_in_place(std::integral_constant<int, 2>{});
    } else if (P <= 4 * ESVO_WAVE) {
      rank_in_place(std::integral_constant<int, 4>{});
    } else if (P <= 8 * ESVO_WAVE) {
      rank_in_place(std::integral_constant<int, 8>{});   // (measured at P = 425: 29 k cycles; the network below: 62 k)
    } else {
      // still more candidates: a bitonic network over the (id, cell) pairs in LDS, O(P log^2 P) -- counting ranks costs
      // P^2 / 64 compares per lane
      u32 N2 = 1024;
      while (N2 < P) N2 <<= 1;
      for (u32 i = P + (u32)lane; i < N2; i += ESVO_WAVE) s_q[i] = 0xffffffffu;  // padding sorts to the end
      __syncthreads();
      for (u32 k = 2; k <= N2; k <<= 1)
        for (u32 j = k >> 1; j > 0; j >>= 1) {
          for (u32 t = (u32)lane; t < N2 / 2; t += ESVO_WAVE) {
            const u32 i = ((t & ~(j - 1u)) << 1) | (t & (j - 1u)), l = i | j;
            const u32 qa = s_q[i], qb = s_q[l];
            const bool up = (i & k) == 0;
            if ((qa > qb) == up) {
              s_q[i] = qb; s_q[l] = qa;
              const u32 ra = s_rc[i], rb = s_rc[l];
              s_rc[i] = rb; s_rc[l] = ra;
            }
          }
          __syncthreads();
        }
    }
    for (u32 r = (u32)lane; r < P; r += ESVO_WAVE) {  // one bit per record
      const u32 rc = s_rc2[r];
      const u32 prow = rc >> 16, pcol = rc & 0xffffu;
      for (int k = 0; k < K; ++k) {
        int row, col;
        if (!fusion_cell(prow, pcol, k, radius, p.W, p.H, row, col)) continue;
        if (row < r0 || row >= r0 + FT || col < c0 || col >= c0 + FT) continue;
        if (row < p.cband_y0 || row >= p.cband_y1) continue;
        atomicOr(&s_bits[((row - r0) * FT + (col - c0)) * FUSE_BITW + (r >> 5)], 1u << (r & 31u));
      }
    }
    __syncthreads();
    const u32 nw = (P + 31u) >> 5;
    for (u32 w = 0; w < nw; ++w) n += (u32)__popc(s_bits[lane * FUSE_BITW + w]);
  } else {
    s_cnt[lane] = 0;
    s_fill[lane] = 0;
    __syncthreads();
  }
  // one candidate point of the dense path: fn(local cell, record id) for each of its records in a cell of this tile
  auto records_of = [&](uint2 e, auto&& fn) {
    const u32 q = e.x, prow = e.y >> 16, pcol = e.y & 0xffffu;
    for (int k = 0; k < K; ++k) {
      int row, col;
      if (!fusion_cell(prow, pcol, k, radius, p.W, p.H, row, col)) continue;
      if (row < r0 || row >= r0 + FT || col < c0 || col >= c0 + FT) continue;
      if (row < p.cband_y0 || row >= p.cband_y1) continue;
      fn((row - r0) * FT + (col - c0), q * (u32)K + (u32)k);
    }
  };
  auto expand = [&](auto&& fn) {
    for_lists([&](const uint2* list, u32 cnt) {
      for (u32 i = (u32)lane; i < cnt; i += ESVO_WAVE) records_of(list[i], fn);
    });
  };
  if (!fast) {
    expand([&](int lc, u32) { atomicAdd(&s_cnt[lc], 1u); });
    __syncthreads();
    n = s_cnt[lane];
  }
  {
    u32 tot;
    ex = block_excl_scan<1>(n, &tot, s_scan);
    total = tot;
    s_off[lane] = ex;
    if (lane == 0) s_off[FT_CELLS] = tot;
  }
  // the tile's lists, contiguous in rec_ids: its own fixed region; a tile with more records reserves behind the regions (a
  // counter every tile of the launch would hit costs ~20 ns per tile at the memory side: 4800 tiles, 0.1 ms)
  u32 gbase = blockIdx.x * a.tile_rec;
  if (total > a.tile_rec) {
    if (lane == 0) gbase = (u32)(gridDim.x * a.tile_rec) + atomicAdd(a.rec_cursor, total);
    gbase = (u32)__shfl((int)gbase, 0);
  }
  // the touched cells by length class: rank inside the tile, one reservation per class
  u32 rank = 0, bk = 0;
  if (n > 0) { bk = fuse_bucket(n); rank = atomicAdd(&s_hist[bk], 1u); }
  __syncthreads();
  // (one counter per class AND slice of the tiles -- tile % FUSE_SLICES -- for the same reason; the slices' lists are disjoint
  //  regions of the class's list, fuse_turn_kernel lays them end to end)
  const u32 slice = blockIdx.x % FUSE_SLICES;
  if (lane < FUSE_NB && s_hist[lane]) s_hbase[lane] = atomicAdd(&a.class_count[lane * FUSE_SLICES + slice], s_hist[lane]);
  if (in_band) {
    a.cell_count[cell] = n;
    if (n == 0) map_flags(a.map, ncell)[cell] = 0;  // nothing lands here: the cell reads empty
  }
  if (total == 0) return;  // (wave-uniform)
  if (n > 0) a.cell_offset[cell] = gbase + ex;
  __syncthreads();
  if (n > 0) a.cell_list[((size_t)bk * FUSE_SLICES + slice) * a.slice_cap + s_hbase[bk] + rank] = (u32)cell;
  u32* gout = a.rec_ids + gbase;
  if (fast) {  // ---- the cell's bit row, lowest rank first ----
    const u32* s_q2 = s_q;
    const u32* s_rc2 = s_rc;
    const u32 nw = (P + 31u) >> 5;
    u32 o = ex;
    for (u32 w = 0; w < nw; ++w) {
      u32 m = s_bits[lane * FUSE_BITW + w];
      while (m) {
        const u32 r = (w << 5) + (u32)__builtin_ctz(m);
        m &= m - 1u;
        const u32 rc = s_rc2[r];
        const int dyc = crow - (int)(rc >> 16), dxc = ccol - (int)(rc & 0xffffu);
        const u32 k = radius == 0 ? (u32)(dyc * 2 + dxc) : (u32)((dyc + 1) * 3 + (dxc + 1));  // fusion_cell's k of this cell
        gout[o++] = s_q2[r] * (u32)K + k;
      }
    }
    return;
  }
  // ---- dense path: runs of consecutive cells whose records fit the LDS buffer together ----
  u32* s_ids = s_buf;
  u32 run0 = 0;
  while (run0 < FT_CELLS) {
    u32 run1 = run0 + 1;  // (every lane derives the same run: s_off is complete)
    while (run1 < FT_CELLS && s_off[run1 + 1] - s_off[run0] <= cap) ++run1;
    const u32 base = s_off[run0];
    const u32 run_n = s_off[run1] - base;
    const bool mine = (u32)lane >= run0 && (u32)lane < run1 && n > 0;
    if (run_n == 0) { run0 = run1; continue; }
    if (run_n <= cap) {
      // ---- fill: the run's records into LDS, each cell's contiguous ----
      expand([&](int lc, u32 id) {
        if ((u32)lc >= run0 && (u32)lc < run1) s_ids[s_off[lc] - base + atomicAdd(&s_fill[lc], 1u)] = id;
      });
      __syncthreads();
      u32* ids = s_ids + (s_off[lane] - base);
      // ---- order: short lists by insertion in place; long ones by the wave, rank-sorted through its scratch ----
      if (mine && n > 1 && n <= 24) {
        for (u32 i = 1; i < n; ++i) {
          const u32 key = ids[i];
          int j = (int)i - 1;
          while (j >= 0 && ids[j] > key) { ids[j + 1] = ids[j]; --j; }
          ids[j + 1] = key;
        }
      }
      u64 longm = __ballot(mine && n > 24);
      while (longm) {
        const int src = __ffsll((long long)longm) - 1;
        longm &= longm - 1;
        const u32 ln = (u32)__shfl((int)n, src), lo = (u32)__shfl((int)(s_off[lane] - base), src);
        u32* lid = s_ids + lo;
        if (ln <= FUSE_SORT_TMP) {
          for (u32 i = (u32)lane; i < ln; i += ESVO_WAVE) {
            const u32 vkey = lid[i];
            u32 rk = 0;
            for (u32 j = 0; j < ln; ++j) rk += (lid[j] < vkey);
            s_tmp[rk] = vkey;
          }
          // (one wave: its LDS operations execute in program order; the barrier only keeps the compiler from moving them)
          __builtin_amdgcn_wave_barrier();
          for (u32 i = (u32)lane; i < ln; i += ESVO_WAVE) lid[i] = s_tmp[i];
          __builtin_amdgcn_wave_barrier();
        } else if (lane == src) {  // longer than the scratch: one lane, insertion sort
          for (u32 i = 1; i < ln; ++i) {
            const u32 key = lid[i];
            int j = (int)i - 1;
            while (j >= 0 && lid[j] > key) { lid[j + 1] = lid[j]; --j; }
            lid[j + 1] = key;
          }
        }
      }
      __syncthreads();
      // ---- out: the run's lists as they lie in LDS (full lines) ----
      for (u32 i = (u32)lane; i < run_n; i += ESVO_WAVE) gout[base + i] = s_ids[i];
      __syncthreads();  // the buffer is reused by the next run
    } else {
      // ---- a single cell with more records than the LDS buffer holds (run1 == run0 + 1): filled straight into its segment
      // of rec_ids by the whole wave, ordered there by its own lane ----
      u32* gids = gout + base;
      expand([&](int lc, u32 id) {
        if ((u32)lc == run0) gids[atomicAdd(&s_fill[lc], 1u)] = id;
      });
      __threadfence_block();
      __syncthreads();
      if (mine) {
        for (u32 i = 1; i < n; ++i) {
          const u32 key = gids[i];
          long long j = (long long)i - 1;
          while (j >= 0 && gids[j] > key) { gids[j + 1] = gids[j]; --j; }
          gids[j + 1] = key;
        }
      }
      __syncthreads();
    }
    run0 = run1;
  }
}

// W: one thread per touched cell, the longest lists first: walk the cell's (sorted) records (DepthFusion::fusion).  The first
// threads also clear the fusion front's counters for the next tick (the tile kernel has read them).
#ifndef FUSE_BLOCK
#define FUSE_BLOCK 256
#endif
template <int MODEL>
__global__ void __launch_bounds__(FUSE_BLOCK, FUSE_WAVES) fuse_cells_kernel(FuseArgs a, DevParams p, int K, int n_tiles) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  const int ncell = p.W * p.H;
  // thread t -> the t-th touched cell in the order "longest class first": class_total[s] = cells before segment s, segments
  // ordered (class descending, slice ascending) by fuse_turn_kernel; [FUSE_NB * FUSE_SLICES] = all touched cells
  constexpr u32 NSEG = FUSE_NB * FUSE_SLICES;
  __shared__ u32 s_seg[NSEG + 1];  // (staged: ten dependent look-ups per thread cost one global latency instead of ten)
  for (u32 i = threadIdx.x; i <= NSEG; i += FUSE_BLOCK) s_seg[i] = a.class_total[i];
  __syncthreads();
  if (t >= s_seg[NSEG]) return;
  u32 lo = 0, hi = NSEG;  // the segment with s_seg[seg] <= t < s_seg[seg + 1]
  while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (s_seg[mid] <= t) lo = mid; else hi = mid; }
  const u32 cls = FUSE_NB - 1u - lo / FUSE_SLICES, slc = lo % FUSE_SLICES;
  const int cell = (int)a.cell_list[((size_t)cls * FUSE_SLICES + slc) * a.slice_cap + (t - s_seg[lo])];
  const int crow = cell / p.W, ccol = cell - crow * p.W;
  const u32 n = a.cell_count[cell];
  const u32* ids = a.rec_ids + a.cell_offset[cell];
  MapCell c;
  bool exists = false;
  u32 numFusion = 0;
  u32 id_nxt = ids[0];
  DevPoint nxt = a.prop[id_nxt / (u32)K];
  for (u32 i = 0; i < n; ++i) {
    const u32 id = id_nxt;
    const DevPoint prop = nxt;
    if (i + 1 < n) {  // software prefetch: the next record does not depend on the cell state
      id_nxt = ids[i + 1];
      nxt = a.prop[id_nxt / (u32)K];
    }
    fuse_record<MODEL>(p, c, exists, numFusion, prop, id, crow, ccol);
  }
  a.map[cell] = c;
  map_flags(a.map, ncell)[cell] = CELL_ALIVE | CELL_GRID;
  if (numFusion) atomicAdd(a.d_num_fusion, numFusion);
  (void)n_tiles;
}

// between T and W (one small workgroup: 4 waves -- a 1024-thread workgroup needs 16 free wave slots on ONE compute unit and,
// beside the LM kernel's long-lived waves, waited ~0.3 ms for them): the (class, slice) sizes become the exclusive scan the walk
// indexes by -- classes in DESCENDING order, slices ascending --, every counter of the front is cleared for the next tick, this
// tick's statistics are cleared / set
#define FUSE_TURN_B 256
__global__ void __launch_bounds__(FUSE_TURN_B) fuse_turn_kernel(FuseArgs a, int n_tiles) {
  __shared__ u32 lds[FUSE_TURN_B / ESVO_WAVE];
  constexpr u32 NSEG = FUSE_NB * FUSE_SLICES;
  constexpr u32 PER = NSEG / FUSE_TURN_B;
  static_assert(NSEG % FUSE_TURN_B == 0, "segments per thread");
  u32 c[PER], sum = 0;
#pragma unroll
  for (u32 k = 0; k < PER; ++k) {
    const u32 seg = threadIdx.x * PER + k;
    const u32 cls = FUSE_NB - 1u - seg / FUSE_SLICES, slc = seg % FUSE_SLICES;
    c[k] = a.class_count[cls * FUSE_SLICES + slc];
    a.class_count[cls * FUSE_SLICES + slc] = 0;
    sum += c[k];
  }
  u32 tot;
  u32 ex = block_excl_scan<FUSE_TURN_B / ESVO_WAVE>(sum, &tot, lds);
#pragma unroll
  for (u32 k = 0; k < PER; ++k) { a.class_total[threadIdx.x * PER + k] = ex; ex += c[k]; }
  for (int i = (int)threadIdx.x; i < n_tiles; i += FUSE_TURN_B) a.tile_count[i] = 0;
  if (threadIdx.x == 0) {
    a.class_total[NSEG] = tot;
    *a.n_touched = tot;
    *a.d_total = *a.rec_cursor;   // (records beyond the tiles' own regions: a density statistic)
    *a.rec_cursor = 0;
    *a.over_count = 0;
    *a.d_num_fusion = 0;
    if (a.n_reg_elems) *a.n_reg_elems = 0;
  }
}

void launch_fuse(const FuseArgs& a, const DevParams& p, hipStream_t s) {
  const int ncell = p.W * p.H;
  const int model = a.naive ? FUSE_NAIVE : (p.ls_norm == ESVO_LSNORM_L2 ? FUSE_L2 : FUSE_TDIST);
  const int K = (model == FUSE_NAIVE || p.fusion_radius == 0) ? 4 : 9;
  const int radius = model == FUSE_NAIVE ? 0 : p.fusion_radius;
  const int n_tiles = fuse_tiles_x(p.W) * fuse_tiles_y(p.H);
  if (a.n_pts) {
    const dim3 g((a.n_pts + 255) / 256), b(256);
    if (model == FUSE_TDIST) hipLaunchKernelGGL(propagate_kernel<FUSE_TDIST>, g, b, 0, s, a, p, K);
    else if (model == FUSE_L2) hipLaunchKernelGGL(propagate_kernel<FUSE_L2>, g, b, 0, s, a, p, K);
    else hipLaunchKernelGGL(propagate_kernel<FUSE_NAIVE>, g, b, 0, s, a, p, K);
  }
  const u32 cap = (a.lds_cap >= 1 && a.lds_cap <= FUSE_LDS_CAP_MAX) ? a.lds_cap : FUSE_LDS_CAP_MAX;
  const u32 pmax = a.pmax_plus1 ? std::min<u32>(a.pmax_plus1 - 1u, FUSE_PMAX) : FUSE_PMAX;
  hipLaunchKernelGGL(tile_lists_kernel, dim3(n_tiles), dim3(FT_CELLS), 0, s, a, p, K, radius, cap, pmax);
  hipLaunchKernelGGL(fuse_turn_kernel, dim3(1), dim3(FUSE_TURN_B), 0, s, a, n_tiles);
  {
    const dim3 g((ncell + FUSE_BLOCK - 1) / FUSE_BLOCK), b(FUSE_BLOCK);
    if (model == FUSE_TDIST) hipLaunchKernelGGL(fuse_cells_kernel<FUSE_TDIST>, g, b, 0, s, a, p, K, n_tiles);
    else if (model == FUSE_L2) hipLaunchKernelGGL(fuse_cells_kernel<FUSE_L2>, g, b, 0, s, a, p, K, n_tiles);
    else hipLaunchKernelGGL(fuse_cells_kernel<FUSE_NAIVE>, g, b, 0, s, a, p, K, n_tiles);
  }
}

// ---- SmartGrid::clean ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) clean_kernel(MapCell* __restrict__ map, DevParams p) {
  const int cell = blockIdx.x * blockDim.x + threadIdx.x;
  if (cell >= p.W * p.H) return;
  { const int row = cell / p.W; if (row < p.cband_y0 || row >= p.cband_y1) return; }
  u32* mf = map_flags(map, p.W * p.H);
  if (!(mf[cell] & CELL_ALIVE)) return;  // (dense flags: a dead cell costs 4 contiguous bytes)
  const MapCell& c = map[cell];
  // DepthPoint::valid(var, age, max, min), DepthPoint.cpp:221-230
  const bool valid = c.inv_depth > -1e-6 && (double)c.age >= p.age_thr && c.variance <= p.var_thr &&
                     c.inv_depth <= p.invdepth_max && c.inv_depth >= p.invdepth_min;
  if (valid) return;
  // erase the element: its true cell reads empty; the reference also NULLs the grid entry of the
  // cell the element believes it occupies (SmartGrid.h:239), orphaning that cell's element.
  atomicAnd(&mf[cell], ~(CELL_ALIVE | CELL_GRID));
  const u32 b = c.row * (u32)p.W + c.col;
  if (b != (u32)cell && c.row < (u32)p.H && c.col < (u32)p.W) atomicAnd(&mf[b], ~CELL_GRID);
}
void launch_clean(MapCell* map, const DevParams& p, hipStream_t s) {
  const int ncell = p.W * p.H;
  hipLaunchKernelGGL(clean_kernel, dim3((ncell + 255) / 256), dim3(256), 0, s, map, p);
}

// ---- DepthRegularization::apply -------------------------------------------------------------------
// "Regularisation view" of the map: what the (2r+1)^2 neighbourhood scan reads, 32 B per cell instead of a 104 B MapCell.
//   ab[c] : (inv_depth, 2*sqrt(variance))  -- the closeness test operands; (NaN, NaN) where the tap is not a neighbour,
//                                             i.e. !(exists(r,c) && at(r,c).valid()): every comparison with it is false
//   cd[c] : (nu, scale2)                   -- read only for close neighbours
// The neighbour's sqrt is computed once per cell instead of once per tap.
// The same pass records, per believed cell, the last and the first element set there (dmTmp.set(it->row(), it->col(), *it)
// in list order: the last one owns the cell, the first one fixes its position in dmTmp's list).
__global__ void __launch_bounds__(256) reg_view_kernel(const MapCell* __restrict__ map, MapCell* __restrict__ out,
                                                       double2* __restrict__ ab, double2* __restrict__ cd,
                                                       u32* __restrict__ owner_max, u32* __restrict__ owner_min,
                                                       u32* __restrict__ n_elems, int ncell, int W,
                                                       int band0, int band1, int view0, int view1, int l2) {
  const int cell = blockIdx.x * blockDim.x + threadIdx.x;
  bool alive = false;
  if (cell < ncell) {
    const u32 fl = map_flags(map, ncell)[cell];  // dense: the record itself is only read where an element lives
    const int row = cell / W;
    if (row >= view0 && row < view1) {  // the view covers the band's halo as well (multi-GPU: computed, not exchanged)
      const double nan = __longlong_as_double(0x7ff8000000000000ll);
      double2 vab = make_double2(nan, nan);
      if (fl & CELL_ALIVE) {
        const MapCell& n = map[cell];
        const u32 b = n.row * (u32)W + n.col;
        atomicMax(&owner_max[b], n.seq + 1u);
        atomicMin(&owner_min[b], n.seq);
        if ((fl & CELL_GRID) && n.inv_depth > -1e-6) {
          vab = make_double2(n.inv_depth, 2.0 * sqrt(n.variance));
          cd[cell] = l2 ? make_double2(n.variance, 0.0) : make_double2(n.nu, n.scale2);
        }
      }
      ab[cell] = vab;
    }
    if (row >= band0 && row < band1) {
      alive = (fl & CELL_ALIVE) != 0;
      if (!alive) map_flags(out, ncell)[cell] = 0;
    }
  }
  // number of elements of the band (a statistic).  n_elems is null when the tile kernel counts them instead (reg_apply_kernel, one
  // atomic per tile spread over its run time): one atomic per wave on ONE address from ~2000-4800 waves that all issue it at once
  // made this kernel last 34 us instead of 6 (the reference-faithful DSEC tick; 57 us in the throughput tick)
  const u64 am = __ballot(alive);
  if (n_elems && (threadIdx.x & 63) == 0 && am) atomicAdd(n_elems, (u32)__popcll(am));
}

// ---- DepthRegularization::apply, fused: neighbourhood scan + sequential Student-t fusion ---------------------------------
// One workgroup per REG_TX x REG_TY tile of the image.  The tile's elements (alive cells) are compacted onto the lanes of
// its waves -- lane = element -- and the rows of the view the tile's neighbourhoods cover are streamed through LDS one
// at a time (double-buffered, one barrier per row).  For a staged row every lane tests the (2r+1) taps of its window:
// a tap is a CLOSE neighbour if |rho_self - rho_n| < 2 sigma_self or < 2 sigma_n (DepthRegularization.cpp:45-48; taps
// that are no neighbours are NaN in the view and fail both).  The close taps of the row, lowest column first, then go
// through the sequential Student-t fusion (:66-98) straight away: rows ascend for every element, so each element sees
// its close neighbours in the reference's row-major order.  Nothing but the result leaves the chip: the per-element row
// masks that the two-kernel version wrote and re-read (328 B per element) live in a register for the duration of a
// row, and a view row is fetched once per tile instead of once per element.
// The counts decide at the end (neighbours > minN, close > minClose), so the fusion runs speculatively.
#define REG_TX 64
#ifndef REG_TY
#define REG_TY 8   // rows per tile.  Lanes are ELEMENTS (compacted), so what a tile costs is its element count rounded up to whole
                   // waves: at the headline workload's density (36 % of the cells alive) 4 rows fill 1.44 waves -> 2, 8 rows
                   // 2.9 -> 3 (lane use 72 % -> 96 %), and a staged row serves twice the cells.  Round 5, sustained headline:
                   // 1.298 -> 1.273 ms per tick (profiles/r05_ab_regty.txt); 16 rows (1024-thread workgroups) is slower (1.69).
#endif
#ifndef REG_RB
#define REG_RB 4   // view rows staged per barrier.  One row per barrier (rounds 2-4) made every one of the 46-50 steps of a tile wait
                   // for a global load issued only two steps earlier: a tile's critical path was ~50 load latencies.  Blocks of 4
                   // rows: 13 steps, each prefetching four rows while four are being scanned.
#endif
#define REG_MARGIN 1   // an element's believed (row, col) is at most one cell away from its true cell (Appendix A-7)
#define REG_MAXW 128   // staged columns: REG_TX + 2 * (R + REG_MARGIN) <= 64 + 2 * 32, two 64-lane halves
static_assert((REG_TX * REG_TY) % 64 == 0 && REG_TX == 64, "a wave is a tile row");
// SPARSE: the closeness test walks the neighbour bits of a window row instead of all its taps -- the layout for maps with few
// elements (launch_reg_apply picks it from the previous tick's element count; same results either way, the dense variant's code
// is untouched: putting both loops into one kernel cost it 8 spilled registers)
template <int RT, bool SPARSE = false>  // RegularizationRadius when it is one of the shipped values (5, 20): the tap loop unrolls; 0: any radius
__global__ void __launch_bounds__(REG_TX * REG_TY, BACK_WAVES) reg_apply_kernel(const MapCell* __restrict__ map, MapCell* __restrict__ out,
                                                                    const u32* __restrict__ owner_max,
                                                                    const u32* __restrict__ owner_min,
                                                                    const double2* __restrict__ ab,
                                                                    const double2* __restrict__ cd, DevParams p, u32* __restrict__ n_elems) {
  constexpr int NW = REG_TY;                        // waves of the workgroup
  constexpr int UNITS = 4 * REG_RB;                 // staging units of a block: (row, ab | cd, column half) x 64 lanes
  constexpr int UPW = (UNITS + NW - 1) / NW;        // units per wave
  __shared__ double2 s_ab[2][REG_RB][REG_MAXW];
  __shared__ double2 s_cd[2][REG_RB][REG_MAXW];
  __shared__ u64 s_vb[2][REG_RB][2];
  __shared__ u32 s_elem[REG_TX * REG_TY];
  __shared__ u32 s_wcount[REG_TY + 1];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int tiles_x = (p.W + REG_TX - 1) / REG_TX;
  const int tile_c0 = (int)(blockIdx.x % tiles_x) * REG_TX, tile_r0 = (int)(blockIdx.x / tiles_x) * REG_TY;
  const int R = RT > 0 ? RT : p.reg_radius, Wn = 2 * R + 1;
  // ---- the tile's elements, compacted in row-major order ----
  const int cr = tile_r0 + wv, cc = tile_c0 + lane;
  bool alive = false;
  if (cr < p.H && cc < p.W && cr >= p.band_y0 && cr < p.band_y1) alive = (map_flags(map, p.W * p.H)[cr * p.W + cc] & CELL_ALIVE) != 0;
  const u64 am = __ballot(alive);
  if (lane == 0) s_wcount[wv] = (u32)__popcll(am);
  __syncthreads();
  u32 base = 0, n_el = 0;
  for (int w = 0; w < REG_TY; ++w) { if (w < wv) base += s_wcount[w]; n_el += s_wcount[w]; }
  if (n_el == 0) return;  // block-uniform
  if (t == 0) atomicAdd(n_elems, n_el);  // the band's element count (reg_view_kernel's note)
  if (alive) s_elem[base + (u32)__popcll(am & ((1ull << lane) - 1ull))] = (u32)(cr * p.W + cc);
  __syncthreads();
  const bool has = (u32)t < n_el;
  int cell = 0, row = 0, col = 0;
  double inv = -1.0, sd_self2 = 0.0;
  u32 seq = 0;
  if (has) {
    cell = (int)s_elem[t];
    const MapCell& c = map[cell];
    row = (int)c.row; col = (int)c.col;
    inv = c.inv_depth;
    sd_self2 = 2.0 * sqrt(c.variance);
    seq = c.seq;
  }
  const u32 b = (u32)row * (u32)p.W + (u32)col;  // dmTmp.set(it->row(), it->col(), *it)
  const bool owner = has && owner_max[b] == seq + 1u;  // else: overwritten by a later element
  // SmartGrid::getNeighbourhood's loop bounds mix int and size_t (SmartGrid.h:373-375): for row < radius or
  // col < radius the loops never execute -> no neighbours at all.
  bool scan = owner && inv > -1e-6 && row >= R && col >= R;
  // staged window: rows [tile_r0 - M - R, tile_r0 + TY + M + R), columns [tile_c0 - M - R, tile_c0 + TX + M + R)
  const int sc0 = tile_c0 - REG_MARGIN - R, sr0 = tile_r0 - REG_MARGIN - R;
  const int sw = REG_TX + 2 * (R + REG_MARGIN), sh = REG_TY + 2 * (R + REG_MARGIN);
  const int off = col - R - sc0;  // first tap of the lane's window inside a staged row
  // believed position outside the staged margins (cannot happen with the reference's fusion, see REG_MARGIN): such an
  // element takes the plain path below, straight from the view in global memory
  bool slow = scan && (off < 0 || off + Wn > sw || row - R < sr0 || row + R >= sr0 + sh);
  if (slow) scan = false;
  const u64 wmask = (Wn >= 64) ? ~0ull : ((1ull << Wn) - 1ull);
  u32 nb = 0, nclose = 0;
  double nu_post = 0, inv_post = 0, s2_post = 0, nu_div = 0;
  Recip rnu = make_recip(1.0);
  bool first = true;
  // one Student-t fusion step (DepthRegularization.cpp:72-86); see fdiv.hpp for the quotients
  auto fuse_step = [&](double inv_obs, double nu_obs, double s2_obs) {
    if (first) {
      first = false;
      nu_post = nu_obs; inv_post = inv_obs; s2_post = s2_obs;
      nu_div = nu_post + 1;
      rnu = make_recip(nu_div);
      return;
    }
    const double nu_prior = nu_post, inv_prior = inv_post, s2_prior = s2_post;
    nu_post = (nu_obs < nu_prior) ? nu_obs : nu_prior;
    // nu_post is a running minimum, so the divisor nu_post + 1 is almost always the previous step's
    if (nu_post + 1 != nu_div) { nu_div = nu_post + 1; rnu = make_recip(nu_div); }
    const double ssum = s2_obs + s2_prior;  // == s2_prior + s2_obs
    const Recip rsum = make_recip(ssum);
    const double a1 = s2_obs * inv_prior + s2_prior * inv_obs;
    const double dd = inv_prior - inv_obs;
    const double a2 = dd * dd;
    const double pp = s2_prior * s2_obs;
    const double q1 = div_fast(a1, rsum);
    const double a3 = nu_post + div_fast(a2, rsum);
    const double a4 = div_fast(a3, rnu) * pp;
    const double q4 = div_fast(a4, rsum);
    const bool ok = (int)rnu.fast & (int)fdiv_ok_b4(ssum, a1, a2, a3, a4);
    if (ok) {
      inv_post = q1;
      s2_post = q4;
    } else {
      inv_post = a1 / ssum;
      s2_post = ((nu_post + a2 / ssum) / (nu_post + 1) * pp) / ssum;
    }
  };
  // ---- stream the view through LDS, REG_RB rows per barrier, double-buffered ----
  // unit u = 4 r + 2 a + hf: row r of the block, a = 0 the (inverse depth, 2 sigma) view / 1 the (nu, scale^2) view, hf the column
  // half; wave w stages the units w, w + NW, ...  The ballot of an `ab` unit is that half row's neighbour word.
  const double nan = __longlong_as_double(0x7ff8000000000000ll);
  double2 ld[UPW];
  const int n_blk = (sh + REG_RB - 1) / REG_RB;
  auto fetch_block = [&](int k) {
#pragma unroll
    for (int j = 0; j < UPW; ++j) {
      const int u = wv + NW * j;
      ld[j] = make_double2(nan, nan);
      if (u < UNITS && k < n_blk) {
        const int r = u >> 2, a = (u >> 1) & 1, hf = u & 1;
        const int y = k * REG_RB + r, sj = hf * 64 + lane;
        const int gy = sr0 + y, gx = sc0 + sj;
        if (sj < sw && y < sh && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) ld[j] = (a ? cd : ab)[gy * p.W + gx];
      }
    }
  };
  auto store_block = [&](int buf) {
#pragma unroll
    for (int j = 0; j < UPW; ++j) {
      const int u = wv + NW * j;
      if (u < UNITS) {  // wave-uniform
        const int r = u >> 2, a = (u >> 1) & 1, hf = u & 1;
        const int sj = hf * 64 + lane;
        (a ? s_cd : s_ab)[buf][r][sj] = ld[j];
        if (!a) {
          const u64 vm = __ballot(sj < sw && ld[j].x == ld[j].x);
          if (lane == 0) s_vb[buf][r][hf] = vm;
        }
      }
    }
  };
  DEV_REG_DECL  // (tools/reg_floor.py; empty in the product: dev_hooks.hpp)
  fetch_block(0);
  store_block(0);
  fetch_block(1);
  __syncthreads();
  for (int k = 0; k < n_blk; ++k) {
    const int buf = k & 1;
    DEV_REG_BLOCK_BEGIN();
#pragma unroll 1
    for (int r = 0; r < REG_RB; ++r) {
      const int gy = sr0 + k * REG_RB + r;
      DEV_REG_ROW_BEGIN(nclose)
      if (scan && gy >= row - R && gy <= row + R) {
        // neighbour bits of the lane's window in this row
        const u64 w0 = s_vb[buf][r][0], w1 = s_vb[buf][r][1];
        u64 bits = (off < 64) ? (w0 >> off) : 0ull;
        if (off > 0 && off < 64) bits |= w1 << (64 - off);
        if (off >= 64) bits = w1 >> (off - 64);
        bits &= wmask;
        nb += (u32)__popcll(bits);
        u32 cm2[2] = {0u, 0u};  // close taps 0..31 and 32..63 of the row
        if (bits) {
          // |rho_self - rho_n| < 2 sigma_self || < 2 sigma_n  ==  < max(2 sigma_self, 2 sigma_n); a NaN tap fails, as
          // fmax returns the other operand and the difference is NaN.  Bits are collected in two 32-bit halves.
          u32 lo = 0, hi = 0;
          const double2* tap = &s_ab[buf][r][off];
          if constexpr (SPARSE) {
            // a sparse map (a reference-faithful tick: 4 % of the cells alive, 1-2 neighbours per window row): test the taps that
            // ARE neighbours instead of all 2r + 1 -- the others are NaN in the view and fail the test anyway: same masks
#pragma unroll 1
            for (int half = 0; half < 2; ++half) {
              u32 m = half ? (u32)(bits >> 32) : (u32)bits;
              u32 acc = 0;
              while (m) {
                const int kk = __builtin_ctz(m);
                m &= m - 1u;
                const double2 q = tap[32 * half + kk];
                acc |= (fabs(inv - q.x) < fmax(sd_self2, q.y)) ? (1u << kk) : 0u;
              }
              if (half) hi = acc; else lo = acc;
            }
          } else if (RT > 0) {
#pragma unroll
            for (int dc = 0; dc < 2 * RT + 1; ++dc) {
              const double2 q = tap[dc];
              const bool close = fabs(inv - q.x) < fmax(sd_self2, q.y);
              if (dc < 32) lo |= close ? (1u << dc) : 0u; else hi |= close ? (1u << (dc - 32)) : 0u;
            }
          } else {
            for (int dc = 0; dc < Wn; ++dc) {
              const double2 q = tap[dc];
              const bool close = fabs(inv - q.x) < fmax(sd_self2, q.y);
              if (dc < 32) lo |= close ? (1u << dc) : 0u; else hi |= close ? (1u << (dc - 32)) : 0u;
            }
          }
          cm2[0] = lo; cm2[1] = hi;
        }
        nclose += (u32)__popc(cm2[0]) + (u32)__popc(cm2[1]);
        // lowest column first; two 32-bit masks (a 64-bit ctz / clear-lowest costs 8 instructions per step, these 3)
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
          u32 m = half ? cm2[1] : cm2[0];
          const int o2 = off + 32 * half;
          while (m) {
            const int kk = __builtin_ctz(m);
            m &= m - 1u;
            const double2 qa = s_ab[buf][r][o2 + kk], qc = s_cd[buf][r][o2 + kk];
            fuse_step(qa.x, qc.x, qc.y);
          }
        }
      }
      DEV_REG_ROW_END(nclose);
    }
    DEV_REG_BLOCK_END(lane, wv);
    if (k + 1 < n_blk) store_block(buf ^ 1);  // the block fetched one iteration ago
    fetch_block(k + 2);
    __syncthreads();
  }
  DEV_REG_DONE(t, lane, nclose);
  if (slow) {  // plain path: the reference's two loops on the view in global memory
    for (int r = row - R; r <= row + R && r < p.H; ++r)
      for (int c2 = col - R; c2 <= col + R && c2 < p.W; ++c2) {
        const double2 q = ab[r * p.W + c2];
        if (!(q.x == q.x)) continue;
        nb++;
        const double diff = fabs(inv - q.x);
        if (diff < sd_self2 || diff < q.y) {
          nclose++;
          const double2 qc = cd[r * p.W + c2];
          fuse_step(q.x, qc.x, qc.y);
        }
      }
  }
  if (!has) return;
  u32* of = map_flags(out, p.W * p.H);
  if (!owner) { of[cell] = 0; return; }
  MapCell c = map[cell];
  if (c.inv_depth > -1e-6)  // it->valid()
    c.inv_depth = (nb > (u32)p.reg_min_nb && nclose > (u32)p.reg_min_close) ? inv_post : -1.0;
  c.seq = owner_min[b];  // position of the first element set at that cell in dmTmp's list
  out[cell] = c;
  of[cell] = CELL_ALIVE | CELL_GRID;
}

// LSnorm "l2" (DepthRegularization.cpp:56-65): the new inverse depth is the inverse-variance weighted mean of the close
// neighbours -- two passes over them (the total first), so one thread per element walks its window in the view twice.  No
// shipped configuration selects it: the plain formulation, not the tile kernel.  cd holds (variance, 0) in this mode.
__global__ void __launch_bounds__(256) reg_apply_l2_kernel(const MapCell* __restrict__ map, MapCell* __restrict__ out,
                                                           const u32* __restrict__ owner_max, const u32* __restrict__ owner_min,
                                                           const double2* __restrict__ ab, const double2* __restrict__ cd, DevParams p) {
  const int cell = blockIdx.x * blockDim.x + threadIdx.x;
  if (cell >= p.W * p.H) return;
  { const int r0 = cell / p.W; if (r0 < p.band_y0 || r0 >= p.band_y1) return; }
  if (!(map_flags(map, p.W * p.H)[cell] & CELL_ALIVE)) return;
  MapCell c = map[cell];
  u32* of = map_flags(out, p.W * p.H);
  const int row = (int)c.row, col = (int)c.col, R = p.reg_radius;
  const u32 b = (u32)row * (u32)p.W + (u32)col;  // dmTmp.set(it->row(), it->col(), *it)
  if (owner_max[b] != c.seq + 1u) { of[cell] = 0; return; }  // overwritten by a later element
  if (c.inv_depth > -1e-6) {  // it->valid()
    u32 nb = 0, nclose = 0;
    double total = 0.0;
    const double sd_self2 = 2.0 * sqrt(c.variance);
    // SmartGrid::getNeighbourhood's loop bounds mix int and size_t (SmartGrid.h:373-375): row < radius or col < radius -> no neighbours
    const bool scan = row >= R && col >= R;
    if (scan)
      for (int r = row - R; r <= row + R && r < p.H; ++r)
        for (int c2 = col - R; c2 <= col + R && c2 < p.W; ++c2) {
          const double2 q = ab[r * p.W + c2];
          if (!(q.x == q.x)) continue;
          nb++;
          const double diff = fabs(c.inv_depth - q.x);
          if (diff < sd_self2 || diff < q.y) { nclose++; total += 1.0 / cd[r * p.W + c2].x; }
        }
    double mean = 0.0;
    const bool set = nb > (u32)p.reg_min_nb && nclose > (u32)p.reg_min_close;
    if (set)
      for (int r = row - R; r <= row + R && r < p.H; ++r)
        for (int c2 = col - R; c2 <= col + R && c2 < p.W; ++c2) {
          const double2 q = ab[r * p.W + c2];
          if (!(q.x == q.x)) continue;
          const double diff = fabs(c.inv_depth - q.x);
          if (diff < sd_self2 || diff < q.y) mean += q.x * (1.0 / cd[r * p.W + c2].x) / total;
        }
    c.inv_depth = set ? mean : -1.0;
  }
  c.seq = owner_min[b];
  out[cell] = c;
  of[cell] = CELL_ALIVE | CELL_GRID;
}

void launch_reg_view(const MapCell* map_in, MapCell* map_out, u32* owner_max, u32* owner_min, double2* ab, double2* cd,
                     u32* n_elems, const DevParams& p, hipStream_t s) {
  const int ncell = p.W * p.H;
  // owner_max / owner_min / n_elems were reset by fuse_reset_kernel (launch_fuse of the same tick)
  const int nb = (ncell + 255) / 256;
  if (p.ls_norm != ESVO_LSNORM_L2) n_elems = nullptr;  // the tile kernel counts (launch_reg_apply)
  hipLaunchKernelGGL(reg_view_kernel, dim3(nb), dim3(256), 0, s, map_in, map_out, ab, cd, owner_max, owner_min, n_elems, ncell, p.W,
                     p.band_y0, p.band_y1, p.cband_y0, p.cband_y1, p.ls_norm == ESVO_LSNORM_L2 ? 1 : 0);
}
void launch_reg_apply(const MapCell* map_in, MapCell* map_out, const u32* owner_max, const u32* owner_min, const double2* ab,
                      const double2* cd, u32* n_elems, const DevParams& p, hipStream_t s, bool sparse) {
  if (p.ls_norm == ESVO_LSNORM_L2) {
    const int ncell = p.W * p.H;
    hipLaunchKernelGGL(reg_apply_l2_kernel, dim3((ncell + 255) / 256), dim3(256), 0, s, map_in, map_out, owner_max, owner_min, ab, cd, p);
    return;
  }
  const int tiles_x = (p.W + REG_TX - 1) / REG_TX;
  const int ty0 = p.band_y0 / REG_TY, ty1 = (p.band_y1 + REG_TY - 1) / REG_TY;  // tile rows that intersect the band
  if (ty1 <= ty0) return;
  // grid = all tile rows (blockIdx -> tile); tiles outside the band find no element and leave at once
  const int tiles_y = (p.H + REG_TY - 1) / REG_TY;
  const dim3 grid(tiles_x * tiles_y), block(REG_TX * REG_TY);
  if (sparse) {
    if (p.reg_radius == 20) hipLaunchKernelGGL((reg_apply_kernel<20, true>), grid, block, 0, s, map_in, map_out, owner_max, owner_min, ab, cd, p, n_elems);
    else if (p.reg_radius == 5) hipLaunchKernelGGL((reg_apply_kernel<5, true>), grid, block, 0, s, map_in, map_out, owner_max, owner_min, ab, cd, p, n_elems);
    else hipLaunchKernelGGL((reg_apply_kernel<0, true>), grid, block, 0, s, map_in, map_out, owner_max, owner_min, ab, cd, p, n_elems);
    return;
  }
  if (p.reg_radius == 20) hipLaunchKernelGGL(reg_apply_kernel<20>, grid, block, 0, s, map_in, map_out, owner_max, owner_min, ab, cd, p, n_elems);
  else if (p.reg_radius == 5) hipLaunchKernelGGL(reg_apply_kernel<5>, grid, block, 0, s, map_in, map_out, owner_max, owner_min, ab, cd, p, n_elems);
  else hipLaunchKernelGGL(reg_apply_kernel<0>, grid, block, 0, s, map_in, map_out, owner_max, owner_min, ab, cd, p, n_elems);
}

// ---- export: alive cells -> esvo_depth_point_t list (cell order; host orders by seq) --------------
__global__ void __launch_bounds__(256) map_flags_kernel(const MapCell* __restrict__ map, u32* __restrict__ flags, int ncell, int W,
                                                        int band0, int band1) {
  const int cell = blockIdx.x * blockDim.x + threadIdx.x;
  if (cell >= ncell) return;
  const int row = cell / W;
  flags[cell] = (row >= band0 && row < band1 && (map_flags(map, ncell)[cell] & CELL_ALIVE)) ? 1u : 0u;
}
__global__ void __launch_bounds__(256) map_export_kernel(const MapCell* __restrict__ map, const u32* __restrict__ flags,
                                                         const u32* __restrict__ prefix, esvo_depth_point_t* __restrict__ out,
                                                         u32* __restrict__ out_cell, int ncell) {
  const int cell = blockIdx.x * blockDim.x + threadIdx.x;
  if (cell >= ncell || !flags[cell]) return;
  const MapCell& c = map[cell];
  esvo_depth_point_t o;
  o.row = c.row; o.col = c.col;
  o.x[0] = c.x[0]; o.x[1] = c.x[1];
  o.inv_depth = c.inv_depth; o.scale2 = c.scale2; o.nu = c.nu; o.variance = c.variance; o.residual = c.residual;
  o.age = c.age;
  o.p_cam[0] = c.p_cam[0]; o.p_cam[1] = c.p_cam[1]; o.p_cam[2] = c.p_cam[2];
  o.pose_idx = 0;
  o.seq = c.seq;
  out[prefix[cell]] = o;
  if (out_cell) out_cell[prefix[cell]] = (map_flags(map, ncell)[cell] & CELL_GRID) ? (u32)cell : 0xffffffffu;
}
void launch_map_compact(const MapCell* map, u32* flags, u32* prefix, u32* d_total, u32* scan_tmp,
                        esvo_depth_point_t* out, u32* out_cell, const DevParams& p, hipStream_t s) {
  const int ncell = p.W * p.H;
  hipLaunchKernelGGL(map_flags_kernel, dim3((ncell + 255) / 256), dim3(256), 0, s, map, flags, ncell, p.W, p.band_y0, p.band_y1);
  launch_exclusive_scan_u32(flags, prefix, d_total, scan_tmp, (size_t)ncell, s);
  hipLaunchKernelGGL(map_export_kernel, dim3((ncell + 255) / 256), dim3(256), 0, s, map, flags, prefix, out, out_cell, ncell);
}

}  // namespace esvo
