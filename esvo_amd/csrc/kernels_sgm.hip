// kernels_sgm.hip — semi-global matching on the Time-Surface pair: the mapper's bootstrap.
//
// Replaces cv::StereoSGBM::compute as esvo_Mapping configures and calls it (esvo_core/src/esvo_Mapping.cpp:102-108,444:
// minDisparity 0, numDisparities 48, blockSize 11, P1 = 8*11*11, P2 = 32*11*11, uniquenessRatio 11, MODE_SGBM) and the
// edge-mask / DepthPoint loop + DepthFusion::naive_propagation of InitializationAtTime (:446-486, DepthFusion.cpp:234-288).
// OpenCV is third-party and absent here: the arithmetic follows the restatement in oracle/esvo_oracle.cpp (orc_sgbm_compute,
// "parity unpinned"), bit for bit -- it is all 8/16/32-bit integer work.
//
// Layout: the cost volumes are [y][x'][d] int16 with x' = x - numDisparities (only those columns are matched) and d
// innermost, so a wave reads / writes the D costs of one pixel as one coalesced row.  Path aggregation is the one
// sequential part: one wave per path, lane = disparity, the previous pixel's costs stay in registers, their minimum is a
// wave reduction, the d-1 / d+1 neighbours a lane shuffle.  The five directions are independent recursions, so all paths
// of all directions run side by side (a few thousand waves) instead of OpenCV's row-by-row raster.
#include "common.hpp"

namespace esvo {

void launch_fill_i16(int16_t* p, int16_t v, size_t n, hipStream_t st);

#define SGM_FTZERO 15
#define SGM_MAX_COST 32767

__device__ inline int sgm_clip(int v) { return min(max(v, -SGM_FTZERO), SGM_FTZERO) + SGM_FTZERO; }

// clipped x-Sobel and raw plane of one image; first / last column = tab[0]
__global__ void __launch_bounds__(256) sgm_prefilter_kernel(const uint8_t* __restrict__ img, uint8_t* __restrict__ sob,
                                                            uint8_t* __restrict__ raw, int W, int H) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= W * H) return;
  const int y = i / W, x = i - y * W;
  int s = sgm_clip(0), r = sgm_clip(0);
  if (x >= 1 && x < W - 1) {
    const uint8_t* r0 = img + (size_t)y * W;
    const uint8_t* rn = img + (size_t)(y > 0 ? y - 1 : y) * W;
    const uint8_t* rs = img + (size_t)(y < H - 1 ? y + 1 : y) * W;
    s = sgm_clip((r0[x + 1] - r0[x - 1]) * 2 + rn[x + 1] - rn[x - 1] + rs[x + 1] - rs[x - 1]);
    r = r0[x];
  }
  sob[i] = (uint8_t)s;
  raw[i] = (uint8_t)r;
}

__device__ inline int sgm_bt(const uint8_t* __restrict__ p1, const uint8_t* __restrict__ p2, int x, int xr, int W) {
  const int u = p1[x];
  const int ul = x > 0 ? (u + p1[x - 1]) / 2 : u, ur = x < W - 1 ? (u + p1[x + 1]) / 2 : u;
  const int u0 = min(min(ul, ur), u), u1 = max(max(ul, ur), u);
  const int v = p2[xr];
  const int vl = xr < W - 1 ? (v + p2[xr + 1]) / 2 : v, vr = xr > 0 ? (v + p2[xr - 1]) / 2 : v;
  const int v0 = min(min(vl, vr), v), v1 = max(max(vl, vr), v);
  const int c0 = max(max(0, u - v1), v0 - u);
  const int c1 = max(max(0, v - u1), u0 - v);
  return min(c0, c1);
}
// Birchfield-Tomasi pixel cost: Sobel plane + (raw plane >> 2)
__global__ void __launch_bounds__(256) sgm_pixcost_kernel(const uint8_t* __restrict__ sobL, const uint8_t* __restrict__ rawL,
                                                          const uint8_t* __restrict__ sobR, const uint8_t* __restrict__ rawR,
                                                          int16_t* __restrict__ pix, int W, int H, int D) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int width1 = W - D;
  if (i >= (size_t)H * width1 * D) return;
  const int d = (int)(i % D);
  const int xp = (int)((i / D) % width1);
  const int y = (int)(i / ((size_t)D * width1));
  const int x = xp + D, xr = x - d;
  const size_t row = (size_t)y * W;
  pix[i] = (int16_t)(sgm_bt(sobL + row, sobR + row, x, xr, W) + (sgm_bt(rawL + row, rawR + row, x, xr, W) >> 2));
}
// box sums with replicated borders: horizontal (over x') and vertical (over y)
__global__ void __launch_bounds__(256) sgm_hsum_kernel(const int16_t* __restrict__ pix, int16_t* __restrict__ hs, int width1, int H,
                                                       int D, int SW2) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)H * width1 * D) return;
  const int d = (int)(i % D);
  const int xp = (int)((i / D) % width1);
  const size_t rowbase = (i / ((size_t)D * width1)) * (size_t)width1 * D;
  int sum = 0;
  for (int k = -SW2; k <= SW2; ++k) sum += pix[rowbase + (size_t)min(max(xp + k, 0), width1 - 1) * D + d];
  hs[i] = (int16_t)sum;
}
__global__ void __launch_bounds__(256) sgm_vsum_kernel(const int16_t* __restrict__ hs, int16_t* __restrict__ C, int width1, int H, int D,
                                                       int SH2) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t rowN = (size_t)width1 * D;
  if (i >= (size_t)H * rowN) return;
  const int y = (int)(i / rowN);
  const size_t o = i - (size_t)y * rowN;
  int sum = 0;
  for (int k = -SH2; k <= SH2; ++k) sum += hs[(size_t)min(max(y + k, 0), H - 1) * rowN + o];
  C[i] = (int16_t)sum;
}

__device__ inline int wave_min_i32(int v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
  return v;
}
// One wave per path of one direction; lane = disparity.  dir: 0 from the left, 1 from up-left, 2 from above,
// 3 from up-right, 4 from the right (the neighbour q = p + (qx, qy)).  L(q, .) = 0 outside the matched range.
__global__ void __launch_bounds__(64) sgm_path_kernel(const int16_t* __restrict__ C, int16_t* __restrict__ L0, int16_t* __restrict__ L1,
                                                      int16_t* __restrict__ L2, int16_t* __restrict__ L3, int16_t* __restrict__ L4,
                                                      int width1, int H, int D, int P1, int P2) {
  const int d = threadIdx.x;
  int id = blockIdx.x;
  int dir, x, y, sx, sy;
  const int n_diag = width1 + H - 1;
  if (id < H) { dir = 0; x = 0; y = id; sx = 1; sy = 0; }
  else if ((id -= H) < n_diag) { dir = 1; sx = 1; sy = 1; if (id < width1) { x = id; y = 0; } else { x = 0; y = id - width1 + 1; } }
  else if ((id -= n_diag) < width1) { dir = 2; x = id; y = 0; sx = 0; sy = 1; }
  else if ((id -= width1) < n_diag) { dir = 3; sx = -1; sy = 1; if (id < width1) { x = id; y = 0; } else { x = width1 - 1; y = id - width1 + 1; } }
  else { id -= n_diag; dir = 4; x = width1 - 1; y = id; sx = -1; sy = 0; }
  int16_t* __restrict__ L = dir == 0 ? L0 : (dir == 1 ? L1 : (dir == 2 ? L2 : (dir == 3 ? L3 : L4)));
  const size_t rowN = (size_t)width1 * D;
  const bool lane_ok = d < D;
  int prev = 0;      // L(q, d); the path starts outside the range: zeros
  int minq = 0;
  while (x >= 0 && x < width1 && y < H) {
    const size_t o = (size_t)y * rowN + (size_t)x * D + d;
    const int c = lane_ok ? C[o] : 0;
    int pm = __shfl_up(prev, 1, 64), pp = __shfl_down(prev, 1, 64);
    pm = d > 0 ? pm : SGM_MAX_COST;
    pp = d < D - 1 ? pp : SGM_MAX_COST;
    const int delta = minq + P2;
    const int v = c + min(prev, min(pm + P1, min(pp + P1, delta))) - delta;
    const int stored = (int)(int16_t)v;  // Lr is kept as short
    if (lane_ok) L[o] = (int16_t)v;
    prev = lane_ok ? stored : 0;
    minq = wave_min_i32(lane_ok ? stored : SGM_MAX_COST);
    x += sx;
    y += sy;
  }
}

__device__ inline int sgm_sat16(int v) { return v > 32767 ? 32767 : (v < -32768 ? -32768 : v); }
// One wave per matched pixel: total cost, winner, uniqueness, sub-pixel step; the right-image candidate of the
// left-right check goes into d2key[y][x2] by atomicMin on (cost, processing order) -- OpenCV scans x downwards and
// replaces only on a strictly smaller cost.
__global__ void __launch_bounds__(64) sgm_select_kernel(const int16_t* __restrict__ L0, const int16_t* __restrict__ L1,
                                                        const int16_t* __restrict__ L2, const int16_t* __restrict__ L3,
                                                        const int16_t* __restrict__ L4, int16_t* __restrict__ d1, u32* __restrict__ d2key,
                                                        int W, int H, int D, int uniqueness) {
  const int d = threadIdx.x;
  const int width1 = W - D;
  const int xp = blockIdx.x % width1, y = blockIdx.x / width1;
  const size_t o = ((size_t)y * width1 + xp) * D + d;
  const bool ok = d < D;
  int S = SGM_MAX_COST;
  if (ok) S = sgm_sat16(sgm_sat16((int)L0[o] + L1[o] + L2[o] + L3[o]) + L4[o]);
  const int minS = wave_min_i32(ok ? S : 0x7fffffff);
  int best = -1;
  if (minS < SGM_MAX_COST) best = __ffsll((unsigned long long)__ballot(ok && S == minS)) - 1;  // first minimum
  const bool bad = ok && S * (100 - uniqueness) < minS * 100 && abs(best - d) > 1;
  if (__ballot(bad) != 0ull) return;  // not unique: the pixel keeps INVALID
  // (best == -1 happens only if every cost is SHRT_MAX; OpenCV would then index Sp[-1]: treated as no match)
  if (best < 0) return;
  const int Sm = __shfl(S, max(best - 1, 0), 64), Sb = __shfl(S, best, 64), Sq = __shfl(S, min(best + 1, 63), 64);
  if (d != 0) return;
  const int x = xp + D;
  const int x2 = x - best;
  atomicMin(&d2key[(size_t)y * W + x2], ((u32)(minS + 32768) << 16) | (u32)(width1 - 1 - xp));
  int dd;
  if (0 < best && best < D - 1) {
    const int denom2 = max(Sm + Sq - 2 * Sb, 1);
    dd = best * 16 + ((Sm - Sq) * 16 + denom2) / (denom2 * 2);
  } else {
    dd = best * 16;
  }
  d1[(size_t)y * W + x] = (int16_t)dd;
}
// left-right consistency (tolerance 1) on the raw disparities
__global__ void __launch_bounds__(256) sgm_lrcheck_kernel(const int16_t* __restrict__ d1, const u32* __restrict__ d2key,
                                                          int16_t* __restrict__ out, int W, int H, int D) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= W * H) return;
  const int y = i / W, x = i - y * W;
  const int width1 = W - D;
  int v = d1[i];
  if (x >= D && v != -16) {
    auto d2 = [&](int xx) -> int {  // disparity of the right-image pixel xx, or INVALID (-1)
      const u32 k = d2key[(size_t)y * W + xx];
      if (k == 0xffffffffu) return -1;
      const int xp = width1 - 1 - (int)(k & 0xffffu);
      return xp + D - xx;
    };
    const int lo = v >> 4, hi = (v + 15) >> 4;
    const int xa = x - lo, xb = x - hi;
    bool reject = false;
    if (0 <= xa && xa < W && 0 <= xb && xb < W) {
      const int da = d2(xa), db = d2(xb);
      reject = da >= 0 && abs(da - lo) > 1 && db >= 0 && abs(db - hi) > 1;
    }
    if (reject) v = -16;
  }
  out[i] = (int16_t)v;
}
// cv::medianBlur(disp, disp, 3) on int16, replicated border
__global__ void __launch_bounds__(256) sgm_median3_kernel(const int16_t* __restrict__ in, int16_t* __restrict__ out, int W, int H) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= W * H) return;
  const int y = i / W, x = i - y * W;
  int v[9];
  int k = 0;
#pragma unroll
  for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx) v[k++] = in[(size_t)min(max(y + dy, 0), H - 1) * W + min(max(x + dx, 0), W - 1)];
  // the median of nine by partial selection
#pragma unroll
  for (int a = 0; a < 5; ++a)
#pragma unroll
    for (int b = a + 1; b < 9; ++b) { const int lo = min(v[a], v[b]), hi = max(v[a], v[b]); v[a] = lo; v[b] = hi; }
  out[i] = (int16_t)v[4];
}

void launch_sgbm(const uint8_t* left, const uint8_t* right, const SgmScratch& s, int16_t* disp, int W, int H, hipStream_t st) {
  const int D = 48, block = 11, P1 = 8 * block * block, P2 = 32 * block * block, uniqueness = 11;
  const int width1 = W - D;
  const int npx = W * H;
  hipLaunchKernelGGL(sgm_prefilter_kernel, dim3((npx + 255) / 256), dim3(256), 0, st, left, s.sobL, s.rawL, W, H);
  hipLaunchKernelGGL(sgm_prefilter_kernel, dim3((npx + 255) / 256), dim3(256), 0, st, right, s.sobR, s.rawR, W, H);
  const size_t nvol = (size_t)H * width1 * D;
  const unsigned vb = (unsigned)((nvol + 255) / 256);
  hipLaunchKernelGGL(sgm_pixcost_kernel, dim3(vb), dim3(256), 0, st, s.sobL, s.rawL, s.sobR, s.rawR, s.vol[0], W, H, D);
  hipLaunchKernelGGL(sgm_hsum_kernel, dim3(vb), dim3(256), 0, st, s.vol[0], s.vol[1], width1, H, D, block / 2);
  hipLaunchKernelGGL(sgm_vsum_kernel, dim3(vb), dim3(256), 0, st, s.vol[1], s.vol[0], width1, H, D, block / 2);  // C = vol[0]
  const int n_paths = 2 * H + width1 + 2 * (width1 + H - 1);
  hipLaunchKernelGGL(sgm_path_kernel, dim3(n_paths), dim3(64), 0, st, s.vol[0], s.vol[1], s.vol[2], s.vol[3], s.vol[4], s.vol[5], width1, H, D,
                     P1, P2);
  hipMemsetAsync(s.d2key, 0xFF, sizeof(u32) * npx, st);
  launch_fill_i16(s.d1, (int16_t)-16, (size_t)npx, st);
  hipLaunchKernelGGL(sgm_select_kernel, dim3(width1 * H), dim3(64), 0, st, s.vol[1], s.vol[2], s.vol[3], s.vol[4], s.vol[5], s.d1, s.d2key, W, H,
                     D, uniqueness);
  hipLaunchKernelGGL(sgm_lrcheck_kernel, dim3((npx + 255) / 256), dim3(256), 0, st, s.d1, s.d2key, s.d1b, W, H, D);
  hipLaunchKernelGGL(sgm_median3_kernel, dim3((npx + 255) / 256), dim3(256), 0, st, s.d1b, disp, W, H);
}

__global__ void __launch_bounds__(256) fill_i16_kernel(int16_t* p, int16_t v, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
void launch_fill_i16(int16_t* p, int16_t v, size_t n, hipStream_t st) {
  hipLaunchKernelGGL(fill_i16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, v, n);
}

// ---- InitializationAtTime: edge mask (createEdgeMask, radius 0) AND disparity -> Gaussian DepthPoints -----------------
// event k of the selection = ring[(first - k) % cap] (newest first, as dataTransferring walks); flags + points in
// that order (the caller compacts them)
__global__ void __launch_bounds__(256) sgm_points_kernel(const esvo_event_t* __restrict__ ring, u64 first, u64 cap, u32 n,
                                                         const float2* __restrict__ lut, const int16_t* __restrict__ disp,
                                                         DevPoint* __restrict__ slots, u32* __restrict__ flags, DevParams p) {
  const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const esvo_event_t e = ring[(first - k) % cap];
  u32 keep = 0;
  if (e.x < p.W && e.y < p.H) {
    const float2 c = lut[(size_t)e.y * p.W + e.x];
    const int xc = (int)floor((double)c.x), yc = (int)floor((double)c.y);
    if (xc >= 0 && xc < p.W && yc >= 0 && yc < p.H) {
      const double dsp = disp[(size_t)yc * p.W + xc] / 16.0;
      const double inv = dsp / p.baseline_f;  // disp / (P(0,0) * baseline), esvo_Mapping.cpp:466
      if (!(dsp < 0) && !(inv < p.invdepth_min || inv > p.invdepth_max)) {
        DevPoint o;
        o.row = (u32)xc;  // DepthPoint dp(x, y): the constructor takes (row, col) -- as the reference wrote it (:463)
        o.col = (u32)yc;
        o.x[0] = xc * 1.0;
        o.x[1] = yc * 1.0;
        cam2World(p.camL, o.x[0], o.x[1], inv, o.p_cam);
        o.inv_depth = inv;
        o.scale2 = 0;   // the Gaussian update leaves scaleSquared_ / nu_ untouched (Appendix A-8: zero here and in the oracle)
        o.nu = 0;
        o.variance = 1e-6;  // pow(0.001, 2), bounded below by 1e-6 (DepthPoint::boundVariance)
        o.residual = 0;
        o.age = (u64)p.age_thr;
        o.pose_idx = 0;
        o.seq = k;
        slots[k] = o;
        keep = 1;
      }
    }
  }
  flags[k] = keep;
}

// DepthFusion::naive_propagation (DepthFusion.cpp:234-288) of the frame into the empty DepthFrame.  All residuals are
// zero, so an occupied cell is never replaced (`prop.residual < existing.residual` is false): a cell belongs to the FIRST
// point (lowest index) that touches it.  Pass 1 finds that point per cell (atomicMin on index*4 + k), pass 2 creates the
// cells; the creation rank (the element list order) comes from a scan over the (point, k) pairs that won.
struct SgmProp { double x[2], inv, var, p_cam[3]; int row, col; bool ok; };
__device__ inline SgmProp sgm_propagate(const DevPoint& pt, const double* T, const DevParams& p) {
  SgmProp o;
  double pp[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) pp[r] = ((T[r * 4 + 0] * pt.p_cam[0] + T[r * 4 + 1] * pt.p_cam[1]) + T[r * 4 + 2] * pt.p_cam[2]) + T[r * 4 + 3];
  world2Cam(p.camL, pp, o.x[0], o.x[1]);
  o.ok = !(o.x[0] < 0 || o.x[0] >= (double)p.W || o.x[1] < 0 || o.x[1] >= (double)p.H) && o.x[0] == o.x[0] && o.x[1] == o.x[1];
  o.row = (int)floor(o.x[1]);
  o.col = (int)floor(o.x[0]);
  double denominator = (T[8] * pt.p_cam[0] + T[9] * pt.p_cam[1]) + T[11];
  denominator /= pt.p_cam[2];
  denominator += T[10];
  const double J = T[10] / (denominator * denominator);
  o.inv = 1.0 / pp[2];
  o.var = J * J * pt.variance;
  if (o.var < 1e-6) o.var = 1e-6;
  o.p_cam[0] = pp[0]; o.p_cam[1] = pp[1]; o.p_cam[2] = pp[2];
  return o;
}
__global__ void __launch_bounds__(256) sgm_naive_owner_kernel(const DevPoint* __restrict__ pts, u32 n, const double* __restrict__ T_frame_obs,
                                                              u32* __restrict__ owner, DevParams p) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const SgmProp pr = sgm_propagate(pts[i], T_frame_obs, p);
  if (!pr.ok) return;
  for (int k = 0; k < 4; ++k) {
    const int row = pr.row + (k >> 1), col = pr.col + (k & 1);
    if (row >= p.H || col >= p.W) continue;
    atomicMin(&owner[(size_t)row * p.W + col], i * 4u + (u32)k);
  }
}
__global__ void __launch_bounds__(256) sgm_naive_flags_kernel(const u32* __restrict__ owner, u32* __restrict__ pair_flags, int ncell) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncell) return;
  const u32 o = owner[c];
  if (o != 0xffffffffu) pair_flags[o] = 1u;
}
__global__ void __launch_bounds__(256) sgm_naive_create_kernel(const DevPoint* __restrict__ pts, const double* __restrict__ T_frame_obs,
                                                               const u32* __restrict__ owner, const u32* __restrict__ pair_rank,
                                                               MapCell* __restrict__ map, DevParams p) {
  const int cell = blockIdx.x * blockDim.x + threadIdx.x;
  if (cell >= p.W * p.H) return;
  const u32 o = owner[cell];
  if (o == 0xffffffffu) return;
  const DevPoint pt = pts[o >> 2];
  const SgmProp pr = sgm_propagate(pt, T_frame_obs, p);
  MapCell c;
  const int row = cell / p.W, col = cell - row * p.W;
  c.x[0] = col + 0.5;
  c.x[1] = row + 0.5;
  c.inv_depth = pr.inv;                       // DepthPoint::update on a new point
  c.variance = pr.var < 1e-6 ? 1e-6 : pr.var;
  c.scale2 = 0;
  c.nu = 0;
  c.residual = pt.residual;
  c.age = pt.age;
  cam2World(p.camL, c.x[0], c.x[1], pr.inv, c.p_cam);
  c.row = (u32)row;
  c.col = (u32)col;
  c.seq = pair_rank[o];
  c.unused_ = 0;
  map[cell] = c;
  map_flags(map, p.W * p.H)[cell] = CELL_ALIVE | CELL_GRID;
}

void launch_sgm_points(const esvo_event_t* ring, u64 first, u64 cap, u32 n, const float2* lut, const int16_t* disp, DevPoint* slots,
                       u32* flags, const DevParams& p, hipStream_t st) {
  if (!n) return;
  hipLaunchKernelGGL(sgm_points_kernel, dim3((n + 255) / 256), dim3(256), 0, st, ring, first, cap, n, lut, disp, slots, flags, p);
}
void launch_sgm_naive(const DevPoint* pts, u32 n, const double* d_T_frame_obs, u32* owner, u32* pair_flags, u32* pair_rank, u32* d_total,
                      u32* scan_tmp, MapCell* map, const DevParams& p, hipStream_t st) {
  const int ncell = p.W * p.H;
  hipMemsetAsync(owner, 0xFF, sizeof(u32) * ncell, st);
  hipMemsetAsync(map, 0, map_buffer_bytes((size_t)ncell), st);  // cells + flags
  if (!n) return;
  hipMemsetAsync(pair_flags, 0, sizeof(u32) * 4 * (size_t)n, st);
  hipLaunchKernelGGL(sgm_naive_owner_kernel, dim3((n + 255) / 256), dim3(256), 0, st, pts, n, d_T_frame_obs, owner, p);
  hipLaunchKernelGGL(sgm_naive_flags_kernel, dim3((ncell + 255) / 256), dim3(256), 0, st, owner, pair_flags, ncell);
  launch_exclusive_scan_u32(pair_flags, pair_rank, d_total, scan_tmp, 4 * (size_t)n, st);
  hipLaunchKernelGGL(sgm_naive_create_kernel, dim3((ncell + 255) / 256), dim3(256), 0, st, pts, d_T_frame_obs, owner, pair_rank, map, p);
}

}  // namespace esvo
