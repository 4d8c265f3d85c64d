// kernels_ts.hip — Time-Surface raster on gfx950.
//
//   K1 ts_scatter : esvo_time_surface eventsCallback / EventQueueMat::insertEvent
//                   (esvo_time_surface/src/TimeSurface.cpp:403-425, TimeSurface.h:39-50)
//   K2 ts_render_fused : TimeSurface::createTimeSurfaceAtTime, BACKWARD mode
//                   (TimeSurface.cpp:52-152): exp decay -> x255 -> u8 -> 3x3 median -> rectifying remap, one LDS-staged pass
//   ts_decay_f64 + ts_forward_gather : the same function in FORWARD mode (TimeSurface.cpp:85-116)
//   gaussian5     : TimeSurfaceObservation::GaussianBlurTS(5) (TimeSurfaceObservation.h:107-116)
//
// Data layout: the per-pixel event queues of the reference (std::deque, length 20) collapse to a
// Surface of Active Events: one u64 per pixel holding (t_ns << 1 | polarity) of the newest
// event, updated with atomicMax.  The host only scatters events with ts < T before rendering at
// T, which is exactly what getMostRecentEventBeforeT (TimeSurface.h:52-75) returns.
#include <cstdlib>
#include "common.hpp"

namespace esvo {

// ---- K1 -----------------------------------------------------------------------------------------
// One thread per event, 16-byte coalesced loads (the reference's in-memory dvs_msgs::Event),
// 8-byte atomicMax into the SAE.  HBM-bound: 16 B read + 8 B atomic per event.
__device__ inline void ts_scatter_range(const uint4* __restrict__ ev, size_t n, u64* __restrict__ sae, int W, int H) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const uint4 e = ev[i];  // {x | y<<16, sec, nsec, polarity | pad}
    const u32 x = e.x & 0xffffu, y = e.x >> 16;
    if (x >= (u32)W || y >= (u32)H) continue;  // EventQueueMat::insideImage
    if (ev_is_late(e.w)) continue;  // arrived out of order: the reference's eventsCallback does not insert it (common.hpp, EV_LATE)
    const u64 t_ns = (u64)e.y * 1000000000ull + (u64)e.z;
    const u64 key = (t_ns << 1) | (u64)((e.w & 0xffu) ? 1u : 0u);
    atomicMax(&sae[(size_t)y * W + x], key);
  }
}
__global__ void __launch_bounds__(256) ts_scatter_kernel(const uint4* __restrict__ ev, size_t n, u64* __restrict__ sae,
                                                         int W, int H) {
  ts_scatter_range(ev, n, sae, W, H);
}
// up to four ring segments (two cameras, each range may wrap the ring once) in one launch: blockIdx.y = segment
__global__ void __launch_bounds__(256) ts_scatter_segs_kernel(TsScatterSegs g, int W, int H) {
  const int k = blockIdx.y;
  ts_scatter_range(reinterpret_cast<const uint4*>(g.ev[k]), g.n[k], g.sae[k], W, H);
}

void launch_ts_scatter(const esvo_event_t* d_ev, size_t n, u64* d_sae, int W, int H, hipStream_t s) {
  if (n == 0) return;
  size_t blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;  // grid-stride above 8 blocks/CU
  hipLaunchKernelGGL(ts_scatter_kernel, dim3((u32)blocks), dim3(256), 0, s, reinterpret_cast<const uint4*>(d_ev), n,
                     d_sae, W, H);
}
void launch_ts_scatter_segs(const TsScatterSegs& g, int n_seg, int W, int H, hipStream_t s) {
  size_t n_max = 0;
  for (int k = 0; k < n_seg; ++k) n_max = n_max > g.n[k] ? n_max : g.n[k];
  if (n_seg == 0 || n_max == 0) return;
  size_t blocks = (n_max + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(ts_scatter_segs_kernel, dim3((u32)blocks, (u32)n_seg), dim3(256), 0, s, g, W, H);
}

// ---- out-of-order packets (api_ts.hip, push_unsorted): the staged tail of the ring merged with a late packet -----------------
// plan[j]: source of merged position j -- bit 31 clear: staged[idx] (a copy of the ring's tail), set: packet[idx]
__global__ void __launch_bounds__(256) ts_merge_kernel(const uint4* __restrict__ staged, const uint4* __restrict__ packet,
                                                       const u32* __restrict__ plan, size_t n, uint4* __restrict__ ring, u64 first_slot,
                                                       u64 ring_cap) {
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const u32 src = plan[j];
  ring[(first_slot + j) % ring_cap] = (src & 0x80000000u) ? packet[src & 0x7fffffffu] : staged[src];
}
void launch_ts_merge(const esvo_event_t* staged, const esvo_event_t* packet, const u32* plan, size_t n, esvo_event_t* ring, u64 first_slot,
                     u64 ring_cap, hipStream_t s) {
  if (n == 0) return;
  hipLaunchKernelGGL(ts_merge_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const uint4*>(staged),
                     reinterpret_cast<const uint4*>(packet), plan, n, reinterpret_cast<uint4*>(ring), first_slot, ring_cap);
}

// ---- per-pixel event queues (max_event_queue_len > 0) --------------------------------------------------------------------
// EventQueueMat (TimeSurface.h:28-97): every received event is appended to its pixel's deque, the deque is trimmed to the
// newest L, and a render at T walks it back to the first event with ts < T.  Events arrive sorted by time (the ingest calls
// refuse anything else), so "the newest L by arrival" is "the L largest stamps", a SET: the queue is kept as an unordered
// set of <= L keys (t_ns << 1 | polarity; 0 = empty slot) in slot-major layout q[slot][pixel], an insertion replaces the
// smallest key once the set is full, and the render takes the largest key below T.  Insertion order inside a batch is
// therefore free, which is what lets a batch be inserted in parallel:
//   tsq_bin    thread / event: the event goes to the list of its 8x8-pixel tile (one atomic per event; lists of fixed
//              capacity, the rare rest to one overflow list every tile scans);
//   tsq_merge  one wave / tile, lane = pixel: the tile's queues in LDS, the tile's list broadcast entry by entry, the
//              owning lane inserts;
//   tsq_view   thread / pixel: the largest key with stamp < T -> the SAE word the render kernels read (they are unchanged).
// (Two events of one pixel with the same nanosecond stamp: the reference keeps arrival order, this keeps key order -- as the
// one-stamp path, visible only with ignore_polarity = false.)
#define TSQ_TILE 8
__global__ void __launch_bounds__(256) tsq_bin_kernel(TsQueueArgs a, int W, int H) {
  const int tiles_x = (W + TSQ_TILE - 1) / TSQ_TILE;
  for (int k = 0; k < 2; ++k) {
    const uint4* __restrict__ ev = reinterpret_cast<const uint4*>(a.ev[k]);
    const size_t n = a.n[k];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
      const uint4 e = ev[i];
      const u32 x = e.x & 0xffffu, y = e.x >> 16;
      if (x >= (u32)W || y >= (u32)H) continue;  // EventQueueMat::insideImage
      if (ev_is_late(e.w)) continue;  // (its place in the queues is taken by a copy of the then-newest event: api_ts.hip, push_unsorted)
      const u64 t_ns = (u64)e.y * 1000000000ull + (u64)e.z;
      const u64 key = (t_ns << 1) | (u64)((e.w & 0xffu) ? 1u : 0u);
      if (key <= 1ull) continue;  // a zero stamp never renders (TimeSurface.cpp:73) and would read as an empty slot
      const u32 tile = (y / TSQ_TILE) * (u32)tiles_x + x / TSQ_TILE;
      const u32 pix = (y % TSQ_TILE) * TSQ_TILE + x % TSQ_TILE;
      const u32 pos = atomicAdd(&a.tcount[tile], 1u);
      if (pos < a.tcap) {
        a.tlist[(size_t)tile * a.tcap + pos] = make_uint4((u32)key, (u32)(key >> 32), pix, 0u);
      } else {
        const u32 o = atomicAdd(a.over_count, 1u);
        if (o < a.over_cap) a.over[o] = make_uint4((u32)key, (u32)(key >> 32), pix, tile);
      }
    }
  }
}
__global__ void __launch_bounds__(64) tsq_merge_kernel(TsQueueArgs a, int W, int H) {
  __shared__ u64 s_q[TSQ_LMAX][64];
  __shared__ uint4 s_e[64];
  const int lane = threadIdx.x;
  const u32 tile = blockIdx.x;
  const u32 cnt = a.tcount[tile];
  const u32 n_over = min(*a.over_count, a.over_cap);
  if (cnt == 0 && n_over == 0) return;
  const u32 P = min(cnt, a.tcap);
  const int tiles_x = (W + TSQ_TILE - 1) / TSQ_TILE;
  const int x = (int)(tile % tiles_x) * TSQ_TILE + lane % TSQ_TILE, y = (int)(tile / tiles_x) * TSQ_TILE + lane / TSQ_TILE;
  const bool in_img = x < W && y < H;
  const size_t npx = (size_t)W * H, px = (size_t)y * W + x;
  const int L = a.L;
  int n = 0;  // keys in the set (the non-empty slots, compacted to the front)
  if (in_img)
    for (int sl = 0; sl < L; ++sl) {
      const u64 k = a.q[(size_t)sl * npx + px];
      if (k) s_q[n++][lane] = k;
    }
  bool changed = false;
  auto insert = [&](u64 key) {
    if (n < L) { s_q[n++][lane] = key; changed = true; return; }
    int m = 0;
    u64 mv = s_q[0][lane];
    for (int sl = 1; sl < L; ++sl) { const u64 v = s_q[sl][lane]; if (v < mv) { mv = v; m = sl; } }
    if (key >= mv) { s_q[m][lane] = key; changed = true; }  // (an equal stamp: the later arrival displaces the earlier one)
  };
  for (u32 base = 0; base < P; base += 64) {
    __syncthreads();
    if (base + lane < P) s_e[lane] = a.tlist[(size_t)tile * a.tcap + base + lane];
    __syncthreads();
    const u32 m = min(64u, P - base);
    for (u32 j = 0; j < m; ++j) {
      const uint4 e = s_e[j];
      if ((int)e.z == lane && in_img) insert(((u64)e.y << 32) | e.x);
    }
  }
  for (u32 base = 0; base < n_over; base += 64) {  // (rare: a tile that received more than its list holds)
    __syncthreads();
    if (base + lane < n_over) s_e[lane] = a.over[base + lane];
    __syncthreads();
    const u32 m = min(64u, n_over - base);
    for (u32 j = 0; j < m; ++j) {
      const uint4 e = s_e[j];
      if (e.w == tile && (int)e.z == lane && in_img) insert(((u64)e.y << 32) | e.x);
    }
  }
  if (changed)
    for (int sl = 0; sl < n; ++sl) a.q[(size_t)sl * npx + px] = s_q[sl][lane];
  if (lane == 0) a.tcount[tile] = 0;  // (only this workgroup reads it: the list is free for the next batch)
}
__global__ void __launch_bounds__(256) tsq_view_kernel(const u64* __restrict__ q, int L, size_t npx, u64 t_ns, u64* __restrict__ sae) {
  const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= npx) return;
  u64 best = 0;
  for (int sl = 0; sl < L; ++sl) {  // EventQueueMat::getMostRecentEventBeforeT, TimeSurface.h:52-75 (strict <)
    const u64 k = q[(size_t)sl * npx + p];
    if ((k >> 1) < t_ns && k > best) best = k;
  }
  sae[p] = best;
}
void launch_tsq_insert(const TsQueueArgs& a, int W, int H, hipStream_t s) {
  const size_t n = a.n[0] + a.n[1];
  if (n == 0) return;
  hipMemsetAsync(a.over_count, 0, sizeof(u32), s);
  size_t blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(tsq_bin_kernel, dim3((u32)blocks), dim3(256), 0, s, a, W, H);
  const int n_tiles = ((W + TSQ_TILE - 1) / TSQ_TILE) * ((H + TSQ_TILE - 1) / TSQ_TILE);
  hipLaunchKernelGGL(tsq_merge_kernel, dim3(n_tiles), dim3(64), 0, s, a, W, H);
}
void launch_tsq_view(const u64* q, int L, int W, int H, u64 t_ns, u64* sae, hipStream_t s) {
  const size_t npx = (size_t)W * H;
  hipLaunchKernelGGL(tsq_view_kernel, dim3((u32)((npx + 255) / 256)), dim3(256), 0, s, q, L, npx, t_ns, sae);
}

// ---- K2: decay + quantise ------------------------------------------------------------------------
// TimeSurface.cpp:65-127.  dt is formed like ros::Duration::toSec() (Appendix A-17); the u8
// conversion is cv::Mat::convertTo = saturate_cast<uchar>(cvRound(v)) = round-half-even.
// The quantised value only depends on which side of a half-integer 255 * exp(-dt / tau) falls, so an f32 evaluation
// (v_exp_f32) decides it whenever its result is further from the half-integer than its error bound allows; the f64
// evaluation -- the reference's arithmetic, ~150 instructions -- runs for the lanes that are not.  Error of the short
// form: the exponent x = dt * kf carries <= 1.8e-7 |x| relative (u64 -> f32, kf in f32, the product, the * log2 e inside
// __expf), v_exp_f32 and the final product 2 ulp, so |g32 - g| <= 255 e^x (1.8e-7 |x| + 1.2e-7) <= 4.8e-5 for every
// x <= 0; the guard band is 5e-4, ten times that.  About 0.1 % of the pixels (6 % of the waves) take both paths.
struct TsDecay { u64 t_ns; double decay_sec; float kf; int ignore_polarity; };  // kf = -1e-9 / decay_sec
__device__ inline int ts_decay_u8(u64 key, const TsDecay& d) {
  const u64 te = key >> 1;
  // no event before T (TimeSurface.cpp:71-73); ns_to_sec(te) > 0 <=> te != 0.  The empty value is 0, or 127.5 -> 128.
  if (!(key > 1ull && te < d.t_ns)) return d.ignore_polarity ? 0 : 128;
  const float e32 = __expf((float)(d.t_ns - te) * d.kf);
  const float g32 = d.ignore_polarity ? 255.0f * e32 : 127.5f * ((key & 1ull) ? 1.0f + e32 : 1.0f - e32);
  const float q32 = rintf(g32);
  if (fabsf(g32 - q32) < 0.5f - 5e-4f) return (int)q32;  // in [0, 255]: e32 in [0, 1]
  const double dt = duration_to_sec(d.t_ns, te);
  double e = exp(-dt / d.decay_sec);
  if (!d.ignore_polarity) e *= (key & 1ull) ? 1.0 : -1.0;
  const double g = d.ignore_polarity ? 255.0 * e : 255.0 * (e + 1.0) / 2.0;
  const int q = (int)rint(g);
  return q < 0 ? 0 : (q > 255 ? 255 : q);
}

// ---- 3x3 median (BORDER_REPLICATE) and the fixed-point bilinear remap, tap by tap from a raw image in memory: the last
// step of FORWARD mode, whose raw image is the output of the splat ---------------------------------------------------------
__device__ inline void cswap(int& a, int& b) { int lo = min(a, b), hi = max(a, b); a = lo; b = hi; }
__device__ inline int median9(int p0, int p1, int p2, int p3, int p4, int p5, int p6, int p7, int p8) {
  // 19-exchange median network
  cswap(p1, p2); cswap(p4, p5); cswap(p7, p8); cswap(p0, p1); cswap(p3, p4); cswap(p6, p7);
  cswap(p1, p2); cswap(p4, p5); cswap(p7, p8); cswap(p0, p3); cswap(p5, p8); cswap(p4, p7);
  cswap(p3, p6); cswap(p1, p4); cswap(p2, p5); cswap(p4, p7); cswap(p4, p2); cswap(p6, p4);
  cswap(p4, p2);
  return p4;
}
// median of n <= 49 values by rank (kernels 5x5 and 7x7: medianBlur(2k + 1), TimeSurface.cpp:130-131; no shipped configuration
// sets k > 1): the element with at most n/2 values below it and more than n/2 values not above it
#define TS_MEDIAN_K_MAX 3
__device__ inline int median_by_rank(const int* v, int n) {
  for (int i = 0; i < n; ++i) {
    int lt = 0, le = 0;
    for (int j = 0; j < n; ++j) { lt += v[j] < v[i]; le += v[j] <= v[i]; }
    if (lt <= n / 2 && le > n / 2) return v[i];
  }
  return v[0];  // (not reached)
}
template <bool BIGMED>
__device__ inline int median_tap(const uint8_t* __restrict__ raw, int W, int H, int x, int y, int median_k) {
  if (x < 0 || x >= W || y < 0 || y >= H) return 0;  // BORDER_CONSTANT 0 of cv::remap
  if (median_k <= 0) return raw[y * W + x];
  if constexpr (BIGMED) {
    int v[(2 * TS_MEDIAN_K_MAX + 1) * (2 * TS_MEDIAN_K_MAX + 1)];
    int n = 0;
    for (int dy = -median_k; dy <= median_k; ++dy)
      for (int dx = -median_k; dx <= median_k; ++dx)
        v[n++] = raw[min(max(y + dy, 0), H - 1) * W + min(max(x + dx, 0), W - 1)];
    return median_by_rank(v, n);
  }
  const int xm = max(x - 1, 0), xp = min(x + 1, W - 1), ym = max(y - 1, 0), yp = min(y + 1, H - 1);
  const uint8_t* r0 = raw + ym * W;
  const uint8_t* r1 = raw + y * W;
  const uint8_t* r2 = raw + yp * W;
  return median9(r0[xm], r0[x], r0[xp], r1[xm], r1[x], r1[xp], r2[xm], r2[x], r2[xp]);
}

// fixmap[i] = (cvRound(map_x*32), cvRound(map_y*32)): OpenCV's INTER_BITS=5 coordinate
// quantisation, precomputed once on the host (Appendix B.2).  Weights are the exact 15-bit
// integers (32-fx)(32-fy)*32 ...; dst = (sum + 16384) >> 15.
template <bool BIGMED>
__device__ inline int ts_median_remap_px(const uint8_t* __restrict__ raw, const int2* __restrict__ fixmap, int W, int H,
                                         int median_k, int x, int y) {
  const int i = y * W + x;
  int v;
  if (fixmap) {
    const int2 m = fixmap[i];
    const int ix = m.x >> 5, iy = m.y >> 5, fx = m.x & 31, fy = m.y & 31;
    const int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
    int acc = 0;
    if (w00) acc += w00 * median_tap<BIGMED>(raw, W, H, ix, iy, median_k);
    if (w01) acc += w01 * median_tap<BIGMED>(raw, W, H, ix + 1, iy, median_k);
    if (w10) acc += w10 * median_tap<BIGMED>(raw, W, H, ix, iy + 1, median_k);
    if (w11) acc += w11 * median_tap<BIGMED>(raw, W, H, ix + 1, iy + 1, median_k);
    v = (acc + 16384) >> 15;
  } else {
    v = median_tap<BIGMED>(raw, W, H, x, y, median_k);
  }
  return v;
}
template <bool BIGMED>
__global__ void __launch_bounds__(256) ts_median_remap_kernel(const uint8_t* __restrict__ raw, const int2* __restrict__ fixmap,
                                                              uint8_t* __restrict__ out, int W, int H, int median_k) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= W || y >= H) return;
  out[y * W + x] = (uint8_t)ts_median_remap_px<BIGMED>(raw, fixmap, W, H, median_k, x, y);
}
// ---- K2, fused: decay -> u8 -> 3x3 median -> rectifying remap in ONE pass, staged through LDS ----------------------------
// A workgroup renders a TSF_TX x TSF_TY tile of the RECTIFIED surface (threads 32 x 8, two rows each).  It first reads its
// pixels' fixed-point source coordinates and reduces the bounding box of the bilinear taps that carry weight; then
//   1. the decayed, quantised raw surface over that box (+ the median's one-pixel ring, coordinates clamped to the image =
//      BORDER_REPLICATE) goes from the SAE into LDS -- one 8-byte stamp and one exp per staged raw pixel, once;
//   2. the 3x3 median of the box goes LDS -> LDS, once per source pixel instead of once per bilinear tap (the two-kernel
//      version above evaluates four medians = 36 global byte loads per output pixel and writes / re-reads the raw image);
//   3. every thread blends its four taps from LDS with the exact 15-bit weights.
// The intermediate raw image never exists in memory: 8 B (stamp) + 8 B (map entry) read and 1-2 B written per output pixel.
// A tile whose box does not fit the staging buffers (it cannot for the lens models of the shipped rigs; a wildly distorted
// map could) takes the direct path: the same arithmetic, taps evaluated from the SAE one by one.
#define TSF_TX 32
#define TSF_TY 16
#define TSF_CAP 2560   // staged raw pixels (e.g. 64 x 40: the 32 x 16 tile sheared by the rectification, + taps, + the ring)
__device__ inline int ts_raw_at(const u64* __restrict__ sae, int W, int H, int x, int y, const TsDecay& d) {
  x = min(max(x, 0), W - 1); y = min(max(y, 0), H - 1);
  return ts_decay_u8(sae[(size_t)y * W + x], d);
}
template <bool BIGMED>
__device__ inline int ts_median_tap_direct(const u64* __restrict__ sae, int W, int H, int x, int y, int median_k, const TsDecay& d) {
  if (x < 0 || x >= W || y < 0 || y >= H) return 0;
  if (median_k <= 0) return ts_raw_at(sae, W, H, x, y, d);
  if constexpr (BIGMED) {
    int w[(2 * TS_MEDIAN_K_MAX + 1) * (2 * TS_MEDIAN_K_MAX + 1)];
    int n = 0;
    for (int dy = -median_k; dy <= median_k; ++dy)
      for (int dx = -median_k; dx <= median_k; ++dx) w[n++] = ts_raw_at(sae, W, H, x + dx, y + dy, d);
    return median_by_rank(w, n);
  }
  int v[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) v[k] = ts_raw_at(sae, W, H, x + k % 3 - 1, y + k / 3 - 1, d);
  return median9(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8]);
}
template <bool BIGMED>
__global__ void __launch_bounds__(256) ts_render_fused_kernel(TsPair c, int W, int H, TsDecay d, int median_k, int stage_cap, int ty0) {
  __shared__ uint8_t s_raw[TSF_CAP];
  __shared__ uint8_t s_med[TSF_CAP];
  __shared__ int s_bb[4];
  const int cam = blockIdx.z;
  const u64* __restrict__ sae = c.sae[cam];
  const int2* __restrict__ fixmap = c.fixmap[cam];
  const int t = threadIdx.x, tx = t & 31, ty = t >> 5;
  const int x = blockIdx.x * TSF_TX + tx;
  if (t < 4) s_bb[t] = (t < 2) ? 0x7fffffff : (int)0x80000000;
  __syncthreads();
  int ix[2], iy[2], fx[2], fy[2];
  bool in[2];
  int bx0 = 0x7fffffff, by0 = 0x7fffffff, bx1 = (int)0x80000000, by1 = (int)0x80000000;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int y = (blockIdx.y + ty0) * TSF_TY + ty + 8 * k;
    in[k] = x < W && y < H;
    ix[k] = iy[k] = fx[k] = fy[k] = 0;
    if (in[k]) {
      const int2 m = fixmap ? fixmap[(size_t)y * W + x] : make_int2(x << 5, y << 5);
      ix[k] = m.x >> 5; iy[k] = m.y >> 5; fx[k] = m.x & 31; fy[k] = m.y & 31;
      // the taps that carry weight (the others are never read: ts_median_remap_px), clipped to the image (outside: constant 0)
      const int lx = max(ix[k], 0), hx = min(ix[k] + (fx[k] ? 1 : 0), W - 1);
      const int ly = max(iy[k], 0), hy = min(iy[k] + (fy[k] ? 1 : 0), H - 1);
      if (lx <= hx && ly <= hy) { bx0 = min(bx0, lx); bx1 = max(bx1, hx); by0 = min(by0, ly); by1 = max(by1, hy); }
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    bx0 = min(bx0, __shfl_xor(bx0, d)); by0 = min(by0, __shfl_xor(by0, d));
    bx1 = max(bx1, __shfl_xor(bx1, d)); by1 = max(by1, __shfl_xor(by1, d));
  }
  if ((t & 63) == 0) { atomicMin(&s_bb[0], bx0); atomicMin(&s_bb[1], by0); atomicMax(&s_bb[2], bx1); atomicMax(&s_bb[3], by1); }
  __syncthreads();
  // the box is workgroup-uniform: kept in scalar registers, and so is everything derived from it
  const int mx0 = __builtin_amdgcn_readfirstlane(s_bb[0]), my0 = __builtin_amdgcn_readfirstlane(s_bb[1]);
  const int mx1 = __builtin_amdgcn_readfirstlane(s_bb[2]), my1 = __builtin_amdgcn_readfirstlane(s_bb[3]);
  const bool any = mx0 <= mx1;
  const int hal = median_k > 0 ? median_k : 0;
  const int mw = any ? mx1 - mx0 + 1 : 0, mh = any ? my1 - my0 + 1 : 0;
  const int rw = mw + 2 * hal, rh = mh + 2 * hal;
  const bool staged = any && (long long)rw * rh <= stage_cap;
  const uint8_t* med = s_raw;
  if (staged) {
    {  // element i = t, t + 256, ... of the rw x rh box: (rx, ry) advance by (256 % rw, 256 / rw), one division per thread
      const int sx = 256 % rw, sy = 256 / rw;
      int ry = t / rw, rx = t - ry * rw;
      for (int i = t; i < rw * rh; i += 256) {
        s_raw[i] = (uint8_t)ts_raw_at(sae, W, H, mx0 - hal + rx, my0 - hal + ry, d);
        rx += sx; ry += sy;
        if (rx >= rw) { rx -= rw; ++ry; }
      }
    }
    __syncthreads();
    if (hal) {
      const int sx = 256 % mw, sy = 256 / mw;
      int my = t / mw, mx = t - my * mw;
      for (int i = t; i < mw * mh; i += 256) {
        const uint8_t* r0 = s_raw + my * rw + mx;
        if constexpr (!BIGMED) {
          const uint8_t* r1 = r0 + rw;
          const uint8_t* r2 = r1 + rw;
          s_med[i] = (uint8_t)median9(r0[0], r0[1], r0[2], r1[0], r1[1], r1[2], r2[0], r2[1], r2[2]);
        } else {  // (2 hal + 1)^2 window, LDS -> LDS
          int v[(2 * TS_MEDIAN_K_MAX + 1) * (2 * TS_MEDIAN_K_MAX + 1)];
          int n = 0;
          for (int dy = 0; dy <= 2 * hal; ++dy)
            for (int dx = 0; dx <= 2 * hal; ++dx) v[n++] = r0[dy * rw + dx];
          s_med[i] = (uint8_t)median_by_rank(v, n);
        }
        mx += sx; my += sy;
        if (mx >= mw) { mx -= mw; ++my; }
      }
      __syncthreads();
      med = s_med;
    }
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    if (!in[k]) continue;
    const int y = (blockIdx.y + ty0) * TSF_TY + ty + 8 * k;
    auto tap = [&](int sx, int sy) -> int {
      if (sx < 0 || sx >= W || sy < 0 || sy >= H) return 0;  // BORDER_CONSTANT 0 of cv::remap
      if (staged) return med[(sy - my0) * mw + (sx - mx0)];
      return ts_median_tap_direct<BIGMED>(sae, W, H, sx, sy, median_k, d);
    };
    const int w00 = (32 - fx[k]) * (32 - fy[k]) * 32, w01 = fx[k] * (32 - fy[k]) * 32, w10 = (32 - fx[k]) * fy[k] * 32,
              w11 = fx[k] * fy[k] * 32;
    int acc = 0;
    if (w00) acc += w00 * tap(ix[k], iy[k]);
    if (w01) acc += w01 * tap(ix[k] + 1, iy[k]);
    if (w10) acc += w10 * tap(ix[k], iy[k] + 1);
    if (w11) acc += w11 * tap(ix[k] + 1, iy[k] + 1);
    const uint8_t v = (uint8_t)((acc + 16384) >> 15);
    c.out[cam][(size_t)y * W + x] = v;
    if (c.out2[cam]) c.out2[cam][(size_t)y * W + x] = v;
  }
}

// ESVO_TS_STAGE_CAP (tests only): a smaller staging capacity, down to 0 = every tile on the direct path
static int ts_stage_cap() {
  static const int cap = [] {
    const char* e = esvo_dev_switch("ESVO_TS_STAGE_CAP");
    const int v = e ? std::atoi(e) : TSF_CAP;
    return v < 0 ? 0 : (v > TSF_CAP ? TSF_CAP : v);
  }();
  return cap;
}
static TsDecay ts_decay_args(u64 t_ns, double decay_sec, int ignore_polarity) {
  return TsDecay{t_ns, decay_sec, (float)(-1e-9 / decay_sec), ignore_polarity};
}
static_assert(TSF_TY == TS_TILE_ROWS, "row bands are whole tiles");
// tile rows [ty0, ty0 + nty) that cover the rectified rows [row0, row1) (row1 < 0: the whole image)
static void ts_tile_rows(int H, int row0, int row1, int& ty0, int& nty) {
  if (row1 < 0 || row1 > H) row1 = H;
  if (row0 < 0) row0 = 0;
  ty0 = row0 / TSF_TY;
  nty = row1 > row0 ? (row1 + TSF_TY - 1) / TSF_TY - ty0 : 0;
}
void launch_ts_render(const u64* d_sae, const int2* d_fixmap, uint8_t* d_raw, uint8_t* d_out, int W, int H, u64 t_ns,
                      double decay_sec, int ignore_polarity, int median_k, hipStream_t s, int row0, int row1) {
  (void)d_raw;  // the raw image is an intermediate of FORWARD mode only
  TsPair c{};
  c.sae[0] = d_sae; c.fixmap[0] = d_fixmap; c.out[0] = d_out;
  launch_ts_render_pair(c, W, H, t_ns, decay_sec, ignore_polarity, median_k, s, row0, row1);
}

// (a TsPair whose second camera is empty renders one camera: grid.z = 1)
void launch_ts_render_pair(const TsPair& c, int W, int H, u64 t_ns, double decay_sec, int ignore_polarity, int median_k,
                           hipStream_t s, int row0, int row1) {
  int ty0, nty;
  ts_tile_rows(H, row0, row1, ty0, nty);
  if (nty <= 0) return;
  const dim3 grid((W + TSF_TX - 1) / TSF_TX, nty, c.sae[1] ? 2 : 1);
  if (median_k > 1)
    hipLaunchKernelGGL(ts_render_fused_kernel<true>, grid, dim3(256), 0, s, c, W, H, ts_decay_args(t_ns, decay_sec, ignore_polarity),
                       median_k, ts_stage_cap(), ty0);
  else
    hipLaunchKernelGGL(ts_render_fused_kernel<false>, grid, dim3(256), 0, s, c, W, H, ts_decay_args(t_ns, decay_sec, ignore_polarity),
                       median_k, ts_stage_cap(), ty0);
}

// ---- FORWARD mode (TimeSurface.cpp:85-116) ------------------------------------------------------------------------------
// Pass 1: the decayed value of every raw pixel as f64 (NaN: no event before T, :71-73).  Pass 2, one thread per DESTINATION
// pixel: the contributions the reference's raster-order splat would add to it, in that order (the list is sorted by source
// index on the host), each `+= w * expVal` followed by the clamp to 1 (:101-113); x255, round-half-even, u8 (:123-127).
__global__ void __launch_bounds__(256) ts_decay_f64_kernel(const u64* __restrict__ sae, double* __restrict__ val, int n_px, u64 t_ns,
                                                           double decay_sec, int ignore_polarity) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_px) return;
  const u64 key = sae[i];
  const u64 te = key >> 1;
  double v = __longlong_as_double(0x7ff8000000000000ll);
  if (key != 0 && te < t_ns && ns_to_sec(te) > 0) {
    const double dt = duration_to_sec(t_ns, te);
    double e = exp(-dt / decay_sec);
    if (!ignore_polarity) e *= (key & 1ull) ? 1.0 : -1.0;
    v = e;
  }
  val[i] = v;
}
__global__ void __launch_bounds__(256) ts_forward_gather_kernel(const u32* __restrict__ off, const u32* __restrict__ src,
                                                                const float2* __restrict__ lut, const double* __restrict__ val,
                                                                uint8_t* __restrict__ raw, int n_px, int ignore_polarity) {
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= n_px) return;
  double acc = 0.0;
  for (u32 k = off[d]; k < off[d + 1]; ++k) {
    const u32 rec = src[k];
    const u32 s = rec & 0x3fffffffu, corner = rec >> 30;
    const double e = val[s];
    if (!(e == e)) continue;  // the source pixel has no event before T
    const float2 uv = lut[s];
    const double u = (double)uv.x, v = (double)uv.y;
    const double fu = u - floor(u), fv = v - floor(v), fu1 = 1.0 - fu, fv1 = 1.0 - fv;
    const double w = corner == 0 ? fu1 * fv1 : (corner == 1 ? fu * fv1 : (corner == 2 ? fu1 * fv : fu * fv));
    acc += w * e;
    if (acc > 1) acc = 1;
  }
  const double g = ignore_polarity ? 255.0 * acc : 255.0 * (acc + 1.0) / 2.0;
  int q = (int)rint(g);
  q = q < 0 ? 0 : (q > 255 ? 255 : q);
  raw[d] = (uint8_t)q;
}
void launch_ts_render_forward(const u64* d_sae, const u32* d_off, const u32* d_src, const float2* d_lut, double* d_val,
                              uint8_t* d_raw, uint8_t* d_out, int W, int H, u64 t_ns, double decay_sec, int ignore_polarity,
                              int median_k, hipStream_t s) {
  const int n = W * H;
  hipLaunchKernelGGL(ts_decay_f64_kernel, dim3((n + 255) / 256), dim3(256), 0, s, d_sae, d_val, n, t_ns, decay_sec, ignore_polarity);
  hipLaunchKernelGGL(ts_forward_gather_kernel, dim3((n + 255) / 256), dim3(256), 0, s, d_off, d_src, d_lut, d_val, d_raw, n,
                     ignore_polarity);
  // the 3x3 median of the BACKWARD path without its remap (fixmap == null)
  if (median_k > 1)
    hipLaunchKernelGGL(ts_median_remap_kernel<true>, dim3((W + 63) / 64, (H + 3) / 4), dim3(256), 0, s, d_raw, (const int2*)nullptr, d_out,
                       W, H, median_k);
  else
    hipLaunchKernelGGL(ts_median_remap_kernel<false>, dim3((W + 63) / 64, (H + 3) / 4), dim3(256), 0, s, d_raw, (const int2*)nullptr, d_out,
                       W, H, median_k);
}

// ---- 5x5 Gaussian, [1 4 6 4 1]^2 / 256, BORDER_REFLECT_101, round-to-nearest once ------------------
__device__ inline int reflect101(int p, int n) {
  if (n == 1) return 0;
  while (p < 0 || p >= n) {
    if (p < 0) p = -p;
    if (p >= n) p = 2 * (n - 1) - p;
  }
  return p;
}
__device__ inline void gaussian5_px(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int W, int H, int row0, int row1) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = row0 + blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= W || y >= row1) return;
  const int k[5] = {1, 4, 6, 4, 1};
  int acc = 0;
#pragma unroll
  for (int dy = -2; dy <= 2; ++dy) {
    const uint8_t* row = in + reflect101(y + dy, H) * W;
    int r = 0;
#pragma unroll
    for (int dx = -2; dx <= 2; ++dx) r += k[dx + 2] * row[reflect101(x + dx, W)];
    acc += k[dy + 2] * r;
  }
  int v = (acc + 128) >> 8;
  out[y * W + x] = (uint8_t)(v > 255 ? 255 : v);
}
__global__ void __launch_bounds__(256) gaussian5_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int W, int H,
                                                        int row0, int row1) {
  gaussian5_px(in, out, W, H, row0, row1);
}
__global__ void __launch_bounds__(256) gaussian5_pair_kernel(const uint8_t* __restrict__ in0, const uint8_t* __restrict__ in1,
                                                             uint8_t* __restrict__ out0, uint8_t* __restrict__ out1, int W, int H,
                                                             int row0, int row1) {
  gaussian5_px(blockIdx.z ? in1 : in0, blockIdx.z ? out1 : out0, W, H, row0, row1);
}
// rows [row0, row1) of the blurred image (row1 < 0: all); they read rows row0 - 2 .. row1 + 1 of the input (reflected at the
// image border only)
void launch_gaussian5(const uint8_t* d_in, uint8_t* d_out, int W, int H, hipStream_t s, int row0, int row1) {
  if (row1 < 0 || row1 > H) row1 = H;
  if (row0 < 0) row0 = 0;
  if (row1 <= row0) return;
  hipLaunchKernelGGL(gaussian5_kernel, dim3((W + 63) / 64, (row1 - row0 + 3) / 4), dim3(256), 0, s, d_in, d_out, W, H, row0, row1);
}
void launch_gaussian5_pair(const uint8_t* in0, const uint8_t* in1, uint8_t* out0, uint8_t* out1, int W, int H, hipStream_t s, int row0,
                           int row1) {
  if (row1 < 0 || row1 > H) row1 = H;
  if (row0 < 0) row0 = 0;
  if (row1 <= row0) return;
  hipLaunchKernelGGL(gaussian5_pair_kernel, dim3((W + 63) / 64, (row1 - row0 + 3) / 4, 2), dim3(256), 0, s, in0, in1, out0, out1, W, H,
                     row0, row1);
}


// ---- event denoising (Denoising: True; rpg / hkust configs) ----------------------------------------
// createDenoisingMask (esvo_Mapping.cpp:1046-1054, Visualization.cpp:96-104): binary event map of the
// selected events (indexed by the RAW pixel, Appendix A-15) -> medianBlur(3); extractDenoisedEvents
// (:1056-1072) keeps the events whose pixel is 255 in the mask, in order.  The 3x3 median of a 0/255
// image with BORDER_REPLICATE is 255 iff at least 5 of the 9 (replicated) taps are set.
__global__ void __launch_bounds__(256) denoise_mark_kernel(const uint4* __restrict__ ring, u64 first, u64 cap, u32 n,
                                                           uint8_t* __restrict__ evmap, int W, int H) {
  const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const uint4 e = ring[(first - k) % cap];
  const u32 x = e.x & 0xffffu, y = e.x >> 16;
  if (x < (u32)W && y < (u32)H) evmap[y * W + x] = 255;
}
__global__ void __launch_bounds__(256) denoise_flag_kernel(const uint4* __restrict__ ring, u64 first, u64 cap, u32 n,
                                                           const uint8_t* __restrict__ evmap, u32* __restrict__ flags, int W, int H) {
  const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const uint4 e = ring[(first - k) % cap];
  const int x = e.x & 0xffffu, y = e.x >> 16;
  u32 keep = 0;
  if (x < W && y < H) {
    int cnt = 0;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) {
        const int yy = min(max(y + dy, 0), H - 1), xx = min(max(x + dx, 0), W - 1);
        cnt += evmap[yy * W + xx] != 0;
      }
    keep = cnt >= 5;
  }
  flags[k] = keep;
}
__global__ void __launch_bounds__(256) denoise_select_kernel(const u32* __restrict__ flags, const u32* __restrict__ prefix, u32 n,
                                                             u32* __restrict__ sel) {
  const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n && flags[k]) sel[prefix[k]] = k;
}
void launch_denoise_flags(const esvo_event_t* ring, u64 first, u64 cap, u32 n, uint8_t* evmap, u32* flags, int W, int H,
                          hipStream_t s) {
  hipMemsetAsync(evmap, 0, (size_t)W * H, s);
  if (n == 0) return;
  const uint4* r = reinterpret_cast<const uint4*>(ring);
  hipLaunchKernelGGL(denoise_mark_kernel, dim3((n + 255) / 256), dim3(256), 0, s, r, first, cap, n, evmap, W, H);
  hipLaunchKernelGGL(denoise_flag_kernel, dim3((n + 255) / 256), dim3(256), 0, s, r, first, cap, n, evmap, flags, W, H);
}
void launch_denoise_select(const u32* flags, const u32* prefix, u32 n, u32* sel, hipStream_t s) {
  if (n == 0) return;
  hipLaunchKernelGGL(denoise_select_kernel, dim3((n + 255) / 256), dim3(256), 0, s, flags, prefix, n, sel);
}

// Routed band mode (the rank's ring holds its rows' events, each with its index in the global sequence): mark the selected events
// of the ring in the event map, then the flag of every selected event whose RAW row lies in [band_y0, band_y1) as bit k of `bits`
// (k = its walk position in the global selection; zeroed by the caller).
__global__ void __launch_bounds__(256) denoise_mark_routed_kernel(const uint4* __restrict__ ring, u64 first, u64 cap, u32 n_loc,
                                                                  const u32* __restrict__ gidx, u32 g_first, u32 n, uint8_t* __restrict__ evmap,
                                                                  int W, int H) {
  const u32 w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n_loc) return;
  const u64 ei = (first - w) % cap;
  if (g_first - gidx[ei] >= n) return;
  const uint4 e = ring[ei];
  const u32 x = e.x & 0xffffu, y = e.x >> 16;
  if (x < (u32)W && y < (u32)H) evmap[y * W + x] = 255;
}
__global__ void __launch_bounds__(256) denoise_bits_routed_kernel(const uint4* __restrict__ ring, u64 first, u64 cap, u32 n_loc,
                                                                  const u32* __restrict__ gidx, u32 g_first, u32 n,
                                                                  const uint8_t* __restrict__ evmap, int W, int H, int band_y0, int band_y1,
                                                                  u32* __restrict__ bits) {
  const u32 w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n_loc) return;
  const u64 ei = (first - w) % cap;
  const u32 k = g_first - gidx[ei];
  if (k >= n) return;
  const uint4 e = ring[ei];
  const int x = e.x & 0xffffu, y = e.x >> 16;
  if (!(x < W && y < H) || y < band_y0 || y >= band_y1) return;  // (an event outside the sensor is never kept: denoise_flag_kernel)
  int cnt = 0;
#pragma unroll
  for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx) {
      const int yy = min(max(y + dy, 0), H - 1), xx = min(max(x + dx, 0), W - 1);
      cnt += evmap[yy * W + xx] != 0;
    }
  if (cnt >= 5) atomicOr(&bits[k >> 5], 1u << (k & 31u));
}
void launch_denoise_bits_routed(const esvo_event_t* ring, u64 first, u64 cap, u32 n_loc, const u32* gidx, u32 g_first, u32 n, uint8_t* evmap,
                                int W, int H, int band_y0, int band_y1, u32* bits, hipStream_t s) {
  hipMemsetAsync(evmap, 0, (size_t)W * H, s);
  if (n_loc == 0) return;
  const uint4* r = reinterpret_cast<const uint4*>(ring);
  hipLaunchKernelGGL(denoise_mark_routed_kernel, dim3((n_loc + 255) / 256), dim3(256), 0, s, r, first, cap, n_loc, gidx, g_first, n, evmap, W, H);
  hipLaunchKernelGGL(denoise_bits_routed_kernel, dim3((n_loc + 255) / 256), dim3(256), 0, s, r, first, cap, n_loc, gidx, g_first, n, evmap, W, H,
                     band_y0, band_y1, bits);
}
// the gathered blocks [N][block_words] -> one flag per walk position (every position has at most one rank that set its bit)
__global__ void __launch_bounds__(256) denoise_bits_unpack_kernel(const u32* __restrict__ blocks, u32 block_words, u32 N, u32 n, u32* __restrict__ flags) {
  const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  u32 word = 0;
  for (u32 r = 0; r < N; ++r) word |= blocks[(size_t)r * block_words + (k >> 5)];
  flags[k] = (word >> (k & 31u)) & 1u;
}
void launch_denoise_bits_unpack(const u32* blocks, u32 block_words, u32 N, u32 n, u32* flags, hipStream_t s) {
  if (n == 0) return;
  hipLaunchKernelGGL(denoise_bits_unpack_kernel, dim3((n + 255) / 256), dim3(256), 0, s, blocks, block_words, N, n, flags);
}

// ---- wire ingest: serialised dvs_msgs/Event records (13 B: u16 x, u16 y, u32 sec, u32 nsec, u8 polarity; ROS1
// little-endian, no padding) -> esvo_event_t records in the event ring --------------------
__global__ void __launch_bounds__(256) ts_unpack_wire_kernel(const uint8_t* __restrict__ wire, size_t n, esvo_event_t* __restrict__ ring,
                                                             u64 first_slot, u64 ring_cap) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint8_t* r = wire + i * 13;
  esvo_event_t e;
  e.x = (uint16_t)(r[0] | (r[1] << 8));
  e.y = (uint16_t)(r[2] | (r[3] << 8));
  e.sec = (u32)r[4] | ((u32)r[5] << 8) | ((u32)r[6] << 16) | ((u32)r[7] << 24);
  e.nsec = (u32)r[8] | ((u32)r[9] << 8) | ((u32)r[10] << 16) | ((u32)r[11] << 24);
  e.polarity = r[12] ? 1 : 0;  // (bit 7 of the byte is the library's: EV_LATE)
  e._pad[0] = e._pad[1] = e._pad[2] = 0;
  ring[(first_slot + i) % ring_cap] = e;
}
void launch_ts_unpack_wire(const uint8_t* wire, size_t n, esvo_event_t* ring, u64 first_slot, u64 ring_cap, hipStream_t s) {
  if (n == 0) return;
  hipLaunchKernelGGL(ts_unpack_wire_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, wire, n, ring, first_slot, ring_cap);
}

}  // namespace esvo
