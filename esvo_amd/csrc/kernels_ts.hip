// kernels_ts.hip — Time-Surface raster on gfx950.
//
//   K1 ts_scatter : esvo_time_surface eventsCallback / EventQueueMat::insertEvent
//                   (esvo_time_surface/src/TimeSurface.cpp:403-425, TimeSurface.h:39-50)
//   K2 ts_decay + ts_median_remap : TimeSurface::createTimeSurfaceAtTime, BACKWARD mode
//                   (TimeSurface.cpp:52-152): exp decay -> x255 -> u8 -> 3x3 median -> rectifying remap
//   ts_decay_f64 + ts_forward_gather : the same function in FORWARD mode (TimeSurface.cpp:85-116)
//   gaussian5     : TimeSurfaceObservation::GaussianBlurTS(5) (TimeSurfaceObservation.h:107-116)
//
// Data layout: the per-pixel event queues of the reference (std::deque, length 20) collapse to a
// Surface of Active Events: one u64 per pixel holding (t_ns << 1 | polarity) of the newest
// event, updated with atomicMax.  The host only scatters events with ts < T before rendering at
// T, which is exactly what getMostRecentEventBeforeT (TimeSurface.h:52-75) returns.
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

// ---- K2a: decay + quantise ------------------------------------------------------------------------
// TimeSurface.cpp:65-127.  dt is formed like ros::Duration::toSec() (Appendix A-17); the u8
// conversion is cv::Mat::convertTo = saturate_cast<uchar>(cvRound(v)) = round-half-even.
__device__ inline void ts_decay_px(const u64* __restrict__ sae, uint8_t* __restrict__ raw, int n_px, u64 t_ns,
                                   double decay_sec, int ignore_polarity) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_px) return;
  const u64 key = sae[i];
  const u64 te = key >> 1;
  double v = 0.0;
  if (key != 0 && te < t_ns && ns_to_sec(te) > 0) {
    const double dt = duration_to_sec(t_ns, te);
    double e = exp(-dt / decay_sec);
    if (!ignore_polarity) e *= (key & 1ull) ? 1.0 : -1.0;
    v = e;
  }
  const double g = ignore_polarity ? 255.0 * v : 255.0 * (v + 1.0) / 2.0;
  int q = (int)rint(g);
  q = q < 0 ? 0 : (q > 255 ? 255 : q);
  raw[i] = (uint8_t)q;
}
__global__ void __launch_bounds__(256) ts_decay_kernel(const u64* __restrict__ sae, uint8_t* __restrict__ raw, int n_px,
                                                       u64 t_ns, double decay_sec, int ignore_polarity) {
  ts_decay_px(sae, raw, n_px, t_ns, decay_sec, ignore_polarity);
}
__global__ void __launch_bounds__(256) ts_decay_pair_kernel(TsPair c, int n_px, u64 t_ns, double decay_sec, int ignore_polarity) {
  ts_decay_px(c.sae[blockIdx.y], c.raw[blockIdx.y], n_px, t_ns, decay_sec, ignore_polarity);
}

// ---- K2b: 3x3 median (BORDER_REPLICATE) fused with the fixed-point bilinear remap --------------------
__device__ inline void cswap(int& a, int& b) { int lo = min(a, b), hi = max(a, b); a = lo; b = hi; }
__device__ inline int median9(int p0, int p1, int p2, int p3, int p4, int p5, int p6, int p7, int p8) {
  // 19-exchange median network
  cswap(p1, p2); cswap(p4, p5); cswap(p7, p8); cswap(p0, p1); cswap(p3, p4); cswap(p6, p7);
  cswap(p1, p2); cswap(p4, p5); cswap(p7, p8); cswap(p0, p3); cswap(p5, p8); cswap(p4, p7);
  cswap(p3, p6); cswap(p1, p4); cswap(p2, p5); cswap(p4, p7); cswap(p4, p2); cswap(p6, p4);
  cswap(p4, p2);
  return p4;
}
__device__ inline int median_tap(const uint8_t* __restrict__ raw, int W, int H, int x, int y, int median_k) {
  if (x < 0 || x >= W || y < 0 || y >= H) return 0;  // BORDER_CONSTANT 0 of cv::remap
  if (median_k <= 0) return raw[y * W + x];
  const int xm = max(x - 1, 0), xp = min(x + 1, W - 1), ym = max(y - 1, 0), yp = min(y + 1, H - 1);
  const uint8_t* r0 = raw + ym * W;
  const uint8_t* r1 = raw + y * W;
  const uint8_t* r2 = raw + yp * W;
  return median9(r0[xm], r0[x], r0[xp], r1[xm], r1[x], r1[xp], r2[xm], r2[x], r2[xp]);
}

// fixmap[i] = (cvRound(map_x*32), cvRound(map_y*32)): OpenCV's INTER_BITS=5 coordinate
// quantisation, precomputed once on the host (Appendix B.2).  Weights are the exact 15-bit
// integers (32-fx)(32-fy)*32 ...; dst = (sum + 16384) >> 15.
__device__ inline int ts_median_remap_px(const uint8_t* __restrict__ raw, const int2* __restrict__ fixmap, int W, int H,
                                         int median_k, int x, int y) {
  const int i = y * W + x;
  int v;
  if (fixmap) {
    const int2 m = fixmap[i];
    const int ix = m.x >> 5, iy = m.y >> 5, fx = m.x & 31, fy = m.y & 31;
    const int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
    int acc = 0;
    if (w00) acc += w00 * median_tap(raw, W, H, ix, iy, median_k);
    if (w01) acc += w01 * median_tap(raw, W, H, ix + 1, iy, median_k);
    if (w10) acc += w10 * median_tap(raw, W, H, ix, iy + 1, median_k);
    if (w11) acc += w11 * median_tap(raw, W, H, ix + 1, iy + 1, median_k);
    v = (acc + 16384) >> 15;
  } else {
    v = median_tap(raw, W, H, x, y, median_k);
  }
  return v;
}
__global__ void __launch_bounds__(256) ts_median_remap_kernel(const uint8_t* __restrict__ raw, const int2* __restrict__ fixmap,
                                                              uint8_t* __restrict__ out, int W, int H, int median_k) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= W || y >= H) return;
  out[y * W + x] = (uint8_t)ts_median_remap_px(raw, fixmap, W, H, median_k, x, y);
}
// both cameras (blockIdx.z); out2 (may be null): a second copy of the surface, the mapper's observation when it is not smoothed
__global__ void __launch_bounds__(256) ts_median_remap_pair_kernel(TsPair c, int W, int H, int median_k) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= W || y >= H) return;
  const int cam = blockIdx.z;
  const uint8_t v = (uint8_t)ts_median_remap_px(c.raw[cam], c.fixmap[cam], W, H, median_k, x, y);
  c.out[cam][y * W + x] = v;
  if (c.out2[cam]) c.out2[cam][y * W + x] = v;
}

void launch_ts_render(const u64* d_sae, const int2* d_fixmap, uint8_t* d_raw, uint8_t* d_out, int W, int H, u64 t_ns,
                      double decay_sec, int ignore_polarity, int median_k, hipStream_t s) {
  const int n = W * H;
  hipLaunchKernelGGL(ts_decay_kernel, dim3((n + 255) / 256), dim3(256), 0, s, d_sae, d_raw, n, t_ns, decay_sec,
                     ignore_polarity);
  hipLaunchKernelGGL(ts_median_remap_kernel, dim3((W + 63) / 64, (H + 3) / 4), dim3(256), 0, s, d_raw, d_fixmap, d_out, W,
                     H, median_k);
}

void launch_ts_render_pair(const TsPair& c, int W, int H, u64 t_ns, double decay_sec, int ignore_polarity, int median_k,
                           hipStream_t s) {
  const int n = W * H;
  hipLaunchKernelGGL(ts_decay_pair_kernel, dim3((n + 255) / 256, 2), dim3(256), 0, s, c, n, t_ns, decay_sec, ignore_polarity);
  hipLaunchKernelGGL(ts_median_remap_pair_kernel, dim3((W + 63) / 64, (H + 3) / 4, 2), dim3(256), 0, s, c, W, H, median_k);
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
  hipLaunchKernelGGL(ts_median_remap_kernel, dim3((W + 63) / 64, (H + 3) / 4), dim3(256), 0, s, d_raw, (const int2*)nullptr, d_out, W,
                     H, median_k);
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
__device__ inline void gaussian5_px(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int W, int H) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= W || y >= H) return;
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
__global__ void __launch_bounds__(256) gaussian5_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int W, int H) {
  gaussian5_px(in, out, W, H);
}
__global__ void __launch_bounds__(256) gaussian5_pair_kernel(const uint8_t* __restrict__ in0, const uint8_t* __restrict__ in1,
                                                             uint8_t* __restrict__ out0, uint8_t* __restrict__ out1, int W, int H) {
  gaussian5_px(blockIdx.z ? in1 : in0, blockIdx.z ? out1 : out0, W, H);
}
void launch_gaussian5(const uint8_t* d_in, uint8_t* d_out, int W, int H, hipStream_t s) {
  hipLaunchKernelGGL(gaussian5_kernel, dim3((W + 63) / 64, (H + 3) / 4), dim3(256), 0, s, d_in, d_out, W, H);
}
void launch_gaussian5_pair(const uint8_t* in0, const uint8_t* in1, uint8_t* out0, uint8_t* out1, int W, int H, hipStream_t s) {
  hipLaunchKernelGGL(gaussian5_pair_kernel, dim3((W + 63) / 64, (H + 3) / 4, 2), dim3(256), 0, s, in0, in1, out0, out1, W, H);
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
  e.polarity = r[12];
  e._pad[0] = e._pad[1] = e._pad[2] = 0;
  ring[(first_slot + i) % ring_cap] = e;
}
void launch_ts_unpack_wire(const uint8_t* wire, size_t n, esvo_event_t* ring, u64 first_slot, u64 ring_cap, hipStream_t s) {
  if (n == 0) return;
  hipLaunchKernelGGL(ts_unpack_wire_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, wire, n, ring, first_slot, ring_cap);
}

}  // namespace esvo
