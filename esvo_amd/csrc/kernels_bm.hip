// kernels_bm.hip — K3: event block matching on the stereo Time Surfaces (gfx950).
//
// Replaces EventBM::match_an_event + epipolarSearching + zncc_cost
// (esvo_core/src/core/EventBM.cpp:80-226,317-333; tools::normalizePatch utils.h:74-92).
//
// Work decomposition: one wave64 per event, 4 events per 256-thread workgroup.  The wave stages
// the event's left patch (wx*wy bytes) and the right epipolar strip (wy rows x (wx + Nd - 1)
// columns, u8) in LDS; lane l then owns disparity candidate dmin + l (+64k) and accumulates the
// integer moments Sr, Srr, Slr of its window (the left moments are wave-reduced once).  Time
// Surface values are integers 0..255, so the ZNCC cost follows exactly from integer moments:
//   cost = 0.5 * (1 - (Slr - Sl*Sr/N) / ((sig_l + 1e-6)(sig_r + 1e-6)) / N)
// evaluated in f64 with the same expression sequence as the oracle's zncc_cost_int.  The
// argmin over candidates is a wave butterfly with the reference's tie rule (`cost <= min_cost`
// while scanning increasing disparity, EventBM.cpp:198: the largest disparity among equal
// minima wins).  Results land in slot w of the reference's thread-stride order
// (EventBM.cpp:289-308), so a stable compaction reproduces vEMP.
#include "common.hpp"

namespace esvo {

__device__ inline int wave_sum_i32(int v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, ESVO_WAVE);
  return v;
}
__device__ inline long long wave_sum_i64(long long v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, ESVO_WAVE);
  return v;
}

__device__ inline double zncc_from_moments(long long Sl, long long Sll, long long Sr, long long Srr, long long Slr,
                                           int N) {
  const double n = (double)N;
  const double varl = (double)((long long)N * Sll - Sl * Sl) / (n * n);
  const double varr = (double)((long long)N * Srr - Sr * Sr) / (n * n);
  const double sigl = sqrt(varl) + 1e-6, sigr = sqrt(varr) + 1e-6;
  const double cov = (double)((long long)N * Slr - Sl * Sr) / n;
  return 0.5 * (1 - cov / (sigl * sigr) / n);
}

extern __shared__ __attribute__((aligned(16))) unsigned char bm_smem[];

__global__ void __launch_bounds__(256) bm_match_kernel(BmArgs a, DevParams p, int lds_per_wave, int left_bytes) {
  const int wave_in_block = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const u32 w = blockIdx.x * 4 + wave_in_block;  // slot in thread-stride order
  unsigned char* ldsL = bm_smem + wave_in_block * lds_per_wave;
  unsigned char* ldsR = ldsL + left_bytes;

  const int W = p.W, H = p.H, wx = p.wx, wy = p.wy, N = wx * wy;
  const int hx = (wx - 1) / 2, hy = (wy - 1) / 2;
  const int Ws = (p.dmax - p.dmin) + wx;

  bool ok = w < a.n;
  u32 k = 0;
  uint4 e = make_uint4(0, 0, 0, 0);
  if (ok) {
    k = stride_item(w, a.n, (u32)p.num_threads);
    const u64 ei = (a.ev_reverse ? (a.ev_first - k) : (a.ev_first + k)) % a.ev_cap;
    e = reinterpret_cast<const uint4*>(a.ev)[ei];
  }
  const int ex = e.x & 0xffffu, ey = e.x >> 16;
  double xr = 0, yr = 0;
  int x1 = 0, y1 = 0;
  if (ok) ok = ex < W && ey < H;
  if (ok) {
    const float2 l = a.lut[ey * W + ex];  // getRectifiedUndistortedCoordinate, EventBM.cpp:88
    xr = (double)l.x;
    yr = (double)l.y;
    ok = !(xr < 0 || xr > (double)(W - 1) || yr < 0 || yr > (double)(H - 1));  // :90-92
  }
  if (ok && a.mask) ok = a.mask[(int)yr * W + (int)xr] > 125;  // :94 (index truncation)
  if (ok) {
    x1 = (int)floor(xr);
    y1 = (int)floor(yr);
    ok = (y1 >= p.band_y0 && y1 < p.band_y1);  // row-band sharding
  }
  if (ok) {  // isValidPatch, EventBM.cpp:251-267
    ok = !(x1 - hx < 1 || y1 - hy < 1 || x1 + hx >= W - 1 || y1 + hy >= H - 1);
  }

  // ---- stage the left patch, left moments, low-texture test (:101-109) ----
  long long Sl = 0, Sll = 0;
  if (ok) {
    int cnt = 0, sl = 0;
    long long sll = 0;
    for (int i = lane; i < N; i += ESVO_WAVE) {
      const int py = i / wx, px = i - py * wx;
      const int v = a.tsL[(y1 - hy + py) * W + (x1 - hx + px)];
      ldsL[i] = (unsigned char)v;
      cnt += (v < 1);
      sl += v;
      sll += v * v;
    }
    cnt = wave_sum_i32(cnt);
    Sl = wave_sum_i32(sl);
    Sll = wave_sum_i64(sll);
    if ((double)cnt > 0.95 * (double)N) ok = false;
  }
  // ---- stage the right strip ----
  const int xs0 = x1 - p.dmax - hx;
  if (ok) {
    for (int i = lane; i < wy * Ws; i += ESVO_WAVE) {
      const int ry = i / Ws, rx = i - ry * Ws;
      const int gx = xs0 + rx, gy = y1 - hy + ry;
      ldsR[i] = (gx >= 0 && gx < W) ? a.tsR[gy * W + gx] : (unsigned char)0;
    }
  }
  __syncthreads();  // all four waves reach this (no early exits); makes the LDS tiles visible

  double best = 1.0;  // ZNCC_MAX_
  int bestd = -1;
  if (ok) {
    for (int d = p.dmin + lane; d <= p.dmax; d += ESVO_WAVE) {
      const int x2 = x1 - d;
      if (x2 - hx < 1 || x2 + hx >= W - 1) continue;  // invalid candidates never update (:186-190)
      const int col0 = p.dmax - d;
      int sr = 0, slr = 0;
      long long srr = 0;
      for (int py = 0; py < wy; ++py) {
        const unsigned char* rrow = ldsR + py * Ws + col0;
        const unsigned char* lrow = ldsL + py * wx;
        int rr = 0;
        for (int px = 0; px < wx; ++px) {
          const int r = rrow[px], l = lrow[px];
          sr += r;
          rr += r * r;
          slr += l * r;
        }
        srr += rr;
      }
      const double cost = zncc_from_moments(Sl, Sll, sr, srr, slr, N);
      if (cost <= best) { best = cost; bestd = d; }  // :198 (lane scans increasing d)
    }
    // wave argmin; ties -> larger disparity
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
      const double oc = __shfl_xor(best, s, ESVO_WAVE);
      const int od = __shfl_xor(bestd, s, ESVO_WAVE);
      if (od >= 0 && (bestd < 0 || oc < best || (oc == best && od > bestd))) { best = oc; bestd = od; }
    }
    ok = bestd >= 0 && best < p.zncc_thr;  // :222 (fine search re-evaluates the same candidate)
  }
  u32 pose_idx = 0;
  if (ok) {  // StampTransformationMap_lower_bound, utils.h:66-71
    const double te = time_to_sec(e.y, e.z);
    u32 lo = 0, hi = a.n_pose;
    while (lo < hi) {
      const u32 mid = (lo + hi) >> 1;
      if (a.pose_sec[mid] < te) lo = mid + 1; else hi = mid;
    }
    pose_idx = lo;
    ok = lo < a.n_pose;  // EventBM.cpp:155-156
  }
  if (w < a.n && lane == 0) {
    a.out_flags[w] = ok ? 1u : 0u;
    if (ok) {
      esvo_match_t m;
      const double disparity = (double)bestd;         // x1(0) - bestMatch(0), :151
      const double depth = p.baseline_f / disparity;  // :152
      m.x_left[0] = xr;
      m.x_left[1] = yr;
      m.inv_depth = 1.0 / depth;  // :158
      m.cost = best;
      m.disp = disparity;
      m.event_idx = k;
      m.pose_idx = pose_idx;
      a.out_slots[w] = m;
    }
  }
}

void launch_bm_match(const BmArgs& a, const DevParams& p, hipStream_t s) {
  if (a.n == 0) return;
  const int left_bytes = ((p.wx * p.wy + 15) / 16) * 16;
  const int Ws = (p.dmax - p.dmin) + p.wx;
  const int right_bytes = ((p.wy * Ws + 15) / 16) * 16;
  const int per_wave = left_bytes + right_bytes;
  const u32 blocks = (a.n + 3) / 4;
  hipLaunchKernelGGL(bm_match_kernel, dim3(blocks), dim3(256), (size_t)per_wave * 4, s, a, p, per_wave, left_bytes);
}

// stable compaction: slot w -> position prefix[w]
__global__ void __launch_bounds__(256) compact_matches_kernel(const esvo_match_t* __restrict__ slots, const u32* __restrict__ flags,
                                                              const u32* __restrict__ prefix, u32 n,
                                                              esvo_match_t* __restrict__ out) {
  const u32 w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n || !flags[w]) return;
  out[prefix[w]] = slots[w];
}
void launch_compact_matches(const esvo_match_t* slots, const u32* flags, const u32* prefix, u32 n, esvo_match_t* out,
                            hipStream_t s) {
  if (n == 0) return;
  hipLaunchKernelGGL(compact_matches_kernel, dim3((n + 255) / 256), dim3(256), 0, s, slots, flags, prefix, n, out);
}

}  // namespace esvo
