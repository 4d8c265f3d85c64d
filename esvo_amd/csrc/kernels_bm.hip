// kernels_bm.hip — K3: event block matching on the stereo Time Surfaces (gfx950).
//
// Replaces EventBM::match_an_event + epipolarSearching + zncc_cost
// (esvo_core/src/core/EventBM.cpp:80-226,317-333; tools::normalizePatch utils.h:74-92).
//
// Work decomposition: G lanes per event (G in {8,16,32,64}, chosen so that the candidate slots
// ceil(Nd/G)*G waste the least; 256/G events per workgroup).  The group stages the event's left
// patch and the right epipolar strip (7 rows x (15 + Nd - 1) columns, u8) in LDS; lane l owns the
// disparity candidates dmin + l (+G...) and accumulates the integer moments Sr, Srr, Slr of its
// window with packed v_dot4_u32_u8 (the left moments are group-reduced once).  Time
// Surface values are integers 0..255, so the ZNCC cost follows exactly from integer moments:
//   cost = 0.5 * (1 - (Slr - Sl*Sr/N) / ((sig_l + 1e-6)(sig_r + 1e-6)) / N)
// evaluated in f64 with the same expression sequence as the oracle's zncc_cost_int.  The
// argmin over candidates is a wave butterfly with the reference's tie rule (`cost <= min_cost`
// while scanning increasing disparity, EventBM.cpp:198: the largest disparity among equal
// minima wins).  Results land in slot w of the reference's thread-stride order
// (EventBM.cpp:289-308), so a stable compaction reproduces vEMP.
#include "common.hpp"
#include "fdiv.hpp"

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

// ZNCC cost from the exact integer moments of the two 105-pixel patches (tools::zncc_cost via normalizePatch,
// utils.h:74-92 / EventBM.cpp:317-333, restated on moments -- see the oracle's integer mode).
//   N*Sxx - Sx*Sx <= 105^2 * 255^2 < 2^31 and |N*Slr - Sl*Sr| < 2^31: 32-bit integer arithmetic is exact and the
//   conversions to f64 are single instructions.
// The quotients by the constants n and n^2 and by sigl*sigr are IEEE quotients through fdiv.hpp's refined reciprocal:
// every operand is 0 or lies in [1e-12, 1e10] by the integer bounds, far inside its window; sqrt_moderate likewise
// (a zero variance is selected around it).
typedef u32 u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));

struct ZnccLeft {
  double sigl;  // sqrt(var_l) + 1e-6
  int Sl;
};
__device__ inline double sqrt_var(double v) { return v > 0 ? sqrt_moderate(v) : 0.0; }
__device__ inline ZnccLeft zncc_left(int Sl, int Sll, int N, const Recip& rn2) {
  ZnccLeft z;
  z.Sl = Sl;
  z.sigl = sqrt_var(div_fast((double)(N * Sll - Sl * Sl), rn2)) + 1e-6;
  return z;
}
__device__ inline double zncc_from_moments(const ZnccLeft& zl, int Sr, int Srr, int Slr, int N, const Recip& rn,
                                           const Recip& rn2) {
  const double varr = div_fast((double)(N * Srr - Sr * Sr), rn2);
  const double sigr = sqrt_var(varr) + 1e-6;
  const double cov = div_fast((double)(N * Slr - zl.Sl * Sr), rn);
  Recip rs;
  rs.b = zl.sigl * sigr;
  rs.y = recip_refined(rs.b);
  return 0.5 * (1 - div_fast(div_fast(cov, rs), rn));
}

extern __shared__ __attribute__((aligned(16))) unsigned char bm_smem[];

template <int G>
__device__ inline int grp_sum_i32(int v) {
#pragma unroll
  for (int d = G / 2; d >= 1; d >>= 1) v += __shfl_xor(v, d, G);
  return v;
}

// G lanes per event (64/G events per wave, 256/G per workgroup).  Inside a group, lane l owns
// the disparity candidates dmin + l, dmin + l + G, ...
//
// LDS per event: the right strip as RD aligned dwords per row (a dword-aligned copy of global
// memory: no shifting while staging) and the left patch as 4 dwords per row (15 pixels + a zero
// pad byte).  A candidate's 15-pixel window starts at an arbitrary byte offset of its row, so
// each lane reads 5 dwords per row, re-aligns them with v_alignbyte_b32 and accumulates
// S_r, S_rr and S_lr with v_dot4_u32_u8 (the pad byte of the left patch is zero and the 16th
// byte of the window is masked out).
//
// Template switches for the two EventBM options no shipped configuration sets:
//   COARSE (BM_step > 1, EventBM.cpp:118-138): the dense cost row [dmin, dmax] is formed as for step 1 -- a superset of
//     what the reference evaluates -- and kept in LDS (cost 1.0 = ZNCC_MAX_ for invalid candidates, as mDispCost holds
//     them); the coarse pass is the argmin over d = dmin + j step with the neighbour rule of :207-219 (both coarse
//     neighbours in the map and < ZNCC_MAX_), the fine pass the argmin over [best - (step-1), best + (step-1)] with the
//     carried minimum (:127-133).  Both use the reference's `cost <= min_cost` tie rule (largest disparity among equal
//     minima).  The neighbour rule implies dmin + step <= best <= dmax - step, so the fine window lies inside the row.
//   UPDOWN (BM_bUpDownConfiguration, :178-186, :148-151): candidates are x2 = (x1.x, x1.y - d); the strip is vertical,
//     7 + Nd - 1 rows of 15 pixels staged like the left patch (16-byte rows), candidate d reads rows dmax - d .. + 6.
#ifndef BM_BLOCK
#define BM_BLOCK 64   // threads per workgroup (a multiple of 64; the two barriers are per workgroup).  64 beats 256 by 20 %:
                      // barriers span one wave and a finished wave frees its slot and LDS at once
#endif

// one 15-pixel row starting at byte address A of a dword-padded image -> 4 dwords (15 pixels + a zero pad byte)
__device__ inline uint4 load_row15(const u32* ts32, int n_dw, int A) {
  const int a0 = A >> 2;  // arithmetic shift = floor
  const u32 sh = (u32)(A & 3);
  u32 d[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    int idx = a0 + j;
    idx = idx < 0 ? 0 : (idx >= n_dw ? n_dw - 1 : idx);  // only bytes that no valid patch reads can be clamped
    d[j] = ts32[idx];
  }
  uint4 wv;
  wv.x = __builtin_amdgcn_alignbyte(d[1], d[0], sh);
  wv.y = __builtin_amdgcn_alignbyte(d[2], d[1], sh);
  wv.z = __builtin_amdgcn_alignbyte(d[3], d[2], sh);
  wv.w = __builtin_amdgcn_alignbyte(d[4], d[3], sh) & 0x00ffffffu;
  return wv;
}

// Which event a group handles (k = its position in the newest-first walk of the tick's selection) and its record.
//   thread-stride slots (default; multi-GPU: dealt by slot): k = stride_item(w)
//   routed band mode (a.gidx): the rank's ring holds only the events of its image rows, each with its index in the GLOBAL event
//     sequence (low 32 bits); w is the position in the newest-first walk of the LOCAL ring and k = g_first - gidx
__device__ inline void bm_item(const BmArgs& a, const DevParams& p, u32 w, u32& k, uint4& e, bool& ok) {
  if (a.gidx) {
    const u64 ei = (a.ev_first - w) % a.ev_cap;
    e = reinterpret_cast<const uint4*>(a.ev)[ei];
    k = a.g_first - a.gidx[ei];
    if (a.keep_flags) {  // Denoising: position in the KEPT sequence (the raw position only selects)
      ok = k < a.n_raw && a.keep_flags[k] != 0u;
      if (ok) k = a.keep_prefix[k];
      return;
    }
    ok = k < a.n;
    return;
  }
  k = stride_item(w, a.n, (u32)p.num_threads);
  const u32 kk = a.sel ? a.sel[k] : k;
  const u64 ei = (a.ev_reverse ? (a.ev_first - kk) : (a.ev_first + kk)) % a.ev_cap;
  e = reinterpret_cast<const uint4*>(a.ev)[ei];
}

template <int G>
__device__ inline void grp_argmin(double& best, int& bestd) {  // ties -> larger disparity (`<=` while scanning upwards)
#pragma unroll
  for (int s = G / 2; s >= 1; s >>= 1) {
    const double oc = __shfl_xor(best, s, G);
    const int od = __shfl_xor(bestd, s, G);
    if (od >= 0 && (bestd < 0 || oc < best || (oc == best && od > bestd))) { best = oc; bestd = od; }
  }
}

template <int G, bool COARSE, bool UPDOWN>
__global__ void __launch_bounds__(BM_BLOCK) bm_match_kernel(BmArgs a, DevParams p, int RD, int lds_per_event) {
  constexpr int EPB = BM_BLOCK / G;  // events per block
  const int grp = threadIdx.x / G, l = threadIdx.x % G;
  // slot in thread-stride order; multi-GPU: slots are dealt round-robin and a rank's launch covers its own ones densely --
  // or (routed band mode, bm_item) position w of the walk over the rank's own ring, results indexed by that position
  const u32 pos = blockIdx.x * EPB + grp;
  const u32 w = a.gidx ? pos : pos * (u32)p.ev_nshards + (u32)p.ev_shard;
  const u32 n_out = a.gidx ? a.n_loc : a.n;
  u32* ldsL = reinterpret_cast<u32*>(bm_smem + grp * lds_per_event);  // [7][4] dwords
  u32* ldsR = ldsL + 28;                                              // [7][RD] dwords / UPDOWN: [Nd + 6][4] dwords
  const int nd = p.dmax - p.dmin + 1;
  double* cost_row = reinterpret_cast<double*>(ldsR + (UPDOWN ? (nd + 6) * 4 : 7 * RD));  // COARSE only: [Nd]

  const int W = p.W, H = p.H;
  constexpr int wx = 15, wy = 7, N = wx * wy, hx = 7, hy = 3;

  bool ok = w < n_out;
  u32 k = 0;
  uint4 e = make_uint4(0, 0, 0, 0);
  if (ok) bm_item(a, p, w, k, e, ok);
  const int ex = e.x & 0xffffu, ey = e.x >> 16;
  double xr = 0, yr = 0;
  int x1 = 0, y1 = 0;
  if (ok) ok = ex < W && ey < H;
  if (ok) {
    const float2 q = a.lut[ey * W + ex];  // getRectifiedUndistortedCoordinate, EventBM.cpp:88
    xr = (double)q.x;
    yr = (double)q.y;
    ok = !(xr < 0 || xr > (double)(W - 1) || yr < 0 || yr > (double)(H - 1));  // :90-92
    // routed band mode: the event belongs to the rank that owns floor(y_rect); the ring also holds the
    // events of the Time Surface's source rows around the band, which other ranks match
    if (a.gidx && ok) { const int yb = (int)floor(yr); ok = yb >= p.band_y0 && yb < p.band_y1; }
  }
  if (ok && a.mask) ok = a.mask[(int)yr * W + (int)xr] > 125;  // :94 (index truncation)
  if (ok) {
    x1 = (int)floor(xr);
    y1 = (int)floor(yr);
  }
  if (ok) ok = !(x1 - hx < 1 || y1 - hy < 1 || x1 + hx >= W - 1 || y1 + hy >= H - 1);  // isValidPatch, :251-267
  int reason = 0;  // why the event failed, as the reference counts it (EventBM.h:89): 1 info-noise ratio, 2 coarse, 3 fine

  // ---- stage the left patch, left moments, low-texture test (:101-109) ----
  // Lane py < 7 owns patch row py: five aligned dwords straight from the image, re-aligned to the row's first byte with
  // v_alignbyte (as the candidates do below), written as one 16-byte row (15 pixels + a zero pad byte).  Sum, sum of
  // squares and the number of zero pixels come from the packed words (dot4, and the exact zero-byte mask).
  int Sl = 0, Sll = 0;
  if (ok) {
    int cnt = 0, sl = 0, sll = 0;
    if (l < wy) {
      const u32* ts32 = reinterpret_cast<const u32*>(a.tsL);
      const int n_dw = (W * H + 3) >> 2;
      const uint4 wv = load_row15(ts32, n_dw, (y1 - hy + l) * W + (x1 - hx));
      *reinterpret_cast<uint4*>(ldsL + l * 4) = wv;
      u32 usl = 0, usll = 0;
      usl = __builtin_amdgcn_udot4(wv.x, 0x01010101u, usl, false);
      usl = __builtin_amdgcn_udot4(wv.y, 0x01010101u, usl, false);
      usl = __builtin_amdgcn_udot4(wv.z, 0x01010101u, usl, false);
      usl = __builtin_amdgcn_udot4(wv.w, 0x01010101u, usl, false);
      usll = __builtin_amdgcn_udot4(wv.x, wv.x, usll, false);
      usll = __builtin_amdgcn_udot4(wv.y, wv.y, usll, false);
      usll = __builtin_amdgcn_udot4(wv.z, wv.z, usll, false);
      usll = __builtin_amdgcn_udot4(wv.w, wv.w, usll, false);
      auto zero_bytes = [](u32 x) {  // bit 7 of every byte that is 0, exactly
        return __popc(~(((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x | 0x7f7f7f7fu));
      };
      cnt = zero_bytes(wv.x) + zero_bytes(wv.y) + zero_bytes(wv.z) + zero_bytes(wv.w | 0xff000000u);
      sl = (int)usl;
      sll = (int)usll;
    }
    cnt = grp_sum_i32<G>(cnt);
    Sl = grp_sum_i32<G>(sl);
    Sll = grp_sum_i32<G>(sll);  // <= 105 * 255^2 < 2^31
    if ((double)cnt > 0.95 * (double)N) { ok = false; reason = 1; }  // infoNoiseRatioLowNum_++, :105-109
  }
  // ---- stage the right strip ----
  const int xs0 = x1 - p.dmax - hx;
  int o_row[wy];  // byte offset of column xs0 inside the first staged dword of each row
  if (ok) {
    const u32* ts32 = reinterpret_cast<const u32*>(a.tsR);
    const int n_dw = (W * H + 3) >> 2;
    if constexpr (UPDOWN) {  // rows y1 - dmax - hy ... y1 - dmin + hy, 15 pixels each, one 16-byte row per lane and trip
      for (int r = l; r < nd + 6; r += G)
        *reinterpret_cast<uint4*>(ldsR + r * 4) = load_row15(ts32, n_dw, (y1 - p.dmax - hy + r) * W + (x1 - hx));
    } else {  // aligned 16-byte pieces straight from the (dword-padded) image; RD is a multiple of 4
#pragma unroll
      for (int py = 0; py < wy; ++py) {
        const int A = (y1 - hy + py) * W + xs0;
        const int a0 = A >> 2;  // arithmetic shift = floor
        o_row[py] = A - (a0 << 2);
        for (int j4 = l; j4 < (RD >> 2); j4 += G) {
          const int idx = a0 + 4 * j4;
          uint4 v;
          if (__builtin_expect(idx >= 0 && idx + 3 < n_dw, 1)) {
            const u32x4_a4 q = *reinterpret_cast<const u32x4_a4*>(ts32 + idx);  // dword-aligned 16-byte load
            v = make_uint4(q.x, q.y, q.z, q.w);
          } else {  // only bytes of invalid candidates can be clamped
            u32 t[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              int i = idx + q;
              i = i < 0 ? 0 : (i >= n_dw ? n_dw - 1 : i);
              t[q] = ts32[i];
            }
            v = make_uint4(t[0], t[1], t[2], t[3]);
          }
          *reinterpret_cast<uint4*>(ldsR + py * RD + 4 * j4) = v;
        }
      }
    }
  }
  __syncthreads();

  // isValidPatch of candidate d (:184-190): only the coordinate that moves can leave the image
  auto cand_valid = [&](int d) {
    if constexpr (UPDOWN) { const int y2 = y1 - d; return !(y2 - hy < 1 || y2 + hy >= H - 1); }
    else { const int x2 = x1 - d; return !(x2 - hx < 1 || x2 + hx >= W - 1); }
  };
  double best = 1.0;  // ZNCC_MAX_
  int bestd = -1;
  if (ok) {
    const Recip rn = make_recip((double)N), rn2 = make_recip((double)N * (double)N);
    const ZnccLeft zl = zncc_left(Sl, Sll, N, rn2);
    u32 L[wy][4];
#pragma unroll
    for (int py = 0; py < wy; ++py) {
      const uint4 v = *reinterpret_cast<const uint4*>(ldsL + py * 4);
      L[py][0] = v.x; L[py][1] = v.y; L[py][2] = v.z; L[py][3] = v.w;
    }
    for (int d = p.dmin + l; d <= p.dmax; d += G) {
      if (!cand_valid(d)) {  // invalid candidates never update (:186-190); the map holds ZNCC_MAX_ for them
        if constexpr (COARSE) cost_row[d - p.dmin] = 1.0;
        continue;
      }
      const int col0 = p.dmax - d;
      u32 sr = 0, srr = 0, slr = 0;
#pragma unroll
      for (int py = 0; py < wy; ++py) {
        u32 w0, w1, w2, w3;
        if constexpr (UPDOWN) {
          const uint4 v = *reinterpret_cast<const uint4*>(ldsR + (col0 + py) * 4);
          w0 = v.x; w1 = v.y; w2 = v.z; w3 = v.w;
        } else {
          const int b = o_row[py] + col0;
          const u32* row = ldsR + py * RD + (b >> 2);
          const u32 sh = (u32)(b & 3);
          const u32 d0 = row[0], d1 = row[1], d2 = row[2], d3 = row[3], d4 = row[4];
          w0 = __builtin_amdgcn_alignbyte(d1, d0, sh);
          w1 = __builtin_amdgcn_alignbyte(d2, d1, sh);
          w2 = __builtin_amdgcn_alignbyte(d3, d2, sh);
          w3 = __builtin_amdgcn_alignbyte(d4, d3, sh) & 0x00ffffffu;
        }
        slr = __builtin_amdgcn_udot4(L[py][0], w0, slr, false);
        slr = __builtin_amdgcn_udot4(L[py][1], w1, slr, false);
        slr = __builtin_amdgcn_udot4(L[py][2], w2, slr, false);
        slr = __builtin_amdgcn_udot4(L[py][3], w3, slr, false);
        sr = __builtin_amdgcn_udot4(w0, 0x01010101u, sr, false);
        sr = __builtin_amdgcn_udot4(w1, 0x01010101u, sr, false);
        sr = __builtin_amdgcn_udot4(w2, 0x01010101u, sr, false);
        sr = __builtin_amdgcn_udot4(w3, 0x01010101u, sr, false);
        srr = __builtin_amdgcn_udot4(w0, w0, srr, false);
        srr = __builtin_amdgcn_udot4(w1, w1, srr, false);
        srr = __builtin_amdgcn_udot4(w2, w2, srr, false);
        srr = __builtin_amdgcn_udot4(w3, w3, srr, false);
      }
      const double cost = zncc_from_moments(zl, (int)sr, (int)srr, (int)slr, N, rn, rn2);
      if constexpr (COARSE) cost_row[d - p.dmin] = cost;
      else if (cost <= best) { best = cost; bestd = d; }  // :198 (lane scans increasing d)
    }
  }
  if constexpr (COARSE) {
    __syncthreads();  // the cost row is complete
    if (ok) {
      const int step = p.step;
      const int nc = (p.dmax - p.dmin) / step + 1;  // coarse candidates dmin + j step (:119-121)
      for (int j = l; j < nc; j += G) {
        const int d = p.dmin + j * step;
        if (!cand_valid(d)) continue;
        const double c = cost_row[d - p.dmin];
        if (c <= best) { best = c; bestd = d; }
      }
      grp_argmin<G>(best, bestd);
      // :207-219: both coarse neighbours are in the map (size_t arithmetic: best - step must not wrap) and below ZNCC_MAX_
      bool found = bestd >= 0 && bestd - step >= p.dmin && bestd + step <= p.dmin + (nc - 1) * step;
      if (found) found = cost_row[bestd - step - p.dmin] < 1.0 && cost_row[bestd + step - p.dmin] < 1.0 && best < p.zncc_thr;
      if (!found) { ok = false; reason = 2; }  // coarseSearchingFailNum_++, :122-126
      else {  // fine pass over [best - (step - 1), best + (step - 1)] with the carried minimum (:127-133)
        const int cd = bestd;
        double fb = 2.0;
        int fd = -1;
        for (int d = cd - (step - 1) + l; d <= cd + (step - 1); d += G) {
          if (!cand_valid(d)) continue;
          const double c = cost_row[d - p.dmin];
          if (c <= fb) { fb = c; fd = d; }
        }
        grp_argmin<G>(fb, fd);
        // the window contains cd itself (valid, cost == best), so its minimum is <= best and the scan's last `<=` hit wins
        best = fb;
        bestd = fd;
        if (!(best < p.zncc_thr)) { ok = false; reason = 3; }  // fineSearchingFailNum_++ (cannot happen: the minimum only falls)
      }
    }
  } else if (ok) {
    grp_argmin<G>(best, bestd);  // group argmin; ties -> larger disparity
    // :222 (the fine search re-evaluates the same candidate); a failure here is the COARSE search's in the reference's count
    if (!(bestd >= 0 && best < p.zncc_thr)) { ok = false; reason = 2; }
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
  if (w < n_out && l == 0) {
    a.out_flags[w] = ok ? 1u : 0u;
    if (ok) {
      esvo_match_t m;
      const double disparity = (double)bestd;         // x1(0) - bestMatch(0) / x1(1) - bestMatch(1), :146-151
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
  // the reference's per-reason failure counters (EventBM.h:89, EventBM.cpp:107,124,135): one atomic per wave and reason,
  // striped over CNT_STRIPES addresses so that a launch of 10^5 waves does not queue on one L2 line
  if (a.fail_counters) {
    const bool lead = w < n_out && l == 0;
#pragma unroll
    for (int r = 1; r <= 3; ++r) {
      const int n = __popcll(__ballot(lead && reason == r));
      if (n && (threadIdx.x & 63) == 0)
        atomicAdd(a.fail_counters + CNT_BM_FAIL + (r - 1) * CNT_STRIPES + (blockIdx.x % CNT_STRIPES), (u32)n);
    }
  }
}

// ---- any patch size (patch_size_X x patch_size_Y: a run-time parameter upstream, code default 25 x 25, esvo_Mapping.cpp:38-39) ----
// The same decisions in the same order as bm_match_kernel, written for generality: one wave per event, the left patch and the
// right strip as plain bytes in LDS, every candidate's moments by byte loops, 64-bit integer moments (N Sxx reaches 2^31 from
// 183 pixels on) and plain IEEE division / square root in the oracle's expression order (zncc_cost_int).
__device__ inline double zncc_any(long long Sl, long long Sll, long long Sr, long long Srr, long long Slr, int N) {
  const double n = (double)N;
  const double varl = (double)(N * Sll - Sl * Sl) / (n * n);
  const double varr = (double)(N * Srr - Sr * Sr) / (n * n);
  const double sigl = sqrt(varl) + 1e-6, sigr = sqrt(varr) + 1e-6;
  const double cov = (double)(N * Slr - Sl * Sr) / n;
  return 0.5 * (1 - cov / (sigl * sigr) / n);
}
__device__ inline uint8_t ts_byte_clamped(const uint8_t* ts, int n_px, int idx) {  // only bytes no valid patch reads can be clamped
  return ts[idx < 0 ? 0 : (idx >= n_px ? n_px - 1 : idx)];
}

template <bool COARSE, bool UPDOWN>
__global__ void __launch_bounds__(64) bm_match_any_kernel(BmArgs a, DevParams p, int lds_strip_bytes) {
  const int l = threadIdx.x;
  const u32 w = a.gidx ? blockIdx.x : blockIdx.x * (u32)p.ev_nshards + (u32)p.ev_shard;
  const u32 n_out = a.gidx ? a.n_loc : a.n;
  const int W = p.W, H = p.H, wx = p.wx, wy = p.wy, N = wx * wy, hx = (wx - 1) / 2, hy = (wy - 1) / 2;
  const int nd = p.dmax - p.dmin + 1;
  uint8_t* ldsL = bm_smem;                                          // [wy][wx]
  uint8_t* ldsR = bm_smem + ((N + 7) & ~7);                         // [wy][wx + nd - 1] / UPDOWN: [wy + nd - 1][wx]
  double* cost_row = reinterpret_cast<double*>(ldsR + lds_strip_bytes);  // COARSE only: [nd]
  const int SW = wx + nd - 1;  // strip row length (horizontal search)

  bool ok = w < n_out;
  u32 k = 0;
  uint4 e = make_uint4(0, 0, 0, 0);
  if (ok) bm_item(a, p, w, k, e, ok);
  const int ex = e.x & 0xffffu, ey = e.x >> 16;
  double xr = 0, yr = 0;
  int x1 = 0, y1 = 0;
  if (ok) ok = ex < W && ey < H;
  if (ok) {
    const float2 q = a.lut[ey * W + ex];  // EventBM.cpp:88
    xr = (double)q.x;
    yr = (double)q.y;
    ok = !(xr < 0 || xr > (double)(W - 1) || yr < 0 || yr > (double)(H - 1));  // :90-92
    // routed band mode: the event belongs to the rank that owns floor(y_rect); the ring also holds the
    // events of the Time Surface's source rows around the band, which other ranks match
    if (a.gidx && ok) { const int yb = (int)floor(yr); ok = yb >= p.band_y0 && yb < p.band_y1; }
  }
  if (ok && a.mask) ok = a.mask[(int)yr * W + (int)xr] > 125;  // :94
  if (ok) {
    x1 = (int)floor(xr);
    y1 = (int)floor(yr);
  }
  if (ok) ok = !(x1 - hx < 1 || y1 - hy < 1 || x1 + hx >= W - 1 || y1 + hy >= H - 1);  // isValidPatch, :251-267
  int reason = 0;
  long long Sl = 0, Sll = 0;
  if (ok) {  // left patch (block of wx x wy from the left-top corner, :101), its moments, the low-texture test (:104-109)
    int cnt = 0, sl = 0, sll = 0;
    for (int i = l; i < N; i += 64) {
      const int py = i / wx, px = i - py * wx;
      const int v = a.tsL[(y1 - hy + py) * W + (x1 - hx + px)];
      ldsL[i] = (uint8_t)v;
      cnt += v < 1;
      sl += v;
      sll += v * v;
    }
    cnt = wave_sum_i32(cnt);
    Sl = wave_sum_i32(sl);
    Sll = wave_sum_i64((long long)sll);
    if ((double)cnt > 0.95 * (double)N) { ok = false; reason = 1; }
  }
  if (ok) {  // right strip
    const int n_px = W * H;
    if constexpr (UPDOWN) {  // rows y1 - dmax - hy ... , wx pixels each
      const int rows = wy + nd - 1;
      for (int i = l; i < rows * wx; i += 64) {
        const int r = i / wx, px = i - r * wx;
        ldsR[i] = ts_byte_clamped(a.tsR, n_px, (y1 - p.dmax - hy + r) * W + (x1 - hx + px));
      }
    } else {
      const int xs0 = x1 - p.dmax - hx;
      for (int i = l; i < wy * SW; i += 64) {
        const int py = i / SW, q = i - py * SW;
        ldsR[i] = ts_byte_clamped(a.tsR, n_px, (y1 - hy + py) * W + xs0 + q);
      }
    }
  }
  __syncthreads();
  auto cand_valid = [&](int d) {  // isValidPatch of candidate d (:184-190)
    if constexpr (UPDOWN) { const int y2 = y1 - d; return !(y2 - hy < 1 || y2 + hy >= H - 1); }
    else { const int x2 = x1 - d; return !(x2 - hx < 1 || x2 + hx >= W - 1); }
  };
  double best = 1.0;  // ZNCC_MAX_
  int bestd = -1;
  if (ok) {
    for (int d = p.dmin + l; d <= p.dmax; d += 64) {
      if (!cand_valid(d)) {
        if constexpr (COARSE) cost_row[d - p.dmin] = 1.0;
        continue;
      }
      const int col0 = p.dmax - d;
      u32 sr = 0, slr = 0;
      unsigned long long srr = 0;
      for (int py = 0; py < wy; ++py) {
        const uint8_t* lrow = ldsL + py * wx;
        const uint8_t* rrow = UPDOWN ? ldsR + (col0 + py) * wx : ldsR + py * SW + col0;
        for (int px = 0; px < wx; ++px) {
          const u32 lv = lrow[px], rv = rrow[px];
          sr += rv;
          srr += rv * rv;
          slr += lv * rv;
        }
      }
      const double cost = zncc_any(Sl, Sll, (long long)sr, (long long)srr, (long long)slr, N);
      if constexpr (COARSE) cost_row[d - p.dmin] = cost;
      else if (cost <= best) { best = cost; bestd = d; }  // :198
    }
  }
  if constexpr (COARSE) {
    __syncthreads();
    if (ok) {
      const int step = p.step;
      const int nc = (p.dmax - p.dmin) / step + 1;
      for (int j = l; j < nc; j += 64) {
        const int d = p.dmin + j * step;
        if (!cand_valid(d)) continue;
        const double c = cost_row[d - p.dmin];
        if (c <= best) { best = c; bestd = d; }
      }
      grp_argmin<64>(best, bestd);
      bool found = bestd >= 0 && bestd - step >= p.dmin && bestd + step <= p.dmin + (nc - 1) * step;  // :207-219
      if (found) found = cost_row[bestd - step - p.dmin] < 1.0 && cost_row[bestd + step - p.dmin] < 1.0 && best < p.zncc_thr;
      if (!found) { ok = false; reason = 2; }
      else {  // fine pass (:127-133)
        const int cd = bestd;
        double fb = 2.0;
        int fd = -1;
        for (int d = cd - (step - 1) + l; d <= cd + (step - 1); d += 64) {
          if (!cand_valid(d)) continue;
          const double c = cost_row[d - p.dmin];
          if (c <= fb) { fb = c; fd = d; }
        }
        grp_argmin<64>(fb, fd);
        best = fb;
        bestd = fd;
        if (!(best < p.zncc_thr)) { ok = false; reason = 3; }
      }
    }
  } else if (ok) {
    grp_argmin<64>(best, bestd);
    if (!(bestd >= 0 && best < p.zncc_thr)) { ok = false; reason = 2; }
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
    ok = lo < a.n_pose;  // :155-156
  }
  if (w < n_out && l == 0) {
    a.out_flags[w] = ok ? 1u : 0u;
    if (ok) {
      esvo_match_t m;
      const double disparity = (double)bestd;
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
    if (a.fail_counters && reason)
      atomicAdd(a.fail_counters + CNT_BM_FAIL + (reason - 1) * CNT_STRIPES + (blockIdx.x % CNT_STRIPES), 1u);
  }
}
template <bool COARSE, bool UPDOWN>
static void launch_bm_any(const BmArgs& a, const DevParams& p, hipStream_t s) {
  const int nd = p.dmax - p.dmin + 1, N = p.wx * p.wy;
  const int strip = ((UPDOWN ? (p.wy + nd - 1) * p.wx : p.wy * (p.wx + nd - 1)) + 7) & ~7;
  const size_t lds = (size_t)((N + 7) & ~7) + strip + (COARSE ? (size_t)nd * 8 : 0);
  const u32 own = a.gidx ? a.n_loc : ((a.n > (u32)p.ev_shard) ? (a.n - (u32)p.ev_shard + (u32)p.ev_nshards - 1) / (u32)p.ev_nshards : 0);
  if (own == 0) return;
  hipLaunchKernelGGL((bm_match_any_kernel<COARSE, UPDOWN>), dim3(own), dim3(64), lds, s, a, p, strip);
}

template <int G, bool COARSE, bool UPDOWN>
static void launch_bm_g(const BmArgs& a, const DevParams& p, int RD, hipStream_t s) {
  const int nd = p.dmax - p.dmin + 1;
  const int per_event = (28 + (UPDOWN ? (nd + 6) * 4 : 7 * RD)) * 4 + (COARSE ? ((nd + 1) & ~1) * 8 : 0);
  const int epb = BM_BLOCK / G;
  const u32 own = a.gidx ? a.n_loc : ((a.n > (u32)p.ev_shard) ? (a.n - (u32)p.ev_shard + (u32)p.ev_nshards - 1) / (u32)p.ev_nshards : 0);
  if (own == 0) return;
  const u32 blocks = (own + epb - 1) / epb;
  hipLaunchKernelGGL((bm_match_kernel<G, COARSE, UPDOWN>), dim3(blocks), dim3(BM_BLOCK), (size_t)per_event * epb, s, a, p, RD, per_event);
}
template <bool COARSE, bool UPDOWN>
static void launch_bm_mode(int G, const BmArgs& a, const DevParams& p, int RD, hipStream_t s) {
  switch (G) {
    case 64: launch_bm_g<64, COARSE, UPDOWN>(a, p, RD, s); break;
    case 32: launch_bm_g<32, COARSE, UPDOWN>(a, p, RD, s); break;
    case 16: launch_bm_g<16, COARSE, UPDOWN>(a, p, RD, s); break;
    default: launch_bm_g<8, COARSE, UPDOWN>(a, p, RD, s); break;
  }
}

void launch_bm_match(const BmArgs& a, const DevParams& p, hipStream_t s) {
  if (a.n == 0) return;
  const int nd = p.dmax - p.dmin + 1;
  const int RD = ((((nd + 2) >> 2) + 5) + 3) & ~3;  // dwords staged per strip row (16-byte pieces)
  // lanes per event: the group size that wastes the fewest candidate slots (ties -> wider)
  int bestG = 64, bestSlots = 1 << 30;
  for (int G = 64; G >= 8; G >>= 1) {
    const int slots = ((nd + G - 1) / G) * G;
    if (slots < bestSlots) { bestSlots = slots; bestG = G; }
  }
  const bool coarse = p.step > 1;
  if (p.wx != 15 || p.wy != 7) {  // any other patch size: the general kernel
    if (p.updown) { if (coarse) launch_bm_any<true, true>(a, p, s); else launch_bm_any<false, true>(a, p, s); }
    else { if (coarse) launch_bm_any<true, false>(a, p, s); else launch_bm_any<false, false>(a, p, s); }
    return;
  }
  if (p.updown) {
    if (coarse) launch_bm_mode<true, true>(bestG, a, p, RD, s);
    else launch_bm_mode<false, true>(bestG, a, p, RD, s);
  } else {
    if (coarse) launch_bm_mode<true, false>(bestG, a, p, RD, s);
    else launch_bm_mode<false, false>(bestG, a, p, RD, s);
  }
}

// stable compaction: slot w -> position prefix[w]; slot_of (optional) remembers w
__global__ void __launch_bounds__(256) compact_matches_kernel(const esvo_match_t* __restrict__ slots, const u32* __restrict__ flags,
                                                              const u32* __restrict__ prefix, u32 n,
                                                              esvo_match_t* __restrict__ out, u32* __restrict__ slot_of) {
  const u32 w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n || !flags[w]) return;
  out[prefix[w]] = slots[w];
  if (slot_of) slot_of[prefix[w]] = w;
}
void launch_compact_matches(const esvo_match_t* slots, const u32* flags, const u32* prefix, u32 n, esvo_match_t* out,
                            u32* slot_of, hipStream_t s) {
  if (n == 0) return;
  hipLaunchKernelGGL(compact_matches_kernel, dim3((n + 255) / 256), dim3(256), 0, s, slots, flags, prefix, n, out, slot_of);
}

// esvo_MVStereo::vEMP2vDP (esvo_MVStereo.cpp:1072-1094): one Gaussian DepthPoint per match for the PURE_BLOCK_MATCHING mode --
// pseudo-variance 0, i.e. the 1e-6 bound of DepthPoint::boundVariance; residual = ZNCC cost; age = age_vis_threshold.
__global__ void __launch_bounds__(256) matches_to_points_kernel(const esvo_match_t* __restrict__ m, const u32* __restrict__ n_ptr,
                                                                u32 max_n, DevPoint* __restrict__ out, DevParams p) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  u32 n = *n_ptr;
  if (n > max_n) n = max_n;
  if (i >= n) return;
  const esvo_match_t e = m[i];
  DevPoint o;
  o.row = (u32)(size_t)floor(e.x_left[1]);
  o.col = (u32)(size_t)floor(e.x_left[0]);
  o.x[0] = e.x_left[0];
  o.x[1] = e.x_left[1];
  cam2World(p.camL, e.x_left[0], e.x_left[1], e.inv_depth, o.p_cam);
  o.inv_depth = e.inv_depth;  // DepthPoint::update on a new point
  o.variance = 1e-6;
  o.scale2 = 0;
  o.nu = 0;
  o.residual = e.cost;
  o.age = (u64)p.age_thr;
  o.pose_idx = e.pose_idx;
  o.seq = i;
  out[i] = o;
}
void launch_matches_to_points(const esvo_match_t* m, const u32* n_ptr, u32 max_n, DevPoint* out, const DevParams& p, hipStream_t s) {
  if (max_n == 0) return;
  hipLaunchKernelGGL(matches_to_points_kernel, dim3((max_n + 255) / 256), dim3(256), 0, s, m, n_ptr, max_n, out, p);
}

}  // namespace esvo
