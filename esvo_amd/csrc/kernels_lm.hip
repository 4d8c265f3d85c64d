// kernels_lm.hip — K4: per-match inverse-depth refinement (1-D Levenberg-Marquardt, f64) on gfx950.
//
// Replaces DepthProblemSolver::solve_multiple_problems / solve_single_problem_numerical
// (esvo_core/src/core/DepthProblemSolver.cpp:80-214), DepthProblem::operator() / warping /
// patchInterpolation (esvo_core/src/core/DepthProblem.cpp:34-262) and the third-party
// Eigen::LevenbergMarquardt<NumericalDiff<...>> for one unknown (MINPACK lmdif logic,
// SURVEY.md Appendix B.1), plus DepthProblemSolver::pointCulling (:216-244).
//
// Work decomposition: the 15x7 patch needs a 16x8 source block per image, so a match maps
// onto a 16-lane group: lane c loads column c of the block (8 rows, both images), takes
// column c+1 from its neighbour with a 16-wide shuffle, and owns the 7 residuals of patch
// column c (lane 15 only feeds its neighbour).  Four matches share one wave64; the LM control
// flow is uniform inside a group and diverges between groups via the exec mask.  Patch sums
// (t-scale update, |f|, |J|, J^T f) are reduced in a fixed order: per-lane over rows, then an
// xor butterfly over the 16 lanes — the oracle's "canonical" order, so results match the CPU
// oracle bit for bit.  All arithmetic is f64 (the forward-difference step is sqrt(eps)*|x|).
//
// The Student-t scale iteration (DepthProblem.cpp:96-124) has no iteration cap in the
// reference.  When at most floor(0.94*N/(nu+1)) residuals are non-zero (and none is below
// 1e-6 in magnitude) the update s2 <- mean(...) contracts by > 6% per step for ever, s2
// underflows to 0, the sum becomes exactly 0 and the reference resets s2 to Tdist_scale^2
// after ~10^4 iterations; that provable outcome is taken directly.  Every
// other case runs the literal loop.
#define DEV_HOOKS_LM_TU
#include "common.hpp"
#include "dev_hooks.hpp"
#include "fdiv.hpp"
#include "lm_common.hpp"


namespace esvo {

#define LM_ROWS 7
#define LM_COLS 15

// ---- 16-lane (DPP row) cross-lane helpers ---------------------------------------------------------
// A DPP row is 16 lanes = one match group, so group reductions need no LDS (ds_bpermute) round
// trip.  The all-reduce adds partner values in four steps: lane^1 and lane^2 by quad_perm, then
// the other quad of the 8-lane half (row_half_mirror) and the other half (row_mirror).  After each
// step all lanes of the combined sub-group hold the same value, so the sums are exactly those of
// the xor butterfly (lane^1, ^2, ^4, ^8) the oracle's canonical order prescribes.
template <int CTRL>
__device__ inline double dpp_f64(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const int lo2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
  const int hi2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi2, lo2);
}
template <int CTRL>
__device__ inline int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
#define DPP_XOR1 0xB1         // quad_perm [1,0,3,2]
#define DPP_XOR2 0x4E         // quad_perm [2,3,0,1]
#define DPP_HALF_MIRROR 0x141 // lane i <-> 7-i inside each 8-lane half
#define DPP_MIRROR 0x140      // lane i <-> 15-i inside the row
#define DPP_SHL1 0x101        // lane i <- lane i+1 (lane 15 gets 0)

__device__ inline double grp_sum(double v) {
  v = v + dpp_f64<DPP_XOR1>(v);
  v = v + dpp_f64<DPP_XOR2>(v);
  v = v + dpp_f64<DPP_HALF_MIRROR>(v);
  v = v + dpp_f64<DPP_MIRROR>(v);
  return v;
}
__device__ inline int grp_sum_int(int v) {
  v += dpp_i32<DPP_XOR1>(v);
  v += dpp_i32<DPP_XOR2>(v);
  v += dpp_i32<DPP_HALF_MIRROR>(v);
  v += dpp_i32<DPP_MIRROR>(v);
  return v;
}
__device__ inline double grp_min(double v) {
  v = fmin(v, dpp_f64<DPP_XOR1>(v));
  v = fmin(v, dpp_f64<DPP_XOR2>(v));
  v = fmin(v, dpp_f64<DPP_HALF_MIRROR>(v));
  v = fmin(v, dpp_f64<DPP_MIRROR>(v));
  return v;
}
// ---- the two lane layouts ---------------------------------------------------------------------------
// NARROW (throughput): a match is a 16-lane group, lane = patch column, 7 rows per lane, four matches per wave.
// WIDE (latency): a match is a whole wave: lane = (row group, column); the four 16-lane DPP rows of the wave hold patch rows
//   {0,1}, {2,3}, {4,5}, {6}.  One match per wave: no lockstep between matches, and 2 instead of 7 rows of dependent
//   f64 work per lane -- for launches that cannot fill the chip anyway (a reference-faithful tick: a few hundred to a few
//   thousand matches on 2048 wave slots) the launch lasts as long as its slowest match, and that match is ~2x shorter.
// Both reduce in the SAME canonical order: per column the sequential sum over rows 0..6, then the xor butterfly over the
// 16 columns.  In the wide layout the column sum travels through the row groups (rows {0,1} -> +{2,3} -> +{4,5} -> +{6}),
// one v_permlane{16,32}_swap per hop: the groups sit in DPP rows 0, 1, 3, 2 of the wave so that every hop is one swap.
template <bool WIDE>
struct Lay {
  static constexpr int RL = WIDE ? 2 : LM_ROWS;  // patch rows per lane
};
// value of the same column in the previous row group (wide layout); hop k leads into group k
__device__ inline double hop_f64(double v, int k) {
  const u32 lo = (u32)__double2loint(v), hi = (u32)__double2hiint(v);
  u32 l2, h2;
  if (k == 2) {        // DPP row 1 -> row 3
    l2 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false)[0];
    h2 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false)[0];
  } else if (k == 1) {  // DPP row 0 -> row 1
    l2 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false)[0];
    h2 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false)[0];
  } else {             // DPP row 3 -> row 2
    l2 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false)[1];
    h2 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false)[1];
  }
  return __hiloint2double((int)h2, (int)l2);
}
#define LM_WIDE_LAST_ROW_LANE 32  // first lane of the DPP row that holds row group 3 (the end of the column-sum chain)
__device__ inline double bcast_f64(double v, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
// canonical sum over the patch of the per-lane values t[]; the result is in every lane of the match
template <bool WIDE>
__device__ inline double patch_sum(const double* t, int rg) {
  if constexpr (!WIDE) {
    double a = t[0];
#pragma unroll
    for (int y = 1; y < LM_ROWS; ++y) a = a + t[y];
    return grp_sum(a);
  } else {
    double a = t[0] + t[1];                    // rows 0, 1                      (meaningful in group 0)
    a = (hop_f64(a, 1) + t[0]) + t[1];          // + rows 2, 3                    (group 1)
    a = (hop_f64(a, 2) + t[0]) + t[1];          // + rows 4, 5                    (group 2)
    a = hop_f64(a, 3) + t[0];                   // + row 6                        (group 3)
    (void)rg;
    return bcast_f64(grp_sum(a), LM_WIDE_LAST_ROW_LANE);
  }
}
template <bool WIDE>
__device__ inline int match_sum_int(int v) {
  v = grp_sum_int(v);
  if constexpr (WIDE)
    v = (__builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16)) + (__builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48));
  return v;
}
template <bool WIDE>
__device__ inline double match_min(double v) {
  v = grp_min(v);
  if constexpr (WIDE) v = fmin(fmin(bcast_f64(v, 0), bcast_f64(v, 16)), fmin(bcast_f64(v, 32), bcast_f64(v, 48)));
  return v;
}
template <bool WIDE>
__device__ inline double patch_dot(const double* a, const double* b, int rg) {
  double t[Lay<WIDE>::RL];
#pragma unroll
  for (int y = 0; y < Lay<WIDE>::RL; ++y) t[y] = a[y] * b[y];
  return patch_sum<WIDE>(t, rg);
}

__device__ inline double in_vgpr(double x) {
  asm volatile("" : "+v"(x));
  return x;
}

template <bool T_LDS> struct LmPose { double T[12]; };            // T_left_virtual (3x4) in registers
template <> struct LmPose<true> { const double* T; };            // ... in LDS: 24 VGPRs less (the persistent kernel)
#define LM_T_LDS_DEFAULT false
template <bool T_LDS>
struct LmProblemT : LmPose<T_LDS> {
  double cx, cy;           // rectified left coordinate of the event
  double ray[3];           // Kinv [cx cy 1]^T: the part of cam2World that does not depend on the inverse depth
  // The two Time Surfaces as buffer resources: a patch byte is  buffer_load_ubyte v, v_off, s[rsrc], s_row offen  with the
  // per-lane offset of the block's first row in v_off and the row stride in a scalar -- no per-row 64-bit address
  // arithmetic on the vector ALU, and an offset outside the image reads 0 instead of faulting, so the loads can be issued
  // before the bounds predicate is resolved (no branch ladder around them).
  __amdgpu_buffer_rsrc_t tsL;
  __amdgpu_buffer_rsrc_t tsR;
  // The 27 camera constants an evaluation reads (P_left, P_right, Kinv_t of the left camera) from LDS instead of the
  // kernel-argument SGPRs: with the two projection matrices, the image descriptors and the driver's state the scalar
  // register file overflowed and every evaluation re-fetched ~60 spilled values with v_readlane (vector-ALU issue slots).
  // An LDS read costs none; the narrow layout may keep the values in VGPRs across the loop (it has the room), the wide one
  // re-reads them per evaluation (128-VGPR budget).
  const double* cam;       // [0..11] P_left, [12..23] P_right, [24..26] Kinv_t (left)
  int c;                   // patch column of the lane
  int rg;                  // row group of the lane (wide layout; 0 in the narrow one)
  int vy0, vy1;            // BAND kernels only: the rows of the observation pair that hold data (routed band mode)
  DEV_LM_PROBLEM_FIELDS
};
typedef LmProblemT<LM_T_LDS_DEFAULT> LmProblem;

// patchInterpolation's geometry (DepthProblem.cpp:193-230) as straight-line code: the block's upper-left corner, the four
// bilinear weights and the predicate "the patch lies inside the image".  Nothing here is guarded: for a location that fails
// the predicate (or is not finite) the integers are garbage, which only ever feeds bounds-checked buffer loads.
struct PatchGeom { int ulx, uly; double q1, q2, q3, q4; bool ok; };
__device__ inline PatchGeom interp_geom(const DevParams& p, double lx, double ly) {
  PatchGeom g;
  const int hx = (LM_COLS - 1) / 2, hy = (LM_ROWS - 1) / 2;
  const double fx = floor(lx), fy = floor(ly);
  const int l1 = (int)fx, l0 = (int)fy;
  g.ulx = l1 - hx;
  g.uly = l0 - hy;
  const int drx = l1 + hx, dry = l0 + hy;
  g.ok = (int)(g.ulx >= 0) & (int)(g.uly >= 0) & (int)(drx < p.W) & (int)(dry < p.H) & (int)(g.uly + LM_ROWS < p.H) & (int)(g.ulx + LM_COLS < p.W);
  // (double)(l1 + 1) - lx etc. (:215-222): floor(lx) already IS (double)l1, and adding 1.0 to an integer-valued double
  // below 2^31 is exact, so the int -> double conversions are not needed (a location outside that range fails g.ok)
  g.q1 = (fx + 1.0) - lx;
  g.q2 = lx - fx;
  g.q3 = (fy + 1.0) - ly;
  g.q4 = ly - fy;
  return g;
}
// The source bytes of the lane's block column, kept across the evaluations of a match as floats (exact: 0..255).  lmdif's
// evaluations come in pairs at x and x + sqrt(eps) |x| -- the same pixel block, other bilinear weights -- and a converging
// trial step rarely leaves the block either, so most evaluations find both images' columns here and issue no load at all
// (measured: a second, dependent batch of these loads costs the tick 8.5 %).  voff = the block column's first byte, the key;
// a wave reloads when ANY of its matches moved (the branch is wave-uniform), and a failed warp's garbage offset is as good
// a key as any: the loads are bounds-checked, the same offset gives the same bytes.
template <bool WIDE>
struct LmCache {
  float l[Lay<WIDE>::RL + 1], r[Lay<WIDE>::RL + 1];
  int vl, vr;
  __device__ inline void clear() {
#pragma unroll
    for (int y = 0; y <= Lay<WIDE>::RL; ++y) l[y] = r[y] = 0.f;  // what an out-of-range offset reads
    vl = vr = 0x7fffffff;
  }
};
__device__ inline float dpp_shl1_f32(float v) { return __int_as_float(dpp_i32<DPP_SHL1>(__float_as_int(v))); }
template <bool WIDE>
__device__ inline void interp_column(__amdgpu_buffer_rsrc_t img, int W, const PatchGeom& g, int c, int rg, double* tau, float* cs,
                                     int& cvoff) {
  constexpr int RL = Lay<WIDE>::RL;
  // first source row of the lane (the wide layout's row group rg owns rows 2 rg, 2 rg + 1 and reads one more; group 3 owns
  // row 6 only: its third source row lies below the block and is not used -- wherever it falls, the load is bounds-checked)
  const int voff = (g.uly + (WIDE ? 2 * rg : 0)) * W + g.ulx + c;
  if (__ballot(voff != cvoff) != 0ull) {
#pragma unroll
    for (int y = 0; y <= RL; ++y) cs[y] = (float)(int)__builtin_amdgcn_raw_buffer_load_b8(img, voff, y * W, 0);
    cvoff = voff;
  }
  double R[RL + 1];
#pragma unroll
  for (int y = 0; y <= RL; ++y) {
    const float s0 = cs[y];
    const float s1 = dpp_shl1_f32(s0);  // column c+1 from the neighbour lane
    R[y] = g.q1 * (double)s0 + g.q2 * (double)s1;
  }
#pragma unroll
  for (int y = 0; y < RL; ++y) tau[y] = g.q3 * R[y] + g.q4 * R[y + 1];
}

// DepthProblem::operator().  fv[y] = residual of patch element (y, c); lane 15 -> 0.
// L2 (LSnorm "l2", DepthProblem.cpp:43-45,67-75,143-147; no shipped configuration sets it): the plain temporal residual
// tau_L - tau_R, 255 where warping or interpolation fails; no weights, no scale iteration.
// Returns whether the evaluation was "tight" (group-uniform): every non-zero residual of the match has 2^-50 <= |r| and the
// final scale lies in [2^-100, 2^100], so every non-zero f = sqrt(w) r has 2^-150 <= |f| < 2^11 -- what lets the caller
// divide differences of two such evaluations through a shared reciprocal (fdiv.hpp's window) without testing them.
// BAND (routed band mode: the rank's observation pair holds its band + halo rows only): an evaluation whose source blocks
// leave those rows would read stale bytes -- it is recorded in `viol` (the tick is then refused with ESVO_ERR_HALO by every
// rank, include/esvo_hip.h), never silently used.
template <bool WIDE, bool L2, bool COUNT = false, bool BAND = false, bool T_LDS = LM_T_LDS_DEFAULT>
__device__ bool lm_eval(const DevParams& p, const LmProblemT<T_LDS>& pr, LmCache<WIDE>& cc, double x, double* fv, int* n_iter = nullptr,
                        bool* viol = nullptr) {
  int iters = 0;  // t-scale iterations of this evaluation (COUNT only: the split launch orders the matches by it)
  constexpr int RL = Lay<WIDE>::RL;
  // element (y, c) of the patch exists: column 15 only feeds its neighbour; row group 3 of the wide layout owns one row
  bool el[RL];
#pragma unroll
  for (int y = 0; y < RL; ++y) el[y] = pr.c < LM_COLS && (!WIDE || 2 * pr.rg + y < LM_ROWS);
  const double nu = in_vgpr(p.td_nu);  // VGPR: otherwise re-loaded from the kernel arguments in every t-scale iteration
  DEV_LM_EVAL(pr);  // (tools: evaluations per group / per wave; empty in the product, dev_hooks.hpp)
  if constexpr (WIDE) asm volatile("" ::: "memory");  // the LDS reads below stay inside the loop (see LmProblem::cam)
  double prv[3], pl[3];
  {  // cam2World(p.camL, cx, cy, x): z * (Kinv [cx cy 1]^T) - Kinv_t with the ray computed once per match
    const double z = 1.0 / x;
#pragma unroll
    for (int r = 0; r < 3; ++r) prv[r] = z * pr.ray[r] - pr.cam[24 + r];
  }
#pragma unroll
  for (int r = 0; r < 3; ++r)
    pl[r] = ((pr.T[r * 4 + 0] * prv[0] + pr.T[r * 4 + 1] * prv[1]) + pr.T[r * 4 + 2] * prv[2]) + pr.T[r * 4 + 3];
  double x1u, x1v, x2u, x2v;
  {  // world2Cam for both cameras (common.hpp), the matrices from LDS
    double hL[3], hR[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const double* PL = pr.cam + r * 4;
      const double* PR = pr.cam + 12 + r * 4;
      hL[r] = ((PL[0] * pl[0] + PL[1] * pl[1]) + PL[2] * pl[2]) + PL[3];
      hR[r] = ((PR[0] * pl[0] + PR[1] * pl[1]) + PR[2] * pl[2]) + PR[3];
    }
    // Rectified cameras share the third row of P, so the four quotients have ONE divisor.  With |h2| in [2^-100, 2^100]
    // (one exponent test) they go through a shared refined reciprocal: a coordinate that passes the bounds test below lies
    // in [3, W], i.e. its numerator lies in fdiv.hpp's window and the quotient is the IEEE one; a numerator outside the
    // window gives a quotient outside [3, W] (or NaN) on either path, and such an evaluation takes the failure fill
    // whatever the coordinate's bits are.
    if (__double_as_longlong(hL[2]) == __double_as_longlong(hR[2]) &&
        (unsigned)(((__double2hiint(hL[2]) >> 20) & 0x7ff) - 923) <= 200u) {
      Recip rz;
      rz.b = hL[2];
      rz.y = recip_refined(hL[2]);
      x1u = div_fast(hL[0], rz);
      x1v = div_fast(hL[1], rz);
      x2u = div_fast(hR[0], rz);
      x2v = div_fast(hR[1], rz);
    } else {
      x1u = hL[0] / hL[2];
      x1v = hL[1] / hL[2];
      x2u = hR[0] / hR[2];
      x2v = hR[1] / hR[2];
    }
  }
  const int hx = (LM_COLS - 1) / 2, hy = (LM_ROWS - 1) / 2;
  // warping bounds (DepthProblem.cpp:186-189) and patchInterpolation's (:205-230), as ONE predicate in straight-line code:
  // every comparison is evaluated, nothing is short-circuited into a branch ladder (a NaN coordinate fails `inside`)
  const double wlo = (double)hx, whi = (double)(p.W - hx), vlo = (double)hy, vhi = (double)(p.H - hy);
  const bool inside = (int)(x1u >= wlo) & (int)(x1u <= whi) & (int)(x1v >= vlo) & (int)(x1v <= vhi) &
                      (int)(x2u >= wlo) & (int)(x2u <= whi) & (int)(x2v >= vlo) & (int)(x2v <= vhi);
  const PatchGeom g1 = interp_geom(p, x1u, x1v), g2 = interp_geom(p, x2u, x2v);
  const bool okw = inside & g1.ok & g2.ok;
  if constexpr (BAND) {  // source rows uly .. uly + LM_ROWS of both blocks
    const bool bad = (int)okw & ((int)(g1.uly < pr.vy0) | (int)(g1.uly + LM_ROWS >= pr.vy1) | (int)(g2.uly < pr.vy0) | (int)(g2.uly + LM_ROWS >= pr.vy1));
    *viol = *viol | bad;
  }
  double tau1[RL], tau2[RL], r[RL], r2[RL];
  interp_column<WIDE>(pr.tsL, p.W, g1, pr.c, pr.rg, tau1, cc.l, cc.vl);
  interp_column<WIDE>(pr.tsR, p.W, g2, pr.c, pr.rg, tau2, cc.r, cc.vr);
  if (!okw) {  // failure fill, DepthProblem.cpp:49-56 / :149-155 (l2: :67-75, :143-147)
    double f = 255;
    if constexpr (!L2) {
      const double q = f / p.td_scale;
      const double weight = (nu + 1) / (nu + q * q);
      f = sqrt(weight) * f;
    }
#pragma unroll
    for (int y = 0; y < RL; ++y) fv[y] = el[y] ? f : 0.0;
    if constexpr (COUNT) *n_iter = 0;
    return false;
  }
  if constexpr (L2) {
#pragma unroll
    for (int y = 0; y < RL; ++y) fv[y] = el[y] ? (tau1[y] - tau2[y]) : 0.0;
    return false;
  }
  int knz = 0;
  double minabs = 1e300, r2max = 0;
  double r2n[RL];  // r^2 (nu + 1): the numerators of the t-scale update
#pragma unroll
  for (int y = 0; y < RL; ++y) {
    r[y] = el[y] ? (tau1[y] - tau2[y]) : 0.0;
    r2[y] = r[y] * r[y];
    r2n[y] = r2[y] * (nu + 1);
    r2max = fmax(r2max, r2[y]);
    if (r[y] != 0) { knz++; minabs = fmin(minabs, fabs(r[y])); }
  }
  const int knz_lane = knz;
  knz = match_sum_int<WIDE>(knz);
  DEV_LM_FIRST_EVAL_KNZ(pr, knz);
  minabs = match_min<WIDE>(minabs);
  const double scale2_0 = in_vgpr(p.td_scale2);
  double s2;
  const int e_r2max = (__double2hiint(r2max) >> 20) & 0x7ff;
  // all non-zero r^2 of the lane in [2^-100, 2^98]: minabs and r2max bound them (r^2 is monotone in |r|)
  const bool r2_tight = knz_lane == 0 || (minabs * minabs >= 0x1p-100 && r2max < 0x1p98);
  const bool nu_mid = nu >= 0x1p-100 && nu < 0x1p100;
  // lane_ok: every r^2 is 0 or of moderate magnitude, so shared-divisor quotients are allowed.  The tight bounds imply
  // it; the per-row exponent tests run only for the rare lane outside them.
  bool lane_ok = r2_tight && nu_mid;
  if (!lane_ok) {
    bool r2_ok = true;
#pragma unroll
    for (int y = 0; y < RL; ++y) r2_ok = r2_ok && fdiv_ok(r2[y]);
    lane_ok = r2_ok && fdiv_ok(nu) && nu > 0 && fdiv_ok(r2max * (nu + 1));
  }
  const int N = LM_ROWS * LM_COLS;
  // "tight" evaluation (all but a handful): nu and every non-zero r^2 of the GROUP lie in [2^-100, 2^98].
  const bool tight = match_sum_int<WIDE>((r2_tight && nu_mid) ? 0 : 1) == 0;
  if ((double)knz * (nu + 1) / (double)N <= 0.94 && minabs >= 1e-6) {
    s2 = scale2_0;  // provable outcome of the uncapped loop (header comment)
    DEV_LM_SHORTCUT(pr);
  } else {
    // DepthProblem.cpp:96-124: s1 <- s2 until |s2 - s1| / s1 <= 5 %.  Every quotient below is the IEEE quotient
    // (fdiv.hpp): r^2 / s1 and the convergence test share the divisor s1, sum / N has a constant divisor, and
    // the weights' own divisors nu + r^2/s1 lie in [nu, 2^333) by the exponent test on the lane's largest r^2.
    double s1 = scale2_0;
    const Recip rN = make_recip((double)N);
    // In a tight evaluation, while s1 stays in [2^-100, 2^100], every operand of the iteration is inside fdiv.hpp's
    // window by construction (r^2/s1 <= 2^198, t >= 2^-299, 2^-299 <= sum <= 2^110, |s2 - s1| is 0 or >= ulp(2^-100)), so
    // the iteration runs without a single per-operand range test or branch: one exponent test on s1 decides, uniformly
    // for the group.
    bool done = false;
    if (tight) {
      // sign bit included in the exponent field: a negative or NaN s1 fails the range test
      while ((unsigned)(((__double2hiint(s1) >> 20) & 0xfff) - 923) <= 200u) {
        if constexpr (COUNT) ++iters;
        DEV_LM_SCALE_ITER(pr);  // (tools: t-scale iterations per group / per wave)
        Recip rs1;
        rs1.b = s1;
        rs1.y = recip_refined(s1);
        double t[RL];
#pragma unroll
        for (int y = 0; y < RL; ++y) {  // r == 0 gives +0 / nu = +0 as the reference's skip does
          Recip rd;
          rd.b = nu + div_fast(r2[y], rs1);
          rd.y = recip_refined(rd.b);
          t[y] = div_fast(r2n[y], rd);
        }
        const double sum = patch_sum<WIDE>(t, pr.rg);
        if (sum == 0) { s2 = scale2_0; done = true; break; }
        s2 = div_fast(sum, rN);
        const double rel = div_fast(fabs(s2 - s1), rs1);
        if (!(rel > 0.05)) { done = true; break; }
        s1 = s2;
      }
    }
    while (!done) {
      if constexpr (COUNT) ++iters;
      DEV_LM_SCALE_ITER(pr);
      double t[RL];
      const Recip rs1 = make_recip(s1);
      const int e_s1 = (__double2hiint(s1) >> 20) & 0x7ff;
      if (lane_ok && rs1.fast && s1 > 0 && e_r2max - e_s1 < 300) {
#pragma unroll
        for (int y = 0; y < RL; ++y)
          t[y] = div_fast(r2n[y], make_recip(nu + div_fast(r2[y], rs1)));
      } else {
#pragma unroll
        for (int y = 0; y < RL; ++y) t[y] = (r[y] != 0) ? r2[y] * (nu + 1) / (nu + r2[y] / s1) : 0.0;
      }
      const double sum = patch_sum<WIDE>(t, pr.rg);
      if (sum == 0) { s2 = scale2_0; break; }
      s2 = div_by(sum, rN);
      const double rel = div_by(fabs(s2 - s1), rs1);
      if (!(rel > 0.05)) break;
      s1 = s2;
    }
  }
  // The weights sqrt((nu + 1) / (nu + r^2 / s2)) (DepthProblem.cpp:127-135).  Tight evaluation with s2 in [2^-100, 2^100]
  // (a group-uniform test; s2 is the group's): r^2/s2 <= 2^198, so every divisor nu + r^2/s2 lies in [2^-100, 2^199] and
  // every weight in [2^-200, (nu+1)/nu] -- fdiv.hpp's window and sqrt_moderate's range -- and the seven rows run as a
  // straight line of shared-reciprocal quotients without a per-row test or select.
  if (tight && (unsigned)(((__double2hiint(s2) >> 20) & 0xfff) - 923) <= 200u) {
    Recip rs2;
    rs2.b = s2;
    rs2.y = recip_refined(s2);
    const double nu1 = nu + 1;
#pragma unroll
    for (int y = 0; y < RL; ++y) {
      Recip rd;
      rd.b = nu + div_fast(r2[y], rs2);
      rd.y = recip_refined(rd.b);
      fv[y] = sqrt_moderate(div_fast(nu1, rd)) * r[y];
    }
    if constexpr (COUNT) *n_iter = iters;
    return true;
  }
  const Recip rs2 = make_recip(s2);
  const bool fast2 = lane_ok && rs2.fast && s2 > 0 && e_r2max - ((__double2hiint(s2) >> 20) & 0x7ff) < 300;
#pragma unroll
  for (int y = 0; y < RL; ++y) {
    // fast2 bounds r^2/s2 below 2^301, so the weight lies in (2^-300, (nu+1)/nu]: sqrt_moderate's range (fdiv.hpp)
    fv[y] = fast2 ? sqrt_moderate(div_fast(nu + 1, make_recip(nu + div_fast(r2[y], rs2)))) * r[y]
                  : sqrt((nu + 1) / (nu + r2[y] / s2)) * r[y];
  }
  if constexpr (COUNT) *n_iter = iters;
  return false;
}

#ifndef LM_WAVES
#define LM_WAVES 2  // waves per SIMD.  2: 175 VGPRs, nothing spilled, kernel 3 % faster than 3 (166 VGPRs, 2 VGPRs + 63
                    // SGPRs spilled) in interleaved same-box runs; 4 (<= 128 VGPRs) spills 71 registers into the solver loop: 1.7x slower
#endif
#ifndef LM_BLOCK
#define LM_BLOCK 64   // threads per workgroup (no LDS, no barriers).  One wave per workgroup: a finished wave's slot is
                      // refilled at once instead of waiting for its three siblings -- 4 % per tick against 256
#endif
// (One match per wave in the NARROW layout -- rows 1..3 idle -- was measured on reference-faithful ticks and is not faster
// than four: 277 vs 250-340 us, 447 vs 414 us.  A small launch lasts as long as its slowest MATCH's own dependent chain;
// the wide layout shortens that chain.)
#ifndef LM_WIDE_WAVES
#define LM_WIDE_WAVES 3  // three waves per SIMD (146 VGPRs; 152 in the pair layout), 3072 waves resident.  Round 2 ran four: the
                         // kernel has grown since and at the 128-register bound it spilled 16-18 VGPRs into the solver loop --
                         // measured against it: 346x260 throughput 55.4 -> 58.1 M events/s, synchronised small ticks -2 %
#endif
#ifndef LM_WIDE_MAX
#define LM_WIDE_MAX 40000u  // launches bounded by this many matches (= events handed to block matching) use the wide layout
#endif
// ---- the split launch (narrow layout, large launches) ---------------------------------------------------------------
// Four matches share a wave in lockstep: the wave executes the LONGEST of their t-scale loops in every evaluation, and a
// launch ends with its slowest waves running alone.  A match's cost is set by its patch (how many iterations the scale
// needs), which its first evaluation F(x0) already shows.  So the launch is split:
//   STAGE 1  F(x0) of every match (minimizeInit) in slot order: residuals, |f|, the iteration count -> a histogram
//   order    matches sorted by that count, longest first (counting sort over 64 bins)
//   STAGE 2  the remaining ~19 evaluations, four NEIGHBOURS of the sorted list per wave
// Waves then hold matches of equal cost (the lockstep loss of the t-scale loop shrinks) and the expensive waves start
// first (the tail of the launch is filled with cheap ones).  Every match is solved by the same instructions on the same
// operands as in the single launch -- only its position in the grid changes -- and results are written by slot, so the
// output is bit-identical.  F(x0) travels through a scratch buffer (LmArgs::split_*: 7 x 16 doubles per match).
struct LmSplit {
  double* fvec0;   // [max_matches][7][16]
  double* fnorm0;  // [max_matches]
  u32* meta;       // [max_matches] iteration count of F(x0) (clamped to 63) | tight << 8
  u32* order;      // [max_matches] slot of the k-th match in processing order
  u32* hist;       // [STRIPES][64] matches per iteration count, then [STRIPES][64] fill counters of the scatter.  Striped by
                   // the first stage's block index: 4 x 10^4 atomics on the five or six bins that occur in practice
                   // serialise on their L2 lines otherwise (measured: 0.18 ms for the sort alone)
};
#define LM_SPLIT_BINS 64
#define LM_SPLIT_STRIPES 32
__device__ inline u32 lm_split_stripe(u32 slot) { return (slot >> 2) & (LM_SPLIT_STRIPES - 1); }  // 4 slots per first-stage wave

// Register footprint of the narrow layout: 199 VGPRs, two waves per SIMD, 112 registers left for one wave of a matching- or
// fusion-stage kernel.  Measured in round 3 and not kept (profiles/HISTORY.md): the match's pose matrix in LDS (~160 VGPRs, three
// waves per SIMD) and the kernel padded to 169..176 VGPRs so that two waves of every fusion-stage kernel fit beside two LM
// waves -- a faster LM launch and a LONGER tick both times: more resident waves of the other stages take issue slots from the LM
// kernel without finishing sooner themselves.
// ---- the pair layout (wide layout, the smallest launches) ----------------------------------------------------------------
// A small launch lasts as long as its slowest match's DEPENDENT CHAIN of evaluations, and lmdif's chain alternates two kinds:
// the trial point F(x + p) and, once the step is accepted, the forward-difference point F(x' + h(x')) at the new x' = x + p.
// The second is known as soon as the first is -- h(x') = sqrt(eps) |x'| -- so TWO waves work on one match: both run the same
// driver on the same state (they stay in lockstep by construction), wave 0 evaluates the point the driver asks for, wave 1
// the difference point that follows if the step is accepted, and they swap residuals through LDS (one barrier).  When the
// driver then asks for exactly the point wave 1 evaluated (compared bit for bit) the residuals are there; otherwise -- a
// rejected step, a restart of the outer loop at an unchanged x -- both waves evaluate as before.  F is a pure function, so
// the result is the same bits; an accepted LM iteration costs one evaluation's latency instead of two (F(x0) and
// F(x0 + h) of minimizeInit likewise).  Twice the waves per match: used while the launch's matches fit the chip twice.
// Whether it pays depends on the data: on the un-smoothed 346x260 surfaces (long chains of accepted steps) a synchronised
// 1000-event tick drops from 0.50 to 0.41 ms and ticks of 2500-7500 events by 14 %; on DSEC's smoothed surfaces the same
// sizes get 2-8 % SLOWER (short chains: the second wave, the exchange and the extra SGPR spills cost more than the few
// reused evaluations save), and above ~2000 matches the doubled waves crowd the SIMDs.  The handle therefore measures
// (api_map.hip, lm_pair_policy): both layouts give the same bits, so it may switch between ticks.
template <bool WIDE, bool L2 = false, int STAGE = 0, bool PAIR = false, bool BAND = false>
__global__ void __launch_bounds__(PAIR ? 128 : LM_BLOCK, WIDE ? LM_WIDE_WAVES : LM_WAVES) lm_refine_kernel(LmArgs a, DevParams p, u32* n_solved, LmSplit sp) {
  static_assert(!PAIR || (WIDE && !L2 && STAGE == 0), "the pair layout is a variant of the wide one");
  static_assert(!BAND || (STAGE == 0 && !PAIR), "routed band launches are single launches of the narrow or the wide layout");
  constexpr int RL = Lay<WIDE>::RL;
  // [exchange parity][wave][row][lane]: residuals of the two points evaluated side by side, + whether each was tight
  __shared__ double lds_pair[PAIR ? 2 : 1][2][RL][64];
  __shared__ int lds_pair_tight[PAIR ? 2 : 1][2];
  const int wv = PAIR ? (int)(threadIdx.x >> 6) : 0;
  const int ln = threadIdx.x & 63;
  __shared__ double lds_cam[28];
  if (threadIdx.x < 27) {
    const int i = threadIdx.x;
    lds_cam[i] = i < 12 ? p.camL.P[i] : (i < 24 ? p.camR.P[i - 12] : p.camL.Kinv_t[i - 24]);
  }
  __syncthreads();
  const u32 pos = WIDE ? blockIdx.x : (blockIdx.x * LM_BLOCK + threadIdx.x) >> 4;  // position in the grid
  const int c = threadIdx.x & 15;
  const int drow = (threadIdx.x >> 4) & 3;
  const int rg = WIDE ? (drow ^ (drow >> 1)) : 0;  // row groups 0,1,2,3 sit in DPP rows 0,1,3,2 (patch_sum)
  const bool lead = WIDE ? threadIdx.x == 0 : c == 0;
  u32 M = *a.n_matches;
  if (M > a.max_matches) M = a.max_matches;
  bool active = pos < M;
  // solver slot (thread-stride order): the grid position, or -- second stage of a split launch -- what the sort put there
  u32 s = pos;
  if constexpr (STAGE == 2) {
    // the sort's counters, for the next launch (its kernels are done): 2 x STRIPES x BINS words, 64 per block
    if (blockIdx.x < 2 * LM_SPLIT_STRIPES * LM_SPLIT_BINS / LM_BLOCK) sp.hist[blockIdx.x * LM_BLOCK + threadIdx.x] = 0u;
    s = active ? sp.order[pos] : 0xffffffffu;
  }
  // The grid is sized for the worst case (every event matched): waves without any match leave at once (~39 000 of 49 000 on the
  // headline workload; they are the LAST in dispatch order, execute ~12 instructions each -- 0.1 % of the launch -- and delay no
  // working wave).  Sizing the grid from the previous tick's count with a strided tail was built in round 5 and dropped: wrapping
  // the body in the loop costs the register allocator 7 VGPRs (199 -> 206: a block-matching wave no longer fits beside two LM
  // waves) and triples the SGPR spills (29 -> 96), for no time to win.
  // Inactive groups of a partially filled wave run the (cheap, failing) code path below with a dummy
  // problem so that the wave's control flow stays simple; they write nothing.
  if (STAGE != 2 && !active && lead && s < a.max_matches) a.out_flags[s] = 0u;  // every slot of the launch gets its flag: no memset
  if (__ballot(active) == 0) return;
  // shader-clock probe (LmArgs::clk): the start values go to memory, not into registers that would stay live across the solver
  // (every 65th workgroup: the dispatcher deals workgroups round-robin over the eight XCDs, a stride coprime with 8 visits them all)
  const bool probe = STAGE != 1 && a.clk != nullptr && threadIdx.x == 0 && blockIdx.x % CLK_STRIDE == 0u;
  if (probe) {
    u64* sc = a.clk + CLK_SCRATCH + 2 * (size_t)(blockIdx.x / CLK_STRIDE);
    sc[0] = __builtin_readcyclecounter();
    sc[1] = __builtin_amdgcn_s_memrealtime();
  }
  u32 j = 0;
  esvo_match_t m;
  m.x_left[0] = m.x_left[1] = -1e9; m.inv_depth = 1.0; m.pose_idx = 0; m.cost = 0; m.disp = 0; m.event_idx = 0;
  if (active) {
    j = a.dense ? s : stride_item(s, M, (u32)p.num_threads);  // DepthProblemSolver.cpp:90 (dense: kernels_shard.hip)
    u32 jm = j;
    if constexpr (WIDE) { if (a.match_index) jm = a.match_index[j]; }  // (LmArgs::match_index: the list as indices into the slots)
    m = a.matches[jm];
  }
  LmProblem pr;
  pr.cx = m.x_left[0];
  pr.cy = m.x_left[1];
  {  // one bounds-checked window over each image (0x00020000: untyped 32-bit data format, the gfx9 raw-buffer descriptor)
    const int n_bytes = p.W * p.H;
    pr.tsL = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a.tsL), 0, n_bytes, 0x00020000);
    pr.tsR = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a.tsR), 0, n_bytes, 0x00020000);
  }
#pragma unroll
  for (int r3 = 0; r3 < 3; ++r3)
    pr.ray[r3] = (p.camL.Kinv[r3 * 3 + 0] * pr.cx + p.camL.Kinv[r3 * 3 + 1] * pr.cy) + p.camL.Kinv[r3 * 3 + 2];
  pr.cam = lds_cam;
  pr.c = c;
  pr.rg = rg;
  pr.vy0 = a.vy0; pr.vy1 = a.vy1;
  bool viol = false;
  DEV_LM_SET_SLOT(pr, active ? s : 0xffffffffu);
  {  // DepthProblem::setProblem, DepthProblem.cpp:17-32
    double Tlw[16], Tlv[16];
    rigid_inverse(a.T_world_obs, Tlw);
    mat4_mul(Tlw, a.pose_T + (size_t)m.pose_idx * 16, Tlv);
#pragma unroll
    for (int i = 0; i < 12; ++i) pr.T[i] = Tlv[i];
  }
  const int N = LM_ROWS * LM_COLS;
  const double ftol = 1e-6, xtol = 1e-6, gtol = 0., factor = 100.;
  const double eps = 2.220446049250313e-16;
  const double sqrt_eps = 1.4901161193847656e-08;  // sqrt(DBL_EPSILON), exact
  const int maxfev = p.lm_maxfev;

  // The solver is written as a state machine around ONE call site of the residual evaluator
  // (phase 0: F(x0) of minimizeInit; phase 1: F(x+h) of NumericalDiff; phase 2: F(x+p) of the
  // trust-region trial).  Arithmetic and control flow are those of Eigen's
  // minimizeInit/minimizeOneStep driven by the loop of DepthProblemSolver.cpp:161-188; a single
  // inlined evaluator keeps the kernel ~3x smaller (I-cache, register pressure).
  double x = m.inv_depth;
  double fvec[RL], out[RL];
  LmCache<WIDE> cc;
  cc.clear();
  double fnorm = 0., par = 0., diag = 0., xnorm = 0., delta = 0., r = 0., qtf = 0., gnorm = 0.;
  double h = 0., xnew = 0., wa1 = 0., pnorm = 0.;
  int nfev = 1, iter = 1, iteration = 0, optState = 0;
  int phase = 0;
  bool need_step = false;
  bool fvec_tight = false;  // fvec comes from a tight evaluation (lm_eval's return value)
  double xe = x;
  // pair layout: the point wave 1 evaluated in the last exchange, which LDS buffer holds it, whether it is still there
  double xs = 0.;
  int xpar = 0;
  bool have_spec = false;
  if constexpr (STAGE == 1) {  // minimizeInit only: F(x0), |F(x0)|, what the evaluation cost
    int n_it = 0;
    const bool tgt = lm_eval<WIDE, L2, true>(p, pr, cc, x, out, &n_it);
    const double f0 = sqrt(patch_dot<WIDE>(out, out, rg));
    if (active) {
      double* dst = sp.fvec0 + (size_t)s * (LM_ROWS * 16) + c;
#pragma unroll
      for (int y = 0; y < RL; ++y) dst[y * 16] = out[y];
      if (lead) {
        const u32 key = n_it < LM_SPLIT_BINS - 1 ? (u32)n_it : (u32)(LM_SPLIT_BINS - 1);
        sp.fnorm0[s] = f0;
        sp.meta[s] = key | (tgt ? 0x100u : 0u);
        atomicAdd(&sp.hist[lm_split_stripe(s) * LM_SPLIT_BINS + key], 1u);
      }
    }
    return;
  }
  if constexpr (STAGE == 2) {  // resume behind minimizeInit (the `phase == 0` branch and the step to phase 1 below)
    if (active) {
      const double* src = sp.fvec0 + (size_t)s * (LM_ROWS * 16) + c;
#pragma unroll
      for (int y = 0; y < RL; ++y) fvec[y] = src[y * 16];
      fnorm = sp.fnorm0[s];
      fvec_tight = (sp.meta[s] & 0x100u) != 0;
    } else {
#pragma unroll
      for (int y = 0; y < RL; ++y) fvec[y] = 0.0;
    }
    par = 0.;
    iter = 1;
    phase = 1;
    h = sqrt_eps * fabs(x);
    if (h == 0.) h = sqrt_eps;
    xe = x + h;
  }
  DEV_LM_JAC_DECL
  while (true) {
    if (need_step) {  // determine the LM parameter and the trial point (minimizeOneStep, inner loop head)
      const double pstep = lm_lmpar2(r, diag, qtf, delta, par);
      wa1 = -pstep;
      xnew = x + wa1;
      pnorm = fabs(diag * wa1);
      if (iter == 1) delta = (pnorm < delta) ? pnorm : delta;
      xe = xnew;
      need_step = false;
    }
    bool out_tight;
    if constexpr (PAIR) {
      const bool reuse = phase == 1 && have_spec && __double_as_longlong(xe) == __double_as_longlong(xs);
      if (reuse) {  // F(x + h): wave 1 evaluated it beside the trial point
#pragma unroll
        for (int y = 0; y < RL; ++y) out[y] = lds_pair[xpar][1][y][ln];
        out_tight = lds_pair_tight[xpar][1] != 0;
        have_spec = false;
      } else {
        const bool side = phase != 1;  // beside F(xe): the difference point that follows if the driver moves to xe
        if (side) {
          double hs = sqrt_eps * fabs(xe);
          if (hs == 0.) hs = sqrt_eps;
          xs = xe + hs;
        }
        const bool t_mine = lm_eval<WIDE, L2>(p, pr, cc, (side && wv) ? xs : xe, out);
        out_tight = t_mine;
        if (side) {
          xpar ^= 1;
#pragma unroll
          for (int y = 0; y < RL; ++y) lds_pair[xpar][wv][y][ln] = out[y];
          if (ln == 0) lds_pair_tight[xpar][wv] = t_mine ? 1 : 0;
          __syncthreads();  // both waves run the same driver on the same state: they arrive here together
#pragma unroll
          for (int y = 0; y < RL; ++y) out[y] = lds_pair[xpar][0][y][ln];
          out_tight = lds_pair_tight[xpar][0] != 0;
          have_spec = true;
        }
      }
    } else {
      out_tight = lm_eval<WIDE, L2, false, BAND>(p, pr, cc, xe, out, nullptr, &viol);
    }
    int status = -1;
    bool outer_tail = false;
    if (phase == 0) {  // minimizeInit
      fvec_tight = out_tight;
#pragma unroll
      for (int y = 0; y < RL; ++y) fvec[y] = out[y];
      fnorm = sqrt(patch_dot<WIDE>(fvec, fvec, rg));
      par = 0.;
      iter = 1;
    } else if (phase == 1) {
      DEV_LM_JAC_PASS(pr, active, x);
      // NumericalDiff<Forward>::df: the reference evaluates F(x) again (val1) and F(x + h); F is a
      // pure function and fvec already holds F(x) at the current x, so val1 == fvec bit for bit and
      // only F(x + h) is computed.  nfev still advances by 2 (it drives the maxfev test).
      double fjac[RL];
      {
        // Both evaluations tight: every entry is 0 or has 2^-150 <= |f| < 2^11, so a non-zero difference has
        // 2^-202 <= |d| < 2^12; with h inside the window too the seven quotients share one refined reciprocal (fdiv.hpp:
        // bit for bit the IEEE quotient), 3 instead of 13 instructions each.
        const Recip rh = make_recip(h);
        if (out_tight && fvec_tight && rh.fast) {
#pragma unroll
          for (int y = 0; y < RL; ++y) fjac[y] = div_fast(out[y] - fvec[y], rh);
        } else {
#pragma unroll
          for (int y = 0; y < RL; ++y) fjac[y] = (out[y] - fvec[y]) / h;
        }
      }
      nfev += 2;
      const double wa2n = sqrt(patch_dot<WIDE>(fjac, fjac, rg));
      const double jtf = patch_dot<WIDE>(fjac, fvec, rg);
      r = wa2n;
      const double fvec0 = WIDE ? bcast_f64(fvec[0], 0) : __shfl(fvec[0], 0, 16);  // element (0, 0) of the patch
      qtf = (r != 0.) ? jtf / r : fvec0;
      if (iter == 1) {
        diag = (wa2n == 0.) ? 1. : wa2n;
        xnorm = fabs(diag * x);
        delta = factor * xnorm;
        if (delta == 0.) delta = factor;
      }
      gnorm = 0.;
      if (fnorm != 0.)
        if (wa2n != 0.) { const double g = fabs(r * (qtf / fnorm) / wa2n); gnorm = (gnorm < g) ? g : gnorm; }
      if (gnorm <= gtol) {
        status = 4;
        outer_tail = true;
      } else {
        diag = (diag < wa2n) ? wa2n : diag;
        need_step = true;
        phase = 2;
      }
    } else {  // phase 2: trust-region trial at xnew
      ++nfev;
      const double fnorm1 = sqrt(patch_dot<WIDE>(out, out, rg));
      double actred = -1.;
      if (0.1 * fnorm1 < fnorm) actred = 1. - (fnorm1 / fnorm) * (fnorm1 / fnorm);
      const double wa3 = r * wa1;
      const double t1 = fabs(wa3) / fnorm, temp1 = t1 * t1;
      const double t2 = sqrt(par) * pnorm / fnorm, temp2 = t2 * t2;
      const double prered = temp1 + temp2 / 0.5;
      const double dirder = -(temp1 + temp2);
      double ratio = 0.;
      if (prered != 0.) ratio = actred / prered;
      if (ratio <= 0.25) {
        double temp = 0.5;
        if (actred >= 0.) temp = 0.5;
        if (actred < 0.) temp = 0.5 * dirder / (dirder + 0.5 * actred);
        if (0.1 * fnorm1 >= fnorm || temp < 0.1) temp = 0.1;
        const double pn = pnorm / 0.1;
        delta = temp * ((pn < delta) ? pn : delta);
        par /= temp;
      } else if (!(par != 0. && ratio < 0.75)) {
        delta = pnorm / 0.5;
        par = 0.5 * par;
      }
      if (ratio >= 1e-4) {
        x = xnew;
        fvec_tight = out_tight;
#pragma unroll
        for (int y = 0; y < RL; ++y) fvec[y] = out[y];
        xnorm = fabs(diag * x);
        fnorm = fnorm1;
        ++iter;
      }
      if (fabs(actred) <= ftol && prered <= ftol && 0.5 * ratio <= 1. && delta <= xtol * xnorm) status = 3;
      else if (fabs(actred) <= ftol && prered <= ftol && 0.5 * ratio <= 1.) status = 1;
      else if (delta <= xtol * xnorm) status = 2;
      else if (nfev >= maxfev) status = 5;
      else if (fabs(actred) <= eps && prered <= eps && 0.5 * ratio <= 1.) status = 6;
      else if (delta <= eps * xnorm) status = 7;
      else if (gnorm <= eps) status = 8;
      if (status >= 0 || !(ratio < 1e-4)) outer_tail = true;  // minimizeOneStep returns (status or Running)
      else need_step = true;                                   // do { ... } while (ratio < 1e-4)
    }
    if (phase == 0) {
      phase = 1;
    } else if (outer_tail) {
      // ---- the reference's outer loop, DepthProblemSolver.cpp:161-188 ----
      iteration++;
      if (iteration >= p.lm_max_iter) break;
      if (status == 2 || status == 3) {
        if (optState == 0) optState++;
        else break;
      }
      phase = 1;
    }
    if (phase == 1) {  // next evaluation: F(x + h) for the forward difference
      h = sqrt_eps * fabs(x);
      if (h == 0.) h = sqrt_eps;
      xe = x + h;
    }
  }

  if (probe) {
    const u64* sc = a.clk + CLK_SCRATCH + 2 * (size_t)(blockIdx.x / CLK_STRIDE);
    const u64 c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    u32 xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
    xcc &= CLK_XCDS - 1;
    atomicAdd(&a.clk[2 * xcc], c1 - sc[0]);
    atomicAdd(&a.clk[2 * xcc + 1], r1 - sc[1]);
    atomicAdd(&a.clk[CLK_SAMPLES], 1ull);
  }
  if (!active || !lead) return;
  if constexpr (BAND) { if (viol) atomicAdd(a.halo_viol, 1u); }
  const bool solved = !(x <= 0.001);  // DepthProblemSolver.cpp:192
  bool keep = solved;
  if (solved) {
    atomicAdd(n_solved, 1u);
    const double invJtJ = (r != 0.) ? (1. / r) * (1. / r) : 0.;  // internal::covar, n == 1
    double variance;
    if constexpr (L2) {  // :200-206: cov = |f|^2 / (values - inputs) * (J^T J)^-1; DepthPoint::update on a new point bounds it
      // (upstream takes lm.fvec.blueNorm() here and lm.fnorm -- stableNorm -- for the residual below: two Eigen algorithms for
      // the same |f| that may round differently in the last place.  Both are third-party code absent from the reference tree;
      // this path and the oracle use the solver's one canonical |f| for both.  Stated deviation, DESIGN.md "Deviations".)
      variance = fnorm * fnorm / (double)(N - 1) * invJtJ;
      if (variance < 1e-6) variance = 1e-6;
    } else {
      variance = p.td_stdvar2 * invJtJ;                          // :210
    }
    const double residual = fnorm * fnorm;                       // :212
    DevPoint o;
    o.row = (u32)(size_t)floor(pr.cy);  // :116
    o.col = (u32)(size_t)floor(pr.cx);
    o.x[0] = pr.cx;
    o.x[1] = pr.cy;
    cam2World(p.camL, pr.cx, pr.cy, x, o.p_cam);                 // :119
    DEV_PERTURB_POINT(o, s);  // (empty in the product: dev_hooks.hpp)
    o.inv_depth = x;                                             // update_studentT, new-point branch

    o.scale2 = L2 ? 0.0 : variance * (p.td_nu - 2) / p.td_nu;    // :125 (l2: the Gaussian update leaves scaleSquared_ / nu_
    o.nu = L2 ? 0.0 : p.td_nu;                                   //  as constructed -- zero here and in the oracle, Appendix A-8)
    o.variance = variance;
    o.residual = residual;
    o.age = 0;
    o.pose_idx = m.pose_idx;
    o.seq = j;
    if (a.cull)  // pointCulling, :230-234
      keep = variance <= p.var_thr && residual <= p.cost_thr && x > -1e-6 && x >= p.invdepth_min && x <= p.invdepth_max;
    if (keep) a.out_slots[s] = o;
  }
  a.out_flags[s] = keep ? 1u : 0u;
}

// ---- the persistent narrow layout (round 5) ------------------------------------------------------------------------------------
// In lm_refine_kernel<false> four matches share a wave from its first instruction to its last: the wave performs as many evaluations
// as the SLOWEST of its four matches needs (22.9 on the headline workload where a match needs 19.2: profiles/r05_lm_attribution.txt),
// and a launch needs one wave per four slots of its bound (~49 000 waves for ~10 000 that find a match).  Here a 16-lane group that
// has finished its match fetches the next one from a counter and starts over while the other three groups carry on: the grid is
// what the chip holds (two waves per SIMD), every evaluation a wave executes serves four live matches until the list runs dry,
// and the groups' different lengths average out inside a wave instead of idling its lanes.  What stays is the lockstep INSIDE an
// evaluation (the t-scale loop runs as long as the slowest of the four groups needs).
// Same arithmetic, same operands, same order for every match as in lm_refine_kernel (the evaluator and the solver's state
// machine are the same code): which wave solves a match, and beside which others, does not enter its result -- bit-identical
// output (tests/test_gpu_parity.py, test_gpu_fullsize.py compare with the oracle element by element).
// The per-match set-up moves INTO the loop, where the solver's whole state is live: T_left_virtual therefore goes to LDS (one
// element per lane, 12 lanes of the group) instead of 24 VGPRs, and its 3 x 4 products are formed by those 12 lanes.
template <bool L2>
__global__ void __launch_bounds__(LM_BLOCK, LM_WAVES) lm_refine_persist_kernel(LmArgs a, DevParams p, u32* n_solved, u32* next) {
  constexpr bool WIDE = false;
  constexpr int RL = LM_ROWS;
  __shared__ double lds_T[LM_BLOCK / 16][12];
  __shared__ double lds_cam[28];
  if (threadIdx.x < 27) {
    const int i = threadIdx.x;
    lds_cam[i] = i < 12 ? p.camL.P[i] : (i < 24 ? p.camR.P[i - 12] : p.camL.Kinv_t[i - 24]);
  }
  __syncthreads();
  const int c = threadIdx.x & 15;
  const bool lead = c == 0;
  u32 M = *a.n_matches;
  if (M > a.max_matches) M = a.max_matches;
  const u32 n_groups = gridDim.x * (LM_BLOCK / 16);
  {  // every slot of the launch gets its flag (no memset): the slots beyond the match count, dealt to all threads of the grid
    const u32 gid = blockIdx.x * LM_BLOCK + threadIdx.x, nth = gridDim.x * LM_BLOCK;
    for (u32 i = M + gid; i < a.max_matches; i += nth) a.out_flags[i] = 0u;
  }
  const bool probe = a.clk != nullptr && threadIdx.x == 0 && blockIdx.x % CLK_STRIDE == 0u;
  if (probe) {
    u64* sc = a.clk + CLK_SCRATCH + 2 * (size_t)(blockIdx.x / CLK_STRIDE);
    sc[0] = __builtin_readcyclecounter();
    sc[1] = __builtin_amdgcn_s_memrealtime();
  }
  LmProblemT<true> pr;
  {
    const int n_bytes = p.W * p.H;
    pr.tsL = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a.tsL), 0, n_bytes, 0x00020000);
    pr.tsR = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a.tsR), 0, n_bytes, 0x00020000);
  }
  pr.cam = lds_cam;
  pr.c = c;
  pr.rg = 0;
  pr.vy0 = 0; pr.vy1 = p.H;
  double* Tm = lds_T[threadIdx.x >> 4];
  pr.T = Tm;
  pr.cx = pr.cy = -1e9;
  pr.ray[0] = pr.ray[1] = pr.ray[2] = 0.;
  const int N = LM_ROWS * LM_COLS;
  const double ftol = 1e-6, xtol = 1e-6, gtol = 0., factor = 100.;
  const double eps = 2.220446049250313e-16;
  const double sqrt_eps = 1.4901161193847656e-08;  // sqrt(DBL_EPSILON), exact
  const int maxfev = p.lm_maxfev;
  double x = 1.0;
  double fvec[RL], out[RL];
  LmCache<WIDE> cc;
  cc.clear();
  double fnorm = 0., par = 0., diag = 0., xnorm = 0., delta = 0., r = 0., qtf = 0., gnorm = 0.;
  double h = 0., xnew = 0., wa1 = 0., pnorm = 0.;
  int nfev = 1, iter = 1, iteration = 0, optState = 0;
  int phase = 0;
  bool need_step = false, fvec_tight = false;
  double xe = x;
  u32 s = (blockIdx.x * LM_BLOCK + threadIdx.x) >> 4;  // the group's first match: its own position; later ones: n_groups + counter
  u32 j = 0, pose_idx = 0;
  bool need_new = true, first = true;
  while (true) {
    if (need_new) {  // (group-uniform)
      if (!first) {
        u32 t = 0;
        if (lead) t = atomicAdd(next, 1u);
        s = n_groups + __shfl(t, 0, 16);
      }
      first = false;
      need_new = false;
      if (s >= M) break;  // the list is exhausted: the group retires (the wave ends when its four groups have)
      j = a.dense ? s : stride_item(s, M, (u32)p.num_threads);  // DepthProblemSolver.cpp:90
      const esvo_match_t m = a.matches[j];
      pose_idx = m.pose_idx;
      pr.cx = m.x_left[0];
      pr.cy = m.x_left[1];
#pragma unroll
      for (int r3 = 0; r3 < 3; ++r3)
        pr.ray[r3] = (p.camL.Kinv[r3 * 3 + 0] * pr.cx + p.camL.Kinv[r3 * 3 + 1] * pr.cy) + p.camL.Kinv[r3 * 3 + 2];
      DEV_LM_SET_SLOT(pr, s);
      {  // DepthProblem::setProblem, DepthProblem.cpp:17-32: T_left_virtual = T_world_obs^-1 T_world_virtual, element c by lane c < 12
         // (mat4_mul's association: ((a0 b0 + a1 b1) + a2 b2) + a3 b3)
        double Tlw[16];
        rigid_inverse(a.T_world_obs, Tlw);
        const int ti = (c >> 2) < 3 ? (c >> 2) : 0, tj = c & 3;
        const double a0 = ti == 0 ? Tlw[0] : (ti == 1 ? Tlw[4] : Tlw[8]), a1 = ti == 0 ? Tlw[1] : (ti == 1 ? Tlw[5] : Tlw[9]);
        const double a2 = ti == 0 ? Tlw[2] : (ti == 1 ? Tlw[6] : Tlw[10]), a3 = ti == 0 ? Tlw[3] : (ti == 1 ? Tlw[7] : Tlw[11]);
        const double* B = a.pose_T + (size_t)pose_idx * 16;
        const double v = ((a0 * B[0 * 4 + tj] + a1 * B[1 * 4 + tj]) + a2 * B[2 * 4 + tj]) + a3 * B[3 * 4 + tj];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the previous match's reads of Tm are done (same wave: LDS is in order)
        if (c < 12) Tm[c] = v;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
      x = m.inv_depth;
      cc.clear();
      fnorm = par = diag = xnorm = delta = r = qtf = gnorm = 0.;
      h = xnew = wa1 = pnorm = 0.;
      nfev = 1; iter = 1; iteration = 0; optState = 0;
      phase = 0;
      need_step = false;
      fvec_tight = false;
      xe = x;
    }
    if (need_step) {  // determine the LM parameter and the trial point (minimizeOneStep, inner loop head)
      const double pstep = lm_lmpar2(r, diag, qtf, delta, par);
      wa1 = -pstep;
      xnew = x + wa1;
      pnorm = fabs(diag * wa1);
      if (iter == 1) delta = (pnorm < delta) ? pnorm : delta;
      xe = xnew;
      need_step = false;
    }
    const bool out_tight = lm_eval<WIDE, L2, false, false, true>(p, pr, cc, xe, out);
    int status = -1;
    bool outer_tail = false;
    if (phase == 0) {  // minimizeInit
      fvec_tight = out_tight;
#pragma unroll
      for (int y = 0; y < RL; ++y) fvec[y] = out[y];
      fnorm = sqrt(patch_dot<WIDE>(fvec, fvec, 0));
      par = 0.;
      iter = 1;
    } else if (phase == 1) {  // NumericalDiff<Forward>::df (see lm_refine_kernel)
      double fjac[RL];
      {
        const Recip rh = make_recip(h);
        if (out_tight && fvec_tight && rh.fast) {
#pragma unroll
          for (int y = 0; y < RL; ++y) fjac[y] = div_fast(out[y] - fvec[y], rh);
        } else {
#pragma unroll
          for (int y = 0; y < RL; ++y) fjac[y] = (out[y] - fvec[y]) / h;
        }
      }
      nfev += 2;
      const double wa2n = sqrt(patch_dot<WIDE>(fjac, fjac, 0));
      const double jtf = patch_dot<WIDE>(fjac, fvec, 0);
      r = wa2n;
      const double fvec0 = __shfl(fvec[0], 0, 16);  // element (0, 0) of the patch
      qtf = (r != 0.) ? jtf / r : fvec0;
      if (iter == 1) {
        diag = (wa2n == 0.) ? 1. : wa2n;
        xnorm = fabs(diag * x);
        delta = factor * xnorm;
        if (delta == 0.) delta = factor;
      }
      gnorm = 0.;
      if (fnorm != 0.)
        if (wa2n != 0.) { const double g = fabs(r * (qtf / fnorm) / wa2n); gnorm = (gnorm < g) ? g : gnorm; }
      if (gnorm <= gtol) {
        status = 4;
        outer_tail = true;
      } else {
        diag = (diag < wa2n) ? wa2n : diag;
        need_step = true;
        phase = 2;
      }
    } else {  // phase 2: trust-region trial at xnew
      ++nfev;
      const double fnorm1 = sqrt(patch_dot<WIDE>(out, out, 0));
      double actred = -1.;
      if (0.1 * fnorm1 < fnorm) actred = 1. - (fnorm1 / fnorm) * (fnorm1 / fnorm);
      const double wa3 = r * wa1;
      const double t1 = fabs(wa3) / fnorm, temp1 = t1 * t1;
      const double t2 = sqrt(par) * pnorm / fnorm, temp2 = t2 * t2;
      const double prered = temp1 + temp2 / 0.5;
      const double dirder = -(temp1 + temp2);
      double ratio = 0.;
      if (prered != 0.) ratio = actred / prered;
      if (ratio <= 0.25) {
        double temp = 0.5;
        if (actred >= 0.) temp = 0.5;
        if (actred < 0.) temp = 0.5 * dirder / (dirder + 0.5 * actred);
        if (0.1 * fnorm1 >= fnorm || temp < 0.1) temp = 0.1;
        const double pn = pnorm / 0.1;
        delta = temp * ((pn < delta) ? pn : delta);
        par /= temp;
      } else if (!(par != 0. && ratio < 0.75)) {
        delta = pnorm / 0.5;
        par = 0.5 * par;
      }
      if (ratio >= 1e-4) {
        x = xnew;
        fvec_tight = out_tight;
#pragma unroll
        for (int y = 0; y < RL; ++y) fvec[y] = out[y];
        xnorm = fabs(diag * x);
        fnorm = fnorm1;
        ++iter;
      }
      if (fabs(actred) <= ftol && prered <= ftol && 0.5 * ratio <= 1. && delta <= xtol * xnorm) status = 3;
      else if (fabs(actred) <= ftol && prered <= ftol && 0.5 * ratio <= 1.) status = 1;
      else if (delta <= xtol * xnorm) status = 2;
      else if (nfev >= maxfev) status = 5;
      else if (fabs(actred) <= eps && prered <= eps && 0.5 * ratio <= 1.) status = 6;
      else if (delta <= eps * xnorm) status = 7;
      else if (gnorm <= eps) status = 8;
      if (status >= 0 || !(ratio < 1e-4)) outer_tail = true;  // minimizeOneStep returns (status or Running)
      else need_step = true;                                   // do { ... } while (ratio < 1e-4)
    }
    bool finished = false;
    if (phase == 0) {
      phase = 1;
    } else if (outer_tail) {  // the reference's outer loop, DepthProblemSolver.cpp:161-188
      iteration++;
      if (iteration >= p.lm_max_iter) finished = true;
      else if (status == 2 || status == 3) {
        if (optState == 0) optState++;
        else finished = true;
      }
      if (!finished) phase = 1;
    }
    if (finished) {  // the match is solved: its point (lm_refine_kernel's epilogue), then the group asks for the next match
      if (lead) {
        const bool solved = !(x <= 0.001);  // DepthProblemSolver.cpp:192
        bool keep = solved;
        if (solved) {
          atomicAdd(n_solved, 1u);
          const double invJtJ = (r != 0.) ? (1. / r) * (1. / r) : 0.;  // internal::covar, n == 1
          double variance;
          if constexpr (L2) {
            variance = fnorm * fnorm / (double)(N - 1) * invJtJ;
            if (variance < 1e-6) variance = 1e-6;
          } else {
            variance = p.td_stdvar2 * invJtJ;                          // :210
          }
          const double residual = fnorm * fnorm;                       // :212
          DevPoint o;
          o.row = (u32)(size_t)floor(pr.cy);  // :116
          o.col = (u32)(size_t)floor(pr.cx);
          o.x[0] = pr.cx;
          o.x[1] = pr.cy;
          cam2World(p.camL, pr.cx, pr.cy, x, o.p_cam);                 // :119
          DEV_PERTURB_POINT(o, s);
          o.inv_depth = x;
          o.scale2 = L2 ? 0.0 : variance * (p.td_nu - 2) / p.td_nu;    // :125
          o.nu = L2 ? 0.0 : p.td_nu;
          o.variance = variance;
          o.residual = residual;
          o.age = 0;
          o.pose_idx = pose_idx;
          o.seq = j;
          if (a.cull)  // pointCulling, :230-234
            keep = variance <= p.var_thr && residual <= p.cost_thr && x > -1e-6 && x >= p.invdepth_min && x <= p.invdepth_max;
          if (keep) a.out_slots[s] = o;
        }
        a.out_flags[s] = keep ? 1u : 0u;
      }
      need_new = true;
      continue;
    }
    if (phase == 1) {  // next evaluation: F(x + h) for the forward difference
      h = sqrt_eps * fabs(x);
      if (h == 0.) h = sqrt_eps;
      xe = x + h;
    }
  }
  if (probe) {
    const u64* sc = a.clk + CLK_SCRATCH + 2 * (size_t)(blockIdx.x / CLK_STRIDE);
    const u64 c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    u32 xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
    xcc &= CLK_XCDS - 1;
    atomicAdd(&a.clk[2 * xcc], c1 - sc[0]);
    atomicAdd(&a.clk[2 * xcc + 1], r1 - sc[1]);
    atomicAdd(&a.clk[CLK_SAMPLES], 1ull);
  }
}

// counting sort of the slots by the cost of F(x0), most expensive first: offsets from the (complete) histogram, computed
// per block in LDS; the order inside a bin is whatever the atomics make it (it does not matter: see above)
__global__ void __launch_bounds__(256) lm_order_kernel(const u32* __restrict__ n_matches, u32 max_matches, LmSplit sp) {
  // base[stripe][key]: first position of the (stripe, key) sub-list -- bins in descending key order, stripes ascending
  // inside a bin.  Every block derives the table from the (complete) histogram: 2048 cached words.
  __shared__ u32 base[LM_SPLIT_STRIPES][LM_SPLIT_BINS];
  __shared__ u32 tot[LM_SPLIT_BINS];
  if (threadIdx.x < LM_SPLIT_BINS) {
    const int k = threadIdx.x;
    u32 run = 0;
    for (int st = 0; st < LM_SPLIT_STRIPES; ++st) {
      base[st][k] = run;
      run += sp.hist[st * LM_SPLIT_BINS + k];
    }
    tot[k] = run;
  }
  __syncthreads();
  u32 off = 0;  // matches in bins above the thread's own (threads >= BINS only help with the adds below)
  if (threadIdx.x < LM_SPLIT_BINS)
    for (int k = LM_SPLIT_BINS - 1; k > (int)threadIdx.x; --k) off += tot[k];
  __syncthreads();
  if (threadIdx.x < LM_SPLIT_BINS)
    for (int st = 0; st < LM_SPLIT_STRIPES; ++st) base[st][threadIdx.x] += off;
  __syncthreads();
  u32 M = *n_matches;
  if (M > max_matches) M = max_matches;
  const u32 s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= M) return;
  const u32 key = sp.meta[s] & 0xffu, st = lm_split_stripe(s);
  u32* fill = sp.hist + LM_SPLIT_STRIPES * LM_SPLIT_BINS;
  sp.order[base[st][key] + atomicAdd(&fill[st * LM_SPLIT_BINS + key], 1u)] = s;
}

void launch_lm_refine_any(const LmArgs& a, const DevParams& p, u32* n_solved, hipStream_t s);  // kernels_lm_any.hip
// whether a launch bounded by max_matches takes the one-wave-per-match layout -- the one that reads LmArgs::match_index
bool lm_launch_is_wide(u32 max_matches, const DevParams& p) {
  return max_matches > 0 && p.wx == LM_COLS && p.wy == LM_ROWS && p.ls_norm != ESVO_LSNORM_L2 && max_matches <= LM_WIDE_MAX && LM_BLOCK == 64;
}
void launch_lm_refine(const LmArgs& a, const DevParams& p, u32* n_solved, hipStream_t s) {
  if (a.max_matches == 0) return;
  if (p.wx != LM_COLS || p.wy != LM_ROWS) { launch_lm_refine_any(a, p, n_solved, s); return; }  // (no clock probe, no layouts)
  LmSplit sp;
  sp.fvec0 = a.split_fvec0; sp.fnorm0 = a.split_fnorm0; sp.meta = a.split_meta; sp.order = a.split_order; sp.hist = a.split_hist;
  const u32 groups_per_block = LM_BLOCK / 16;
  const u32 blocks = (a.max_matches + groups_per_block - 1) / groups_per_block;
  if (p.ls_norm == ESVO_LSNORM_L2) {  // no shipped configuration: the narrow layout only
    if (a.halo_viol) hipLaunchKernelGGL((lm_refine_kernel<false, true, 0, false, true>), dim3(blocks), dim3(LM_BLOCK), 0, s, a, p, n_solved, sp);
    else hipLaunchKernelGGL((lm_refine_kernel<false, true, 0>), dim3(blocks), dim3(LM_BLOCK), 0, s, a, p, n_solved, sp);
    return;
  }
  if (a.halo_viol) {  // routed band mode (the observation pair holds the band's rows only): the guarded kernels, one launch
    if (a.max_matches <= LM_WIDE_MAX && LM_BLOCK == 64)
      hipLaunchKernelGGL((lm_refine_kernel<true, false, 0, false, true>), dim3(a.max_matches), dim3(64), 0, s, a, p, n_solved, sp);
    else
      hipLaunchKernelGGL((lm_refine_kernel<false, false, 0, false, true>), dim3(blocks), dim3(LM_BLOCK), 0, s, a, p, n_solved, sp);
    return;
  }
  // the match count lives on the device; the layout is chosen by the launch's bound (the events handed to block matching)
  if (a.max_matches <= LM_WIDE_MAX && LM_BLOCK == 64) {
    if (a.pair)
      hipLaunchKernelGGL((lm_refine_kernel<true, false, 0, true>), dim3(a.max_matches), dim3(128), 0, s, a, p, n_solved, sp);
    else
      hipLaunchKernelGGL((lm_refine_kernel<true, false, 0>), dim3(a.max_matches), dim3(64), 0, s, a, p, n_solved, sp);
    return;
  }
  if (a.split_fvec0 && !a.dense) {  // the split launch (see LmSplit)
    hipLaunchKernelGGL((lm_refine_kernel<false, false, 1>), dim3(blocks), dim3(LM_BLOCK), 0, s, a, p, n_solved, sp);
    hipLaunchKernelGGL(lm_order_kernel, dim3((a.max_matches + 255) / 256), dim3(256), 0, s, a.n_matches, a.max_matches, sp);
    hipLaunchKernelGGL((lm_refine_kernel<false, false, 2>), dim3(blocks), dim3(LM_BLOCK), 0, s, a, p, n_solved, sp);
    return;
  }
  if (a.persist_next) {  // the persistent layout: as many workgroups as the chip holds beside the other stages (two waves per SIMD)
    const u32 grid = blocks < a.persist_blocks ? blocks : a.persist_blocks;
    hipLaunchKernelGGL((lm_refine_persist_kernel<false>), dim3(grid), dim3(LM_BLOCK), 0, s, a, p, n_solved, a.persist_next);
    return;
  }
  hipLaunchKernelGGL((lm_refine_kernel<false, false, 0>), dim3(blocks), dim3(LM_BLOCK), 0, s, a, p, n_solved, sp);
}

// stable compaction of the solver slots into a frame buffer
__global__ void __launch_bounds__(256) compact_points_kernel(const DevPoint* __restrict__ slots, const u32* __restrict__ flags,
                                                             const u32* __restrict__ prefix, const u32* __restrict__ n_in,
                                                             u32 max_n, DevPoint* __restrict__ out) {
  const u32 s = blockIdx.x * blockDim.x + threadIdx.x;
  u32 n = *n_in;
  if (n > max_n) n = max_n;
  if (s >= n || !flags[s]) return;
  DevPoint o = slots[s];
  o.seq = prefix[s];
  out[prefix[s]] = o;
}
void launch_compact_points(const DevPoint* slots, const u32* flags, const u32* prefix, const u32* n_in, u32 max_n,
                           DevPoint* out, hipStream_t s) {
  if (max_n == 0) return;
  hipLaunchKernelGGL(compact_points_kernel, dim3((max_n + 255) / 256), dim3(256), 0, s, slots, flags, prefix, n_in, max_n,
                     out);
}

}  // namespace esvo
