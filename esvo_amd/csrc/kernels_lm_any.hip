// kernels_lm_any.hip — the inverse-depth refinement for ANY patch size (patch_size_X x patch_size_Y: a run-time parameter of
// the reference, code default 25 x 25 -- esvo_core/src/esvo_Mapping.cpp:38-39,93-94,127; every shipped configuration sets
// 15 x 7, which kernels_lm.hip serves with its register layouts).
//
// Same algorithm, same order of every floating-point operation as kernels_lm.hip and the oracle's canonical mode
// (DepthProblem::operator() / warping / patchInterpolation, DepthProblem.cpp:34-262; Eigen's LevenbergMarquardt over
// NumericalDiff for one unknown, DepthProblemSolver.cpp:138-214; pointCulling :216-244) -- written for generality, not for
// speed: ONE wave per match, lane = patch column (wx <= 64), the rows in a loop, the three per-element arrays of the solver
// (raw residuals, F(x), the evaluation in flight) in LDS as [row][64 lanes] (no bank conflicts), plain IEEE division and
// square root everywhere.  Patch sums reduce in the canonical order: per column the sequential sum over the rows top to
// bottom, then the xor butterfly over the columns padded with zeros to P = the next power of two (oracle: reduce_patch).
#include "common.hpp"
#include "lm_common.hpp"

namespace esvo {

extern __shared__ __attribute__((aligned(16))) double lm_any_smem[];

struct AnyProblem {
  double cx, cy;
  double T[12];  // T_left_virtual (3x4)
  const uint8_t* tsL;
  const uint8_t* tsR;
  int wx, wy, P, c;
  int vy0, vy1;  // rows of the observation pair that hold data (routed band mode; 0, H otherwise)
};

__device__ inline double any_butterfly(double a, int P) {
  for (int w = 1; w < P; w <<= 1) a = a + __shfl_xor(a, w, 64);
  return a;
}
__device__ inline int any_butterfly_int(int a, int P) {
  for (int w = 1; w < P; w <<= 1) a = a + __shfl_xor(a, w, 64);
  return a;
}
__device__ inline double any_butterfly_min(double a, int P) {
  for (int w = 1; w < P; w <<= 1) a = fmin(a, __shfl_xor(a, w, 64));
  return a;
}
// canonical sum over the patch of a[y][c] * b[y][c]
__device__ inline double any_dot(const double* a, const double* b, int wy, int c, int P) {
  double s = a[c] * b[c];
  for (int y = 1; y < wy; ++y) s = s + a[y * 64 + c] * b[y * 64 + c];
  return any_butterfly(s, P);
}

struct AnyGeom { int ulx, uly; double q1, q2, q3, q4; bool ok; };
__device__ inline AnyGeom any_geom(const DevParams& p, int wx, int wy, double lx, double ly) {  // DepthProblem.cpp:193-230
  AnyGeom g;
  const int hx = (wx - 1) / 2, hy = (wy - 1) / 2;
  const double fx = floor(lx), fy = floor(ly);
  const int l1 = (int)fx, l0 = (int)fy;
  g.ulx = l1 - hx;
  g.uly = l0 - hy;
  const int drx = l1 + hx, dry = l0 + hy;
  g.ok = g.ulx >= 0 && g.uly >= 0 && drx < p.W && dry < p.H && g.uly + wy < p.H && g.ulx + wx < p.W;
  g.q1 = (fx + 1.0) - lx;  // (double)(l1 + 1) - lx: exact either way
  g.q2 = lx - fx;
  g.q3 = (fy + 1.0) - ly;
  g.q4 = ly - fy;
  return g;
}

// DepthProblem::operator(): fv[y][c] = residual of patch element (y, c) (0 in the padding lanes); rr = scratch for the raw
// residuals.  L2: LSnorm "l2" (the plain temporal residual, 255 on failure).
template <bool L2>
__device__ void any_eval(const DevParams& p, const AnyProblem& pr, double x, double* fv, double* rr, bool& viol) {
  const int wx = pr.wx, wy = pr.wy, c = pr.c, P = pr.P;
  const bool el = c < wx;
  const double nu = p.td_nu;
  double prv[3], pl[3], x1u, x1v, x2u, x2v;
  cam2World(p.camL, pr.cx, pr.cy, x, prv);
#pragma unroll
  for (int r = 0; r < 3; ++r)
    pl[r] = ((pr.T[r * 4 + 0] * prv[0] + pr.T[r * 4 + 1] * prv[1]) + pr.T[r * 4 + 2] * prv[2]) + pr.T[r * 4 + 3];
  world2Cam(p.camL, pl, x1u, x1v);
  world2Cam(p.camR, pl, x2u, x2v);
  const int hx = (wx - 1) / 2, hy = (wy - 1) / 2;
  const double wlo = (double)hx, whi = (double)(p.W - hx), vlo = (double)hy, vhi = (double)(p.H - hy);
  // warping's bounds (:186-189; a NaN coordinate fails every comparison) and patchInterpolation's (:205-230)
  const bool inside = x1u >= wlo && x1u <= whi && x1v >= vlo && x1v <= vhi && x2u >= wlo && x2u <= whi && x2v >= vlo && x2v <= vhi;
  bool okw = inside;
  AnyGeom g1, g2;
  if (okw) {
    g1 = any_geom(p, wx, wy, x1u, x1v);
    g2 = any_geom(p, wx, wy, x2u, x2v);
    okw = g1.ok && g2.ok;
    // source rows uly .. uly + wy of both blocks must hold data (kernels_lm.hip, BAND)
    if (okw && (g1.uly < pr.vy0 || g1.uly + wy >= pr.vy1 || g2.uly < pr.vy0 || g2.uly + wy >= pr.vy1)) viol = true;
  }
  if (!okw) {  // failure fill, :49-56 / :149-155 (l2: :67-75, :143-147)
    double f = 255;
    if constexpr (!L2) {
      const double q = f / p.td_scale;
      const double weight = (nu + 1) / (nu + q * q);
      f = sqrt(weight) * f;
    }
    for (int y = 0; y < wy; ++y) fv[y * 64 + c] = el ? f : 0.0;
    return;
  }
  // bilinear interpolation, rows rolling: R[y] = q1 Src[y][c] + q2 Src[y][c + 1]; tau[y] = q3 R[y] + q4 R[y + 1]
  int knz = 0;
  double minabs = 1e300;
  {
    const uint8_t* s1 = pr.tsL + (size_t)g1.uly * p.W + g1.ulx + c;
    const uint8_t* s2 = pr.tsR + (size_t)g2.uly * p.W + g2.ulx + c;
    double R1 = 0, R2 = 0;
    if (el) {
      R1 = g1.q1 * (double)s1[0] + g1.q2 * (double)s1[1];
      R2 = g2.q1 * (double)s2[0] + g2.q2 * (double)s2[1];
    }
    for (int y = 0; y < wy; ++y) {
      double r = 0.0;
      if (el) {
        s1 += p.W;
        s2 += p.W;
        const double N1 = g1.q1 * (double)s1[0] + g1.q2 * (double)s1[1];
        const double N2 = g2.q1 * (double)s2[0] + g2.q2 * (double)s2[1];
        const double t1 = g1.q3 * R1 + g1.q4 * N1;
        const double t2 = g2.q3 * R2 + g2.q4 * N2;
        R1 = N1;
        R2 = N2;
        r = t1 - t2;
      }
      if constexpr (L2) fv[y * 64 + c] = r;
      else {
        rr[y * 64 + c] = r;
        if (r != 0) { knz++; minabs = fmin(minabs, fabs(r)); }
      }
    }
  }
  if constexpr (L2) return;
  knz = any_butterfly_int(knz, P);
  minabs = any_butterfly_min(minabs, P);
  const int N = wx * wy;
  const double scale2_0 = p.td_scale2;
  double s2;
  if ((double)knz * (nu + 1) / (double)N <= 0.94 && minabs >= 1e-6) {
    s2 = scale2_0;  // provable outcome of the reference's uncapped loop (kernels_lm.hip header)
  } else {          // DepthProblem.cpp:96-124: s1 <- s2 until |s2 - s1| / s1 <= 5 %
    double s1 = scale2_0;
    while (true) {
      double a = 0.0;
      for (int y = 0; y < wy; ++y) {
        const double r = rr[y * 64 + c];
        const double r2 = r * r;
        const double t = (r != 0) ? r2 * (nu + 1) / (nu + r2 / s1) : 0.0;
        a = y ? a + t : t;
      }
      const double sum = any_butterfly(a, P);
      if (sum == 0) { s2 = scale2_0; break; }
      s2 = sum / (double)N;
      if (!(fabs(s2 - s1) / s1 > 0.05)) break;
      s1 = s2;
    }
  }
  for (int y = 0; y < wy; ++y) {  // :127-135
    const double r = rr[y * 64 + c];
    fv[y * 64 + c] = sqrt((nu + 1) / (nu + r * r / s2)) * r;
  }
}

template <bool L2>
__global__ void __launch_bounds__(64) lm_refine_any_kernel(LmArgs a, DevParams p, u32* n_solved) {
  const int wx = p.wx, wy = p.wy;
  int P = 1;
  while (P < wx) P <<= 1;
  const int c = threadIdx.x;
  double* fvec = lm_any_smem;            // [wy][64] F at the current x
  double* out = fvec + (size_t)wy * 64;  // [wy][64] the evaluation in flight
  double* rr = out + (size_t)wy * 64;    // [wy][64] raw residuals of the evaluation in flight
  const u32 s = blockIdx.x;              // solver slot (thread-stride order); one match per wave
  u32 M = *a.n_matches;
  if (M > a.max_matches) M = a.max_matches;
  if (s >= M) {  // every slot of the launch gets its flag
    if (c == 0 && s < a.max_matches) a.out_flags[s] = 0u;
    return;
  }
  const u32 j = a.dense ? s : stride_item(s, M, (u32)p.num_threads);  // DepthProblemSolver.cpp:90 (dense: kernels_shard.hip)
  const esvo_match_t m = a.matches[j];
  AnyProblem pr;
  pr.cx = m.x_left[0];
  pr.cy = m.x_left[1];
  pr.tsL = a.tsL;
  pr.tsR = a.tsR;
  pr.wx = wx; pr.wy = wy; pr.P = P; pr.c = c;
  pr.vy0 = a.halo_viol ? a.vy0 : 0; pr.vy1 = a.halo_viol ? a.vy1 : p.H;
  bool viol = false;
  {  // DepthProblem::setProblem, DepthProblem.cpp:17-32
    double Tlw[16], Tlv[16];
    rigid_inverse(a.T_world_obs, Tlw);
    mat4_mul(Tlw, a.pose_T + (size_t)m.pose_idx * 16, Tlv);
#pragma unroll
    for (int i = 0; i < 12; ++i) pr.T[i] = Tlv[i];
  }
  const int N = wx * wy;
  const double ftol = 1e-6, xtol = 1e-6, gtol = 0., factor = 100.;
  const double eps = 2.220446049250313e-16;
  const double sqrt_eps = 1.4901161193847656e-08;  // sqrt(DBL_EPSILON), exact
  const int maxfev = p.lm_maxfev;

  // Eigen's minimizeInit / minimizeOneStep driven by the loop of DepthProblemSolver.cpp:161-188, as a state machine around one
  // call site of the evaluator (phase 0: F(x0); 1: F(x + h) of NumericalDiff; 2: F(x + p) of the trust-region trial) -- the
  // same control flow and arithmetic as kernels_lm.hip's driver.
  double x = m.inv_depth;
  double fnorm = 0., par = 0., diag = 0., xnorm = 0., delta = 0., r = 0., qtf = 0., gnorm = 0.;
  double h = 0., xnew = 0., wa1 = 0., pnorm = 0.;
  int nfev = 1, iter = 1, iteration = 0, optState = 0;
  int phase = 0;
  bool need_step = false;
  double xe = x;
  while (true) {
    if (need_step) {
      const double pstep = lm_lmpar2(r, diag, qtf, delta, par);
      wa1 = -pstep;
      xnew = x + wa1;
      pnorm = fabs(diag * wa1);
      if (iter == 1) delta = (pnorm < delta) ? pnorm : delta;
      xe = xnew;
      need_step = false;
    }
    any_eval<L2>(p, pr, xe, out, rr, viol);
    __syncthreads();  // one wave per workgroup: orders the LDS writes of all lanes before the cross-lane read of element (0, 0)
    int status = -1;
    bool outer_tail = false;
    if (phase == 0) {  // minimizeInit
      double* t = fvec; fvec = out; out = t;
      fnorm = sqrt(any_dot(fvec, fvec, wy, c, P));
      par = 0.;
      iter = 1;
    } else if (phase == 1) {  // NumericalDiff<Forward>::df: val1 == fvec (F is pure); nfev advances by 2
      double sjj, sjf;
      {
        const double f0 = (out[c] - fvec[c]) / h;
        sjj = f0 * f0;
        sjf = f0 * fvec[c];
        for (int y = 1; y < wy; ++y) {
          const double fj = (out[y * 64 + c] - fvec[y * 64 + c]) / h;
          sjj = sjj + fj * fj;
          sjf = sjf + fj * fvec[y * 64 + c];
        }
      }
      nfev += 2;
      const double wa2n = sqrt(any_butterfly(sjj, P));
      const double jtf = any_butterfly(sjf, P);
      r = wa2n;
      const double fvec0 = fvec[0];  // element (0, 0) of the patch
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
      const double fnorm1 = sqrt(any_dot(out, out, wy, c, P));
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
        double* t = fvec; fvec = out; out = t;
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
      if (status >= 0 || !(ratio < 1e-4)) outer_tail = true;
      else need_step = true;
    }
    if (phase == 0) {
      phase = 1;
    } else if (outer_tail) {  // the reference's outer loop, DepthProblemSolver.cpp:161-188
      iteration++;
      if (iteration >= p.lm_max_iter) break;
      if (status == 2 || status == 3) {
        if (optState == 0) optState++;
        else break;
      }
      phase = 1;
    }
    if (phase == 1) {
      h = sqrt_eps * fabs(x);
      if (h == 0.) h = sqrt_eps;
      xe = x + h;
    }
  }
  if (c != 0) return;
  if (a.halo_viol && viol) atomicAdd(a.halo_viol, 1u);
  const bool solved = !(x <= 0.001);  // DepthProblemSolver.cpp:192
  bool keep = solved;
  if (solved) {
    atomicAdd(n_solved, 1u);
    const double invJtJ = (r != 0.) ? (1. / r) * (1. / r) : 0.;  // internal::covar, n == 1
    double variance;
    if constexpr (L2) {  // :200-206 (the solver's |f| for both norms: kernels_lm.hip)
      variance = fnorm * fnorm / (double)(N - 1) * invJtJ;
      if (variance < 1e-6) variance = 1e-6;
    } else {
      variance = p.td_stdvar2 * invJtJ;  // :210
    }
    const double residual = fnorm * fnorm;  // :212
    DevPoint o;
    o.row = (u32)(size_t)floor(pr.cy);  // :116
    o.col = (u32)(size_t)floor(pr.cx);
    o.x[0] = pr.cx;
    o.x[1] = pr.cy;
    cam2World(p.camL, pr.cx, pr.cy, x, o.p_cam);  // :119
    o.inv_depth = x;
    o.scale2 = L2 ? 0.0 : variance * (p.td_nu - 2) / p.td_nu;  // :125
    o.nu = L2 ? 0.0 : p.td_nu;
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

void launch_lm_refine_any(const LmArgs& a, const DevParams& p, u32* n_solved, hipStream_t s) {
  if (a.max_matches == 0) return;
  const size_t lds = (size_t)3 * p.wy * 64 * sizeof(double);
  if (p.ls_norm == ESVO_LSNORM_L2)
    hipLaunchKernelGGL((lm_refine_any_kernel<true>), dim3(a.max_matches), dim3(64), lds, s, a, p, n_solved);
  else
    hipLaunchKernelGGL((lm_refine_any_kernel<false>), dim3(a.max_matches), dim3(64), lds, s, a, p, n_solved);
}

}  // namespace esvo
