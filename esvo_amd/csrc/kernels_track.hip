// kernels_track.hip — tracker residual / Jacobian evaluation on the device-resident Time Surface.
//
// Reference: esvo_core/src/core/RegProblemLM.cpp (operator() :91-136, thread :138-176, df :178-269,
// reprojection / isValidPatch / patchInterpolation :366-483) and TimeSurfaceObservation.h:118-147
// (getTimeSurfaceNegative + computeTsNegativeGrad), for the settings every shipped tracking config uses:
// patch 1x1, kernelSize 5 (or 0), l2 / Huber, analytical Jacobian at x = 0.
//
// The negated blurred TS is an 8-bit image and its 3x3 Sobel derivatives are integers in [-1020, 1020], so the
// three images are kept as u8 / i16 (5 B per pixel instead of the reference's three f64 matrices); a point costs one
// 2x2 fetch per image.  One thread per point; every f64 operation is written in the order of the reference's
// expressions (no contraction), so results equal the CPU restatement bit for bit.
#include "common.hpp"

namespace esvo {

__device__ inline int trk_reflect101(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

// neg = 255 - blurred; du, dv = cv::Sobel(neg, CV_64F, 1,0 / 0,1), ksize 3, BORDER_REFLECT_101
__global__ void __launch_bounds__(256) track_images_kernel(const uint8_t* __restrict__ blurred, uint8_t* __restrict__ neg,
                                                           int16_t* __restrict__ du, int16_t* __restrict__ dv, int W, int H) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= W * H) return;
  const int y = i / W, x = i - y * W;
  int v[3][3];
#pragma unroll
  for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx)
      v[dy + 1][dx + 1] = 255 - (int)blurred[trk_reflect101(y + dy, H) * W + trk_reflect101(x + dx, W)];
  neg[i] = (uint8_t)v[1][1];
  du[i] = (int16_t)((v[0][2] - v[0][0]) + 2 * (v[1][2] - v[1][0]) + (v[2][2] - v[2][0]));
  dv[i] = (int16_t)((v[2][0] - v[0][0]) + 2 * (v[2][1] - v[0][1]) + (v[2][2] - v[0][2]));
}
void launch_track_images(const uint8_t* blurred, uint8_t* neg, int16_t* du, int16_t* dv, int W, int H, hipStream_t s) {
  hipLaunchKernelGGL(track_images_kernel, dim3((W * H + 255) / 256), dim3(256), 0, s, blurred, neg, du, dv, W, H);
}

// the point loop of RegProblemLM::setProblem (:44-56): p_cam = R_world_ref^T (p - t_world_ref)
__global__ void __launch_bounds__(256) track_reference_kernel(const float* __restrict__ xyz, u32 n, TrackRef r,
                                                              double* __restrict__ pts) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double d[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) d[k] = (double)xyz[3 * i + k] - r.T[k * 4 + 3];
#pragma unroll
  for (int c = 0; c < 3; ++c) pts[3 * i + c] = (r.T[0 * 4 + c] * d[0] + r.T[1 * 4 + c] * d[1]) + r.T[2 * 4 + c] * d[2];
}
void launch_track_reference(const float* xyz, u32 n, const TrackRef& r, double* pts, hipStream_t s) {
  if (n == 0) return;
  hipLaunchKernelGGL(track_reference_kernel, dim3((n + 255) / 256), dim3(256), 0, s, xyz, n, r, pts);
}

// reprojection (:387-401) + isValidPatch (:366-385) for a 1x1 patch
__device__ inline bool trk_reproject(const TrackArgs& a, const double p[3], const double* T, double x[2]) {
  double pl[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) pl[r] = ((T[r * 4 + 0] * p[0] + T[r * 4 + 1] * p[1]) + T[r * 4 + 2] * p[2]) + T[r * 4 + 3];
  double hm[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) hm[r] = ((a.P[r * 4 + 0] * pl[0] + a.P[r * 4 + 1] * pl[1]) + a.P[r * 4 + 2] * pl[2]) + a.P[r * 4 + 3];
  x[0] = hm[0] / hm[2];
  x[1] = hm[1] / hm[2];
  if (!isfinite(x[0]) || !isfinite(x[1])) return false;
  if (x[0] < 0.0 || x[0] > (double)(a.W - 1) || x[1] < 0.0 || x[1] > (double)(a.H - 1)) return false;
  if (a.mask && a.mask[(int)x[1] * a.W + (int)x[0]] < 125) return false;
  return true;
}
// patchInterpolation (:403-483), 1x1 patch, on an integer-valued image
template <typename Px>
__device__ inline bool trk_interp(const TrackArgs& a, const Px* __restrict__ img, const double loc[2], double& out) {
  const int ux = (int)floor(loc[0]), uy = (int)floor(loc[1]);
  if (ux < 0 || uy < 0 || ux >= a.W || uy >= a.H) return false;
  const double q1 = (double)(ux + 1) - loc[0], q2 = loc[0] - (double)ux;
  const double q3 = (double)(uy + 1) - loc[1], q4 = loc[1] - (double)uy;
  if (uy + 1 >= a.H || ux + 1 >= a.W) return false;
  const Px* s = img + (size_t)uy * a.W + ux;
  const double r0 = q1 * (double)s[0] + q2 * (double)s[1];
  const double r1 = q1 * (double)s[a.W] + q2 * (double)s[a.W + 1];
  out = q3 * r0 + q4 * r1;
  return true;
}

// one value of RegProblemLM::operator() (:91-136) + thread() (:138-176)
__device__ inline double trk_residual(const TrackArgs& a, const double* T_left_ref, const double pp[3], int huber, double huber_threshold) {
  double x[2], r = 255.0, tau;
  if (trk_reproject(a, pp, T_left_ref, x) && trk_interp(a, a.neg, x, tau)) r = tau;
  if (huber) {
    double w = 1.0;
    if (r > huber_threshold) w = huber_threshold / r;
    r = sqrt(w) * r;
  }
  return r;
}
__global__ void __launch_bounds__(256) track_residual_kernel(TrackArgs a, TrackPose pose, u32 offset, u32 count, int huber,
                                                             double huber_threshold, double* __restrict__ fvec) {
  const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= count) return;
  const double* p = a.pts + 3 * (size_t)(offset + k);
  const double pp[3] = {p[0], p[1], p[2]};
  fvec[k] = trk_residual(a, pose.T, pp, huber, huber_threshold);
}
void launch_track_residuals(const TrackArgs& a, const TrackPose& pose, u32 offset, u32 count, int huber, double thr, double* fvec,
                            hipStream_t s) {
  if (count == 0) return;
  hipLaunchKernelGGL(track_residual_kernel, dim3((count + 255) / 256), dim3(256), 0, s, a, pose, offset, count, huber, thr, fvec);
}

// one row of RegProblemLM::df at x = 0 (:178-269); pose.T = T_left_ref, pose.Jc = J_constPart (3x2)
__device__ inline void trk_jacobian_row(const TrackArgs& a, const TrackPose& pose, const double p[3], double row[6]) {
  double e[12];
  double x[2];
  if (!trk_reproject(a, p, pose.T, x)) {
#pragma unroll
    for (int j = 0; j < 12; ++j) e[j] = 0.0;
  } else {
    double gx = 0.0, gy = 0.0;  // border pixel: the reference reads an unset matrix here; defined as 0 (oracle)
    trk_interp(a, a.du, x, gx);
    trk_interp(a, a.dv, x, gy);
    const double g0 = gx / 8, g1 = gy / 8;
    const double z = p[2], z2 = z * z;
    const double P11 = a.P[0], P12 = a.P[1], P14 = a.P[3], P21 = a.P[4], P22 = a.P[5], P24 = a.P[7];
    double D[2][3];
    D[0][0] = P11 / z; D[0][1] = P12 / z; D[0][2] = -((P11 * p[0] + P12 * p[1]) + P14) / z2;
    D[1][0] = P21 / z; D[1][1] = P22 / z; D[1][2] = -((P21 * p[0] + P22 * p[1]) + P24) / z2;
    double av[3], b[2], c[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) av[j] = g0 * D[0][j] + g1 * D[1][j];
#pragma unroll
    for (int j = 0; j < 2; ++j) b[j] = (av[0] * pose.Jc[0 * 2 + j] + av[1] * pose.Jc[1 * 2 + j]) + av[2] * pose.Jc[2 * 2 + j];
#pragma unroll
    for (int j = 0; j < 3; ++j) c[j] = b[0] * D[0][j] + b[1] * D[1][j];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      e[j] = (c[j] * p[0]) * z; e[3 + j] = (c[j] * p[1]) * z; e[6 + j] = (c[j] * p[2]) * z; e[9 + j] = c[j] * z;
    }
  }
  row[0] = -(2.0 * e[5] - 2.0 * e[7]);
  row[1] = -(2.0 * e[6] - 2.0 * e[2]);
  row[2] = -(2.0 * e[1] - 2.0 * e[3]);
  row[3] = -e[9];
  row[4] = -e[10];
  row[5] = -e[11];
}
// fjac is count x 6 column-major
__global__ void __launch_bounds__(256) track_jacobian_kernel(TrackArgs a, TrackPose pose, u32 offset, u32 count,
                                                             double* __restrict__ fjac) {
  const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= count) return;
  const double* pq = a.pts + 3 * (size_t)(offset + k);
  const double p[3] = {pq[0], pq[1], pq[2]};
  double row[6];
  trk_jacobian_row(a, pose, p, row);
  const size_t m = count;
#pragma unroll
  for (int j = 0; j < 6; ++j) fjac[(size_t)j * m + k] = row[j];
}
void launch_track_jacobian(const TrackArgs& a, const TrackPose& pose, u32 offset, u32 count, double* fjac, hipStream_t s) {
  if (count == 0) return;
  hipLaunchKernelGGL(track_jacobian_kernel, dim3((count + 255) / 256), dim3(256), 0, s, a, pose, offset, count, fjac);
}

// ---- normal equations of one Gauss-Newton / LM iteration in ONE launch --------------------------------------------------
// What RegProblemSolverLM::solve_analytical's iteration consumes (esvo_core/src/core/RegProblemSolverLM.cpp:148-215:
// minimizeInit -> F(0), minimizeOneStep -> df at 0, then the 6 x 6 system): f = operator()(0) with the Huber weights
// (:121-131), J = df(0) (the reference's Jacobian carries no weight), and their products J^T J (21 upper-triangle entries,
// row-major i <= j), J^T f (6) and |f|^2 -- 28 doubles instead of count x 7 over PCIe and two synchronous calls.
// Summation order (the oracle restates it: orc_tracker_normal_equations): thread t of the single 256-thread workgroup adds
// the terms of points t, t + 256, t + 512, ... in that order; the 256 partial sums are then folded by the tree
// s[t] += s[t + 128], s[t] += s[t + 64], ..., s[0] += s[1].
// One workgroup per POSE (blockIdx.x): the trial steps of an LM iteration (several damping values) cost one launch and one
// read-back together -- the sums of a pose do not depend on how many others ride along.
__global__ void __launch_bounds__(TRK_NE_THREADS) track_normal_kernel(TrackArgs a, TrackPoseSet poses, u32 offset, u32 count, int huber,
                                                                    double huber_threshold, double* __restrict__ out) {
  __shared__ double red[TRK_NE_TERMS][TRK_NE_THREADS];
  const TrackPose& pose = poses.p[blockIdx.x];
  out += (size_t)blockIdx.x * TRK_NE_TERMS;
  double acc[TRK_NE_TERMS];
#pragma unroll
  for (int n = 0; n < TRK_NE_TERMS; ++n) acc[n] = 0.0;
  for (u32 k = threadIdx.x; k < count; k += TRK_NE_THREADS) {
    const double* pq = a.pts + 3 * (size_t)(offset + k);
    const double p[3] = {pq[0], pq[1], pq[2]};
    const double f = trk_residual(a, pose.T, p, huber, huber_threshold);
    double row[6];
    trk_jacobian_row(a, pose, p, row);
    int n = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = i; j < 6; ++j) { acc[n] = acc[n] + row[i] * row[j]; ++n; }
#pragma unroll
    for (int i = 0; i < 6; ++i) acc[21 + i] = acc[21 + i] + row[i] * f;
    acc[27] = acc[27] + f * f;
  }
#pragma unroll
  for (int n = 0; n < TRK_NE_TERMS; ++n) red[n][threadIdx.x] = acc[n];
  for (u32 s = TRK_NE_THREADS / 2; s > 0; s >>= 1) {
    __syncthreads();
    if (threadIdx.x < s)
      for (int n = 0; n < TRK_NE_TERMS; ++n) red[n][threadIdx.x] = red[n][threadIdx.x] + red[n][threadIdx.x + s];
  }
  __syncthreads();
  if (threadIdx.x < TRK_NE_TERMS) {
    out[threadIdx.x] = red[threadIdx.x][0];
    __threadfence_system();  // `out` may be pinned host memory the caller polls the stream for (api_track.hip)
  }
}
void launch_track_normal(const TrackArgs& a, const TrackPoseSet& poses, int n_poses, u32 offset, u32 count, int huber, double thr,
                         double* out, hipStream_t s) {
  hipLaunchKernelGGL(track_normal_kernel, dim3(n_poses), dim3(TRK_NE_THREADS), 0, s, a, poses, offset, count, huber, thr, out);
}

}  // namespace esvo
