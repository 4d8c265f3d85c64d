// api_track.hip — tracker residual / Jacobian evaluation on the device-resident Time Surface (see context.hpp).
#include "context.hpp"
#include "../../include/esvo_hip.hpp"  // esvo_hip::gauss_newton_register: the host-side driver esvo_track_register runs

// ---- Tracker residual / Jacobian evaluation ---------------------------------------
extern "C" {
int esvo_track_set_current(esvo_handle h, const uint8_t* ts_left, int kernel_size) {
  if (!h) return ESVO_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> _trk_lock(h->mu_track);
  if (kernel_size != 0 && kernel_size != 5)
    FAIL(ESVO_ERR_UNSUPPORTED, "tracker kernelSize must be 0 or 5 (the shipped configs); other sizes take OpenCV's float kernel path");
  HIPCHK(hipSetDevice(h->device));
  const size_t npx = (size_t)h->W * h->H;
  if (!h->d_trk_neg) {
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&h->d_trk_blur), npx));
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&h->d_trk_neg), npx));
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&h->d_trk_du), npx * sizeof(int16_t)));
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&h->d_trk_dv), npx * sizeof(int16_t)));
  }
  const uint8_t* src = h->d_ts[0];
  if (ts_left) {  // host image (TS node in another process)
    HIPCHK(hipMemcpyAsync(h->d_trk_neg, ts_left, npx, hipMemcpyHostToDevice, h->stream_t));
    src = h->d_trk_neg;  // staged here, consumed by the blur / copy below before track_images writes it
    if (kernel_size == 5) launch_gaussian5(src, h->d_trk_blur, h->W, h->H, h->stream_t);
    else HIPCHK(hipMemcpyAsync(h->d_trk_blur, src, npx, hipMemcpyDeviceToDevice, h->stream_t));
  } else {
    // The resident left surface belongs to the mapper group, whose thread may render the next one at any moment: the read
    // is enqueued behind the newest render (EV_R1) and marked, under mu_ts, so that the next render queues behind it.
    std::lock_guard<std::mutex> lt(h->mu_ts);
    if (!h->ts_valid[0]) FAIL(ESVO_ERR_STATE, "no device-resident left Time Surface: call esvo_ts_render(h, 0, ...) first");
    // (a render that runs alone records EV_R1 only once the tracker has shown up -- context.hpp, trk_used; renders are enqueued
    //  under mu_ts, held here: the first call drains the front stream instead, every later render has recorded the event)
    if (!h->trk_used.exchange(true)) HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipStreamWaitEvent(h->stream_t, h->evt[EV_R1], 0));  // the render of camera 0 on the front stream
    if (kernel_size == 5) launch_gaussian5(src, h->d_trk_blur, h->W, h->H, h->stream_t);
    else HIPCHK(hipMemcpyAsync(h->d_trk_blur, src, npx, hipMemcpyDeviceToDevice, h->stream_t));
    HIPCHK(hipEventRecord(h->evt_trk_read, h->stream_t));
    h->trk_read_pending = true;
  }
  launch_track_images(h->d_trk_blur, h->d_trk_neg, h->d_trk_du, h->d_trk_dv, h->W, h->H, h->stream_t);
  HIPCHK(hipGetLastError());
  // a host image is borrowed for the call; the resident surface needs no host wait (the evaluation calls run on this stream)
  if (ts_left) HIPCHK(hipStreamSynchronize(h->stream_t));
  h->trk_cur = true;
  return ESVO_OK;
}

int esvo_track_get_images(esvo_handle h, uint8_t* neg, int16_t* du, int16_t* dv) {
  if (!h) return ESVO_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> _trk_lock(h->mu_track);
  if (!h->trk_cur) FAIL(ESVO_ERR_STATE, "esvo_track_set_current has not been called");
  HIPCHK(hipSetDevice(h->device));
  const size_t npx = (size_t)h->W * h->H;
  if (neg) HIPCHK(hipMemcpyAsync(neg, h->d_trk_neg, npx, hipMemcpyDeviceToHost, h->stream_t));
  if (du) HIPCHK(hipMemcpyAsync(du, h->d_trk_du, npx * sizeof(int16_t), hipMemcpyDeviceToHost, h->stream_t));
  if (dv) HIPCHK(hipMemcpyAsync(dv, h->d_trk_dv, npx * sizeof(int16_t), hipMemcpyDeviceToHost, h->stream_t));
  HIPCHK(hipStreamSynchronize(h->stream_t));
  return ESVO_OK;
}

int esvo_track_set_reference(esvo_handle h, const float* xyz_world, size_t n, const double T_world_ref[16]) {
  if (!h || (n && !xyz_world) || !T_world_ref) return ESVO_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> _trk_lock(h->mu_track);
  HIPCHK(hipSetDevice(h->device));
  if (n > h->trk_cap) {
    HIPCHK(hipStreamSynchronize(h->stream_t));
    for (void* q : {(void*)h->d_trk_xyz, (void*)h->d_trk_pts, (void*)h->d_trk_out}) if (q) hipFree(q);
    if (h->h_trk_xyz) hipHostFree(h->h_trk_xyz);
    h->d_trk_xyz = nullptr; h->d_trk_pts = nullptr; h->d_trk_out = nullptr; h->h_trk_xyz = nullptr;
    const size_t cap = std::max<size_t>(n, 4096);
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&h->h_trk_xyz), cap * 3 * sizeof(float)));
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&h->d_trk_xyz), cap * 3 * sizeof(float)));
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&h->d_trk_pts), cap * 3 * sizeof(double)));
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&h->d_trk_out), cap * 6 * sizeof(double)));
    h->trk_cap = cap;
  }
  h->trk_n = n;
  if (n) {
    TrackRef r;
    std::memcpy(r.T, T_world_ref, sizeof(r.T));
    // xyz_world is borrowed for the call: it is copied into a pinned staging buffer (24 KB for 2000 points) instead of waiting
    // for the device -- the previous upload out of that buffer finished long ago (every evaluation call ends with a wait on
    // this stream), the stream wait below only covers a caller that sets two references in a row
    if (h->trk_xyz_inflight) HIPCHK(hipStreamSynchronize(h->stream_t));
    std::memcpy(h->h_trk_xyz, xyz_world, n * 3 * sizeof(float));
    h->trk_xyz_inflight = true;
    HIPCHK(hipMemcpyAsync(h->d_trk_xyz, h->h_trk_xyz, n * 3 * sizeof(float), hipMemcpyHostToDevice, h->stream_t));
    launch_track_reference(h->d_trk_xyz, (u32)n, r, h->d_trk_pts, h->stream_t);
    HIPCHK(hipGetLastError());
  }
  return ESVO_OK;
}
}  // extern "C"

namespace {
void fill_track_args(esvo_context* h, TrackArgs& a) {
  a.pts = h->d_trk_pts; a.neg = h->d_trk_neg; a.du = h->d_trk_du; a.dv = h->d_trk_dv; a.mask = h->d_mask;
  std::memcpy(a.P, h->dp.camL.P, sizeof(a.P));
  a.W = h->W; a.H = h->H;
}
}  // namespace

extern "C" {
int esvo_track_residuals(esvo_handle h, const double T_left_ref[16], size_t offset, size_t count, int ls_norm,
                         double huber_threshold, double* fvec, size_t* n_out) {
  if (!h || !T_left_ref || !n_out || (ls_norm != ESVO_TRACK_L2 && ls_norm != ESVO_TRACK_HUBER)) return ESVO_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> _trk_lock(h->mu_track);
  if (!h->trk_cur) FAIL(ESVO_ERR_STATE, "esvo_track_set_current has not been called");
  HIPCHK(hipSetDevice(h->device));
  const size_t m = offset >= h->trk_n ? 0 : std::min(count, h->trk_n - offset);  // setStochasticSampling, :71-88
  *n_out = m;
  if (m == 0) return ESVO_OK;
  if (!fvec) return ESVO_ERR_INVALID_ARG;
  TrackArgs a;
  fill_track_args(h, a);
  TrackPose pose;
  std::memcpy(pose.T, T_left_ref, sizeof(pose.T));
  std::memset(pose.Jc, 0, sizeof(pose.Jc));
  launch_track_residuals(a, pose, (u32)offset, (u32)m, ls_norm == ESVO_TRACK_HUBER, huber_threshold, h->d_trk_out, h->stream_t);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(fvec, h->d_trk_out, m * sizeof(double), hipMemcpyDeviceToHost, h->stream_t));
  HIPCHK(hipStreamSynchronize(h->stream_t));
  h->trk_xyz_inflight = false;
  return ESVO_OK;
}

int esvo_track_jacobian(esvo_handle h, const double R[9], const double t[3], size_t offset, size_t count, double* fjac,
                        size_t* n_out) {
  if (!h || !R || !t || !n_out) return ESVO_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> _trk_lock(h->mu_track);
  if (!h->trk_cur) FAIL(ESVO_ERR_STATE, "esvo_track_set_current has not been called");
  HIPCHK(hipSetDevice(h->device));
  const size_t m = offset >= h->trk_n ? 0 : std::min(count, h->trk_n - offset);
  *n_out = m;
  if (m == 0) return ESVO_OK;
  if (!fjac) return ESVO_ERR_INVALID_ARG;
  TrackArgs a;
  fill_track_args(h, a);
  TrackPose pose;  // T_left_ref = [R^T | -R^T t] (:203-205), J_constPart = R^T diag(1/P11, 1/P22; 0) (:189-194)
  std::memset(pose.T, 0, sizeof(pose.T));
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) pose.T[r * 4 + c] = R[c * 3 + r];
    pose.T[r * 4 + 3] = (-R[0 * 3 + r] * t[0] + -R[1 * 3 + r] * t[1]) + -R[2 * 3 + r] * t[2];
  }
  pose.T[15] = 1.0;
  const double iP11 = 1.0 / a.P[0], iP22 = 1.0 / a.P[5];
  for (int r = 0; r < 3; ++r) { pose.Jc[r * 2 + 0] = R[0 * 3 + r] * iP11; pose.Jc[r * 2 + 1] = R[1 * 3 + r] * iP22; }
  launch_track_jacobian(a, pose, (u32)offset, (u32)m, h->d_trk_out, h->stream_t);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(fjac, h->d_trk_out, m * 6 * sizeof(double), hipMemcpyDeviceToHost, h->stream_t));
  HIPCHK(hipStreamSynchronize(h->stream_t));
  h->trk_xyz_inflight = false;
  return ESVO_OK;
}

static_assert(ESVO_TRACK_MAX_POSES == TRK_NE_MAX_POSES, "the ABI's pose limit is the kernel's");
// f = operator()(0), J = df(0) and their products in one launch: H = J^T J (6 x 6, row-major, symmetric), b = J^T f, cost = |f|^2;
// n_poses of them (one workgroup each) share the launch and the read-back
int esvo_track_normal_equations_batch(esvo_handle h, int n_poses, const double* R, const double* t, size_t offset, size_t count,
                                      int ls_norm, double huber_threshold, double* H, double* b, double* cost, size_t* n_out) {
  if (!h || !R || !t || !H || !b || !n_out || n_poses < 1 || n_poses > ESVO_TRACK_MAX_POSES ||
      (ls_norm != ESVO_TRACK_L2 && ls_norm != ESVO_TRACK_HUBER))
    return ESVO_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> _trk_lock(h->mu_track);
  if (!h->trk_cur) FAIL(ESVO_ERR_STATE, "esvo_track_set_current has not been called");
  HIPCHK(hipSetDevice(h->device));
  const size_t m = offset >= h->trk_n ? 0 : std::min(count, h->trk_n - offset);
  *n_out = m;
  std::memset(H, 0, sizeof(double) * 36 * n_poses);
  std::memset(b, 0, sizeof(double) * 6 * n_poses);
  if (cost) std::memset(cost, 0, sizeof(double) * n_poses);
  if (m == 0) return ESVO_OK;
  if (!h->h_trk_ne) HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&h->h_trk_ne), sizeof(double) * TRK_NE_TERMS * TRK_NE_MAX_POSES));
  TrackArgs a;
  fill_track_args(h, a);
  TrackPoseSet set;  // as esvo_track_jacobian: T_left_ref = [R^T | -R^T t] (RegProblemLM.cpp:203-205), J_constPart (:189-194)
  std::memset(&set, 0, sizeof(set));
  const double iP11 = 1.0 / a.P[0], iP22 = 1.0 / a.P[5];
  for (int q = 0; q < n_poses; ++q) {
    TrackPose& pose = set.p[q];
    const double *Rq = R + 9 * q, *tq = t + 3 * q;
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) pose.T[r * 4 + c] = Rq[c * 3 + r];
      pose.T[r * 4 + 3] = (-Rq[0 * 3 + r] * tq[0] + -Rq[1 * 3 + r] * tq[1]) + -Rq[2 * 3 + r] * tq[2];
    }
    pose.T[15] = 1.0;
    for (int r = 0; r < 3; ++r) { pose.Jc[r * 2 + 0] = Rq[0 * 3 + r] * iP11; pose.Jc[r * 2 + 1] = Rq[1 * 3 + r] * iP22; }
  }
  // The kernel's one write per pose (28 sums) goes straight into the pinned host row -- no copy operation behind the launch -- and the
  // host polls for the end of the launch (10-20 us of kernel: sleeping on the completion interrupt would cost as much again).  One
  // of these round trips per iteration is the whole latency of the tracker's loop.
  launch_track_normal(a, set, n_poses, (u32)offset, (u32)m, ls_norm == ESVO_TRACK_HUBER, huber_threshold, h->h_trk_ne, h->stream_t);
  HIPCHK(hipGetLastError());
  HIPCHK(esvo_wait_stream(h->stream_t, true));
  h->trk_xyz_inflight = false;
  for (int q = 0; q < n_poses; ++q) {
    const double* s = h->h_trk_ne + (size_t)q * TRK_NE_TERMS;
    double* Hq = H + 36 * q;
    int n = 0;
    for (int i = 0; i < 6; ++i)
      for (int j = i; j < 6; ++j) { Hq[i * 6 + j] = Hq[j * 6 + i] = s[n]; ++n; }
    for (int i = 0; i < 6; ++i) b[6 * q + i] = s[21 + i];
    if (cost) cost[q] = s[27];
  }
  return ESVO_OK;
}
int esvo_track_normal_equations(esvo_handle h, const double R[9], const double t[3], size_t offset, size_t count, int ls_norm,
                                double huber_threshold, double H[36], double b[6], double* cost, size_t* n_out) {
  return esvo_track_normal_equations_batch(h, 1, R, t, offset, count, ls_norm, huber_threshold, H, b, cost, n_out);
}

// The registration of one frame: esvo_hip::gauss_newton_register (include/esvo_hip.hpp, host C++) over
// esvo_track_normal_equations_batch -- the (up to three) trial steps of an iteration in one launch, 224 bytes back per trial.
int esvo_track_register(esvo_handle h, size_t n_points, double R[9], double t[3], int ls_norm, double huber_threshold,
                        int max_iterations, double damping, double* rms, int* iterations) {
  if (!h || !R || !t || max_iterations < 1) return ESVO_ERR_INVALID_ARG;
  int rc = ESVO_OK;
  auto ne = [&](int, int k, const double* Rc, const double* tc, double* H, double* b, double* cost, size_t* n) {
    rc = esvo_track_normal_equations_batch(h, k, Rc, tc, 0, n_points, ls_norm, huber_threshold, H, b, cost, n);
    return rc == ESVO_OK;
  };
  const esvo_hip::Registration res = esvo_hip::gauss_newton_register(ne, R, t, max_iterations, damping);
  if (rc) return rc;
  std::memcpy(R, res.R, sizeof(res.R));
  std::memcpy(t, res.t, sizeof(res.t));
  if (rms) *rms = res.rms;
  if (iterations) *iterations = res.iterations;
  return ESVO_OK;
}
}  // extern "C"
