// api.hip — host side of the C-ABI declared in include/esvo_hip.h.
//
// One esvo_context owns every device buffer of one GPU: SAE + staged event rings (Time
// Surface), the observation pair, per-tick BM/LM scratch, the fusion window ring and the dense
// DepthMap.  Each entry point replays, on the handle's HIP stream, the call sequence of the
// reference seam it replaces (cited in esvo_hip.h); nothing here computes on the CPU except
// bookkeeping (time-stamp binary searches, window policy, output ordering).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <string>
#include <vector>

#include "common.hpp"

using namespace esvo;

namespace {
thread_local std::string g_create_error;
}

// front-stage events live on the front stream; the back-stage set exists once per tick parity (two ticks in flight)
enum { EV_SC0 = 0, EV_SC1, EV_R1, EV_SC0b, EV_SC1b, EV_R1b, EV_FRAME,
       EV_T0, EV_BM0, EV_BM1, EV_S1, EV_LM0, EV_LM1, EV_S2, EV_CNT, EV_STG, EV_T0b, EV_BM0b, EV_BM1b, EV_S1b, EV_LM0b, EV_LM1b, EV_S2b, EV_CNTb, EV_STGb,
       EV_FU0, EV_FU1, EV_CL1, EV_RG1, EV_POSE, EV_FU0b, EV_FU1b, EV_CL1b, EV_RG1b, EV_POSEb, EV_N };
constexpr int EV_BACK_STRIDE = EV_FU0b - EV_FU0;  // evt[EV_x + par * EV_BACK_STRIDE]
constexpr int EV_FRONT_STRIDE = EV_T0b - EV_T0;   // evt[EV_x + fpar * EV_FRONT_STRIDE]
constexpr int EV_TS_STRIDE = EV_SC0b - EV_SC0;    // evt[EV_x + cam * EV_TS_STRIDE]

struct FrameRec {
  u32 off;    // offset in the window ring
  u32 count;  // points
  u32 slot;   // pose-table slot
};

struct esvo_context {
  esvo_params_t prm;
  DevParams dp;
  int W = 0, H = 0, device = 0;
  // Two streams, two ticks in flight: the front stage (TS, block matching, LM, frame assembly) of tick k+1 runs on
  // `stream` while the back stage (propagate, fuse, clean, regularise) of tick k runs on `stream_b`.  The back
  // stage only reads what the front stage finished (the frame in the window ring, the tick's pose table).
  hipStream_t stream = nullptr;
  hipStream_t stream_b = nullptr;
  hipStream_t stream_i = nullptr;  // event ingest (H2D into the ring): staging new events never waits for a running tick
  bool own_stream = false;
  int par = 0;                    // parity of the tick being assembled
  bool back_pending[2] = {false, false};  // back-stage timings / counters of that parity not collected yet
  u32 back_frames[2] = {0, 0};
  std::string err;
  double baseline = 0;

  // calibration
  float2* d_lut = nullptr;
  uint8_t* d_mask = nullptr;
  int2* d_fixmap[2] = {nullptr, nullptr};

  // Time Surface
  u64* d_sae[2] = {nullptr, nullptr};
  uint8_t* d_raw = nullptr;
  uint8_t* d_ts[2] = {nullptr, nullptr};
  bool ts_valid[2] = {false, false};
  esvo_event_t* d_ring[2] = {nullptr, nullptr};
  uint8_t* d_wire = nullptr;    // staging of serialised 13-byte event records (esvo_ts_push_event_array)
  size_t wire_cap = 0;
  u64 ring_cap = 0;
  std::deque<u64> ts_host[2];   // time stamps of staged events [ring_base, ring_base + size)
  u64 ring_base[2] = {0, 0};    // absolute index of ts_host[cam].front()
  u64 ring_next[2] = {0, 0};    // absolute index of the next event to stage
  u64 scattered[2] = {0, 0};    // absolute index of the first event not yet in the SAE
  u64 scatter_pending_lo[2] = {~0ull, ~0ull};  // oldest event a possibly still running scatter kernel reads

  // observation
  uint8_t* d_obs[2] = {nullptr, nullptr};
  uint8_t* d_obs_tmp = nullptr;
  double T_world_obs[16];
  double* d_T_world_obs = nullptr;
  u64 obs_t_ns = 0;
  bool obs_set = false;

  // pose table of the tick
  double* d_pose_sec = nullptr;
  double* d_pose_T = nullptr;     // the tick's table (one of d_pose_T2, alternating)
  double* d_pose_T2[2] = {nullptr, nullptr};
  int pose_buf = 0;
  std::vector<double> h_pose_T;
  double* h_pin = nullptr;        // pinned staging: 2 slots x (max_poses x 17 + 16) doubles
  int pin_slot = 0;
  bool stats_pending = false;     // the last tick's counters / timings have not been read back yet
  u32 n_pose = 0;

  // per-tick scratch
  u32 max_ev = 0;
  esvo_event_t* d_tick_ev = nullptr;
  esvo_match_t* d_match_slots = nullptr;
  u32* d_match_flags = nullptr;
  u32* d_match_prefix = nullptr;
  esvo_match_t* d_matches = nullptr;
  DevPoint* d_pt_slots = nullptr;
  u32* d_pt_flags = nullptr;
  u32* d_pt_prefix = nullptr;
  DevPoint* d_pts_tmp = nullptr;  // stage-wise refine output
  DevPoint* d_stage[2] = {nullptr, nullptr};  // a lazily completed tick's frame (by parity) until its count is known
  u32* d_counters = nullptr;      // [0] n_matches [1] n_points [2] n_solved [3] n_fusion [4] n_records [5] n_map
                                  // [6] touched cells [7] regulariser elements [8] own matches (sharded)
  u32* h_counters = nullptr;      // pinned
  u32* d_scan_tmp = nullptr;
  u32* d_cnt_b = nullptr;         // back stage: [3] n_fusion [4] n_records [5] n_map [6] touched cells [7] regulariser elements
  u32* h_cnt_b = nullptr;         // pinned, one row of 8 per parity + one for exports
  u32* d_scan_tmp_b = nullptr;

  // fusion window
  DevPoint* d_win = nullptr;
  u32 win_cap = 0;
  std::deque<FrameRec> frames;    // oldest first
  u32 n_pose_slots = 0;
  std::vector<char> slot_used;
  double* d_frame_pose_T = nullptr;
  u32 max_poses = 0;
  u32* d_fr_table = nullptr;      // fr_cum | fr_off | fr_slot
  u32* h_fr_table = nullptr;      // pinned
  u32 max_frames = 0;

  // DepthMap
  DevPoint* d_prop = nullptr;
  u32* d_cell_count = nullptr;
  u32* d_cell_offset = nullptr;
  u32* d_cell_fill = nullptr;
  u32* d_rec_ids = nullptr;
  MapCell* d_map = nullptr;
  MapCell* d_map2 = nullptr;
  MapCell* d_map_cur = nullptr;
  u32* d_owner_max = nullptr;
  u32* d_owner_min = nullptr;
  u32* d_bucket = nullptr;
  u32* d_sel = nullptr;           // denoising: walk positions of the kept events
  uint8_t* d_evmap = nullptr;     // denoising: binary event map
  // sharded mode (kernels_shard.hip): dense local lists + the (matched, kept) byte per slot that is exchanged
  u32* d_own_w = nullptr;         // slot w of the k-th own match
  u32* d_lkeep = nullptr;         // keep flag of the k-th own match after LM + culling
  uint8_t* d_codes = nullptr;     // [codes_bytes] one byte per slot, zero for other ranks' slots
  size_t codes_bytes = 0;
  void* xchg_ptr = nullptr;       // what the caller must sum across the ranks before the next phase
  size_t xchg_bytes = 0;
  u64* d_reg_valid = nullptr;     // regulariser view: 1 bit per cell
  bool sharded = false;
  u32 reg_words = 0;
  // A tick's state between its phases.  Unsharded ticks are finished lazily: esvo_map_tick(k) enqueues the front
  // stage of tick k and only then completes tick k-1 (point count -> window policy -> back stage), so the host
  // never waits on the front stream while it still has work to enqueue there.
  struct TickState {
    u32 n = 0, off = 0, points = 0, n_pose = 0;
    int pose_buf = 0;
    u64 t_ns = 0;
    double T_world_obs[16];
  } tk[2];
  int fpar = 0;                   // parity of the newest front stage
  bool tick_pending = false;      // tk[fpar] has its front stage enqueued but is not committed yet
  u64 committed_t_ns = 0;         // stamp of the newest tick whose back stage is enqueued (0: none)
  u64 sh_first = 0;
  u32* d_cell_list = nullptr;
  u64* d_reg_bits = nullptr;    // close-neighbour masks of the regulariser scan: [elements][words]
  u32* d_reg_counts = nullptr;  // (n_neighbours, n_close) per element
  double2* d_reg_ab = nullptr;
  double2* d_reg_cd = nullptr;
  double T_world_frame[16];
  // export
  u32* d_exp_flags = nullptr;
  u32* d_exp_prefix = nullptr;
  esvo_depth_point_t* d_export = nullptr;
  u32* d_export_cell = nullptr;

  // tracker residual / Jacobian evaluation (kernels_track.hip): own stream, own images, synchronous calls
  hipStream_t stream_t = nullptr;
  uint8_t* d_trk_blur = nullptr;
  uint8_t* d_trk_neg = nullptr;
  int16_t* d_trk_du = nullptr;
  int16_t* d_trk_dv = nullptr;
  float* d_trk_xyz = nullptr;
  double* d_trk_pts = nullptr;
  double* d_trk_out = nullptr;
  size_t trk_cap = 0, trk_n = 0;
  bool trk_cur = false;

  // pinned staging slots for frame pose tables that arrive from the host (push_frame variants): a slot is reused only
  // after the back stream has consumed it
  static constexpr int POSE_POOL = 32;
  double* h_pose_pool = nullptr;
  hipEvent_t pool_evt[POSE_POOL];
  bool pool_ok = false;
  int pool_next = 0;

  hipEvent_t evt[EV_N];
  bool evt_ok = false;
  esvo_stats_t stats;
  bool ts_timing_pending[2] = {false, false};
};

namespace {
int flush_pending_tick(esvo_context* h);  // completes a lazily finished tick (see TickState)
}

#define HIPCHK(call)                                                                              \
  do {                                                                                            \
    hipError_t _e = (call);                                                                       \
    if (_e != hipSuccess) {                                                                       \
      char _b[512];                                                                               \
      snprintf(_b, sizeof(_b), "%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); \
      if (h) h->err = _b; else g_create_error = _b;                                               \
      return ESVO_ERR_HIP;                                                                        \
    }                                                                                             \
  } while (0)

#define FAIL(code, msg)                      \
  do {                                       \
    if (h) h->err = (msg); else g_create_error = (msg); \
    return (code);                           \
  } while (0)

namespace {

void invert3x3(const double* P, double* Kinv, double* Kinv_t) {
  const double a = P[0], b = P[1], cc = P[2], d = P[4], e = P[5], f = P[6], g = P[8], hh = P[9], i = P[10];
  const double det = a * (e * i - f * hh) - b * (d * i - f * g) + cc * (d * hh - e * g);
  const double id = 1.0 / det;
  Kinv[0] = (e * i - f * hh) * id; Kinv[1] = (cc * hh - b * i) * id; Kinv[2] = (b * f - cc * e) * id;
  Kinv[3] = (f * g - d * i) * id;  Kinv[4] = (a * i - cc * g) * id;  Kinv[5] = (cc * d - a * f) * id;
  Kinv[6] = (d * hh - e * g) * id; Kinv[7] = (b * g - a * hh) * id;  Kinv[8] = (a * e - b * d) * id;
  for (int r = 0; r < 3; ++r) Kinv_t[r] = (Kinv[r * 3 + 0] * P[3] + Kinv[r * 3 + 1] * P[7]) + Kinv[r * 3 + 2] * P[11];
}

int validate_params(const esvo_params_t* p, std::string& why) {
  if (p->ls_norm != ESVO_LSNORM_TDIST) { why = "only LSnorm == Tdist is supported (every shipped config uses it)"; return ESVO_ERR_UNSUPPORTED; }
  if (p->bm_updown) { why = "BM_bUpDownConfiguration is not supported"; return ESVO_ERR_UNSUPPORTED; }
  if (p->bm_step != 1) { why = "BM_step != 1 is not supported (every shipped config uses 1)"; return ESVO_ERR_UNSUPPORTED; }
  if (p->patch_size_x != 15 || p->patch_size_y != 7) { why = "patch size must be 15x7 (every shipped config)"; return ESVO_ERR_UNSUPPORTED; }
  if (p->median_blur_kernel_size < 0 || p->median_blur_kernel_size > 1) { why = "median_blur_kernel_size must be 0 or 1"; return ESVO_ERR_UNSUPPORTED; }
  if (p->bm_max_disparity < p->bm_min_disparity || p->bm_min_disparity < 0) { why = "bad disparity range"; return ESVO_ERR_INVALID_ARG; }
  if (p->td_nu <= 2.0 || p->td_scale <= 0) { why = "Tdist_nu must be > 2 and Tdist_scale > 0"; return ESVO_ERR_INVALID_ARG; }
  if (p->num_threads < 1 || p->num_threads > 64) { why = "num_threads out of range"; return ESVO_ERR_INVALID_ARG; }
  if (p->lm_max_iteration < 1) { why = "lm_max_iteration must be >= 1"; return ESVO_ERR_INVALID_ARG; }
  if (p->reg_radius < 0 || p->reg_radius > 31) { why = "RegularizationRadius out of range [0,31] (one 64-bit mask per tap row)"; return ESVO_ERR_INVALID_ARG; }
  return ESVO_OK;
}

void fill_dev_params(esvo_context* h) {
  const esvo_params_t& p = h->prm;
  DevParams& d = h->dp;
  d.W = h->W; d.H = h->H;
  d.wx = p.patch_size_x; d.wy = p.patch_size_y;
  d.dmin = p.bm_min_disparity; d.dmax = p.bm_max_disparity; d.step = p.bm_step;
  d.zncc_thr = p.bm_zncc_threshold;
  d.baseline_f = h->baseline * d.camL.P[0];
  d.td_nu = p.td_nu; d.td_scale = p.td_scale; d.td_scale2 = p.td_scale * p.td_scale;
  const double td_stdvar = std::sqrt(p.td_nu / (p.td_nu - 2) * (p.td_scale * p.td_scale));  // DepthProblem.h:34
  d.td_stdvar2 = td_stdvar * td_stdvar;
  d.lm_max_iter = p.lm_max_iteration; d.lm_maxfev = p.lm_max_iteration * 3;
  d.invdepth_min = p.invdepth_min; d.invdepth_max = p.invdepth_max;
  d.var_thr = p.stdvar_vis_threshold * p.stdvar_vis_threshold;
  d.cost_thr = (p.residual_vis_threshold * p.residual_vis_threshold) * (double)(p.patch_size_x * p.patch_size_y);
  d.age_thr = p.age_vis_threshold;
  d.fusion_radius = p.fusion_radius;
  d.reg_radius = p.reg_radius; d.reg_min_nb = p.reg_min_neighbours; d.reg_min_close = p.reg_min_close_neighbours;
  d.num_threads = p.num_threads;
}

template <typename T>
hipError_t dalloc(T** p, size_t n) { return hipMalloc(reinterpret_cast<void**>(p), (n ? n : 1) * sizeof(T)); }

// lower_bound over the staged time stamps with the reference's toSec() comparison
// (tools::EventBuffer_lower_bound, utils.h:51-56); returns an absolute index
u64 lower_bound_sec(const esvo_context* h, int cam, double t) {
  const auto& v = h->ts_host[cam];
  size_t lo = 0, hi = v.size();
  while (lo < hi) {
    size_t mid = (lo + hi) / 2;
    if (ns_to_sec(v[mid]) < t) lo = mid + 1; else hi = mid;
  }
  return h->ring_base[cam] + lo;
}
// ros::Time(double)  (TimeBase::fromSec)
u64 ros_time_from_sec(double t) {
  long long sec64 = (long long)std::floor(t);
  u32 sec = (u32)sec64;
  u32 nsec = (u32)std::round((t - sec) * 1e9);
  sec += (nsec / 1000000000ul);
  nsec %= 1000000000ul;
  return (u64)sec * 1000000000ull + nsec;
}

// compute band of the per-cell stages: the owned rows + a halo of 2 rows for the displaced-element side
// effects (+ the regulariser's radius), see DevParams::cband_y0
void set_compute_band(esvo_context* h) {
  const int halo = 2 + (h->prm.regularization ? h->prm.reg_radius : 0);
  h->dp.cband_y0 = std::max(0, h->dp.band_y0 - halo);
  h->dp.cband_y1 = std::min(h->H, h->dp.band_y1 + halo);
}

int upload_poses(esvo_context* h, const uint64_t* pose_t_ns, const double* pose_T, size_t m) {
  if (m > h->max_poses) FAIL(ESVO_ERR_CAPACITY, "pose table larger than max_poses_per_tick");
  // staged through pinned memory (two alternating slots): no host synchronisation on the tick path
  h->pin_slot ^= 1;
  double* pin = h->h_pin + (size_t)h->pin_slot * ((size_t)h->max_poses * 17 + 16);
  double* sec = pin;
  double* T = pin + h->max_poses;
  for (size_t i = 0; i < m; ++i) sec[i] = ns_to_sec(pose_t_ns[i]);
  std::memcpy(T, pose_T, sizeof(double) * 16 * m);
  h->h_pose_T.assign(pose_T, pose_T + 16 * m);
  h->n_pose = (u32)m;
  // the back stage copies the previous table of this buffer into its frame slot: not before that is done
  h->pose_buf ^= 1;
  h->d_pose_T = h->d_pose_T2[h->pose_buf];
  HIPCHK(hipStreamWaitEvent(h->stream, h->evt[EV_POSE + h->pose_buf * EV_BACK_STRIDE], 0));
  if (m) {
    HIPCHK(hipMemcpyAsync(h->d_pose_sec, sec, sizeof(double) * m, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_pose_T, T, sizeof(double) * 16 * m, hipMemcpyHostToDevice, h->stream));
  }
  return ESVO_OK;
}

// BM over n events starting at absolute ring index `first` (reverse walk) or over d_tick_ev:
// flags + match records in slot (thread-stride) order
int run_bm(esvo_context* h, const esvo_event_t* d_ev, u64 first, u64 cap, int reverse, u32 n, const u32* sel = nullptr) {
  BmArgs a;
  a.ev = d_ev; a.n = n; a.ev_first = first; a.ev_cap = cap; a.ev_reverse = reverse; a.sel = sel;
  a.tsL = h->d_obs[0]; a.tsR = h->d_obs[1];
  a.lut = h->d_lut; a.mask = h->d_mask;
  a.pose_sec = h->d_pose_sec; a.n_pose = h->n_pose;
  a.out_slots = h->d_match_slots; a.out_flags = h->d_match_flags;
  hipEventRecord(h->evt[EV_BM0 + h->fpar * EV_FRONT_STRIDE], h->stream);
  launch_bm_match(a, h->dp, h->stream);
  hipEventRecord(h->evt[EV_BM1 + h->fpar * EV_FRONT_STRIDE], h->stream);
  HIPCHK(hipGetLastError());
  return ESVO_OK;
}
// stable compaction of the match slots into vEMP order.  Sharded mode: the flags are this rank's own
// ones, the list is its dense local list (count -> counters[8]) and slot_of remembers each entry's slot.
int run_order_matches(esvo_context* h, u32 n, bool local) {
  launch_exclusive_scan_u32(h->d_match_flags, h->d_match_prefix, h->d_counters + (local ? 8 : 0), h->d_scan_tmp, n, h->stream);
  launch_compact_matches(h->d_match_slots, h->d_match_flags, h->d_match_prefix, n, h->d_matches, local ? h->d_own_w : nullptr,
                         h->stream);
  hipEventRecord(h->evt[EV_S1 + h->fpar * EV_FRONT_STRIDE], h->stream);
  HIPCHK(hipGetLastError());
  return ESVO_OK;
}
int run_match(esvo_context* h, const esvo_event_t* d_ev, u64 first, u64 cap, int reverse, u32 n) {
  int rc = run_bm(h, d_ev, first, cap, reverse, n);
  if (rc) return rc;
  return run_order_matches(h, n, false);
}

// LM (+cull) over the compacted matches: point records + flags in solver-slot order (dense: in list order)
int run_lm(esvo_context* h, u32 max_matches, int cull, bool dense) {
  u32* flags = dense ? h->d_lkeep : h->d_pt_flags;  // the kernel writes every flag of its launch range
  LmArgs a;
  a.matches = h->d_matches; a.n_matches = h->d_counters + (dense ? 8 : 0); a.max_matches = max_matches;
  a.tsL = h->d_obs[0]; a.tsR = h->d_obs[1];
  a.pose_T = h->d_pose_T; a.T_world_obs = h->d_T_world_obs;
  a.out_slots = h->d_pt_slots; a.out_flags = flags; a.cull = cull; a.dense = dense ? 1 : 0;
  hipEventRecord(h->evt[EV_LM0 + h->fpar * EV_FRONT_STRIDE], h->stream);
  launch_lm_refine(a, h->dp, h->d_counters + 2, h->stream);
  hipEventRecord(h->evt[EV_LM1 + h->fpar * EV_FRONT_STRIDE], h->stream);
  HIPCHK(hipGetLastError());
  return ESVO_OK;
}
// stable compaction of the solver slots: the culled points go to `dst` in the reference's order
int run_order_points(esvo_context* h, u32 max_matches, DevPoint* dst) {
  launch_exclusive_scan_u32(h->d_pt_flags, h->d_pt_prefix, h->d_counters + 1, h->d_scan_tmp, max_matches, h->stream);
  launch_compact_points(h->d_pt_slots, h->d_pt_flags, h->d_pt_prefix, h->d_counters + 0, max_matches, dst, h->stream);
  hipEventRecord(h->evt[EV_S2 + h->fpar * EV_FRONT_STRIDE], h->stream);
  HIPCHK(hipGetLastError());
  return ESVO_OK;
}
int run_refine(esvo_context* h, u32 max_matches, int cull, DevPoint* dst) {
  HIPCHK(hipMemsetAsync(h->d_counters + 2, 0, sizeof(u32), h->stream));  // n_solved (a tick zeroes all counters at once)
  int rc = run_lm(h, max_matches, cull, false);
  if (rc) return rc;
  return run_order_points(h, max_matches, dst);
}

// Time-Surface kernel timings of the last render of a camera (-1: both); the caller knows their events are complete
void collect_ts_timing(esvo_context* h, int only = -1) {
  for (int cam = 0; cam < 2; ++cam) {
    if (!h->ts_timing_pending[cam] || (only >= 0 && cam != only)) continue;
    h->ts_timing_pending[cam] = false;
    const int o = cam * EV_TS_STRIDE;
    float sc = 0, rd = 0;
    if (hipEventElapsedTime(&sc, h->evt[EV_SC0 + o], h->evt[EV_SC1 + o]) == hipSuccess &&
        hipEventElapsedTime(&rd, h->evt[EV_SC1 + o], h->evt[EV_R1 + o]) == hipSuccess) {
      h->stats.ms_ts_scatter = h->stats.ms_kernel[0] = sc;
      h->stats.ms_ts_render = h->stats.ms_kernel[1] = rd;
      h->stats.sum_ms_kernel[0] += sc;
      h->stats.sum_ms_kernel[1] += rd;
      h->stats.sum_ms_kernel[7] += 1;
    }
  }
}

int read_counters(esvo_context* h) {
  HIPCHK(hipMemcpyAsync(h->h_counters, h->d_counters, sizeof(u32) * 16, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return ESVO_OK;
}
// back-stage counters into row `row` of the pinned table (0/1: the tick parities, 2: exports)
int read_counters_b(esvo_context* h, int row, bool sync) {
  HIPCHK(hipMemcpyAsync(h->h_cnt_b + 8 * row, h->d_cnt_b, sizeof(u32) * 8, hipMemcpyDeviceToHost, h->stream_b));
  if (sync) HIPCHK(hipStreamSynchronize(h->stream_b));
  return ESVO_OK;
}
// the back stage starts when everything enqueued on the front stream so far is done
int back_after_front(esvo_context* h) {
  HIPCHK(hipEventRecord(h->evt[EV_FRAME], h->stream));
  HIPCHK(hipStreamWaitEvent(h->stream_b, h->evt[EV_FRAME], 0));
  return ESVO_OK;
}
// timings and counters of a finished back stage
void collect_back(esvo_context* h, int par) {
  if (!h->back_pending[par]) return;
  h->back_pending[par] = false;
  const int o = par * EV_BACK_STRIDE;
  esvo_stats_t& s = h->stats;
  s.last_fusions = h->h_cnt_b[8 * par + 3];
  if (h->prm.regularization) s.last_map_size = h->h_cnt_b[8 * par + 7];  // alive cells of the band (exports refresh it)
  float fu = 0, cl = 0, rg = 0;
  hipEventElapsedTime(&fu, h->evt[EV_FU0 + o], h->evt[EV_FU1 + o]);
  hipEventElapsedTime(&cl, h->evt[EV_FU1 + o], h->evt[EV_CL1 + o]);
  hipEventElapsedTime(&rg, h->evt[EV_CL1 + o], h->evt[EV_RG1 + o]);
  s.ms_fusion = fu + cl;
  s.ms_regularization = rg;
  s.ms_kernel[4] = fu; s.ms_kernel[5] = cl; s.ms_kernel[6] = rg;
  s.sum_ms_kernel[4] += fu; s.sum_ms_kernel[5] += cl; s.sum_ms_kernel[6] += rg;
}

// place a frame of n points in the window ring (frames stay contiguous: [oldest frame, newest frame) modulo the wrap)
int window_reserve(esvo_context* h, u32 n, u32* off_out) {
  u32 off = 0;
  if (!h->frames.empty()) {
    const FrameRec& back = h->frames.back();
    const FrameRec& front = h->frames.front();
    const u32 tail = back.off + back.count;
    if (back.off >= front.off) {  // not wrapped: [front.off, tail)
      if (tail + n <= h->win_cap) off = tail;
      else if (n <= front.off) off = 0;
      else FAIL(ESVO_ERR_CAPACITY, "fusion window ring full (raise max_window_points)");
    } else {  // wrapped: free space is [tail, front.off)
      if (tail + n <= front.off) off = tail;
      else FAIL(ESVO_ERR_CAPACITY, "fusion window ring full (raise max_window_points)");
    }
  } else if (n > h->win_cap) {
    FAIL(ESVO_ERR_CAPACITY, "frame larger than the fusion window ring");
  }
  *off_out = off;
  return ESVO_OK;
}
int alloc_pose_slot(esvo_context* h, u32* slot) {
  for (u32 i = 0; i < h->n_pose_slots; ++i)
    if (!h->slot_used[i]) { h->slot_used[i] = 1; *slot = i; return ESVO_OK; }
  FAIL(ESVO_ERR_CAPACITY, "no free pose-table slot (too many frames in the fusion window)");
}
void pop_front_frame(esvo_context* h) {
  h->slot_used[h->frames.front().slot] = 0;
  h->frames.pop_front();
}
// window policy, esvo_Mapping.cpp:341-368
void apply_window_policy(esvo_context* h) {
  if (h->prm.fusion_strategy == ESVO_FUSION_CONST_POINTS) {
    auto total = [&]() { size_t s = 0; for (auto& f : h->frames) s += f.count; return s; };
    size_t np = total();
    while ((double)np > 1.5 * (double)h->prm.max_fusion_points) { pop_front_frame(h); np = total(); }
  } else {
    while (h->frames.size() > (size_t)h->prm.max_fusion_frames) pop_front_frame(h);
  }
}

// pose table of the frame: from the host (stage-wise API) or, in a tick, the front stage's device table
int commit_frame(esvo_context* h, u32 off, u32 count, const double* pose_T_host, u32 m, int pose_buf = 0) {
  u32 slot;
  int rc = alloc_pose_slot(h, &slot);
  if (rc) return rc;
  if (m) {
    double* dst = h->d_frame_pose_T + (size_t)slot * h->max_poses * 16;
    if (pose_T_host) {  // through a pinned slot: an async copy from pageable memory would stall the host behind the stream
      const int ps = h->pool_next;
      h->pool_next = (ps + 1) % esvo_context::POSE_POOL;
      HIPCHK(hipEventSynchronize(h->pool_evt[ps]));
      double* pin = h->h_pose_pool + (size_t)ps * h->max_poses * 16;
      std::memcpy(pin, pose_T_host, sizeof(double) * 16 * m);
      HIPCHK(hipMemcpyAsync(dst, pin, sizeof(double) * 16 * m, hipMemcpyHostToDevice, h->stream_b));
      HIPCHK(hipEventRecord(h->pool_evt[ps], h->stream_b));
    } else {
      HIPCHK(hipMemcpyAsync(dst, h->d_pose_T2[pose_buf], sizeof(double) * 16 * m, hipMemcpyDeviceToDevice, h->stream_b));
      HIPCHK(hipEventRecord(h->evt[EV_POSE + pose_buf * EV_BACK_STRIDE], h->stream_b));
    }
  }
  h->frames.push_back(FrameRec{off, count, slot});
  apply_window_policy(h);
  if (h->frames.size() > h->max_frames) FAIL(ESVO_ERR_CAPACITY, "too many frames in the fusion window");
  return ESVO_OK;
}

// fusion loop + clean + regularisation on the current window, on the back stream; `par` selects the
// pinned frame table and the event set (two ticks may be in flight)
int run_fuse(esvo_context* h, int par, const double* T_world_obs) {
  // frames newest -> oldest (esvo_Mapping.cpp:372-377)
  const u32 nf = (u32)h->frames.size();
  const size_t tab = 3 * (size_t)h->max_frames + 1;
  u32* cum = h->h_fr_table + (size_t)par * tab;
  u32* off = cum + (h->max_frames + 1);
  u32* slot = off + h->max_frames;
  u32 total = 0;
  for (u32 i = 0; i < nf; ++i) {
    const FrameRec& f = h->frames[nf - 1 - i];
    cum[i] = total; off[i] = f.off; slot[i] = f.slot;
    total += f.count;
  }
  cum[nf] = total;
  hipStream_t sb = h->stream_b;
  u32* dtab = h->d_fr_table + (size_t)par * tab;
  HIPCHK(hipMemcpyAsync(dtab, cum, sizeof(u32) * tab, hipMemcpyHostToDevice, sb));
  std::memcpy(h->T_world_frame, T_world_obs, sizeof(double) * 16);  // new DepthFrame at the TS pose (:268-272)
  FuseArgs a;
  a.win = h->d_win;
  a.fr_cum = dtab; a.fr_off = dtab + (h->max_frames + 1); a.fr_slot = a.fr_off + h->max_frames;
  a.n_frames = nf; a.n_pts = total;
  a.frame_pose_T = h->d_frame_pose_T; a.max_poses = h->max_poses;
  rigid_inverse(h->T_world_frame, a.T_frame_world);
  a.prop = h->d_prop;
  a.cell_count = h->d_cell_count; a.cell_offset = h->d_cell_offset; a.cell_fill = h->d_cell_fill;
  a.rec_ids = h->d_rec_ids; a.scan_tmp = h->d_scan_tmp_b; a.d_total = h->d_cnt_b + 4;
  a.map = h->d_map; a.d_num_fusion = h->d_cnt_b + 3;
  a.bucket = h->d_bucket; a.cell_list = h->d_cell_list; a.n_touched = h->d_cnt_b + 6;
  a.owner_max = h->prm.regularization ? h->d_owner_max : nullptr;
  a.owner_min = h->d_owner_min; a.n_reg_elems = h->prm.regularization ? h->d_cnt_b + 7 : nullptr;
  if (total > h->win_cap) FAIL(ESVO_ERR_CAPACITY, "window points exceed capacity");
  const int o = par * EV_BACK_STRIDE;
  hipEventRecord(h->evt[EV_FU0 + o], sb);
  launch_fuse(a, h->dp, sb);
  hipEventRecord(h->evt[EV_FU1 + o], sb);
  h->d_map_cur = h->d_map;
  const bool do_clean = h->prm.clean_requires_full_window ? (h->frames.size() >= (size_t)h->prm.max_fusion_frames) : true;
  if (do_clean) launch_clean(h->d_map, h->dp, sb);
  hipEventRecord(h->evt[EV_CL1 + o], sb);
  if (h->prm.regularization) {
    launch_reg_view(h->d_map, h->d_map2, h->d_owner_max, h->d_owner_min, h->d_reg_valid, h->d_reg_ab, h->d_reg_cd,
                    h->d_cell_list, h->d_cnt_b + 7, h->dp, sb);
    launch_reg_apply(h->d_map, h->d_map2, h->d_owner_max, h->d_owner_min, h->d_reg_valid, h->d_reg_bits, h->d_reg_counts,
                     h->d_reg_ab, h->d_reg_cd, h->d_cell_list, h->d_cnt_b + 7,
                     (u32)((size_t)(h->dp.band_y1 - h->dp.band_y0) * h->W), h->dp, sb);
    h->d_map_cur = h->d_map2;
  }
  HIPCHK(hipMemcpyAsync(h->h_cnt_b + 8 * par, h->d_cnt_b, sizeof(u32) * 8, hipMemcpyDeviceToHost, sb));
  hipEventRecord(h->evt[EV_RG1 + o], sb);  // also "back stage of this parity done"
  HIPCHK(hipGetLastError());
  h->back_pending[par] = true;
  return ESVO_OK;
}

int export_map(esvo_context* h, std::vector<esvo_depth_point_t>& out, std::vector<u32>* cells) {
  launch_map_compact(h->d_map_cur, h->d_exp_flags, h->d_exp_prefix, h->d_cnt_b + 5, h->d_scan_tmp_b, h->d_export,
                     h->d_export_cell, h->dp, h->stream_b);
  int rc = read_counters_b(h, 2, true);
  if (rc) return rc;
  const u32 n = h->h_cnt_b[8 * 2 + 5];
  out.resize(n);
  std::vector<u32> cell(n);
  if (n) {
    HIPCHK(hipMemcpy(out.data(), h->d_export, sizeof(esvo_depth_point_t) * n, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(cell.data(), h->d_export_cell, sizeof(u32) * n, hipMemcpyDeviceToHost));
  }
  // the reference iterates its element list in creation order
  std::vector<u32> order(n);
  for (u32 i = 0; i < n; ++i) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](u32 a, u32 b) { return out[a].seq < out[b].seq; });
  std::vector<esvo_depth_point_t> sorted(n);
  if (cells) cells->resize(n);
  for (u32 i = 0; i < n; ++i) {
    sorted[i] = out[order[i]];
    if (!h->sharded) sorted[i].seq = i;  // sharded: keep the global creation id so that bands can be merged
    if (cells) (*cells)[i] = cell[order[i]];
  }
  out.swap(sorted);
  h->stats.last_map_size = n;
  return ESVO_OK;
}

}  // namespace

// =================================================================================================
extern "C" {

void esvo_default_params(esvo_params_t* p) {
  std::memset(p, 0, sizeof(*p));
  p->decay_ms = 30; p->median_blur_kernel_size = 1; p->ignore_polarity = 1;
  p->patch_size_x = 25; p->patch_size_y = 25; p->ls_norm = ESVO_LSNORM_TDIST;
  p->td_nu = 0; p->td_scale = 0; p->lm_max_iteration = 10;
  p->reg_radius = 5; p->reg_min_neighbours = 8; p->reg_min_close_neighbours = 8;
  p->bm_min_disparity = 3; p->bm_max_disparity = 40; p->bm_step = 1; p->bm_zncc_threshold = 0.1;
  p->invdepth_min = 0.16; p->invdepth_max = 2.0; p->stdvar_vis_threshold = 0.005; p->residual_vis_threshold = 15;
  p->age_vis_threshold = 0; p->fusion_radius = 0; p->fusion_strategy = ESVO_FUSION_CONST_FRAMES;
  p->max_fusion_frames = 10; p->max_fusion_points = 2000; p->clean_requires_full_window = 1;
  p->process_event_num = 500; p->bm_half_slice_thickness = 0.001; p->num_threads = 4;
  p->max_events_per_tick = 1024; p->max_window_points = 20000; p->max_poses_per_tick = 256;
  p->event_ring_capacity = 1 << 24;
}

const char* esvo_last_error(esvo_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int esvo_create(const esvo_params_t* params, const esvo_calib_t* left, const esvo_calib_t* right, int device,
                esvo_handle* out) {
  esvo_context* h = nullptr;
  if (!params || !left || !right || !out) FAIL(ESVO_ERR_INVALID_ARG, "null argument");
  if (left->width != right->width || left->height != right->height || left->width <= 0 || left->height <= 0)
    FAIL(ESVO_ERR_INVALID_ARG, "left/right image sizes differ or are empty");
  if (!left->rect_lut || !left->map_x || !left->map_y || !right->map_x || !right->map_y)
    FAIL(ESVO_ERR_INVALID_ARG, "calibration arrays missing");
  {
    std::string why;
    int rc = validate_params(params, why);
    if (rc) FAIL(rc, why);
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    FAIL(ESVO_ERR_NO_DEVICE, "no HIP device visible: the ESVO hot path has no CPU fallback");
  if (device < 0 || device >= ndev) FAIL(ESVO_ERR_INVALID_ARG, "device ordinal out of range");
  HIPCHK(hipSetDevice(device));
  h = new esvo_context();
  h->prm = *params;
  h->device = device;
  h->W = left->width; h->H = left->height;
  const size_t npx = (size_t)h->W * h->H;
  std::memset(&h->stats, 0, sizeof(h->stats));
  std::memcpy(h->dp.camL.P, left->P, sizeof(double) * 12);
  std::memcpy(h->dp.camR.P, right->P, sizeof(double) * 12);
  invert3x3(left->P, h->dp.camL.Kinv, h->dp.camL.Kinv_t);
  invert3x3(right->P, h->dp.camR.Kinv, h->dp.camR.Kinv_t);
  {  // CameraSystem::computeBaseline, CameraSystem.cpp:161-166
    const double* t = h->dp.camR.Kinv_t;
    h->baseline = std::sqrt((t[0] * t[0] + t[1] * t[1]) + t[2] * t[2]);
  }
  h->dp.band_y0 = 0; h->dp.band_y1 = h->H;
  h->dp.cband_y0 = 0; h->dp.cband_y1 = h->H;
  h->dp.ev_shard = 0; h->dp.ev_nshards = 1;
  fill_dev_params(h);

#define CK(call) do { hipError_t _e = (call); if (_e != hipSuccess) { g_create_error = std::string(#call) + ": " + hipGetErrorString(_e); esvo_destroy(h); return ESVO_ERR_HIP; } } while (0)
  {
    int prio_lo = 0, prio_hi = 0;  // numerically lower = higher priority
    CK(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
    // The front stage is the critical path of the two-stream pipeline: its stream gets the high priority, the
    // (latency-bound, gap-filling) back stage the low one: -2 % per tick; the reverse costs +6 %.  ESVO_STREAM_PRIO
    // = 0 (no priorities) / 2 (reversed) exist for that A/B.
    const char* pe = std::getenv("ESVO_STREAM_PRIO");
    const int mode = pe ? std::atoi(pe) : 1;
    CK(hipStreamCreateWithPriority(&h->stream, hipStreamNonBlocking, mode == 1 ? prio_hi : (mode == 2 ? prio_lo : 0)));
    h->own_stream = true;
    CK(hipStreamCreateWithPriority(&h->stream_b, hipStreamNonBlocking, mode == 1 ? prio_lo : (mode == 2 ? prio_hi : 0)));
  }
  CK(hipStreamCreateWithFlags(&h->stream_t, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&h->stream_i, hipStreamNonBlocking));
  // calibration -> device
  CK(dalloc(&h->d_lut, npx));
  CK(hipMemcpy(h->d_lut, left->rect_lut, sizeof(float2) * npx, hipMemcpyHostToDevice));
  if (left->rect_mask) {
    CK(dalloc(&h->d_mask, npx));
    CK(hipMemcpy(h->d_mask, left->rect_mask, npx, hipMemcpyHostToDevice));
  }
  for (int cam = 0; cam < 2; ++cam) {
    const esvo_calib_t* c = cam ? right : left;
    std::vector<int2> fm(npx);
    for (size_t i = 0; i < npx; ++i) {  // OpenCV remap's INTER_BITS=5 coordinate quantisation (Appendix B.2)
      fm[i].x = (int)std::nearbyintf(c->map_x[i] * 32.f);
      fm[i].y = (int)std::nearbyintf(c->map_y[i] * 32.f);
    }
    CK(dalloc(&h->d_fixmap[cam], npx));
    CK(hipMemcpy(h->d_fixmap[cam], fm.data(), sizeof(int2) * npx, hipMemcpyHostToDevice));
    CK(dalloc(&h->d_sae[cam], npx));
    CK(hipMemset(h->d_sae[cam], 0, sizeof(u64) * npx));
    CK(dalloc(&h->d_ts[cam], npx + 64));
    CK(dalloc(&h->d_obs[cam], npx + 64));
  }
  CK(dalloc(&h->d_raw, npx + 64));
  CK(dalloc(&h->d_obs_tmp, npx + 64));
  h->ring_cap = (u64)std::max<int64_t>(params->event_ring_capacity, 1024);
  for (int cam = 0; cam < 2; ++cam) CK(dalloc(&h->d_ring[cam], h->ring_cap));
  CK(dalloc(&h->d_T_world_obs, 16));
  h->max_poses = (u32)std::max(params->max_poses_per_tick, 2);
  CK(dalloc(&h->d_pose_sec, h->max_poses));
  CK(dalloc(&h->d_pose_T2[0], (size_t)h->max_poses * 16));
  CK(dalloc(&h->d_pose_T2[1], (size_t)h->max_poses * 16));
  h->d_pose_T = h->d_pose_T2[0];
  h->max_ev = (u32)std::max(params->max_events_per_tick, params->process_event_num);
  if (h->max_ev > 4000000u) { g_create_error = "max_events_per_tick too large (scan limit 4M)"; esvo_destroy(h); return ESVO_ERR_CAPACITY; }
  if (npx > 4000000u) { g_create_error = "image too large (scan limit 4M pixels)"; esvo_destroy(h); return ESVO_ERR_CAPACITY; }
  const size_t E = h->max_ev;
  CK(dalloc(&h->d_tick_ev, E));
  CK(dalloc(&h->d_match_slots, E));
  CK(dalloc(&h->d_match_flags, E));
  CK(dalloc(&h->d_match_prefix, E));
  CK(dalloc(&h->d_matches, E));
  CK(dalloc(&h->d_pt_slots, E));
  CK(dalloc(&h->d_pt_flags, E));
  CK(dalloc(&h->d_pt_prefix, E));
  CK(dalloc(&h->d_pts_tmp, E));
  CK(dalloc(&h->d_stage[0], E));
  CK(dalloc(&h->d_stage[1], E));
  CK(dalloc(&h->d_counters, 16));
  CK(hipMemset(h->d_counters, 0, sizeof(u32) * 16));
  CK(hipHostMalloc(reinterpret_cast<void**>(&h->h_counters), sizeof(u32) * 16 * 2));
  std::memset(h->h_counters, 0, sizeof(u32) * 16 * 2);
  CK(hipHostMalloc(reinterpret_cast<void**>(&h->h_pin), sizeof(double) * 2 * ((size_t)h->max_poses * 17 + 16)));
  CK(dalloc(&h->d_scan_tmp, scan_scratch_elems(std::max(E, npx)) + 8));
  CK(dalloc(&h->d_scan_tmp_b, scan_scratch_elems(std::max(E, npx)) + 8));
  CK(dalloc(&h->d_cnt_b, 8));
  CK(hipMemset(h->d_cnt_b, 0, sizeof(u32) * 8));
  CK(hipHostMalloc(reinterpret_cast<void**>(&h->h_cnt_b), sizeof(u32) * 8 * 3));
  std::memset(h->h_cnt_b, 0, sizeof(u32) * 8 * 3);
  // fusion window
  h->win_cap = (u32)std::max<int64_t>((int64_t)params->max_window_points, (int64_t)E) + 2 * (u32)E;
  CK(dalloc(&h->d_win, h->win_cap));
  h->max_frames = (u32)std::max(params->max_fusion_frames + 2, 512);
  h->n_pose_slots = h->max_frames + 1;
  h->slot_used.assign(h->n_pose_slots, 0);
  CK(dalloc(&h->d_frame_pose_T, (size_t)h->n_pose_slots * h->max_poses * 16));
  CK(dalloc(&h->d_fr_table, 2 * (3 * (size_t)h->max_frames + 1)));
  CK(hipHostMalloc(reinterpret_cast<void**>(&h->h_fr_table), sizeof(u32) * 2 * (3 * (size_t)h->max_frames + 1)));
  // map
  CK(dalloc(&h->d_prop, h->win_cap));
  CK(dalloc(&h->d_cell_count, npx));
  CK(dalloc(&h->d_cell_offset, npx));
  CK(dalloc(&h->d_cell_fill, npx));
  CK(dalloc(&h->d_rec_ids, (size_t)h->win_cap * 9));
  CK(dalloc(&h->d_map, npx));
  CK(dalloc(&h->d_map2, npx));
  CK(hipMemset(h->d_map, 0, sizeof(MapCell) * npx));
  CK(hipMemset(h->d_map2, 0, sizeof(MapCell) * npx));
  h->d_map_cur = h->d_map;
  CK(dalloc(&h->d_owner_max, npx));
  CK(dalloc(&h->d_owner_min, npx));
  CK(dalloc(&h->d_bucket, 3 * 128));
  CK(dalloc(&h->d_own_w, E));
  CK(dalloc(&h->d_lkeep, E));
  h->codes_bytes = (E + 7) / 8 * 8;
  CK(dalloc(&h->d_codes, h->codes_bytes));
  CK(dalloc(&h->d_sel, E));
  CK(dalloc(&h->d_evmap, npx + 64));
  CK(dalloc(&h->d_reg_valid, npx / 64 + 8));
  CK(hipMemset(h->d_reg_valid, 0, sizeof(u64) * (npx / 64 + 8)));
  CK(dalloc(&h->d_cell_list, npx));
  {
    h->reg_words = (u32)(2 * std::max(params->reg_radius, 1) + 1);  // one mask per tap row and element
    CK(dalloc(&h->d_reg_bits, npx * (size_t)h->reg_words));
    CK(dalloc(&h->d_reg_counts, 2 * npx));
  }
  CK(dalloc(&h->d_reg_ab, npx));
  CK(dalloc(&h->d_reg_cd, npx));
  CK(dalloc(&h->d_exp_flags, npx));
  CK(dalloc(&h->d_exp_prefix, npx));
  CK(dalloc(&h->d_export, npx));
  CK(dalloc(&h->d_export_cell, npx));
  for (int i = 0; i < EV_N; ++i) CK(hipEventCreate(&h->evt[i]));
  h->evt_ok = true;
  CK(hipHostMalloc(reinterpret_cast<void**>(&h->h_pose_pool), sizeof(double) * 16 * (size_t)h->max_poses * esvo_context::POSE_POOL));
  for (int i = 0; i < esvo_context::POSE_POOL; ++i) CK(hipEventCreate(&h->pool_evt[i]));
  h->pool_ok = true;
  for (int i = 0; i < 16; ++i) h->T_world_obs[i] = h->T_world_frame[i] = (i % 5 == 0) ? 1.0 : 0.0;
  CK(hipMemcpy(h->d_T_world_obs, h->T_world_obs, sizeof(double) * 16, hipMemcpyHostToDevice));
#undef CK
  *out = h;
  return ESVO_OK;
}

int esvo_destroy(esvo_handle h) {
  if (!h) return ESVO_OK;
  hipSetDevice(h->device);
  if (h->stream) hipStreamSynchronize(h->stream);
  if (h->stream_b) hipStreamSynchronize(h->stream_b);
  void* ptrs[] = {h->d_lut, h->d_mask, h->d_fixmap[0], h->d_fixmap[1], h->d_sae[0], h->d_sae[1], h->d_raw, h->d_ts[0],
                  h->d_ts[1], h->d_ring[0], h->d_ring[1], h->d_obs[0], h->d_obs[1], h->d_obs_tmp, h->d_T_world_obs,
                  h->d_pose_sec, h->d_pose_T2[0], h->d_pose_T2[1], h->d_scan_tmp_b, h->d_cnt_b, h->d_tick_ev, h->d_match_slots, h->d_match_flags, h->d_match_prefix,
                  h->d_matches, h->d_pt_slots, h->d_pt_flags, h->d_pt_prefix, h->d_pts_tmp, h->d_stage[0], h->d_stage[1], h->d_counters, h->d_scan_tmp,
                  h->d_win, h->d_frame_pose_T, h->d_fr_table, h->d_prop, h->d_cell_count, h->d_cell_offset,
                  h->d_cell_fill, h->d_rec_ids, h->d_map, h->d_map2, h->d_owner_max, h->d_owner_min, h->d_exp_flags,
                  h->d_exp_prefix, h->d_export, h->d_export_cell, h->d_reg_bits, h->d_reg_ab, h->d_reg_cd, h->d_bucket, h->d_cell_list, h->d_own_w, h->d_lkeep, h->d_codes,
                  h->d_reg_valid, h->d_reg_counts, h->d_sel, h->d_evmap};
  for (void* p : ptrs) if (p) hipFree(p);
  if (h->h_counters) hipHostFree(h->h_counters);
  if (h->h_cnt_b) hipHostFree(h->h_cnt_b);
  if (h->h_pin) hipHostFree(h->h_pin);
  if (h->h_fr_table) hipHostFree(h->h_fr_table);
  if (h->evt_ok) for (int i = 0; i < EV_N; ++i) hipEventDestroy(h->evt[i]);
  if (h->pool_ok) for (int i = 0; i < esvo_context::POSE_POOL; ++i) hipEventDestroy(h->pool_evt[i]);
  if (h->h_pose_pool) hipHostFree(h->h_pose_pool);
  if (h->d_wire) hipFree(h->d_wire);
  if (h->own_stream && h->stream) hipStreamDestroy(h->stream);
  if (h->stream_b) hipStreamDestroy(h->stream_b);
  if (h->stream_t) { hipStreamSynchronize(h->stream_t); hipStreamDestroy(h->stream_t); }
  if (h->stream_i) { hipStreamSynchronize(h->stream_i); hipStreamDestroy(h->stream_i); }
  for (void* q : {(void*)h->d_trk_blur, (void*)h->d_trk_neg, (void*)h->d_trk_du, (void*)h->d_trk_dv, (void*)h->d_trk_xyz, (void*)h->d_trk_pts,
                  (void*)h->d_trk_out})
    if (q) hipFree(q);
  delete h;
  return ESVO_OK;
}

int esvo_reset(esvo_handle h) {
  if (!h) return ESVO_ERR_INVALID_ARG;
  HIPCHK(hipSetDevice(h->device));
  { int rcp = flush_pending_tick(h); if (rcp) return rcp; }
  const size_t npx = (size_t)h->W * h->H;
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipStreamSynchronize(h->stream_b));
  for (int cam = 0; cam < 2; ++cam) {
    HIPCHK(hipMemsetAsync(h->d_sae[cam], 0, sizeof(u64) * npx, h->stream));
    h->ts_host[cam].clear();
    h->ring_base[cam] = h->ring_next[cam] = h->scattered[cam] = 0;
    h->ts_valid[cam] = false;
  }
  h->frames.clear();
  std::fill(h->slot_used.begin(), h->slot_used.end(), 0);
  HIPCHK(hipMemsetAsync(h->d_map, 0, sizeof(MapCell) * npx, h->stream));
  HIPCHK(hipMemsetAsync(h->d_map2, 0, sizeof(MapCell) * npx, h->stream));
  h->d_map_cur = h->d_map;
  h->obs_set = false;
  h->n_pose = 0;
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipStreamSynchronize(h->stream_b));
  h->back_pending[0] = h->back_pending[1] = false;
  h->committed_t_ns = 0;
  h->ts_timing_pending[0] = h->ts_timing_pending[1] = false;
  h->stats_pending = false;
  std::memset(&h->stats, 0, sizeof(h->stats));
  return ESVO_OK;
}

int esvo_set_params(esvo_handle h, const esvo_params_t* params) {
  if (!h || !params) return ESVO_ERR_INVALID_ARG;
  { int rcp = flush_pending_tick(h); if (rcp) return rcp; }
  std::string why;
  int rc = validate_params(params, why);
  if (rc) FAIL(rc, why);
  if ((u32)std::max(params->max_events_per_tick, params->process_event_num) > h->max_ev)
    FAIL(ESVO_ERR_CAPACITY, "process_event_num exceeds the capacity fixed at esvo_create");
  {
    if ((u32)(2 * std::max(params->reg_radius, 1) + 1) > h->reg_words) FAIL(ESVO_ERR_CAPACITY, "RegularizationRadius exceeds the capacity fixed at esvo_create");
  }
  esvo_params_t np = *params;
  np.max_events_per_tick = h->prm.max_events_per_tick;
  np.max_window_points = h->prm.max_window_points;
  np.max_poses_per_tick = h->prm.max_poses_per_tick;
  np.event_ring_capacity = h->prm.event_ring_capacity;
  h->prm = np;
  fill_dev_params(h);
  set_compute_band(h);
  return ESVO_OK;
}

int esvo_set_stream(esvo_handle h, void* hip_stream) {
  if (!h) return ESVO_ERR_INVALID_ARG;
  { int rcp = flush_pending_tick(h); if (rcp) return rcp; }
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipStreamSynchronize(h->stream_b));
  if (h->own_stream && h->stream) hipStreamDestroy(h->stream);
  h->stream = reinterpret_cast<hipStream_t>(hip_stream);
  h->own_stream = false;
  return ESVO_OK;
}

int esvo_synchronize(esvo_handle h) {
  if (!h) return ESVO_ERR_INVALID_ARG;
  { int rcp = flush_pending_tick(h); if (rcp) return rcp; }
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipStreamSynchronize(h->stream_b));
  return ESVO_OK;
}

// ---- Time Surface ---------------------------------------------------------------------------------
namespace {
// Staging runs on its own stream.  The slots it overwrites hold events older than ring_cap; a front stage still in flight
// reads at most the max_ev events before its selection point, so only a (nearly) full ring needs the front stream drained.
int ring_overwrite_guard(esvo_context* h, int cam, size_t n) {
  if (h->ring_next[cam] + n <= h->ring_cap) return ESVO_OK;
  const u64 evict_end = h->ring_next[cam] + n - h->ring_cap;  // first absolute index that survives
  u64 oldest_read = h->scatter_pending_lo[cam];              // scatter kernels enqueued since the last drain
  if (cam == 0) oldest_read = std::min(oldest_read, h->sh_first > (u64)h->max_ev ? h->sh_first - (u64)h->max_ev : 0);
  if (evict_end > oldest_read) {
    HIPCHK(hipStreamSynchronize(h->stream));
    h->scatter_pending_lo[0] = h->scatter_pending_lo[1] = ~0ull;
  }
  return ESVO_OK;
}
}  // namespace

int esvo_ts_push_events(esvo_handle h, int cam, const esvo_event_t* ev, size_t n) {
  if (!h || cam < 0 || cam > 1 || (n && !ev)) return ESVO_ERR_INVALID_ARG;
  if (n == 0) return ESVO_OK;
  if (n > h->ring_cap) FAIL(ESVO_ERR_CAPACITY, "event block larger than the event ring");
  HIPCHK(hipSetDevice(h->device));
  auto& tsq = h->ts_host[cam];
  u64 last = tsq.empty() ? 0 : tsq.back();
  for (size_t i = 0; i < n; ++i) {
    const u64 t = (u64)ev[i].sec * 1000000000ull + ev[i].nsec;
    if (t < last) FAIL(ESVO_ERR_INVALID_ARG, "events must be sorted by time stamp (SURVEY Appendix A-1)");
    last = t;
  }
  // the ring must not overwrite events that are not yet scattered into the SAE
  if (h->ring_next[cam] + n - h->scattered[cam] > h->ring_cap)
    FAIL(ESVO_ERR_CAPACITY, "event ring full: render (scatter) before staging more events");
  { int rcg = ring_overwrite_guard(h, cam, n); if (rcg) return rcg; }
  const u64 slot = h->ring_next[cam] % h->ring_cap;
  const size_t first = (size_t)std::min<u64>(n, h->ring_cap - slot);
  HIPCHK(hipMemcpyAsync(h->d_ring[cam] + slot, ev, sizeof(esvo_event_t) * first, hipMemcpyHostToDevice, h->stream_i));
  if (first < n)
    HIPCHK(hipMemcpyAsync(h->d_ring[cam], ev + first, sizeof(esvo_event_t) * (n - first), hipMemcpyHostToDevice, h->stream_i));
  HIPCHK(hipStreamSynchronize(h->stream_i));  // `ev` is borrowed for the duration of the call only; later work sees the copy
  for (size_t i = 0; i < n; ++i) tsq.push_back((u64)ev[i].sec * 1000000000ull + ev[i].nsec);
  h->ring_next[cam] += n;
  while (tsq.size() > h->ring_cap) { tsq.pop_front(); h->ring_base[cam]++; }
  h->stats.events_staged[cam] += n;
  return ESVO_OK;
}

// A serialised dvs_msgs/EventArray (ROS1 wire format): std_msgs/Header {u32 seq, u32 sec, u32 nsec, string frame_id},
// u32 height, u32 width, Event[] {u32 count, count x 13 B}.  The 13-byte records go to the device as they are and are
// widened to esvo_event_t in the ring by a kernel; the host only walks the time stamps (order check + selection index).
int esvo_ts_push_event_array(esvo_handle h, int cam, const uint8_t* msg, size_t n_bytes, size_t* n_events) {
  if (!h || cam < 0 || cam > 1 || !msg) return ESVO_ERR_INVALID_ARG;
  auto rd32 = [&](size_t off) { return (u32)msg[off] | ((u32)msg[off + 1] << 8) | ((u32)msg[off + 2] << 16) | ((u32)msg[off + 3] << 24); };
  if (n_bytes < 16) FAIL(ESVO_ERR_INVALID_ARG, "EventArray message shorter than its header");
  const size_t id_len = rd32(12);
  size_t off = 16 + id_len;
  if (off < 16 || off + 12 > n_bytes) FAIL(ESVO_ERR_INVALID_ARG, "EventArray message truncated (frame_id / height / width / count)");
  const u32 height = rd32(off), width = rd32(off + 4), n = rd32(off + 8);
  off += 12;
  if (n_events) *n_events = n;
  if ((size_t)n * 13 != n_bytes - off) FAIL(ESVO_ERR_INVALID_ARG, "EventArray message length does not match its event count");
  if ((height && (int)height != h->H) || (width && (int)width != h->W)) FAIL(ESVO_ERR_INVALID_ARG, "EventArray sensor size differs from the handle's");
  if (n == 0) return ESVO_OK;
  if (n > h->ring_cap) FAIL(ESVO_ERR_CAPACITY, "event block larger than the event ring");
  HIPCHK(hipSetDevice(h->device));
  const uint8_t* rec = msg + off;
  auto stamp = [&](size_t i) {
    const uint8_t* r = rec + i * 13 + 4;
    const u32 sec = (u32)r[0] | ((u32)r[1] << 8) | ((u32)r[2] << 16) | ((u32)r[3] << 24);
    const u32 nsec = (u32)r[4] | ((u32)r[5] << 8) | ((u32)r[6] << 16) | ((u32)r[7] << 24);
    return (u64)sec * 1000000000ull + nsec;
  };
  auto& tsq = h->ts_host[cam];
  u64 last = tsq.empty() ? 0 : tsq.back();
  for (size_t i = 0; i < n; ++i) {
    const u64 t = stamp(i);
    if (t < last) FAIL(ESVO_ERR_INVALID_ARG, "events must be sorted by time stamp (SURVEY Appendix A-1)");
    last = t;
  }
  if (h->ring_next[cam] + n - h->scattered[cam] > h->ring_cap)
    FAIL(ESVO_ERR_CAPACITY, "event ring full: render (scatter) before staging more events");
  if ((size_t)n * 13 > h->wire_cap) {
    if (h->d_wire) { hipFree(h->d_wire); h->d_wire = nullptr; }  // the ingest stream is idle between calls
    h->wire_cap = std::max<size_t>((size_t)n * 13, (size_t)1 << 20);
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&h->d_wire), h->wire_cap));
  }
  { int rcg = ring_overwrite_guard(h, cam, n); if (rcg) return rcg; }
  HIPCHK(hipMemcpyAsync(h->d_wire, rec, (size_t)n * 13, hipMemcpyHostToDevice, h->stream_i));
  launch_ts_unpack_wire(h->d_wire, n, h->d_ring[cam], h->ring_next[cam] % h->ring_cap, h->ring_cap, h->stream_i);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(h->stream_i));  // `msg` is borrowed for the duration of the call only
  for (size_t i = 0; i < n; ++i) tsq.push_back(stamp(i));
  h->ring_next[cam] += n;
  while (tsq.size() > h->ring_cap) { tsq.pop_front(); h->ring_base[cam]++; }
  h->stats.events_staged[cam] += n;
  return ESVO_OK;
}

int esvo_ts_render(esvo_handle h, int cam, uint64_t t_ns, uint8_t* out_mono8) {
  if (!h || cam < 0 || cam > 1) return ESVO_ERR_INVALID_ARG;
  HIPCHK(hipSetDevice(h->device));
  // events with ts < T (strict, TimeSurface.h:68) that are not in the SAE yet
  const auto& tsq = h->ts_host[cam];
  const size_t k = std::lower_bound(tsq.begin(), tsq.end(), (u64)t_ns) - tsq.begin();
  const u64 upto = h->ring_base[cam] + k;
  const int evo = cam * EV_TS_STRIDE;
  if (h->ts_timing_pending[cam] && hipEventQuery(h->evt[EV_R1 + evo]) == hipSuccess) collect_ts_timing(h, cam);
  hipEventRecord(h->evt[EV_SC0 + evo], h->stream);
  if (upto > h->scattered[cam]) {
    u64 a = h->scattered[cam];
    h->scatter_pending_lo[cam] = std::min(h->scatter_pending_lo[cam], a);
    const u64 total = upto - a;
    while (a < upto) {
      const u64 slot = a % h->ring_cap;
      const u64 cnt = std::min<u64>(upto - a, h->ring_cap - slot);
      launch_ts_scatter(h->d_ring[cam] + slot, (size_t)cnt, h->d_sae[cam], h->W, h->H, h->stream);
      a += cnt;
    }
    h->scattered[cam] = upto;
    h->stats.events_scattered[cam] += total;
  }
  hipEventRecord(h->evt[EV_SC1 + evo], h->stream);
  launch_ts_render(h->d_sae[cam], h->d_fixmap[cam], h->d_raw, h->d_ts[cam], h->W, h->H, (u64)t_ns, h->prm.decay_ms / 1000.0,
                   h->prm.ignore_polarity, h->prm.median_blur_kernel_size, h->stream);
  hipEventRecord(h->evt[EV_R1 + evo], h->stream);
  HIPCHK(hipGetLastError());
  h->ts_valid[cam] = true;
  h->ts_timing_pending[cam] = true;
  h->stats.ts_frames[cam]++;
  if (out_mono8) {
    HIPCHK(hipMemcpyAsync(out_mono8, h->d_ts[cam], (size_t)h->W * h->H, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    collect_ts_timing(h);
  }
  return ESVO_OK;
}

// ---- Mapper: stage-wise ---------------------------------------------------------------------------
int esvo_map_set_observation(esvo_handle h, uint64_t t_ns, const uint8_t* ts_left, const uint8_t* ts_right,
                             const double T_world_cam[16]) {
  if (!h || !T_world_cam) return ESVO_ERR_INVALID_ARG;
  HIPCHK(hipSetDevice(h->device));
  const size_t npx = (size_t)h->W * h->H;
  const uint8_t* src[2] = {ts_left, ts_right};
  for (int cam = 0; cam < 2; ++cam) {
    uint8_t* dst = h->prm.smooth_time_surface ? h->d_obs_tmp : h->d_obs[cam];
    if (src[cam]) {
      HIPCHK(hipMemcpyAsync(dst, src[cam], npx, hipMemcpyHostToDevice, h->stream));
      HIPCHK(hipStreamSynchronize(h->stream));
    } else {
      if (!h->ts_valid[cam]) FAIL(ESVO_ERR_STATE, "no device-resident Time Surface: call esvo_ts_render first");
      HIPCHK(hipMemcpyAsync(dst, h->d_ts[cam], npx, hipMemcpyDeviceToDevice, h->stream));
    }
    // createMatchProblem applies GaussianBlurTS(5) when SmoothTimeSurface (EventBM.cpp:68-72)
    if (h->prm.smooth_time_surface) launch_gaussian5(h->d_obs_tmp, h->d_obs[cam], h->W, h->H, h->stream);
  }
  std::memcpy(h->T_world_obs, T_world_cam, sizeof(double) * 16);
  {
    double* pinT = h->h_pin + (size_t)(h->pin_slot ^ 1) * ((size_t)h->max_poses * 17 + 16) + (size_t)h->max_poses * 17;
    std::memcpy(pinT, T_world_cam, sizeof(double) * 16);
    HIPCHK(hipMemcpyAsync(h->d_T_world_obs, pinT, sizeof(double) * 16, hipMemcpyHostToDevice, h->stream));
  }
  h->obs_t_ns = t_ns;
  h->obs_set = true;
  return ESVO_OK;
}

int esvo_map_set_poses(esvo_handle h, const uint64_t* pose_t_ns, const double* pose_T, size_t m) {
  if (!h || (m && (!pose_t_ns || !pose_T))) return ESVO_ERR_INVALID_ARG;
  HIPCHK(hipSetDevice(h->device));
  return upload_poses(h, pose_t_ns, pose_T, m);
}

int esvo_map_match(esvo_handle h, const esvo_event_t* ev, size_t n, const uint64_t* pose_t_ns, const double* pose_T,
                   size_t m, esvo_match_t* out, size_t cap, size_t* n_out) {
  if (!h || (n && !ev) || !n_out) return ESVO_ERR_INVALID_ARG;
  if (!h->obs_set) FAIL(ESVO_ERR_STATE, "esvo_map_set_observation has not been called");
  if (n > h->max_ev) FAIL(ESVO_ERR_CAPACITY, "more events than max_events_per_tick");
  HIPCHK(hipSetDevice(h->device));
  { int rcp = flush_pending_tick(h); if (rcp) return rcp; }
  if (pose_t_ns) { int rc = upload_poses(h, pose_t_ns, pose_T, m); if (rc) return rc; }
  *n_out = 0;
  if (n == 0) { HIPCHK(hipMemsetAsync(h->d_counters, 0, sizeof(u32), h->stream)); return ESVO_OK; }
  HIPCHK(hipMemcpyAsync(h->d_tick_ev, ev, sizeof(esvo_event_t) * n, hipMemcpyHostToDevice, h->stream));
  int rc = run_match(h, h->d_tick_ev, 0, (u64)h->max_ev, 0, (u32)n);
  if (rc) return rc;
  rc = read_counters(h);
  if (rc) return rc;
  const u32 nm = h->h_counters[0];
  *n_out = nm;
  h->stats.last_events_in = (u32)n;
  h->stats.last_matches = nm;
  if (out && nm) {
    if (nm > cap) FAIL(ESVO_ERR_CAPACITY, "output array too small for the matches");
    HIPCHK(hipMemcpy(out, h->d_matches, sizeof(esvo_match_t) * nm, hipMemcpyDeviceToHost));
  }
  return ESVO_OK;
}

int esvo_map_refine(esvo_handle h, const esvo_match_t* matches, size_t n, int cull, esvo_depth_point_t* out, size_t cap,
                    size_t* n_out) {
  if (!h || (n && !matches) || !n_out) return ESVO_ERR_INVALID_ARG;
  if (!h->obs_set) FAIL(ESVO_ERR_STATE, "esvo_map_set_observation has not been called");
  if (n > h->max_ev) FAIL(ESVO_ERR_CAPACITY, "more matches than max_events_per_tick");
  for (size_t i = 0; i < n; ++i)
    if (matches[i].pose_idx >= h->n_pose) FAIL(ESVO_ERR_INVALID_ARG, "match refers to a pose outside the pose table");
  HIPCHK(hipSetDevice(h->device));
  { int rcp = flush_pending_tick(h); if (rcp) return rcp; }
  *n_out = 0;
  if (n == 0) return ESVO_OK;
  const u32 n32 = (u32)n;
  HIPCHK(hipMemcpyAsync(h->d_matches, matches, sizeof(esvo_match_t) * n, hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(h->d_counters, &n32, sizeof(u32), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  int rc = run_refine(h, n32, cull, h->d_pts_tmp);
  if (rc) return rc;
  rc = read_counters(h);
  if (rc) return rc;
  const u32 np = h->h_counters[1];
  *n_out = np;
  h->stats.last_solved = h->h_counters[2];
  h->stats.last_points = np;
  if (out && np) {
    if (np > cap) FAIL(ESVO_ERR_CAPACITY, "output array too small for the depth points");
    HIPCHK(hipMemcpy(out, h->d_pts_tmp, sizeof(esvo_depth_point_t) * np, hipMemcpyDeviceToHost));
  }
  return ESVO_OK;
}

int esvo_map_push_frame(esvo_handle h, const esvo_depth_point_t* pts, size_t n, const double* pose_T, size_t m) {
  if (!h || (n && !pts) || (m && !pose_T)) return ESVO_ERR_INVALID_ARG;
  if (m > h->max_poses) FAIL(ESVO_ERR_CAPACITY, "pose table larger than max_poses_per_tick");
  for (size_t i = 0; i < n; ++i)
    if (pts[i].pose_idx >= m) FAIL(ESVO_ERR_INVALID_ARG, "depth point refers to a pose outside the frame's pose table");
  HIPCHK(hipSetDevice(h->device));
  { int rcp = flush_pending_tick(h); if (rcp) return rcp; }
  u32 off;
  int rc = window_reserve(h, (u32)n, &off);
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(h->stream_b));  // the ring space may have been read by a fusion still in flight
  if (n) HIPCHK(hipMemcpyAsync(h->d_win + off, pts, sizeof(esvo_depth_point_t) * n, hipMemcpyHostToDevice, h->stream));
  static const double ident[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  rc = commit_frame(h, off, (u32)n, m ? pose_T : ident, (u32)m);
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipStreamSynchronize(h->stream_b));
  return ESVO_OK;
}

int esvo_map_fuse(esvo_handle h, size_t* n_fusions) {
  if (!h) return ESVO_ERR_INVALID_ARG;
  if (!h->obs_set) FAIL(ESVO_ERR_STATE, "esvo_map_set_observation has not been called");
  HIPCHK(hipSetDevice(h->device));
  { int rcp = flush_pending_tick(h); if (rcp) return rcp; }
  int rc = back_after_front(h);
  if (rc) return rc;
  const int par = h->par;
  h->par ^= 1;
  HIPCHK(hipEventSynchronize(h->evt[EV_RG1 + par * EV_BACK_STRIDE]));
  collect_back(h, par);
  rc = run_fuse(h, par, h->T_world_obs);
  if (rc) return rc;
  h->committed_t_ns = h->obs_t_ns;
  HIPCHK(hipStreamSynchronize(h->stream_b));
  collect_back(h, par);
  h->stats.last_window_frames = (u32)h->frames.size();
  u32 np = 0;
  for (auto& f : h->frames) np += f.count;
  h->stats.last_window_points = np;
  if (n_fusions) *n_fusions = h->stats.last_fusions;
  return ESVO_OK;
}

}  // extern "C"

// ---- Mapper: fused tick ---------------------------------------------------------------------------
namespace {
// event selection, esvo_Mapping.cpp:562-574 (Appendix A-3): walk back from lower_bound(t_end) to
// lower_bound(t_begin), newest first, at most PROCESS_EVENT_NUM
int select_events(esvo_context* h, uint64_t t_ns, u64* first_out, u32* n_out) {
  const double t_end = ns_to_sec(t_ns);
  const u64 t_begin_ns = ros_time_from_sec(std::max(0.0, t_end - 10 * h->prm.bm_half_slice_thickness));
  const double t_begin = ns_to_sec(t_begin_ns);
  u64 it_end = lower_bound_sec(h, 0, t_end);
  const u64 it_begin = lower_bound_sec(h, 0, t_begin);
  const u64 staged_end = h->ring_base[0] + h->ts_host[0].size();
  u64 avail = it_end - it_begin;
  u64 first = it_end;
  if (it_end == staged_end && avail > 0) { first = it_end - 1; avail -= 1; }  // end() is skipped (oracle definition)
  const u32 n = (u32)std::min<u64>(avail, (u64)h->prm.process_event_num);
  if (n > h->max_ev) FAIL(ESVO_ERR_CAPACITY, "more events than max_events_per_tick");
  if (n && first - (n - 1) < h->ring_next[0] - std::min<u64>(h->ring_next[0], h->ring_cap))
    FAIL(ESVO_ERR_STATE, "selected events were already overwritten in the event ring");
  *first_out = first;
  *n_out = n;
  return ESVO_OK;
}

// phase 0 (front stage): poses, event selection, block matching + LM of the events of this handle's shard
int tick_phase0(esvo_context* h, uint64_t t_ns, const uint64_t* pose_t_ns, const double* pose_T, size_t m) {
  int rc = upload_poses(h, pose_t_ns, pose_T, m);
  if (rc) return rc;
  u32 n = 0;
  rc = select_events(h, t_ns, &h->sh_first, &n);
  if (rc) return rc;
  h->fpar ^= 1;
  esvo_context::TickState& tk = h->tk[h->fpar];
  tk.n = n; tk.off = 0; tk.points = 0; tk.t_ns = t_ns;
  tk.pose_buf = h->pose_buf; tk.n_pose = h->n_pose;
  std::memcpy(tk.T_world_obs, h->T_world_obs, sizeof(double) * 16);
  // two ticks in flight at most: what this tick's front stage overwrites (ring space of popped frames, the pose
  // table buffer) was last read by the back stage two ticks ago
  HIPCHK(hipStreamWaitEvent(h->stream, h->evt[EV_RG1 + h->par * EV_BACK_STRIDE], 0));
  hipEventRecord(h->evt[EV_T0 + h->fpar * EV_FRONT_STRIDE], h->stream);
  HIPCHK(hipMemsetAsync(h->d_counters, 0, sizeof(u32) * 16, h->stream));
  const u32* sel = nullptr;
  if (h->prm.denoising && n) {
    // Denoising (esvo_Mapping.cpp:282-296): mask from the selected events, keep those on it, in order.
    // One extra read-back (the kept count sizes the BM launch); only the small DAVIS configs use it.
    launch_denoise_flags(h->d_ring[0], h->sh_first, h->ring_cap, n, h->d_evmap, h->d_match_flags, h->W, h->H, h->stream);
    launch_exclusive_scan_u32(h->d_match_flags, h->d_match_prefix, h->d_counters + 5, h->d_scan_tmp, n, h->stream);
    launch_denoise_select(h->d_match_flags, h->d_match_prefix, n, h->d_sel, h->stream);
    rc = read_counters(h);
    if (rc) return rc;
    n = tk.n = h->h_counters[5];
    sel = h->d_sel;
  }
  h->xchg_ptr = nullptr;
  h->xchg_bytes = 0;
  if (n && !h->sharded) {
    rc = run_bm(h, h->d_ring[0], h->sh_first, h->ring_cap, 1, n, sel);
    if (rc) return rc;
    rc = run_order_matches(h, n, false);
    if (rc) return rc;
    rc = run_lm(h, n, 1, false);
    if (rc) return rc;
  } else if (n) {
    // own slots only (w % n_shards == shard): BM, dense local list, LM + cull on it, then the (matched, kept)
    // byte of every own slot; the other ranks' bytes stay zero and arrive with the caller's sum
    const u32 N = (u32)h->dp.ev_nshards, r = (u32)h->dp.ev_shard;
    const u32 own = n > r ? (n - r + N - 1) / N : 0;
    HIPCHK(hipMemsetAsync(h->d_match_flags, 0, sizeof(u32) * n, h->stream));
    rc = run_bm(h, h->d_ring[0], h->sh_first, h->ring_cap, 1, n, sel);
    if (rc) return rc;
    rc = run_order_matches(h, n, true);
    if (rc) return rc;
    rc = run_lm(h, own, 1, true);
    if (rc) return rc;
    const size_t nb = ((size_t)n + 7) / 8 * 8;
    HIPCHK(hipMemsetAsync(h->d_codes, 0, nb, h->stream));
    launch_shard_codes(h->d_own_w, h->d_lkeep, h->d_counters + 8, own, h->d_codes, h->stream);
    HIPCHK(hipGetLastError());
    h->xchg_ptr = h->d_codes;
    h->xchg_bytes = nb;
  }
  return ESVO_OK;
}
// phase 1a (front stage, enqueue only): the tick's frame (culled points in the reference's order) goes straight
// into the window ring (capacity for the worst case: n points); the counters follow into the pinned row of the
// tick's parity and EV_CNT marks "frame and counters ready"
int tick_phase1_enqueue(esvo_context* h) {
  esvo_context::TickState& tk = h->tk[h->fpar];
  const u32 n = tk.n;
  int rc = ESVO_OK;
  DevPoint* frame = nullptr;
  if (h->sharded) {  // committed right away: straight into the ring (worst case n points)
    rc = window_reserve(h, n, &tk.off);
    if (rc) return rc;
    frame = h->d_win + tk.off;
  }
  h->xchg_ptr = nullptr;
  h->xchg_bytes = 0;
  if (n && !h->sharded) {
    // the frame waits in the staging buffer of its parity until the tick is committed and its size is known; the
    // buffer's previous frame (two ticks ago) has been copied into the ring by then
    HIPCHK(hipStreamWaitEvent(h->stream, h->evt[EV_STG + h->fpar * EV_FRONT_STRIDE], 0));
    rc = run_order_points(h, n, h->d_stage[h->fpar]);
    if (rc) return rc;
  } else if (n) {
    const u32 N = (u32)h->dp.ev_nshards, r = (u32)h->dp.ev_shard, T = (u32)h->dp.num_threads;
    const u32 own = n > r ? (n - r + N - 1) / N : 0;
    launch_shard_match_flags(h->d_codes, n, h->d_match_flags, h->stream);
    launch_exclusive_scan_u32(h->d_match_flags, h->d_match_prefix, h->d_counters + 0, h->d_scan_tmp, n, h->stream);
    HIPCHK(hipMemsetAsync(h->d_pt_flags, 0, sizeof(u32) * n, h->stream));
    launch_shard_keep_flags(h->d_codes, h->d_match_prefix, h->d_counters + 0, n, T, h->d_pt_flags, h->stream);
    launch_exclusive_scan_u32(h->d_pt_flags, h->d_pt_prefix, h->d_counters + 1, h->d_scan_tmp, n, h->stream);
    launch_shard_place(h->d_own_w, h->d_lkeep, h->d_pt_slots, h->d_counters + 8, own, h->d_match_prefix, h->d_counters + 0,
                       h->d_pt_prefix, h->d_counters + 1, T, frame, n, h->stream);
    hipEventRecord(h->evt[EV_S2 + h->fpar * EV_FRONT_STRIDE], h->stream);
    HIPCHK(hipGetLastError());
  }
  HIPCHK(hipMemcpyAsync(h->h_counters + 16 * h->fpar, h->d_counters, sizeof(u32) * 16, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipEventRecord(h->evt[EV_CNT + h->fpar * EV_FRONT_STRIDE], h->stream));
  h->tick_pending = true;
  return ESVO_OK;
}
// phase 1b (host): wait for the counters of the tick of parity fp (one small D2H per tick: the window policy
// needs the point count), book-keeping, front-stage timings
int tick_phase1_collect(esvo_context* h, int fp) {
  esvo_context::TickState& tk = h->tk[fp];
  HIPCHK(hipEventSynchronize(h->evt[EV_CNT + fp * EV_FRONT_STRIDE]));
  const u32* cnt = h->h_counters + 16 * fp;
  const u32 n = tk.n;
  const u32 n_points = n ? cnt[1] : 0;
  esvo_stats_t& s = h->stats;
  s.last_events_in = n;
  s.last_matches = cnt[0];
  s.last_solved = cnt[2];  // sharded: this rank's share
  s.last_points = n_points;
  s.total_events_in += n;
  s.total_matches += cnt[0];
  s.total_points += n_points;
  tk.points = n_points;
  const int o = fp * EV_FRONT_STRIDE;
  s.ms_bm = s.ms_refine = 0;
  s.ms_kernel[2] = s.ms_kernel[3] = 0;
  if (n) {
    hipEventElapsedTime(&s.ms_bm, h->evt[EV_T0 + o], h->evt[EV_S1 + o]);
    hipEventElapsedTime(&s.ms_refine, h->evt[EV_S1 + o], h->evt[EV_S2 + o]);
    hipEventElapsedTime(&s.ms_kernel[2], h->evt[EV_BM0 + o], h->evt[EV_BM1 + o]);
    hipEventElapsedTime(&s.ms_kernel[3], h->evt[EV_LM0 + o], h->evt[EV_LM1 + o]);
    s.sum_ms_kernel[2] += s.ms_kernel[2];
    s.sum_ms_kernel[3] += s.ms_kernel[3];
  }
  if (h->sharded && n_points) {
    h->xchg_ptr = h->d_win + tk.off;
    h->xchg_bytes = (size_t)n_points * sizeof(DevPoint);
  }
  return ESVO_OK;
}
// phase 2 (back stage): window policy, fusion + clean + regularisation of this band (halo rows recomputed locally),
// enqueued on the back stream behind the frame of the tick of parity fp.  Nothing here waits for the GPU except for
// the back stage of two ticks ago (long finished), whose pinned table and event set are reused; its timings are
// collected then.
int tick_phase2(esvo_context* h, int fp) {
  esvo_context::TickState& tk = h->tk[fp];
  h->xchg_ptr = nullptr;
  h->xchg_bytes = 0;
  if (h->sharded) {  // the caller's frame sum was issued on the front stream after EV_CNT
    int rc = back_after_front(h);
    if (rc) return rc;
  } else {
    HIPCHK(hipStreamWaitEvent(h->stream_b, h->evt[EV_CNT + fp * EV_FRONT_STRIDE], 0));
  }
  const int par = h->par;
  h->par ^= 1;
  HIPCHK(hipEventSynchronize(h->evt[EV_RG1 + par * EV_BACK_STRIDE]));
  collect_back(h, par);
  int rc;
  if (!h->sharded) {  // now that the size is known: exact ring space, frame copied behind the fusion that may still read it
    rc = window_reserve(h, tk.points, &tk.off);
    if (rc) return rc;
    if (tk.points)
      HIPCHK(hipMemcpyAsync(h->d_win + tk.off, h->d_stage[fp], sizeof(DevPoint) * tk.points, hipMemcpyDeviceToDevice, h->stream_b));
    HIPCHK(hipEventRecord(h->evt[EV_STG + fp * EV_FRONT_STRIDE], h->stream_b));
  }
  rc = commit_frame(h, tk.off, tk.points, nullptr, tk.n_pose, tk.pose_buf);
  if (rc) return rc;
  rc = run_fuse(h, par, tk.T_world_obs);
  if (rc) return rc;
  h->stats.ticks++;
  h->stats.last_window_frames = (u32)h->frames.size();
  u32 np = 0;
  for (auto& f : h->frames) np += f.count;
  h->stats.last_window_points = np;
  h->stats_pending = true;
  h->tick_pending = false;
  h->committed_t_ns = tk.t_ns;
  return ESVO_OK;
}
// complete the tick whose front stage is enqueued but which is not committed yet (unsharded ticks are lazy)
int flush_pending_tick(esvo_context* h) {
  if (!h->tick_pending || h->sharded) return ESVO_OK;
  const int fp = h->fpar;
  h->tick_pending = false;  // also when completing it fails (e.g. window ring full): the error is reported once
  int rc = tick_phase1_collect(h, fp);
  if (rc) return rc;
  return tick_phase2(h, fp);
}
// drain the back stream and collect what is pending (older parity first)
int finalize_tick_stats(esvo_context* h) {
  int rcf = flush_pending_tick(h);
  if (rcf) return rcf;
  if (!h->stats_pending && !h->back_pending[0] && !h->back_pending[1]) return ESVO_OK;
  const bool tick_done = h->stats_pending;
  h->stats_pending = false;
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipStreamSynchronize(h->stream_b));
  collect_ts_timing(h);
  collect_back(h, h->par);
  collect_back(h, h->par ^ 1);
  if (tick_done) hipEventElapsedTime(&h->stats.ms_tick_total, h->evt[EV_T0 + h->fpar * EV_FRONT_STRIDE], h->evt[EV_RG1 + (h->par ^ 1) * EV_BACK_STRIDE]);
  return ESVO_OK;
}
}  // namespace

extern "C" int esvo_map_tick(esvo_handle h, uint64_t t_ns, const uint64_t* pose_t_ns, const double* pose_T, size_t m) {
  if (!h || !pose_t_ns || !pose_T) return ESVO_ERR_INVALID_ARG;
  if (!h->obs_set) FAIL(ESVO_ERR_STATE, "esvo_map_set_observation has not been called");
  if (h->sharded) FAIL(ESVO_ERR_STATE, "handle is sharded: drive it with esvo_shard_tick_phase");
  HIPCHK(hipSetDevice(h->device));
  // the previous tick (if still pending) is completed AFTER this tick's front stage is enqueued: its point count
  // arrived long ago, and the front stream never runs dry while the host works
  if (h->prm.denoising) {  // its kept-event count is read back inside phase 0: no point in deferring anything
    int rcp = flush_pending_tick(h);
    if (rcp) return rcp;
  }
  const bool prev = h->tick_pending;
  const int prev_fp = h->fpar;
  int rc = tick_phase0(h, t_ns, pose_t_ns, pose_T, m);
  if (rc) return rc;
  rc = tick_phase1_enqueue(h);
  if (rc) return rc;
  if (prev) {
    rc = tick_phase1_collect(h, prev_fp);
    if (!rc) rc = tick_phase2(h, prev_fp);
    h->tick_pending = true;  // this tick (its front stage is enqueued whatever happened to the previous one)
    if (rc) return rc;
  }
  return ESVO_OK;
}

// ---- device-resident stage calls: the building blocks of tick-interleaved multi-GPU operation ---------------------
// (rank r maps the ticks k with k % N == r completely; a tick needs nothing from the previous DepthMaps -- the
// DepthFrame is rebuilt from the window at every tick, esvo_Mapping.cpp:266-272 -- only the frames of the last
// ticks, which the ranks all-gather; see esvo_amd/dist.py)
extern "C" int esvo_map_front(esvo_handle h, uint64_t t_ns, const uint64_t* pose_t_ns, const double* pose_T, size_t m,
                              size_t* n_points) {
  if (!h || !pose_t_ns || !pose_T || !n_points) return ESVO_ERR_INVALID_ARG;
  if (!h->obs_set) FAIL(ESVO_ERR_STATE, "esvo_map_set_observation has not been called");
  if (h->sharded) FAIL(ESVO_ERR_STATE, "handle is sharded by slot/band: esvo_map_front maps whole ticks");
  HIPCHK(hipSetDevice(h->device));
  int rc = flush_pending_tick(h);
  if (rc) return rc;
  rc = tick_phase0(h, t_ns, pose_t_ns, pose_T, m);
  if (rc) return rc;
  const u32 n = h->tk[h->fpar].n;
  if (n) { rc = run_order_points(h, n, h->d_pts_tmp); if (rc) return rc; }
  HIPCHK(hipMemcpyAsync(h->h_counters + 16 * h->fpar, h->d_counters, sizeof(u32) * 16, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipEventRecord(h->evt[EV_CNT + h->fpar * EV_FRONT_STRIDE], h->stream));
  rc = tick_phase1_collect(h, h->fpar);
  if (rc) return rc;
  *n_points = h->tk[h->fpar].points;
  return ESVO_OK;
}

extern "C" int esvo_map_front_frame(esvo_handle h, const esvo_depth_point_t** d_frame) {
  if (!h || !d_frame) return ESVO_ERR_INVALID_ARG;
  *d_frame = h->d_pts_tmp;
  return ESVO_OK;
}

extern "C" int esvo_map_push_frame_device(esvo_handle h, const esvo_depth_point_t* d_pts, size_t n, const double* pose_T,
                                          size_t m) {
  if (!h || (n && !d_pts) || (m && !pose_T)) return ESVO_ERR_INVALID_ARG;
  if (m > h->max_poses) FAIL(ESVO_ERR_CAPACITY, "pose table larger than max_poses_per_tick");
  HIPCHK(hipSetDevice(h->device));
  int rc = flush_pending_tick(h);
  if (rc) return rc;
  u32 off;
  rc = window_reserve(h, (u32)n, &off);
  if (rc) return rc;
  // the points were produced on the front stream (or by a collective the caller issued there); the copy runs on the
  // back stream, behind any fusion that still reads ring space freed by earlier pops
  rc = back_after_front(h);
  if (rc) return rc;
  if (n) HIPCHK(hipMemcpyAsync(h->d_win + off, d_pts, sizeof(esvo_depth_point_t) * n, hipMemcpyDeviceToDevice, h->stream_b));
  static const double ident[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  return commit_frame(h, off, (u32)n, m ? pose_T : ident, (u32)m);
}

extern "C" int esvo_map_fuse_async(esvo_handle h) {
  if (!h) return ESVO_ERR_INVALID_ARG;
  if (!h->obs_set) FAIL(ESVO_ERR_STATE, "esvo_map_set_observation has not been called");
  HIPCHK(hipSetDevice(h->device));
  int rc = flush_pending_tick(h);
  if (rc) return rc;
  rc = back_after_front(h);
  if (rc) return rc;
  const int par = h->par;
  h->par ^= 1;
  HIPCHK(hipEventSynchronize(h->evt[EV_RG1 + par * EV_BACK_STRIDE]));
  collect_back(h, par);
  rc = run_fuse(h, par, h->T_world_obs);
  if (rc) return rc;
  h->committed_t_ns = h->obs_t_ns;
  h->stats.ticks++;
  h->stats.last_window_frames = (u32)h->frames.size();
  u32 np = 0;
  for (auto& f : h->frames) np += f.count;
  h->stats.last_window_points = np;
  h->stats_pending = true;
  return ESVO_OK;
}

extern "C" int esvo_shard_tick_phase(esvo_handle h, int phase, uint64_t t_ns, const uint64_t* pose_t_ns, const double* pose_T,
                                     size_t m) {
  if (!h) return ESVO_ERR_INVALID_ARG;
  if (!h->obs_set) FAIL(ESVO_ERR_STATE, "esvo_map_set_observation has not been called");
  if (!h->sharded) FAIL(ESVO_ERR_STATE, "call esvo_shard_set_band first");
  HIPCHK(hipSetDevice(h->device));
  switch (phase) {
    case 0:
      if (!pose_t_ns || !pose_T) return ESVO_ERR_INVALID_ARG;
      return tick_phase0(h, t_ns, pose_t_ns, pose_T, m);
    case 1: {
      int rc = tick_phase1_enqueue(h);
      if (rc) return rc;
      return tick_phase1_collect(h, h->fpar);
    }
    case 2: return tick_phase2(h, h->fpar);
    default: FAIL(ESVO_ERR_INVALID_ARG, "phase must be 0..2");
  }
}

extern "C" {
// ---- Outputs -----------------------------------------------------------------------------------------
int esvo_map_get_depth_points(esvo_handle h, esvo_depth_point_t* out, size_t cap, size_t* n) {
  if (!h || !n) return ESVO_ERR_INVALID_ARG;
  HIPCHK(hipSetDevice(h->device));
  { int rcp = flush_pending_tick(h); if (rcp) return rcp; }
  std::vector<esvo_depth_point_t> v;
  int rc = export_map(h, v, nullptr);
  if (rc) return rc;
  *n = v.size();
  if (out) {
    if (v.size() > cap) FAIL(ESVO_ERR_CAPACITY, "output array too small for the DepthMap");
    if (!v.empty()) std::memcpy(out, v.data(), sizeof(esvo_depth_point_t) * v.size());
  }
  return ESVO_OK;
}

int esvo_map_get_committed(esvo_handle h, esvo_depth_point_t* out, size_t cap, size_t* n, uint64_t* t_ns) {
  if (!h || !n) return ESVO_ERR_INVALID_ARG;
  HIPCHK(hipSetDevice(h->device));
  if (t_ns) *t_ns = h->committed_t_ns;
  *n = 0;
  if (h->committed_t_ns == 0) return ESVO_OK;
  std::vector<esvo_depth_point_t> v;
  int rc = export_map(h, v, nullptr);  // back stream only: a pending tick's front stage keeps running
  if (rc) return rc;
  *n = v.size();
  if (out) {
    if (v.size() > cap) FAIL(ESVO_ERR_CAPACITY, "output array too small for the DepthMap");
    std::memcpy(out, v.data(), sizeof(esvo_depth_point_t) * v.size());
  }
  return ESVO_OK;
}

int esvo_map_get_pointcloud_xyz(esvo_handle h, float* out_xyz, size_t cap_points, size_t* n) {
  if (!h || !n) return ESVO_ERR_INVALID_ARG;
  HIPCHK(hipSetDevice(h->device));
  { int rcp = flush_pending_tick(h); if (rcp) return rcp; }
  std::vector<esvo_depth_point_t> v;
  int rc = export_map(h, v, nullptr);
  if (rc) return rc;
  *n = v.size();
  if (out_xyz) {
    if (v.size() > cap_points) FAIL(ESVO_ERR_CAPACITY, "output array too small for the point cloud");
    const double* T = h->T_world_frame;  // publishPointCloud, esvo_Mapping.cpp:925-932
    for (size_t i = 0; i < v.size(); ++i)
      for (int r = 0; r < 3; ++r)
        out_xyz[3 * i + r] = (float)(((T[r * 4 + 0] * v[i].p_cam[0] + T[r * 4 + 1] * v[i].p_cam[1]) + T[r * 4 + 2] * v[i].p_cam[2]) + T[r * 4 + 3]);
  }
  return ESVO_OK;
}

int esvo_map_get_last_frame(esvo_handle h, esvo_depth_point_t* out, size_t cap, size_t* n) {
  if (!h || !n) return ESVO_ERR_INVALID_ARG;
  HIPCHK(hipSetDevice(h->device));
  { int rcp = flush_pending_tick(h); if (rcp) return rcp; }
  *n = 0;
  if (h->frames.empty()) return ESVO_OK;
  const FrameRec& f = h->frames.back();
  *n = f.count;
  if (out && f.count) {
    if (f.count > cap) FAIL(ESVO_ERR_CAPACITY, "output array too small for the frame");
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(out, h->d_win + f.off, sizeof(esvo_depth_point_t) * f.count, hipMemcpyDeviceToHost));
  }
  return ESVO_OK;
}

int esvo_get_stats(esvo_handle h, esvo_stats_t* out) {
  if (!h || !out) return ESVO_ERR_INVALID_ARG;
  HIPCHK(hipSetDevice(h->device));
  int rc = finalize_tick_stats(h);
  if (rc) return rc;
  *out = h->stats;
  return ESVO_OK;
}

// ---- Multi-GPU row-band sharding ------------------------------------------------------------------
int esvo_shard_set_band(esvo_handle h, int row_begin, int row_end, int shard, int n_shards) {
  if (!h || row_begin < 0 || row_end > h->H || row_begin >= row_end || n_shards < 1 || shard < 0 || shard >= n_shards)
    return ESVO_ERR_INVALID_ARG;
  { int rcp = flush_pending_tick(h); if (rcp) return rcp; }
  h->dp.ev_shard = shard;
  h->dp.ev_nshards = n_shards;
  h->dp.band_y0 = row_begin;
  h->dp.band_y1 = row_end;
  set_compute_band(h);
  h->sharded = !(row_begin == 0 && row_end == h->H) || n_shards > 1;
  return ESVO_OK;
}

int esvo_shard_exchange(esvo_handle h, void** d_ptr, size_t* n_bytes) {
  if (!h || !d_ptr || !n_bytes) return ESVO_ERR_INVALID_ARG;
  *d_ptr = h->xchg_ptr;
  *n_bytes = h->xchg_bytes;
  return ESVO_OK;
}

}  // extern "C"

// ---- Tracker residual / Jacobian evaluation ---------------------------------------
extern "C" {
int esvo_track_set_current(esvo_handle h, const uint8_t* ts_left, int kernel_size) {
  if (!h) return ESVO_ERR_INVALID_ARG;
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
  } else {
    if (!h->ts_valid[0]) FAIL(ESVO_ERR_STATE, "no device-resident left Time Surface: call esvo_ts_render(h, 0, ...) first");
    HIPCHK(hipStreamWaitEvent(h->stream_t, h->evt[EV_R1], 0));  // the render of camera 0 on the front stream
  }
  if (kernel_size == 5) launch_gaussian5(src, h->d_trk_blur, h->W, h->H, h->stream_t);
  else HIPCHK(hipMemcpyAsync(h->d_trk_blur, src, npx, hipMemcpyDeviceToDevice, h->stream_t));
  launch_track_images(h->d_trk_blur, h->d_trk_neg, h->d_trk_du, h->d_trk_dv, h->W, h->H, h->stream_t);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(h->stream_t));  // the left TS may be re-rendered right after this call
  h->trk_cur = true;
  return ESVO_OK;
}

int esvo_track_get_images(esvo_handle h, uint8_t* neg, int16_t* du, int16_t* dv) {
  if (!h) return ESVO_ERR_INVALID_ARG;
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
  HIPCHK(hipSetDevice(h->device));
  if (n > h->trk_cap) {
    HIPCHK(hipStreamSynchronize(h->stream_t));
    for (void* q : {(void*)h->d_trk_xyz, (void*)h->d_trk_pts, (void*)h->d_trk_out}) if (q) hipFree(q);
    h->d_trk_xyz = nullptr; h->d_trk_pts = nullptr; h->d_trk_out = nullptr;
    const size_t cap = std::max<size_t>(n, 4096);
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&h->d_trk_xyz), cap * 3 * sizeof(float)));
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&h->d_trk_pts), cap * 3 * sizeof(double)));
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&h->d_trk_out), cap * 6 * sizeof(double)));
    h->trk_cap = cap;
  }
  h->trk_n = n;
  if (n) {
    TrackRef r;
    std::memcpy(r.T, T_world_ref, sizeof(r.T));
    HIPCHK(hipMemcpyAsync(h->d_trk_xyz, xyz_world, n * 3 * sizeof(float), hipMemcpyHostToDevice, h->stream_t));
    launch_track_reference(h->d_trk_xyz, (u32)n, r, h->d_trk_pts, h->stream_t);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream_t));  // xyz_world is borrowed for the call
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
  return ESVO_OK;
}

int esvo_track_jacobian(esvo_handle h, const double R[9], const double t[3], size_t offset, size_t count, double* fjac,
                        size_t* n_out) {
  if (!h || !R || !t || !n_out) return ESVO_ERR_INVALID_ARG;
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
  return ESVO_OK;
}
}  // extern "C"

// sizeof() of every POD of the ABI (binding self-check)
extern "C" void esvo_abi_sizes(size_t out[8]) {
  out[0] = sizeof(esvo_event_t); out[1] = sizeof(esvo_calib_t); out[2] = sizeof(esvo_params_t);
  out[3] = sizeof(esvo_match_t); out[4] = sizeof(esvo_depth_point_t); out[5] = sizeof(esvo_stats_t);
  out[6] = 0; out[7] = ESVO_HIP_ABI_VERSION;
}

// ---- device self-test: div_by(a, make_recip(b)) == a / b and sqrt_moderate(x) == sqrt(x), bit for bit ----------
#include "fdiv.hpp"
namespace {
__device__ inline unsigned long long sm64(unsigned long long& s) {
  unsigned long long z = (s += 0x9e3779b97f4a7c15ull);
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}
__global__ void selftest_div_kernel(unsigned long long n_per_thread, unsigned long long seed, unsigned long long* mismatches) {
  unsigned long long s = seed + 0x1234567ull * (blockIdx.x * blockDim.x + threadIdx.x + 1);
  unsigned long long bad = 0;
  for (unsigned long long i = 0; i < n_per_thread; ++i) {
    const unsigned long long ra = sm64(s), rb = sm64(s);
    // mantissas random; exponents: mostly moderate, sometimes extreme / zero / denormal
    const int mode = (int)(sm64(s) & 15);
    int ea = (int)(ra % 600) - 300 + 1023, eb = (int)(rb % 600) - 300 + 1023;
    if (mode == 0) ea = (int)(ra % 2046) + 1;
    if (mode == 1) eb = (int)(rb % 2046) + 1;
    if (mode == 2) ea = 0;
    if (mode == 3) eb = 0;
    unsigned long long ba = ((unsigned long long)ea << 52) | (ra >> 12);
    unsigned long long bb = ((unsigned long long)eb << 52) | (rb >> 12);
    if (mode == 4) ba = 0;  // a == 0
    if (mode == 5) ba |= 1ull << 63;
    if (mode == 6) bb |= 1ull << 63;
    const double a = __longlong_as_double((long long)ba), b = __longlong_as_double((long long)bb);
    const double q_ref = a / b;
    const double q = esvo::div_by(a, esvo::make_recip(b));
    const bool same = (__double_as_longlong(q) == __double_as_longlong(q_ref)) || (q != q && q_ref != q_ref);
    bad += !same;
    // sqrt_moderate(x) == sqrt(x) for x in [2^-700, 2^700]
    const int es = (int)(sm64(s) % 1400) - 700 + 1023;
    const double xs = __longlong_as_double((long long)(((unsigned long long)es << 52) | (ra >> 12)));
    bad += __double_as_longlong(esvo::sqrt_moderate(xs)) != __double_as_longlong(sqrt(xs));
  }
  if (bad) atomicAdd(mismatches, bad);
}
}  // namespace
extern "C" int esvo_selftest_division(unsigned long long n, unsigned long long seed, unsigned long long* mismatches) {
  esvo_context* h = nullptr;
  unsigned long long* d = nullptr;
  HIPCHK(hipMalloc(reinterpret_cast<void**>(&d), sizeof(unsigned long long)));
  HIPCHK(hipMemset(d, 0, sizeof(unsigned long long)));
  const unsigned threads = 256, blocks = 1024;
  const unsigned long long per = (n + (unsigned long long)threads * blocks - 1) / ((unsigned long long)threads * blocks);
  hipLaunchKernelGGL(selftest_div_kernel, dim3(blocks), dim3(threads), 0, 0, per, seed, d);
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(mismatches, d, sizeof(unsigned long long), hipMemcpyDeviceToHost));
  hipFree(d);
  return ESVO_OK;
}
