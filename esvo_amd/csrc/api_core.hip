// api_core.hip — lifecycle of the handle, parameters, ABI self-checks (see context.hpp).
#include "context.hpp"

namespace esvo_host {
thread_local std::string g_create_error;
}

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
  if (p->ls_norm != ESVO_LSNORM_TDIST && p->ls_norm != ESVO_LSNORM_L2) { why = "LSnorm must be Tdist or l2 (DepthProblemSolver.cpp:121-131 exits on anything else)"; return ESVO_ERR_UNSUPPORTED; }
  if (p->bm_step < 1) { why = "BM_step must be >= 1"; return ESVO_ERR_INVALID_ARG; }
  // 15 x 7 (every shipped configuration) runs the register-layout kernels; any other size the general ones (kernels_lm_any.hip:
  // lane = column, three [rows][64] f64 arrays in LDS)
  if (p->patch_size_x < 1 || p->patch_size_x > 64 || p->patch_size_y < 1 || p->patch_size_y > 40) { why = "patch size must be within 1..64 x 1..40"; return ESVO_ERR_UNSUPPORTED; }
  if (p->median_blur_kernel_size < 0 || p->median_blur_kernel_size > 3) { why = "median_blur_kernel_size must be 0..3 (kernel 2k + 1)"; return ESVO_ERR_UNSUPPORTED; }
  if (p->max_event_queue_len < 0 || p->max_event_queue_len > TSQ_LMAX) { why = "max_event_queue_len must be 0 (one stamp per pixel) or 1..32"; return ESVO_ERR_UNSUPPORTED; }
  if (p->bm_max_disparity < p->bm_min_disparity || p->bm_min_disparity < 0) { why = "bad disparity range"; return ESVO_ERR_INVALID_ARG; }
  if (p->td_nu <= 2.0 || p->td_scale <= 0) { why = "Tdist_nu must be > 2 and Tdist_scale > 0"; return ESVO_ERR_INVALID_ARG; }
  if (p->num_threads < 1 || p->num_threads > 64) { why = "num_threads out of range"; return ESVO_ERR_INVALID_ARG; }
  if (p->lm_max_iteration < 1) { why = "lm_max_iteration must be >= 1"; return ESVO_ERR_INVALID_ARG; }
  if (p->reg_radius < 0 || p->reg_radius > 31) { why = "RegularizationRadius out of range [0,31] (one 64-bit mask per tap row)"; return ESVO_ERR_INVALID_ARG; }
  return ESVO_OK;
}

}  // namespace

namespace esvo_host {

void fill_dev_params(esvo_context* h) {
  const esvo_params_t& p = h->prm;
  DevParams& d = h->dp;
  d.W = h->W; d.H = h->H;
  d.wx = p.patch_size_x; d.wy = p.patch_size_y;
  d.dmin = p.bm_min_disparity; d.dmax = p.bm_max_disparity; d.step = p.bm_step; d.updown = p.bm_updown ? 1 : 0;
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
  d.ls_norm = p.ls_norm;
}

template <typename T>
hipError_t dalloc(T** p, size_t n) { return hipMalloc(reinterpret_cast<void**>(p), (n ? n : 1) * sizeof(T)); }

// compute band of the per-cell stages: the owned rows + a halo of 2 rows for the displaced-element side
// effects (+ the regulariser's radius), see DevParams::cband_y0
void set_compute_band(esvo_context* h) {
  const int halo = 2 + (h->prm.regularization ? h->prm.reg_radius : 0);
  h->dp.cband_y0 = std::max(0, h->dp.band_y0 - halo);
  h->dp.cband_y1 = std::min(h->H, h->dp.band_y1 + halo);
}

// pinned staging + the global-index ring of the routed band mode
void release_routing(esvo_context* h) {
  for (int cam = 0; cam < 2; ++cam) {
    if (h->h_route_ev[cam]) hipHostFree(h->h_route_ev[cam]);
    h->h_route_ev[cam] = nullptr;
    h->route_cap[cam] = 0;
  }
  if (h->h_route_gidx) hipHostFree(h->h_route_gidx);
  h->h_route_gidx = nullptr;
}

}  // namespace esvo_host

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

const char* esvo_last_error(esvo_handle h) { (void)h; return g_create_error.c_str(); }  // the calling thread's last error

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
    // Three stages on three streams share the chip.  The LM kernel is the bulk of the vector-ALU work and tolerates waiting
    // (two dense waves per SIMD); the matching stage and the fusion stage are short, latency-bound kernels that run one wave
    // per SIMD beside it.  They get the HIGH priority and the LM stream the LOWEST: whenever one of their waves is ready it
    // issues, the LM waves fill every other slot.  Measured in round 3: 1.40 ms per tick against 1.70 ms with the LM stream
    // high and the fusion stream low.  (Other priority assignments and a SPATIAL partition of the compute units by
    // hipExtStreamCreateWithCUMask were measured in rounds 3-4 -- all slower, 1.5-4 ms for the partition; profiles/HISTORY.md.)
    CK(hipStreamCreateWithPriority(&h->stream, hipStreamNonBlocking, prio_hi));
    CK(hipStreamCreateWithPriority(&h->stream_b, hipStreamNonBlocking, prio_hi));
    h->own_stream = true;
    CK(hipStreamCreateWithPriority(&h->stream_l, hipStreamNonBlocking, prio_lo));
    CK(hipStreamCreateWithPriority(&h->stream_l1, hipStreamNonBlocking, prio_lo));
    if (const char* ep = esvo_dev_switch("ESVO_LM_PAIR")) h->lm_pair_forced = std::atoi(ep) == 1 ? 1 : (std::atoi(ep) == 0 ? 0 : -1);
    if (const char* eq = esvo_dev_switch("ESVO_LM_QUEUES")) h->lm_queues = std::atoi(eq) == 1 ? 1 : (std::atoi(eq) == 2 ? 2 : 0);
    if (const char* em = esvo_dev_switch("ESVO_LM_QUEUES_MAX_EVENTS")) h->lm_two_max = (u32)std::strtoul(em, nullptr, 10);  // A/B only
    const char* e = esvo_dev_switch("ESVO_LM_STREAM");
    h->lm_split = !(e && std::atoi(e) == 0);
  }
  if (const char* ef = esvo_dev_switch("ESVO_FRONT_THROTTLE")) h->front_throttle = std::atoi(ef) != 0;
  if (const char* ea = esvo_dev_switch("ESVO_RESYNC")) h->resync_on = std::atoi(ea) != 0;
  if (const char* ea = esvo_dev_switch("ESVO_COLLECT_ASIDE")) h->collect_aside = std::atoi(ea) != 0;
  if (const char* et = esvo_dev_switch("ESVO_TIMELINE")) h->tl_on = std::atoi(et) != 0;
  if (const char* el = esvo_dev_switch("ESVO_LOWLAT")) h->lat_mode = std::atoi(el) != 0;
  if (const char* el = esvo_dev_switch("ESVO_LOWLAT_TIMED_EVERY")) h->lat_timed_every = std::max(1, std::atoi(el));
  if (const char* el = esvo_dev_switch("ESVO_PIPE_BIG_TIMED_EVERY")) h->pipe_big_every = std::max(1, std::atoi(el));
  if (const char* el = esvo_dev_switch("ESVO_REG_SPARSE")) h->reg_sparse_forced = std::atoi(el) != 0 ? 1 : 0;
  if (const char* el = esvo_dev_switch("ESVO_BACK_PROLOGUE")) h->pro_always = std::atoi(el) != 0;
  if (const char* el = esvo_dev_switch("ESVO_PIPE_TIMED_EVERY")) h->pipe_timed_every = std::max(1, std::atoi(el));
  if (const char* el = esvo_dev_switch("ESVO_LOWLAT_MAX_EVENTS")) h->lat_max_events = (u32)std::strtoul(el, nullptr, 10);
  if (const char* e1 = esvo_dev_switch("ESVO_ONE_STREAM")) {  // A/B only: the three stages in one queue (no cross-queue hand-offs)
    if (std::atoi(e1) == 1) {
      hipStreamDestroy(h->stream_b);
      h->stream_b = h->stream;
      h->one_stream = true;
      h->lm_split = false;
    }
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
    // per rectified row: the raw rows its bilinear taps reach (what a row band of the Time Surface needs of the SAE;
    // esvo_shard_set_routing).  A tap outside the image reads the constant 0 and needs nothing.
    h->fix_row_lo[cam].assign(h->H, h->H);
    h->fix_row_hi[cam].assign(h->H, -1);
    for (int y = 0; y < h->H; ++y)
      for (int x = 0; x < h->W; ++x) {
        const int2 m = fm[(size_t)y * h->W + x];
        const int ix = m.x >> 5, iy = m.y >> 5, fx = m.x & 31, fy = m.y & 31;
        if (ix + (fx ? 1 : 0) < 0 || ix >= h->W) continue;
        const int lo = std::max(iy, 0), hi = std::min(iy + (fy ? 1 : 0), h->H - 1);
        if (lo > hi) continue;
        h->fix_row_lo[cam][y] = std::min(h->fix_row_lo[cam][y], lo);
        h->fix_row_hi[cam][y] = std::max(h->fix_row_hi[cam][y], hi);
      }
    CK(dalloc(&h->d_sae[cam], npx));
    CK(hipMemset(h->d_sae[cam], 0, sizeof(u64) * npx));
    if (params->max_event_queue_len > 0) {  // EventQueueMat semantics: a set of <= L keys per pixel, slot-major
      CK(dalloc(&h->d_tsq[cam], npx * (size_t)params->max_event_queue_len));
      CK(hipMemset(h->d_tsq[cam], 0, sizeof(u64) * npx * (size_t)params->max_event_queue_len));
    }
    CK(dalloc(&h->d_ts[cam], npx + 64));
    CK(dalloc(&h->d_obs2[0][cam], npx + 64));
    CK(dalloc(&h->d_obs2[1][cam], npx + 64));
    h->d_obs[cam] = h->d_obs2[0][cam];
  }
  if (params->max_event_queue_len > 0) {
    h->tsq_len = params->max_event_queue_len;
    const size_t tiles = (size_t)((h->W + 7) / 8) * ((h->H + 7) / 8);
    h->tsq_tcap = 1024;
    if (const char* e = esvo_dev_switch("ESVO_TSQ_TILE_CAP")) h->tsq_tcap = (u32)std::max(1, std::atoi(e));  // tests: force the overflow list
    CK(dalloc(&h->d_tsq_tcount, tiles));
    CK(hipMemset(h->d_tsq_tcount, 0, sizeof(u32) * tiles));
    CK(dalloc(&h->d_tsq_tlist, tiles * h->tsq_tcap));
    CK(dalloc(&h->d_tsq_over, (size_t)esvo_context::TSQ_ROUND));
    CK(dalloc(&h->d_tsq_over_count, 1));
  }
  CK(dalloc(&h->d_raw, npx + 64));
  CK(dalloc(&h->d_raw1, npx + 64));
  h->h_rect_lut[0].assign(left->rect_lut, left->rect_lut + 2 * npx);
  if (right->rect_lut) h->h_rect_lut[1].assign(right->rect_lut, right->rect_lut + 2 * npx);
  CK(dalloc(&h->d_obs_tmp, npx + 64));
  h->ring_cap = (u64)std::max<int64_t>(params->event_ring_capacity, 1024);
  for (int cam = 0; cam < 2; ++cam) CK(dalloc(&h->d_ring[cam], h->ring_cap));
  h->max_poses = (u32)std::max(params->max_poses_per_tick, 2);
  CK(dalloc(&h->d_pose_T2[0], (size_t)h->max_poses * 17));  // [T | toSec]
  CK(dalloc(&h->d_pose_T2[1], (size_t)h->max_poses * 17));
  h->d_pose_T = h->d_pose_T2[0];
  // + 1: the SGM bootstrap selects up to PROCESS_EVENT_NUM + 1 events (esvo_Mapping.cpp:547: `size() <= PROCESS_EVENT_NUM_`)
  h->max_ev = (u32)std::max(params->max_events_per_tick, params->process_event_num) + 1u;
  if (h->max_ev > 4000000u) { g_create_error = "max_events_per_tick too large (scan limit 4M)"; esvo_destroy(h); return ESVO_ERR_CAPACITY; }
  if (npx > 4000000u) { g_create_error = "image too large (scan limit 4M pixels)"; esvo_destroy(h); return ESVO_ERR_CAPACITY; }
  const size_t E = h->max_ev;
  CK(dalloc(&h->d_tick_ev, E));
  CK(dalloc(&h->d_match_slots, E));
  CK(dalloc(&h->d_match_flags, E));
  CK(dalloc(&h->d_match_prefix, E));
  CK(dalloc(&h->d_matches2[0], E));
  CK(dalloc(&h->d_matches2[1], E));
  h->d_matches = h->d_matches2[0];
  {
    const char* es0 = esvo_dev_switch("ESVO_LM_SPLIT");
    h->lm_split_mode = es0 ? (std::atoi(es0) == 1 ? 1 : 0) : -1;
  }
  // (scratch of the split launch: 7 x 16 doubles per match -- only where the launch can be used)
  if (E > LM_SPLIT_MIN_EVENTS && (h->lm_split_mode == 1 || (h->lm_split_mode < 0 && E >= 400000u))) {
    // The split LM launch (kernels_lm.hip, LmSplit) executes 10 % fewer vector instructions (3.47e8 against 3.85e8 per launch
    // of the bench workload) but does not shorten the tick (1.40 against 1.38 ms): what it removes are the partially masked
    // instructions of lockstep execution, and the tick is bound by the chip's throughput at its sustained f64 clock.  It is
    // On the 1280x720 stress stream (4.9e5 events, 2.3e5 matches per tick: five times the waves) it does pay: 7.7 against
    // 8.4 ms per tick (profiles/r03_split_launch_other_workloads.txt).  So: used for launches bounded by >= 400 000 events,
    // ESVO_LM_SPLIT=0 / 1 forces it off / on.
    const char* es = esvo_dev_switch("ESVO_LM_SPLIT");
    h->lm_split_mode = es ? (std::atoi(es) == 1 ? 1 : 0) : -1;
    CK(dalloc(&h->d_lm_fvec0, E * 7 * 16));
    CK(dalloc(&h->d_lm_fnorm0, E));
    CK(dalloc(&h->d_lm_meta, E));
    CK(dalloc(&h->d_lm_order, E));
    CK(dalloc(&h->d_lm_hist, 2 * 32 * 64));  // kernels_lm.hip: 2 x LM_SPLIT_STRIPES x LM_SPLIT_BINS
    CK(hipMemset(h->d_lm_hist, 0, sizeof(u32) * 2 * 32 * 64));
  }
  // (two blocks, one per front parity: two LM launches in flight -- the two LM queues -- must not share the probe's scratch, where
  //  a wave leaves its start stamps: a launch that read the other one's stamp added a wrapped difference to the sums)
  CK(dalloc(&h->d_clk, 2 * clk_words(h->max_ev)));
  CK(hipMemset(h->d_clk, 0, sizeof(u64) * 2 * clk_words(h->max_ev)));
  if (const char* ec = esvo_dev_switch("ESVO_CLK_PROBE")) h->clk_probe = std::atoi(ec) != 0;
  if (const char* ep = esvo_dev_switch("ESVO_LM_PERSIST")) h->lm_persist = std::atoi(ep) != 0;
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) h->lm_persist_blocks = (u32)cus * 4u * 2u;
    if (const char* eb = esvo_dev_switch("ESVO_LM_PERSIST_BLOCKS")) h->lm_persist_blocks = (u32)std::max(1, std::atoi(eb));
  }
  {
    int khz = 0;  // rate of s_memrealtime (wall_clock64): the constant reference clock the probe divides by
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device) != hipSuccess || khz <= 0) khz = 100000;
    h->stats.clk_ref_khz = (uint32_t)khz;
  }
  for (int k = 0; k < 2; ++k) {
    CK(dalloc(&h->d_pt_slots2[k], E));
    CK(dalloc(&h->d_pt_flags2[k], E));
    CK(dalloc(&h->d_pt_prefix2[k], E));
    CK(dalloc(&h->d_scan_tmp_l2[k], scan_scratch_elems(std::max(E, npx)) + 8));
  }
  h->d_pt_slots = h->d_pt_slots2[0]; h->d_pt_flags = h->d_pt_flags2[0]; h->d_pt_prefix = h->d_pt_prefix2[0];
  h->d_scan_tmp_l = h->d_scan_tmp_l2[0];
  CK(dalloc(&h->d_pts_tmp, E));
  CK(dalloc(&h->d_stage[0], E));
  CK(dalloc(&h->d_stage[1], E));
  CK(dalloc(&h->d_counters2[0], CNT_ROW));
  CK(dalloc(&h->d_counters2[1], CNT_ROW));
  CK(hipMemset(h->d_counters2[0], 0, sizeof(u32) * CNT_ROW));
  CK(hipMemset(h->d_counters2[1], 0, sizeof(u32) * CNT_ROW));
  h->d_counters = h->d_counters2[0];
  CK(hipHostMalloc(reinterpret_cast<void**>(&h->h_counters), sizeof(u32) * CNT_ROW * 2));
  std::memset(h->h_counters, 0, sizeof(u32) * CNT_ROW * 2);
  CK(hipHostMalloc(reinterpret_cast<void**>(&h->h_pin), sizeof(double) * 2 * ((size_t)h->max_poses * 17 + 16)));
  CK(dalloc(&h->d_scan_tmp, scan_scratch_elems(std::max(E, npx)) + 8));
  CK(dalloc(&h->d_scan_tmp_b, scan_scratch_elems(std::max(E, npx)) + 8));
  CK(dalloc(&h->d_cnt_b, 8));
  CK(hipMemset(h->d_cnt_b, 0, sizeof(u32) * 8));
  CK(hipHostMalloc(reinterpret_cast<void**>(&h->h_cnt_b), sizeof(u32) * 8 * 4));  // rows 0, 1: tick parities; 2: exports; 3: [par] halo violations
  std::memset(h->h_cnt_b, 0, sizeof(u32) * 8 * 4);
  CK(dalloc(&h->d_halo_viol, 2));
  CK(hipMemset(h->d_halo_viol, 0, sizeof(u32) * 2));
  // fusion window
  h->win_cap = (u32)std::max<int64_t>((int64_t)params->max_window_points, (int64_t)E) + 2 * (u32)E;
  CK(dalloc(&h->d_win, h->win_cap));
  // CONST_POINTS keeps frames until their points exceed 1.5 maxNumFusionPoints (esvo_Mapping.cpp:346-353): every non-empty
  // frame holds at least one point, which bounds their number; empty frames are run-length records (context.hpp)
  h->max_frames = (u32)std::max(params->max_fusion_frames + 2, 512);
  if (params->fusion_strategy == ESVO_FUSION_CONST_POINTS)
    h->max_frames = std::max(h->max_frames, (u32)(1.5 * params->max_fusion_points) + 4u);
  // pose-table slots of the window's non-empty frames: max_frames + 1 in the worst case (CONST_POINTS with one point per
  // frame: 1 GB of tables at 20 000 points x 256 poses), a handful in practice -- allocated for 1024 frames and doubled on
  // demand (alloc_pose_slot, api_map.hip).  ESVO_POSE_SLOTS0 (tests): a smaller first allocation.
  h->slot_used.assign(h->max_frames + 1, 0);
  h->n_pose_slots = std::min<u32>(h->max_frames + 1, 1024u);
  if (const char* e0 = esvo_dev_switch("ESVO_POSE_SLOTS0")) h->n_pose_slots = std::min<u32>(h->max_frames + 1, (u32)std::max(1, std::atoi(e0)));
  CK(dalloc(&h->d_frame_pose_T, (size_t)h->n_pose_slots * h->max_poses * 16));
  CK(dalloc(&h->d_fr_table, 2 * (3 * (size_t)h->max_frames + 1)));
  CK(hipHostMalloc(reinterpret_cast<void**>(&h->h_fr_table), sizeof(u32) * 2 * (3 * (size_t)h->max_frames + 1)));
  // map
  CK(dalloc(&h->d_prop, h->win_cap));
  {  // the fusion front (kernels_fuse.hip): FUSE_TILE x FUSE_TILE-cell tiles
    const size_t n_tiles = (size_t)((h->W + FUSE_TILE - 1) / FUSE_TILE) * ((h->H + FUSE_TILE - 1) / FUSE_TILE);
    if (const char* et = esvo_dev_switch("ESVO_FUSE_TILE_CAP")) h->fuse_tile_cap = (u32)std::max(1L, std::atol(et));
    if (const char* ep = esvo_dev_switch("ESVO_FUSE_PMAX")) h->fuse_pmax_plus1 = (u32)std::max(0L, std::atol(ep)) + 1u;
    CK(dalloc(&h->d_tile_pts, n_tiles * h->fuse_tile_cap));
    CK(dalloc(&h->d_over_pts, h->win_cap));
    CK(dalloc(&h->d_tile_count, n_tiles));
    CK(dalloc(&h->d_cell_count, npx));
    CK(dalloc(&h->d_cell_offset, npx));
    h->fuse_slice_cap = (u32)((n_tiles + 63) / 64) * FUSE_TILE * FUSE_TILE;   // the cells of the tiles t with t % 64 == slice
    CK(dalloc(&h->d_cell_list, (size_t)16 * 64 * h->fuse_slice_cap));
    CK(dalloc(&h->d_fuse_ctr, 2112 + 64));
    CK(hipMemset(h->d_tile_count, 0, sizeof(u32) * n_tiles));  // zero between ticks: fuse_turn_kernel clears what was read
    CK(hipMemset(h->d_fuse_ctr, 0, sizeof(u32) * (2112 + 64)));
    if (const char* er = esvo_dev_switch("ESVO_FUSE_TILE_REC")) h->fuse_tile_rec = (u32)std::max(1L, std::atol(er));
    CK(dalloc(&h->d_rec_ids, n_tiles * h->fuse_tile_rec + (size_t)h->win_cap * 9));
  }
  if (const char* ef = esvo_dev_switch("ESVO_FUSE_LDS_CAP")) h->fuse_lds_cap = (u32)std::max(0L, std::atol(ef));
  CK(hipMalloc(reinterpret_cast<void**>(&h->d_map), map_buffer_bytes(npx)));  // cells + their dense flags (common.hpp: map_flags)
  CK(hipMalloc(reinterpret_cast<void**>(&h->d_map2), map_buffer_bytes(npx)));  // cells + their dense flags (common.hpp: map_flags)
  CK(hipMemset(h->d_map, 0, map_buffer_bytes(npx)));
  CK(hipMemset(h->d_map2, 0, map_buffer_bytes(npx)));
  h->d_map_cur = h->d_map;
  CK(dalloc(&h->d_owner_max, npx));
  CK(dalloc(&h->d_owner_min, npx));
  CK(dalloc(&h->d_own_w, E));
  CK(dalloc(&h->d_lkeep, E));
  h->codes_bytes = (E + 7) / 8 * 8;
  CK(dalloc(&h->d_codes, h->codes_bytes));
  CK(dalloc(&h->d_sel, E));
  CK(dalloc(&h->d_evmap, npx + 64));
  CK(dalloc(&h->d_reg_ab, npx));
  CK(dalloc(&h->d_reg_cd, npx));
  CK(dalloc(&h->d_exp_flags, npx));
  CK(dalloc(&h->d_exp_prefix, npx));
  CK(dalloc(&h->d_export, npx));
  CK(dalloc(&h->d_export_cell, npx));
  for (int i = 0; i < EV_N; ++i) CK(hipEventCreate(&h->evt[i]));
  if (h->tl_on) { CK(hipEventCreate(&h->tl_ref)); CK(hipEventRecord(h->tl_ref, h->stream)); CK(hipEventSynchronize(h->tl_ref)); }
  h->evt_ok = true;
  CK(hipEventCreateWithFlags(&h->evt_trk_read, hipEventDisableTiming));
  for (int cam = 0; cam < 2; ++cam) CK(hipEventCreateWithFlags(&h->evt_ingest[cam], hipEventDisableTiming));
  CK(hipHostMalloc(reinterpret_cast<void**>(&h->h_pose_pool), sizeof(double) * 16 * (size_t)h->max_poses * esvo_context::POSE_POOL));
  for (int i = 0; i < esvo_context::POSE_POOL; ++i) CK(hipEventCreate(&h->pool_evt[i]));
  h->pool_ok = true;
  for (int i = 0; i < 16; ++i) h->T_world_obs[i] = h->T_world_frame[i] = (i % 5 == 0) ? 1.0 : 0.0;
#undef CK
  *out = h;
  return ESVO_OK;
}

int esvo_destroy(esvo_handle h) {
  if (!h) return ESVO_OK;
  hipSetDevice(h->device);
  if (h->stream) hipStreamSynchronize(h->stream);
  if (h->stream_l) hipStreamSynchronize(h->stream_l);
  if (h->stream_l1) hipStreamSynchronize(h->stream_l1);
  if (h->stream_b) hipStreamSynchronize(h->stream_b);
  comm_release(h);
  void* ptrs[] = {h->d_lut, h->d_mask, h->d_fixmap[0], h->d_fixmap[1], h->d_sae[0], h->d_sae[1], h->d_raw, h->d_raw1, h->d_fwd_lut[0], h->d_fwd_lut[1], h->d_fwd_off[0], h->d_fwd_off[1],
                  h->d_fwd_src[0], h->d_fwd_src[1], h->d_fwd_val, h->d_ts[0],
                  h->d_ts[1], h->d_ring[0], h->d_ring[1], h->d_obs2[0][0], h->d_obs2[0][1], h->d_obs2[1][0], h->d_obs2[1][1], h->d_obs_tmp,
                  h->d_pose_T2[0], h->d_pose_T2[1], h->d_scan_tmp_b, h->d_cnt_b, h->d_tick_ev, h->d_match_slots, h->d_match_flags, h->d_match_prefix,
                  h->d_matches2[0], h->d_matches2[1], h->d_scan_tmp_l2[0], h->d_scan_tmp_l2[1], h->d_pt_slots2[0], h->d_pt_slots2[1], h->d_pt_flags2[0], h->d_pt_flags2[1], h->d_pt_prefix2[0], h->d_pt_prefix2[1], h->d_pts_tmp, h->d_stage[0], h->d_stage[1], h->d_counters2[0], h->d_counters2[1], h->d_scan_tmp,
                  h->d_win, h->d_frame_pose_T, h->d_fr_table, h->d_prop, h->d_tile_pts, h->d_tile_count, h->d_over_pts,
                  h->d_cell_count, h->d_cell_offset, h->d_cell_list, h->d_fuse_ctr, h->d_rec_ids, h->d_map, h->d_map2, h->d_owner_max, h->d_owner_min, h->d_exp_flags,
                  h->d_exp_prefix, h->d_export, h->d_export_cell, h->d_reg_ab, h->d_reg_cd, h->d_tsq[0], h->d_tsq[1], h->d_tsq_tcount, h->d_tsq_tlist, h->d_tsq_over, h->d_tsq_over_count, h->d_own_w, h->d_lkeep, h->d_codes, h->d_codes_send, h->d_codes_all, h->d_pts_send, h->d_pts_all, h->d_rank_kept,
                  h->d_sel, h->d_evmap, h->d_lm_fvec0, h->d_lm_fnorm0, h->d_lm_meta, h->d_lm_order, h->d_lm_hist, h->d_clk, h->d_ring_gidx, h->d_halo_viol, h->d_dn_flags, h->d_merge_a, h->d_merge_b, h->d_merge_plan, h->d_tsq_dup};
  for (void* p : ptrs) if (p) hipFree(p);
  if (h->h_counters) hipHostFree(h->h_counters);
  if (h->h_cnt_b) hipHostFree(h->h_cnt_b);
  if (h->h_pin) hipHostFree(h->h_pin);
  if (h->h_fr_table) hipHostFree(h->h_fr_table);
  if (h->evt_ok) for (int i = 0; i < EV_N; ++i) hipEventDestroy(h->evt[i]);
  if (h->pool_ok) for (int i = 0; i < esvo_context::POSE_POOL; ++i) hipEventDestroy(h->pool_evt[i]);
  if (h->h_pose_pool) hipHostFree(h->h_pose_pool);
  if (h->h_trk_ne) hipHostFree(h->h_trk_ne);
  if (h->h_trk_xyz) hipHostFree(h->h_trk_xyz);
  release_routing(h);
  for (int cam = 0; cam < 2; ++cam) if (h->d_wire[cam]) hipFree(h->d_wire[cam]);
  if (h->evt_trk_read) hipEventDestroy(h->evt_trk_read);
  if (h->tl_ref) hipEventDestroy(h->tl_ref);
  for (int cam = 0; cam < 2; ++cam) if (h->evt_ingest[cam]) hipEventDestroy(h->evt_ingest[cam]);
  for (void* q : {(void*)h->d_viz_bgr, (void*)h->d_viz_jet, (void*)h->d_viz_owner}) if (q) hipFree(q);
  for (void* q : {(void*)h->sgm.sobL, (void*)h->sgm.rawL, (void*)h->sgm.sobR, (void*)h->sgm.rawR, (void*)h->sgm.vol[0], (void*)h->sgm.vol[1],
                  (void*)h->sgm.vol[2], (void*)h->sgm.vol[3], (void*)h->sgm.vol[4], (void*)h->sgm.vol[5], (void*)h->sgm.d1, (void*)h->sgm.d1b,
                  (void*)h->sgm.d2key, (void*)h->d_sgm_img[0], (void*)h->d_sgm_img[1], (void*)h->d_sgm_disp, (void*)h->d_sgm_pair, (void*)h->d_sgm_T})
    if (q) hipFree(q);
  if (h->own_stream && h->stream) hipStreamDestroy(h->stream);
  if (h->stream_b && !h->one_stream) hipStreamDestroy(h->stream_b);
  if (h->stream_l) { hipStreamSynchronize(h->stream_l); hipStreamDestroy(h->stream_l); }
  if (h->stream_l1) { hipStreamSynchronize(h->stream_l1); hipStreamDestroy(h->stream_l1); }
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
  API_LOCK(h);
  // a reset excludes the other two groups as well: pushers in flight finish first, the tracker's images go
  std::lock_guard<std::mutex> lp0(h->mu_push[0]), lp1(h->mu_push[1]), ltk(h->mu_track), lts(h->mu_ts), lr(h->mu_ring);
  HIPCHK(hipSetDevice(h->device));
  { int rcp = flush_pending_tick(h); if (rcp) return rcp; }
  const size_t npx = (size_t)h->W * h->H;
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipStreamSynchronize(h->stream_l)); HIPCHK(hipStreamSynchronize(h->stream_l1));
  HIPCHK(hipStreamSynchronize(h->stream_b));
  HIPCHK(hipStreamSynchronize(h->stream_t));
  HIPCHK(hipStreamSynchronize(h->stream_i));
  comm_reset(h);  // (drains the exchange stream of a handle with a communicator)
  for (int cam = 0; cam < 2; ++cam) {
    HIPCHK(hipMemsetAsync(h->d_sae[cam], 0, sizeof(u64) * npx, h->stream));
    if (h->d_tsq[cam]) HIPCHK(hipMemsetAsync(h->d_tsq[cam], 0, sizeof(u64) * npx * (size_t)h->tsq_len, h->stream));
    h->ts_host[cam].clear();
    h->ring_base[cam] = h->ring_next[cam] = h->ring_reserved[cam] = h->scattered[cam] = 0;
    h->scatter_pending_lo[cam] = ~0ull;
    h->ingest_pending[cam] = false;
    h->ts_valid[cam] = false;
    h->last_stamp[cam] = 0;
    h->tsq_dup[cam].clear();
  }
  h->glob_ts.clear();
  h->kept_g.clear();
  h->own_before.clear();
  h->own_total = 0;
  h->glob_base = 0;
  h->halo_error = false;
  h->dn_pending = false;
  h->resync = esvo_context::Resync();
  HIPCHK(hipMemsetAsync(h->d_halo_viol, 0, sizeof(u32) * 2, h->stream));
  std::memset(h->h_cnt_b + 8 * 3, 0, sizeof(u32) * 8);
  h->sh_first = 0;
  h->sh_first_prev = 0;
  h->trk_read_pending = false;
  h->ema_lm_ms = h->ema_back_ms = 0.f;
  h->lm_two_on = false;
  std::memset(h->lm_pair_ms, 0, sizeof(h->lm_pair_ms));
  h->lm_pair_n[0] = h->lm_pair_n[1] = 0u;
  h->lm_pair_decisions = 0;
  h->lm_pair_current = -1;
  h->frames.clear();
  h->n_window_frames = 0;
  std::fill(h->slot_used.begin(), h->slot_used.end(), 0);
  HIPCHK(hipMemsetAsync(h->d_map, 0, map_buffer_bytes(npx), h->stream));
  HIPCHK(hipMemsetAsync(h->d_map2, 0, map_buffer_bytes(npx), h->stream));
  HIPCHK(hipMemsetAsync(h->d_tile_count, 0, sizeof(u32) * (size_t)((h->W + FUSE_TILE - 1) / FUSE_TILE) * ((h->H + FUSE_TILE - 1) / FUSE_TILE), h->stream));
  HIPCHK(hipMemsetAsync(h->d_fuse_ctr, 0, sizeof(u32) * 2112, h->stream));
  if (h->d_rank_kept) HIPCHK(hipMemsetAsync(h->d_rank_kept, 0, sizeof(u32) * esvo_context::SHARD_MAX_RANKS, h->stream));
  h->d_map_cur = h->d_map;
  h->obs_set = false;
  h->n_pose = 0;
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipStreamSynchronize(h->stream_l)); HIPCHK(hipStreamSynchronize(h->stream_l1));
  HIPCHK(hipStreamSynchronize(h->stream_b));
  h->back_pending[0] = h->back_pending[1] = false;
  h->committed_t_ns = 0;
  h->ts_timing_pending[0] = h->ts_timing_pending[1] = false;
  h->ts_pair_sample = false;
  h->stats_pending = false;
  {
    const uint32_t khz = h->stats.clk_ref_khz;
    std::memset(&h->stats, 0, sizeof(h->stats));
    h->stats.clk_ref_khz = khz;
  }
  HIPCHK(hipMemset(h->d_clk, 0, sizeof(u64) * CLK_SCRATCH));
  HIPCHK(hipMemset(h->d_clk + clk_words(h->max_ev), 0, sizeof(u64) * CLK_SCRATCH));
  return ESVO_OK;
}

int esvo_set_params(esvo_handle h, const esvo_params_t* params) {
  if (!h || !params) return ESVO_ERR_INVALID_ARG;
  API_LOCK(h);
  { int rcp = flush_pending_tick(h); if (rcp) return rcp; }
  std::string why;
  int rc = validate_params(params, why);
  if (rc) FAIL(rc, why);
  if ((u32)std::max(params->max_events_per_tick, params->process_event_num) > h->max_ev)
    FAIL(ESVO_ERR_CAPACITY, "process_event_num exceeds the capacity fixed at esvo_create");
  {
    const u32 need = params->fusion_strategy == ESVO_FUSION_CONST_POINTS ? (u32)(1.5 * params->max_fusion_points) + 4u
                                                                         : (u32)params->max_fusion_frames + 2u;
    if (need > h->max_frames) FAIL(ESVO_ERR_CAPACITY, "fusion window (frames) exceeds the capacity fixed at esvo_create");
  }
  if (h->routed && (params->smooth_time_surface != h->prm.smooth_time_surface || params->median_blur_kernel_size != h->prm.median_blur_kernel_size ||
                    params->patch_size_y != h->prm.patch_size_y || params->denoising != h->prm.denoising || params->bm_updown))
    FAIL(ESVO_ERR_STATE, "the handle routes events by row (esvo_shard_set_routing): SmoothTimeSurface, median_blur_kernel_size, patch_size_Y and Denoising "
                         "fix the rows it renders and the events it keeps; esvo_reset + esvo_shard_set_routing to change them");
  esvo_params_t np = *params;
  np.max_events_per_tick = h->prm.max_events_per_tick;
  np.max_window_points = h->prm.max_window_points;
  np.max_poses_per_tick = h->prm.max_poses_per_tick;
  np.event_ring_capacity = h->prm.event_ring_capacity;
  if (np.max_event_queue_len != h->prm.max_event_queue_len) FAIL(ESVO_ERR_CAPACITY, "max_event_queue_len is fixed at esvo_create (the per-pixel queues are allocated there)");
  h->prm = np;
  fill_dev_params(h);
  set_compute_band(h);
  return ESVO_OK;
}

int esvo_set_stream(esvo_handle h, void* hip_stream) {
  if (!h) return ESVO_ERR_INVALID_ARG;
  API_LOCK(h);
  std::lock_guard<std::mutex> lp0(h->mu_push[0]), lp1(h->mu_push[1]);  // a pusher may be draining the front stream
  { int rcp = flush_pending_tick(h); if (rcp) return rcp; }
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipStreamSynchronize(h->stream_l)); HIPCHK(hipStreamSynchronize(h->stream_l1));
  HIPCHK(hipStreamSynchronize(h->stream_b));
  if (h->own_stream && h->stream) hipStreamDestroy(h->stream);
  h->stream = reinterpret_cast<hipStream_t>(hip_stream);
  if (h->one_stream) h->stream_b = h->stream;
  h->own_stream = false;
  return ESVO_OK;
}

int esvo_synchronize(esvo_handle h) {
  if (!h) return ESVO_ERR_INVALID_ARG;
  API_LOCK(h);
  { int rcp = flush_pending_tick(h); if (rcp) return rcp; }
  const bool poll = h->lat_last;  // latency mode (context.hpp): the caller waits for a tick that ran alone -- polled, not slept
  HIPCHK(esvo_wait_stream(h->stream, poll));
  HIPCHK(esvo_wait_stream(h->stream_l, poll)); HIPCHK(esvo_wait_stream(h->stream_l1, poll));
  HIPCHK(esvo_wait_stream(h->stream_b, poll));
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
namespace {
__global__ void debug_stall_kernel(unsigned long long ref_ticks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ref_ticks) __builtin_amdgcn_s_sleep(32);
}
}  // namespace
// tools only (tools/regime_probe.py, ESVO_TIMELINE=1): when every stage of the last `max_rows` ticks ran, in ms since esvo_create:
// rows of 12 -- front stage T0, BM0, BM1, S1, LM0, LM1, S2, CNT; back stage FU0, FU1, CL1, RG1 -- collected from the handle's HIP
// events as the ticks completed (no extra synchronisation while the run lasts; this call drains the handle).  Not part of the
// documented ABI.
extern "C" int esvo_debug_timeline(esvo_handle h, float* out, int max_rows, int* n_rows) {
  if (!h || !out || !n_rows) return ESVO_ERR_INVALID_ARG;
  API_LOCK(h);
  HIPCHK(hipSetDevice(h->device));
  int rc = finalize_tick_stats(h);
  if (rc) return rc;
  const size_t n = std::min(h->tl_front.size(), h->tl_back.size());
  const size_t take = std::min<size_t>(n, (size_t)std::max(max_rows, 0));
  for (size_t k = 0; k < take; ++k) {
    const size_t i = n - take + k;
    for (int j = 0; j < 8; ++j) out[k * 12 + j] = h->tl_front[i][j];
    for (int j = 0; j < 4; ++j) out[k * 12 + 8 + j] = h->tl_back[i][j];
  }
  *n_rows = (int)take;
  return ESVO_OK;
}

// tools only (tools/regime_probe.py): occupy one of the handle's queues for `microseconds` with a kernel that does nothing --
// a stage of the tick pipeline falls behind by that much.  which: 0 front (Time Surfaces, block matching), 1 LM, 2 back (fusion,
// regulariser).  Not part of the documented ABI.
extern "C" int esvo_debug_stall(esvo_handle h, int which, unsigned microseconds) {
  if (!h || which < 0 || which > 2) return ESVO_ERR_INVALID_ARG;
  API_LOCK(h);
  HIPCHK(hipSetDevice(h->device));
  const unsigned long long ticks = (unsigned long long)microseconds * (h->stats.clk_ref_khz ? h->stats.clk_ref_khz : 100000u) / 1000ull;
  hipStream_t s = which == 0 ? h->stream : (which == 1 ? h->stream_l : h->stream_b);
  hipLaunchKernelGGL(debug_stall_kernel, dim3(1), dim3(1), 0, s, ticks);
  HIPCHK(hipGetLastError());
  return ESVO_OK;
}

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
