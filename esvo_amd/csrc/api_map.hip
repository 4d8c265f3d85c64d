// api_map.hip — the mapper: stage-wise calls, the fused (lazily completed, two-stream) tick, device-resident stage
// calls, multi-GPU sharding, outputs and statistics (see context.hpp).
#include <chrono>

#include "context.hpp"

namespace esvo_host {

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

int upload_poses(esvo_context* h, const uint64_t* pose_t_ns, const double* pose_T, size_t m, u32* d_zero_row = nullptr) {
  if (m > h->max_poses) FAIL(ESVO_ERR_CAPACITY, "pose table larger than max_poses_per_tick");
  // staged through pinned memory (two alternating slots): no host synchronisation on the tick path
  h->pin_slot ^= 1;
  double* pin = h->h_pin + (size_t)h->pin_slot * ((size_t)h->max_poses * 17 + 16);
  double* T = pin;            // [T (16 m) | toSec (m)]: one contiguous upload
  double* sec = pin + 16 * m;
  for (size_t i = 0; i < m; ++i) sec[i] = ns_to_sec(pose_t_ns[i]);
  std::memcpy(T, pose_T, sizeof(double) * 16 * m);
  h->h_pose_T.assign(pose_T, pose_T + 16 * m);
  h->n_pose = (u32)m;
  // the back stage copies the previous table of this buffer into its frame slot: not before that is done
  h->pose_buf ^= 1;
  h->d_pose_T = h->d_pose_T2[h->pose_buf];
  HIPCHK(hipStreamWaitEvent(h->stream, h->evt[EV_POSE + h->pose_buf * EV_BACK_STRIDE], 0));
  h->d_pose_sec = h->d_pose_T + 16 * m;
  // d_zero_row: the counter row of the tick that follows, cleared by the same launch
  launch_upload_words(T, h->d_pose_T, sizeof(double) * 17 * m, h->stream, d_zero_row, CNT_ROW);
  return ESVO_OK;
}

// BM over n events starting at absolute ring index `first` (reverse walk) or over d_tick_ev:
// flags + match records in slot (thread-stride) order
int run_bm(esvo_context* h, const esvo_event_t* d_ev, u64 first, u64 cap, int reverse, u32 n, const u32* sel) {
  BmArgs a;
  a.ev = d_ev; a.n = n; a.ev_first = first; a.ev_cap = cap; a.ev_reverse = reverse; a.sel = sel;
  a.tsL = h->d_obs[0]; a.tsR = h->d_obs[1];
  a.lut = h->d_lut; a.mask = h->d_mask;
  a.pose_sec = h->d_pose_sec; a.n_pose = h->n_pose;
  a.out_slots = h->d_match_slots; a.out_flags = h->d_match_flags;
  a.fail_counters = h->d_counters;
  if (h->stage_events_on) hipEventRecord(h->evt[EV_BM0 + h->fpar * EV_FRONT_STRIDE], h->stream);
  launch_bm_match(a, h->dp, h->stream);
  if (h->stage_events_on) hipEventRecord(h->evt[EV_BM1 + h->fpar * EV_FRONT_STRIDE], h->stream);
  HIPCHK(hipGetLastError());
  return ESVO_OK;
}
// stable compaction of the match slots into vEMP order.  Sharded mode: the flags are this rank's own
// ones, the list is its dense local list (count -> counters[8]) and slot_of remembers each entry's slot.
int run_order_matches(esvo_context* h, u32 n, bool local) {
  if (scan_compact_is_small(n)) {  // a small tick: one launch (scan.hip)
    // (latency mode: the list as indices into the slots -- d_own_w -- which the wide LM layout reads; no record is copied)
    const bool by_index = h->match_by_index && !local;
    launch_scan_compact_matches_small(h->d_match_flags, h->d_match_prefix, h->d_counters + (local ? 8 : 0), n, h->d_match_slots,
                                      by_index ? nullptr : h->d_matches, (local || by_index) ? h->d_own_w : nullptr, h->stream);
  } else {
    launch_exclusive_scan_u32(h->d_match_flags, h->d_match_prefix, h->d_counters + (local ? 8 : 0), h->d_scan_tmp, n, h->stream);
    launch_compact_matches(h->d_match_slots, h->d_match_flags, h->d_match_prefix, n, h->d_matches, local ? h->d_own_w : nullptr,
                           h->stream);
  }
  if (h->stage_events_on) hipEventRecord(h->evt[EV_S1 + h->fpar * EV_FRONT_STRIDE], h->stream);
  HIPCHK(hipGetLastError());
  return ESVO_OK;
}
int run_match(esvo_context* h, const esvo_event_t* d_ev, u64 first, u64 cap, int reverse, u32 n) {
  int rc = run_bm(h, d_ev, first, cap, reverse, n);
  if (rc) return rc;
  return run_order_matches(h, n, false);
}

// LM (+cull) over the compacted matches: point records + flags in solver-slot order (dense: in list order)
// the LM stage's own buffers follow the front parity (context.hpp: two LM queues)
static void set_lm_parity(esvo_context* h) {
  h->d_pt_slots = h->d_pt_slots2[h->fpar]; h->d_pt_flags = h->d_pt_flags2[h->fpar]; h->d_pt_prefix = h->d_pt_prefix2[h->fpar];
  h->d_scan_tmp_l = h->d_scan_tmp_l2[h->fpar];
}
// which LM layout the next lazy tick uses (context.hpp: lm_pair_*): -1 not a candidate, else 0 wide / 1 pair
static int lm_pair_policy(esvo_context* h, u32 n_events) {
  if (n_events == 0 || n_events > esvo::LM_PAIR_MAX_EVENTS || h->prm.ls_norm == ESVO_LSNORM_L2) return -1;
  if (h->lm_pair_forced >= 0) return h->lm_pair_forced;
  const u32 k = h->lm_pair_decisions++;
  if (k < 8u || h->lm_pair_n[0] == 0u || h->lm_pair_n[1] == 0u) return (int)(k & 1u);
  auto recent_min = [&](int mode) {
    float m = 1e30f;
    for (u32 i = 0; i < 4u && i < h->lm_pair_n[mode]; ++i) m = std::min(m, h->lm_pair_ms[mode][i]);
    return m;
  };
  // The samples are EV_LM0..EV_LM1 intervals on the lowest-priority stream: contention with the other stages only ever ADDS
  // time, so the minimum of the recent four is the estimate least touched by it; and the layout in use is left only for one
  // that is 5 % faster by that estimate (hysteresis: two layouts within noise of each other do not alternate run to run).
  if (h->lm_pair_current < 0) h->lm_pair_current = recent_min(1) < recent_min(0) ? 1 : 0;
  else if (recent_min(h->lm_pair_current ^ 1) < 0.95f * recent_min(h->lm_pair_current)) h->lm_pair_current ^= 1;
  return (k % 64u == 63u) ? h->lm_pair_current ^ 1 : h->lm_pair_current;  // (a periodic sample of the other one keeps its estimate fresh)
}
int run_lm(esvo_context* h, u32 max_matches, int cull, bool dense, hipStream_t st = nullptr, int pair = -1) {
  if (!st) st = h->stream;
  if (h->gather_guard[h->fpar]) {  // a back stage's first launch reads this parity's solver slots (latency mode, tick_phase2): not
    h->gather_guard[h->fpar] = false;  // before it is done (an event long complete when ticks are waited for one by one)
    HIPCHK(hipStreamWaitEvent(st, h->evt[EV_STG + h->fpar * EV_FRONT_STRIDE], 0));
  }
  u32* flags = dense ? h->d_lkeep : h->d_pt_flags;  // the kernel writes every flag of its launch range
  LmArgs a;
  a.matches = h->d_matches; a.n_matches = h->d_counters + (dense ? 8 : 0); a.max_matches = max_matches;
  a.match_index = nullptr;
  if (h->match_by_index && !dense) { a.matches = h->d_match_slots; a.match_index = h->d_own_w; }
  a.tsL = h->d_obs[0]; a.tsR = h->d_obs[1];
  a.pose_T = h->d_pose_T; std::memcpy(a.T_world_obs, h->T_world_obs, sizeof(double) * 16);
  a.out_slots = h->d_pt_slots; a.out_flags = flags; a.cull = cull; a.dense = dense ? 1 : 0;
  const bool split = h->d_lm_fvec0 != nullptr && (h->lm_split_mode == 1 || (h->lm_split_mode < 0 && max_matches >= 400000u));
  a.split_fvec0 = split ? h->d_lm_fvec0 : nullptr; a.split_fnorm0 = h->d_lm_fnorm0; a.split_meta = h->d_lm_meta;
  a.split_order = h->d_lm_order; a.split_hist = h->d_lm_hist;
  a.pair = pair >= 0 ? pair : (h->lm_pair_forced == 1 && max_matches <= esvo::LM_PAIR_MAX_EVENTS ? 1 : 0);
  a.clk = h->clk_probe ? h->d_clk + (size_t)h->fpar * clk_words(h->max_ev) : nullptr;  // (a block per front parity: api_core.hip)
  if (h->routed && dense) { a.halo_viol = h->d_counters + 10; a.vy0 = h->oband_y0; a.vy1 = h->oband_y1; }
  // launches of the throughput layout: persistent groups that pull matches from a counter (kernels_lm.hip); counters[11] is
  // zero at this point (a tick clears its counter row with the pose upload, run_refine clears it itself)
  if (h->lm_persist && !dense) { a.persist_next = h->d_counters + 11; a.persist_blocks = h->lm_persist_blocks; }
  const bool timed_lm = h->stage_events_on || h->tk[h->fpar].timed_lm;
  if (timed_lm) hipEventRecord(h->evt[EV_LM0 + h->fpar * EV_FRONT_STRIDE], st);
  launch_lm_refine(a, h->dp, h->d_counters + 2, st);
  // (EV_LM1 is also what the point compaction waits for when it runs on the other LM queue -- collect_aside)
  if (timed_lm || h->tk[h->fpar].cnt_stream != st) hipEventRecord(h->evt[EV_LM1 + h->fpar * EV_FRONT_STRIDE], st);
  HIPCHK(hipGetLastError());
  return ESVO_OK;
}
// stable compaction of the solver slots: the culled points go to `dst` in the reference's order
int run_order_points(esvo_context* h, u32 max_matches, DevPoint* dst, hipStream_t st) {
  u32* scratch = (st && st != h->stream) ? h->d_scan_tmp_l : h->d_scan_tmp;  // the LM stage scans beside the next tick's BM
  if (!st) st = h->stream;
  if (scan_compact_is_small(max_matches)) {
    // (the refinement kernel writes a flag for every slot of its launch, 0 beyond the match count: no count to clip to)
    launch_scan_compact_points_small(h->d_pt_flags, h->d_pt_prefix, h->d_counters + 1, max_matches, h->d_pt_slots, dst, st,
                                     h->d_counters, h->cnt_row_host, h->cnt_row_host ? (u32)CNT_ROW : 0u);
    h->cnt_row_sent = h->cnt_row_host != nullptr;
  } else {
    launch_exclusive_scan_u32(h->d_pt_flags, h->d_pt_prefix, h->d_counters + 1, scratch, max_matches, st);
    launch_compact_points(h->d_pt_slots, h->d_pt_flags, h->d_pt_prefix, h->d_counters + 0, max_matches, dst, st);
  }
  if (h->stage_events_on) hipEventRecord(h->evt[EV_S2 + h->fpar * EV_FRONT_STRIDE], st);
  HIPCHK(hipGetLastError());
  return ESVO_OK;
}
int run_refine(esvo_context* h, u32 max_matches, int cull, DevPoint* dst) {
  HIPCHK(hipMemsetAsync(h->d_counters + 2, 0, sizeof(u32), h->stream));  // n_solved (a tick zeroes all counters at once)
  HIPCHK(hipMemsetAsync(h->d_counters + 11, 0, sizeof(u32), h->stream));  // the persistent LM layout's work counter
  int rc = run_lm(h, max_matches, cull, false);
  if (rc) return rc;
  return run_order_points(h, max_matches, dst);
}

// EventBM's per-reason failure counters (EventBM.h:89) from a counter row read back from the device
void collect_bm_failures(esvo_context* h, const u32* row, bool accumulate) {
  u32 r[3] = {0, 0, 0};
  for (int k = 0; k < 3; ++k)
    for (int i = 0; i < CNT_STRIPES; ++i) r[k] += row[CNT_BM_FAIL + k * CNT_STRIPES + i];
  esvo_stats_t& s = h->stats;
  s.last_bm_info_noise_low = r[0]; s.last_bm_coarse_fail = r[1]; s.last_bm_fine_fail = r[2];
  if (accumulate) { s.total_bm_info_noise_low += r[0]; s.total_bm_coarse_fail += r[1]; s.total_bm_fine_fail += r[2]; }
}
int read_counters(esvo_context* h) {
  HIPCHK(hipMemcpyAsync(h->h_counters, h->d_counters, sizeof(u32) * CNT_ROW, hipMemcpyDeviceToHost, h->stream));
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
  if (h->routed && h->h_cnt_b[8 * 3 + par]) {  // (the running total over all ranks: identical on every rank at this point of the call sequence)
    s.halo_violations = h->h_cnt_b[8 * 3 + par];
    h->halo_error = true;
  }
  if (h->prm.regularization) s.last_map_size = h->h_cnt_b[8 * par + 7];  // alive cells of the band (exports refresh it)
  if (!h->back_timed[par]) return;  // latency mode: this back stage's timings were not sampled (context.hpp, lat_ticks)
  float fu = 0, cl = 0, rg = 0;
  hipEventElapsedTime(&fu, h->evt[EV_FU0 + o], h->evt[EV_FU1 + o]);
  hipEventElapsedTime(&cl, h->evt[EV_FU1 + o], h->evt[EV_CL1 + o]);
  hipEventElapsedTime(&rg, h->evt[EV_CL1 + o], h->evt[EV_RG1 + o]);
  s.ms_fusion = fu + cl;
  s.ms_regularization = rg;
  s.ms_kernel[4] = fu; s.ms_kernel[5] = cl; s.ms_kernel[6] = rg;
  s.sum_ms_kernel[4] += fu; s.sum_ms_kernel[5] += cl; s.sum_ms_kernel[6] += rg;
  h->ema_back_ms = h->ema_back_ms > 0.f ? 0.75f * h->ema_back_ms + 0.25f * (fu + cl + rg) : fu + cl + rg;
  if (h->tl_on && h->tl_ref) {
    const int bk[4] = {EV_FU0, EV_FU1, EV_CL1, EV_RG1};
    std::array<float, 4> row;
    for (int i = 0; i < 4; ++i) { row[i] = -1.f; if (hipEventElapsedTime(&row[i], h->tl_ref, h->evt[bk[i] + o]) != hipSuccess) (void)hipGetLastError(); }
    h->tl_back.push_back(row);
  }
}

// place a frame of n points in the window ring (frames stay contiguous: [oldest frame, newest frame) modulo the wrap)
// (`frames`: the window to place it behind -- the handle's own, or a copy on which a caller has already dropped the frames that
// will leave, to learn whether a frame fits BEFORE it changes anything)
static int window_reserve_in(esvo_context* h, const std::deque<FrameRec>& frames, u32 n, u32* off_out) {
  u32 off = 0;
  const FrameRec* first = nullptr;  // oldest and newest frames that occupy ring space (empty frames hold none)
  const FrameRec* last = nullptr;
  for (const FrameRec& f : frames)
    if (f.count) { if (!first) first = &f; last = &f; }
  if (first) {
    const FrameRec& back = *last;
    const FrameRec& front = *first;
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
int window_reserve(esvo_context* h, u32 n, u32* off_out) { return window_reserve_in(h, h->frames, n, off_out); }
// would a frame of n points fit once the window has been cut down to fewer than `keep_below` frames (the pops themselves are
// left to the caller, after its last fallible step)?
static int window_probe_after_pops(esvo_context* h, size_t keep_below, u32 n) {
  std::deque<FrameRec> fr = h->frames;
  size_t nwf = h->n_window_frames;
  while (nwf && nwf >= keep_below) {
    nwf--;
    if (fr.front().run > 1) fr.front().run--; else fr.pop_front();
  }
  u32 off;
  return window_reserve_in(h, fr, n, &off);
}
int alloc_pose_slot(esvo_context* h, u32* slot) {
  for (u32 i = 0; i < h->n_pose_slots; ++i)
    if (!h->slot_used[i]) { h->slot_used[i] = 1; *slot = i; return ESVO_OK; }
  // every allocated slot holds a frame of the window: double the table (a rare, synchronising step; kernels take the pointer
  // at launch, so nothing in flight may still read the old one)
  const u32 cap = h->max_frames + 1;
  if (h->n_pose_slots >= cap) FAIL(ESVO_ERR_CAPACITY, "no free pose-table slot (too many frames in the fusion window)");
  const u32 n_new = (u32)std::min<u64>(cap, 2ull * h->n_pose_slots);
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipStreamSynchronize(h->stream_l)); HIPCHK(hipStreamSynchronize(h->stream_l1));
  HIPCHK(hipStreamSynchronize(h->stream_b));
  double* d_new = nullptr;
  const size_t per = (size_t)h->max_poses * 16 * sizeof(double);
  if (hipMalloc(reinterpret_cast<void**>(&d_new), per * n_new) != hipSuccess) {
    (void)hipGetLastError();
    FAIL(ESVO_ERR_CAPACITY, "out of device memory growing the pose-table slots");
  }
  if (hipMemcpy(d_new, h->d_frame_pose_T, per * h->n_pose_slots, hipMemcpyDeviceToDevice) != hipSuccess) {
    (void)hipGetLastError();
    (void)hipFree(d_new);  // the old table stays in place and in use
    FAIL(ESVO_ERR_HIP, "copying the pose-table slots into the grown table failed");
  }
  double* d_old = h->d_frame_pose_T;
  h->d_frame_pose_T = d_new;  // the copy succeeded: from here on the handle owns the new table whatever the free says
  HIPCHK(hipFree(d_old));
  *slot = h->n_pose_slots;
  h->slot_used[*slot] = 1;
  h->n_pose_slots = n_new;
  return ESVO_OK;
}
void pop_front_frame(esvo_context* h) {
  FrameRec& f = h->frames.front();
  h->n_window_frames--;
  if (f.run > 1) { f.run--; return; }
  if (f.slot != NO_SLOT) h->slot_used[f.slot] = 0;
  h->frames.pop_front();
}
// window policy, esvo_Mapping.cpp:341-368
void apply_window_policy(esvo_context* h) {
  if (h->prm.fusion_strategy == ESVO_FUSION_CONST_POINTS) {
    auto total = [&]() { size_t s = 0; for (auto& f : h->frames) s += f.count; return s; };
    size_t np = total();
    while ((double)np > 1.5 * (double)h->prm.max_fusion_points) { pop_front_frame(h); np = total(); }
  } else {
    while (h->n_window_frames > (size_t)h->prm.max_fusion_frames) pop_front_frame(h);
  }
}

// latency mode (context.hpp, DeferredCopies): the copies a tick's back stage opens with, if run_fuse did not get to carry them in
// its first launch (an error on the way), are enqueued the plain way -- the events behind them release buffers the next ticks wait for
static int flush_deferred_copies(esvo_context* h) {
  esvo_context::DeferredCopies d = h->pro;
  h->pro = esvo_context::DeferredCopies();
  if (!d.active) return ESVO_OK;
  if (d.a_flags) launch_back_prologue(nullptr, nullptr, 0, d.a_src, d.a_dst, d.a_bytes, nullptr, nullptr, 0, h->stream_b, d.a_flags, d.a_prefix, d.a_slots);
  else if (d.a_bytes) HIPCHK(hipMemcpyAsync(d.a_dst, d.a_src, d.a_bytes, hipMemcpyDeviceToDevice, h->stream_b));
  if (d.ev_a >= 0) HIPCHK(hipEventRecord(h->evt[d.ev_a], h->stream_b));
  if (d.b_bytes) HIPCHK(hipMemcpyAsync(d.b_dst, d.b_src, d.b_bytes, hipMemcpyDeviceToDevice, h->stream_b));
  if (d.ev_b >= 0) HIPCHK(hipEventRecord(h->evt[d.ev_b], h->stream_b));
  return ESVO_OK;
}
// pose table of the frame: from the host (stage-wise API) or, in a tick, the front stage's device table
int commit_frame(esvo_context* h, u32 off, u32 count, const double* pose_T_host, u32 m, int pose_buf, bool apply_policy) {
  if (count == 0) {  // an empty frame: no pose table, no ring space; consecutive ones share a record
    if (!h->frames.empty() && h->frames.back().count == 0) h->frames.back().run++;
    else h->frames.push_back(FrameRec{off, 0, NO_SLOT, 1});
    h->n_window_frames++;
    if (apply_policy) apply_window_policy(h);
    return ESVO_OK;
  }
  u32 slot;
  int rc = alloc_pose_slot(h, &slot);  // before the frame enters the deque: a failure leaves the window as it was
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
    } else if (h->pro.active) {  // latency mode: carried by run_fuse's first launch
      h->pro.b_src = h->d_pose_T2[pose_buf]; h->pro.b_dst = dst; h->pro.b_bytes = sizeof(double) * 16 * m;
      h->pro.ev_b = EV_POSE + pose_buf * EV_BACK_STRIDE;
    } else {
      HIPCHK(hipMemcpyAsync(dst, h->d_pose_T2[pose_buf], sizeof(double) * 16 * m, hipMemcpyDeviceToDevice, h->stream_b));
      HIPCHK(hipEventRecord(h->evt[EV_POSE + pose_buf * EV_BACK_STRIDE], h->stream_b));
    }
  }
  h->frames.push_back(FrameRec{off, count, slot, 1});
  h->n_window_frames++;
  if (apply_policy) apply_window_policy(h);
  return ESVO_OK;
}

// fusion loop + clean + regularisation on the current window, on the back stream; `par` selects the
// pinned frame table and the event set (two ticks may be in flight)
int run_fuse(esvo_context* h, int par, const double* T_world_obs, bool naive) {
  // frames newest -> oldest (esvo_Mapping.cpp:372-377)
  // The table is laid out COMPACTLY for the frames in use -- [cum (nf + 1) | off (nf) | slot (nf)] -- so that one small
  // upload carries it (max_frames is sized for the worst case of CONST_POINTS, one point per frame; a tick uses a handful).
  const size_t tab = 3 * (size_t)h->max_frames + 1;
  u32* host = h->h_fr_table + (size_t)par * tab;
  u32 nf = 0;
  for (size_t q = h->frames.size(); q-- > 0;)
    if (h->frames[q].count) ++nf;  // empty frames contribute no point (DepthFusion::update loops over none)
  if (nf > h->max_frames) { (void)flush_deferred_copies(h); FAIL(ESVO_ERR_CAPACITY, "too many non-empty frames in the fusion window"); }
  u32* cum = host;
  u32* off = host + (nf + 1);
  u32* slot = off + nf;
  u32 total = 0, i = 0;
  for (size_t q = h->frames.size(); q-- > 0;) {
    const FrameRec& f = h->frames[q];
    if (f.count == 0) continue;
    cum[i] = total; off[i] = f.off; slot[i] = f.slot;
    total += f.count;
    ++i;
  }
  cum[nf] = total;
  hipStream_t sb = h->stream_b;
  int tail_ev[2] = {-1, -1};
  u32* dtab = h->d_fr_table + (size_t)par * tab;
  if (h->pro.active) {  // latency mode: the frame's points and its pose table travel with the table (one launch, not three operations)
    const esvo_context::DeferredCopies d = h->pro;
    h->pro = esvo_context::DeferredCopies();
    launch_back_prologue(host, dtab, sizeof(u32) * (3 * (size_t)nf + 1), d.a_src, d.a_dst, d.a_bytes, d.b_src, d.b_dst, d.b_bytes, sb,
                         d.a_flags, d.a_prefix, d.a_slots);
    // "staging buffer / pose table free again": recorded at the END of this back stage, not here between two dependent launches
    // (~5 us each); who waits for them -- the front stage two ticks on -- comes long after either point
    tail_ev[0] = d.ev_a;
    if (d.tail_b) tail_ev[1] = d.ev_b;
    else if (d.ev_b >= 0) HIPCHK(hipEventRecord(h->evt[d.ev_b], sb));
  } else {
    launch_upload_words(host, dtab, sizeof(u32) * (3 * (size_t)nf + 1), sb);
  }
  std::memcpy(h->T_world_frame, T_world_obs, sizeof(double) * 16);  // new DepthFrame at the TS pose (:268-272)
  FuseArgs a;
  a.win = h->d_win;
  a.fr_cum = dtab; a.fr_off = dtab + (nf + 1); a.fr_slot = a.fr_off + nf;
  a.n_frames = nf; a.n_pts = total;
  a.frame_pose_T = h->d_frame_pose_T; a.max_poses = h->max_poses;
  rigid_inverse(h->T_world_frame, a.T_frame_world);
  a.prop = h->d_prop;
  a.tile_count = h->d_tile_count; a.tile_pts = h->d_tile_pts; a.tile_cap = h->fuse_tile_cap;
  a.over_pts = h->d_over_pts; a.over_count = h->d_fuse_ctr + 2081;
  a.rec_ids = h->d_rec_ids; a.tile_rec = h->fuse_tile_rec; a.rec_cursor = h->d_fuse_ctr + 2080;
  a.cell_count = h->d_cell_count; a.cell_offset = h->d_cell_offset; a.cell_list = h->d_cell_list; a.slice_cap = h->fuse_slice_cap;
  a.class_count = h->d_fuse_ctr; a.class_total = h->d_fuse_ctr + 1024;
  a.lds_cap = h->fuse_lds_cap; a.pmax_plus1 = h->fuse_pmax_plus1; a.d_total = h->d_cnt_b + 4;
  a.map = h->d_map; a.d_num_fusion = h->d_cnt_b + 3;
  a.n_touched = h->d_cnt_b + 6;
  a.naive = naive ? 1 : 0;
  a.owner_max = h->prm.regularization ? h->d_owner_max : nullptr;
  a.owner_min = h->d_owner_min; a.n_reg_elems = h->prm.regularization ? h->d_cnt_b + 7 : nullptr;
  if (total > h->win_cap) {
    for (int e : tail_ev) if (e >= 0) hipEventRecord(h->evt[e], sb);
    FAIL(ESVO_ERR_CAPACITY, "window points exceed capacity");
  }
  const int o = par * EV_BACK_STRIDE;
  const bool timed = h->stage_events_on;
  h->back_timed[par] = timed;
  if (timed) hipEventRecord(h->evt[EV_FU0 + o], sb);
  launch_fuse(a, h->dp, sb);
  if (timed) hipEventRecord(h->evt[EV_FU1 + o], sb);
  h->d_map_cur = h->d_map;
  // (naive propagation, esvo_MVStereo.cpp:416-428: the map is published as it is, neither cleaned nor regularised)
  const bool do_clean = naive ? false : (h->prm.clean_requires_full_window ? (h->n_window_frames >= (size_t)h->prm.max_fusion_frames) : true);
  if (do_clean) launch_clean(h->d_map, h->dp, sb);
  if (timed) hipEventRecord(h->evt[EV_CL1 + o], sb);
  if (h->prm.regularization && !naive) {
    launch_reg_view(h->d_map, h->d_map2, h->d_owner_max, h->d_owner_min, h->d_reg_ab, h->d_reg_cd, h->d_cnt_b + 7, h->dp, sb);
    // (the tile kernel's layout for sparse maps when the newest known element count -- the previous tick's -- is below a tenth of
    //  the band's cells: scheduling only, same bits; ESVO_REG_SPARSE = 0 / 1 forces never / always)
    const u64 band_cells = (u64)std::max(h->dp.band_y1 - h->dp.band_y0, 1) * (u64)h->W;
    const bool sparse = h->reg_sparse_forced >= 0 ? h->reg_sparse_forced == 1 : (u64)h->stats.last_map_size * 10u < band_cells;
    launch_reg_apply(h->d_map, h->d_map2, h->d_owner_max, h->d_owner_min, h->d_reg_ab, h->d_reg_cd, h->d_cnt_b + 7, h->dp, sb, sparse);
    h->d_map_cur = h->d_map2;
  }
  HIPCHK(hipMemcpyAsync(h->h_cnt_b + 8 * par, h->d_cnt_b, sizeof(u32) * 8, hipMemcpyDeviceToHost, sb));
  if (h->routed) HIPCHK(hipMemcpyAsync(h->h_cnt_b + 8 * 3 + par, h->d_halo_viol, sizeof(u32), hipMemcpyDeviceToHost, sb));
  hipEventRecord(h->evt[EV_RG1 + o], sb);  // also "back stage of this parity done"
  for (int e : tail_ev) if (e >= 0) hipEventRecord(h->evt[e], sb);
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

}  // namespace esvo_host

// =================================================================================================
namespace esvo_host {
// a new observation goes into the OTHER pair of buffers: an LM stage still in flight keeps reading its own
// (the one before that has finished: the call that enqueued it completed its predecessor, context.hpp)
void begin_observation(esvo_context* h) {
  h->obs_par ^= 1;
  h->d_obs[0] = h->d_obs2[h->obs_par][0];
  h->d_obs[1] = h->d_obs2[h->obs_par][1];
  // (observations set twice between two ticks: the pending tick's LM stage reads this very pair -- write behind it)
  if (h->tick_pending && h->tk[h->fpar].obs_par == h->obs_par)
    hipStreamWaitEvent(h->stream, h->evt[EV_CNT + h->fpar * EV_FRONT_STRIDE], 0);
}
// nothing was rendered into the pair begin_observation switched to: the previous observation stays current
void revert_observation(esvo_context* h) {
  h->obs_par ^= 1;
  h->d_obs[0] = h->d_obs2[h->obs_par][0];
  h->d_obs[1] = h->d_obs2[h->obs_par][1];
}
}  // namespace esvo_host

extern "C" {

// ---- Mapper: stage-wise ---------------------------------------------------------------------------
int esvo_map_set_observation(esvo_handle h, uint64_t t_ns, const uint8_t* ts_left, const uint8_t* ts_right,
                             const double T_world_cam[16]) {
  if (!h || !T_world_cam) return ESVO_ERR_INVALID_ARG;
  API_LOCK(h);
  HIPCHK(hipSetDevice(h->device));
  const size_t npx = (size_t)h->W * h->H;
  const uint8_t* src[2] = {ts_left, ts_right};
  begin_observation(h);
  for (int cam = 0; cam < 2; ++cam) {
    uint8_t* dst = h->prm.smooth_time_surface ? h->d_obs_tmp : h->d_obs[cam];
    if (src[cam]) {
      HIPCHK(hipMemcpyAsync(dst, src[cam], npx, hipMemcpyHostToDevice, h->stream));
      HIPCHK(hipStreamSynchronize(h->stream));
    } else {
      if (!h->ts_valid[cam]) FAIL(ESVO_ERR_STATE, "no device-resident Time Surface: call esvo_ts_render first");
      if (h->prm.smooth_time_surface) dst = h->d_ts[cam];  // the blur reads the resident surface directly
      else HIPCHK(hipMemcpyAsync(dst, h->d_ts[cam], npx, hipMemcpyDeviceToDevice, h->stream));
    }
    // createMatchProblem applies GaussianBlurTS(5) when SmoothTimeSurface (EventBM.cpp:68-72)
    if (h->prm.smooth_time_surface)
      launch_gaussian5(dst, h->d_obs[cam], h->W, h->H, h->stream, h->routed ? h->oband_y0 : 0, h->routed ? h->oband_y1 : -1);
  }
  std::memcpy(h->T_world_obs, T_world_cam, sizeof(double) * 16);  // handed to the LM kernel by value
  h->obs_t_ns = t_ns;
  h->obs_set = true;
  return ESVO_OK;
}

int esvo_map_set_poses(esvo_handle h, const uint64_t* pose_t_ns, const double* pose_T, size_t m) {
  if (!h || (m && (!pose_t_ns || !pose_T))) return ESVO_ERR_INVALID_ARG;
  API_LOCK(h);
  HIPCHK(hipSetDevice(h->device));
  return upload_poses(h, pose_t_ns, pose_T, m);
}

int esvo_map_match(esvo_handle h, const esvo_event_t* ev, size_t n, const uint64_t* pose_t_ns, const double* pose_T,
                   size_t m, esvo_match_t* out, size_t cap, size_t* n_out) {
  if (!h || (n && !ev) || !n_out) return ESVO_ERR_INVALID_ARG;
  API_LOCK(h);
  if (!h->obs_set) FAIL(ESVO_ERR_STATE, "esvo_map_set_observation has not been called");
  if (n > h->max_ev) FAIL(ESVO_ERR_CAPACITY, "more events than max_events_per_tick");
  HIPCHK(hipSetDevice(h->device));
  { int rcp = flush_pending_tick(h); if (rcp) return rcp; }
  if (pose_t_ns) { int rc = upload_poses(h, pose_t_ns, pose_T, m); if (rc) return rc; }
  *n_out = 0;
  if (n == 0) { HIPCHK(hipMemsetAsync(h->d_counters, 0, sizeof(u32), h->stream)); return ESVO_OK; }
  HIPCHK(hipMemcpyAsync(h->d_tick_ev, ev, sizeof(esvo_event_t) * n, hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemsetAsync(h->d_counters + CNT_BM_FAIL, 0, sizeof(u32) * 3 * CNT_STRIPES, h->stream));
  int rc = run_match(h, h->d_tick_ev, 0, (u64)h->max_ev, 0, (u32)n);
  if (rc) return rc;
  rc = read_counters(h);
  if (rc) return rc;
  const u32 nm = h->h_counters[0];
  *n_out = nm;
  h->stats.last_events_in = (u32)n;
  h->stats.last_matches = nm;
  collect_bm_failures(h, h->h_counters, false);
  if (out && nm) {
    if (nm > cap) FAIL(ESVO_ERR_CAPACITY, "output array too small for the matches");
    HIPCHK(hipMemcpy(out, h->d_matches, sizeof(esvo_match_t) * nm, hipMemcpyDeviceToHost));
  }
  return ESVO_OK;
}

int esvo_map_refine(esvo_handle h, const esvo_match_t* matches, size_t n, int cull, esvo_depth_point_t* out, size_t cap,
                    size_t* n_out) {
  if (!h || (n && !matches) || !n_out) return ESVO_ERR_INVALID_ARG;
  API_LOCK(h);
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
  API_LOCK(h);
  if (m > h->max_poses) FAIL(ESVO_ERR_CAPACITY, "pose table larger than max_poses_per_tick");
  for (size_t i = 0; i < n; ++i)
    if (pts[i].pose_idx >= m) FAIL(ESVO_ERR_INVALID_ARG, "depth point refers to a pose outside the frame's pose table");
  HIPCHK(hipSetDevice(h->device));
  { int rcp = flush_pending_tick(h); if (rcp) return rcp; }
  u32 off;
  int rc = window_reserve(h, (u32)n, &off);
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(h->stream_l)); HIPCHK(hipStreamSynchronize(h->stream_l1));
  HIPCHK(hipStreamSynchronize(h->stream_b));  // the ring space may have been read by a fusion still in flight
  if (n) HIPCHK(hipMemcpyAsync(h->d_win + off, pts, sizeof(esvo_depth_point_t) * n, hipMemcpyHostToDevice, h->stream));
  static const double ident[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  rc = commit_frame(h, off, (u32)n, m ? pose_T : ident, (u32)m);
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipStreamSynchronize(h->stream_l)); HIPCHK(hipStreamSynchronize(h->stream_l1));
  HIPCHK(hipStreamSynchronize(h->stream_b));
  return ESVO_OK;
}

int esvo_map_fuse(esvo_handle h, size_t* n_fusions) {
  if (!h) return ESVO_ERR_INVALID_ARG;
  API_LOCK(h);
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
  HIPCHK(hipStreamSynchronize(h->stream_l)); HIPCHK(hipStreamSynchronize(h->stream_l1));
  HIPCHK(hipStreamSynchronize(h->stream_b));
  collect_back(h, par);
  h->stats.last_window_frames = (u32)h->n_window_frames;
  u32 np = 0;
  for (auto& f : h->frames) np += f.count;
  h->stats.last_window_points = np;
  if (n_fusions) *n_fusions = h->stats.last_fusions;
  return ESVO_OK;
}

}  // extern "C"

// ---- Mapper: fused tick ---------------------------------------------------------------------------
namespace esvo_host {
// event selection, esvo_Mapping.cpp:562-574 (Appendix A-3): walk back from lower_bound(t_end) to
// lower_bound(t_begin), newest first, at most PROCESS_EVENT_NUM
int select_events(esvo_context* h, uint64_t t_ns, u64* first_out, u32* n_out) {
  std::lock_guard<std::mutex> lr(h->mu_ring);  // the ingest thread appends to ts_host / advances the ring meanwhile
  ingest_fence(h, 0);  // block matching reads the left camera's ring on the front stream
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
  // (against ring_reserved: a pusher on another thread may be overwriting the slots of its block right now)
  if (n && first - (n - 1) < h->ring_reserved[0] - std::min<u64>(h->ring_reserved[0], h->ring_cap))
    FAIL(ESVO_ERR_STATE, "selected events were already overwritten in the event ring");
  *first_out = first;
  *n_out = n;
  h->sh_first_prev = h->sh_first;  // the previous tick's front stage may still be in flight on the front stream (lazy ticks)
  h->sh_first = first;             // under mu_ring: what the ingest thread's overwrite guard reads
  return ESVO_OK;
}
// The same selection on a routed band handle: the walk is defined on the WHOLE left stream (glob_ts: every stamp, kept on the
// host), the rank's ring holds the events of its rows.  n / g_first: size of the global selection and the global index of its
// newest event; loc_first / n_loc: the newest of them in this rank's ring (absolute local index) and how many the ring holds.
int select_events_routed(esvo_context* h, uint64_t t_ns, u32* n_out, u32* g_first_out, u64* loc_first_out, u32* n_loc_out, u32* n_own_out) {
  std::lock_guard<std::mutex> lr(h->mu_ring);
  ingest_fence(h, 0);
  const double t_end = ns_to_sec(t_ns);
  const u64 t_begin_ns = ros_time_from_sec(std::max(0.0, t_end - 10 * h->prm.bm_half_slice_thickness));
  const double t_begin = ns_to_sec(t_begin_ns);
  auto lower = [&](double t) {
    const auto& v = h->glob_ts;
    size_t lo = 0, hi = v.size();
    while (lo < hi) {
      const size_t mid = (lo + hi) / 2;
      if (ns_to_sec(v[mid]) < t) lo = mid + 1; else hi = mid;
    }
    return h->glob_base + lo;
  };
  const u64 it_end = lower(t_end), it_begin = lower(t_begin);
  const u64 staged_end = h->glob_base + h->glob_ts.size();
  u64 avail = it_end - it_begin;
  u64 first = it_end;
  if (it_end == staged_end && avail > 0) { first = it_end - 1; avail -= 1; }  // end() is skipped (oracle definition)
  const u32 n = (u32)std::min<u64>(avail, (u64)h->prm.process_event_num);
  if (n > h->max_ev) FAIL(ESVO_ERR_CAPACITY, "more events than max_events_per_tick");
  *n_out = n;
  *g_first_out = (u32)first;
  *loc_first_out = 0;
  *n_loc_out = 0;
  *n_own_out = 0;
  if (n == 0) return ESVO_OK;
  // the kept events with a global index in [first - n + 1, first]
  const auto& kg = h->kept_g;
  const size_t lo = std::lower_bound(kg.begin(), kg.end(), first - (n - 1)) - kg.begin();
  const size_t hi = std::upper_bound(kg.begin(), kg.end(), first) - kg.begin();
  if (hi <= lo) return ESVO_OK;
  const u64 loc_first = h->ring_base[0] + hi - 1;
  const u32 n_loc = (u32)(hi - lo);
  if (loc_first - (n_loc - 1) < h->ring_reserved[0] - std::min<u64>(h->ring_reserved[0], h->ring_cap))
    FAIL(ESVO_ERR_STATE, "selected events were already overwritten in the event ring");
  *loc_first_out = loc_first;
  *n_loc_out = n_loc;
  *n_own_out = (u32)((hi < h->own_before.size() ? h->own_before[hi] : h->own_total) - h->own_before[lo]);
  h->sh_first_prev = h->sh_first;
  h->sh_first = loc_first;
  return ESVO_OK;
}

// block length of exchange 1 (kernels_shard.hip): the bytes of a rank's own slots, whole 64-bit words
static inline size_t shard_codes_block(u32 n, u32 N) { return (((size_t)n + N - 1) / N + 7) / 8 * 8; }
// the same in routed band mode: two bits per slot of the whole tick, whole 64-bit words
static inline size_t shard_codes_block_routed(u32 n) { return (((size_t)n + 15) / 16 * 4 + 7) / 8 * 8; }

// Routed band mode, phase 0 proper: the events of the band's rows (the rank's own ring): BM over them, dense local list of the own
// matches, LM + cull on it, then the (matched, kept) bits of the own slots in a block that spans the whole tick.
// keep_flags / keep_prefix (Denoising): per walk position of the RAW selection (n_raw events) whether the event is kept and how
// many kept ones precede it -- the slots are those of the kept sequence (n of them), as on one GPU.
static int routed_front(esvo_context* h, esvo_context::TickState& tk, u32 n, const u32* keep_flags, const u32* keep_prefix, u32 n_raw = 0) {
  const u32 N = (u32)h->dp.ev_nshards;
  const u32 n_loc = tk.n_loc, n_own = tk.n_own;
  int rc;
  if (n_loc) {
    BmArgs a;
    a.ev = h->d_ring[0]; a.n = n; a.ev_first = h->sh_first; a.ev_cap = h->ring_cap; a.ev_reverse = 1; a.sel = nullptr;
    a.gidx = h->d_ring_gidx; a.g_first = tk.g_first; a.n_loc = n_loc;
    a.keep_flags = keep_flags; a.keep_prefix = keep_prefix; a.n_raw = keep_flags ? n_raw : n;
    a.tsL = h->d_obs[0]; a.tsR = h->d_obs[1];
    a.lut = h->d_lut; a.mask = h->d_mask;
    a.pose_sec = h->d_pose_sec; a.n_pose = h->n_pose;
    a.out_slots = h->d_match_slots; a.out_flags = h->d_match_flags;
    a.fail_counters = h->d_counters;
    if (h->stage_events_on) hipEventRecord(h->evt[EV_BM0 + h->fpar * EV_FRONT_STRIDE], h->stream);
    launch_bm_match(a, h->dp, h->stream);
    if (h->stage_events_on) hipEventRecord(h->evt[EV_BM1 + h->fpar * EV_FRONT_STRIDE], h->stream);
    HIPCHK(hipGetLastError());
    // dense list of the own matches (count -> counters[8]); the slot of each follows from its walk position (shard_codes_routed)
    if (scan_compact_is_small(n_loc)) {
      launch_scan_compact_matches_small(h->d_match_flags, h->d_match_prefix, h->d_counters + 8, n_loc, h->d_match_slots, h->d_matches, nullptr,
                                        h->stream);
    } else {
      launch_exclusive_scan_u32(h->d_match_flags, h->d_match_prefix, h->d_counters + 8, h->d_scan_tmp, n_loc, h->stream);
      launch_compact_matches(h->d_match_slots, h->d_match_flags, h->d_match_prefix, n_loc, h->d_matches, nullptr, h->stream);
    }
    if (h->stage_events_on) hipEventRecord(h->evt[EV_S1 + h->fpar * EV_FRONT_STRIDE], h->stream);
    HIPCHK(hipGetLastError());
    // (the ring also holds the raster's halo events: the launch -- and with it the kernel's layout -- is bounded by the OWN
    //  events of the selection, counted at ingest)
    rc = run_lm(h, n_own, 1, true);
    if (rc) return rc;
  } else {  // no event of this tick in the band: the stage events the statistics read are still recorded
    if (h->stage_events_on)
      for (int e : {EV_BM0, EV_BM1, EV_S1, EV_LM0, EV_LM1}) hipEventRecord(h->evt[e + h->fpar * EV_FRONT_STRIDE], h->stream);
  }
  const size_t nb = shard_codes_block_routed(n);
  HIPCHK(hipMemsetAsync(h->d_codes_send, 0, nb, h->stream));
  launch_shard_codes_routed(h->d_matches, h->d_lkeep, h->d_counters + 8, n_own, n, (u32)h->dp.num_threads, h->d_own_w,
                            reinterpret_cast<u32*>(h->d_codes_send), h->stream);
  HIPCHK(hipGetLastError());
  h->xchg_send = h->d_codes_send;
  h->xchg_recv = N > 1 ? h->d_codes_all : h->d_codes_send;
  h->xchg_block = nb;
  return ESVO_OK;
}
// Denoising on a routed band handle (esvo_Mapping.cpp:1046-1072: the mask is the 3 x 3 median of the selected events' map; an
// event is kept when its pixel is set in it).  An event's flag needs the selected events of its raw row and the two next to it; a
// rank's ring holds the raw rows of its band + 1 (keep_px bit 2, esvo_shard_set_routing), so it computes the flags of the events
// whose RAW row lies in its band -- every selected event has exactly one such rank -- and the ranks all-gather them as one bit per
// walk position of the selection.  The kept sequence (which events, in which order, how many) is then the one-GPU one on every rank.
static inline size_t denoise_bits_block(u32 n) { return (((size_t)n + 31) / 32 * 4 + 7) / 8 * 8; }
static int routed_denoise_begin(esvo_context* h, esvo_context::TickState& tk) {
  const u32 N = (u32)h->dp.ev_nshards;
  const size_t nb = denoise_bits_block(tk.n);
  HIPCHK(hipMemsetAsync(h->d_codes_send, 0, nb, h->stream));
  launch_denoise_bits_routed(h->d_ring[0], h->sh_first, h->ring_cap, tk.n_loc, h->d_ring_gidx, tk.g_first, tk.n, h->d_evmap, h->W, h->H,
                             h->dp.band_y0, h->dp.band_y1, reinterpret_cast<u32*>(h->d_codes_send), h->stream);
  HIPCHK(hipGetLastError());
  h->xchg_send = h->d_codes_send;
  h->xchg_recv = N > 1 ? h->d_codes_all : h->d_codes_send;
  h->xchg_block = nb;
  h->dn_pending = true;
  return ESVO_OK;
}
static int routed_denoise_resume(esvo_context* h) {
  h->dn_pending = false;
  esvo_context::TickState& tk = h->tk[h->fpar];
  const u32 N = (u32)h->dp.ev_nshards, n_raw = tk.n;
  if (!h->d_dn_flags) {
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&h->d_dn_flags), sizeof(u32) * 2 * (size_t)h->max_ev));
  }
  u32* flags = h->d_dn_flags;
  u32* prefix = h->d_dn_flags + h->max_ev;
  launch_denoise_bits_unpack(reinterpret_cast<const u32*>(N > 1 ? h->d_codes_all : h->d_codes_send), (u32)(denoise_bits_block(n_raw) / 4), N, n_raw,
                             flags, h->stream);
  launch_exclusive_scan_u32(flags, prefix, h->d_counters + 5, h->d_scan_tmp, n_raw, h->stream);
  int rc = read_counters(h);  // the kept count sizes everything behind it (as on one GPU: one read-back)
  if (rc) return rc;
  const u32 n = tk.n = h->h_counters[5];
  h->xchg_send = h->xchg_recv = nullptr;
  h->xchg_block = 0;
  if (!n) return ESVO_OK;
  return routed_front(h, tk, n, flags, prefix, n_raw);
}

// phase 0 (front stage): poses, event selection, block matching + LM of the events of this handle's shard
int tick_phase0(esvo_context* h, uint64_t t_ns, const uint64_t* pose_t_ns, const double* pose_T, size_t m) {
  // Everything that can refuse the tick (pose table too large, events beyond the capacity or already overwritten in
  // the ring) is checked BEFORE any per-tick state is switched: a refused tick must leave no trace, in particular not in
  // the pose-table double buffer, which the LM stage of a still pending tick reads and whose content the back stage
  // copies into that tick's frame slot.
  if (h->dn_pending) {  // phase 0 called again behind the exchange of the denoising bits (ESVO_AGAIN)
    h->stage_events_on = h->tk[h->fpar].timed;
    return routed_denoise_resume(h);
  }
  if (m > h->max_poses) FAIL(ESVO_ERR_CAPACITY, "pose table larger than max_poses_per_tick");
  if (h->routed && h->halo_error)
    FAIL(ESVO_ERR_HALO, "a refinement of an earlier tick read outside the Time-Surface rows some rank renders (stats.halo_violations): "
                        "raise ts_halo_rows or use ESVO_ROUTE_BROADCAST");
  u32 n = 0, n_loc = 0, n_own = 0, g_first = 0;
  u64 first = 0;
  int rc = h->routed ? select_events_routed(h, t_ns, &n, &g_first, &first, &n_loc, &n_own) : select_events(h, t_ns, &first, &n);
  if (rc) return rc;
  // (the counter row of this tick's parity -- last used two ticks ago, collected since -- is cleared with the pose upload)
  rc = upload_poses(h, pose_t_ns, pose_T, m, h->d_counters2[h->fpar ^ 1]);
  if (rc) return rc;
  h->fpar ^= 1;
  h->d_matches = h->d_matches2[h->fpar];    // the tick's own match list and counters (the LM stage of the previous tick
  h->d_counters = h->d_counters2[h->fpar];  // may still be running on its own)
  set_lm_parity(h);
  esvo_context::TickState& tk = h->tk[h->fpar];
  tk.n = n; tk.off = 0; tk.points = 0; tk.t_ns = t_ns;
  tk.n_loc = n_loc; tk.n_own = n_own; tk.g_first = g_first;
  tk.lm_stream = tk.cnt_stream = h->stream;
  tk.lm_pair = -1;
  // latency mode: nothing of an earlier tick is pending, so this tick has nothing to run beside -- its LM launch stays in the front
  // queue (one cross-queue hand-off less on the path the caller waits for), the host polls for its counters and its end, and its
  // stage timings are sampled, not recorded tick by tick (context.hpp)
  tk.lat = h->lat_now && !h->sharded && n && n <= h->lat_max_events;
  tk.gather = false;
  h->match_by_index = false;
  // (sampled for every tick that runs alone, whatever its size; a band-sharded tick is waited for phase by phase)
  tk.timed = (h->lat_now || h->sharded) ? esvo_stage_timed(h) : true;
  if (h->pipe_now && !h->sharded && !h->comm && !h->tl_on && n && n <= h->lat_max_events) {
    // a small tick that overlaps the previous one (esvo_map_tick's lazy path): host-paced -- one tick in pipe_timed_every is timed
    tk.timed = h->pipe_seq % h->pipe_timed_every == 0u;
    h->pipe_seq++;
  } else if (h->pipe_now && !h->sharded && !h->comm && !h->tl_on && n > h->lat_max_events) {
    tk.timed = h->pipe_big_seq % h->pipe_big_every == 0u;
    h->pipe_big_seq++;
  }
  tk.timed_lm = tk.timed;
  h->stage_events_on = tk.timed;  // (esvo_map_tick's scope switches it back on)
  tk.obs_par = h->obs_par;
  tk.pose_buf = h->pose_buf; tk.n_pose = h->n_pose;
  std::memcpy(tk.T_world_obs, h->T_world_obs, sizeof(double) * 16);
  // two ticks in flight at most: what this tick's front stage overwrites (ring space of popped frames, the pose
  // table buffer) was last read by the back stage two ticks ago.
  // (For the ordinary tick this device-side wait is only a throttle: the tick writes parity buffers that are released by events of
  // their own -- EV_STG, EV_POSE, the collected EV_CNT -- and the host never runs more than two back stages ahead.  Off since
  // round 5 (ESVO_FRONT_THROTTLE=1 restores it): it ties every front stage to the END of a back stage, which is what makes a
  // lagging back chain stay behind -- see pipeline_resync.  Sharded ticks keep it: their frame goes straight into the ring.)
  if (h->sharded || h->front_throttle) HIPCHK(hipStreamWaitEvent(h->stream, h->evt[EV_RG1 + h->par * EV_BACK_STRIDE], 0));
  if (tk.timed) hipEventRecord(h->evt[EV_T0 + h->fpar * EV_FRONT_STRIDE], h->stream);
  const u32* sel = nullptr;
  if (h->prm.denoising && n && !h->routed) {
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
  h->xchg_send = h->xchg_recv = nullptr;
  h->xchg_block = 0;
  if (n && !h->sharded) {
    rc = run_bm(h, h->d_ring[0], h->sh_first, h->ring_cap, 1, n, sel);
    if (rc) return rc;
    // latency mode: the compacted match list is not materialised -- the scan leaves the kept slots' indices and the (wide) LM
    // layout reads the block matcher's slots through them (one workgroup copying 48-byte records: 17 us for DSEC's 10 000 slots)
    h->match_by_index = tk.lat && scan_compact_is_small(n) && lm_launch_is_wide(n, h->dp);
    rc = run_order_matches(h, n, false);
    if (rc) { h->match_by_index = false; return rc; }
    hipStream_t sl = h->stream;
    if (h->split_now && !tk.lat) {  // the LM stage on its own stream, behind this tick's matches
      HIPCHK(hipEventRecord(h->evt[EV_A1 + h->fpar * EV_FRONT_STRIDE], h->stream));
      // launches in the latency-bound (wide) layout alternate between the two LM queues; the split launch's scratch and
      // the throughput layout (which fills the chip by itself) stay on one
      const bool split_scratch = h->d_lm_fvec0 != nullptr && (h->lm_split_mode == 1 || (h->lm_split_mode < 0 && n >= 400000u));
      if (h->ema_lm_ms > 0.f && h->ema_back_ms > 0.f)
        h->lm_two_on = h->ema_lm_ms > (h->lm_two_on ? 0.7f : 0.9f) * h->ema_back_ms;  // (context.hpp: round 6's thresholds)
      const bool two = (h->lm_queues == 2 || (h->lm_queues == 0 && h->lm_two_on)) && n <= h->lm_two_max && !split_scratch;
      h->lm_two_now = two;
      sl = (two && h->fpar) ? h->stream_l1 : h->stream_l;
      HIPCHK(hipStreamWaitEvent(sl, h->evt[EV_A1 + h->fpar * EV_FRONT_STRIDE], 0));
    }
    if (h->resync.lm_wait_back) {  // pipeline_resync: this LM launch starts together with the back stage after the newest enqueued one
      h->resync.lm_wait_back = false;
      HIPCHK(hipStreamWaitEvent(sl, h->evt[EV_RG1 + (h->par ^ 1) * EV_BACK_STRIDE], 0));
    }
    tk.lm_stream = tk.cnt_stream = sl;
    // With ONE LM queue, what follows the launch on the device (point compaction, counters) goes to the idle second queue: the
    // next tick's launch -- enqueued before this one has finished -- then follows this one directly instead of waiting out
    // ~40 us of small dependent launches at the head of the stream that paces the pipeline.
    // (only while ticks overlap -- the previous one is still pending: a tick that is waited for gains nothing from it and would pay
    //  one more cross-queue hand-off)
    if (h->collect_aside && h->split_now && !tk.lat && h->tick_pending && (sl == h->stream_l || sl == h->stream_l1) && !h->lm_two_now)
      tk.cnt_stream = sl == h->stream_l ? h->stream_l1 : h->stream_l;
    tk.lm_pair = lm_pair_policy(h, n);
    // (the layout policy's feedback is the LM launch time: sampled ticks aside, whenever it explores or tries the other layout)
    if (tk.lm_pair >= 0 && h->lm_pair_forced < 0 && tk.lm_pair != h->lm_pair_current) tk.timed_lm = true;
    rc = run_lm(h, n, 1, false, sl, tk.lm_pair);
    h->match_by_index = false;
    if (rc) return rc;
  } else if (n && h->routed) {
    if (h->prm.denoising) {  // the denoising mask first: its bits are exchanged, phase 0 is called again behind that (ESVO_AGAIN)
      rc = routed_denoise_begin(h, tk);
      return rc ? rc : (int)ESVO_AGAIN;
    }
    rc = routed_front(h, tk, n, nullptr, nullptr);
    if (rc) return rc;
  } else if (n) {
    // own slots only (w % n_shards == shard): BM, dense local list, LM + cull on it, then the (matched, kept)
    // byte of every own slot, back to back: this rank's block of the caller's all-gather
    const u32 N = (u32)h->dp.ev_nshards, r = (u32)h->dp.ev_shard;
    const u32 own = n > r ? (n - r + N - 1) / N : 0;
    HIPCHK(hipMemsetAsync(h->d_match_flags, 0, sizeof(u32) * n, h->stream));
    rc = run_bm(h, h->d_ring[0], h->sh_first, h->ring_cap, 1, n, sel);
    if (rc) return rc;
    rc = run_order_matches(h, n, true);
    if (rc) return rc;
    rc = run_lm(h, own, 1, true);
    if (rc) return rc;
    const size_t nb = shard_codes_block(n, N);
    HIPCHK(hipMemsetAsync(h->d_codes_send, 0, nb, h->stream));
    launch_shard_codes(h->d_own_w, h->d_lkeep, h->d_counters + 8, own, N, h->d_codes_send, h->stream);
    HIPCHK(hipGetLastError());
    h->xchg_send = h->d_codes_send;
    h->xchg_recv = N > 1 ? h->d_codes_all : h->d_codes_send;
    h->xchg_block = nb;
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
  h->xchg_send = h->xchg_recv = nullptr;
  h->xchg_block = 0;
  if (n && !h->sharded) {
    // the frame waits in the staging buffer of its parity until the tick is committed and its size is known; the
    // buffer's previous frame (two ticks ago) has been copied into the ring by then
    if (tk.cnt_stream != tk.lm_stream) HIPCHK(hipStreamWaitEvent(tk.cnt_stream, h->evt[EV_LM1 + h->fpar * EV_FRONT_STRIDE], 0));
    HIPCHK(hipStreamWaitEvent(tk.cnt_stream, h->evt[EV_STG + h->fpar * EV_FRONT_STRIDE], 0));
    // latency mode: the compaction kernel leaves the counter row in the pinned host row itself (no copy operation behind it)
    h->cnt_row_host = tk.lat ? h->h_counters + CNT_ROW * h->fpar : nullptr;
    h->cnt_row_sent = false;
    // ... and only scans: the frame stays in the solver slots until the back stage's first launch -- which knows where in the window
    // ring it goes -- compacts it straight into place (one copy of the records instead of two, and that one over the whole grid)
    tk.gather = tk.lat && scan_compact_is_small(n);
    rc = run_order_points(h, n, tk.gather ? nullptr : h->d_stage[h->fpar], tk.cnt_stream);
    h->cnt_row_host = nullptr;
    if (rc) return rc;
  } else if (n) {
    const u32 N = (u32)h->dp.ev_nshards, r = (u32)h->dp.ev_shard, T = (u32)h->dp.num_threads;
    const u32 own = h->routed ? tk.n_own : (n > r ? (n - r + N - 1) / N : 0);
    // Six dependent launches for a routed tick above the single-workgroup scans' size (round 6; thirteen before): unpack (+ the
    // matched slots per scan tile), down-sweep of the matched bits (+ clearing the keep flags), keep flags by solver slot (+ clearing
    // the exchange block's cursor), their scan (2), pack.  This chain is the same on every rank whatever the number of ranks -- the
    // part of a band-mode tick that does not shrink.
    const bool tiled = h->routed && !scan_is_small(n);
    if (h->routed)
      launch_shard_unpack_routed(reinterpret_cast<const u32*>(N > 1 ? h->d_codes_all : h->d_codes_send), (u32)(shard_codes_block_routed(n) / 4), N,
                                 n, h->d_codes, h->d_rank_kept, tiled ? h->d_scan_tmp : nullptr, h->stream);
    else
      launch_shard_unpack_codes(N > 1 ? h->d_codes_all : h->d_codes_send, (u32)shard_codes_block(n, N), N, n, h->d_codes, h->d_rank_kept,
                                h->stream);
    if (tiled) launch_scan_down_code_bit0(h->d_codes, h->d_match_prefix, h->d_counters + 0, h->d_scan_tmp, n, h->d_pt_flags, h->stream);
    else launch_exclusive_scan_code_bit0(h->d_codes, h->d_match_prefix, h->d_counters + 0, h->d_scan_tmp, n, h->d_pt_flags, h->stream);
    launch_shard_keep_flags(h->d_codes, h->d_match_prefix, h->d_counters + 0, n, T, h->d_pt_flags, h->d_pts_send, h->stream);
    launch_exclusive_scan_u32(h->d_pt_flags, h->d_pt_prefix, h->d_counters + 1, h->d_scan_tmp, n, h->stream);
    (void)frame;  // filled after exchange 2 (tick_phase2)
    launch_shard_pack(h->d_own_w, h->d_lkeep, h->d_pt_slots, h->d_counters + 8, own, h->d_match_prefix, h->d_counters + 0,
                      h->d_pt_prefix, T, h->d_pts_send, own, n, h->d_rank_kept, N, h->d_counters + 9, h->stream,
                      h->routed ? h->d_counters + 10 : nullptr);
    if (tk.timed) hipEventRecord(h->evt[EV_S2 + h->fpar * EV_FRONT_STRIDE], h->stream);
    HIPCHK(hipGetLastError());
  }
  hipStream_t sc = (n && !h->sharded) ? tk.cnt_stream : h->stream;
  if (!(n && !h->sharded && h->cnt_row_sent))
    HIPCHK(hipMemcpyAsync(h->h_counters + CNT_ROW * h->fpar, h->d_counters, sizeof(u32) * CNT_ROW, hipMemcpyDeviceToHost, sc));
  h->cnt_row_sent = false;
  HIPCHK(hipEventRecord(h->evt[EV_CNT + h->fpar * EV_FRONT_STRIDE], sc));
  h->tick_pending = true;
  return ESVO_OK;
}
// phase 1b (host): wait for the counters of the tick of parity fp (one small D2H per tick: the window policy
// needs the point count), book-keeping, front-stage timings
int tick_phase1_collect(esvo_context* h, int fp) {
  return collect_front_stats(h, h->tk[fp], h->h_counters + CNT_ROW * fp, &h->evt[EV_T0 + fp * EV_FRONT_STRIDE]);
}
// the same for a front stage whose counters and events live elsewhere (`ev`: its EV_T0 .. EV_A1 set, indexed EV_x - EV_T0) --
// the tick-interleaved mode keeps them per own tick, four deep, so that a tick can be collected after the front stage two
// own ticks later has been enqueued on the same parity
int collect_front_stats(esvo_context* h, esvo_context::TickState& tk, const u32* cnt, const hipEvent_t* ev) {
  auto E = [&](int id) { return ev[id - EV_T0]; };
  HIPCHK(esvo_wait_event(E(EV_CNT), tk.lat));
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
  collect_bm_failures(h, cnt, true);
  tk.points = n_points;
  if (tk.timed) {
    s.ms_bm = s.ms_refine = 0;
    s.ms_kernel[2] = s.ms_kernel[3] = 0;
  }
  if (n && !tk.timed && tk.timed_lm && tk.lm_pair >= 0) {  // an unsampled tick whose LM launch was timed for the layout policy alone
    float lm = 0.f;
    if (hipEventElapsedTime(&lm, E(EV_LM0), E(EV_LM1)) == hipSuccess && lm > 0.f) {
      h->lm_pair_ms[tk.lm_pair][h->lm_pair_n[tk.lm_pair] & 3u] = lm;
      h->lm_pair_n[tk.lm_pair]++;
    } else {
      (void)hipGetLastError();
    }
  }
  if (n && tk.timed) {
    s.stage_timing_samples++;
    hipEventElapsedTime(&s.ms_bm, E(EV_T0), E(EV_S1));
    hipEventElapsedTime(&s.ms_refine, E(EV_S1), E(EV_S2));
    hipEventElapsedTime(&s.ms_kernel[2], E(EV_BM0), E(EV_BM1));
    hipEventElapsedTime(&s.ms_kernel[3], E(EV_LM0), E(EV_LM1));
    s.sum_ms_kernel[2] += s.ms_kernel[2];
    s.sum_ms_kernel[3] += s.ms_kernel[3];
    h->ema_lm_ms = h->ema_lm_ms > 0.f ? 0.75f * h->ema_lm_ms + 0.25f * s.ms_kernel[3] : s.ms_kernel[3];
    if (tk.lm_pair >= 0 && s.ms_kernel[3] > 0.f) {  // feedback for lm_pair_policy
      h->lm_pair_ms[tk.lm_pair][h->lm_pair_n[tk.lm_pair] & 3u] = s.ms_kernel[3];  // ring of the last four
      h->lm_pair_n[tk.lm_pair]++;
    }
  }
  if (h->tl_on && h->tl_ref && n && tk.timed) {
    const int fr[8] = {EV_T0, EV_BM0, EV_BM1, EV_S1, EV_LM0, EV_LM1, EV_S2, EV_CNT};
    std::array<float, 8> row;
    for (int i = 0; i < 8; ++i) { row[i] = -1.f; if (hipEventElapsedTime(&row[i], h->tl_ref, E(fr[i])) != hipSuccess) (void)hipGetLastError(); }
    h->tl_front.push_back(row);
  }
  tk.max_kept = (h->sharded && n) ? cnt[9] : 0;
  // exchange 2: [count | kept points], block length from the largest kept count among the ranks.  Routed band mode: the count
  // word also carries the rank's halo violations, so the exchange takes place whenever the tick had events -- a tick whose
  // violating matches were all culled (no kept point anywhere) still reports them (n is the global selection: every rank agrees)
  if (h->sharded && (n_points || (h->routed && n))) {
    h->xchg_send = h->d_pts_send;
    h->xchg_recv = h->dp.ev_nshards > 1 ? h->d_pts_all : h->d_pts_send;
    h->xchg_block = 8 + (size_t)tk.max_kept * sizeof(DevPoint);
  }
  return ESVO_OK;
}
// The tick pipeline has two stable operating points (profiles/r05_regime_timeline.txt: the stage times of both).  Normally LM
// launches run back to back and a tick's back stage (fusion 0.6 ms, then the regulariser) starts the moment its frame is ready,
// i.e. together with the NEXT LM launch: the regulariser runs beside that launch's draining tail (0.6-0.7 ms) and the back chain
// (1.25-1.3 ms) keeps pace with the LM chain (1.27-1.3).  If the back chain ever falls behind by a few milliseconds (a stall of
// its queue does it on demand; a hiccup did it in 2 of 25 sustained runs of round 4) it stays behind: the host, which may not run
// more than two back stages ahead, then issues every front stage when a back stage ENDS, so each LM launch begins ~0.4 ms into a
// fusion stage and the regulariser spends its whole launch beside the LM kernel at full occupancy -- 1.03 instead of 0.69 ms --
// which makes the back chain 1.6 ms per tick: the pace-setter, for good.
// The way back: when the symptom shows (the tick period well above the LM launch time while the back stage fills the period, three
// ticks running) the NEXT LM launch is made to wait, once, for the end of the newest enqueued back stage.  The following back
// stage and that LM launch then start together -- the fast state's alignment, one tick of lag further back, which the window
// policy does not care about.  A workload whose back chain is the slower one by nature (a reference-faithful DSEC tick) shows the
// same symptom; there the wait buys nothing, which the period after it shows, and the attempt is not repeated for 5000 ticks.
static int pipeline_resync(esvo_context* h, float wait_ms, double now_ms) {
  (void)wait_ms;
  esvo_context::Resync& r = h->resync;
  if (r.last_ms > 0.0) { const float dt = (float)(now_ms - r.last_ms); r.period_ema = r.period_ema > 0.f ? 0.8f * r.period_ema + 0.2f * dt : dt; }
  r.last_ms = now_ms;
  if (!h->resync_on) return ESVO_OK;
  if (r.check_in > 0 && --r.check_in == 0)   // did the last attempt shorten the tick?  if not, the back chain IS the pace: stop trying
    r.cooldown = (r.period_ema > 0.93f * r.period_before) ? 5000u : 50u;
  if (r.cooldown > 0) { --r.cooldown; r.streak = 0; return ESVO_OK; }
  if (r.check_in > 0) return ESVO_OK;
  const bool symptom = h->ema_lm_ms > 0.f && h->ema_back_ms > 0.f && r.period_ema > 1.2f * (h->ema_lm_ms + 0.05f) &&
                       h->ema_back_ms > 0.85f * r.period_ema;
  r.streak = symptom ? r.streak + 1 : 0;
  if (r.streak < 3) return ESVO_OK;
  r.streak = 0;
  r.period_before = r.period_ema;
  r.check_in = 24;
  r.lm_wait_back = true;   // consumed by the next tick_phase0
  h->stats.pipeline_resyncs++;
  return ESVO_OK;
}
// phase 2 (back stage): window policy, fusion + clean + regularisation of this band (halo rows recomputed locally),
// enqueued on the back stream behind the frame of the tick of parity fp.  Nothing here waits for the GPU except for
// the back stage of two ticks ago (long finished), whose pinned table and event set are reused; its timings are
// collected then.
int tick_phase2(esvo_context* h, int fp) {
  esvo_context::TickState& tk = h->tk[fp];
  h->xchg_send = h->xchg_recv = nullptr;
  h->xchg_block = 0;
  if (h->sharded) {  // the caller's all-gather was issued on the front stream after EV_CNT: every block's points to frame[seq]
    const u32 N = (u32)h->dp.ev_nshards;
    launch_shard_scatter(N > 1 ? h->d_pts_all : h->d_pts_send, 1 + (size_t)tk.max_kept * (sizeof(DevPoint) / 8), N, tk.max_kept,
                         h->d_win + tk.off, tk.n, h->stream, h->routed ? h->d_halo_viol : nullptr);
    HIPCHK(hipGetLastError());
    int rc = back_after_front(h);
    if (rc) return rc;
  } else {
    HIPCHK(hipStreamWaitEvent(h->stream_b, h->evt[EV_CNT + fp * EV_FRONT_STRIDE], 0));
  }
  const int par = h->par;
  h->par ^= 1;
  const auto t_wait0 = std::chrono::steady_clock::now();
  HIPCHK(hipEventSynchronize(h->evt[EV_RG1 + par * EV_BACK_STRIDE]));
  if (!h->sharded) {
    const auto t_wait1 = std::chrono::steady_clock::now();
    int rcr = pipeline_resync(h, std::chrono::duration<float, std::milli>(t_wait1 - t_wait0).count(),
                              std::chrono::duration<double, std::milli>(t_wait1.time_since_epoch()).count());
    if (rcr) return rcr;
  }
  collect_back(h, par);
  int rc;
  if (!h->sharded) {  // now that the size is known: exact ring space, frame copied behind the fusion that may still read it
    rc = window_reserve(h, tk.points, &tk.off);
    if (rc) return rc;
    if (tk.lat || h->pro_always) {  // the copy rides on run_fuse's first launch (context.hpp, DeferredCopies)
      h->pro = esvo_context::DeferredCopies();
      h->pro.active = true;
      h->pro.tail_b = tk.lat;
      h->pro.a_src = h->d_stage[fp]; h->pro.a_dst = h->d_win + tk.off; h->pro.a_bytes = sizeof(DevPoint) * tk.points;
      h->pro.ev_a = EV_STG + fp * EV_FRONT_STRIDE;
      if (tk.gather && tk.points) {
        h->pro.a_src = h->d_pt_slots2[fp]; h->pro.a_flags = h->d_pt_flags2[fp]; h->pro.a_prefix = h->d_pt_prefix2[fp]; h->pro.a_slots = tk.n;
        h->gather_guard[fp] = true;  // the next LM launch of this parity waits for EV_STG (tick_phase0)
      }
    } else {
      if (tk.points)
        HIPCHK(hipMemcpyAsync(h->d_win + tk.off, h->d_stage[fp], sizeof(DevPoint) * tk.points, hipMemcpyDeviceToDevice, h->stream_b));
      HIPCHK(hipEventRecord(h->evt[EV_STG + fp * EV_FRONT_STRIDE], h->stream_b));
    }
  }
  rc = commit_frame(h, tk.off, tk.points, nullptr, tk.n_pose, tk.pose_buf);
  if (rc) { (void)flush_deferred_copies(h); return rc; }
  {
    StageEventsScope timed_scope(h, tk.timed);
    rc = run_fuse(h, par, tk.T_world_obs);
  }
  if (rc) { (void)flush_deferred_copies(h); return rc; }
  h->stats.ticks++;
  if (h->sharded) h->lat_ticks++;
  h->stats.last_window_frames = (u32)h->n_window_frames;
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
  const bool poll = h->lat_last;  // the newest tick ran in latency mode: somebody is waiting for exactly this
  HIPCHK(esvo_wait_stream(h->stream, poll));
  HIPCHK(esvo_wait_stream(h->stream_l, poll)); HIPCHK(esvo_wait_stream(h->stream_l1, poll));
  HIPCHK(esvo_wait_stream(h->stream_b, poll));
  collect_ts_timing(h);
  collect_back(h, h->par);
  collect_back(h, h->par ^ 1);
  if (tick_done && h->tk[h->fpar].timed && h->back_timed[h->par ^ 1]) hipEventElapsedTime(&h->stats.ms_tick_total, h->evt[EV_T0 + h->fpar * EV_FRONT_STRIDE], h->evt[EV_RG1 + (h->par ^ 1) * EV_BACK_STRIDE]);
  return ESVO_OK;
}
}  // namespace esvo_host

extern "C" int esvo_map_tick_resident(esvo_handle h, uint64_t t_ns, const double T_world_cam[16], const uint64_t* pose_t_ns,
                                      const double* pose_T, size_t m) {
  if (!h || !T_world_cam || !pose_t_ns || !pose_T) return ESVO_ERR_INVALID_ARG;
  API_LOCK(h);
  HIPCHK(hipSetDevice(h->device));
  // = esvo_ts_render x2 + esvo_map_set_observation + esvo_map_tick with both cameras in one launch per kernel; an
  // un-smoothed observation is written by the remap itself (no device-to-device copies)
  begin_observation(h);
  uint8_t* obs[2] = {h->d_obs[0], h->d_obs[1]};
  int rc = ts_render_pair(h, t_ns, h->prm.smooth_time_surface ? nullptr : obs);
  if (rc) { revert_observation(h); return rc; }
  if (h->prm.smooth_time_surface)  // createMatchProblem applies GaussianBlurTS(5) when SmoothTimeSurface (EventBM.cpp:68-72)
    launch_gaussian5_pair(h->d_ts[0], h->d_ts[1], h->d_obs[0], h->d_obs[1], h->W, h->H, h->stream, h->routed ? h->oband_y0 : 0,
                          h->routed ? h->oband_y1 : -1);
  HIPCHK(hipGetLastError());
  std::memcpy(h->T_world_obs, T_world_cam, sizeof(double) * 16);
  h->obs_t_ns = t_ns;
  h->obs_set = true;
  return esvo_map_tick(h, t_ns, pose_t_ns, pose_T, m);
}

extern "C" int esvo_map_tick(esvo_handle h, uint64_t t_ns, const uint64_t* pose_t_ns, const double* pose_T, size_t m) {
  if (!h || !pose_t_ns || !pose_T) return ESVO_ERR_INVALID_ARG;
  API_LOCK(h);
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
  h->split_now = h->lm_split && !h->prm.denoising;
  h->lat_now = h->lat_mode && !prev;
  h->pipe_now = h->lat_mode && prev;
  int rc = tick_phase0(h, t_ns, pose_t_ns, pose_T, m);
  if (!rc) rc = tick_phase1_enqueue(h);
  h->split_now = false;
  h->lat_now = h->pipe_now = false;
  h->stage_events_on = true;
  if (!rc) {
    h->lat_last = h->tk[h->fpar].lat;
    if (!prev) h->lat_ticks++;
  }
  if (rc) {  // the failed tick leaves no trace: the previous one (if pending) stays pending on ITS parity and is
    h->fpar = prev_fp;  // completed -- with its own counters and staging buffer -- by the next call that needs it
    return rc;
  }
  if (prev) {
    rc = tick_phase1_collect(h, prev_fp);
    if (!rc) rc = tick_phase2(h, prev_fp);
    h->tick_pending = true;  // this tick (its front stage is enqueued whatever happened to the previous one)
    if (rc) return rc;
  }
  return ESVO_OK;
}

// ---- esvo_MVStereo's PURE_BLOCK_MATCHING mode (MVStereoMode 1, esvo_MVStereo.cpp:383-432) -------------------------------------
// Event selection + (denoising) + block matching as in every tick; then vEMP2vDP (:1072-1094) instead of the nonlinear
// refinement, a window of maxNumFusionFrames frames whatever the fusion strategy (:419-421), and
// DepthFusion::naive_propagation of every frame, newest first, into a new DepthFrame (:422-423) -- no culling, no clean, no
// regularisation.  Synchronous (a visualisation baseline: nothing is pipelined).
extern "C" int esvo_map_tick_bm_only(esvo_handle h, uint64_t t_ns, const uint64_t* pose_t_ns, const double* pose_T, size_t m) {
  if (!h || !pose_t_ns || !pose_T) return ESVO_ERR_INVALID_ARG;
  API_LOCK(h);
  if (!h->obs_set) FAIL(ESVO_ERR_STATE, "esvo_map_set_observation has not been called");
  if (h->sharded) FAIL(ESVO_ERR_STATE, "handle is sharded");
  HIPCHK(hipSetDevice(h->device));
  int rc = flush_pending_tick(h);
  if (rc) return rc;
  // phase 0 up to the match list (tick_phase0 also enqueues the LM kernel, which this mode does not run), on the front stream
  if (m > h->max_poses) FAIL(ESVO_ERR_CAPACITY, "pose table larger than max_poses_per_tick");
  u32 n = 0;
  u64 first = 0;
  rc = select_events(h, t_ns, &first, &n);
  if (!rc) rc = upload_poses(h, pose_t_ns, pose_T, m, h->d_counters2[h->fpar ^ 1]);
  if (rc) return rc;
  // This mode keeps maxNumFusionFrames frames of up to PROCESS_EVENT_NUM un-culled matches whatever the fusion strategy, while
  // the window ring is sized for the normal policy (max_window_points): a CONST_POINTS preset with a small point budget and
  // many frames can run out of ring.  Find that out HERE, before any tick state flips: the frame that leaves at this tick
  // leaves first (push_back + pop_front while size > max == pop while size >= max, then push), and the ring must take a
  // frame of n points (n = the selected events bounds the matches).
  // (probed on a COPY of the window: a refused tick, or one that fails further down, has dropped no frame)
  const size_t keep_below = (size_t)std::max(1, h->prm.max_fusion_frames);
  if (window_probe_after_pops(h, keep_below, n) != ESVO_OK)
    FAIL(ESVO_ERR_CAPACITY, "PURE_BLOCK_MATCHING window (maxNumFusionFrames frames of up to PROCESS_EVENT_NUM matches) "
                            "does not fit the fusion window ring: raise max_window_points");
  h->fpar ^= 1;
  h->d_matches = h->d_matches2[h->fpar];
  h->d_counters = h->d_counters2[h->fpar];
  set_lm_parity(h);
  HIPCHK(hipStreamWaitEvent(h->stream, h->evt[EV_RG1 + h->par * EV_BACK_STRIDE], 0));
  hipEventRecord(h->evt[EV_T0 + h->fpar * EV_FRONT_STRIDE], h->stream);
  const u32* sel = nullptr;
  if (h->prm.denoising && n) {
    launch_denoise_flags(h->d_ring[0], h->sh_first, h->ring_cap, n, h->d_evmap, h->d_match_flags, h->W, h->H, h->stream);
    launch_exclusive_scan_u32(h->d_match_flags, h->d_match_prefix, h->d_counters + 5, h->d_scan_tmp, n, h->stream);
    launch_denoise_select(h->d_match_flags, h->d_match_prefix, n, h->d_sel, h->stream);
    rc = read_counters(h);
    if (rc) return rc;
    n = h->h_counters[5];
    sel = h->d_sel;
  }
  u32 n_matches = 0;
  if (n) {
    rc = run_bm(h, h->d_ring[0], h->sh_first, h->ring_cap, 1, n, sel);
    if (rc) return rc;
    rc = run_order_matches(h, n, false);
    if (rc) return rc;
    launch_matches_to_points(h->d_matches, h->d_counters + 0, n, h->d_pts_tmp, h->dp, h->stream);
    HIPCHK(hipGetLastError());
    rc = read_counters(h);
    if (rc) return rc;
    n_matches = h->h_counters[0];
    collect_bm_failures(h, h->h_counters, true);
  }
  esvo_stats_t& s = h->stats;
  s.last_events_in = n; s.last_matches = n_matches; s.last_solved = 0; s.last_points = n_matches;
  s.total_events_in += n; s.total_matches += n_matches; s.total_points += n_matches;
  // dqvDepthPoints_.push_back(vdp_em); while (size > maxNumFusionFrames_) pop_front()
  rc = back_after_front(h);
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(h->stream_b));  // the ring space may still be read by a fusion in flight
  // the frame that leaves at this tick leaves first (push_back + pop_front while size > max == pop while size >= max, then
  // push) -- now that every fallible step of the front stage is behind us; the probe above guarantees the space
  while (h->n_window_frames && h->n_window_frames >= keep_below) pop_front_frame(h);
  u32 off;
  rc = window_reserve(h, n_matches, &off);
  if (rc) return rc;
  if (n_matches)
    HIPCHK(hipMemcpyAsync(h->d_win + off, h->d_pts_tmp, sizeof(DevPoint) * n_matches, hipMemcpyDeviceToDevice, h->stream_b));
  rc = commit_frame(h, off, n_matches, nullptr, h->n_pose, h->pose_buf, false);
  if (rc) return rc;
  while (h->n_window_frames > (size_t)h->prm.max_fusion_frames) pop_front_frame(h);
  const int par = h->par;
  h->par ^= 1;
  HIPCHK(hipEventSynchronize(h->evt[EV_RG1 + par * EV_BACK_STRIDE]));
  collect_back(h, par);
  rc = run_fuse(h, par, h->T_world_obs, true);
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(h->stream_b));
  collect_back(h, par);
  h->committed_t_ns = t_ns;
  s.ticks++;
  s.last_window_frames = (u32)h->n_window_frames;
  u32 np = 0;
  for (auto& f : h->frames) np += f.count;
  s.last_window_points = np;
  return ESVO_OK;
}

// The stage-wise seam of the same mode: what follows match_all_HyperThread in PURE_BLOCK_MATCHING (esvo_MVStereo.cpp:411-423)
// on matches the caller holds (esvo_map_match gave them): vEMP2vDP, dqvDepthPoints_.push_back + pop to maxNumFusionFrames_,
// naive_propagation of the window (newest first) into a new DepthFrame at the observation's pose.  Synchronous.
extern "C" int esvo_map_fuse_matches_naive(esvo_handle h, const esvo_match_t* matches, size_t n, const double* pose_T, size_t m) {
  if (!h || (n && !matches) || (m && !pose_T)) return ESVO_ERR_INVALID_ARG;
  API_LOCK(h);
  if (!h->obs_set) FAIL(ESVO_ERR_STATE, "esvo_map_set_observation has not been called");
  if (h->sharded) FAIL(ESVO_ERR_STATE, "handle is sharded");
  if (n > h->max_ev) FAIL(ESVO_ERR_CAPACITY, "more matches than max_events_per_tick");
  if (m > h->max_poses) FAIL(ESVO_ERR_CAPACITY, "pose table larger than max_poses_per_tick");
  for (size_t i = 0; i < n; ++i)
    if (matches[i].pose_idx >= m) FAIL(ESVO_ERR_INVALID_ARG, "match refers to a pose outside the pose table");
  HIPCHK(hipSetDevice(h->device));
  int rc = flush_pending_tick(h);
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(h->stream_l)); HIPCHK(hipStreamSynchronize(h->stream_l1));
  HIPCHK(hipStreamSynchronize(h->stream_b));  // the staging buffers and the ring may still be read by work in flight
  const u32 n32 = (u32)n;
  if (n) {
    HIPCHK(hipMemcpyAsync(h->d_matches, matches, sizeof(esvo_match_t) * n, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_counters, &n32, sizeof(u32), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));  // n32 / matches are borrowed
    launch_matches_to_points(h->d_matches, h->d_counters + 0, n32, h->d_pts_tmp, h->dp, h->stream);
    HIPCHK(hipGetLastError());
  }
  u32 off;
  // the frame that leaves at this call leaves first (its ring space is free: stream_b was drained above) -- probed on a copy
  // of the window, popped for real only when nothing can fail any more before the frame is committed
  const size_t keep_below = (size_t)std::max(1, h->prm.max_fusion_frames);
  rc = window_probe_after_pops(h, keep_below, n32);
  if (rc) return rc;
  rc = back_after_front(h);
  if (rc) return rc;
  while (h->n_window_frames && h->n_window_frames >= keep_below) pop_front_frame(h);
  rc = window_reserve(h, n32, &off);
  if (rc) return rc;
  if (n) HIPCHK(hipMemcpyAsync(h->d_win + off, h->d_pts_tmp, sizeof(DevPoint) * n, hipMemcpyDeviceToDevice, h->stream_b));
  static const double ident[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  rc = commit_frame(h, off, n32, m ? pose_T : ident, (u32)m, 0, false);
  if (rc) return rc;
  while (h->n_window_frames > (size_t)h->prm.max_fusion_frames) pop_front_frame(h);
  const int par = h->par;
  h->par ^= 1;
  HIPCHK(hipEventSynchronize(h->evt[EV_RG1 + par * EV_BACK_STRIDE]));
  collect_back(h, par);
  rc = run_fuse(h, par, h->T_world_obs, true);
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(h->stream_b));
  collect_back(h, par);
  h->committed_t_ns = h->obs_t_ns;
  h->stats.last_points = n32;
  h->stats.last_window_frames = (u32)h->n_window_frames;
  u32 np = 0;
  for (auto& f : h->frames) np += f.count;
  h->stats.last_window_points = np;
  return ESVO_OK;
}

// ---- SGM initialisation -----------------------------------------------------------------------
// Replaces esvo_Mapping::InitializationAtTime (esvo_Mapping.cpp:433-492) with the SGM branch of dataTransferring (:537-552):
// cv::StereoSGBM on the UN-smoothed Time-Surface pair, the rectified pixels of the newest <= PROCESS_EVENT_NUM + 1 left
// events of the last 2 * BM_half_slice_thickness as edge mask, one Gaussian DepthPoint (variance 1e-6, age =
// age_vis_threshold) per masked event with a disparity inside the inverse-depth range; if at least min_points
// (INIT_SGM_DP_NUM_THRESHOLD) come out they open the fusion window and DepthFusion::naive_propagation fills the DepthFrame.
extern "C" int esvo_map_init_sgm(esvo_handle h, const uint8_t* ts_left, const uint8_t* ts_right, size_t min_points, size_t* n_points,
                                 int16_t* disp_out) {
  if (!h || !n_points) return ESVO_ERR_INVALID_ARG;
  API_LOCK(h);
  if (!h->obs_set) FAIL(ESVO_ERR_STATE, "esvo_map_set_observation has not been called (time stamp and pose of the Time-Surface pair)");
  if (h->sharded) FAIL(ESVO_ERR_STATE, "handle is sharded");
  if (h->W <= 48 + 2) FAIL(ESVO_ERR_UNSUPPORTED, "image narrower than numDisparities");
  HIPCHK(hipSetDevice(h->device));
  int rc = flush_pending_tick(h);
  if (rc) return rc;
  const size_t npx = (size_t)h->W * h->H;
  if (!h->sgm_ok) {
    const size_t nvol = (size_t)h->H * (h->W - 48) * 48;
    uint8_t** planes[4] = {&h->sgm.sobL, &h->sgm.rawL, &h->sgm.sobR, &h->sgm.rawR};
    for (auto pp : planes) HIPCHK(hipMalloc(reinterpret_cast<void**>(pp), npx));
    for (int i = 0; i < 6; ++i) HIPCHK(hipMalloc(reinterpret_cast<void**>(&h->sgm.vol[i]), nvol * sizeof(int16_t)));
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&h->sgm.d1), npx * sizeof(int16_t)));
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&h->sgm.d1b), npx * sizeof(int16_t)));
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&h->sgm.d2key), npx * sizeof(u32)));
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&h->d_sgm_img[0]), npx));
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&h->d_sgm_img[1]), npx));
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&h->d_sgm_disp), npx * sizeof(int16_t)));
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&h->d_sgm_pair), sizeof(u32) * 8 * (size_t)h->max_ev));
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&h->d_sgm_T), sizeof(double) * 16));
    h->sgm_ok = true;
  }
  const uint8_t* src[2] = {ts_left, ts_right};
  const uint8_t* img[2];
  for (int cam = 0; cam < 2; ++cam) {
    if (src[cam]) {
      HIPCHK(hipMemcpyAsync(h->d_sgm_img[cam], src[cam], npx, hipMemcpyHostToDevice, h->stream));
      img[cam] = h->d_sgm_img[cam];
    } else {
      if (!h->ts_valid[cam]) FAIL(ESVO_ERR_STATE, "no device-resident Time Surface: call esvo_ts_render first");
      img[cam] = h->d_ts[cam];
    }
  }
  HIPCHK(hipStreamSynchronize(h->stream_l)); HIPCHK(hipStreamSynchronize(h->stream_l1));
  HIPCHK(hipStreamSynchronize(h->stream_b));  // the DepthMap and the window are rebuilt below
  launch_sgbm(img[0], img[1], h->sgm, h->d_sgm_disp, h->W, h->H, h->stream);
  HIPCHK(hipGetLastError());
  // the SGM event selection (esvo_Mapping.cpp:541-551): newest first from lower_bound(t), 2 * BM_half_slice_thickness back
  u64 first = 0;
  u32 n = 0;
  {
    std::lock_guard<std::mutex> lr(h->mu_ring);
    ingest_fence(h, 0);
    const double t_end = ns_to_sec(h->obs_t_ns);
    const double t_begin = ns_to_sec(ros_time_from_sec(std::max(0.0, t_end - 2 * h->prm.bm_half_slice_thickness)));
    const u64 it_end = lower_bound_sec(h, 0, t_end), it_begin = lower_bound_sec(h, 0, t_begin);
    const u64 staged_end = h->ring_base[0] + h->ts_host[0].size();
    u64 avail = it_end - it_begin;
    first = it_end;
    if (it_end == staged_end && avail > 0) { first = it_end - 1; avail -= 1; }  // end() is skipped (oracle definition)
    n = (u32)std::min<u64>(avail, (u64)h->prm.process_event_num + 1);
    if (n > h->max_ev) FAIL(ESVO_ERR_CAPACITY, "more events than max_events_per_tick");
    if (n && first - (n - 1) < h->ring_reserved[0] - std::min<u64>(h->ring_reserved[0], h->ring_cap))
      FAIL(ESVO_ERR_STATE, "selected events were already overwritten in the event ring");
    if (n) { h->sh_first_prev = h->sh_first; h->sh_first = first; }  // the ingest thread's overwrite guard protects this selection like a tick's
  }
  HIPCHK(hipMemsetAsync(h->d_counters, 0, sizeof(u32) * CNT_ROW, h->stream));
  u32 count = 0;
  if (n) {
    HIPCHK(hipStreamWaitEvent(h->stream, h->evt[EV_RG1 + h->par * EV_BACK_STRIDE], 0));
    launch_sgm_points(h->d_ring[0], first, h->ring_cap, n, h->d_lut, h->d_sgm_disp, h->d_pt_slots, h->d_pt_flags, h->dp, h->stream);
    launch_exclusive_scan_u32(h->d_pt_flags, h->d_pt_prefix, h->d_counters + 1, h->d_scan_tmp, n, h->stream);
    HIPCHK(hipMemcpyAsync(h->d_counters + 0, &n, sizeof(u32), hipMemcpyHostToDevice, h->stream));  // compaction bound (n_in of compact_points), behind the memset above
    launch_compact_points(h->d_pt_slots, h->d_pt_flags, h->d_pt_prefix, h->d_counters + 0, n, h->d_pts_tmp, h->stream);
    rc = read_counters(h);
    if (rc) return rc;
    count = h->h_counters[1];
  }
  if (disp_out) {
    HIPCHK(hipMemcpyAsync(disp_out, h->d_sgm_disp, npx * sizeof(int16_t), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
  }
  *n_points = 0;
  if (count < min_points) return ESVO_OK;  // InitializationAtTime returns false: nothing is pushed (:482-483)
  u32 off;
  rc = window_reserve(h, count, &off);
  if (rc) return rc;
  rc = back_after_front(h);
  if (rc) return rc;
  if (count) HIPCHK(hipMemcpyAsync(h->d_win + off, h->d_pts_tmp, sizeof(DevPoint) * count, hipMemcpyDeviceToDevice, h->stream_b));
  rc = commit_frame(h, off, count, h->T_world_obs, 1, 0, false);  // dqvDepthPoints_.push_back(vdp_sgm): no window policy (:485)
  if (rc) return rc;
  // DepthFusion::naive_propagation into a new DepthFrame at the observation's pose (:436-440, :486)
  std::memcpy(h->T_world_frame, h->T_world_obs, sizeof(double) * 16);
  double Tfw[16], Tfo[16];
  rigid_inverse(h->T_world_frame, Tfw);
  mat4_mul(Tfw, h->T_world_obs, Tfo);
  HIPCHK(hipMemcpy(h->d_sgm_T, Tfo, sizeof(double) * 16, hipMemcpyHostToDevice));
  launch_sgm_naive(h->d_win + off, count, h->d_sgm_T, h->d_owner_max, h->d_sgm_pair, h->d_sgm_pair + 4 * (size_t)h->max_ev, h->d_cnt_b + 4,
                   h->d_scan_tmp_b, h->d_map, h->dp, h->stream_b);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(h->stream_l)); HIPCHK(hipStreamSynchronize(h->stream_l1));
  HIPCHK(hipStreamSynchronize(h->stream_b));
  h->d_map_cur = h->d_map;
  h->committed_t_ns = h->obs_t_ns;
  h->stats.last_points = count;
  h->stats.last_window_frames = (u32)h->n_window_frames;
  *n_points = count;
  return ESVO_OK;
}

// ---- device-resident stage calls: the building blocks of tick-interleaved multi-GPU operation ---------------------
// (rank r maps the ticks k with k % N == r completely; a tick needs nothing from the previous DepthMaps -- the
// DepthFrame is rebuilt from the window at every tick, esvo_Mapping.cpp:266-272 -- only the frames of the last
// ticks, which the ranks all-gather; see esvo_amd/dist.py)
extern "C" int esvo_map_front(esvo_handle h, uint64_t t_ns, const uint64_t* pose_t_ns, const double* pose_T, size_t m,
                              size_t* n_points) {
  if (!h || !pose_t_ns || !pose_T || !n_points) return ESVO_ERR_INVALID_ARG;
  API_LOCK(h);
  if (!h->obs_set) FAIL(ESVO_ERR_STATE, "esvo_map_set_observation has not been called");
  if (h->sharded) FAIL(ESVO_ERR_STATE, "handle is sharded by slot/band: esvo_map_front maps whole ticks");
  HIPCHK(hipSetDevice(h->device));
  int rc = flush_pending_tick(h);
  if (rc) return rc;
  rc = tick_phase0(h, t_ns, pose_t_ns, pose_T, m);
  if (rc) return rc;
  const u32 n = h->tk[h->fpar].n;
  if (n) { rc = run_order_points(h, n, h->d_pts_tmp); if (rc) return rc; }
  HIPCHK(hipMemcpyAsync(h->h_counters + CNT_ROW * h->fpar, h->d_counters, sizeof(u32) * CNT_ROW, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipEventRecord(h->evt[EV_CNT + h->fpar * EV_FRONT_STRIDE], h->stream));
  rc = tick_phase1_collect(h, h->fpar);
  if (rc) return rc;
  *n_points = h->tk[h->fpar].points;
  return ESVO_OK;
}

extern "C" int esvo_map_front_frame(esvo_handle h, const esvo_depth_point_t** d_frame) {
  if (!h || !d_frame) return ESVO_ERR_INVALID_ARG;
  API_LOCK(h);
  *d_frame = h->d_pts_tmp;
  return ESVO_OK;
}

extern "C" int esvo_map_push_frame_device(esvo_handle h, const esvo_depth_point_t* d_pts, size_t n, const double* pose_T,
                                          size_t m) {
  if (!h || (n && !d_pts) || (m && !pose_T)) return ESVO_ERR_INVALID_ARG;
  API_LOCK(h);
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
  API_LOCK(h);
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
  h->stats.last_window_frames = (u32)h->n_window_frames;
  u32 np = 0;
  for (auto& f : h->frames) np += f.count;
  h->stats.last_window_points = np;
  h->stats_pending = true;
  return ESVO_OK;
}

extern "C" int esvo_shard_tick_phase(esvo_handle h, int phase, uint64_t t_ns, const uint64_t* pose_t_ns, const double* pose_T,
                                     size_t m) {
  if (!h) return ESVO_ERR_INVALID_ARG;
  API_LOCK(h);
  if (!h->obs_set) FAIL(ESVO_ERR_STATE, "esvo_map_set_observation has not been called");
  if (!h->sharded) FAIL(ESVO_ERR_STATE, "call esvo_shard_set_band first");
  HIPCHK(hipSetDevice(h->device));
  // (stage-timing events are sampled, context.hpp lat_ticks: phase 0 decides for the tick; switched back on when the call returns)
  StageEventsScope timed_scope(h, phase == 0 ? true : h->tk[h->fpar].timed);
  switch (phase) {
    case 0:
      if (!h->dn_pending && (!pose_t_ns || !pose_T)) return ESVO_ERR_INVALID_ARG;
      return tick_phase0(h, t_ns, pose_t_ns, pose_T, m);  // (ESVO_AGAIN: exchange, then phase 0 once more -- Denoising on a routed handle)
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
  API_LOCK(h);
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
  API_LOCK(h);
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
  API_LOCK(h);
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

// pc_near_ of publishPointCloud (esvo_Mapping.cpp:925-932): what the global-cloud voxel filter is fed
int esvo_map_get_pointcloud_near_xyz(esvo_handle h, double visualize_range, float* out_xyz, size_t cap_points, size_t* n) {
  if (!h || !n) return ESVO_ERR_INVALID_ARG;
  API_LOCK(h);
  HIPCHK(hipSetDevice(h->device));
  { int rcp = flush_pending_tick(h); if (rcp) return rcp; }
  std::vector<esvo_depth_point_t> v;
  int rc = export_map(h, v, nullptr);
  if (rc) return rc;
  const double* T = h->T_world_frame;
  size_t k = 0;
  for (size_t i = 0; i < v.size(); ++i) {
    const double* q = v[i].p_cam;
    if (!(std::sqrt((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) < visualize_range)) continue;
    if (out_xyz) {
      if (k >= cap_points) FAIL(ESVO_ERR_CAPACITY, "output array too small for the point cloud");
      for (int r = 0; r < 3; ++r)
        out_xyz[3 * k + r] = (float)(((T[r * 4 + 0] * q[0] + T[r * 4 + 1] * q[1]) + T[r * 4 + 2] * q[2]) + T[r * 4 + 3]);
    }
    ++k;
  }
  *n = k;
  return ESVO_OK;
}

// pcl::VoxelGrid<PointXYZ> with a cubic leaf (esvo_Mapping.cpp:960-964): host code, as in the reference -- it runs on a
// few ten thousand points once per visualizeGPC_interval.  Float arithmetic throughout; one centroid per occupied voxel
// in ascending voxel index (x fastest); the points of a voxel are summed in input order.
int esvo_voxel_filter_xyz(const float* xyz, size_t n, float leaf, float* out_xyz, size_t cap_points, size_t* n_out) {
  esvo_context* h = nullptr;
  if ((n && !xyz) || !n_out || !(leaf > 0)) return ESVO_ERR_INVALID_ARG;
  std::vector<size_t> fin;
  float mn[3] = {0, 0, 0}, mx[3] = {0, 0, 0};
  for (size_t i = 0; i < n; ++i) {
    const float* p = xyz + 3 * i;
    if (!std::isfinite(p[0]) || !std::isfinite(p[1]) || !std::isfinite(p[2])) continue;
    if (fin.empty()) for (int c = 0; c < 3; ++c) mn[c] = mx[c] = p[c];
    for (int c = 0; c < 3; ++c) { mn[c] = std::min(mn[c], p[c]); mx[c] = std::max(mx[c], p[c]); }
    fin.push_back(i);
  }
  *n_out = 0;
  if (fin.empty()) return ESVO_OK;
  const float inv = 1.0f / leaf;
  long long minb[3], divb[3];
  for (int c = 0; c < 3; ++c) {
    minb[c] = (long long)std::floor(mn[c] * inv);
    divb[c] = (long long)std::floor(mx[c] * inv) - minb[c] + 1;
  }
  if ((double)divb[0] * (double)divb[1] * (double)divb[2] > 2147483647.0)
    FAIL(ESVO_ERR_CAPACITY, "leaf size too small for the extent of the cloud (voxel index overflows, as in pcl::VoxelGrid)");
  std::vector<std::pair<long long, size_t>> idx;
  idx.reserve(fin.size());
  for (size_t i : fin) {
    const float* p = xyz + 3 * i;
    const long long a = (long long)std::floor(p[0] * inv) - minb[0], b = (long long)std::floor(p[1] * inv) - minb[1],
                    c = (long long)std::floor(p[2] * inv) - minb[2];
    idx.emplace_back(a + b * divb[0] + c * divb[0] * divb[1], i);
  }
  std::stable_sort(idx.begin(), idx.end(),
                   [](const std::pair<long long, size_t>& x, const std::pair<long long, size_t>& y) { return x.first < y.first; });
  size_t k = 0;
  for (size_t a = 0; a < idx.size();) {
    size_t b = a;
    float c[3] = {0, 0, 0};
    while (b < idx.size() && idx[b].first == idx[a].first) {
      for (int d = 0; d < 3; ++d) c[d] += xyz[3 * idx[b].second + d];
      ++b;
    }
    if (out_xyz) {
      if (k >= cap_points) FAIL(ESVO_ERR_CAPACITY, "output array too small for the filtered cloud");
      for (int d = 0; d < 3; ++d) out_xyz[3 * k + d] = c[d] / (float)(b - a);
    }
    ++k;
    a = b;
  }
  *n_out = k;
  return ESVO_OK;
}

// Visualization::plot_map x 4 with publishMappingResults' arguments (esvo_Mapping.cpp:868-884)
int esvo_map_get_debug_images(esvo_handle h, double age_max_range, uint8_t* inv_depth_bgr, uint8_t* stdvar_bgr, uint8_t* age_bgr,
                              uint8_t* cost_bgr) {
  if (!h) return ESVO_ERR_INVALID_ARG;
  API_LOCK(h);
  HIPCHK(hipSetDevice(h->device));
  { int rcp = flush_pending_tick(h); if (rcp) return rcp; }
  const size_t npx = (size_t)h->W * h->H;
  if (!h->d_viz_bgr) {
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&h->d_viz_bgr), npx * 3));
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&h->d_viz_owner), npx * sizeof(u32)));
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&h->d_viz_jet), 768));
    // the reference's colour tables are jet on i / 255 (Visualization.cpp:128-226): 255 * channel =
    // clamp(min(4 i + a, -4 i + b), 0, 255), stored in an 8-bit image by rounding half to even
    uint8_t jet[768];
    const double ab[3][2] = {{127.5, 637.5}, {-127.5, 892.5}, {-382.5, 1147.5}};  // B, G, R
    for (int i = 0; i < 256; ++i)
      for (int c = 0; c < 3; ++c) {
        double v = std::min(4.0 * i + ab[c][0], -4.0 * i + ab[c][1]);
        v = v < 0 ? 0 : (v > 255 ? 255 : v);
        jet[3 * i + c] = (uint8_t)std::nearbyint(v);
      }
    HIPCHK(hipMemcpy(h->d_viz_jet, jet, 768, hipMemcpyHostToDevice));
  }
  const esvo_params_t& p = h->prm;
  const double cost_thr = p.residual_vis_threshold * p.residual_vis_threshold * (p.patch_size_x * p.patch_size_y);  // esvo_Mapping.cpp:97
  struct { uint8_t* out; int type; double mx, mn, t1, t2; } img[4] = {
      {inv_depth_bgr, 0, p.invdepth_max, p.invdepth_min, p.stdvar_vis_threshold, p.age_vis_threshold},
      {stdvar_bgr, 1, p.stdvar_vis_threshold, 0.0, p.stdvar_vis_threshold, 0.0},
      {age_bgr, 3, age_max_range, 0.0, p.age_vis_threshold, 0.0},
      {cost_bgr, 2, cost_thr, 0.0, cost_thr, 0.0}};
  for (auto& im : img) {
    if (!im.out) continue;
    launch_debug_image(h->d_map_cur, h->d_viz_owner, h->d_viz_bgr, h->d_viz_jet, im.type, im.mx, im.mn, im.t1, im.t2, h->dp, h->stream_b);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(im.out, h->d_viz_bgr, npx * 3, hipMemcpyDeviceToHost, h->stream_b));
    HIPCHK(hipStreamSynchronize(h->stream_l)); HIPCHK(hipStreamSynchronize(h->stream_l1));
  HIPCHK(hipStreamSynchronize(h->stream_b));
  }
  return ESVO_OK;
}

// esvo_MVStereo::saveDepthMap (esvo_MVStereo.cpp:982-1000), the reference's only DepthMap dump: the file <save_dir><t_ns>.txt with
// one line per valid element (inverse depth > -1e-6, DepthPoint::valid() without arguments) in list order:
//     of << it->x().transpose() << " " << it->p_cam()(2) << "\n"
// Eigen's operator<< with the default IOFormat prints the 1 x 2 row vector with the stream's precision (6 significant digits,
// general format) and ALIGNED columns: both coefficients right-aligned to the longer one's width, separated by one blank; the
// depth follows as a plain double.  (Eigen is third-party and absent here: restated from its documented default format.)
int esvo_map_save_depth_map(esvo_handle h, const char* save_dir, uint64_t t_ns, size_t* n_written) {
  if (!h || !save_dir) return ESVO_ERR_INVALID_ARG;
  API_LOCK(h);
  HIPCHK(hipSetDevice(h->device));
  { int rcp = flush_pending_tick(h); if (rcp) return rcp; }
  std::vector<esvo_depth_point_t> v;
  int rc = export_map(h, v, nullptr);
  if (rc) return rc;
  const std::string path = std::string(save_dir) + std::to_string((unsigned long long)t_ns) + ".txt";
  FILE* f = std::fopen(path.c_str(), "w");
  if (!f) FAIL(ESVO_ERR_INVALID_ARG, "cannot open " + path);
  size_t n = 0;
  for (const esvo_depth_point_t& e : v) {
    if (!(e.inv_depth > -1e-6)) continue;
    char a[64], b[64];
    std::snprintf(a, sizeof(a), "%g", e.x[0]);
    std::snprintf(b, sizeof(b), "%g", e.x[1]);
    const int w = (int)std::max(std::strlen(a), std::strlen(b));
    std::fprintf(f, "%*s %*s %g\n", w, a, w, b, e.p_cam[2]);
    ++n;
  }
  std::fclose(f);
  if (n_written) *n_written = n;
  return ESVO_OK;
}

int esvo_map_get_last_frame(esvo_handle h, esvo_depth_point_t* out, size_t cap, size_t* n) {
  if (!h || !n) return ESVO_ERR_INVALID_ARG;
  API_LOCK(h);
  HIPCHK(hipSetDevice(h->device));
  { int rcp = flush_pending_tick(h); if (rcp) return rcp; }
  *n = 0;
  if (h->frames.empty()) return ESVO_OK;
  const FrameRec& f = h->frames.back();
  *n = f.count;
  if (out && f.count) {
    if (f.count > cap) FAIL(ESVO_ERR_CAPACITY, "output array too small for the frame");
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipStreamSynchronize(h->stream_b));  // (the frame was copied into the ring on the back stream)
    HIPCHK(hipMemcpy(out, h->d_win + f.off, sizeof(esvo_depth_point_t) * f.count, hipMemcpyDeviceToHost));
  }
  return ESVO_OK;
}

int esvo_get_stats(esvo_handle h, esvo_stats_t* out) {
  if (!h || !out) return ESVO_ERR_INVALID_ARG;
  API_LOCK(h);
  HIPCHK(hipSetDevice(h->device));
  int rc = finalize_tick_stats(h);
  if (rc) return rc;
  {  // the LM kernel's clock probe (every stream is drained here): running sums since esvo_create / esvo_reset
    u64 acc[CLK_SCRATCH], acc1[CLK_SCRATCH];
    HIPCHK(hipMemcpy(acc, h->d_clk, sizeof(acc), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(acc1, h->d_clk + clk_words(h->max_ev), sizeof(acc1), hipMemcpyDeviceToHost));
    for (u32 x = 0; x < CLK_XCDS; ++x) { h->stats.clk_cycles[x] = acc[2 * x] + acc1[2 * x]; h->stats.clk_ref_ticks[x] = acc[2 * x + 1] + acc1[2 * x + 1]; }
    h->stats.clk_samples = acc[CLK_SAMPLES] + acc1[CLK_SAMPLES];
  }
  std::lock_guard<std::mutex> lr(h->mu_ring);  // events_staged is written by the ingest thread
  *out = h->stats;
  return ESVO_OK;
}

// ---- Multi-GPU row-band sharding ------------------------------------------------------------------
namespace {
bool rings_empty(esvo_context* h) {
  std::lock_guard<std::mutex> lr(h->mu_ring);
  return h->ring_next[0] == 0 && h->ring_next[1] == 0 && h->glob_ts.empty();
}
void free_shard_blocks(esvo_context* h) {
  for (void** p : {(void**)&h->d_codes_send, (void**)&h->d_codes_all, (void**)&h->d_pts_send, (void**)&h->d_pts_all, (void**)&h->d_rank_kept, (void**)&h->d_ring_gidx})
    if (*p) { (void)hipFree(*p); *p = nullptr; }
}
}  // namespace

int esvo_shard_set_band(esvo_handle h, int row_begin, int row_end, int shard, int n_shards) {
  if (!h || row_begin < 0 || row_end > h->H || row_begin >= row_end || n_shards < 1 || shard < 0 || shard >= n_shards ||
      n_shards > (int)esvo_context::SHARD_MAX_RANKS)
    return ESVO_ERR_INVALID_ARG;
  API_LOCK(h);
  { int rcp = flush_pending_tick(h); if (rcp) return rcp; }
  if (h->routed && !rings_empty(h))
    FAIL(ESVO_ERR_STATE, "the handle routes events by row and holds staged events: esvo_reset before changing its band");
  h->routed = false;  // (esvo_shard_set_routing follows)
  h->dp.ev_shard = shard;
  h->dp.ev_nshards = n_shards;
  h->dp.band_y0 = row_begin;
  h->dp.band_y1 = row_end;
  set_compute_band(h);
  h->sharded = !(row_begin == 0 && row_end == h->H) || n_shards > 1;
  if (h->sharded && !h->d_rank_kept) {  // exchange blocks, sized for any rank count up to SHARD_MAX_RANKS (lazily: unsharded handles never pay)
    const size_t E = h->max_ev, R = esvo_context::SHARD_MAX_RANKS, WP = sizeof(DevPoint) / 8;
    HIPCHK(hipSetDevice(h->device));
    auto alloc = [&](auto** p, size_t bytes) { return hipMalloc(reinterpret_cast<void**>(p), bytes) == hipSuccess; };
    // (d_rank_kept, the guard above, is allocated LAST: a failure in the chain frees what came before and leaves the guard null)
    if (!alloc(&h->d_codes_send, (E + 7) / 8 * 8) || !alloc(&h->d_codes_all, E + 8 * R) || !alloc(&h->d_pts_send, 8 * (1 + E * WP)) ||
        !alloc(&h->d_pts_all, 8 * (R + (E + R) * WP)) || !alloc(&h->d_rank_kept, sizeof(u32) * R)) {
      (void)hipGetLastError();
      free_shard_blocks(h);
      h->sharded = false;
      h->dp.ev_shard = 0; h->dp.ev_nshards = 1; h->dp.band_y0 = 0; h->dp.band_y1 = h->H;
      set_compute_band(h);
      FAIL(ESVO_ERR_CAPACITY, "out of device memory for the shard exchange blocks");
    }
    HIPCHK(hipMemset(h->d_rank_kept, 0, sizeof(u32) * R));
  }
  return ESVO_OK;
}

int esvo_shard_set_routing(esvo_handle h, int mode, int ts_halo_rows) {
  if (!h || (mode != ESVO_ROUTE_BROADCAST && mode != ESVO_ROUTE_Y_RECT)) return ESVO_ERR_INVALID_ARG;
  API_LOCK(h);
  { int rcp = flush_pending_tick(h); if (rcp) return rcp; }
  if (!h->sharded) FAIL(ESVO_ERR_STATE, "call esvo_shard_set_band first");
  if (!rings_empty(h)) FAIL(ESVO_ERR_STATE, "events are already staged: choose the routing before the first esvo_ts_push_events (or esvo_reset)");
  if (mode == ESVO_ROUTE_BROADCAST) { h->routed = false; return ESVO_OK; }
  const esvo_params_t& p = h->prm;
  if (h->tsq_len) FAIL(ESVO_ERR_UNSUPPORTED, "per-pixel event queues (max_event_queue_len) are not routed: use ESVO_ROUTE_BROADCAST");
  if (p.bm_updown) FAIL(ESVO_ERR_UNSUPPORTED, "up-down stereo searches along y, across the bands: use ESVO_ROUTE_BROADCAST");
  const int H = h->H, W = h->W;
  const int hy = (p.patch_size_y - 1) / 2;
  int halo = ts_halo_rows < 0 ? 24 : ts_halo_rows;
  // block matching reads the band + hy rows; the refinement's blocks reach one row further before any motion
  if (halo < hy + 2) FAIL(ESVO_ERR_INVALID_ARG, "ts_halo_rows must be at least patch_size_Y / 2 + 2");
  // rows of the observation pair that must hold data, in whole 4-row tiles of the blur; the Time-Surface rows they are made
  // from (+ 2 under SmoothTimeSurface: GaussianBlurTS(5)), in whole tiles of the render kernel
  const int o0 = std::max(0, h->dp.band_y0 - halo) / 4 * 4;
  const int o1 = std::min(H, (std::min(H, h->dp.band_y1 + halo) + 3) / 4 * 4);
  const int pad = p.smooth_time_surface ? 2 : 0;
  const int r0 = std::max(0, o0 - pad) / TS_TILE_ROWS * TS_TILE_ROWS;
  const int r1 = std::min(H, (std::min(H, o1 + pad) + TS_TILE_ROWS - 1) / TS_TILE_ROWS * TS_TILE_ROWS);
  const int k = std::max(0, p.median_blur_kernel_size);
  for (int cam = 0; cam < 2; ++cam) {  // raw rows the remap taps of [r0, r1) reach, + the median's ring
    int lo = H, hi = -1;
    for (int y = r0; y < r1; ++y) { lo = std::min(lo, h->fix_row_lo[cam][y]); hi = std::max(hi, h->fix_row_hi[cam][y]); }
    h->sband_y0[cam] = hi < lo ? 0 : std::max(0, lo - k);
    h->sband_y1[cam] = hi < lo ? 0 : std::min(H, hi + k + 1);
  }
  h->keep_px.assign((size_t)W * H, 0);
  const float* lut = h->h_rect_lut[0].data();
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      uint8_t f = (y >= h->sband_y0[0] && y < h->sband_y1[0]) ? 1 : 0;
      const int yb = (int)std::floor((double)lut[2 * ((size_t)y * W + x) + 1]);  // kernels_bm.hip: the rank that owns floor(y_rect)
      if (yb >= h->dp.band_y0 && yb < h->dp.band_y1) f |= 2;
      // Denoising: the rank decides the mask's verdict for the events whose RAW row is in its band; the 3 x 3 median reads the
      // selected events of one more row on either side (routed_denoise_begin)
      if (p.denoising && y >= h->dp.band_y0 - 1 && y < h->dp.band_y1 + 1) f |= 4;
      h->keep_px[(size_t)y * W + x] = f;
    }
  HIPCHK(hipSetDevice(h->device));
  if (!h->d_ring_gidx) HIPCHK(hipMalloc(reinterpret_cast<void**>(&h->d_ring_gidx), sizeof(u32) * h->ring_cap));
  {  // exchange 1 spans the whole tick in this mode: n_shards blocks of two bits per slot
    const size_t need = (size_t)h->dp.ev_nshards * shard_codes_block_routed(h->max_ev);
    if (need > (size_t)h->max_ev + 8 * esvo_context::SHARD_MAX_RANKS) {
      HIPCHK(hipStreamSynchronize(h->stream));
      uint8_t* d_new = nullptr;
      if (hipMalloc(reinterpret_cast<void**>(&d_new), need) != hipSuccess) { (void)hipGetLastError(); FAIL(ESVO_ERR_CAPACITY, "out of device memory for the routed exchange blocks"); }
      (void)hipFree(h->d_codes_all);
      h->d_codes_all = d_new;
    }
  }
  h->ts_halo = halo;
  h->oband_y0 = o0; h->oband_y1 = o1;
  h->rband_y0 = r0; h->rband_y1 = r1;
  h->routed = true;
  return ESVO_OK;
}

int esvo_shard_get_rows(esvo_handle h, int render_rows[2], int observation_rows[2], int source_rows_left[2], int source_rows_right[2]) {
  if (!h) return ESVO_ERR_INVALID_ARG;
  API_LOCK(h);
  const bool r = h->routed;
  if (render_rows) { render_rows[0] = r ? h->rband_y0 : 0; render_rows[1] = r ? h->rband_y1 : h->H; }
  if (observation_rows) { observation_rows[0] = r ? h->oband_y0 : 0; observation_rows[1] = r ? h->oband_y1 : h->H; }
  if (source_rows_left) { source_rows_left[0] = r ? h->sband_y0[0] : 0; source_rows_left[1] = r ? h->sband_y1[0] : h->H; }
  if (source_rows_right) { source_rows_right[0] = r ? h->sband_y0[1] : 0; source_rows_right[1] = r ? h->sband_y1[1] : h->H; }
  return ESVO_OK;
}

int esvo_shard_exchange(esvo_handle h, void** d_send, void** d_recv, size_t* block_bytes) {
  if (!h || !d_send || !d_recv || !block_bytes) return ESVO_ERR_INVALID_ARG;
  API_LOCK(h);
  *d_send = h->xchg_send;
  *d_recv = h->xchg_recv;
  *block_bytes = h->xchg_block;
  return ESVO_OK;
}

}  // extern "C"
