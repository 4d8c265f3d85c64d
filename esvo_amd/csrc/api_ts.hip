// api_ts.hip — event ingest and Time-Surface render (see context.hpp).
#include "context.hpp"

namespace esvo_host {

// Time-Surface kernel timings of the last render of a camera (-1: both); the caller knows their events are complete
void collect_ts_timing(esvo_context* h, int only) {
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
      h->stats.sum_ms_kernel[7] += (cam == 0 && h->ts_pair_sample) ? 2 : 1;
    }
    if (cam == 0) h->ts_pair_sample = false;
  }
}

// A copy enqueued by esvo_ts_push_events_async may still be in flight on the ingest stream: whatever the front stream launches
// next that reads the camera's ring (scatter, block matching, the SGM points) waits for it on the DEVICE.  Caller holds mu_ring.
void ingest_fence(esvo_context* h, int cam) {
  if (!h->ingest_pending[cam]) return;
  hipStreamWaitEvent(h->stream, h->evt_ingest[cam], 0);
  h->ingest_pending[cam] = false;
}

// Tick-interleaved multi-GPU operation (esvo_comm_tick on a tick another rank maps): every staged event with ts < t_ns that is
// not in the SAE yet is scattered NOW -- on the front stream, which is idle between this rank's block matching and its next
// own render, i.e. beside the own tick's LM launch -- so that the next own render finds only its own tick's events left to
// scatter.  Same bookkeeping as the scatter of a render; stream order keeps it behind the last render and before the next.
// (A side stream was tried first: it ended up in the hardware queue of the exchange stream, behind that stream's wait for the
//  LM launch, and the next own render behind it -- 1.66 instead of 1.3 ms per round.)
int ts_scatter_ahead(esvo_context* h, uint64_t t_ns) {
  if (h->tsq_len || h->routed) return ESVO_OK;  // (queue mode inserts and derives the SAE at render time)
  std::lock_guard<std::mutex> lr(h->mu_ring);
  TsScatterSegs g;
  int n_seg = 0;
  for (int cam = 0; cam < 2; ++cam) {
    const auto& tsq = h->ts_host[cam];
    const size_t k = std::lower_bound(tsq.begin(), tsq.end(), (u64)t_ns) - tsq.begin();
    const u64 upto = h->ring_base[cam] + k;
    u64 a = h->scattered[cam];
    if (upto <= a) continue;
    ingest_fence(h, cam);
    h->scatter_pending_lo[cam] = std::min(h->scatter_pending_lo[cam], a);
    h->scatter_seq++;
    h->stats.events_scattered[cam] += upto - a;
    while (a < upto) {
      const u64 slot = a % h->ring_cap;
      const u64 cnt = std::min<u64>(upto - a, h->ring_cap - slot);
      if (n_seg == 4) { launch_ts_scatter_segs(g, n_seg, h->W, h->H, h->stream); n_seg = 0; }
      g.ev[n_seg] = h->d_ring[cam] + slot; g.n[n_seg] = (size_t)cnt; g.sae[n_seg] = h->d_sae[cam];
      ++n_seg;
      a += cnt;
    }
    h->scattered[cam] = upto;
  }
  if (!n_seg) return ESVO_OK;
  launch_ts_scatter_segs(g, n_seg, h->W, h->H, h->stream);
  HIPCHK(hipGetLastError());
  return ESVO_OK;
}

// The resident surface of `cam` is about to be overwritten on the front stream (caller holds mu_ts): a tracker thread's
// esvo_track_set_current may still be reading the left one on the tracker stream.
void resident_write_begin(esvo_context* h, int cam) {
  if (cam == 0 && h->trk_read_pending) {
    hipStreamWaitEvent(h->stream, h->evt_trk_read, 0);
    h->trk_read_pending = false;
  }
}

// Queue mode (max_event_queue_len > 0; caller holds mu_ring): every staged event of `cam` that is not in its pixel's queue yet
// goes in -- future ones included, as eventsCallback inserts them on arrival (TimeSurface.cpp:403-425) -- and the SAE word
// of every pixel is derived for THIS render time (getMostRecentEventBeforeT); the render kernels read the SAE as ever.
void queue_prepare(esvo_context* h, int cam, u64 t_ns) {
  u64 a = h->scattered[cam];
  const u64 upto = h->ring_next[cam];
  if (upto > a) {
    h->scatter_pending_lo[cam] = std::min(h->scatter_pending_lo[cam], a);
    h->scatter_seq++;
    h->stats.events_scattered[cam] += upto - a;
    while (a < upto) {  // rounds of at most TSQ_ROUND events: the overflow list can hold a whole round
      TsQueueArgs g{};
      g.q = h->d_tsq[cam]; g.L = h->tsq_len;
      g.tcount = h->d_tsq_tcount; g.tlist = h->d_tsq_tlist; g.tcap = h->tsq_tcap;
      g.over = h->d_tsq_over; g.over_count = h->d_tsq_over_count; g.over_cap = (u32)esvo_context::TSQ_ROUND;
      u64 left = std::min<u64>(upto - a, esvo_context::TSQ_ROUND);
      for (int k = 0; k < 2 && left; ++k) {
        const u64 slot = a % h->ring_cap;
        const u64 cnt = std::min<u64>(left, h->ring_cap - slot);
        g.ev[k] = h->d_ring[cam] + slot; g.n[k] = (size_t)cnt;
        a += cnt; left -= cnt;
      }
      launch_tsq_insert(g, h->W, h->H, h->stream);
    }
    h->scattered[cam] = upto;
  }
  if (!h->tsq_dup[cam].empty()) {  // what eventsCallback inserted in place of late events: copies of the then-newest event (push_unsorted)
    auto& dup = h->tsq_dup[cam];
    if (dup.size() > h->tsq_dup_cap) {
      hipStreamSynchronize(h->stream);
      if (h->d_tsq_dup) hipFree(h->d_tsq_dup);
      h->tsq_dup_cap = std::max<size_t>(dup.size(), 4096);
      if (hipMalloc(reinterpret_cast<void**>(&h->d_tsq_dup), sizeof(esvo_event_t) * h->tsq_dup_cap) != hipSuccess) { h->d_tsq_dup = nullptr; h->tsq_dup_cap = 0; }
    }
    if (h->d_tsq_dup) {
      for (size_t a0 = 0; a0 < dup.size(); a0 += esvo_context::TSQ_ROUND) {
        const size_t cnt = std::min<size_t>(dup.size() - a0, esvo_context::TSQ_ROUND);
        hipMemcpyAsync(h->d_tsq_dup, dup.data() + a0, sizeof(esvo_event_t) * cnt, hipMemcpyHostToDevice, h->stream);
        hipStreamSynchronize(h->stream);  // (pageable source; the rare path)
        TsQueueArgs g{};
        g.q = h->d_tsq[cam]; g.L = h->tsq_len;
        g.tcount = h->d_tsq_tcount; g.tlist = h->d_tsq_tlist; g.tcap = h->tsq_tcap;
        g.over = h->d_tsq_over; g.over_count = h->d_tsq_over_count; g.over_cap = (u32)esvo_context::TSQ_ROUND;
        g.ev[0] = h->d_tsq_dup; g.n[0] = cnt;
        launch_tsq_insert(g, h->W, h->H, h->stream);
      }
    }
    dup.clear();
  }
  launch_tsq_view(h->d_tsq[cam], h->tsq_len, h->W, h->H, t_ns, h->d_sae[cam], h->stream);
}

// Both cameras' surfaces at t_ns with one launch per kernel (scatter segments, decay, median + remap): what two
// esvo_ts_render calls do, in four launches less.  obs_out[cam] (may be null) receives a second copy of the surface.
int ts_render_pair(esvo_context* h, uint64_t t_ns, uint8_t* const obs_out[2]) {
  // stage timings of a render that runs alone are sampled (context.hpp, lat_ticks): each event costs the queue ~5 us
  const bool timed = esvo_stage_timed(h);
  {
    std::lock_guard<std::mutex> lr(h->mu_ring);  // what a pusher on another thread reads and writes (context.hpp)
    ingest_fence(h, 0);
    ingest_fence(h, 1);
    u64 upto[2];
    for (int cam = 0; cam < 2 && !h->tsq_len; ++cam) {
      const auto& tsq = h->ts_host[cam];
      const size_t k = std::lower_bound(tsq.begin(), tsq.end(), (u64)t_ns) - tsq.begin();
      upto[cam] = h->ring_base[cam] + k;
      if (upto[cam] < h->scattered[cam])
        FAIL(ESVO_ERR_STATE, "esvo_ts_render: t_ns precedes events of an earlier render (render times must not decrease; esvo_reset to replay)");
    }
    for (int cam = 0; cam < 2; ++cam)
      if (h->ts_timing_pending[cam] && hipEventQuery(h->evt[EV_R1 + cam * EV_TS_STRIDE]) == hipSuccess) collect_ts_timing(h, cam);
    if (timed) hipEventRecord(h->evt[EV_SC0], h->stream);
    TsScatterSegs g;
    int n_seg = 0;
    if (h->tsq_len) { queue_prepare(h, 0, (u64)t_ns); queue_prepare(h, 1, (u64)t_ns); }
    for (int cam = 0; cam < 2 && !h->tsq_len; ++cam) {
      u64 a = h->scattered[cam];
      if (upto[cam] <= a) continue;
      h->scatter_pending_lo[cam] = std::min(h->scatter_pending_lo[cam], a);
      h->scatter_seq++;
      h->stats.events_scattered[cam] += upto[cam] - a;
      while (a < upto[cam]) {
        const u64 slot = a % h->ring_cap;
        const u64 cnt = std::min<u64>(upto[cam] - a, h->ring_cap - slot);
        if (n_seg == 4) {  // a range longer than the ring (cannot happen: staging refuses it) -- flush and go on
          launch_ts_scatter_segs(g, n_seg, h->W, h->H, h->stream);
          n_seg = 0;
        }
        g.ev[n_seg] = h->d_ring[cam] + slot; g.n[n_seg] = (size_t)cnt; g.sae[n_seg] = h->d_sae[cam];
        ++n_seg;
        a += cnt;
      }
      h->scattered[cam] = upto[cam];
    }
    launch_ts_scatter_segs(g, n_seg, h->W, h->H, h->stream);
    if (timed) hipEventRecord(h->evt[EV_SC1], h->stream);
  }
  TsPair c;
  for (int cam = 0; cam < 2; ++cam) {
    c.sae[cam] = h->d_sae[cam]; c.fixmap[cam] = h->d_fixmap[cam]; c.out[cam] = h->d_ts[cam];
    c.out2[cam] = obs_out ? obs_out[cam] : nullptr;
  }
  c.raw[0] = h->d_raw; c.raw[1] = h->d_raw1;
  {
    std::lock_guard<std::mutex> lt(h->mu_ts);
    resident_write_begin(h, 0);
    // (a routed band handle renders the rows of its band + halo only: h->rband_*, whole tiles)
    launch_ts_render_pair(c, h->W, h->H, (u64)t_ns, h->prm.decay_ms / 1000.0, h->prm.ignore_polarity, h->prm.median_blur_kernel_size,
                          h->stream, h->routed ? h->rband_y0 : 0, h->routed ? h->rband_y1 : -1);
    if (timed || h->trk_used.load()) hipEventRecord(h->evt[EV_R1], h->stream);  // (also what esvo_track_set_current waits for)
    HIPCHK(hipGetLastError());
    h->ts_valid[0] = h->ts_valid[1] = true;
  }
  h->ts_timing_pending[0] = timed;
  h->ts_pair_sample = timed;
  h->stats.ts_frames[0]++;
  h->stats.ts_frames[1]++;
  return ESVO_OK;
}

}  // namespace esvo_host

// ---- Time Surface ---------------------------------------------------------------------------------
namespace {
// ---- ingest protocol (INGEST group: may run on another thread than the renders and ticks of the same handle) ----------
// Staging runs on its own stream.  A pusher (one per camera at a time, mu_push) works in three steps:
//   begin   (mu_ring) order + capacity checks, then it RESERVES [ring_next, ring_next + n): from here on selections are
//           validated against ring_reserved, so no new kernel can be pointed at the slots about to be overwritten;
//   drain   the slots hold events older than ring_cap; kernels already enqueued read at most the max_ev events before the
//           last selection point and the not yet completed scatter ranges, so only a (nearly) full ring needs the front
//           stream drained -- done WITHOUT holding mu_ring;
//   copy + commit  host-to-device on the ingest stream (no lock), then (mu_ring) the stamps and ring_next are published.
struct PushTicket {
  u64 slot = 0;
  bool drain = false;
  u64 seq = 0;
};
int push_begin(esvo_context* h, int cam, size_t n, u64 t_first, PushTicket& tk) {
  std::lock_guard<std::mutex> lr(h->mu_ring);
  const auto& tsq = h->ts_host[cam];
  if (!tsq.empty() && t_first < tsq.back()) FAIL(ESVO_ERR_INVALID_ARG, "events must be sorted by time stamp (SURVEY Appendix A-1)");
  // the ring must not overwrite events that are not yet scattered into the SAE
  if (h->ring_next[cam] + n - h->scattered[cam] > h->ring_cap)
    FAIL(ESVO_ERR_CAPACITY, "event ring full: render (scatter) before staging more events");
  tk.slot = h->ring_next[cam] % h->ring_cap;
  h->ring_reserved[cam] = h->ring_next[cam] + n;
  if (h->ring_reserved[cam] > h->ring_cap) {
    const u64 evict_end = h->ring_reserved[cam] - h->ring_cap;  // first absolute index that survives
    u64 oldest_read = h->scatter_pending_lo[cam];               // scatter kernels enqueued since the last drain
    if (cam == 0) {  // the selections of the (up to) two ticks whose front stages may not have completed: each reads at most
                     // max_ev events up to its selection point
      const u64 sel_lo = std::min(h->sh_first, h->sh_first_prev);
      oldest_read = std::min(oldest_read, sel_lo > (u64)h->max_ev ? sel_lo - (u64)h->max_ev : 0);
    }
    tk.drain = evict_end > oldest_read;
    tk.seq = h->scatter_seq;
  }
  return ESVO_OK;
}
void push_abort(esvo_context* h, int cam) {
  std::lock_guard<std::mutex> lr(h->mu_ring);
  h->ring_reserved[cam] = h->ring_next[cam];
}
int push_drain(esvo_context* h, int cam, const PushTicket& tk) {
  if (!tk.drain) return ESVO_OK;
  if (hipStreamSynchronize(h->stream) != hipSuccess) { push_abort(h, cam); FAIL(ESVO_ERR_HIP, "draining the front stream failed"); }
  std::lock_guard<std::mutex> lr(h->mu_ring);
  if (h->scatter_seq == tk.seq) h->scatter_pending_lo[0] = h->scatter_pending_lo[1] = ~0ull;  // nothing enqueued meanwhile
  return ESVO_OK;
}
template <typename StampFn>
void push_commit(esvo_context* h, int cam, size_t n, StampFn stamp, bool copy_in_flight = false) {
  std::lock_guard<std::mutex> lr(h->mu_ring);
  if (copy_in_flight) h->ingest_pending[cam] = true;
  auto& tsq = h->ts_host[cam];
  for (size_t i = 0; i < n; ++i) tsq.push_back(stamp(i));
  h->ring_next[cam] += n;
  h->ring_reserved[cam] = h->ring_next[cam];
  while (tsq.size() > h->ring_cap) { tsq.pop_front(); h->ring_base[cam]++; }
  h->stats.events_staged[cam] += n;
}
// ---- routed band mode (esvo_shard_set_routing): the packet is filtered on the host -----------------------------------------------
// Of the n events of a packet the rank keeps those its Time Surfaces need (raw row inside the camera's source rows) and, for the
// left camera, those it block-matches (floor(y_rect) inside the band): keep_px / sband_*.  Only they are copied to the device
// ring; each kept left event carries its index in the GLOBAL left sequence (d_ring_gidx), because the reference's event selection
// (esvo_Mapping.cpp:562-574) and its thread-stride order are defined on the whole stream -- whose stamps stay on the host
// (glob_ts).  Synchronous (the staging buffer is the library's): esvo_ts_push_events_async behaves like esvo_ts_push_events.
// Caller holds mu_push[cam]; the packet is sorted.
template <typename GetEv>
int push_routed(esvo_context* h, int cam, size_t n, GetEv get) {
  auto stamp_of = [](const esvo_event_t& e) { return (u64)e.sec * 1000000000ull + e.nsec; };
  u64 g0 = 0;
  {
    std::lock_guard<std::mutex> lr(h->mu_ring);
    if (h->last_stamp[cam] && stamp_of(get(0)) < h->last_stamp[cam])
      FAIL(ESVO_ERR_INVALID_ARG, "events must be sorted by time stamp (SURVEY Appendix A-1)");
    g0 = h->glob_base + h->glob_ts.size();
  }
  if (n > h->route_cap[cam]) {  // (idle: every routed push ends with a wait for its copies)
    const size_t cap = std::max<size_t>(n, (size_t)1 << 16);
    if (h->h_route_ev[cam]) { hipHostFree(h->h_route_ev[cam]); h->h_route_ev[cam] = nullptr; h->route_cap[cam] = 0; }
    if (cam == 0 && h->h_route_gidx) { hipHostFree(h->h_route_gidx); h->h_route_gidx = nullptr; }
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&h->h_route_ev[cam]), sizeof(esvo_event_t) * cap, hipHostMallocDefault));
    if (cam == 0) HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&h->h_route_gidx), sizeof(u32) * cap, hipHostMallocDefault));
    h->route_cap[cam] = cap;
  }
  esvo_event_t* kept = h->h_route_ev[cam];
  std::vector<u64> kept_ts, kept_gl, all_ts;
  std::vector<uint8_t> kept_own;
  kept_ts.reserve(n / 2 + 16);
  if (cam == 0) { kept_gl.reserve(n / 2 + 16); all_ts.resize(n); }
  size_t m = 0;
  const int W = h->W, H = h->H, sy0 = h->sband_y0[cam], sy1 = h->sband_y1[cam];
  const uint8_t* keep_px = h->keep_px.data();
  u64 last = 0;
  for (size_t i = 0; i < n; ++i) {
    const esvo_event_t e = get(i);
    const u64 t = stamp_of(e);
    last = t;
    if (cam == 0) all_ts[i] = t;
    bool keep = false;
    if ((int)e.x < W && (int)e.y < H) keep = cam == 0 ? keep_px[(size_t)e.y * W + e.x] != 0 : ((int)e.y >= sy0 && (int)e.y < sy1);
    if (!keep) continue;
    kept[m] = e;
    kept[m].polarity = e.polarity ? 1 : 0;   // (bit 7 of the byte is the library's: EV_LATE)
    kept[m]._pad[0] = kept[m]._pad[1] = kept[m]._pad[2] = 0;
    if (cam == 0) { h->h_route_gidx[m] = (u32)(g0 + i); kept_gl.push_back(g0 + i); kept_own.push_back((keep_px[(size_t)e.y * W + e.x] >> 1) & 1); }
    kept_ts.push_back(t);
    ++m;
  }
  if (m) {
    PushTicket tk;
    { int rc = push_begin(h, cam, m, kept_ts[0], tk); if (rc) return rc; }
    { int rc = push_drain(h, cam, tk); if (rc) return rc; }
    const size_t first = (size_t)std::min<u64>(m, h->ring_cap - tk.slot);
    hipError_t e = hipMemcpyAsync(h->d_ring[cam] + tk.slot, kept, sizeof(esvo_event_t) * first, hipMemcpyHostToDevice, h->stream_i);
    if (e == hipSuccess && first < m)
      e = hipMemcpyAsync(h->d_ring[cam], kept + first, sizeof(esvo_event_t) * (m - first), hipMemcpyHostToDevice, h->stream_i);
    if (e == hipSuccess && cam == 0) {
      e = hipMemcpyAsync(h->d_ring_gidx + tk.slot, h->h_route_gidx, sizeof(u32) * first, hipMemcpyHostToDevice, h->stream_i);
      if (e == hipSuccess && first < m)
        e = hipMemcpyAsync(h->d_ring_gidx, h->h_route_gidx + first, sizeof(u32) * (m - first), hipMemcpyHostToDevice, h->stream_i);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream_i);
    if (e != hipSuccess) { push_abort(h, cam); FAIL(ESVO_ERR_HIP, std::string("staging the routed events failed: ") + hipGetErrorString(e)); }
  }
  std::lock_guard<std::mutex> lr(h->mu_ring);
  auto& tsq = h->ts_host[cam];
  for (size_t i = 0; i < m; ++i) tsq.push_back(kept_ts[i]);
  h->ring_next[cam] += m;
  h->ring_reserved[cam] = h->ring_next[cam];
  if (cam == 0) {
    for (size_t i = 0; i < m; ++i) { h->kept_g.push_back(kept_gl[i]); h->own_before.push_back(h->own_total); h->own_total += kept_own[i]; }
    h->glob_ts.insert(h->glob_ts.end(), all_ts.begin(), all_ts.end());   // (one splice: the tick thread's selection takes this lock too)
    if (h->glob_ts.size() > h->ring_cap) {
      const size_t drop = h->glob_ts.size() - h->ring_cap;
      h->glob_ts.erase(h->glob_ts.begin(), h->glob_ts.begin() + (std::ptrdiff_t)drop);
      h->glob_base += drop;
    }
  }
  while (tsq.size() > h->ring_cap) {
    tsq.pop_front();
    h->ring_base[cam]++;
    if (cam == 0) { h->kept_g.pop_front(); h->own_before.pop_front(); }
  }
  h->last_stamp[cam] = last;
  h->stats.events_staged[cam] += m;
  return ESVO_OK;
}

// ---- out-of-order input ---------------------------------------------------------------------------------------------------
// The reference sorts on arrival: both nodes insertion-sort every event into their queue (TimeSurface.cpp:412-420,
// esvo_Mapping.cpp:692-702: `while (EQ[i].ts > e.ts)` -- strict, so equal stamps keep their arrival order: a STABLE sort).
//   mapper side   the ring holds the events in that sorted order; a packet that reaches back into already staged events is merged
//                 into the ring's tail on the device (ts_merge_kernel).  (esvo_Mapping::eventsCallback also RESETS the whole mapper
//                 when a message's FIRST stamp lies before its queue's newest, :678-687: that decision stays with the caller --
//                 stats.late_events tells it -- the library never throws a map away on its own.)
//   Time Surface  eventsCallback then inserts events_.back() -- the NEWEST event, not the one that just arrived (:421-422, SURVEY
//                 Appendix A-1).  For an event that is late on arrival (stamp < the newest stamp seen before it) that is a second
//                 copy of the newest event and the late event itself never enters the per-pixel queues: the late event is marked
//                 (EV_LATE, common.hpp) and skipped by the scatter; the second copy is a no-op for the one-stamp-per-pixel SAE and,
//                 in queue mode, is inserted with the next batch (tsq_dup).
// The rare path: it takes the mapper group's lock and drains the front stream before it moves anything.
template <typename GetEv>
int push_unsorted(esvo_context* h, int cam, size_t n, GetEv get) {
  std::lock_guard<std::recursive_mutex> la(h->mu_api);  // (before mu_push: the order esvo_reset takes them in)
  std::lock_guard<std::mutex> lp(h->mu_push[cam]);
  HIPCHK(hipSetDevice(h->device));
  if (h->routed) FAIL(ESVO_ERR_UNSUPPORTED, "out-of-order packets on a row-routed band handle: sort them first (or use ESVO_ROUTE_BROADCAST)");
  auto stamp_of = [](const esvo_event_t& e) { return (u64)e.sec * 1000000000ull + e.nsec; };
  std::vector<esvo_event_t> S(n);
  std::vector<u64> t(n);
  std::vector<u32> order(n);
  u64 newest = 0;
  esvo_event_t newest_ev;
  std::memset(&newest_ev, 0, sizeof(newest_ev));
  size_t K = 0, pos_rel = 0;
  // nothing enqueued may still read the ring's tail (block matching of a pending tick, a scatter) while it moves
  if (hipStreamSynchronize(h->stream) != hipSuccess || hipStreamSynchronize(h->stream_i) != hipSuccess) FAIL(ESVO_ERR_HIP, "draining the streams failed");
  bool have_newest_ev = false;
  {
    u64 last_idx = 0;
    bool any = false;
    {
      std::lock_guard<std::mutex> lr(h->mu_ring);
      if (!h->ts_host[cam].empty()) { newest = h->ts_host[cam].back(); last_idx = h->ring_next[cam] - 1; any = true; }
    }
    if (any && h->tsq_len) {  // queue mode: the record of the newest staged event (the ring's last: sorted), whose copies stand in for late events
      HIPCHK(hipMemcpy(&newest_ev, h->d_ring[cam] + last_idx % h->ring_cap, sizeof(esvo_event_t), hipMemcpyDeviceToHost));
      have_newest_ev = true;
    }
  }
  // late on arrival: below the running maximum of everything that arrived before (a tie is NOT late: it becomes back())
  size_t n_late = 0;
  std::vector<esvo_event_t> dups;
  for (size_t i = 0; i < n; ++i) {
    esvo_event_t e = get(i);
    t[i] = stamp_of(e);
    order[i] = (u32)i;
    e.polarity = e.polarity ? 1 : 0;
    e._pad[0] = e._pad[1] = e._pad[2] = 0;
    if (t[i] < newest) {
      e.polarity |= (uint8_t)EV_LATE;   // the library's mark: the byte and the padding magic together (common.hpp, ev_is_late)
      e._pad[0] = (uint8_t)(EV_LATE_PAD & 0xffu); e._pad[1] = (uint8_t)((EV_LATE_PAD >> 8) & 0xffu); e._pad[2] = (uint8_t)((EV_LATE_PAD >> 16) & 0xffu);
      ++n_late;
      if (h->tsq_len && have_newest_ev) dups.push_back(newest_ev);
    } else {
      newest = t[i];
      newest_ev = e;
      have_newest_ev = true;
    }
    S[i] = e;
  }
  std::stable_sort(order.begin(), order.end(), [&](u32 a, u32 b) { return t[a] < t[b]; });
  std::vector<esvo_event_t> P(n);
  std::vector<u64> tp(n);
  for (size_t j = 0; j < n; ++j) { P[j] = S[order[j]]; tp[j] = t[order[j]]; }
  // the staged events the packet reaches back into: those with a stamp ABOVE the packet's smallest (equal ones arrived earlier: they stay in front)
  std::vector<u64> ts_tail;
  u64 pos_abs = 0, ring_next = 0;
  {
    std::lock_guard<std::mutex> lr(h->mu_ring);
    const auto& tsq = h->ts_host[cam];
    pos_rel = std::upper_bound(tsq.begin(), tsq.end(), tp[0]) - tsq.begin();
    K = tsq.size() - pos_rel;
    pos_abs = h->ring_base[cam] + pos_rel;
    ring_next = h->ring_next[cam];
    if (h->ring_next[cam] + n - h->scattered[cam] > h->ring_cap) FAIL(ESVO_ERR_CAPACITY, "event ring full: render (scatter) before staging more events");
    if (K > ((size_t)1 << 22)) FAIL(ESVO_ERR_CAPACITY, "packet reaches back over more than 4 M staged events");
    // the merged tail (K staged + n new events) is written back into the ring: positions j and j + ring_cap would share a slot
    if (K + n > h->ring_cap)
      FAIL(ESVO_ERR_CAPACITY, "an out-of-order packet reaches back over more staged events than the event ring holds with it (a clock jump "
                              "or a looped bag: the reference resets its mapper here, esvo_Mapping.cpp:680-687 -- esvo_reset and re-stage)");
    ts_tail.assign(tsq.begin() + pos_rel, tsq.end());
    h->ring_reserved[cam] = ring_next + n;
  }
  auto grow = [&](auto** p, size_t& cap, size_t need, size_t elem) {
    if (need <= cap) return true;
    if (*p) (void)hipFree(*p);
    *p = nullptr; cap = 0;
    const size_t c = std::max<size_t>(need + need / 2, 4096);
    if (hipMalloc(reinterpret_cast<void**>(p), c * elem) != hipSuccess) { (void)hipGetLastError(); return false; }
    cap = c;
    return true;
  };
  if (!grow(&h->d_merge_a, h->merge_cap_a, K, sizeof(esvo_event_t)) || !grow(&h->d_merge_b, h->merge_cap_b, n, sizeof(esvo_event_t)) ||
      !grow(&h->d_merge_plan, h->merge_cap_plan, K + n, sizeof(u32))) {
    push_abort(h, cam);
    FAIL(ESVO_ERR_CAPACITY, "out of device memory for the out-of-order merge");
  }
  // plan of the merged tail: staged first on equal stamps
  std::vector<u32> plan(K + n);
  std::vector<u64> merged(K + n);
  size_t a = 0, b = 0, n_before_scattered = 0;
  u64 scattered_rel = 0;
  {
    std::lock_guard<std::mutex> lr(h->mu_ring);
    scattered_rel = h->scattered[cam] > pos_abs ? h->scattered[cam] - pos_abs : 0;  // staged events of the tail already in the SAE
  }
  for (size_t j = 0; j < K + n; ++j) {
    if (b >= n || (a < K && ts_tail[a] <= tp[b])) { plan[j] = (u32)a; merged[j] = ts_tail[a]; ++a; }
    else {
      plan[j] = 0x80000000u | (u32)b; merged[j] = tp[b];
      if (a < scattered_rel) ++n_before_scattered;  // lands among events that are scattered already: it is late (never scattered)
      ++b;
    }
  }
  hipError_t e = hipSuccess;
  const u64 slot0 = pos_abs % h->ring_cap;
  if (K) {  // a copy of the tail (it may wrap)
    const size_t first = (size_t)std::min<u64>(K, h->ring_cap - slot0);
    e = hipMemcpyAsync(h->d_merge_a, h->d_ring[cam] + slot0, sizeof(esvo_event_t) * first, hipMemcpyDeviceToDevice, h->stream_i);
    if (e == hipSuccess && first < K)
      e = hipMemcpyAsync(h->d_merge_a + first, h->d_ring[cam], sizeof(esvo_event_t) * (K - first), hipMemcpyDeviceToDevice, h->stream_i);
  }
  if (e == hipSuccess) e = hipMemcpyAsync(h->d_merge_b, P.data(), sizeof(esvo_event_t) * n, hipMemcpyHostToDevice, h->stream_i);
  if (e == hipSuccess) e = hipMemcpyAsync(h->d_merge_plan, plan.data(), sizeof(u32) * (K + n), hipMemcpyHostToDevice, h->stream_i);
  if (e == hipSuccess) {
    launch_ts_merge(h->d_merge_a, h->d_merge_b, h->d_merge_plan, K + n, h->d_ring[cam], slot0, h->ring_cap, h->stream_i);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream_i);
  if (e != hipSuccess) { push_abort(h, cam); FAIL(ESVO_ERR_HIP, std::string("out-of-order merge failed: ") + hipGetErrorString(e)); }
  std::lock_guard<std::mutex> lr(h->mu_ring);
  auto& tsq = h->ts_host[cam];
  tsq.erase(tsq.begin() + pos_rel, tsq.end());
  for (u64 v : merged) tsq.push_back(v);
  h->ring_next[cam] += n;
  h->ring_reserved[cam] = h->ring_next[cam];
  if (h->scattered[cam] > pos_abs) h->scattered[cam] += n_before_scattered;
  while (tsq.size() > h->ring_cap) { tsq.pop_front(); h->ring_base[cam]++; }
  h->stats.events_staged[cam] += n;
  h->stats.late_events[cam] += n_late;
  for (const esvo_event_t& d : dups) h->tsq_dup[cam].push_back(d);
  return ESVO_OK;
}

#define PUSH_HIPCHK(call)                                     \
  do {                                                        \
    hipError_t _pe = (call);                                  \
    if (_pe != hipSuccess) {                                  \
      push_abort(h, cam);                                     \
      FAIL(ESVO_ERR_HIP, std::string(#call) + " failed: " + hipGetErrorString(_pe)); \
    }                                                         \
  } while (0)
}  // namespace

extern "C" {

static int push_events_impl(esvo_handle h, int cam, const esvo_event_t* ev, size_t n, bool wait);
int esvo_ts_push_events(esvo_handle h, int cam, const esvo_event_t* ev, size_t n) { return push_events_impl(h, cam, ev, n, true); }
int esvo_ts_push_events_async(esvo_handle h, int cam, const esvo_event_t* ev, size_t n) { return push_events_impl(h, cam, ev, n, false); }
int esvo_ts_push_wait(esvo_handle h, int cam) {
  if (!h || cam < 0 || cam > 1) return ESVO_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> lp(h->mu_push[cam]);
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipStreamSynchronize(h->stream_i));
  return ESVO_OK;
}
int esvo_host_alloc(size_t bytes, void** out) {
  esvo_context* h = nullptr;
  if (!out || !bytes) return ESVO_ERR_INVALID_ARG;
  HIPCHK(hipHostMalloc(out, bytes, hipHostMallocDefault));
  return ESVO_OK;
}
int esvo_host_free(void* p) {
  esvo_context* h = nullptr;
  if (p) HIPCHK(hipHostFree(p));
  return ESVO_OK;
}
static int push_events_impl(esvo_handle h, int cam, const esvo_event_t* ev, size_t n, bool wait) {
  if (!h || cam < 0 || cam > 1 || (n && !ev)) return ESVO_ERR_INVALID_ARG;
  if (n == 0) return ESVO_OK;
  if (n > h->ring_cap) FAIL(ESVO_ERR_CAPACITY, "event block larger than the event ring");
  auto stamp = [&](size_t i) { return (u64)ev[i].sec * 1000000000ull + ev[i].nsec; };
  {  // sorted, and not older than what is staged: the fast path; anything else is sorted in as the reference's callbacks do
    bool in_order = true;
    u64 last = stamp(0);
    for (size_t i = 1; i < n && in_order; ++i) {
      const u64 t = stamp(i);
      in_order = t >= last;
      last = t;
    }
    if (in_order) {
      std::lock_guard<std::mutex> lr(h->mu_ring);
      const u64 newest = h->routed ? h->last_stamp[cam] : (h->ts_host[cam].empty() ? 0 : h->ts_host[cam].back());
      in_order = stamp(0) >= newest;
    }
    if (!in_order) return push_unsorted(h, cam, n, [&](size_t i) { return ev[i]; });
  }
  std::lock_guard<std::mutex> lp(h->mu_push[cam]);
  HIPCHK(hipSetDevice(h->device));
  if (h->routed) return push_routed(h, cam, n, [&](size_t i) { return ev[i]; });
  PushTicket tk;
  { int rc = push_begin(h, cam, n, stamp(0), tk); if (rc) return rc; }
  { int rc = push_drain(h, cam, tk); if (rc) return rc; }
  const size_t first = (size_t)std::min<u64>(n, h->ring_cap - tk.slot);
  PUSH_HIPCHK(hipMemcpyAsync(h->d_ring[cam] + tk.slot, ev, sizeof(esvo_event_t) * first, hipMemcpyHostToDevice, h->stream_i));
  if (first < n)
    PUSH_HIPCHK(hipMemcpyAsync(h->d_ring[cam], ev + first, sizeof(esvo_event_t) * (n - first), hipMemcpyHostToDevice, h->stream_i));
  if (wait) {
    PUSH_HIPCHK(hipStreamSynchronize(h->stream_i));  // `ev` is borrowed for the duration of the call only; later work sees the copy
    push_commit(h, cam, n, stamp);
  } else {
    // esvo_ts_push_events_async: the caller keeps `ev` alive until esvo_ts_push_wait; whoever reads these ring slots on the
    // front stream queues behind the copy (ingest_fence)
    PUSH_HIPCHK(hipEventRecord(h->evt_ingest[cam], h->stream_i));
    push_commit(h, cam, n, stamp, true);
  }
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
  const uint8_t* rec = msg + off;
  auto stamp = [&](size_t i) {
    const uint8_t* r = rec + i * 13 + 4;
    const u32 sec = (u32)r[0] | ((u32)r[1] << 8) | ((u32)r[2] << 16) | ((u32)r[3] << 24);
    const u32 nsec = (u32)r[4] | ((u32)r[5] << 8) | ((u32)r[6] << 16) | ((u32)r[7] << 24);
    return (u64)sec * 1000000000ull + nsec;
  };
  auto widen = [&](size_t i) {  // a 13-byte record as esvo_event_t, on the host (the paths that filter or sort there)
    const uint8_t* r = rec + i * 13;
    esvo_event_t e;
    std::memset(&e, 0, sizeof(e));
    e.x = (uint16_t)(r[0] | (r[1] << 8));
    e.y = (uint16_t)(r[2] | (r[3] << 8));
    e.sec = (u32)r[4] | ((u32)r[5] << 8) | ((u32)r[6] << 16) | ((u32)r[7] << 24);
    e.nsec = (u32)r[8] | ((u32)r[9] << 8) | ((u32)r[10] << 16) | ((u32)r[11] << 24);
    e.polarity = r[12] ? 1 : 0;   // (bit 7 of the byte is the library's: EV_LATE, common.hpp)
    return e;
  };
  {  // in order (as esvo_ts_push_events): the fast path; else sorted in on the host
    bool in_order = true;
    u64 last = stamp(0);
    for (size_t i = 1; i < n && in_order; ++i) {
      const u64 t = stamp(i);
      in_order = t >= last;
      last = t;
    }
    if (in_order) {
      std::lock_guard<std::mutex> lr(h->mu_ring);
      const u64 newest = h->routed ? h->last_stamp[cam] : (h->ts_host[cam].empty() ? 0 : h->ts_host[cam].back());
      in_order = stamp(0) >= newest;
    }
    if (!in_order) return push_unsorted(h, cam, n, widen);
  }
  std::lock_guard<std::mutex> lp(h->mu_push[cam]);
  HIPCHK(hipSetDevice(h->device));
  if (h->routed) return push_routed(h, cam, n, widen);
  if ((size_t)n * 13 > h->wire_cap[cam]) {  // this camera's staging buffer: its pusher is the only user (mu_push)
    if (h->d_wire[cam]) { hipFree(h->d_wire[cam]); h->d_wire[cam] = nullptr; }
    h->wire_cap[cam] = std::max<size_t>((size_t)n * 13, (size_t)1 << 20);
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&h->d_wire[cam]), h->wire_cap[cam]));
  }
  PushTicket tk;
  { int rc = push_begin(h, cam, n, stamp(0), tk); if (rc) return rc; }
  { int rc = push_drain(h, cam, tk); if (rc) return rc; }
  PUSH_HIPCHK(hipMemcpyAsync(h->d_wire[cam], rec, (size_t)n * 13, hipMemcpyHostToDevice, h->stream_i));
  launch_ts_unpack_wire(h->d_wire[cam], n, h->d_ring[cam], tk.slot, h->ring_cap, h->stream_i);
  PUSH_HIPCHK(hipGetLastError());
  PUSH_HIPCHK(hipStreamSynchronize(h->stream_i));  // `msg` is borrowed for the duration of the call only
  push_commit(h, cam, n, stamp);
  return ESVO_OK;
}

namespace {
// FORWARD mode, first use for a camera: the contribution lists of TimeSurface.cpp:85-116.  A source pixel s with rectified
// position (u, v), u, v >= 0, u_i + 1 < W, v_i + 1 < H adds to the four pixels around (u, v); every destination pixel gets the
// list of (source, corner) pairs that reach it, in raster order of the sources (a stable counting sort).
int build_forward_lists(esvo_context* h, int cam) {
  if (h->d_fwd_off[cam]) return ESVO_OK;
  const std::vector<float>& lut = h->h_rect_lut[cam];
  const size_t npx = (size_t)h->W * h->H;
  if (lut.size() != 2 * npx) FAIL(ESVO_ERR_STATE, "esvo_ts_render_forward: this camera's rect_lut was not given to esvo_create");
  if (npx >= (1u << 30)) FAIL(ESVO_ERR_UNSUPPORTED, "image too large for the packed contribution records");
  const size_t W = (size_t)h->W, H = (size_t)h->H;
  std::vector<u32> off(npx + 1, 0);
  auto corners = [&](size_t s, size_t dst[4]) -> bool {
    const double u = (double)lut[2 * s], v = (double)lut[2 * s + 1];
    if (!(u >= 0 && v >= 0)) return false;
    const double fu = std::floor(u), fv = std::floor(v);
    if (!(fu < 4e9 && fv < 4e9)) return false;
    const size_t u_i = (size_t)fu, v_i = (size_t)fv;
    if (!(u_i + 1 < W && v_i + 1 < H)) return false;
    dst[0] = v_i * W + u_i; dst[1] = v_i * W + u_i + 1; dst[2] = (v_i + 1) * W + u_i; dst[3] = (v_i + 1) * W + u_i + 1;
    return true;
  };
  size_t dst[4];
  for (size_t s = 0; s < npx; ++s)
    if (corners(s, dst))
      for (int c = 0; c < 4; ++c) off[dst[c] + 1]++;
  for (size_t i = 0; i < npx; ++i) off[i + 1] += off[i];
  std::vector<u32> src(std::max<size_t>(off[npx], 1)), fill(off.begin(), off.end() - 1);
  for (size_t s = 0; s < npx; ++s)
    if (corners(s, dst))
      for (int c = 0; c < 4; ++c) src[fill[dst[c]]++] = (u32)s | ((u32)c << 30);
  HIPCHK(hipMalloc(reinterpret_cast<void**>(&h->d_fwd_off[cam]), sizeof(u32) * (npx + 1)));
  HIPCHK(hipMalloc(reinterpret_cast<void**>(&h->d_fwd_src[cam]), sizeof(u32) * src.size()));
  HIPCHK(hipMalloc(reinterpret_cast<void**>(&h->d_fwd_lut[cam]), sizeof(float2) * npx));
  if (!h->d_fwd_val) HIPCHK(hipMalloc(reinterpret_cast<void**>(&h->d_fwd_val), sizeof(double) * npx));
  HIPCHK(hipMemcpy(h->d_fwd_off[cam], off.data(), sizeof(u32) * (npx + 1), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(h->d_fwd_src[cam], src.data(), sizeof(u32) * src.size(), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(h->d_fwd_lut[cam], lut.data(), sizeof(float2) * npx, hipMemcpyHostToDevice));
  return ESVO_OK;
}
}  // namespace

int esvo_ts_render_forward(esvo_handle h, int cam, uint64_t t_ns, uint8_t* out_mono8) {
  if (!h || cam < 0 || cam > 1) return ESVO_ERR_INVALID_ARG;
  API_LOCK(h);
  HIPCHK(hipSetDevice(h->device));
  if (h->routed) FAIL(ESVO_ERR_UNSUPPORTED, "FORWARD mode on a routed band handle (the splat needs every raw row): use ESVO_ROUTE_BROADCAST");
  { int rc = build_forward_lists(h, cam); if (rc) return rc; }
  {
    std::lock_guard<std::mutex> lr(h->mu_ring);
    ingest_fence(h, cam);
    const auto& tsq = h->ts_host[cam];
    const size_t k = std::lower_bound(tsq.begin(), tsq.end(), (u64)t_ns) - tsq.begin();
    const u64 upto = h->ring_base[cam] + k;
    if (h->tsq_len) queue_prepare(h, cam, (u64)t_ns);
    else if (upto < h->scattered[cam])
      FAIL(ESVO_ERR_STATE, "esvo_ts_render_forward: t_ns precedes events of an earlier render (render times must not decrease; esvo_reset to replay)");
    if (!h->tsq_len && upto > h->scattered[cam]) {  // events with ts < T that are not in the SAE yet (as esvo_ts_render)
      u64 a = h->scattered[cam];
      h->scatter_pending_lo[cam] = std::min(h->scatter_pending_lo[cam], a);
      h->scatter_seq++;
      h->stats.events_scattered[cam] += upto - a;
      while (a < upto) {
        const u64 slot = a % h->ring_cap;
        const u64 cnt = std::min<u64>(upto - a, h->ring_cap - slot);
        launch_ts_scatter(h->d_ring[cam] + slot, (size_t)cnt, h->d_sae[cam], h->W, h->H, h->stream);
        a += cnt;
      }
      h->scattered[cam] = upto;
    }
  }
  {
    std::lock_guard<std::mutex> lt(h->mu_ts);
    resident_write_begin(h, cam);
    launch_ts_render_forward(h->d_sae[cam], h->d_fwd_off[cam], h->d_fwd_src[cam], h->d_fwd_lut[cam], h->d_fwd_val, cam ? h->d_raw1 : h->d_raw,
                             h->d_ts[cam], h->W, h->H, (u64)t_ns, h->prm.decay_ms / 1000.0, h->prm.ignore_polarity,
                             h->prm.median_blur_kernel_size, h->stream);
    if (cam == 0) hipEventRecord(h->evt[EV_R1], h->stream);  // what esvo_track_set_current waits for
    HIPCHK(hipGetLastError());
    h->ts_valid[cam] = true;
  }
  if (cam == 0) h->ts_timing_pending[0] = false;  // EV_R1 no longer belongs to a timed scatter / render pair
  h->stats.ts_frames[cam]++;
  if (out_mono8) {
    HIPCHK(hipMemcpyAsync(out_mono8, h->d_ts[cam], (size_t)h->W * h->H, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
  }
  return ESVO_OK;
}

int esvo_ts_render(esvo_handle h, int cam, uint64_t t_ns, uint8_t* out_mono8) {
  if (!h || cam < 0 || cam > 1) return ESVO_ERR_INVALID_ARG;
  API_LOCK(h);
  HIPCHK(hipSetDevice(h->device));
  const int evo = cam * EV_TS_STRIDE;
  const bool timed = esvo_stage_timed(h);  // sampled for renders that run alone (context.hpp, lat_ticks)
  {
    std::lock_guard<std::mutex> lr(h->mu_ring);
    ingest_fence(h, cam);
    // events with ts < T (strict, TimeSurface.h:68) that are not in the SAE yet
    const auto& tsq = h->ts_host[cam];
    const size_t k = std::lower_bound(tsq.begin(), tsq.end(), (u64)t_ns) - tsq.begin();
    const u64 upto = h->ring_base[cam] + k;
    // The SAE keeps ONE stamp per pixel, the reference a queue of 20 (TimeSurface.h:28-96): rendering at a T that precedes
    // events already scattered would read pixels as empty where getMostRecentEventBeforeT finds the older event.
    if (!h->tsq_len && upto < h->scattered[cam])
      FAIL(ESVO_ERR_STATE, "esvo_ts_render: t_ns precedes events of an earlier render (render times must not decrease; esvo_reset to replay)");
    if (h->ts_timing_pending[cam] && hipEventQuery(h->evt[EV_R1 + evo]) == hipSuccess) collect_ts_timing(h, cam);
    if (timed) hipEventRecord(h->evt[EV_SC0 + evo], h->stream);
    if (h->tsq_len) queue_prepare(h, cam, (u64)t_ns);
    else if (upto > h->scattered[cam]) {
      u64 a = h->scattered[cam];
      h->scatter_pending_lo[cam] = std::min(h->scatter_pending_lo[cam], a);
      h->scatter_seq++;
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
    if (timed) hipEventRecord(h->evt[EV_SC1 + evo], h->stream);
  }
  {
    std::lock_guard<std::mutex> lt(h->mu_ts);
    resident_write_begin(h, cam);
    launch_ts_render(h->d_sae[cam], h->d_fixmap[cam], h->d_raw, h->d_ts[cam], h->W, h->H, (u64)t_ns, h->prm.decay_ms / 1000.0,
                     h->prm.ignore_polarity, h->prm.median_blur_kernel_size, h->stream, h->routed ? h->rband_y0 : 0,
                     h->routed ? h->rband_y1 : -1);
    if (timed || (cam == 0 && h->trk_used.load())) hipEventRecord(h->evt[EV_R1 + evo], h->stream);
    HIPCHK(hipGetLastError());
    h->ts_valid[cam] = true;
  }
  h->ts_timing_pending[cam] = timed;
  h->stats.ts_frames[cam]++;
  if (out_mono8) {
    HIPCHK(hipMemcpyAsync(out_mono8, h->d_ts[cam], (size_t)h->W * h->H, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    collect_ts_timing(h);
  }
  return ESVO_OK;
}

}  // extern "C"
