// context.hpp — the state behind an esvo_handle and what the api_*.hip translation units share.
//
// One esvo_context owns every device buffer of one GPU: SAE + staged event rings (Time Surface), the observation pair,
// per-tick BM/LM scratch, the fusion window ring and the dense DepthMap.  Each entry point replays, on the handle's HIP
// streams, the call sequence of the reference seam it replaces (cited in include/esvo_hip.h); nothing computes on the
// CPU except bookkeeping (time-stamp binary searches, window policy, output ordering).
//   api_core.hip   lifecycle, parameters, self-tests           api_ts.hip     event ingest, Time-Surface render
//   api_map.hip    mapper: stage-wise calls, ticks, sharding   api_track.hip  tracker residual / Jacobian evaluation
#pragma once
#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <vector>

#include "common.hpp"

using namespace esvo;

// front-stage events live on the front stream; the back-stage set exists once per tick parity (two ticks in flight)
enum { EV_SC0 = 0, EV_SC1, EV_R1, EV_SC0b, EV_SC1b, EV_R1b, EV_FRAME,
       EV_T0, EV_BM0, EV_BM1, EV_S1, EV_LM0, EV_LM1, EV_S2, EV_CNT, EV_STG, EV_A1, EV_T0b, EV_BM0b, EV_BM1b, EV_S1b, EV_LM0b, EV_LM1b, EV_S2b, EV_CNTb, EV_STGb, EV_A1b,
       EV_FU0, EV_FU1, EV_CL1, EV_RG1, EV_POSE, EV_FU0b, EV_FU1b, EV_CL1b, EV_RG1b, EV_POSEb, EV_N };
constexpr int EV_BACK_STRIDE = EV_FU0b - EV_FU0;  // evt[EV_x + par * EV_BACK_STRIDE]
constexpr int EV_FRONT_STRIDE = EV_T0b - EV_T0;   // evt[EV_x + fpar * EV_FRONT_STRIDE]
constexpr int EV_TS_STRIDE = EV_SC0b - EV_SC0;    // evt[EV_x + cam * EV_TS_STRIDE]

struct FrameRec {
  u32 off;    // offset in the window ring
  u32 count;  // points
  u32 slot;   // pose-table slot (NO_SLOT for empty frames)
  u32 run;    // 1, or the number of consecutive EMPTY frames this record stands for: a frame without points adds
              // nothing to the fusion but counts as a frame of the window (esvo_Mapping.cpp:341-368,385), and a
              // sparse stretch under CONST_POINTS may queue any number of them
};
enum : u32 { NO_SLOT = 0xffffffffu };

struct esvo_context {
  esvo_params_t prm;
  DevParams dp;
  int W = 0, H = 0, device = 0;
  // Two streams, two ticks in flight: the front stage (TS, block matching, LM, frame assembly) of tick k+1 runs on
  // `stream` while the back stage (propagate, fuse, clean, regularise) of tick k runs on `stream_b`.  The back
  // stage only reads what the front stage finished (the frame in the window ring, the tick's pose table).
  hipStream_t stream = nullptr;
  hipStream_t stream_b = nullptr;
  // LM stage of a lazy tick (refinement, frame assembly, counters): block matching and the Time Surfaces of the NEXT tick run
  // beside it on `stream`.  What the two stages share is double-buffered by POINTER SWAP: d_obs / d_matches / d_counters
  // below alias one of two buffers each (kernels capture the pointer at launch, so work in flight keeps its own).
  hipStream_t stream_l = nullptr;
  // A SECOND queue for the LM stage (round 3): the LM stages of consecutive lazy ticks alternate between stream_l and
  // stream_l1 by the tick's parity, so the head of tick k+1's launch fills the chip while the tail of tick k's -- a few
  // long dependent chains -- drains.  Used for launches in the latency-bound (wide) layout; everything the stage writes
  // (d_pt_slots / d_pt_flags / d_pt_prefix / the scan scratch, besides the buffers listed above) exists once per parity.
  hipStream_t stream_l1 = nullptr;
  bool collect_aside = true;   // one LM queue in use: a tick's compaction + counters go to the other one (api_map.hip, tick_phase0)
  bool lm_two_now = false;     // this tick's LM launch alternates between the two queues
  // Whether the second queue pays depends on what else the tick holds: where the LM launch is much longer than the fusion
  // stage (346x260, no regulariser: 0.37 ms against 0.12) two launches in flight raise the rate by 18 %; where the two are
  // of similar length (DSEC's reference-faithful tick: 0.25 against 0.30 ms) the fusion stage is the bottleneck either way
  // and a second resident LM launch only takes its registers (0.33 -> 0.375 ms).  So the handle decides from its own stage
  // timings (HIP events of the completed ticks, smoothed).  Scheduling only -- results do not depend on it.  ESVO_LM_QUEUES = 1 / 2
  // forces never / always.  Thresholds: on above 1.5 x, off below 1.2 x until round 6; since the back chain lost ~60 us (no markers
  // between its kernels on most ticks, the one-launch prologue, reg_view's counter) the DSEC tick gains from the second queue as well
  // (0.289 -> 0.264 ms per overlapping tick; 346x260 1000 events 0.203 -> 0.191): on above 0.9 x, off below 0.7 x
  // (profiles/r06_lowlat_tick.txt).
  int lm_queues = 0;              // 0 auto, 1 never, 2 always
  float ema_lm_ms = 0.f, ema_back_ms = 0.f;
  bool lm_two_on = false;
  u32 lm_two_max = esvo::LM_TWO_QUEUES_MAX_EVENTS;  // launches bounded by more events stay on one queue
  bool lm_split = true;           // ESVO_LM_STREAM=0: everything of the front stage on `stream`
  bool one_stream = false;       // ESVO_ONE_STREAM=1 (A/B): stream_b aliases stream
  // ESVO_TIMELINE=1 (tools/regime_probe.py): when every stage of every tick ran, collected from the HIP events as they complete
  bool tl_on = false;
  hipEvent_t tl_ref = nullptr;
  std::vector<std::array<float, 8>> tl_front;
  std::vector<std::array<float, 4>> tl_back;
  // the pipeline's way back from its slow operating point (api_map.hip, pipeline_resync); ESVO_RESYNC=0 (A/B) switches it off
  bool resync_on = true;
  struct Resync { double last_ms = 0; float period_ema = 0, period_before = 0; u32 streak = 0, cooldown = 0, check_in = 0; bool lm_wait_back = false; } resync;
  bool front_throttle = false;   // ESVO_FRONT_THROTTLE=1 (A/B): an unsharded tick's front stage waits for the back stage two ticks ago (api_map.hip; the default until round 4)
  bool split_now = false;         // set by esvo_map_tick around its front stage: only the lazy tick path splits
  // Latency mode (round 6, api_map.hip): a tick that arrives while nothing of the previous one is pending -- the caller reads every
  // tick's result before it hands in the next, as the ROS node does -- has nothing to overlap with.  Its LM launch stays in the
  // front queue (no cross-queue hand-off: ~25 us) and the host polls for its counters and its end instead of sleeping on the
  // completion interrupt (~10-20 us per wake-up), for ticks of at most lat_max_events events.  ESVO_LOWLAT=0 (A/B) switches it off.
  bool lat_mode = true;
  bool lat_now = false;           // set by esvo_map_tick around its front stage
  bool pipe_now = false;          // ... when the previous tick is still pending (the two overlap)
  bool lat_last = false;          // the newest tick was enqueued in latency mode (what synchronising calls look at)
  u32 lat_max_events = 40000u;
  // Stage timings are SAMPLED on that path.  Every hipEventRecord between two dependent kernels costs the queue ~5 us (a marker
  // packet the next dispatch waits for: kernels with no event between them follow each other with no gap at all --
  // profiles/r06_lowlat_tick.txt), and a tick recorded fourteen of them for nothing but esvo_stats_t's ms_* fields.  A tick that
  // runs alone records them for its first 8 ticks and for one tick in lat_timed_every afterwards; ms_* / ms_kernel[] hold the latest
  // sample, sum_ms_kernel[] sums the samples, stats.stage_timing_samples counts them.  The LM layout policy (lm_pair_*) lives on LM
  // launch times: the two events around the LM launch are recorded as well whenever the policy is exploring or trying the layout it
  // is not using (TickState::timed_lm).  Ticks that overlap (the throughput path) record all of them as before.
  u32 lat_ticks = 0;              // ticks enqueued with nothing pending before them (band-sharded ticks: all of them -- every
                                  // phase of such a tick is waited for by the exchange that follows it)
  // Overlapping SMALL ticks (at most lat_max_events events; two in flight, nobody waits) are paced by the host: ~45 runtime calls per
  // tick, a third of them event records.  They sample their stage timings one tick in pipe_timed_every; the throughput path (larger
  // ticks, paced by the LM kernel) keeps recording every tick.
  u32 pipe_seq = 0;
  u32 pipe_timed_every = 4;       // ESVO_PIPE_TIMED_EVERY (A/B; 1 = every tick)
  u32 pipe_big_seq = 0;
  u32 pipe_big_every = 1;         // ESVO_PIPE_BIG_TIMED_EVERY (A/B): the same for large overlapping ticks -- the throughput path, paced by the LM
                                  // queue.  One in four measured -0.5 % on one box and +0.3 % on another, a four-way A/B with the prologue
                                  // switch nothing at all (profiles/r06_throughput_ab.txt): every tick stays timed there
  u32 lat_timed_every = 31;       // ESVO_LOWLAT_TIMED_EVERY (A/B; 1 = every tick)
  u32* cnt_row_host = nullptr;    // latency mode: where the tick's point compaction leaves the counter row (null: a copy follows)
  bool cnt_row_sent = false;
  int reg_sparse_forced = -1;     // ESVO_REG_SPARSE (A/B): the regulariser's sparse-map layout never (0) / always (1); -1: by the element count
  bool pro_always = false;        // ESVO_BACK_PROLOGUE=1 (A/B): overlapping ticks open their back stage with the one-launch prologue too (neutral:
                                  // profiles/r06_throughput_ab.txt; the path that has run for four rounds stays)
  bool match_by_index = false;    // latency mode: this tick's match list is d_own_w (indices into d_match_slots), not d_matches
  bool gather_guard[2] = {false, false};  // the solver-slot buffers of that parity are read by a back stage's first launch (EV_STG releases them)
  bool stage_events_on = true;    // false while a tick whose stage timings are not sampled is being enqueued (api_map.hip)
  bool back_timed[2] = {true, true};  // the back stage of that parity recorded its stage events
  std::atomic<bool> trk_used{false};  // esvo_track_set_current has read the resident surface: renders record EV_R1 for it
  // ... and the copies that open its back stage (frame into the ring, pose table into the frame's slot) are not enqueued one by one
  // but handed to run_fuse, whose frame-table upload carries them in the same launch (scan.hip, back_prologue_kernel)
  struct DeferredCopies {
    bool active = false;
    const void* a_src = nullptr; void* a_dst = nullptr; size_t a_bytes = 0; int ev_a = -1;  // frame points; event recorded behind it
    const u32* a_flags = nullptr; const u32* a_prefix = nullptr; u32 a_slots = 0;          // gather mode: a_src = the solver slots
    bool tail_b = false;  // ev_b recorded at the end of the back stage too (a tick that runs alone); else right behind the launch:
                          // the front stage two ticks on, which overlaps this back stage, waits for the pose table's copy
    const void* b_src = nullptr; void* b_dst = nullptr; size_t b_bytes = 0; int ev_b = -1;  // pose table
  } pro;
  uint8_t* d_obs2[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
  int obs_par = 0;
  esvo_match_t* d_matches2[2] = {nullptr, nullptr};
  u32* d_counters2[2] = {nullptr, nullptr};
  u32* d_scan_tmp_l = nullptr;    // scan scratch of the LM stage (of the current parity)
  u32* d_scan_tmp_l2[2] = {nullptr, nullptr};
  hipStream_t stream_i = nullptr;  // event ingest (H2D into the ring): staging new events never waits for a running tick
  bool own_stream = false;
  int par = 0;                    // parity of the tick being assembled
  bool back_pending[2] = {false, false};  // back-stage timings / counters of that parity not collected yet
  u32 back_frames[2] = {0, 0};
  double baseline = 0;

  // ---- threading contract (include/esvo_hip.h, "Threads"): three groups of calls may run concurrently on one handle --
  // INGEST (esvo_ts_push_events / _event_array / _bag: the ROS spinner's eventsCallback), TRACKER (esvo_track_*) and
  // everything else (the MAPPER group: renders, ticks, outputs, parameters).
  std::recursive_mutex mu_api;  // held by every mapper-group call for its duration (they call each other: recursive)
  std::mutex mu_ring;           // event-ring bookkeeping: ts_host, ring_base / _next / _reserved, scattered, scatter_pending_lo,
                                // scatter_seq, sh_first, stats.events_staged / _scattered.  Never held across a host wait.
  std::mutex mu_push[2];        // one pusher per camera at a time (held across its host-to-device copy)
  std::mutex mu_track;          // tracker-group calls
  std::mutex mu_ts;             // the resident left Time Surface (d_ts[0], ts_valid[0], EV_R1) between a render and a
                                // tracker read (esvo_track_set_current without a host image)

  // calibration
  float2* d_lut = nullptr;
  uint8_t* d_mask = nullptr;
  int2* d_fixmap[2] = {nullptr, nullptr};

  // Time Surface
  u64* d_sae[2] = {nullptr, nullptr};
  uint8_t* d_raw = nullptr;
  uint8_t* d_raw1 = nullptr;
  // FORWARD-mode Time Surface (esvo_ts_render_forward): host copy of each camera's rect_lut, and -- built on first use --
  // its device copy, the per-destination contribution lists (CSR: offsets, source index | corner << 30) and a f64 scratch
  std::vector<float> h_rect_lut[2];
  float2* d_fwd_lut[2] = {nullptr, nullptr};
  uint32_t* d_fwd_off[2] = {nullptr, nullptr};
  uint32_t* d_fwd_src[2] = {nullptr, nullptr};
  double* d_fwd_val = nullptr;  // the right camera's raw surface when both cameras render in one launch (esvo_map_tick_resident)
  uint8_t* d_ts[2] = {nullptr, nullptr};
  bool ts_valid[2] = {false, false};
  esvo_event_t* d_ring[2] = {nullptr, nullptr};
  uint8_t* d_wire[2] = {nullptr, nullptr};  // staging of serialised 13-byte event records (esvo_ts_push_event_array), per camera
  size_t wire_cap[2] = {0, 0};
  u64 ring_cap = 0;
  std::deque<u64> ts_host[2];   // time stamps of staged events [ring_base, ring_base + size)
  u64 ring_base[2] = {0, 0};    // absolute index of ts_host[cam].front()
  u64 ring_next[2] = {0, 0};    // absolute index of the next event to stage
  u64 ring_reserved[2] = {0, 0};  // >= ring_next: end of the block a pusher is copying right now (its slots are being
                                  // overwritten: selections are validated against THIS, not against ring_next)
  u64 scatter_seq = 0;          // scatter launches so far (a pusher that drained the front stream resets scatter_pending_lo
                                // only if no scatter was enqueued meanwhile)
  u64 scattered[2] = {0, 0};    // absolute index of the first event not yet in the SAE
  u64 scatter_pending_lo[2] = {~0ull, ~0ull};  // oldest event a possibly still running scatter kernel reads
  // esvo_ts_push_events_async: the newest enqueued (not awaited) copy of a camera; consumers of the ring on the front stream
  // queue behind it (ingest_fence, api_ts.hip) -- under mu_ring
  hipEvent_t evt_ingest[2] = {nullptr, nullptr};
  bool ingest_pending[2] = {false, false};

  // observation
  uint8_t* d_obs[2] = {nullptr, nullptr};
  uint8_t* d_obs_tmp = nullptr;
  double T_world_obs[16];
  u64 obs_t_ns = 0;
  bool obs_set = false;

  // pose table of the tick
  double* d_pose_sec = nullptr;   // toSec() of the stamps: the tail of the tick's table (d_pose_T + 16 m)
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
  // split LM launch (kernels_lm.hip): F(x0) of every match, its cost, the processing order; allocated when the handle can see
  // launches above the wide layout's bound
  double* d_lm_fvec0 = nullptr;
  double* d_lm_fnorm0 = nullptr;
  u32* d_lm_meta = nullptr;
  u32* d_lm_order = nullptr;
  u32* d_lm_hist = nullptr;
  u64* d_clk = nullptr;           // in-run shader-clock probe of the LM kernel (LmArgs::clk, common.hpp); read by esvo_get_stats
  bool clk_probe = true;          // ESVO_CLK_PROBE=0 (A/B only) launches the LM kernel without it
  // ESVO_LM_PERSIST=1 (A/B): launches of the throughput layout use lm_refine_persist_kernel (kernels_lm.hip).  Bit-identical and
  // 13 % shorter as a launch (1.24 -> 1.075 ms beside the other stages), but OFF by default: a persistent grid has no draining
  // tail, and the regulariser -- which runs beside that tail in the pipelined tick -- then takes 0.87 instead of 0.60 ms, so the
  // back chain paces the tick at 1.6 ms (1.31 at best with a smaller grid; profiles/r05_ab_lm_persist.txt)
  bool lm_persist = false;
  u32 lm_persist_blocks = 2048;   // workgroups (= waves) of the persistent layout: two per SIMD
  int lm_split_mode = -1;         // the split launch: -1 by launch size (>= 400 000 events), 0 never, 1 always (ESVO_LM_SPLIT)
  DevPoint* d_pt_slots = nullptr;   // LM output by slot + keep flags + their scan: alias one of two sets (front parity)
  u32* d_pt_flags = nullptr;
  u32* d_pt_prefix = nullptr;
  DevPoint* d_pt_slots2[2] = {nullptr, nullptr};
  u32* d_pt_flags2[2] = {nullptr, nullptr};
  u32* d_pt_prefix2[2] = {nullptr, nullptr};
  DevPoint* d_pts_tmp = nullptr;  // stage-wise refine output
  DevPoint* d_stage[2] = {nullptr, nullptr};  // a lazily completed tick's frame (by parity) until its count is known
  u32* d_counters = nullptr;      // [0] n_matches [1] n_points [2] n_solved [3] n_fusion [4] n_records [5] n_map
                                  // [6] touched cells [7] regulariser elements [8] own matches (sharded)
  u32* h_counters = nullptr;      // pinned
  u32* d_scan_tmp = nullptr;
  u32* d_cnt_b = nullptr;         // back stage: [2] overflow cursor of the fusion front [3] n_fusion [4] n_records [5] n_map [6] touched cells [7] regulariser elements
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
  u32 max_frames = 0;             // capacity in NON-EMPTY frames (pose slots, frame tables)
  size_t n_window_frames = 0;     // frames of the window as the reference counts them (empty ones included)

  // DepthMap
  DevPoint* d_prop = nullptr;
  // fusion front (kernels_fuse.hip): per tile a fixed-capacity list of point ids + one shared overflow list; per cell the
  // (offset, count) of its sorted record list; the touched cells by length class; counters (class sizes, cursors)
  u32* d_tile_count = nullptr;
  uint2* d_tile_pts = nullptr;
  uint2* d_over_pts = nullptr;
  u32* d_cell_count = nullptr;
  u32* d_cell_offset = nullptr;
  u32* d_cell_list = nullptr;
  u32* d_fuse_ctr = nullptr;      // [0..1023] class_count [1024..2048] class_total [2080] rec_cursor [2081] over_count
  u32 fuse_tile_rec = 4096;       // entries of a tile's own region of d_rec_ids (ESVO_FUSE_TILE_REC: tests)
  u32 fuse_slice_cap = 0;         // entries of one (class, slice) segment of d_cell_list
  u32 fuse_tile_cap = 1024;       // ESVO_FUSE_TILE_CAP (tests): entries per tile list
  u32 fuse_pmax_plus1 = 0;        // ESVO_FUSE_PMAX (tests) + 1: candidates up to which a tile takes the bit-row path
  u32* d_rec_ids = nullptr;       // record ids of cells whose list does not fit LDS (degenerate scenes)
  u32 fuse_lds_cap = 0;           // ESVO_FUSE_LDS_CAP (tests): record ids per tile kept in LDS; 0 = the maximum
  MapCell* d_map = nullptr;
  MapCell* d_map2 = nullptr;
  MapCell* d_map_cur = nullptr;
  u32* d_owner_max = nullptr;
  u32* d_owner_min = nullptr;
  u32* d_sel = nullptr;           // denoising: walk positions of the kept events
  uint8_t* d_evmap = nullptr;     // denoising: binary event map
  // sharded mode (kernels_shard.hip): dense local lists + the (matched, kept) byte per slot that is exchanged
  u32* d_own_w = nullptr;         // slot w of the k-th own match
  u32* d_lkeep = nullptr;         // keep flag of the k-th own match after LM + culling
  uint8_t* d_codes = nullptr;     // [codes_bytes] one byte per slot (all ranks' slots, after exchange 1)
  size_t codes_bytes = 0;
  // what the caller must all-gather across the ranks before the next phase (esvo_shard_exchange): xchg_block bytes from
  // xchg_send of every rank into xchg_recv, rank-major.  One rank (n_shards == 1): recv aliases send, nothing to exchange.
  void* xchg_send = nullptr;
  void* xchg_recv = nullptr;
  size_t xchg_block = 0;
  uint8_t* d_codes_send = nullptr;             // exchange 1: [roundup8(ceil(n / N))] the bytes of the own slots r, r + N ...
  uint8_t* d_codes_all = nullptr;              //             [N][that]
  unsigned long long* d_pts_send = nullptr;    // exchange 2: [1 + own * 13] count | kept points (final index in seq)
  unsigned long long* d_pts_all = nullptr;     //             [N][1 + max_kept * 13]
  u32* d_rank_kept = nullptr;                  // [SHARD_MAX_RANKS] kept count of every rank (from exchange 1)
  static constexpr u32 SHARD_MAX_RANKS = 1024;
  bool sharded = false;
  // ---- routed band mode (esvo_shard_set_routing): events are routed by image row at ingest, the Time Surfaces are
  // rendered for the band + halo only, block matching and refinement belong to the rank that owns floor(y_rect) of the event.
  bool routed = false;
  int ts_halo = 0;                       // rows beyond the band for which the observation pair is valid
  int rband_y0 = 0, rband_y1 = 0;        // rectified rows the Time Surfaces are rendered for (whole tiles of TS_TILE_ROWS)
  int oband_y0 = 0, oband_y1 = 0;        // rows of the observation pair that hold data: what block matching and LM may read
  int sband_y0[2] = {0, 0}, sband_y1[2] = {0, 0};  // RAW rows whose events a camera's SAE needs for rband (remap + median taps)
  std::vector<int> fix_row_lo[2], fix_row_hi[2];   // per rectified row: the raw rows its remap taps reach (from the fixed-point maps)
  std::vector<uint8_t> keep_px;          // left camera, [W * H]: bit 0 the SAE needs the pixel's events, bit 1 floor(y_rect) is in the band
  // event selection stays GLOBAL (dataTransferring walks the whole left stream): every left stamp is kept on the host, each
  // kept event remembers its index in that sequence -- under mu_ring like ts_host
  std::deque<u64> glob_ts;               // stamps of ALL left events [glob_base, glob_base + size)
  u64 glob_base = 0;
  std::deque<u64> kept_g;                // global index of each kept left event, aligned with ts_host[0]
  std::deque<u64> own_before;            // how many OWN events (floor(y_rect) in the band) were kept before this one: the count of a
  u64 own_total = 0;                     //   selection's own events bounds its LM launch (the ring also holds the raster's halo events)
  u64 last_stamp[2] = {0, 0};            // newest stamp seen per camera (kept or not): the order check of the push calls
  u32* d_ring_gidx = nullptr;            // [ring_cap] low 32 bits of the global index of the left ring's events
  esvo_event_t* h_route_ev[2] = {nullptr, nullptr};  // pinned staging of the kept events of one push, per camera
  u32* h_route_gidx = nullptr;
  size_t route_cap[2] = {0, 0};
  // out-of-order packets (api_ts.hip, push_unsorted): scratch of the ring merge; queue mode: the copies of the then-newest event
  // that take a late event's place in the per-pixel queues, inserted with the next batch
  esvo_event_t* d_merge_a = nullptr;
  esvo_event_t* d_merge_b = nullptr;
  u32* d_merge_plan = nullptr;
  size_t merge_cap_a = 0, merge_cap_b = 0, merge_cap_plan = 0;
  std::vector<esvo_event_t> tsq_dup[2];
  esvo_event_t* d_tsq_dup = nullptr;
  size_t tsq_dup_cap = 0;
  u32* d_halo_viol = nullptr;            // [2] matches whose refinement read outside oband, all ranks, summed over the ticks | scratch
  bool halo_error = false;               // sticky (esvo_reset clears it): ticks are refused with ESVO_ERR_HALO
  bool dn_pending = false;               // Denoising on a routed handle: phase 0 returned ESVO_AGAIN, the mask bits are being exchanged
  u32* d_dn_flags = nullptr;             // [2][max_ev] kept flag + exclusive prefix per walk position of the raw selection (lazily)
  // A tick's state between its phases.  Unsharded ticks are finished lazily: esvo_map_tick(k) enqueues the front
  // stage of tick k and only then completes tick k-1 (point count -> window policy -> back stage), so the host
  // never waits on the front stream while it still has work to enqueue there.
  struct TickState {
    u32 n = 0, off = 0, points = 0, n_pose = 0;
    u32 n_loc = 0;                    // routed band mode: events of the selection in this rank's ring (n stays the global count)
    u32 n_own = 0;                    //   ... of which the rank owns (block-matches and refines) this many
    u32 g_first = 0;                  //   global index (low 32 bits) of the selection's newest event
    u32 max_kept = 0;                 // sharded: largest kept count among the ranks (block length of exchange 2)
    int pose_buf = 0;
    u64 t_ns = 0;
    double T_world_obs[16];
    hipStream_t lm_stream = nullptr;  // where the tick's LM launch was enqueued
    hipStream_t cnt_stream = nullptr; // where its frame (point compaction) and counters follow: the same, or the idle second LM queue
    int obs_par = 0;                  // which observation pair it reads
    int lm_pair = -1;                 // LM layout of the tick: 1 pair, 0 wide, -1 not a candidate (policy feedback)
    bool lat = false;                 // enqueued in latency mode: LM in the front queue, polled waits
    bool timed = true;                // its stage-timing events were recorded
    bool gather = false;              // latency mode: its frame is still in the solver slots (flags + prefix): the back stage's first launch compacts it
    bool timed_lm = true;             // ... at least the two around the LM launch (the layout policy's feedback)
  } tk[2];
  // pair layout of the LM kernel (kernels_lm.hip): chosen per tick from the LM launch times the handle measures anyway
  // (HIP events).  The first eight candidate ticks alternate between the layouts, then the faster one is used, with one tick
  // of the other every 64 so that a change of scene is noticed; a layout's figure is the MINIMUM of its last four launches
  // (the first ticks of a handle are slow whatever the layout).  ESVO_LM_PAIR = 0 / 1 forces never / always.  Scheduling
  // only: same bits either way.
  int lm_pair_forced = -1;
  float lm_pair_ms[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  u32 lm_pair_n[2] = {0u, 0u};
  u32 lm_pair_decisions = 0;
  int lm_pair_current = -1;       // the layout the policy settled on (-1: still exploring)
  int fpar = 0;                   // parity of the newest front stage
  bool tick_pending = false;      // tk[fpar] has its front stage enqueued but is not committed yet
  u64 committed_t_ns = 0;         // stamp of the newest tick whose back stage is enqueued (0: none)
  // per-pixel event queues (max_event_queue_len > 0; kernels_ts.hip): key sets of both cameras + the batch's tile lists
  int tsq_len = 0;
  u64* d_tsq[2] = {nullptr, nullptr};
  u32* d_tsq_tcount = nullptr;
  uint4* d_tsq_tlist = nullptr;
  uint4* d_tsq_over = nullptr;
  u32* d_tsq_over_count = nullptr;
  u32 tsq_tcap = 0;
  static constexpr u64 TSQ_ROUND = 1ull << 20;  // events per insertion round = capacity of the overflow list
  u64 sh_first = 0;
  u64 sh_first_prev = 0;  // the selection before it (two ticks may be in flight)
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
  double* h_trk_ne = nullptr;     // pinned: the 28 sums of esvo_track_normal_equations
  float* h_trk_xyz = nullptr;     // pinned staging of esvo_track_set_reference's point cloud
  bool trk_xyz_inflight = false;  // an upload out of it has been enqueued and no call has waited for the tracker stream since
  size_t trk_cap = 0, trk_n = 0;
  bool trk_cur = false;
  hipEvent_t evt_trk_read = nullptr;  // the tracker stream has read the resident left Time Surface (mu_ts)
  bool trk_read_pending = false;

  // pinned staging slots for frame pose tables that arrive from the host (push_frame variants): a slot is reused only
  // after the back stream has consumed it
  static constexpr int POSE_POOL = 32;
  double* h_pose_pool = nullptr;
  hipEvent_t pool_evt[POSE_POOL];
  bool pool_ok = false;
  int pool_next = 0;

  // SGM initialisation (kernels_sgm.hip): allocated on first use, released by esvo_destroy
  SgmScratch sgm = {};
  bool sgm_ok = false;
  uint8_t* d_sgm_img[2] = {nullptr, nullptr};
  int16_t* d_sgm_disp = nullptr;
  u32* d_sgm_pair = nullptr;      // [2][4 * max_ev] winner flags / ranks of naive_propagation
  double* d_sgm_T = nullptr;

  // debug images (kernels_viz.hip): allocated on first use
  uint8_t* d_viz_bgr = nullptr;
  uint8_t* d_viz_jet = nullptr;
  u32* d_viz_owner = nullptr;

  struct esvo_comm* comm = nullptr;  // multi-GPU exchange (api_comm.hip), null on single-GPU handles

  hipEvent_t evt[EV_N];
  bool evt_ok = false;
  esvo_stats_t stats;
  bool ts_timing_pending[2] = {false, false};
  bool ts_pair_sample = false;  // the pending sample of camera 0 covers both cameras (pair render)
};

namespace esvo_host {
// esvo_last_error: the message of the calling THREAD's last failed call (with or without a handle) -- three threads may
// use one handle at the same time, a string inside the handle could be rewritten while another thread reads it
extern thread_local std::string g_create_error;
// api_core.hip
void fill_dev_params(esvo_context* h);
void set_compute_band(esvo_context* h);
void release_routing(esvo_context* h);
// api_ts.hip
void collect_ts_timing(esvo_context* h, int only = -1);
void ingest_fence(esvo_context* h, int cam);  // caller holds mu_ring
int ts_scatter_ahead(esvo_context* h, uint64_t t_ns);
void resident_write_begin(esvo_context* h, int cam);
int ts_render_pair(esvo_context* h, uint64_t t_ns, uint8_t* const obs_out[2]);
// api_map.hip
int flush_pending_tick(esvo_context* h);  // completes a lazily finished tick (see esvo_context::TickState)
int finalize_tick_stats(esvo_context* h);
int run_bm(esvo_context* h, const esvo_event_t* d_ev, u64 first, u64 cap, int reverse, u32 n, const u32* sel = nullptr);
int run_order_points(esvo_context* h, u32 max_matches, DevPoint* dst, hipStream_t st = nullptr);
int back_after_front(esvo_context* h);
void collect_back(esvo_context* h, int par);
int window_reserve(esvo_context* h, u32 n, u32* off_out);
int commit_frame(esvo_context* h, u32 off, u32 count, const double* pose_T_host, u32 m, int pose_buf = 0, bool apply_policy = true);
int run_fuse(esvo_context* h, int par, const double* T_world_obs, bool naive = false);
int export_map(esvo_context* h, std::vector<esvo_depth_point_t>& out, std::vector<u32>* cells);
int tick_phase0(esvo_context* h, uint64_t t_ns, const uint64_t* pose_t_ns, const double* pose_T, size_t m);
int tick_phase1_enqueue(esvo_context* h);
int tick_phase1_collect(esvo_context* h, int fp);
int collect_front_stats(esvo_context* h, esvo_context::TickState& tk, const u32* cnt, const hipEvent_t* ev);
int tick_phase2(esvo_context* h, int fp);
void begin_observation(esvo_context* h);
void revert_observation(esvo_context* h);
// api_comm.hip
void comm_release(esvo_context* h);
void comm_reset(esvo_context* h);
}  // namespace esvo_host
using namespace esvo_host;

#define HIPCHK(call)                                                                              \
  do {                                                                                            \
    hipError_t _e = (call);                                                                       \
    if (_e != hipSuccess) {                                                                       \
      char _b[512];                                                                               \
      snprintf(_b, sizeof(_b), "%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); \
      (void)h; g_create_error = _b;                                                               \
      return ESVO_ERR_HIP;                                                                        \
    }                                                                                             \
  } while (0)

// Polled waits (latency mode, api_map.hip): hipEventQuery / hipStreamQuery in a loop for at most `budget_us`, then the blocking
// call.  The blocking calls sleep on the completion interrupt (10-20 us from the signal to the woken thread); a tick the caller
// waits for pays that twice (counters, end of tick).  "Not ready" is an error code the runtime remembers: it is cleared here, or
// the next hipGetLastError() behind a launch would report it.
inline hipError_t esvo_wait_event(hipEvent_t e, bool poll, double budget_us = 3000.0) {
  if (poll) {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
      const hipError_t q = hipEventQuery(e);
      if (q == hipSuccess) return q;
      (void)hipGetLastError();
      if (q != hipErrorNotReady) return q;
      if (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() > budget_us) break;
    }
  }
  return hipEventSynchronize(e);
}
inline hipError_t esvo_wait_stream(hipStream_t s, bool poll, double budget_us = 3000.0) {
  if (poll) {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
      const hipError_t q = hipStreamQuery(s);
      if (q == hipSuccess) return q;
      (void)hipGetLastError();
      if (q != hipErrorNotReady) return q;
      if (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() > budget_us) break;
    }
  }
  return hipStreamSynchronize(s);
}

struct StageEventsScope {  // stage-timing events off (or on) for the calls of one scope, back on at its end whatever the exit
  esvo_context* h;
  StageEventsScope(esvo_context* hh, bool on) : h(hh) { h->stage_events_on = on; }
  ~StageEventsScope() { h->stage_events_on = true; }
};
// whether the operations enqueued NOW (renders, a tick's stages) record their stage-timing events (context.hpp, lat_ticks)
inline bool esvo_stage_timed(const esvo_context* h) {
  if (!h->lat_mode || h->tl_on || (h->comm && !h->sharded)) return true;  // (tick-interleaved ranks: every tick)
  if (h->tick_pending) {  // overlapping ticks: the small ones are paced by the HOST's enqueueing (pipe_seq); the large ones: pipe_big_*
    if (h->tk[h->fpar].n && h->tk[h->fpar].n <= h->lat_max_events) return h->pipe_seq % h->pipe_timed_every == 0u;
    return h->pipe_big_seq % h->pipe_big_every == 0u;
  }
  return h->lat_ticks < 8u || h->lat_ticks % h->lat_timed_every == 0u;
}

#define ESVO_SET_ERR(msg) (esvo_host::g_create_error = (msg))
#define API_LOCK(h) std::lock_guard<std::recursive_mutex> _api_lock((h)->mu_api)

#define FAIL(code, msg)                      \
  do {                                       \
    (void)h; g_create_error = (msg);                    \
    return (code);                           \
  } while (0)
