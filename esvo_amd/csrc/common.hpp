// common.hpp — shared host/device definitions of the MI355X ESVO hot path (gfx950 only).
//
// Build flags (see __graft_entry__.build): hipcc --offload-arch=gfx950 -O3 -std=c++17
// -ffp-contract=off.  Contraction is off on purpose: the f64 stages restate the reference's
// arithmetic expression by expression so that results are reproducible against the CPU oracle.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/esvo_hip.h"

#define ESVO_WAVE 64

// A/B and test switches (ESVO_LM_PAIR, ESVO_FUSE_TILE_CAP, ...; tools/README.md lists them) are read from the environment ONLY
// when ESVO_DEV_SWITCHES=1 is set as well: a deployed library ignores stray variables.  tests/conftest.py and the tools set it.
#include <cstdlib>
#include <cstring>
inline const char* esvo_dev_switch(const char* name) {
  const char* on = std::getenv("ESVO_DEV_SWITCHES");
  return (on && std::strcmp(on, "1") == 0) ? std::getenv(name) : nullptr;
}

namespace esvo {

typedef unsigned long long u64;
typedef unsigned int u32;

// Rectified pin-hole model of one camera + the closed-form cam2World constants
// (PerspectiveCamera::cam2World / world2Cam, CameraSystem.cpp:121-148).
struct CamConst {
  double P[12];      // 3x4 projection, row-major
  double Kinv[9];    // inverse of P[:, :3]
  double Kinv_t[3];  // Kinv * P[:, 3]
};

// Parameters the kernels read (a by-value kernel argument: lives in SGPRs / kernarg segment).
// One row of per-tick device counters (esvo_context::d_counters): [0] n_matches [1] n_points [2] n_solved [3] n_fusion
// [4] n_records [5] n_map [6] touched cells [7] regulariser elements [8] own matches (sharded); from CNT_BM_FAIL on, three
// blocks of CNT_STRIPES partial sums: block matching failures by reason (info-noise ratio, coarse search, fine search).
constexpr int CNT_STRIPES = 16;
constexpr int CNT_BM_FAIL = 16;
constexpr int CNT_ROW = CNT_BM_FAIL + 3 * CNT_STRIPES;  // 64 words

struct DevParams {
  int W, H;
  int wx, wy;                    // patch size
  int dmin, dmax, step;          // effective disparity range; BM_step (1: dense; > 1: coarse-to-fine, EventBM.cpp:118-138)
  int updown;                    // BM_bUpDownConfiguration: the epipolar search runs along y (EventBM.cpp:178-186)
  double zncc_thr;               // BM_ZNCC_Threshold
  double baseline_f;             // baseline * P_left(0,0)
  double td_nu, td_scale, td_scale2, td_stdvar2;
  int lm_max_iter, lm_maxfev;
  double invdepth_min, invdepth_max;
  double var_thr;                // stdVar_vis_threshold^2
  double cost_thr;               // residual_vis_threshold^2 * patch area
  double age_thr;
  int fusion_radius;
  int reg_radius, reg_min_nb, reg_min_close;
  int num_threads;               // stride-N output permutation
  int ls_norm;                   // ESVO_LSNORM_TDIST / ESVO_LSNORM_L2 (DepthProblemConfig::LSnorm_)
  int band_y0, band_y1;          // row band owned by this handle (0,H when unsharded)
  int ev_shard, ev_nshards;      // per-event work (BM, LM) of slot w belongs to shard w % ev_nshards (balanced
                                 // whatever the scene; every rank holds the full Time Surfaces)
  int cband_y0, cband_y1;        // compute band of fusion / clean / regulariser view: the owned band plus a halo of
                                 // 2 rows (the 1-row side effects of displaced elements, Appendix A-7) and, with
                                 // regularisation, reg_radius more rows, so that the band's (2r+1)^2 neighbourhoods
                                 // are computed locally -- the DepthFrame is rebuilt from the window every tick,
                                 // so halo cells are recomputed, never exchanged
  CamConst camL, camR;
};

// ---- ros::Time arithmetic (roscpp) ---------------------------------------------------------
__host__ __device__ inline double time_to_sec(u32 sec, u32 nsec) { return (double)sec + 1e-9 * (double)nsec; }
__host__ __device__ inline double ns_to_sec(u64 ns) {
  return time_to_sec((u32)(ns / 1000000000ull), (u32)(ns % 1000000000ull));
}
__host__ __device__ inline double duration_to_sec(u64 later_ns, u64 earlier_ns) {
  long long d = (long long)(later_ns - earlier_ns);
  long long sec = d / 1000000000ll, nsec = d % 1000000000ll;
  if (nsec < 0) { nsec += 1000000000ll; sec -= 1; }
  return (double)sec + 1e-9 * (double)nsec;
}

// ---- stride-N thread emulation (EventBM.cpp:289-308, DepthProblemSolver.cpp:75-90) ---------
// The reference hands item i to thread i % T and concatenates the per-thread outputs.  Slot w
// of that concatenation holds item stride_item(w, n, T).
__host__ __device__ inline u32 stride_item(u32 w, u32 n, u32 T) {
  u32 off = 0;
  for (u32 t = 0; t < T; ++t) {
    u32 cnt = (n > t) ? (n - t + T - 1) / T : 0;
    if (w < off + cnt) return t + (w - off) * T;
    off += cnt;
  }
  return n;  // out of range
}
// inverse: the slot of item i
__host__ __device__ inline u32 stride_slot(u32 i, u32 n, u32 T) {
  u32 t = i % T, off = 0;
  for (u32 q = 0; q < t; ++q) off += (n > q) ? (n - q + T - 1) / T : 0;
  return off + i / T;
}

// ---- camera -------------------------------------------------------------------------------
__device__ inline void cam2World(const CamConst& c, double x, double y, double invDepth, double p[3]) {
  const double z = 1.0 / invDepth;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const double ray = (c.Kinv[r * 3 + 0] * x + c.Kinv[r * 3 + 1] * y) + c.Kinv[r * 3 + 2];
    p[r] = z * ray - c.Kinv_t[r];
  }
}
__device__ inline void world2Cam(const CamConst& c, const double p[3], double& u, double& v) {
  double h[3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
    h[r] = ((c.P[r * 4 + 0] * p[0] + c.P[r * 4 + 1] * p[1]) + c.P[r * 4 + 2] * p[2]) + c.P[r * 4 + 3];
  u = h[0] / h[2];
  v = h[1] / h[2];
}

// 4x4 row-major helpers (same association order as the oracle's mat4_mul / rigid_inverse)
__host__ __device__ inline void mat4_mul(const double* a, const double* b, double* c) {
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j)
      c[i * 4 + j] = ((a[i * 4 + 0] * b[0 * 4 + j] + a[i * 4 + 1] * b[1 * 4 + j]) + a[i * 4 + 2] * b[2 * 4 + j]) +
                     a[i * 4 + 3] * b[3 * 4 + j];
}
__host__ __device__ inline void rigid_inverse(const double* a, double* c) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) c[i * 4 + j] = a[j * 4 + i];
  for (int i = 0; i < 3; ++i)
    c[i * 4 + 3] = -((c[i * 4 + 0] * a[3] + c[i * 4 + 1] * a[7]) + c[i * 4 + 2] * a[11]);
  c[12] = c[13] = c[14] = 0.0;
  c[15] = 1.0;
}

// Dense DepthMap cell (one per pixel).  `flags` bit0: element alive; bit1: grid pointer valid
// (SmartGrid's pointer grid, see kernels_fuse.hip); row/col are the coordinates the element
// BELIEVES it has (they differ from the true cell only after DepthFusion's replace branch,
// SURVEY Appendix A-7).
struct MapCell {
  double x[2];
  double inv_depth, scale2, nu, variance, residual;
  double p_cam[3];
  u64 age;
  u32 row, col;
  u32 seq;     // creation order (id of the record that created the element)
  u32 unused_; // (the flags lived here until round 3)
};
static_assert(sizeof(MapCell) == 104, "MapCell layout");
enum { CELL_ALIVE = 1u, CELL_GRID = 2u };
// The cells' flags are a dense u32 array BEHIND the cell array, in the same allocation (a DepthMap buffer is ncell cells +
// ncell words): the passes that visit every cell -- the untouched-cell reset, clean, the regulariser's view and tile set-up,
// the export -- read 4 contiguous bytes per cell instead of one word of every 104-byte record (32 MB of lines for a 640x480
// map that holds 12 k elements on a reference-faithful tick).
__host__ __device__ inline u32* map_flags(MapCell* map, int ncell) { return reinterpret_cast<u32*>(map + ncell); }
__host__ __device__ inline const u32* map_flags(const MapCell* map, int ncell) { return reinterpret_cast<const u32*>(map + ncell); }
constexpr size_t map_buffer_bytes(size_t ncell) { return ncell * (sizeof(MapCell) + sizeof(u32)); }

// Window point (what DepthFusion::update reads from a stored DepthPoint)
typedef esvo_depth_point_t DevPoint;

// ---- launchers (one per kernel file) --------------------------------------------------------
// scan.hip
void launch_exclusive_scan_u32(const u32* d_in, u32* d_out, u32* d_total, u32* d_block_sums, size_t n,
                               hipStream_t s);
// the same over bit 0 of one byte per element; d_zero (nullable): n words cleared by the same launches
void launch_exclusive_scan_code_bit0(const uint8_t* d_codes, u32* d_out, u32* d_total, u32* d_block_sums, size_t n, u32* d_zero,
                                     hipStream_t s);
bool scan_is_small(size_t n);   // one single-workgroup launch does the whole scan
u32 scan_tiles(size_t n);       // tiles (= sums) of the two-launch scan; SCAN_TILE_SLOTS elements each
static constexpr u32 SCAN_TILE_SLOTS = 2048;
void launch_scan_down_code_bit0(const uint8_t* d_codes, u32* d_out, u32* d_total, const u32* d_tile_sums, size_t n, u32* d_zero, hipStream_t s);
size_t scan_scratch_elems(size_t n);
// scan + stable compaction in ONE single-workgroup launch, for inputs of at most 10 240 flags (scan.hip)
bool scan_compact_is_small(size_t n);
void launch_scan_compact_matches_small(const u32* flags, u32* prefix, u32* d_total, size_t n, const esvo_match_t* slots,
                                       esvo_match_t* out, u32* slot_of, hipStream_t s);
// (row_host, nullable: the device counter row `d_total` belongs to -- row_src, row_n words -- lands in that pinned host row as well)
void launch_scan_compact_points_small(const u32* flags, u32* prefix, u32* d_total, size_t n, const esvo_depth_point_t* slots,
                                      esvo_depth_point_t* out, hipStream_t s, const u32* row_src = nullptr, u32* row_host = nullptr,
                                      u32 row_n = 0);
// upload of a small pinned host buffer by a kernel (never blocks the host; scan.hip)
void launch_upload_words(const void* pinned_src, void* d_dst, size_t bytes, hipStream_t s, u32* d_zero = nullptr, u32 n_zero = 0);
void launch_back_prologue(const void* pinned_src, void* d_dst, size_t bytes, const void* a_src, void* a_dst, size_t a_bytes,
                          const void* b_src, void* b_dst, size_t b_bytes, hipStream_t s, const u32* a_flags = nullptr,
                          const u32* a_prefix = nullptr, u32 a_slots = 0);

// An event that arrived OUT OF ORDER (its stamp below the newest stamp staged before it) keeps its sorted place in the ring -- the
// mapper's queue is insertion-sorted, esvo_Mapping.cpp:692-702 -- but never reaches the Time Surface: TimeSurface::eventsCallback
// inserts events_.back(), the newest event, in its stead (TimeSurface.cpp:412-422).  The library marks such
// an event in its own copy through the WHOLE last word of the record -- polarity byte 0x80 | polarity AND the three padding bytes
// set to a magic -- so that no caller byte can be read as the mark: a caller's polarity byte means ON whenever it is non-zero (0x80
// included), its padding is whatever the compiler left there, and the records of the in-order paths are copied as they are (no
// sanitising pass on the ingest stream: one was measured at -8 % on the PCIe-inclusive rate).  A caller's record is misread only
// if its polarity byte is 0x80 / 0x81 and its three padding bytes equal the magic.
constexpr unsigned EV_LATE = 0x80u;                 // in the polarity byte
constexpr unsigned EV_LATE_PAD = 0x5ac3e7u;         // in _pad[0..2] (bits 8..31 of the record's last word)
__host__ __device__ inline bool ev_is_late(unsigned w) { return (w & 0xfffffffeu) == ((EV_LATE_PAD << 8) | EV_LATE); }
// kernels_ts.hip
void launch_ts_merge(const esvo_event_t* staged, const esvo_event_t* packet, const u32* plan, size_t n, esvo_event_t* ring, u64 first_slot,
                     u64 ring_cap, hipStream_t s);
void launch_ts_unpack_wire(const uint8_t* wire, size_t n, esvo_event_t* ring, u64 first_slot, u64 ring_cap, hipStream_t s);
void launch_ts_scatter(const esvo_event_t* d_ev, size_t n, u64* d_sae, int W, int H, hipStream_t s);
// (row0, row1: the rectified rows to render -- whole tiles of TS_TILE_ROWS; a routed band handle renders its band + halo only)
#define TS_TILE_ROWS 16
void launch_ts_render(const u64* d_sae, const int2* d_fixmap, uint8_t* d_raw, uint8_t* d_out, int W, int H,
                      u64 t_ns, double decay_sec, int ignore_polarity, int median_k, hipStream_t s, int row0 = 0, int row1 = -1);
void launch_gaussian5(const uint8_t* d_in, uint8_t* d_out, int W, int H, hipStream_t s, int row0 = 0, int row1 = -1);
// both cameras per launch (esvo_map_tick_resident): scatter segments, decay + median/remap, blur
struct TsScatterSegs { const esvo_event_t* ev[4]; size_t n[4]; u64* sae[4]; };
// per-pixel event queues (kernels_ts.hip): up to two ring segments of ONE camera per insertion
#define TSQ_LMAX 32
struct TsQueueArgs {
  const esvo_event_t* ev[2]; size_t n[2];
  u64* q;                       // [L][W * H] the camera's key sets, slot-major; 0 = empty slot
  int L;
  u32* tcount; uint4* tlist; u32 tcap;  // [tiles], [tiles][tcap] (key lo, key hi, pixel in tile, -)
  uint4* over; u32* over_count; u32 over_cap;  // (key lo, key hi, pixel in tile, tile)
};
void launch_tsq_insert(const TsQueueArgs& a, int W, int H, hipStream_t s);
void launch_tsq_view(const u64* q, int L, int W, int H, u64 t_ns, u64* sae, hipStream_t s);
struct TsPair { const u64* sae[2]; uint8_t* raw[2]; const int2* fixmap[2]; uint8_t* out[2]; uint8_t* out2[2]; };
void launch_ts_scatter_segs(const TsScatterSegs& g, int n_seg, int W, int H, hipStream_t s);
void launch_ts_render_pair(const TsPair& c, int W, int H, u64 t_ns, double decay_sec, int ignore_polarity, int median_k,
                           hipStream_t s, int row0 = 0, int row1 = -1);
void launch_ts_render_forward(const u64* d_sae, const u32* d_off, const u32* d_src, const float2* d_lut, double* d_val,
                              uint8_t* d_raw, uint8_t* d_out, int W, int H, u64 t_ns, double decay_sec, int ignore_polarity,
                              int median_k, hipStream_t s);
void launch_gaussian5_pair(const uint8_t* in0, const uint8_t* in1, uint8_t* out0, uint8_t* out1, int W, int H, hipStream_t s,
                           int row0 = 0, int row1 = -1);
// createDenoisingMask + extractDenoisedEvents (esvo_Mapping.cpp:1046-1072) on the n selected events
void launch_denoise_flags(const esvo_event_t* ring, u64 first, u64 cap, u32 n, uint8_t* evmap, u32* flags, int W, int H,
                          hipStream_t s);
void launch_denoise_select(const u32* flags, const u32* prefix, u32 n, u32* sel, hipStream_t s);
// routed band mode (api_map.hip, routed_denoise_begin / _resume): the rank's bits of the kept flags, one per walk position
void launch_denoise_bits_routed(const esvo_event_t* ring, u64 first, u64 cap, u32 n_loc, const u32* gidx, u32 g_first, u32 n, uint8_t* evmap,
                                int W, int H, int band_y0, int band_y1, u32* bits, hipStream_t s);
void launch_denoise_bits_unpack(const u32* blocks, u32 block_words, u32 N, u32 n, u32* flags, hipStream_t s);

// kernels_bm.hip
struct BmArgs {
  const esvo_event_t* ev;   // event buffer (ring of ev_cap slots)
  u32 n;                    // number of events handed to BM this tick
  u64 ev_first;             // absolute index of tick event 0
  u64 ev_cap;               // ring capacity (slot = absolute index % ev_cap)
  int ev_reverse;           // 1: newest-first walk (dataTransferring): event k = ev_first - k; 0: ev_first + k
  const u32* sel;           // optional indirection (denoised events): tick event k is walk position sel[k]
  const uint8_t* tsL;
  const uint8_t* tsR;
  const float2* lut;
  const uint8_t* mask;      // may be null
  const double* pose_sec;   // toSec() of the pose stamps
  u32 n_pose;
  esvo_match_t* out_slots;  // [n] slot w (thread-stride order)
  u32* out_flags;           // [n]
  u32* fail_counters;       // nullable: the tick's counter row (CNT_BM_FAIL: per-reason failures, EventBM.h:89)
  // routed band mode (kernels_bm.hip, bm_item): `ev` is the rank's OWN ring (the events of its image rows); the launch walks
  // its n_loc newest-first events from ev_first, gidx[slot] is each event's index in the global sequence (low 32 bits) and
  // g_first that of the tick's newest selected event; n stays the GLOBAL selection size; results are indexed by walk position
  const u32* gidx = nullptr;
  u32 g_first = 0;
  u32 n_loc = 0;
  // ... with Denoising: the walk positions are those of the RAW selection (n_raw events); keep_flags says which are kept,
  // keep_prefix how many kept ones precede -- the event's position in the kept sequence, which is what n counts
  const u32* keep_flags = nullptr;
  const u32* keep_prefix = nullptr;
  u32 n_raw = 0;
};
void launch_bm_match(const BmArgs& a, const DevParams& p, hipStream_t s);
void launch_compact_matches(const esvo_match_t* slots, const u32* flags, const u32* prefix, u32 n,
                            esvo_match_t* out, u32* slot_of, hipStream_t s);
void launch_matches_to_points(const esvo_match_t* m, const u32* n_ptr, u32 max_n, esvo_depth_point_t* out, const DevParams& p,
                              hipStream_t s);

// kernels_track.hip: tracker residual / Jacobian evaluation (RegProblemLM.cpp), SURVEY.md section 8(f).1
struct TrackRef { double T[16]; };            // T_world_ref
struct TrackPose { double T[16]; double Jc[6]; };  // T_left_ref; J_constPart (3x2, row-major) for the Jacobian
#define TRK_NE_MAX_POSES 4   // poses one launch of track_normal_kernel evaluates (one workgroup each): the trial steps of an LM iteration
struct TrackPoseSet { TrackPose p[TRK_NE_MAX_POSES]; };
struct TrackArgs {
  const double* pts;      // [n][3] points in the reference camera frame
  const uint8_t* neg;     // TS_negative_left_
  const int16_t* du;      // dTS_negative_du_left_
  const int16_t* dv;      // dTS_negative_dv_left_
  const uint8_t* mask;    // UndistortRectify_mask_ (or nullptr)
  double P[12];           // left projection matrix
  int W, H;
};
void launch_track_images(const uint8_t* blurred, uint8_t* neg, int16_t* du, int16_t* dv, int W, int H, hipStream_t s);
void launch_track_reference(const float* xyz, u32 n, const TrackRef& r, double* pts, hipStream_t s);
void launch_track_residuals(const TrackArgs& a, const TrackPose& pose, u32 offset, u32 count, int huber, double thr, double* fvec,
                            hipStream_t s);
void launch_track_jacobian(const TrackArgs& a, const TrackPose& pose, u32 offset, u32 count, double* fjac, hipStream_t s);
#define TRK_NE_THREADS 256
#define TRK_NE_TERMS 28   // 21 upper-triangle entries of J^T J, 6 of J^T f, |f|^2
void launch_track_normal(const TrackArgs& a, const TrackPoseSet& poses, int n_poses, u32 offset, u32 count, int huber, double thr,
                         double* out28_per_pose, hipStream_t s);

// kernels_shard.hip: ordering of a tick's frame from the ranks' (matched, kept) bits
void launch_shard_codes(const u32* own_w, const u32* keep, const u32* n_local, u32 max_local, u32 N, uint8_t* block, hipStream_t s);
void launch_shard_unpack_codes(const uint8_t* blocks, u32 block_bytes, u32 N, u32 n, uint8_t* codes, u32* rank_kept, hipStream_t s);
void launch_shard_keep_flags(const uint8_t* codes, const u32* prefix_f, const u32* n_matches, u32 n, u32 T, u32* keep_by_slot,
                             unsigned long long* cursor, hipStream_t s);
void launch_shard_pack(const u32* own_w, const u32* keep, const DevPoint* local_pts, const u32* n_local, u32 max_local,
                       const u32* prefix_f, const u32* n_matches, const u32* prefix_g, u32 T, unsigned long long* block, u32 block_cap,
                       u32 frame_cap, u32* rank_kept, u32 N, u32* max_kept_out, hipStream_t s, const u32* halo_viol = nullptr);
void launch_shard_scatter(const unsigned long long* blocks, size_t block_words, u32 N, u32 max_kept, DevPoint* frame, u32 frame_cap,
                          hipStream_t s, u32* viol_total = nullptr);
// routed band mode (two bits per slot of the whole tick)
void launch_shard_codes_routed(const esvo_match_t* own_matches, const u32* keep, const u32* n_local, u32 max_local, u32 n, u32 T, u32* own_w,
                               u32* block, hipStream_t s);
void launch_shard_unpack_routed(const u32* blocks, u32 block_words, u32 N, u32 n, uint8_t* codes, u32* rank_kept, u32* tile_sums, hipStream_t s);

// kernels_lm.hip
bool lm_launch_is_wide(u32 max_matches, const struct DevParams& p);  // kernels_lm.hip
struct LmArgs {
  const esvo_match_t* matches;  // compacted vEMP (thread-stride order of BM)
  const u32* n_matches;         // device count
  u32 max_matches;
  const uint8_t* tsL;
  const uint8_t* tsR;
  const double* pose_T;         // [n_pose][16] T_world_virtual
  double T_world_obs[16];       // by value: no upload per tick
  DevPoint* out_slots;          // [max_matches] slot s (thread-stride order of the solver)
  u32* out_flags;               // [max_matches] 1 = solved (and kept when cull)
  int cull;
  int dense;                    // sharded mode: `matches` is this rank's own dense list; slot s solves match s
  // scratch of the split launch (kernels_lm.hip, LmSplit); split_fvec0 == nullptr: single launch
  double* split_fvec0;
  double* split_fnorm0;
  u32* split_meta;
  u32* split_order;
  u32* split_hist;
  int pair;                     // wide layout only: two waves per match (kernels_lm.hip "pair layout"); the caller's choice
  // wide layout only (latency mode, api_map.hip): `matches` is the block matcher's SLOT array and match_index[j] the slot of the
  // j-th match of the compacted list -- the list itself (48 B per record through one workgroup) is never written.  nullptr: `matches`
  // is the compacted list.
  const u32* match_index;
  // In-run shader-clock probe (nullable): lane 0 of every 65th workgroup reads s_memtime (shader cycles) and s_memrealtime
  // (the constant reference clock) when it starts and when it ends and adds the two differences to clk[2 xcc], clk[2 xcc + 1]
  // (xcc = the XCD the wave ran on), one sample to clk[16]; the start values wait in clk[CLK_SCRATCH + 2 * (block / 65) ...].
  // Sum of cycle differences / sum of reference differences x the reference rate = the clock the LM waves really ran at,
  // averaged over their lifetimes -- under the load of the whole tick, without a profiler attached.
  u64* clk;
  // routed band mode: the rows [vy0, vy1) of tsL / tsR hold data; matches whose evaluations read outside are counted here
  // (non-null selects the guarded kernels)
  u32* halo_viol = nullptr;
  int vy0 = 0, vy1 = 0;
  // the persistent narrow layout (kernels_lm.hip, lm_refine_persist_kernel): a zeroed work counter (non-null selects it for
  // launches of the throughput layout) and the number of workgroups the chip holds
  u32* persist_next = nullptr;
  u32 persist_blocks = 0;
};
constexpr u32 CLK_XCDS = 8, CLK_SAMPLES = 16, CLK_SCRATCH = 32, CLK_STRIDE = 65;
inline size_t clk_words(u32 max_ev) { return CLK_SCRATCH + 2 * ((size_t)max_ev / 64 + 2); }
constexpr u32 LM_PAIR_MAX_EVENTS = 10000u;       // launches bounded by more events never use the pair layout (2 waves per match)
constexpr u32 LM_TWO_QUEUES_MAX_EVENTS = 40000u;  // = LM_WIDE_MAX (kernels_lm.hip): launches that use the wide layout
constexpr u32 LM_SPLIT_MIN_EVENTS = 40000u;  // launches bounded by fewer events use the wide layout (LM_WIDE_MAX), never the split
void launch_lm_refine(const LmArgs& a, const DevParams& p, u32* n_solved, hipStream_t s);
void launch_compact_points(const DevPoint* slots, const u32* flags, const u32* prefix, const u32* n_in,
                           u32 max_n, DevPoint* out, hipStream_t s);

// kernels_fuse.hip
#define FUSE_TILE 8   // the fusion front sorts per FUSE_TILE x FUSE_TILE-cell tile
struct FuseArgs {
  const DevPoint* win;          // window point ring
  // frames in FUSION order (newest -> oldest): point q belongs to frame f with
  // fr_cum[f] <= q < fr_cum[f+1]; its record is win[fr_off[f] + (q - fr_cum[f])]
  const u32* fr_cum;            // [n_frames + 1]
  const u32* fr_off;            // [n_frames]
  const u32* fr_slot;           // [n_frames] pose-table slot
  u32 n_frames;
  u32 n_pts;
  const double* frame_pose_T;   // [n_pose_slots][max_poses][16]
  u32 max_poses;
  double T_frame_world[16];     // inverse of the depth frame's pose
  // scratch (kernels_fuse.hip, "the fusion front: tiles")
  DevPoint* prop;               // [n_pts] propagated points (row == 0xffffffff: rejected)
  u32* tile_count;              // [n_tiles] points appended per tile (may exceed tile_cap: readers clamp); zero between ticks
  uint2* tile_pts;              // [n_tiles][tile_cap] the points of each tile: (id, row << 16 | col)
  u32 tile_cap;
  uint2* over_pts;              // [n_pts] points whose tile list was full
  u32* over_count;
  u32* rec_ids;                 // record ids grouped by cell, each cell's in increasing order: [n_tiles][tile_rec] the tiles' own
  u32 tile_rec;                 //   regions, then [n_pts * K] for the tiles that outgrow theirs (rec_cursor)
  u32* rec_cursor;
  u32* cell_count;              // [W*H] records of the cell
  u32* cell_offset;             // [W*H] its first entry in rec_ids
  u32* cell_list;               // [16 classes][64 slices][slice_cap] touched cells by length class and tile slice
  u32 slice_cap;                //   = cells of the tiles of one slice
  u32* class_count;             // [16 * 64] cells per (class, slice), being filled
  u32* class_total;             // [16 * 64 + 1] their exclusive scan in walk order (fuse_turn_kernel)
  u32 lds_cap;                  // record ids a tile may order in LDS at a time (0: the maximum; smaller values: tests)
  u32 pmax_plus1;               // 0: default; else 1 + the largest candidate count that takes the tile kernel's fast path (tests)
  u32* d_total;                 // total records (a statistic)
  MapCell* map;                 // [W*H]
  u32* d_num_fusion;            // fusion counter
  u32* n_touched;               // number of touched cells (a statistic)
  int naive;                    // 1: DepthFusion::naive_propagation instead of update (esvo_MVStereo's PURE_BLOCK_MATCHING mode)
  u32* owner_max;               // regulariser scratch reset together with the per-cell counters (or nullptr)
  u32* owner_min;
  u32* n_reg_elems;
};
void launch_fuse(const FuseArgs& a, const DevParams& p, hipStream_t s);
void launch_clean(MapCell* map, const DevParams& p, hipStream_t s);
void launch_reg_view(const MapCell* map_in, MapCell* map_out, u32* owner_max, u32* owner_min, double2* ab, double2* cd,
                     u32* n_elems, const DevParams& p, hipStream_t s);
void launch_reg_apply(const MapCell* map_in, MapCell* map_out, const u32* owner_max, const u32* owner_min, const double2* ab,
                      const double2* cd, u32* n_elems, const DevParams& p, hipStream_t s, bool sparse = false);
// kernels_sgm.hip: semi-global matching of the Time-Surface pair + the Gaussian DepthPoints of InitializationAtTime
struct SgmScratch {
  uint8_t *sobL, *rawL, *sobR, *rawR;  // pre-filtered planes, W*H each
  int16_t* vol[6];                     // cost volumes [H][W - 48][48]: pixel cost / C, horizontal sums / L0, L1..L4
  int16_t *d1, *d1b;                   // raw disparities before / after the left-right check
  u32* d2key;                          // right-image candidates of the left-right check
};
void launch_sgbm(const uint8_t* left, const uint8_t* right, const SgmScratch& s, int16_t* disp, int W, int H, hipStream_t st);
void launch_sgm_points(const esvo_event_t* ring, u64 first, u64 cap, u32 n, const float2* lut, const int16_t* disp, DevPoint* slots,
                       u32* flags, const DevParams& p, hipStream_t st);
void launch_sgm_naive(const DevPoint* pts, u32 n, const double* d_T_frame_obs, u32* owner, u32* pair_flags, u32* pair_rank, u32* d_total,
                      u32* scan_tmp, MapCell* map, const DevParams& p, hipStream_t st);
// kernels_viz.hip
void launch_debug_image(const MapCell* map, u32* owner, uint8_t* bgr, const uint8_t* jet, int type, double max_range,
                        double min_range, double thr1, double thr2, const DevParams& p, hipStream_t s);
void launch_map_compact(const MapCell* map, u32* flags, u32* prefix, u32* d_total, u32* scan_tmp,
                        esvo_depth_point_t* out, u32* out_cell, const DevParams& p, hipStream_t s);

}  // namespace esvo
