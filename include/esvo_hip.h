/*
 * esvo_hip.h — C-ABI of the MI355X-native ESVO hot path (Time-Surface raster +
 * semi-dense stereo mapper).
 *
 * This header is the drop-in boundary.  The reference (ESVO) has
 * no FFI/plugin interface: its algorithms are C++ classes called directly by the
 * ROS nodes.  Every entry point below therefore names the reference *seam* it
 * stands behind (file:line relative to the ESVO repository) — the call a
 * maintainer replaces in esvo_time_surface/src/TimeSurface.cpp and
 * esvo_core/src/esvo_Mapping.cpp / esvo_MVStereo.cpp.
 *
 * Conventions
 *   - plain C, opaque handle, no exceptions, no exit(): every call returns
 *     ESVO_OK (0) or a negative esvo_status_t; esvo_last_error() gives text.
 *   - all input buffers are borrowed for the duration of the call only.
 *   - outputs go to caller-allocated arrays (capacity + returned count).
 *   - Threads.  Distinct handles are independent.  On ONE handle the calls form three groups that may run
 *     concurrently, one thread per group -- the threading of the reference's nodes (esvo_Mapping.cpp:160,179-247: the
 *     MappingLoop worker thread; :669-703 and TimeSurface.cpp:403-425: eventsCallback on the ROS spinner under
 *     data_mutex_; esvo_Tracking's own loop):
 *       INGEST   esvo_ts_push_events(_async), esvo_ts_push_wait, esvo_ts_push_event_array, esvo_ts_push_bag (one thread per camera is fine)
 *       TRACKER  esvo_track_*
 *       MAPPER   every other call that takes the handle (renders, observation, ticks, stage-wise calls, outputs,
 *                parameters, esvo_comm_*, esvo_reset -- which excludes the other two groups while it runs)
 *     Calls of the same group are serialised by the library (safe from several threads, never concurrent).  The node's
 *     data_mutex_ is therefore NOT needed around these calls.  What the caller still owns is causality: a render or
 *     tick at time t sees the events staged when it is called, exactly like the reference's TS_history_/events_left_
 *     snapshot (dataTransferring) -- stage the events up to t before asking for t.  An ingest call never waits for a
 *     running tick (own HIP stream; the ring bookkeeping is locked for microseconds); the tracker calls run on a HIP
 *     stream of their own and wait only for the render that produced the Time Surface they read.
 *     esvo_last_error returns the message of the calling thread's last failed call.
 *   - every call is synchronous w.r.t. its host-visible outputs; calls without
 *     host outputs only enqueue work on the handle's HIP stream.
 *   - pointers named d_* are DEVICE pointers (HBM), everything else is host.
 *   - there is no CPU fallback: esvo_create() fails with ESVO_ERR_NO_DEVICE
 *     when no gfx950 device is visible.
 *   - Scheduling is the library's business and never changes a result: how many HIP queues a tick's stages use, which
 *     layout the refinement kernel runs in, whether two refinement launches are in flight -- the handle decides from the
 *     sizes of the launches and from its own stage timings (HIP events).  The ESVO_* environment variables read at
 *     esvo_create (tools/README.md lists them) pin those choices for measurements; none of them alters an output bit.
 *     ABI 8: the handle tells a tick that runs ALONE (its predecessor was completed and read before it was handed in -- the ROS
 *     node's pattern, esvo_Mapping.cpp:261-431 once per MappingLoop turn) from ticks that OVERLAP (a throughput loop that hands in
 *     tick k + 1 while tick k is in flight).  The first kind takes the latency path: one queue for the front stage and the LM
 *     launch, no stage-timing event between dependent kernels except on sampled ticks (esvo_stats_t.stage_timing_samples), the
 *     tick's frame compacted straight into the fusion window, polled host waits (the calling thread spins for up to 3 ms instead
 *     of sleeping on the completion interrupt) -- 0.54 instead of 0.68 ms for a DSEC tick of 10 000 events, same bits.
 */
#ifndef ESVO_HIP_H
#define ESVO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ESVO_HIP_ABI_VERSION 8

typedef enum esvo_status_t {
  ESVO_OK = 0,
  ESVO_ERR_INVALID_ARG = -1,
  ESVO_ERR_NO_DEVICE = -2,
  ESVO_ERR_HIP = -3,
  ESVO_ERR_CAPACITY = -4,
  ESVO_ERR_UNSUPPORTED = -5,
  ESVO_ERR_STATE = -6,
  ESVO_ERR_HALO = -7,  /* routed band mode: a refinement read outside the rank's Time-Surface rows (esvo_shard_set_routing) */
  ESVO_AGAIN = 1       /* not an error -- esvo_shard_tick_phase(h, 0, ...) on a routed handle with Denoising: all-gather the block of
                          esvo_shard_exchange (one bit per selected event: the rank's share of the denoising mask's verdicts), then call
                          phase 0 again (its arguments are ignored the second time) */
} esvo_status_t;

typedef struct esvo_context* esvo_handle;

enum { ESVO_CAM_LEFT = 0, ESVO_CAM_RIGHT = 1 };
enum { ESVO_FUSION_CONST_FRAMES = 0, ESVO_FUSION_CONST_POINTS = 1 };
enum { ESVO_LSNORM_TDIST = 0, ESVO_LSNORM_L2 = 1 /* Gaussian model; no shipped config uses it */ };

/* Layout-identical to the in-memory dvs_msgs::Event (uint16 x, uint16 y,
 * ros::Time{uint32 sec, uint32 nsec}, bool polarity; sizeof == 16), so ROS glue
 * passes msg->events.data() without a copy. */
typedef struct esvo_event_t {
  uint16_t x, y;
  uint32_t sec, nsec;
  uint8_t polarity;
  uint8_t _pad[3];
} esvo_event_t;

/* Per-camera calibration products.  They are what the reference's
 * PerspectiveCamera::preComputeRectifiedCoordinate (CameraSystem.cpp:36-111) and
 * TimeSurface::cameraInfoCallback (TimeSurface.cpp:313-401) compute once with
 * OpenCV; the caller passes them in (ROS glue: straight from its CameraSystem;
 * ROS-free harness: esvo_amd/calib.py). */
typedef struct esvo_calib_t {
  int32_t width, height;
  double P[12];              /* rectified 3x4 projection matrix, row-major (CameraSystem.cpp:31) */
  const float* rect_lut;     /* [H*W*2] raw pixel -> rectified (x,y), float32 as returned by
                                cv::undistortPoints (CameraSystem.cpp:62,106-110) */
  const uint8_t* rect_mask;  /* [H*W] 0/255 UndistortRectify_mask_ (CameraSystem.cpp:67-72);
                                may be NULL for the right camera */
  const float* map_x;        /* [H*W] undistort_map1_ CV_32FC1 (TimeSurface.cpp:341-351) */
  const float* map_y;        /* [H*W] undistort_map2_ */
} esvo_calib_t;

/* One POD carrying every yaml key the hot path reads. */
typedef struct esvo_params_t {
  /* esvo_time_surface (cfg/time_surface/ts_parameters.yaml; TimeSurface.cpp:23-30) */
  double decay_ms;                 /* 30 */
  int32_t median_blur_kernel_size; /* k: medianBlur(2k + 1); 1 -> 3x3 (every shipped config); 0 disables; 0..3 */
  int32_t ignore_polarity;         /* 1 */
  /* DepthProblemConfig (DepthProblem.h:15-51) */
  int32_t patch_size_x;            /* 15 in every shipped config (register-layout kernels); any 1..64 otherwise */
  int32_t patch_size_y;            /* 7 in every shipped config; any 1..40 otherwise */
  int32_t ls_norm;                 /* ESVO_LSNORM_TDIST */
  double td_nu;                    /* Tdist_nu */
  double td_scale;                 /* Tdist_scale */
  int32_t lm_max_iteration;        /* ITERATION_OPTIMIZATION (10); maxfev = 3x */
  int32_t reg_radius;              /* RegularizationRadius */
  int32_t reg_min_neighbours;      /* RegularizationMinNeighbours */
  int32_t reg_min_close_neighbours;/* RegularizationMinCloseNeighbours */
  /* EventBM (EventBM.cpp:34-54).  min/max are the EFFECTIVE range after the
   * depth-range clamp of esvo_Mapping.cpp:110-116 (esvo_amd/params.py does it). */
  int32_t bm_min_disparity;
  int32_t bm_max_disparity;
  int32_t bm_step;                 /* 1 in every shipped config; > 1: coarse-to-fine search (EventBM.cpp:118-138) */
  double bm_zncc_threshold;        /* 0.1 */
  int32_t bm_updown;               /* BM_bUpDownConfiguration: vertical epipolar search (EventBM.cpp:178-186) */
  int32_t smooth_time_surface;     /* SmoothTimeSurface: 5x5 Gaussian before BM (DSEC) */
  /* mapping node (esvo_Mapping.cpp:63-99) */
  double invdepth_min, invdepth_max;
  double stdvar_vis_threshold;     /* sigma; culling/clean compare against its square */
  double residual_vis_threshold;   /* cost threshold = thr^2 * patch area */
  double age_vis_threshold;
  int32_t fusion_radius;           /* 0 -> 2x2, !=0 -> 3x3 (DepthFusion.cpp:98-117) */
  int32_t fusion_strategy;         /* ESVO_FUSION_CONST_FRAMES / _CONST_POINTS */
  int32_t max_fusion_frames;
  int32_t max_fusion_points;
  int32_t clean_requires_full_window; /* 1: esvo_Mapping.cpp:385; 0: esvo_MVStereo.cpp:496 */
  int32_t regularization;          /* Regularization */
  int32_t denoising;               /* Denoising: event-map median mask on the selected events
                                      (esvo_Mapping.cpp:282-296,1046-1072); applied by esvo_map_tick */
  int32_t process_event_num;       /* PROCESS_EVENT_NUM (events block-matched per tick) */
  double bm_half_slice_thickness;  /* 0.001 s; event window = 10x (esvo_Mapping.cpp:563) */
  int32_t num_threads;             /* NUM_THREAD_MAPPING (4): reproduces the stride-N output
                                      permutation of EventBM.cpp:289-308 and
                                      DepthProblemSolver.cpp:75-90 */
  /* capacities (device allocations; 288 GB of HBM makes generous defaults cheap) */
  int32_t max_events_per_tick;     /* >= process_event_num */
  int32_t max_window_points;       /* total DepthPoints the fusion window may hold */
  int32_t max_poses_per_tick;      /* >= 201 */
  int64_t event_ring_capacity;     /* staged events per camera */
  /* esvo_time_surface: max_event_queue_len (TimeSurface.cpp:30, default 20).  0 (default here): ONE stamp per pixel -- the
   * fast path; render times must not decrease and the queue's eviction artefact (a render older than max_event_queue_len
   * newer events of a pixel reads that pixel empty, TimeSurface.h:39-75) is not reproduced.  1..32: EventQueueMat semantics
   * exactly -- every staged event enters its pixel's queue at the next render, renders at any time in any order. */
  int32_t max_event_queue_len;
  int32_t pad_params_;
} esvo_params_t;

/* What flows EventBM -> DepthProblemSolver.  The reference's EventMatchPair
 * (EventMatchPair.h:16-38) carries a full pose; consumers read only x_left_,
 * trans_, invDepth_ (DepthProblemSolver.cpp:92-94) and cost_/disp_ in BM-only mode. */
typedef struct esvo_match_t {
  double x_left[2];   /* rectified left coordinate */
  double inv_depth;   /* disparity / (baseline * P(0,0)) */
  double cost;        /* ZNCC cost of the best candidate */
  double disp;
  uint32_t event_idx; /* index into the events passed to the call */
  uint32_t pose_idx;  /* index into the tick's pose table (first stamp >= event ts) */
} esvo_match_t;

/* Every field of DepthPoint (DepthPoint.h:70-88); the 4x4 T_world_cam_ is
 * carried as an index into the pose table of the frame the point came from. */
typedef struct esvo_depth_point_t {
  uint32_t row, col;
  double x[2];
  double inv_depth;
  double scale2;
  double nu;
  double variance;
  double residual;
  uint64_t age;
  double p_cam[3];
  uint32_t pose_idx;
  uint32_t seq;       /* DepthMap outputs: creation order in the reference's element list */
} esvo_depth_point_t;

/* Stage timings use the reference's own stage boundaries (esvo_Mapping.cpp:405-430). */
typedef struct esvo_stats_t {
  uint64_t ticks;
  uint64_t events_staged[2];
  uint64_t events_scattered[2];
  uint64_t ts_frames[2];
  uint32_t last_events_in;      /* events handed to block matching in the last tick */
  uint32_t last_matches;        /* BM successes */
  uint32_t last_solved;         /* LM problems solved */
  uint32_t last_points;         /* after pointCulling */
  uint32_t last_window_frames;
  uint32_t last_window_points;
  uint32_t last_fusions;        /* DepthFusion::update return, summed over the window */
  uint32_t last_map_size;       /* DepthMap::size() after clean/regularisation */
  float ms_ts_scatter, ms_ts_render;
  float ms_bm, ms_refine, ms_fusion, ms_regularization, ms_tick_total;
  /* HIP-event time of single kernels / stages of the last tick (for roofline accounting):
   * [0] ts_scatter [1] ts_decay+median_remap [2] bm_match [3] lm_refine
   * [4] propagate+bucket+fuse_cells [5] clean [6] regularize [7] reserved.
   * The matching stage ([0]-[2]), the LM stage ([3]) and the fusion stage ([4]-[6]) of consecutive ticks run on three
   * streams at the same time, so the values are not additive and each includes the slowdown from the other streams'
   * kernels. */
  float ms_kernel[8];
  float pad_;
  /* Running totals over all ticks since esvo_create / esvo_reset.  A throughput loop reads them once at its
   * end: esvo_get_stats drains the streams, so calling it after every tick serialises the overlap of one
   * tick's LM and fusion stages with the next tick's matching stage. */
  uint64_t total_events_in;
  uint64_t total_matches;
  uint64_t total_points;
  double sum_ms_kernel[8];      /* same slots as ms_kernel; [7] = launches of ts kernels summed in [0],[1] */
  /* EventBM's failure counters by reason (EventBM.h:89; incremented at EventBM.cpp:107, :124, :135), of the last block
   * matching and summed over the ticks: left patch with > 95 % of its pixels below 1, coarse search without a match
   * (with BM_step 1: no candidate below the ZNCC threshold), fine search without a match. */
  uint32_t last_bm_info_noise_low, last_bm_coarse_fail, last_bm_fine_fail, pad2_;
  uint64_t total_bm_info_noise_low, total_bm_coarse_fail, total_bm_fine_fail;
  /* ABI 3 -- the shader clock the refinement kernel really ran at, measured inside the run (no profiler): lane 0 of every
   * 65th workgroup of the kernel reads s_memtime (shader cycles) and s_memrealtime (constant reference clock, clk_ref_khz)
   * when it starts and when it ends; the differences are summed per XCD since esvo_create / esvo_reset.
   * clock [MHz] of XCD x = clk_cycles[x] / clk_ref_ticks[x] * clk_ref_khz / 1000; all XCDs: the ratio of the sums. */
  uint64_t clk_cycles[8];
  uint64_t clk_ref_ticks[8];
  uint64_t clk_samples;
  uint32_t clk_ref_khz;
  /* ABI 8 -- how many ticks contributed to ms_bm .. ms_regularization, ms_kernel[2..6] and sum_ms_kernel[2..6].  Stage timings come
   * from HIP events recorded between the stages, and every such record costs the queue about 5 us (the next dispatch waits for
   * the marker; kernels with no event between them follow each other without a gap).  So stage timings are SAMPLED where those
   * gaps are on the path somebody waits for: a tick that runs ALONE -- the caller read the previous tick's result before handing it
   * in, as the ROS node does -- records its events on the handle's first 8 such ticks and on one in 31 afterwards; small ticks that
   * overlap (at most 40 000 events: the host's enqueueing is their pace) on one tick in four.  Large overlapping ticks (the
   * throughput path, paced by the LM kernel) and the own ticks of a tick-interleaved multi-GPU rank record every tick.  ms_* hold the latest sample, sum_ms_kernel[2..6] the sum
   * over the samples, the mean per tick is sum / stage_timing_samples.  ms_ts_* / sum_ms_kernel[0..1] are sampled the same way with
   * their own count in [7]. */
  uint32_t stage_timing_samples;
  /* ABI 6 -- routed band mode (esvo_shard_set_routing): matches, summed over ALL ranks and the ticks since esvo_create /
   * esvo_reset, whose refinement read rows of the Time Surfaces this rank does not render.  Non-zero: the handle refuses
   * further ticks with ESVO_ERR_HALO. */
  uint64_t halo_violations;
  /* Events that arrived out of order (stamp below the newest stamp staged before them), per camera: sorted into the ring for
   * the mapper, withheld from the Time Surface as the reference's eventsCallback withholds them (esvo_ts_push_events). */
  uint64_t late_events[2];
  /* How often the tick pipeline let its back stream drain to get out of its slow operating point (api_map.hip, pipeline_resync):
   * 0 in an undisturbed run. */
  uint64_t pipeline_resyncs;
} esvo_stats_t;

/* ---- lifecycle -------------------------------------------------------------------- */

/* Fill *p with the reference's code defaults (esvo_Mapping.cpp:37-99, TimeSurface.cpp:23-30). */
void esvo_default_params(esvo_params_t* p);

/* Replaces: TimeSurface ctor + cameraInfoCallback (TimeSurface.cpp:12-50,313-401) and the
 * esvo_Mapping ctor's CameraSystem/EventBM/DepthProblemSolver/DepthFusion setup
 * (esvo_Mapping.cpp:26-128).  `device` is the HIP device ordinal. */
int esvo_create(const esvo_params_t* params, const esvo_calib_t* left, const esvo_calib_t* right,
                int device, esvo_handle* out);
int esvo_destroy(esvo_handle h);
/* Replaces esvo_Mapping::reset (esvo_Mapping.cpp:764-804): clears SAE, event rings, window, map. */
int esvo_reset(esvo_handle h);
/* Replaces EventBM::resetParameters (EventBM.cpp:34-54) + onlineParameterChangeCallback
 * (esvo_Mapping.cpp:806-866).  Capacities and image size cannot change. */
int esvo_set_params(esvo_handle h, const esvo_params_t* params);
const char* esvo_last_error(esvo_handle h); /* h may be NULL for create-time errors */
/* Run the handle's kernels on an external HIP stream (e.g. torch's current stream). */
int esvo_set_stream(esvo_handle h, void* hip_stream);
int esvo_synchronize(esvo_handle h);

/* ---- Time Surface ------------------------------------------------------------------ */

/* Replaces TimeSurface::eventsCallback + EventQueueMat::insertEvent (TimeSurface.cpp:403-425,
 * TimeSurface.h:39-50) and, for the left camera, esvo_Mapping::eventsCallback
 * (esvo_Mapping.cpp:669-703).  Events must be time-sorted (Appendix A-1). */
int esvo_ts_push_events(esvo_handle h, int cam, const esvo_event_t* ev, size_t n);
/* The same without waiting for the copy: the call returns once the host-to-device copy of `ev` is ENQUEUED on the ingest stream
 * (time stamps and order are checked during the call).  `ev` must stay valid and unchanged until esvo_ts_push_wait(h, cam) has
 * returned.  Meant for PINNED buffers (esvo_host_alloc below, or the node's own hipHostRegister'ed message pool): from pinned
 * memory the copy is a DMA that overlaps the running tick, so a node that stages every tick's events pays no PCIe time on
 * its mapping thread; from pageable memory the runtime stages the copy itself and the call is as synchronous as
 * esvo_ts_push_events.  Renders and ticks that read the events are ordered behind the copy on the device (an event wait on the
 * handle's front stream), never by a host wait. */
int esvo_ts_push_events_async(esvo_handle h, int cam, const esvo_event_t* ev, size_t n);
int esvo_ts_push_wait(esvo_handle h, int cam);
/* Pinned host memory for such buffers (hipHostMalloc / hipHostFree behind the C-ABI; no handle needed). */
int esvo_host_alloc(size_t bytes, void** out);
int esvo_host_free(void* p);
/* The same, straight from the ROS1 wire format: `msg` is one serialised
 * dvs_msgs/EventArray (Header, u32 height, u32 width, u32 count, count x 13-byte dvs_msgs/Event: u16 x, u16 y,
 * u32 sec, u32 nsec, u8 polarity) as a rosbag chunk or a TCPROS connection delivers it.  The packed records are copied to
 * the device as they are (13 instead of 16 B/event over PCIe) and widened into the ring by a kernel; no
 * std::vector<dvs_msgs::Event> is materialised (rosbag::MessageInstance::instantiate does that at
 * events_repacking_helper/src/EventMessageEditor.cpp:111).  *n_events (nullable) receives the message's event count. */
int esvo_ts_push_event_array(esvo_handle h, int cam, const uint8_t* msg, size_t n_bytes, size_t* n_events);
/* rosbag (format 2.0) ingest, SURVEY.md §8(f).2: the reading side of events_repacking_helper
 * (events_repacking_helper/src/EventMessageEditor.cpp:66-119: rosbag::Bag::open, rosbag::View over an event topic,
 * MessageInstance::instantiate<dvs_msgs::EventArray>).  Chunks (compression none / bz2 / lz4) are walked in file order;
 * each serialised dvs_msgs/EventArray of the topic is returned as a byte range of the reader's buffer (valid until the next
 * call) -- the input of esvo_ts_push_event_array.  Returns ESVO_OK, 1 at the end of the bag, or a negative status
 * (esvo_bag_last_error).  topic == NULL: every topic of type dvs_msgs/EventArray.  Host code, no GPU needed. */
typedef struct esvo_bag* esvo_bag_handle;
int esvo_bag_open(const char* path, esvo_bag_handle* out);
int esvo_bag_close(esvo_bag_handle b);
const char* esvo_bag_last_error(esvo_bag_handle b);
int esvo_bag_next_event_array(esvo_bag_handle b, const char* topic, const uint8_t** msg, size_t* n_bytes, uint64_t* stamp_ns,
                              const char** topic_out);
/* Stage the messages of `topic` with a bag time stamp below until_ns (0: all that are left) for camera `cam`.  *n_events
 * (nullable) receives the events staged by this call, also when it fails.  A message the handle refuses (typically
 * ESVO_ERR_CAPACITY "event ring full: render before staging more") is NOT consumed: render and call again. */
int esvo_ts_push_bag(esvo_handle h, int cam, esvo_bag_handle b, const char* topic, uint64_t until_ns, size_t* n_events);
/* Replaces TimeSurface::createTimeSurfaceAtTime (TimeSurface.cpp:52-152), BACKWARD mode.
 * Uses every staged event with ts < t_ns.  out_mono8 (W*H) may be NULL: the rectified TS
 * also stays device-resident as the camera's latest frame.
 * Deviation from the reference, by construction: the device keeps ONE time stamp per pixel (the newest event with
 * ts < t_ns at render time) where the reference keeps a queue of the last 20 events per pixel and scans it backwards
 * (EventQueueMat, TimeSurface.h:28-96).  The two agree whenever renders are requested in non-decreasing time order --
 * what the sync topic of the ROS graph delivers.  (a) A render at a t_ns EARLIER than events of a previous render is
 * refused with ESVO_ERR_STATE (the queue would still find the older event, the single stamp cannot); esvo_reset and
 * re-stage to replay.  (b) The reference's corner case "more than 20 events newer than t_ns are already queued at a
 * pixel, so it reads as empty" (TimeSurface.h:52-75) cannot occur here: events with ts >= t_ns are never scattered
 * before the render at t_ns. */
int esvo_ts_render(esvo_handle h, int cam, uint64_t t_ns, uint8_t* out_mono8);
/* The same in FORWARD mode (time_surface_mode: 1, TimeSurface.cpp:85-116): every raw pixel's decayed value is splatted
 * bilinearly at the pixel's rectified position (the camera's rect_lut, which must have been given to esvo_create) with a
 * clamp to 1 after every add, in raster order of the source pixels -- reproduced by a gather over per-pixel contribution
 * lists sorted by source index (built on first use).  x255, round to u8, 3x3 median; the image is rectified by
 * construction (no remap).  Same staging / monotonic-time rules as esvo_ts_render; the result is the camera's resident
 * frame as well.  No shipped configuration selects this mode. */
int esvo_ts_render_forward(esvo_handle h, int cam, uint64_t t_ns, uint8_t* out_mono8);

/* ---- Mapper: stage-wise seams ------------------------------------------------------ */

/* Replaces the TS_obs_ selection of dataTransferring (esvo_Mapping.cpp:501-534) +
 * DepthFrame setup (esvo_Mapping.cpp:268-272).  ts_left/ts_right: host mono8 W*H, or NULL
 * to use the device-resident frames rendered last by esvo_ts_render. */
int esvo_map_set_observation(esvo_handle h, uint64_t t_ns, const uint8_t* ts_left,
                             const uint8_t* ts_right, const double T_world_cam[16]);
/* Replaces EventBM::createMatchProblem + match_all_HyperThread (EventBM.cpp:56-78,269-315).
 * pose_t_ns/pose_T: the st_map_ of esvo_Mapping.cpp:585-599 (m stamps, m row-major 4x4). */
int esvo_map_match(esvo_handle h, const esvo_event_t* ev, size_t n, const uint64_t* pose_t_ns,
                   const double* pose_T, size_t m, esvo_match_t* out, size_t cap, size_t* n_out);
/* Replaces DepthProblemSolver::solve (+ pointCulling when cull != 0)
 * (DepthProblemSolver.cpp:28-136,216-244).  Uses the pose table of the last esvo_map_match
 * or esvo_map_set_poses call. */
int esvo_map_set_poses(esvo_handle h, const uint64_t* pose_t_ns, const double* pose_T, size_t m);
int esvo_map_refine(esvo_handle h, const esvo_match_t* matches, size_t n, int cull,
                    esvo_depth_point_t* out, size_t cap, size_t* n_out);
/* Replaces dqvDepthPoints_.push_back(vdp) + window policy (esvo_Mapping.cpp:341-368). */
int esvo_map_push_frame(esvo_handle h, const esvo_depth_point_t* pts, size_t n,
                        const double* pose_T, size_t m);
/* Replaces the fusion loop + clean + regularisation (esvo_Mapping.cpp:370-395;
 * DepthFusion.cpp:71-192; SmartGrid.h:222-243; DepthRegularization.cpp:19-110). */
int esvo_map_fuse(esvo_handle h, size_t* n_fusions);

/* ---- Mapper: SGM bootstrap ---------------------------------------------- */

/* Replaces esvo_Mapping::InitializationAtTime (esvo_Mapping.cpp:433-492) and the SGM branch of dataTransferring (:537-552):
 * cv::StereoSGBM (0, 48, 11, P1 = 8*11*11, P2 = 32*11*11, -1, 0, 11; MODE_SGBM; :102-108) on the UN-smoothed Time-Surface
 * pair -- ts_left / ts_right: host mono8 W*H, or NULL for the device-resident frames of esvo_ts_render -- masked by the
 * rectified pixels of the newest <= PROCESS_EVENT_NUM + 1 staged left events of the last 2 * BM_half_slice_thickness
 * (createEdgeMask, :1000-1044), one Gaussian DepthPoint (variance 1e-6, age = age_vis_threshold) per masked event whose
 * disparity lies inside the inverse-depth range.  With at least min_points (INIT_SGM_DP_NUM_THRESHOLD, 500) of them the
 * points open the fusion window and DepthFusion::naive_propagation (DepthFusion.cpp:234-288) fills the DepthFrame of the
 * observation set last (esvo_map_set_observation gives its stamp and pose); otherwise *n_points is 0 and nothing
 * changes.  disp_out (nullable): the W*H int16 disparity image (x16, -16 = none).  StereoSGBM is third-party code that
 * is restated here from its published algorithm ("parity unpinned"). */
int esvo_map_init_sgm(esvo_handle h, const uint8_t* ts_left, const uint8_t* ts_right, size_t min_points, size_t* n_points,
                      int16_t* disp_out);

/* ---- Mapper: fused tick (everything stays in HBM) ---------------------------------- */

/* Replaces dataTransferring's event selection (esvo_Mapping.cpp:555-575) + MappingAtTime
 * (esvo_Mapping.cpp:261-431) on the staged left events and the current observation. */
int esvo_map_tick(esvo_handle h, uint64_t t_ns, const uint64_t* pose_t_ns, const double* pose_T,
                  size_t m);
/* Replaces esvo_MVStereo::MappingAtTime in MVStereoMode 1, PURE_BLOCK_MATCHING (esvo_MVStereo.cpp:383-432): the same event
 * selection (+ Denoising) and block matching, then vEMP2vDP (:1072-1094: one Gaussian DepthPoint per match, variance at the
 * 1e-6 bound, residual = ZNCC cost, age = age_vis_threshold), a window of max_fusion_frames frames and
 * DepthFusion::naive_propagation (DepthFusion.cpp:234-288) of every frame, newest first, into a new DepthFrame.  The map is
 * read with the usual output calls; nothing is culled, cleaned or regularised.  Synchronous. */
int esvo_map_tick_bm_only(esvo_handle h, uint64_t t_ns, const uint64_t* pose_t_ns, const double* pose_T, size_t m);
/* The stage-wise seam of that mode: what follows match_all_HyperThread (esvo_MVStereo.cpp:411-423) on the matches of
 * esvo_map_match -- vEMP2vDP, dqvDepthPoints_.push_back(vdp_em) + pop to maxNumFusionFrames_, naive_propagation of the window.
 * pose_T: the m virtual views the matches' pose_idx refer to (the st_map_ handed to esvo_map_match). */
int esvo_map_fuse_matches_naive(esvo_handle h, const esvo_match_t* matches, size_t n, const double* pose_T, size_t m);
/* One call for a node that hosts the Time Surfaces and the mapper on the same handle: esvo_ts_render of both cameras at
 * t_ns (device-resident, no download), esvo_map_set_observation on them with the pose T_world_cam, esvo_map_tick.  Same
 * results as the four calls; a reference-faithful tick is short enough for their host overhead to show. */
int esvo_map_tick_resident(esvo_handle h, uint64_t t_ns, const double T_world_cam[16], const uint64_t* pose_t_ns,
                           const double* pose_T, size_t m);

/* Device-resident variants of the stage-wise calls (no host copies of points): the front stage of a tick on the
 * events staged by esvo_ts_push_events (selection + match + refine + cull; the frame stays on the device,
 * esvo_map_front_frame), a frame pushed from device memory, and the fusion stage enqueued without waiting for it.
 * esvo_map_tick == front + push_frame_device(front_frame) + fuse_async, minus the point-count read-back stall.
 * They exist for tick-interleaved multi-GPU operation (esvo_amd/dist.py: TickShardedEsvo): a tick depends on the
 * previous ticks only through the frames in its window (the DepthFrame is rebuilt every tick,
 * esvo_Mapping.cpp:266-272), so rank r maps the ticks k = r (mod N) and the ranks all-gather their frames. */
int esvo_map_front(esvo_handle h, uint64_t t_ns, const uint64_t* pose_t_ns, const double* pose_T, size_t m,
                   size_t* n_points);
int esvo_map_front_frame(esvo_handle h, const esvo_depth_point_t** d_frame);
int esvo_map_push_frame_device(esvo_handle h, const esvo_depth_point_t* d_pts, size_t n, const double* pose_T,
                               size_t m);
int esvo_map_fuse_async(esvo_handle h);

/* ---- Outputs ------------------------------------------------------------------------ */

/* DepthMap iteration (SmartGrid.h:346-358) as consumed by the publishers
 * (esvo_Mapping.cpp:925-932).  Elements are returned in the reference's list order. */
int esvo_map_get_depth_points(esvo_handle h, esvo_depth_point_t* out, size_t cap, size_t* n);
/* The DepthMap of the newest COMMITTED tick (*t_ns: its stamp, 0 if none) without completing a pending one:
 * esvo_map_tick(k) commits tick k-1 and leaves tick k's front stage running, so a node that publishes after every
 * tick reads map k-1 here while tick k computes (one tick of latency for the overlap). */
int esvo_map_get_committed(esvo_handle h, esvo_depth_point_t* out, size_t cap, size_t* n, uint64_t* t_ns);
/* Replaces the loop of publishPointCloud (esvo_Mapping.cpp:925-932): p_world = R p_cam + t
 * as float32 xyz triples, the payload of /esvo_mapping/pointcloud_local. */
int esvo_map_get_pointcloud_xyz(esvo_handle h, float* out_xyz, size_t cap_points, size_t* n);
/* ---- Debug images and global-cloud helpers ----
 * Replaces Visualization::plot_map / DrawPoint (Visualization.cpp:13-94) as esvo_Mapping::publishMappingResults calls
 * them for the topics Inverse_Depth_Map, Standard_Variance_Map, Age_Map and cost_map (esvo_Mapping.cpp:868-884): BGR8
 * images of H*W*3 bytes, each pointer may be NULL.  age_max_range is the yaml key of that name. */
int esvo_map_get_debug_images(esvo_handle h, double age_max_range, uint8_t* inv_depth_bgr, uint8_t* stdvar_bgr,
                              uint8_t* age_bgr, uint8_t* cost_bgr);
/* pc_near_ of publishPointCloud (esvo_Mapping.cpp:925-932): world points of the elements with |p_cam| < visualize_range. */
int esvo_map_get_pointcloud_near_xyz(esvo_handle h, double visualize_range, float* out_xyz, size_t cap_points, size_t* n);
/* pcl::VoxelGrid<pcl::PointXYZ> with setLeafSize(leaf, leaf, leaf) (esvo_Mapping.cpp:960-964): host code, no handle.
 * One centroid per occupied voxel, ascending voxel index; the node appends the last NumGPC_added_per_refresh - 1 of them
 * to its global cloud (:966-969). */
int esvo_voxel_filter_xyz(const float* xyz, size_t n, float leaf, float* out_xyz, size_t cap_points, size_t* n_out);
/* Replaces esvo_MVStereo::saveDepthMap (esvo_MVStereo.cpp:982-1000; the call sites at :302, :373, :521 are compiled out upstream
 * with `if (false)`: "to save the depth result, set it to true"): writes <save_dir><t_ns>.txt, one line "x y depth" per valid
 * element of the DepthMap in list order, formatted as Eigen's operator<< and the ofstream format it.  save_dir is used as a
 * prefix exactly as upstream does (it must end with the path separator).  n_written (nullable): lines written. */
int esvo_map_save_depth_map(esvo_handle h, const char* save_dir, uint64_t t_ns, size_t* n_written);
/* The newest frame of the fusion window (culled DepthPoints of the last tick). */
int esvo_map_get_last_frame(esvo_handle h, esvo_depth_point_t* out, size_t cap, size_t* n);
int esvo_get_stats(esvo_handle h, esvo_stats_t* out);
/* sizeof() of {event, calib, params, match, depth_point, stats}, 0, ABI version:
 * lets a foreign-language binding check its struct mirrors. */
void esvo_abi_sizes(size_t out[8]);

/* ---- Multi-GPU: one tick split over the GPUs by image row band ------------------------------------ */

/* Make this handle shard `shard` of `n_shards`: its per-cell work (fuse / clean / regularise) is the image rows
 * [row_begin, row_end).  (0, H, 0, 1) = unsharded.  How the per-event work (match + refine), the event ingest and the
 * Time-Surface raster are divided is esvo_shard_set_routing's choice; this call selects ESVO_ROUTE_BROADCAST. */
int esvo_shard_set_band(esvo_handle h, int row_begin, int row_end, int shard, int n_shards);
/* How a band handle divides the work that is not per cell.  Call after esvo_shard_set_band, before any event is staged
 * (or after esvo_reset): ESVO_ERR_STATE otherwise.
 *   ESVO_ROUTE_Y_RECT (what SURVEY.md §8(e) specifies; the default of esvo_amd/dist.py and bench.py):
 *     - ingest: esvo_ts_push_events* keeps, of the events it is handed, those this rank needs -- for both cameras the raw
 *       rows that the rectifying remap and the median of its rendered rows read (derived from the calibration maps), for
 *       the left camera also every event whose rectified row floor(y_rect) lies in the band.  Only they cross PCIe, sit in
 *       the device ring and are scattered; the host still walks every stamp (the reference's event selection,
 *       esvo_Mapping.cpp:562-574, counts the whole left stream).  The caller hands EVERY rank the full packets, exactly as a
 *       ROS topic would deliver them to eight subscribers.
 *     - raster: the Time Surfaces / the observation pair are rendered for the band + ts_halo_rows rows only (rounded
 *       outwards to whole 16-row tiles; + 2 rows under SmoothTimeSurface).  Rows outside are undefined in every image
 *       the handle returns.
 *     - match + refine: an event belongs to the rank that owns floor(y_rect) (the epipolar search is horizontal,
 *       EventBM.cpp:170-226, so block matching reads the band + patch_size_Y / 2 rows).  The refinement warps the patch by
 *       the camera motion since the event (DepthProblem.cpp:157-191) and may leave those rows: every evaluation is
 *       checked, a match whose blocks leave the rendered rows is COUNTED (stats.halo_violations, summed over all ranks by
 *       the second exchange), and once that count is non-zero every rank refuses its next tick with ESVO_ERR_HALO --
 *       never a silently different result.  Raise ts_halo_rows (DSEC / DAVIS hand-held sequences move < 10 rows in the
 *       10 ms a tick looks back) or fall back to ESVO_ROUTE_BROADCAST.
 *     - Denoising (ABI 7; esvo_Mapping.cpp:1046-1072): a rank decides the mask's verdict for the selected events whose RAW row lies
 *       in its band (its ring keeps one more raw row on either side), the verdicts travel as one bit per selected event
 *       (esvo_shard_tick_phase(h, 0) returns ESVO_AGAIN with that block due), and phase 0's second part matches the kept sequence
 *       of the unsharded tick.
 *     - a push that fails (ESVO_ERR_CAPACITY: event ring full) has consumed nothing -- not the events, not their stamps in the
 *       global sequence: render and hand the SAME packet in again, or the rank's global indices part from its peers'.
 *     Not available with per-pixel event queues, FORWARD mode, up-down stereo: ESVO_ERR_UNSUPPORTED.
 *   ESVO_ROUTE_BROADCAST (A/B switch; exact whatever the motion): every rank stages ALL events and renders the full Time
 *     Surfaces; the per-event work is the slots w with w % n_shards == shard of the tick's thread-stride order.
 * ts_halo_rows < 0: the default (24). */
enum { ESVO_ROUTE_BROADCAST = 0, ESVO_ROUTE_Y_RECT = 1 };
int esvo_shard_set_routing(esvo_handle h, int mode, int ts_halo_rows);
/* What the routing resolved to (each pointer may be NULL): rendered rectified rows, valid rows of the observation pair, the
 * raw rows whose events are kept per camera.  (0, H) everywhere for ESVO_ROUTE_BROADCAST. */
int esvo_shard_get_rows(esvo_handle h, int render_rows[2], int observation_rows[2], int source_rows_left[2], int source_rows_right[2]);
/* Three-phase tick for sharded operation.  The fusion window is replicated (every rank holds every frame; the DepthFrame
 * is rebuilt from it at every tick, so a band needs nothing of its neighbours' maps).  After phase 0 and after phase 1
 * the caller ALL-GATHERS one fixed-size block per rank (esvo_shard_exchange; ncclAllGather / torch.distributed
 * all_gather_into_tensor issued on the handle's stream, see esvo_amd/dist.py):
 *   phase 0: poses + event selection + block matching + LM refinement + culling of the rank's events
 *            -> exchange 1: (bit 0 matched, bit 1 point kept) of the tick's slots --
 *               Y_RECT:    two bits per slot of the WHOLE tick, own slots set: ceil(n / 16) 32-bit words rounded up to 8 bytes
 *               BROADCAST: one byte per OWN slot, slots shard, shard + n_shards ... back to back: ceil(n / n_shards)
 *                          bytes rounded up to 8
 *   phase 1: the tick's frame order (EventBM.cpp:289-308 and DepthProblemSolver.cpp:75-90 permutations, derived
 *            from all ranks' bits), own kept points packed with their final index
 *            -> exchange 2: [count (8 B) | points], 8 + K x sizeof(esvo_depth_point_t) bytes with K = the largest
 *               kept count among the ranks (every rank derives it from exchange 1)
 *   phase 2: every block's points to their place in the frame; window policy (identical on every rank), fusion +
 *            clean + regularisation of the band; the halo rows the band's neighbourhoods read
 *            (2 + RegularizationRadius) are recomputed locally, which is exact because the DepthFrame is rebuilt
 *            from the window at every tick.
 * The DepthMap stays sharded; esvo_map_get_depth_points returns the band's elements (seq = global
 * creation order, so bands merge by sorting on it).  stats.last_solved counts the shard's own problems. */
int esvo_shard_tick_phase(esvo_handle h, int phase, uint64_t t_ns, const uint64_t* pose_t_ns,
                          const double* pose_T, size_t m);
/* The exchange due before the next phase: all-gather block_bytes (a multiple of 8; 0 = nothing to do) from d_send of
 * every rank into d_recv, rank-major.  With n_shards == 1 d_recv aliases d_send and the call may be skipped. */
int esvo_shard_exchange(esvo_handle h, void** d_send, void** d_recv, size_t* block_bytes);

/* ---- Multi-GPU exchange behind the C-ABI: one process per GPU, RCCL over xGMI ------------------

 * The reference is a single-process CPU program (std::thread fan-out only); these calls exist because this
 * implementation can spread its per-tick work over the GPUs of a node.  RCCL is loaded (dlopen) by esvo_comm_unique_id /
 * esvo_comm_init only; the frame exchange of the tick-interleaved mode runs on a stream of its own, every other collective on
 * the handle's front stream.  All calls below are COLLECTIVE: every
 * rank makes them in the same order with the same arguments. */
#define ESVO_COMM_ID_BYTES 128
/* ncclGetUniqueId on one rank; hand the bytes to the others by any means (ROS parameter server, MPI, a file). */
int esvo_comm_unique_id(uint8_t id[ESVO_COMM_ID_BYTES]);
/* Which RCCL the calls above resolved: ncclGetVersion's code and the path of the shared object (dladdr).  The library is
 * loaded with dlopen on first use: ESVO_RCCL_PATH if set, else the librccl.so.1 already mapped into the process (a
 * process that imported torch carries torch's copy), else the loader path, else /opt/rocm/lib. */
int esvo_comm_rccl_info(int* version, char* path, size_t path_cap);
/* ncclCommInitRank on the handle's device. */
int esvo_comm_init(esvo_handle h, const uint8_t id[ESVO_COMM_ID_BYTES], int rank, int world);
/* The same with the collective supplied by the caller (another transport; tests that run several ranks on one GPU):
 * all_gather: bytes_per_rank from d_send into d_recv[rank * bytes_per_rank] on every rank.  It must be ordered after the work
 * already enqueued on hip_stream and must have completed (or be enqueued on hip_stream) when it returns 0. */
typedef int (*esvo_all_gather_fn)(void* user, const void* d_send, void* d_recv, size_t bytes_per_rank, void* hip_stream);
int esvo_comm_init_callbacks(esvo_handle h, int rank, int world, esvo_all_gather_fn all_gather, void* user);
int esvo_comm_destroy(esvo_handle h);
/* Tick-interleaved operation (throughput scales with the GPUs; the latency of one tick does not change): rank r maps
 * the ticks k = r (mod world) completely.  A tick depends on earlier ticks only through the frames of its fusion window
 * (MappingAtTime builds a new DepthFrame at every tick, esvo_Mapping.cpp:266-272,341-377), so after every `world` ticks the
 * ranks exchange that round's frames with ONE ncclAllGather (each block carries its point count: one host wait per
 * round, for an exchange enqueued a round earlier), push them into their windows in tick order and fuse at their own tick.
 * esvo_comm_tick replaces esvo_map_set_observation + esvo_map_tick: every rank calls it for every tick; the rank for which
 * esvo_comm_owns_next_tick() is 1 must have rendered both Time Surfaces of that tick (esvo_ts_render) beforehand. */
int esvo_comm_owns_next_tick(esvo_handle h);
int esvo_comm_tick(esvo_handle h, uint64_t t_ns, const double T_world_cam[16], const uint64_t* pose_t_ns, const double* pose_T,
                   size_t m);
/* The same with the owner's Time Surfaces rendered inside the call (= esvo_map_tick_resident's front: both cameras in one
 * launch per kernel); no esvo_ts_render beforehand. */
int esvo_comm_tick_resident(esvo_handle h, uint64_t t_ns, const double T_world_cam[16], const uint64_t* pose_t_ns,
                            const double* pose_T, size_t m);
/* Two rounds are in flight per rank (ABI 7): the call that completes round j enqueues that round's exchange on a stream of
 * its own -- pack, ncclAllGather, the blocks' counts to the host -- and then pushes + fuses round j - 1, whose exchange went
 * out a whole round earlier; the front stage of round j + 1 follows on the front and LM streams while exchange j travels.
 * The block length of a gather is the largest frame of the last four rounds + 25 % (the buffers' capacity until counts have
 * been seen; a frame that does not fit is gathered again with grown blocks, identically on every rank).  On a tick another
 * rank maps the call scatters that tick's events into this rank's SAE on the front stream (idle on such a tick), so an own
 * render only has its own tick's events left.  esvo_comm_flush completes a partial round and everything in flight. */
int esvo_comm_flush(esvo_handle h);
typedef struct esvo_comm_stats_t {
  uint64_t rounds;             /* rounds collected (frames pushed) */
  uint64_t gathers;            /* frame all-gathers issued (rounds + repeats after a regrow) */
  uint64_t regrows;            /* rounds that were gathered again because a frame exceeded its block */
  uint64_t bytes_sent;         /* bytes this rank contributed to those all-gathers (block length each) */
  uint64_t points_gathered;    /* depth points of all ranks' frames in the collected rounds (104 B each) */
  uint64_t host_wait_us;       /* host time spent waiting for the counts of a round (the one host wait per round): the slack the
                                  host has when the device sets the pace; near zero = the calling thread is the pace */
  uint32_t last_stride_points; /* block length of the last collected round, in points */
  uint32_t stride_cap_points;  /* what the exchange buffers hold per block */
} esvo_comm_stats_t;
int esvo_comm_get_stats(esvo_handle h, esvo_comm_stats_t* out);
/* The DepthMap of the newest tick on every rank (it lives on the rank that mapped it: all-gather of the newest maps). */
int esvo_comm_newest_map(esvo_handle h, esvo_depth_point_t* out, size_t cap, size_t* n, long long* tick_index);
/* One tick split over the ranks (esvo_shard_set_band + esvo_shard_set_routing first): the three phases of
 * esvo_shard_tick_phase with their two ncclAllGather exchanges, and the all-gather of the DepthMap bands, merged on the
 * creation order. */
int esvo_comm_shard_tick(esvo_handle h, uint64_t t_ns, const uint64_t* pose_t_ns, const double* pose_T, size_t m);
int esvo_comm_gather_map(esvo_handle h, esvo_depth_point_t* out, size_t cap, size_t* n);
/* ABI 7: the band exchange is device-resident -- the band's alive cells compacted on the device, their counts gathered first
 * (16 B per rank, the one host wait), then the elements with the largest band as block length, merged on the device by the
 * elements' global creation ids (a scatter by id, one scan, a gather).
 * esvo_comm_gather_pointcloud_xyz: publishPointCloud's cloud (esvo_Mapping.cpp:909-934) of the whole map on every rank -- the
 * points, order and float bits of esvo_map_get_pointcloud_xyz on one GPU: the tracker's reference cloud in the closed loop.
 * esvo_comm_gather_ts: routed band mode renders a rank's rows only, the tracker reads the whole left Time Surface: all-gather of
 * the bands' rows of camera cam's resident surface (standard partition: rank r owns rows [r ceil(H / world), ...)); afterwards
 * esvo_track_set_current(h, NULL, k) and esvo_map_init_sgm(h, NULL, NULL, ...) see the full image on every rank.  A no-op
 * without row routing. */
int esvo_comm_gather_pointcloud_xyz(esvo_handle h, float* out_xyz, size_t cap_points, size_t* n);
int esvo_comm_gather_ts(esvo_handle h, int cam);

/* ---- Tracker residual / Jacobian evaluation ---------------------- */

/* The adjacent consumer of the hot path: esvo_Tracking's RegProblemLM evaluates, for <= 2000 map points, a
 * bilinear fetch on the negated blurred left Time Surface (and on its Sobel derivatives for the analytical
 * Jacobian).  The device already holds that Time Surface, so these calls replace the per-point loops of
 * RegProblemLM.cpp and spare the tracker the TS download; the 6-DoF LM driver (Eigen, Cayley + SVD) stays on the
 * host, as north_star has it.  Supported: patch 1x1, kernelSize 0 or 5, l2 / Huber -- what every shipped
 * cfg/tracking yaml uses.  All calls are synchronous and run on a stream of their own. */
#define ESVO_TRACK_L2 0
#define ESVO_TRACK_HUBER 1
/* TimeSurfaceObservation::getTimeSurfaceNegative(kernelSize) + computeTsNegativeGrad
 * (TimeSurfaceObservation.h:118-147).  ts_left == NULL: the device-resident left Time Surface of the last
 * esvo_ts_render(h, 0, ...). */
int esvo_track_set_current(esvo_handle h, const uint8_t* ts_left, int kernel_size);
/* TS_negative_left_ (mono8, what the reprojection-map publisher draws on) and its derivatives as int16. */
int esvo_track_get_images(esvo_handle h, uint8_t* neg, int16_t* du, int16_t* dv);
/* The point loop of RegProblemLM::setProblem (RegProblemLM.cpp:44-56): p = R_world_ref^T (p_world - t_world_ref).
 * xyz_world: n x 3 float32 in the caller's order, i.e. after the stochastic swaps of :48-49 (rand() stays with
 * the caller). */
int esvo_track_set_reference(esvo_handle h, const float* xyz_world, size_t n, const double T_world_ref[16]);
/* RegProblemLM::operator() (:91-136) on the batch [offset, offset+count) of setStochasticSampling (:71-88):
 * fvec[i] = sqrt(w_i) r_i, r_i = TS_negative(x_i) or 255 when the point does not reproject. */
int esvo_track_residuals(esvo_handle h, const double T_left_ref[16], size_t offset, size_t count, int ls_norm,
                         double huber_threshold, double* fvec, size_t* n_out);
/* RegProblemLM::df at x = 0 (:178-269) for the problem's current R_, t_: fjac is n_out x 6, column-major. */
int esvo_track_jacobian(esvo_handle h, const double R[9], const double t[3], size_t offset, size_t count,
                        double* fjac, size_t* n_out);
/* What one iteration of RegProblemSolverLM::solve_analytical consumes (RegProblemSolverLM.cpp:148-215: minimizeInit -> F(0),
 * minimizeOneStep -> df at 0, then a 6 x 6 system), in ONE launch and 43 doubles back: f = operator()(0) on the batch with its
 * Huber weights (RegProblemLM.cpp:121-131), J = df(0) (:178-269; the reference's Jacobian carries no weight), H = J^T J (6 x 6,
 * row-major, symmetric), b = J^T f, *cost = |f|^2.  R, t: the problem's R_, t_ as for esvo_track_jacobian (the warp of
 * operator()(0) is [R^T | -R^T t]).  The sums are taken in a fixed order (kernels_track.hip), reproduced by the CPU oracle. */
int esvo_track_normal_equations(esvo_handle h, const double R[9], const double t[3], size_t offset, size_t count, int ls_norm,
                                double huber_threshold, double H[36], double b[6], double* cost, size_t* n_out);
/* The same for n_poses (1..ESVO_TRACK_MAX_POSES) poses in ONE launch and one read-back: what Eigen's minimizeOneStep spends one
 * functor evaluation each on -- the trial points of its inner loop (a step per damping value until one is accepted) -- evaluated
 * together.  R: n_poses x 9, t: n_poses x 3, H: n_poses x 36, b: n_poses x 6, cost: n_poses; every pose's sums are the ones
 * esvo_track_normal_equations gives for it alone, bit for bit. */
#define ESVO_TRACK_MAX_POSES 4
int esvo_track_normal_equations_batch(esvo_handle h, int n_poses, const double* R, const double* t, size_t offset, size_t count,
                                      int ls_norm, double huber_threshold, double* H, double* b, double* cost, size_t* n_out);
/* The registration loop on top of it, host C++ inside the library (esvo_hip::gauss_newton_register of esvo_hip.hpp -- the same
 * code a C++ node calls directly): Levenberg-damped Gauss-Newton on (Cayley, translation) with the reference's motion update
 * (RegProblemLM::addMotionUpdate, :347-360: R <- orth(dR R), t <- dt + dR t), over the first n_points of the reference cloud;
 * a step is accepted when its actual cost reduction is at least 1e-4 of the predicted one (Eigen's ratio test), the damping
 * is raised tenfold otherwise and lowered tenfold after an accepted step; three trial dampings are evaluated per launch.
 * R, t: in = the start (R_, t_ of setProblem: identity / zero, or the previous frame's), out = the registered motion.
 * The reference's own driver is Eigen's LevenbergMarquardt (third-party, absent here); this one is what the closed-loop test
 * and bench.py use in its place. */
int esvo_track_register(esvo_handle h, size_t n_points, double R[9], double t[3], int ls_norm, double huber_threshold,
                        int max_iterations, double damping, double* rms, int* iterations);

#ifdef __cplusplus
}
#endif
#endif /* ESVO_HIP_H */
