// api_comm.hip — multi-GPU exchange behind the C-ABI: one process per GPU, RCCL over xGMI.
//
// The reference has no distributed code (its parallelism is std::thread fan-out inside one process); what is exchanged
// here follows from how this implementation splits ESVO's per-tick work over GPUs:
//   * tick-interleaved (esvo_comm_tick): rank r maps the ticks k = r (mod N) completely; a tick depends on earlier ticks
//     only through the DepthPoint frames of its fusion window (MappingAtTime builds a new DepthFrame every tick,
//     esvo_Mapping.cpp:266-272, :341-377), so the ONE exchange per round of N ticks is an ncclAllGather of the round's
//     frames.  Each block carries its own point count in-band: one collective and one host wait per round.
//   * one tick split over the ranks (esvo_comm_shard_tick): per-event work by image row of the rectified event (or, as
//     an A/B switch, dealt by slot), per-cell work by image row band; the two exchanges of esvo_shard_tick_phase -- the
//     (matched, kept) bits of the tick's slots, then [count | kept points] -- and the DepthMap bands
//     (esvo_comm_gather_map: north_star's "all-gather of per-tile depth estimates") are all ncclAllGather.
// ncclAllGather is the ONLY collective this library issues.  RCCL is loaded with dlopen at esvo_comm_init, so single-GPU
// users of libesvo_hip.so neither link nor initialise it.  esvo_comm_init_callbacks takes the collective as a function
// pointer instead: tests drive several ranks on ONE GPU through exactly this code with an in-process transport, and
// another transport can be plugged in without touching it.
#include <chrono>
#include <dlfcn.h>
#include <rccl/rccl.h>

#include "context.hpp"

namespace esvo {
// a block = [point count | 0 | the frame: DevPoint is 8-byte aligned, 104 B]; the frame is compacted straight into its place
// behind the header (run_order_points), this writes the header once the count is on the device
__global__ void comm_block_header_kernel(const u32* __restrict__ n_points, u64* __restrict__ block) {
  block[0] = *n_points;
  block[1] = 0;
}
// ---- the DepthMap bands merged on the device (esvo_comm_gather_map / _pointcloud_xyz) ------------------------------------------
// Every element carries the id of the record that created it (MapCell::seq: its position in the reference's sequential fusion
// order, < window points x 9), unique over all bands; the unsharded map's list is the elements in ascending id.  So the merge is a
// scatter of (present, where) by id, an exclusive scan over the id range and a gather -- O(elements + ids), no comparison sort.
__global__ void __launch_bounds__(256) band_counts_kernel(const u32* __restrict__ n_mine, u64* __restrict__ head) {
  if (threadIdx.x == 0) { head[0] = *n_mine; head[1] = 0; }
}
__global__ void __launch_bounds__(256) band_mark_kernel(const esvo_depth_point_t* __restrict__ blocks, size_t block_pts, const u64* __restrict__ counts,
                                                        int world, u32 id_cap, u32* __restrict__ present, u32* __restrict__ where) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int r = (int)(e / block_pts);
  if (r >= world) return;
  const size_t i = e - (size_t)r * block_pts;
  if (i >= counts[2 * r]) return;
  const u32 id = blocks[e].seq;
  if (id >= id_cap) return;  // (cannot happen: ids are below window points x 9)
  present[id] = 1u;
  where[id] = (u32)e;
}
__global__ void __launch_bounds__(256) band_merge_kernel(const esvo_depth_point_t* __restrict__ blocks, const u32* __restrict__ present,
                                                         const u32* __restrict__ prefix, const u32* __restrict__ where, u32 id_cap,
                                                         esvo_depth_point_t* __restrict__ out) {
  const u32 id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= id_cap || !present[id]) return;
  esvo_depth_point_t o = blocks[where[id]];
  o.seq = prefix[id];   // list position, as the unsharded export numbers them
  out[prefix[id]] = o;
}
// publishPointCloud's transform (esvo_Mapping.cpp:925-932) of the merged list: the operations of the host loop in
// esvo_map_get_pointcloud_xyz, un-fused, so the float coordinates are the same bits
__global__ void __launch_bounds__(256) band_xyz_kernel(const esvo_depth_point_t* __restrict__ pts, const u32* __restrict__ n, double T0, double T1,
                                                       double T2, double T3, double T4, double T5, double T6, double T7, double T8, double T9,
                                                       double T10, double T11, float* __restrict__ xyz) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= *n) return;
  const double a = pts[i].p_cam[0], b = pts[i].p_cam[1], c = pts[i].p_cam[2];
  auto row = [&](double t0, double t1, double t2, double t3) {
    return (float)__dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(t0, a), __dmul_rn(t1, b)), __dmul_rn(t2, c)), t3);
  };
  xyz[3 * i + 0] = row(T0, T1, T2, T3);
  xyz[3 * i + 1] = row(T4, T5, T6, T7);
  xyz[3 * i + 2] = row(T8, T9, T10, T11);
}
// the rows [y0, y1) of a rank's band out of / into a full image (esvo_comm_gather_ts)
__global__ void __launch_bounds__(256) ts_bands_scatter_kernel(const uint8_t* __restrict__ blocks, size_t block_bytes, int world, int rows, int W, int H,
                                                               int skip_rank, uint8_t* __restrict__ img) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t per = (size_t)rows * W;
  const int r = (int)(t / per);
  if (r >= world || r == skip_rank) return;
  const size_t o = t - (size_t)r * per;
  const size_t y = (size_t)r * rows + o / W;
  if (y >= (size_t)H) return;
  img[y * W + o % W] = blocks[(size_t)r * block_bytes + o];
}
__global__ void comm_headers_kernel(const u64* __restrict__ recv, size_t block_words, int world, u64* __restrict__ out) {
  const int r = threadIdx.x;
  if (r < world) out[r] = recv[(size_t)r * block_words];
}
}  // namespace esvo

struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*GetVersion)(int*) = nullptr;
  std::string path;  // the shared object the symbols resolved into (dladdr)
};
static RcclApi g_rccl;
static const char* load_rccl() {
  if (g_rccl.lib) return nullptr;
  void* lib = nullptr;
  // ESVO_RCCL_PATH pins the library; otherwise the soname already mapped into the process wins (a process that imported
  // torch has torch's bundled RCCL, and two RCCL copies in one process must not be mixed), then the loader path, then ROCm's.
  const char* pinned = getenv("ESVO_RCCL_PATH");
  if (pinned && *pinned) {
    lib = dlopen(pinned, RTLD_NOW | RTLD_GLOBAL);
    if (!lib) return "ESVO_RCCL_PATH could not be loaded (dlopen)";
  }
  if (!lib)
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"})
      if ((lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL))) break;
  if (!lib) return "librccl.so not found (dlopen)";
  RcclApi a;
  a.lib = lib;
  a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
  a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(lib, "ncclCommInitRank"));
  a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
  a.AllGather = reinterpret_cast<decltype(a.AllGather)>(dlsym(lib, "ncclAllGather"));
  a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
  a.GetVersion = reinterpret_cast<decltype(a.GetVersion)>(dlsym(lib, "ncclGetVersion"));
  Dl_info info;
  if (a.GetUniqueId && dladdr(reinterpret_cast<void*>(a.GetUniqueId), &info) && info.dli_fname) a.path = info.dli_fname;
  if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllGather || !a.GetErrorString)
    return "librccl.so lacks an expected symbol";
  g_rccl = a;
  return nullptr;
}

struct RoundTick {
  u64 t_ns;
  std::vector<double> poses;  // m x 16
  u32 m;
};
// One round of `world` ticks.  Up to two rounds are in flight per rank: the exchange of round j is enqueued when the round's last
// tick has been handed in; its counts are looked at -- the one host wait of a round -- only after this rank's front stage of
// round j + 1 has been enqueued (or right before the exchange of round j + 1 goes out, whichever comes first).  So the
// exchange, the host's wake-up and the enqueueing of the next front stage are all off the chain LM -> LM that paces a rank.
struct CommRound {
  std::vector<RoundTick> ticks;
  u64 k0 = 0;              // index of its first tick
  int own_slot = -1;       // this rank mapped one of them: which OwnTick slot holds its frame, counters and events
  double own_T[16];
  int buf = 0;             // which d_recv / h_heads / event set its gather uses
  u32 stride = 0;          // points per block of its gather
};
// What a rank keeps of each of its own ticks until the tick's round has been collected -- four deep, because the front stage
// of the own tick two rounds later (same front parity) is enqueued BEFORE that happens: the block the frame is compacted
// into (and re-gathered from, should the blocks have to grow), the pinned counter row, the front-stage events, the tick state.
struct OwnTick {
  u64* d_block = nullptr;              // [2 + max_ev * 13] header + frame
  u32* h_cnt = nullptr;                // pinned, CNT_ROW
  hipEvent_t ev[EV_FRONT_STRIDE];      // installed into h->evt[EV_T0 + fp * EV_FRONT_STRIDE ...] while the tick's front stage is enqueued
  esvo_context::TickState tk;
  int fp = 0;
  bool live = false;                   // enqueued; EV_CNT of this slot marks "frame, header and counters ready"
};
struct esvo_comm {
  int rank = 0, world = 1;
  ncclComm_t nccl = nullptr;
  esvo_all_gather_fn cb_gather = nullptr;
  void* cb_user = nullptr;
  // tick-interleaved mode
  u64 k = 0;                       // index of the next tick
  CommRound cur;                   // the round being assembled
  std::deque<CommRound> inflight;  // gather enqueued, frames not pushed yet (oldest first; at most two)
  long long last_own = -1;         // index of the last tick this rank fused
  u32 stride_cap = 65536;          // points per block the receive buffers hold (grows on demand, identically on every rank)
  u32 recent_max[4] = {0, 0, 0, 0};  // largest frame of each of the last four rounds whose counts the host has seen
  u32 n_recent = 0;
  hipStream_t sc = nullptr;        // the exchange: all-gather, counts to the host -- beside the front stages of the next rounds
  OwnTick own[4];
  u64 own_seq = 0;                 // own ticks so far
  int last_slot_of_fp[2] = {-1, -1};
  hipEvent_t orig_ev[2][EV_FRONT_STRIDE];  // the handle's own front-stage events, put back by esvo_comm_destroy
  u64* d_empty = nullptr;          // the block of a round without an own tick (a flushed partial round): zeros
  u64* d_recv[2] = {nullptr, nullptr};
  u64* d_heads = nullptr;          // [2][world] point counts of the gathered blocks
  u64* d_map_heads = nullptr;      // esvo_comm_newest_map: [2] send + [2 * world] gathered (size, tick index + 1) -- a buffer of
                                   // its own: the back stream may still copy frames out of d_recv[] while the maps are exchanged
  u64* h_heads = nullptr;          // pinned, [2][world]
  hipEvent_t gathered[2];          // the gather into d_recv[i] and the copy of its counts have completed
  hipEvent_t pushed[2];            // the back stream has copied every frame out of d_recv[i]
  bool events_ok = false;
  u64 rounds = 0;                  // rounds whose gather has been enqueued
  esvo_comm_stats_t st = {};
  // band mode: all-gather of the band maps, merged on the device; of the Time-Surface bands
  esvo_depth_point_t* d_band_recv = nullptr;  // [world][band_block_pts]
  size_t band_block_pts = 0;
  esvo_depth_point_t* d_merged = nullptr;     // [merged_cap] the unsharded element list
  size_t merged_cap = 0;
  u32* d_id_present = nullptr;                // [id_cap] x 3: present, prefix, where
  u32* d_id_scan_tmp = nullptr;
  size_t id_cap = 0;
  u32* d_merged_n = nullptr;
  float* d_xyz = nullptr;
  size_t xyz_cap = 0;
  uint8_t* d_ts_send = nullptr;
  uint8_t* d_ts_recv = nullptr;
  size_t ts_block = 0;
  hipEvent_t ev_map = nullptr;                // the band's export is on the back stream, the exchange on the front stream
  static size_t block_words(u32 stride) { return 2 + (size_t)stride * 13; }
};

namespace esvo_host {
void comm_release(esvo_context* h) {
  esvo_comm* c = h->comm;
  if (!c) return;
  if (c->sc) hipStreamSynchronize(c->sc);
  for (void* p : {(void*)c->d_empty, (void*)c->d_recv[0], (void*)c->d_recv[1], (void*)c->d_heads, (void*)c->d_map_heads, (void*)c->d_band_recv, (void*)c->d_merged, (void*)c->d_id_present, (void*)c->d_id_scan_tmp, (void*)c->d_merged_n, (void*)c->d_xyz, (void*)c->d_ts_send, (void*)c->d_ts_recv})
    if (p) hipFree(p);
  if (c->h_heads) hipHostFree(c->h_heads);
  if (c->ev_map) hipEventDestroy(c->ev_map);
  if (c->events_ok) {
    for (int i = 0; i < 2; ++i) { hipEventDestroy(c->pushed[i]); hipEventDestroy(c->gathered[i]); }
    // the handle gets its own front-stage events back (nothing is in flight: the callers drained every stream)
    for (int fp = 0; fp < 2; ++fp)
      for (int i = 0; i < EV_FRONT_STRIDE; ++i) h->evt[EV_T0 + fp * EV_FRONT_STRIDE + i] = c->orig_ev[fp][i];
    for (OwnTick& o : c->own) {
      for (int i = 0; i < EV_FRONT_STRIDE; ++i) hipEventDestroy(o.ev[i]);
      if (o.d_block) hipFree(o.d_block);
      if (o.h_cnt) hipHostFree(o.h_cnt);
    }
  }
  if (c->nccl && g_rccl.CommDestroy) g_rccl.CommDestroy(c->nccl);
  if (c->sc) hipStreamDestroy(c->sc);
  delete c;
  h->comm = nullptr;
}
// esvo_reset on a handle with a communicator (collective like every esvo_comm_* call): rounds in flight are dropped with
// the window they would have entered
void comm_reset(esvo_context* h) {
  esvo_comm* c = h->comm;
  if (!c) return;
  if (c->sc) hipStreamSynchronize(c->sc);
  c->cur = CommRound();
  c->inflight.clear();
  c->k = 0;
  c->last_own = -1;
  for (OwnTick& o : c->own) o.live = false;
  c->last_slot_of_fp[0] = c->last_slot_of_fp[1] = -1;
}
}  // namespace esvo_host

namespace {
#define NCCLCHK(call)                                                                                         \
  do {                                                                                                        \
    ncclResult_t _r = (call);                                                                                 \
    if (_r != ncclSuccess) {                                                                                  \
      g_create_error = std::string(#call) + " failed: " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(_r) : "?");  \
      return ESVO_ERR_HIP;                                                                                    \
    }                                                                                                         \
  } while (0)

int comm_all_gather(esvo_context* h, const void* d_send, void* d_recv, size_t bytes, hipStream_t st = nullptr) {
  esvo_comm* c = h->comm;
  if (!st) st = h->stream;
  if (c->cb_gather) {
    if (c->cb_gather(c->cb_user, d_send, d_recv, bytes, st)) FAIL(ESVO_ERR_HIP, "all-gather callback failed");
    return ESVO_OK;
  }
  NCCLCHK(g_rccl.AllGather(d_send, d_recv, bytes, ncclUint8, c->nccl, st));
  return ESVO_OK;
}

int comm_alloc(esvo_context* h) {
  esvo_comm* c = h->comm;
  for (void* p : {(void*)c->d_empty, (void*)c->d_recv[0], (void*)c->d_recv[1]})
    if (p) hipFree(p);
  c->d_empty = c->d_recv[0] = c->d_recv[1] = nullptr;
  const size_t bw = esvo_comm::block_words(c->stride_cap);
  HIPCHK(hipMalloc(reinterpret_cast<void**>(&c->d_empty), bw * 8));
  HIPCHK(hipMemsetAsync(c->d_empty, 0, bw * 8, c->sc));  // (on the stream that reads it: the null stream is not ordered with a non-blocking one)
  HIPCHK(hipMalloc(reinterpret_cast<void**>(&c->d_recv[0]), bw * 8 * c->world));
  HIPCHK(hipMalloc(reinterpret_cast<void**>(&c->d_recv[1]), bw * 8 * c->world));
  return ESVO_OK;
}

int comm_setup(esvo_context* h, int rank, int world) {
  esvo_comm* c = h->comm;
  c->rank = rank;
  c->world = world;
  if (world > 64) FAIL(ESVO_ERR_CAPACITY, "more than 64 ranks");
  if (h->prm.max_events_per_tick > 0 && (u32)h->prm.max_events_per_tick < c->stride_cap)
    c->stride_cap = (u32)h->prm.max_events_per_tick;  // a tick cannot produce more points than events
  if (const char* e0 = esvo_dev_switch("ESVO_COMM_STRIDE0")) c->stride_cap = (u32)std::max(1, std::atoi(e0));  // tests: force the regrow path
  {
    // The exchange stream gets the HIGH priority of the front and back streams: short, latency-critical work -- and HIP keeps a
    // pool of hardware queues per priority, so it does not end up in the hardware queue of the ingest or tracker stream, whose
    // packets would then sit behind this stream's wait for the LM launch.
    int prio_lo = 0, prio_hi = 0;
    HIPCHK(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
    HIPCHK(hipStreamCreateWithPriority(&c->sc, hipStreamNonBlocking, prio_hi));
  }
  int rc = comm_alloc(h);
  if (rc) return rc;
  HIPCHK(hipMalloc(reinterpret_cast<void**>(&c->d_heads), sizeof(u64) * 2 * world));
  HIPCHK(hipMalloc(reinterpret_cast<void**>(&c->d_map_heads), sizeof(u64) * (2 + 2 * (size_t)world)));
  HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&c->h_heads), sizeof(u64) * 2 * world));
  for (OwnTick& o : c->own) {  // (a block holds a frame of any size a tick can produce: re-gathered from here when blocks grow)
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&o.d_block), esvo_comm::block_words(h->max_ev) * 8));
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&o.h_cnt), sizeof(u32) * CNT_ROW));
    std::memset(o.h_cnt, 0, sizeof(u32) * CNT_ROW);
    for (int i = 0; i < EV_FRONT_STRIDE; ++i) HIPCHK(hipEventCreate(&o.ev[i]));
  }
  for (int fp = 0; fp < 2; ++fp)
    for (int i = 0; i < EV_FRONT_STRIDE; ++i) c->orig_ev[fp][i] = h->evt[EV_T0 + fp * EV_FRONT_STRIDE + i];
  for (int i = 0; i < 2; ++i) {
    HIPCHK(hipEventCreateWithFlags(&c->pushed[i], hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&c->gathered[i], hipEventDisableTiming));
  }
  c->events_ok = true;
  // one untimed round trip sets up the communicator's channels (and proves the transport works)
  rc = comm_all_gather(h, c->d_empty, c->d_recv[0], 16, c->sc);
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(c->sc));
  return ESVO_OK;
}

// Block length of the next gather: the largest frame of the last four rounds + 25 % (whole 256 points), the buffers' capacity
// before any count has been seen.  Every rank sees the same counts at the same point of the call sequence, so all of them
// choose the same length; a frame that does not fit is noticed when its round is collected (round_collect regrows and gathers again).
u32 pick_stride(const esvo_comm* c) {
  if (c->n_recent == 0) return c->stride_cap;
  u32 m = 0;
  for (u32 i = 0; i < 4u && i < c->n_recent; ++i) m = std::max(m, c->recent_max[i]);
  const u64 want = ((u64)m + m / 4 + 255) / 256 * 256;
  return (u32)std::min<u64>(std::max<u64>(want, 256), c->stride_cap);
}

// enqueue the exchange of a round on the exchange stream: all-gather of the blocks (this rank's own frame sits in its OwnTick
// block, compacted there by the LM stage; a round without an own tick contributes the empty block), counts to the host
int round_gather(esvo_context* h, CommRound& R) {
  esvo_comm* c = h->comm;
  const int buf = R.buf;
  // d_recv[buf] was last read by the back stream's frame copies of the round two before this one
  HIPCHK(hipStreamWaitEvent(c->sc, c->pushed[buf], 0));
  const u64* d_send = c->d_empty;
  if (R.own_slot >= 0) {  // frame, header and count: behind the LM launch on that tick's queues
    HIPCHK(hipStreamWaitEvent(c->sc, c->own[R.own_slot].ev[EV_CNT - EV_T0], 0));
    d_send = c->own[R.own_slot].d_block;
  }
  const size_t bw = esvo_comm::block_words(R.stride);
  int rc = comm_all_gather(h, d_send, c->d_recv[buf], bw * 8, c->sc);
  if (rc) return rc;
  hipLaunchKernelGGL(esvo::comm_headers_kernel, dim3(1), dim3(64), 0, c->sc, c->d_recv[buf], bw, c->world, c->d_heads + (size_t)buf * c->world);
  HIPCHK(hipMemcpyAsync(c->h_heads + (size_t)buf * c->world, c->d_heads + (size_t)buf * c->world, sizeof(u64) * c->world, hipMemcpyDeviceToHost, c->sc));
  HIPCHK(hipEventRecord(c->gathered[buf], c->sc));
  HIPCHK(hipGetLastError());
  c->st.gathers++;
  c->st.bytes_sent += bw * 8;
  return ESVO_OK;
}

// the oldest round in flight: wait for its counts (the one host wait of a round -- for a gather that was enqueued a round or
// more ago), push its frames in tick order, fuse at the own tick
int round_collect(esvo_context* h) {
  esvo_comm* c = h->comm;
  if (c->inflight.empty()) return ESVO_OK;
  CommRound& R = c->inflight.front();
  const u64* heads = c->h_heads + (size_t)R.buf * c->world;
  {
    const auto t0 = std::chrono::steady_clock::now();
    HIPCHK(hipEventSynchronize(c->gathered[R.buf]));
    c->st.host_wait_us += (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
  }
  u64 max_n = 0, sum_n = 0;
  for (int r = 0; r < c->world; ++r) { max_n = std::max(max_n, heads[r]); sum_n += heads[r]; }
  if (max_n > R.stride) {
    // A frame did not fit its block: every rank sees the same counts, grows its blocks alike and gathers again -- this round
    // and the one enqueued behind it with the same block length.  The owners' frames are still in place (an OwnTick block is
    // rewritten four own ticks later), and nothing has been pushed out of these gathers yet.
    HIPCHK(hipStreamSynchronize(c->sc));
    HIPCHK(hipStreamSynchronize(h->stream_b));
    const u32 grown = (u32)std::min<u64>((u64)h->max_ev, max_n + max_n / 4);
    if (grown > c->stride_cap) {
      c->stride_cap = grown;
      int rc = comm_alloc(h);
      if (rc) return rc;
    }
    c->st.regrows++;
    for (CommRound& L : c->inflight) {
      L.stride = std::max(L.stride, grown);
      int rc = round_gather(h, L);
      if (rc) return rc;
    }
    HIPCHK(hipEventSynchronize(c->gathered[R.buf]));
    max_n = 0;
    for (int r = 0; r < c->world; ++r) max_n = std::max(max_n, heads[r]);
    if (max_n > R.stride) FAIL(ESVO_ERR_CAPACITY, "a frame is larger than max_events_per_tick");
  }
  c->recent_max[c->n_recent & 3u] = (u32)max_n;
  c->n_recent++;
  c->st.rounds++;
  c->st.points_gathered += sum_n;
  c->st.last_stride_points = R.stride;
  if (R.own_slot >= 0) {  // front-stage statistics of the own tick (its counters arrived before the gather finished)
    OwnTick& o = c->own[R.own_slot];
    int rc = collect_front_stats(h, o.tk, o.h_cnt, o.ev);
    if (rc) return rc;
    o.live = false;
  }
  const size_t bw = esvo_comm::block_words(R.stride);
  for (size_t j = 0; j < R.ticks.size(); ++j) {
    const int owner = (int)((R.k0 + j) % (u64)c->world);  // a round may start anywhere (partial rounds are flushed)
    const RoundTick& tk = R.ticks[j];
    const u32 n = (u32)heads[owner];
    u32 off;
    int rc = window_reserve(h, n, &off);
    if (rc) return rc;
    // (the frames come out of the gather, not off the front stream: the back stage must not queue behind the front stages of
    //  the next rounds, which are already enqueued there)
    HIPCHK(hipStreamWaitEvent(h->stream_b, c->gathered[R.buf], 0));
    if (n)
      HIPCHK(hipMemcpyAsync(h->d_win + off, c->d_recv[R.buf] + (size_t)owner * bw + 2, sizeof(DevPoint) * n,
                            hipMemcpyDeviceToDevice, h->stream_b));
    static const double ident[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    rc = commit_frame(h, off, n, tk.m ? tk.poses.data() : ident, tk.m);
    if (rc) return rc;
    if (owner == c->rank) {  // MappingAtTime's fusion for the own tick (esvo_Mapping.cpp:370-395)
      const int par = h->par;
      h->par ^= 1;
      HIPCHK(hipEventSynchronize(h->evt[EV_RG1 + par * EV_BACK_STRIDE]));
      collect_back(h, par);
      rc = run_fuse(h, par, R.own_T);
      if (rc) return rc;
      h->committed_t_ns = tk.t_ns;
      h->stats.ticks++;
      h->stats.last_window_frames = (u32)h->n_window_frames;
      u32 np = 0;
      for (auto& f : h->frames) np += f.count;
      h->stats.last_window_points = np;
      h->stats_pending = true;
      c->last_own = (long long)(R.k0 + j);
    }
  }
  HIPCHK(hipEventRecord(c->pushed[R.buf], h->stream_b));
  c->inflight.pop_front();
  return ESVO_OK;
}

// the round being assembled is complete (or flushed): its exchange starts.  At most two rounds are in flight: the receive
// buffer (and the counts) this gather writes belonged to the round two before it, which is collected first if it still waits.
int round_launch(esvo_context* h) {
  esvo_comm* c = h->comm;
  if (c->cur.ticks.empty()) return ESVO_OK;
  while (c->inflight.size() > 1) { int rc = round_collect(h); if (rc) return rc; }
  CommRound R = std::move(c->cur);
  c->cur = CommRound();
  R.k0 = c->k - R.ticks.size();
  R.buf = (int)(c->rounds & 1);
  R.stride = pick_stride(c);
  int rc = round_gather(h, R);
  if (rc) return rc;
  c->rounds++;
  c->inflight.push_back(std::move(R));
  return ESVO_OK;
}

// everything assembled or in flight: exchanged, pushed, fused (esvo_comm_flush, the map read-outs, esvo_comm_destroy)
int finish_round(esvo_context* h) {
  int rc = round_launch(h);
  while (!rc && !h->comm->inflight.empty()) rc = round_collect(h);
  return rc;
}

// the front stage of an own tick: Time Surfaces (render == true), observation, block matching on the front stream; LM on its
// queue; the frame compacted straight into the tick's exchange block, header and counters behind it
int own_front(esvo_context* h, uint64_t t_ns, const double T_world_cam[16], const uint64_t* pose_t_ns, const double* pose_T, size_t m,
              bool render) {
  esvo_comm* c = h->comm;
  const int nfp = h->fpar ^ 1;                 // the parity tick_phase0 switches to
  const int slot = (int)(c->own_seq & 3u);
  OwnTick& o = c->own[slot];
  if (o.live) FAIL(ESVO_ERR_STATE, "four own ticks in flight");  // (cannot happen: a round is collected before the one two later goes out)
  // This front stage rewrites what the own tick two rounds ago (same parity) used -- observation pair, pose table, match list,
  // counter row, LM output -- and that tick's round may not have been collected yet: behind its frame on the DEVICE.
  if (c->last_slot_of_fp[nfp] >= 0) HIPCHK(hipStreamWaitEvent(h->stream, c->own[c->last_slot_of_fp[nfp]].ev[EV_CNT - EV_T0], 0));
  for (int i = 0; i < EV_FRONT_STRIDE; ++i) h->evt[EV_T0 + nfp * EV_FRONT_STRIDE + i] = o.ev[i];
  int rc;
  if (render) {  // = esvo_map_tick_resident's front: both cameras in one launch per kernel, the observation written by the remap
    begin_observation(h);
    uint8_t* obs[2] = {h->d_obs[0], h->d_obs[1]};
    rc = ts_render_pair(h, t_ns, h->prm.smooth_time_surface ? nullptr : obs);
    if (rc) { revert_observation(h); return rc; }
    if (h->prm.smooth_time_surface)
      launch_gaussian5_pair(h->d_ts[0], h->d_ts[1], h->d_obs[0], h->d_obs[1], h->W, h->H, h->stream, 0, -1);
    HIPCHK(hipGetLastError());
    std::memcpy(h->T_world_obs, T_world_cam, sizeof(double) * 16);
    h->obs_t_ns = t_ns;
    h->obs_set = true;
  } else {
    rc = esvo_map_set_observation(h, t_ns, nullptr, nullptr, T_world_cam);  // the Time Surfaces this rank rendered last
    if (rc) return rc;
  }
  // the LM launch on its own queue (as in the lazy single-GPU tick): the front stages of the next rounds, enqueued while this
  // launch runs, overlap it
  h->split_now = h->lm_split && !h->prm.denoising;
  rc = tick_phase0(h, t_ns, pose_t_ns, pose_T, m);
  const bool split = h->split_now;
  h->split_now = false;
  if (rc) return rc;
  const esvo_context::TickState& tk = h->tk[h->fpar];
  // frame compaction, header and counters: on the idle second LM queue when there is one (the next LM launch then follows this
  // one directly on its queue), behind the LM kernel
  hipStream_t sl = tk.n ? tk.lm_stream : h->stream;
  hipStream_t sn = sl;
  if (tk.n && split && h->collect_aside && !h->lm_two_now && (sl == h->stream_l || sl == h->stream_l1)) {
    sn = sl == h->stream_l ? h->stream_l1 : h->stream_l;
    HIPCHK(hipStreamWaitEvent(sn, h->evt[EV_LM1 + h->fpar * EV_FRONT_STRIDE], 0));
  }
  if (tk.n) { rc = run_order_points(h, tk.n, reinterpret_cast<DevPoint*>(o.d_block + 2), sn); if (rc) return rc; }
  hipLaunchKernelGGL(esvo::comm_block_header_kernel, dim3(1), dim3(1), 0, sn, h->d_counters + 1, o.d_block);
  HIPCHK(hipMemcpyAsync(o.h_cnt, h->d_counters, sizeof(u32) * CNT_ROW, hipMemcpyDeviceToHost, sn));
  HIPCHK(hipEventRecord(h->evt[EV_CNT + h->fpar * EV_FRONT_STRIDE], sn));
  HIPCHK(hipGetLastError());
  o.tk = tk;
  o.fp = h->fpar;
  o.live = true;
  c->last_slot_of_fp[h->fpar] = slot;
  c->own_seq++;
  c->cur.own_slot = slot;
  std::memcpy(c->cur.own_T, T_world_cam, sizeof(double) * 16);
  return ESVO_OK;
}

// one tick of the tick-interleaved mode; `render`: the owner renders both Time Surfaces itself (esvo_comm_tick_resident)
int comm_tick(esvo_context* h, uint64_t t_ns, const double T_world_cam[16], const uint64_t* pose_t_ns, const double* pose_T, size_t m,
              bool render) {
  esvo_comm* c = h->comm;
  int rc = flush_pending_tick(h);
  if (rc) return rc;
  if ((int)(c->k % (u64)c->world) == c->rank) {
    rc = own_front(h, t_ns, T_world_cam, pose_t_ns, pose_T, m, render);
    if (rc) return rc;
  } else {
    static const bool ahead = !(esvo_dev_switch("ESVO_COMM_SCATTER_AHEAD") && std::atoi(esvo_dev_switch("ESVO_COMM_SCATTER_AHEAD")) == 0);
    if (ahead) rc = ts_scatter_ahead(h, t_ns);  // the tick's events reach the SAE now (front stream, idle beside the own tick's LM launch)
    if (rc) return rc;
  }
  RoundTick rt;
  rt.t_ns = t_ns;
  rt.m = (u32)m;
  rt.poses.assign(pose_T, pose_T + 16 * m);
  c->cur.ticks.push_back(std::move(rt));
  c->k++;
  // The round before the newest exchange comes in once this rank's front stage of the CURRENT round is enqueued: the host then
  // waits for counts that left a round ago with two front stages queued behind them on the device.
  if (c->cur.own_slot >= 0)
    while (!rc && c->inflight.size() > 1) rc = round_collect(h);
  if (!rc && (int)c->cur.ticks.size() == c->world) rc = round_launch(h);  // this round's exchange goes out
  return rc;
}
}  // namespace

extern "C" {

int esvo_comm_unique_id(uint8_t id[ESVO_COMM_ID_BYTES]) {
  if (!id) return ESVO_ERR_INVALID_ARG;
  static_assert(sizeof(ncclUniqueId) == ESVO_COMM_ID_BYTES, "ncclUniqueId size");
  if (const char* e = load_rccl()) { g_create_error = e; return ESVO_ERR_UNSUPPORTED; }
  ncclUniqueId u;
  if (g_rccl.GetUniqueId(&u) != ncclSuccess) { g_create_error = "ncclGetUniqueId failed"; return ESVO_ERR_HIP; }
  std::memcpy(id, &u, sizeof(u));
  return ESVO_OK;
}

int esvo_comm_rccl_info(int* version, char* path, size_t path_cap) {
  if (const char* e = load_rccl()) { g_create_error = e; return ESVO_ERR_UNSUPPORTED; }
  if (version) {
    *version = 0;
    if (g_rccl.GetVersion) g_rccl.GetVersion(version);
  }
  if (path && path_cap) {
    std::strncpy(path, g_rccl.path.c_str(), path_cap - 1);
    path[path_cap - 1] = 0;
  }
  return ESVO_OK;
}

int esvo_comm_init(esvo_handle h, const uint8_t id[ESVO_COMM_ID_BYTES], int rank, int world) {
  if (!h || !id || world < 1 || rank < 0 || rank >= world) return ESVO_ERR_INVALID_ARG;
  API_LOCK(h);
  if (h->comm) FAIL(ESVO_ERR_STATE, "the handle already has a communicator");
  if (const char* e = load_rccl()) FAIL(ESVO_ERR_UNSUPPORTED, e);
  HIPCHK(hipSetDevice(h->device));
  { int rcp = flush_pending_tick(h); if (rcp) return rcp; }
  h->comm = new esvo_comm();
  ncclUniqueId u;
  std::memcpy(&u, id, sizeof(u));
  ncclResult_t r = g_rccl.CommInitRank(&h->comm->nccl, world, u, rank);
  if (r != ncclSuccess) {
    g_create_error = std::string("ncclCommInitRank failed: ") + g_rccl.GetErrorString(r);
    delete h->comm;
    h->comm = nullptr;
    return ESVO_ERR_HIP;
  }
  int rc = comm_setup(h, rank, world);
  if (rc) comm_release(h);
  return rc;
}

int esvo_comm_init_callbacks(esvo_handle h, int rank, int world, esvo_all_gather_fn all_gather, void* user) {
  if (!h || !all_gather || world < 1 || rank < 0 || rank >= world) return ESVO_ERR_INVALID_ARG;
  API_LOCK(h);
  if (h->comm) FAIL(ESVO_ERR_STATE, "the handle already has a communicator");
  HIPCHK(hipSetDevice(h->device));
  { int rcp = flush_pending_tick(h); if (rcp) return rcp; }
  h->comm = new esvo_comm();
  h->comm->cb_gather = all_gather;
  h->comm->cb_user = user;
  int rc = comm_setup(h, rank, world);
  if (rc) comm_release(h);
  return rc;
}

int esvo_comm_destroy(esvo_handle h) {
  if (!h) return ESVO_ERR_INVALID_ARG;
  API_LOCK(h);
  if (!h->comm) return ESVO_OK;
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipStreamSynchronize(h->stream_l)); HIPCHK(hipStreamSynchronize(h->stream_l1));
  HIPCHK(hipStreamSynchronize(h->comm->sc));
  HIPCHK(hipStreamSynchronize(h->stream_b));
  comm_release(h);
  return ESVO_OK;
}

int esvo_comm_owns_next_tick(esvo_handle h) {
  if (!h || !h->comm) return 0;
  return (int)(h->comm->k % (u64)h->comm->world) == h->comm->rank ? 1 : 0;
}

int esvo_comm_tick(esvo_handle h, uint64_t t_ns, const double T_world_cam[16], const uint64_t* pose_t_ns, const double* pose_T,
                   size_t m) {
  if (!h || !T_world_cam || !pose_t_ns || !pose_T) return ESVO_ERR_INVALID_ARG;
  API_LOCK(h);
  if (!h->comm) FAIL(ESVO_ERR_STATE, "esvo_comm_init has not been called");
  if (h->sharded) FAIL(ESVO_ERR_STATE, "handle is sharded by slot/band: use esvo_comm_shard_tick");
  if (m > h->max_poses) FAIL(ESVO_ERR_CAPACITY, "pose table larger than max_poses_per_tick");
  HIPCHK(hipSetDevice(h->device));
  return comm_tick(h, t_ns, T_world_cam, pose_t_ns, pose_T, m, false);
}

int esvo_comm_tick_resident(esvo_handle h, uint64_t t_ns, const double T_world_cam[16], const uint64_t* pose_t_ns, const double* pose_T,
                            size_t m) {
  if (!h || !T_world_cam || !pose_t_ns || !pose_T) return ESVO_ERR_INVALID_ARG;
  API_LOCK(h);
  if (!h->comm) FAIL(ESVO_ERR_STATE, "esvo_comm_init has not been called");
  if (h->sharded) FAIL(ESVO_ERR_STATE, "handle is sharded by slot/band: use esvo_comm_shard_tick");
  if (m > h->max_poses) FAIL(ESVO_ERR_CAPACITY, "pose table larger than max_poses_per_tick");
  HIPCHK(hipSetDevice(h->device));
  return comm_tick(h, t_ns, T_world_cam, pose_t_ns, pose_T, m, true);
}

int esvo_comm_get_stats(esvo_handle h, esvo_comm_stats_t* out) {
  if (!h || !out) return ESVO_ERR_INVALID_ARG;
  API_LOCK(h);
  if (!h->comm) FAIL(ESVO_ERR_STATE, "esvo_comm_init has not been called");
  *out = h->comm->st;
  out->stride_cap_points = h->comm->stride_cap;
  return ESVO_OK;
}

int esvo_comm_flush(esvo_handle h) {
  if (!h) return ESVO_ERR_INVALID_ARG;
  API_LOCK(h);
  if (!h->comm) return ESVO_OK;
  HIPCHK(hipSetDevice(h->device));
  return finish_round(h);
}

int esvo_comm_newest_map(esvo_handle h, esvo_depth_point_t* out, size_t cap, size_t* n, long long* tick_index) {
  if (!h || !n) return ESVO_ERR_INVALID_ARG;
  API_LOCK(h);
  if (!h->comm) FAIL(ESVO_ERR_STATE, "esvo_comm_init has not been called");
  HIPCHK(hipSetDevice(h->device));
  esvo_comm* c = h->comm;
  int rc = finish_round(h);
  if (rc) return rc;
  // every rank exports its newest own map; the blocks carry (tick index + 1) so that all ranks pick the same one
  std::vector<esvo_depth_point_t> mine;
  if (c->last_own >= 0) { rc = export_map(h, mine, nullptr); if (rc) return rc; }
  const size_t W = 13;
  // sizes first (8 B per rank), then the data with the largest size as block length
  u64 head[2] = {(u64)mine.size(), (u64)(c->last_own + 1)};
  u64* d_hs = c->d_map_heads;
  u64* d_hr = c->d_map_heads + 2;
  HIPCHK(hipMemcpyAsync(d_hs, head, 16, hipMemcpyHostToDevice, h->stream));
  rc = comm_all_gather(h, d_hs, d_hr, 16);
  if (rc) return rc;
  std::vector<u64> heads(2 * (size_t)c->world);
  HIPCHK(hipMemcpyAsync(heads.data(), d_hr, 16 * (size_t)c->world, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  int best = 0;
  u64 max_n = 0;
  for (int r = 0; r < c->world; ++r) {
    if (heads[2 * r + 1] > heads[2 * best + 1]) best = r;
    max_n = std::max(max_n, heads[2 * r]);
  }
  const size_t count = (size_t)heads[2 * best];
  if (tick_index) *tick_index = (long long)heads[2 * best + 1] - 1;
  *n = count;
  if (!count) return ESVO_OK;  // the same on every rank
  // Every rank takes the same path from here on, whatever its `out` / `cap` and whatever its local allocations did: the
  // ranks first agree (one more 16-byte gather) that all of them could stage the exchange, and only then gather the maps --
  // a rank that skipped a collective the others make would hang them.
  u64 *d_s = nullptr, *d_r = nullptr;
  const size_t bw = std::max<size_t>(max_n * W, 1);
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&d_s), bw * 8);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_r), bw * 8 * c->world);
  if (e == hipSuccess && !mine.empty())
    e = hipMemcpyAsync(d_s, mine.data(), mine.size() * sizeof(esvo_depth_point_t), hipMemcpyHostToDevice, h->stream);
  u64 ok[2] = {e == hipSuccess ? 0u : 1u, 0u};
  bool all_ok = false;
  if (hipMemcpyAsync(d_hs, ok, 16, hipMemcpyHostToDevice, h->stream) != hipSuccess) rc = ESVO_ERR_HIP;
  if (!rc) rc = comm_all_gather(h, d_hs, d_hr, 16);
  if (!rc && hipMemcpyAsync(heads.data(), d_hr, 16 * (size_t)c->world, hipMemcpyDeviceToHost, h->stream) != hipSuccess) rc = ESVO_ERR_HIP;
  if (!rc && hipStreamSynchronize(h->stream) != hipSuccess) rc = ESVO_ERR_HIP;
  if (!rc) {
    all_ok = true;
    for (int r = 0; r < c->world; ++r) all_ok = all_ok && heads[2 * r] == 0;
    if (!all_ok) { g_create_error = "staging of the map exchange failed on a rank"; rc = ESVO_ERR_HIP; }
  } else {
    g_create_error = "map exchange failed";
  }
  if (all_ok) rc = comm_all_gather(h, d_s, d_r, bw * 8);
  if (!rc && out) {
    if (cap < count) { g_create_error = "output capacity too small"; rc = ESVO_ERR_CAPACITY; }
    else {
      e = hipMemcpyAsync(out, d_r + (size_t)best * bw, count * sizeof(esvo_depth_point_t), hipMemcpyDeviceToHost, h->stream);
      if (e != hipSuccess) { g_create_error = "copy of the gathered map failed"; rc = ESVO_ERR_HIP; }
    }
  }
  if (hipStreamSynchronize(h->stream) != hipSuccess && !rc) { g_create_error = "map exchange failed"; rc = ESVO_ERR_HIP; }
  if (d_s) hipFree(d_s);
  if (d_r) hipFree(d_r);
  return rc;
}

// ---- one tick split over the ranks: the three phases of esvo_shard_tick_phase with their two all-gathers -------------------
int esvo_comm_shard_tick(esvo_handle h, uint64_t t_ns, const uint64_t* pose_t_ns, const double* pose_T, size_t m) {
  if (!h || !pose_t_ns || !pose_T) return ESVO_ERR_INVALID_ARG;
  API_LOCK(h);
  if (!h->comm) FAIL(ESVO_ERR_STATE, "esvo_comm_init has not been called");
  for (int phase = 0; phase < 3; ++phase) {
    int rc = esvo_shard_tick_phase(h, phase, t_ns, pose_t_ns, pose_T, m);
    if (rc == ESVO_AGAIN) {  // Denoising on a routed handle: the mask bits are exchanged first, phase 0 then runs its second part
      --phase;
    } else if (rc) {
      return rc;
    }
    if (phase < 2 && h->comm->world > 1) {
      void *snd = nullptr, *rcv = nullptr;
      size_t nb = 0;
      esvo_shard_exchange(h, &snd, &rcv, &nb);
      // the block lengths are identical on every rank (slots of the tick / ranks; the largest kept count, which every rank
      // derives from the first exchange)
      if (nb) { rc = comm_all_gather(h, snd, rcv, nb); if (rc) return rc; }
    }
  }
  return ESVO_OK;
}

}  // extern "C"

// all-gather of the DepthMap bands (north_star: "all-gather of per-tile depth estimates"), merged on the global creation
// order so that the result is the unsharded map's element list.  Device-resident since round 6: the band's alive cells are
// compacted on the device, the counts travel first (16 B per rank, one host wait: they size the second exchange), then the
// elements -- block = the largest band -- and the merge runs on the device (band_mark / scan / band_merge above).  The list stays
// in c->d_merged (count in c->d_merged_n); *n_total receives the count.  Collective.
namespace {
int gather_map_device(esvo_context* h, size_t* n_total) {
  esvo_comm* c = h->comm;
  int rc = flush_pending_tick(h);
  if (rc) return rc;
  if (!c->ev_map) HIPCHK(hipEventCreateWithFlags(&c->ev_map, hipEventDisableTiming));
  if (!c->d_merged_n) HIPCHK(hipMalloc(reinterpret_cast<void**>(&c->d_merged_n), sizeof(u32) * 2));
  // the band's alive cells, in cell order (esvo_depth_point_t with the global creation id in seq), on the back stream
  launch_map_compact(h->d_map_cur, h->d_exp_flags, h->d_exp_prefix, h->d_cnt_b + 5, h->d_scan_tmp_b, h->d_export, nullptr, h->dp, h->stream_b);
  u64* d_hs = c->d_map_heads;
  u64* d_hr = c->d_map_heads + 2;
  hipLaunchKernelGGL(esvo::band_counts_kernel, dim3(1), dim3(64), 0, h->stream_b, h->d_cnt_b + 5, d_hs);
  HIPCHK(hipEventRecord(c->ev_map, h->stream_b));
  HIPCHK(hipStreamWaitEvent(h->stream, c->ev_map, 0));
  rc = comm_all_gather(h, d_hs, d_hr, 16);
  if (rc) return rc;
  std::vector<u64> heads(2 * (size_t)c->world);
  HIPCHK(hipMemcpyAsync(heads.data(), d_hr, 16 * (size_t)c->world, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  size_t max_n = 0, total = 0;
  for (int r = 0; r < c->world; ++r) { max_n = std::max<size_t>(max_n, heads[2 * r]); total += heads[2 * r]; }
  *n_total = total;
  h->stats.last_map_size = (u32)heads[2 * (size_t)c->rank];
  if (!total) { HIPCHK(hipMemsetAsync(c->d_merged_n, 0, sizeof(u32), h->stream)); return ESVO_OK; }  // the same on every rank
  // ids are below (points of the window) x 9 (kernels_fuse.hip: record id q K + k)
  size_t win_pts = 0;
  for (auto& f : h->frames) win_pts += f.count;
  const size_t id_need = std::max<size_t>(win_pts * 9 + 1, 1024);
  // (a failed allocation below would leave the other ranks in a collective this one skips; the buffers are small against the
  //  handle's own and grow geometrically, so it is treated like any other out-of-memory condition: the call fails)
  if (max_n > c->band_block_pts) {
    if (c->d_band_recv) hipFree(c->d_band_recv);
    c->d_band_recv = nullptr;
    c->band_block_pts = max_n + max_n / 4 + 256;
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&c->d_band_recv), sizeof(esvo_depth_point_t) * c->band_block_pts * c->world));
  }
  if (total > c->merged_cap) {
    if (c->d_merged) hipFree(c->d_merged);
    c->d_merged = nullptr;
    c->merged_cap = total + total / 4 + 256;
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&c->d_merged), sizeof(esvo_depth_point_t) * c->merged_cap));
  }
  if (id_need > c->id_cap) {
    if (c->d_id_present) hipFree(c->d_id_present);
    if (c->d_id_scan_tmp) hipFree(c->d_id_scan_tmp);
    c->d_id_present = c->d_id_scan_tmp = nullptr;
    c->id_cap = id_need + id_need / 4;
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&c->d_id_present), sizeof(u32) * 3 * c->id_cap));
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&c->d_id_scan_tmp), sizeof(u32) * (scan_scratch_elems(c->id_cap) + 8)));
  }
  // every rank's block has the SAME length -- the largest band of this read-out -- whatever its own buffers hold
  rc = comm_all_gather(h, h->d_export, c->d_band_recv, max_n * sizeof(esvo_depth_point_t));
  if (rc) return rc;
  u32* present = c->d_id_present;
  u32* prefix = present + c->id_cap;
  u32* where = prefix + c->id_cap;
  const u32 idc = (u32)id_need;
  HIPCHK(hipMemsetAsync(present, 0, sizeof(u32) * idc, h->stream));
  const size_t slots = max_n * (size_t)c->world;
  hipLaunchKernelGGL(esvo::band_mark_kernel, dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, h->stream, c->d_band_recv, max_n, d_hr, c->world, idc,
                     present, where);
  launch_exclusive_scan_u32(present, prefix, c->d_merged_n, c->d_id_scan_tmp, idc, h->stream);
  hipLaunchKernelGGL(esvo::band_merge_kernel, dim3((idc + 255) / 256), dim3(256), 0, h->stream, c->d_band_recv, present, prefix, where, idc, c->d_merged);
  HIPCHK(hipGetLastError());
  return ESVO_OK;
}
}  // namespace

extern "C" {
int esvo_comm_gather_map(esvo_handle h, esvo_depth_point_t* out, size_t cap, size_t* n) {
  if (!h || !n) return ESVO_ERR_INVALID_ARG;
  API_LOCK(h);
  if (!h->comm) FAIL(ESVO_ERR_STATE, "esvo_comm_init has not been called");
  HIPCHK(hipSetDevice(h->device));
  size_t total = 0;
  int rc = gather_map_device(h, &total);
  if (rc) return rc;
  *n = total;
  if (!out || !total) { HIPCHK(hipStreamSynchronize(h->stream)); return ESVO_OK; }
  if (cap < total) { HIPCHK(hipStreamSynchronize(h->stream)); FAIL(ESVO_ERR_CAPACITY, "output capacity too small"); }
  HIPCHK(hipMemcpyAsync(out, h->comm->d_merged, sizeof(esvo_depth_point_t) * total, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return ESVO_OK;
}

// publishPointCloud's cloud (esvo_Mapping.cpp:909-934) of the WHOLE map on every rank: what esvo_map_get_pointcloud_xyz returns
// on one GPU, same points, same order, same bits -- the tracker's reference cloud in the closed loop on row bands
int esvo_comm_gather_pointcloud_xyz(esvo_handle h, float* out_xyz, size_t cap_points, size_t* n) {
  if (!h || !n) return ESVO_ERR_INVALID_ARG;
  API_LOCK(h);
  if (!h->comm) FAIL(ESVO_ERR_STATE, "esvo_comm_init has not been called");
  HIPCHK(hipSetDevice(h->device));
  esvo_comm* c = h->comm;
  size_t total = 0;
  int rc = gather_map_device(h, &total);
  if (rc) return rc;
  *n = total;
  if (!out_xyz || !total) { HIPCHK(hipStreamSynchronize(h->stream)); return ESVO_OK; }
  if (cap_points < total) { HIPCHK(hipStreamSynchronize(h->stream)); FAIL(ESVO_ERR_CAPACITY, "output array too small for the point cloud"); }
  if (total > c->xyz_cap) {
    if (c->d_xyz) hipFree(c->d_xyz);
    c->d_xyz = nullptr;
    c->xyz_cap = total + total / 4 + 256;
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&c->d_xyz), sizeof(float) * 3 * c->xyz_cap));
  }
  const double* T = h->T_world_frame;
  hipLaunchKernelGGL(esvo::band_xyz_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->stream, c->d_merged, c->d_merged_n, T[0], T[1], T[2], T[3],
                     T[4], T[5], T[6], T[7], T[8], T[9], T[10], T[11], c->d_xyz);
  HIPCHK(hipMemcpyAsync(out_xyz, c->d_xyz, sizeof(float) * 3 * total, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return ESVO_OK;
}

// Routed band mode renders a rank's rows only; the tracker (esvo_track_set_current) reads the WHOLE left Time Surface.  All-gather
// of the bands' rows of camera `cam`'s resident surface: afterwards every rank holds the full image, bit for bit the unsharded
// render (the banded raster is, tests/test_gpu_shard.py).  Bands must be the standard partition (rank r owns rows
// [r ceil(H / world), (r + 1) ceil(H / world))).  Without row routing the surfaces are whole already: nothing is exchanged.
int esvo_comm_gather_ts(esvo_handle h, int cam) {
  if (!h || cam < 0 || cam > 1) return ESVO_ERR_INVALID_ARG;
  API_LOCK(h);
  if (!h->comm) FAIL(ESVO_ERR_STATE, "esvo_comm_init has not been called");
  if (!h->routed) return ESVO_OK;
  HIPCHK(hipSetDevice(h->device));
  esvo_comm* c = h->comm;
  const int rows = (h->H + c->world - 1) / c->world;
  if (h->dp.band_y0 != std::min(c->rank * rows, h->H) || h->dp.band_y1 != std::min((c->rank + 1) * rows, h->H))
    FAIL(ESVO_ERR_UNSUPPORTED, "esvo_comm_gather_ts: the band is not rank * ceil(H / world) rows");
  if (!h->ts_valid[cam]) FAIL(ESVO_ERR_STATE, "no device-resident Time Surface: call esvo_ts_render first");
  const size_t block = ((size_t)rows * h->W + 7) / 8 * 8;
  if (c->ts_block != block) {
    if (c->d_ts_send) hipFree(c->d_ts_send);
    if (c->d_ts_recv) hipFree(c->d_ts_recv);
    c->d_ts_send = c->d_ts_recv = nullptr;
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&c->d_ts_send), block));
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&c->d_ts_recv), block * c->world));
    HIPCHK(hipMemsetAsync(c->d_ts_send, 0, block, h->stream));  // (the pad bytes; on the stream the copies below run on: hipMemset's null stream is not ordered with it)
    c->ts_block = block;
  }
  std::lock_guard<std::mutex> lt(h->mu_ts);
  resident_write_begin(h, cam);  // a tracker read of the previous surface may be in flight
  const size_t own = (size_t)(h->dp.band_y1 - h->dp.band_y0) * h->W;
  if (own) HIPCHK(hipMemcpyAsync(c->d_ts_send, h->d_ts[cam] + (size_t)h->dp.band_y0 * h->W, own, hipMemcpyDeviceToDevice, h->stream));
  int rc = comm_all_gather(h, c->d_ts_send, c->d_ts_recv, block);
  if (rc) return rc;
  const size_t cells = (size_t)rows * h->W * c->world;
  hipLaunchKernelGGL(esvo::ts_bands_scatter_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, h->stream, c->d_ts_recv, block, c->world, rows, h->W,
                     h->H, c->rank, h->d_ts[cam]);
  HIPCHK(hipGetLastError());
  if (cam == 0) HIPCHK(hipEventRecord(h->evt[EV_R1], h->stream));  // what esvo_track_set_current waits for
  return ESVO_OK;
}

}  // extern "C"
