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
#include <dlfcn.h>
#include <rccl/rccl.h>

#include "context.hpp"

namespace esvo {
// header: point count; the frame follows at byte 16 (DevPoint is 8-byte aligned, 104 B)
__global__ void __launch_bounds__(256) comm_pack_kernel(const u32* __restrict__ n_points, u32 have, const u64* __restrict__ src,
                                                        u64* __restrict__ block, u32 first, u32 stride_pts) {
  const u32 n = have ? *n_points : 0u;
  const u32 words = 13;  // sizeof(DevPoint) / 8
  const u32 lo = first < n ? first : n;
  const u32 cnt = (n - lo) < stride_pts ? (n - lo) : stride_pts;
  const u64 total = (u64)cnt * words;
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (u64)gridDim.x * blockDim.x)
    block[2 + i] = src[(u64)lo * words + i];
  if (blockIdx.x == 0 && threadIdx.x == 0) { block[0] = n; block[1] = first; }
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
struct esvo_comm {
  int rank = 0, world = 1;
  ncclComm_t nccl = nullptr;
  esvo_all_gather_fn cb_gather = nullptr;
  void* cb_user = nullptr;
  // tick-interleaved mode
  u64 k = 0;                       // index of the next tick
  std::vector<RoundTick> round;    // ticks of the round being assembled
  bool have_own = false;           // this rank mapped one of them ...
  int own_fp = 0;                  // ... whose front-stage parity this is
  double own_T[16];
  long long last_own = -1;         // index of the last tick this rank fused
  u32 stride_pts = 65536;          // points per block of the all-gather (grows on demand, identically on every rank)
  u64* d_send = nullptr;
  u64* d_recv[2] = {nullptr, nullptr};
  u64* d_heads = nullptr;          // [world] point counts of the gathered blocks
  u64* d_map_heads = nullptr;      // esvo_comm_newest_map: [2] send + [2 * world] gathered (size, tick index + 1) -- a buffer of
                                   // its own: the back stream may still copy frames out of d_recv[] while the maps are exchanged
  u64* h_heads = nullptr;          // pinned
  hipEvent_t pushed[2];            // the back stream has copied every frame out of d_recv[i]
  bool pushed_ok = false;
  u64 rounds = 0;
  // band mode: all-gather of the band maps
  u64* d_band_send = nullptr;
  u64* d_band_recv = nullptr;
  size_t band_block_words = 0;
  size_t block_words() const { return 2 + (size_t)stride_pts * 13; }
};

namespace esvo_host {
void comm_release(esvo_context* h) {
  esvo_comm* c = h->comm;
  if (!c) return;
  for (void* p : {(void*)c->d_send, (void*)c->d_recv[0], (void*)c->d_recv[1], (void*)c->d_heads, (void*)c->d_map_heads, (void*)c->d_band_send, (void*)c->d_band_recv})
    if (p) hipFree(p);
  if (c->h_heads) hipHostFree(c->h_heads);
  if (c->pushed_ok) { hipEventDestroy(c->pushed[0]); hipEventDestroy(c->pushed[1]); }
  if (c->nccl && g_rccl.CommDestroy) g_rccl.CommDestroy(c->nccl);
  delete c;
  h->comm = nullptr;
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

int comm_all_gather(esvo_context* h, const void* d_send, void* d_recv, size_t bytes) {
  esvo_comm* c = h->comm;
  if (c->cb_gather) {
    if (c->cb_gather(c->cb_user, d_send, d_recv, bytes, h->stream)) FAIL(ESVO_ERR_HIP, "all-gather callback failed");
    return ESVO_OK;
  }
  NCCLCHK(g_rccl.AllGather(d_send, d_recv, bytes, ncclUint8, c->nccl, h->stream));
  return ESVO_OK;
}

int comm_alloc(esvo_context* h) {
  esvo_comm* c = h->comm;
  for (void* p : {(void*)c->d_send, (void*)c->d_recv[0], (void*)c->d_recv[1]})
    if (p) hipFree(p);
  c->d_send = c->d_recv[0] = c->d_recv[1] = nullptr;
  const size_t bw = c->block_words();
  HIPCHK(hipMalloc(reinterpret_cast<void**>(&c->d_send), bw * 8));
  HIPCHK(hipMalloc(reinterpret_cast<void**>(&c->d_recv[0]), bw * 8 * c->world));
  HIPCHK(hipMalloc(reinterpret_cast<void**>(&c->d_recv[1]), bw * 8 * c->world));
  return ESVO_OK;
}

int comm_setup(esvo_context* h, int rank, int world) {
  esvo_comm* c = h->comm;
  c->rank = rank;
  c->world = world;
  if (h->prm.max_events_per_tick > 0 && (u32)h->prm.max_events_per_tick < c->stride_pts)
    c->stride_pts = (u32)h->prm.max_events_per_tick;  // a tick cannot produce more points than events
  int rc = comm_alloc(h);
  if (rc) return rc;
  HIPCHK(hipMalloc(reinterpret_cast<void**>(&c->d_heads), sizeof(u64) * world));
  HIPCHK(hipMalloc(reinterpret_cast<void**>(&c->d_map_heads), sizeof(u64) * (2 + 2 * (size_t)world)));
  HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&c->h_heads), sizeof(u64) * world));
  HIPCHK(hipEventCreateWithFlags(&c->pushed[0], hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&c->pushed[1], hipEventDisableTiming));
  c->pushed_ok = true;
  // one untimed round trip sets up the communicator's channels (and proves the transport works)
  HIPCHK(hipMemsetAsync(c->d_send, 0, 16, h->stream));
  rc = comm_all_gather(h, c->d_send, c->d_recv[0], 16);
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(h->stream));
  return ESVO_OK;
}

// pack block [first, first + stride) of this rank's frame of the round (or an empty block) and gather the blocks
int round_gather(esvo_context* h, u32 first, int buf) {
  esvo_comm* c = h->comm;
  const u32 blocks = std::min<u32>(1024u, (c->stride_pts * 13u + 255u) / 256u);
  hipLaunchKernelGGL(esvo::comm_pack_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, h->stream, h->d_counters + 1,
                     c->have_own ? 1u : 0u, reinterpret_cast<const u64*>(h->d_pts_tmp), c->d_send, first, c->stride_pts);
  const size_t bw = c->block_words();
  int rc = comm_all_gather(h, c->d_send, c->d_recv[buf], bw * 8);
  if (rc) return rc;
  hipLaunchKernelGGL(esvo::comm_headers_kernel, dim3(1), dim3(64), 0, h->stream, c->d_recv[buf], bw, c->world, c->d_heads);
  HIPCHK(hipMemcpyAsync(c->h_heads, c->d_heads, sizeof(u64) * c->world, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));  // the one host wait of the round: the window policy needs the counts
  return ESVO_OK;
}

// all-gather the frames of the round, push them in tick order, fuse at the own tick
int finish_round(esvo_context* h) {
  esvo_comm* c = h->comm;
  if (c->round.empty()) return ESVO_OK;
  if (c->world > 64) FAIL(ESVO_ERR_CAPACITY, "more than 64 ranks");
  const int buf = (int)(c->rounds & 1);
  // d_recv[buf] was last read by the back stream's frame copies two rounds ago
  HIPCHK(hipStreamWaitEvent(h->stream, c->pushed[buf], 0));
  int rc = round_gather(h, 0, buf);
  if (rc) return rc;
  u64 max_n = 0;
  for (int r = 0; r < c->world; ++r) max_n = std::max(max_n, c->h_heads[r]);
  if (max_n > c->stride_pts) {
    // a frame did not fit its block: every rank sees the same counts, grows its blocks alike and gathers again
    // (the owners' frames are still in place: nothing was enqueued behind the wait above)
    c->stride_pts = (u32)std::min<u64>((u64)h->max_ev, max_n + max_n / 4);
    HIPCHK(hipStreamSynchronize(h->stream_l)); HIPCHK(hipStreamSynchronize(h->stream_l1));
  HIPCHK(hipStreamSynchronize(h->stream_b));
    rc = comm_alloc(h);
    if (rc) return rc;
    rc = round_gather(h, 0, buf);
    if (rc) return rc;
  }
  if (c->have_own) {  // front-stage statistics of the own tick (its counters arrived before the gather finished)
    rc = tick_phase1_collect(h, c->own_fp);
    if (rc) return rc;
  }
  const size_t bw = c->block_words();
  const u64 k0 = c->k - c->round.size();
  for (size_t j = 0; j < c->round.size(); ++j) {
    const int owner = (int)((k0 + j) % (u64)c->world);  // a round may start anywhere (partial rounds are flushed)
    const RoundTick& tk = c->round[j];
    const u32 n = (u32)c->h_heads[owner];
    u32 off;
    rc = window_reserve(h, n, &off);
    if (rc) return rc;
    rc = back_after_front(h);
    if (rc) return rc;
    if (n)
      HIPCHK(hipMemcpyAsync(h->d_win + off, c->d_recv[buf] + (size_t)owner * bw + 2, sizeof(DevPoint) * n,
                            hipMemcpyDeviceToDevice, h->stream_b));
    static const double ident[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    rc = commit_frame(h, off, n, tk.m ? tk.poses.data() : ident, tk.m);
    if (rc) return rc;
    if (owner == c->rank) {  // MappingAtTime's fusion for the own tick (esvo_Mapping.cpp:370-395)
      const int par = h->par;
      h->par ^= 1;
      HIPCHK(hipEventSynchronize(h->evt[EV_RG1 + par * EV_BACK_STRIDE]));
      collect_back(h, par);
      rc = run_fuse(h, par, c->own_T);
      if (rc) return rc;
      h->committed_t_ns = tk.t_ns;
      h->stats.ticks++;
      h->stats.last_window_frames = (u32)h->n_window_frames;
      u32 np = 0;
      for (auto& f : h->frames) np += f.count;
      h->stats.last_window_points = np;
      h->stats_pending = true;
      c->last_own = (long long)(k0 + j);
    }
  }
  HIPCHK(hipEventRecord(c->pushed[buf], h->stream_b));
  c->round.clear();
  c->have_own = false;
  c->rounds++;
  return ESVO_OK;
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
  esvo_comm* c = h->comm;
  int rc = flush_pending_tick(h);
  if (rc) return rc;
  if (esvo_comm_owns_next_tick(h)) {
    rc = esvo_map_set_observation(h, t_ns, nullptr, nullptr, T_world_cam);  // the Time Surfaces this rank rendered last
    if (rc) return rc;
    rc = tick_phase0(h, t_ns, pose_t_ns, pose_T, m);
    if (rc) return rc;
    const u32 n = h->tk[h->fpar].n;
    if (n) { rc = run_order_points(h, n, h->d_pts_tmp); if (rc) return rc; }
    HIPCHK(hipMemcpyAsync(h->h_counters + CNT_ROW * h->fpar, h->d_counters, sizeof(u32) * CNT_ROW, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipEventRecord(h->evt[EV_CNT + h->fpar * EV_FRONT_STRIDE], h->stream));
    c->have_own = true;
    c->own_fp = h->fpar;
    std::memcpy(c->own_T, T_world_cam, sizeof(double) * 16);
  }
  RoundTick rt;
  rt.t_ns = t_ns;
  rt.m = (u32)m;
  rt.poses.assign(pose_T, pose_T + 16 * m);
  c->round.push_back(std::move(rt));
  c->k++;
  if ((int)c->round.size() == c->world) return finish_round(h);
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
    if (rc) return rc;
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

// all-gather of the DepthMap bands (north_star: "all-gather of per-tile depth estimates"), merged on the global creation
// order so that the result is the unsharded map's element list
int esvo_comm_gather_map(esvo_handle h, esvo_depth_point_t* out, size_t cap, size_t* n) {
  if (!h || !n) return ESVO_ERR_INVALID_ARG;
  API_LOCK(h);
  if (!h->comm) FAIL(ESVO_ERR_STATE, "esvo_comm_init has not been called");
  HIPCHK(hipSetDevice(h->device));
  esvo_comm* c = h->comm;
  std::vector<esvo_depth_point_t> mine;
  int rc = export_map(h, mine, nullptr);
  if (rc) return rc;
  // a band holds at most ceil(H / world) + 1 rows of cells: fixed block length, count in-band
  const size_t rows = ((size_t)h->H + c->world - 1) / c->world + 1;
  const size_t bw = 2 + rows * h->W * 13;
  if (mine.size() > rows * h->W) FAIL(ESVO_ERR_CAPACITY, "band larger than its block");
  if (c->band_block_words != bw) {
    if (c->d_band_send) hipFree(c->d_band_send);
    if (c->d_band_recv) hipFree(c->d_band_recv);
    c->d_band_send = c->d_band_recv = nullptr;
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&c->d_band_send), bw * 8));
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&c->d_band_recv), bw * 8 * c->world));
    c->band_block_words = bw;
  }
  u64 head[2] = {(u64)mine.size(), 0};
  HIPCHK(hipMemcpyAsync(c->d_band_send, head, 16, hipMemcpyHostToDevice, h->stream));
  if (!mine.empty())
    HIPCHK(hipMemcpyAsync(c->d_band_send + 2, mine.data(), mine.size() * sizeof(esvo_depth_point_t), hipMemcpyHostToDevice, h->stream));
  rc = comm_all_gather(h, c->d_band_send, c->d_band_recv, bw * 8);
  if (rc) return rc;
  std::vector<u64> all(bw * c->world);
  HIPCHK(hipMemcpyAsync(all.data(), c->d_band_recv, bw * 8 * c->world, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  std::vector<esvo_depth_point_t> merged;
  for (int r = 0; r < c->world; ++r) {
    const u64 cnt = all[(size_t)r * bw];
    const esvo_depth_point_t* p = reinterpret_cast<const esvo_depth_point_t*>(&all[(size_t)r * bw + 2]);
    merged.insert(merged.end(), p, p + cnt);
  }
  std::stable_sort(merged.begin(), merged.end(), [](const esvo_depth_point_t& a, const esvo_depth_point_t& b) { return a.seq < b.seq; });
  for (size_t i = 0; i < merged.size(); ++i) merged[i].seq = (u32)i;
  *n = merged.size();
  if (!out) return ESVO_OK;
  if (cap < merged.size()) FAIL(ESVO_ERR_CAPACITY, "output capacity too small");
  std::memcpy(out, merged.data(), merged.size() * sizeof(esvo_depth_point_t));
  return ESVO_OK;
}

}  // extern "C"
