// scan.hip — device-wide exclusive prefix sum of u32 (stable compaction / counting sort support).
//
// Two launches (reduce -> down-sweep with the block sums' scan folded in), 256-thread blocks, 8 items per
// thread, wave64 shuffles + one LDS hop per block.
#include <cstddef>
#include "common.hpp"
#include "scan.hpp"

namespace esvo {

// what a scan reads: 32-bit flags, or bit 0 of one byte per element (the "matched" bit of the band mode's slot codes)
struct ScanInU32 { const u32* p; __device__ u32 operator()(size_t i) const { return p[i]; } };
struct ScanInCodeBit0 { const uint8_t* p; __device__ u32 operator()(size_t i) const { return p[i] & 1u; } };

template <class In>
__global__ void __launch_bounds__(SCAN_B) scan_reduce_kernel(In in, u32* __restrict__ block_sums, size_t n) {
  __shared__ u32 lds[4];
  const size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * SCAN_V;
  u32 s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_V; ++k)
    if (base + k < n) s += in(base + k);
  u32 tot;
  block_excl_scan(s, &tot, lds);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

// Down-sweep with the scan of the block sums folded in (round 6: two dependent launches instead of three, 5-6 us of queue
// latency each on chains that are nothing but small launches -- the band mode's frame order, the compactions of a tick): every
// block adds up the sums of the blocks before it (at most 2048 words, L2-resident) instead of reading them from a third kernel.
// zero (nullable): n words cleared on the way (the next kernel's scatter target), saving that memset's launch as well.
template <class In>
__global__ void __launch_bounds__(SCAN_B) scan_down_kernel(In in, u32* __restrict__ out, const u32* __restrict__ block_sums, size_t n,
                                                           u32* __restrict__ total, u32* __restrict__ zero) {
  __shared__ u32 lds[4];
  u32 before = 0;
  for (u32 i = threadIdx.x; i < blockIdx.x; i += SCAN_B) before += block_sums[i];
  u32 carry;
  block_excl_scan(before, &carry, lds);
  const size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * SCAN_V;
  u32 v[SCAN_V];
  u32 s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_V; ++k) {
    v[k] = (base + k < n) ? in(base + k) : 0u;
    s += v[k];
  }
  u32 tot;
  u32 ex = block_excl_scan(s, &tot, lds) + carry;
#pragma unroll
  for (int k = 0; k < SCAN_V; ++k) {
    if (base + k < n) {
      out[base + k] = ex;
      if (zero) zero[base + k] = 0u;
    }
    ex += v[k];
  }
  if (total && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *total = carry + tot;
}

// Small inputs (a reference-faithful tick scans a few thousand flags): one workgroup walks the array tile by tile
// with a running carry -- one launch instead of three dependent ones (each costs 4-6 us of dispatch latency).
static constexpr int SCAN_SB = 1024;
static constexpr size_t SCAN_SMALL_MAX = 32768;
template <class In>
__global__ void __launch_bounds__(SCAN_SB) scan_small_kernel(In in, u32* __restrict__ out, u32* __restrict__ total, size_t n,
                                                             u32* __restrict__ zero) {
  __shared__ u32 lds[SCAN_SB / ESVO_WAVE];
  u32 carry = 0;
  for (size_t tile = 0; tile < n; tile += (size_t)SCAN_SB * SCAN_V) {
    const size_t base = tile + (size_t)threadIdx.x * SCAN_V;
    u32 v[SCAN_V];
    u32 s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_V; ++k) {
      v[k] = (base + k < n) ? in(base + k) : 0u;
      s += v[k];
    }
    u32 tot;
    u32 ex = block_excl_scan<SCAN_SB / ESVO_WAVE>(s, &tot, lds) + carry;
#pragma unroll
    for (int k = 0; k < SCAN_V; ++k) {
      if (base + k < n) {
        out[base + k] = ex;
        if (zero) zero[base + k] = 0u;
      }
      ex += v[k];
    }
    carry += tot;
  }
  if (threadIdx.x == 0 && total) *total = carry;
}

// The same walk with the COMPACTION folded in (round 3): a small tick's "flags -> exclusive scan -> stable compaction of the
// records" pairs (matches after block matching, depth points after the refinement) were two dependent launches each, 4 us of
// kernel + 6 us of queue latency per launch on the critical path of a reference-faithful tick.  The workgroup that scans
// the flags writes the records it keeps straight away.  SEQ: the record's `seq` field takes its position (DevPoint);
// slot_of (nullable): the source slot of every kept record.  The prefix array is still written: other kernels read it.
// Round 4: the copy is cooperative.  A thread that copied the records of its own kept flags ran up to SCAN_V dependent
// load -> store round trips of 48 / 104 bytes one after the other (23 us for 10 000 flags, the slowest thread deciding); now the
// scan leaves the kept flags' source slots in LDS and ALL threads copy the tile's records 64-bit word by word -- coalesced,
// independent loads (the records are 8-byte aligned multiples of 8 bytes).
template <class Rec, bool SEQ>
__global__ void __launch_bounds__(SCAN_SB) scan_compact_small_kernel(const u32* __restrict__ flags, u32* __restrict__ prefix,
                                                                     u32* __restrict__ total, size_t n, const Rec* __restrict__ slots,
                                                                     Rec* __restrict__ out, u32* __restrict__ slot_of,
                                                                     const u32* __restrict__ row_src, u32* __restrict__ row_host, u32 row_n) {
  static_assert(sizeof(Rec) % 8 == 0 && alignof(Rec) == 8, "records are copied as 64-bit words");
  constexpr u32 WPR = sizeof(Rec) / 8;
  __shared__ u32 lds[SCAN_SB / ESVO_WAVE];
  __shared__ u32 src_slot[SCAN_SB * SCAN_V];  // source slot of the k-th kept flag of the tile
  const unsigned long long* __restrict__ in64 = reinterpret_cast<const unsigned long long*>(slots);
  unsigned long long* __restrict__ out64 = reinterpret_cast<unsigned long long*>(out);
  u32 carry = 0;
  for (size_t tile = 0; tile < n; tile += (size_t)SCAN_SB * SCAN_V) {
    const size_t base = tile + (size_t)threadIdx.x * SCAN_V;
    u32 v[SCAN_V];
    u32 s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_V; ++k) {
      v[k] = (base + k < n) ? flags[base + k] : 0u;
      s += v[k];
    }
    u32 tot;
    u32 ex = block_excl_scan<SCAN_SB / ESVO_WAVE>(s, &tot, lds);
#pragma unroll
    for (int k = 0; k < SCAN_V; ++k) {
      if (base + k < n) {
        prefix[base + k] = ex + carry;
        if (v[k]) src_slot[ex] = (u32)(base + k);
      }
      ex += v[k];
    }
    __syncthreads();
    // (out == nullptr: prefix, total and the counter row only -- the consumer gathers the records itself, back_prologue_kernel)
    for (u32 w = threadIdx.x; out && w < tot * WPR; w += SCAN_SB) {
      const u32 r = w / WPR, q = w - r * WPR;
      const u32 src = src_slot[r];
      unsigned long long word = in64[(size_t)src * WPR + q];
      if constexpr (SEQ) {  // the record's `seq` field takes its position
        constexpr size_t so = offsetof(Rec, seq);
        if (q == so / 8) {
          const unsigned long long m = 0xffffffffull << ((so % 8) * 8);
          word = (word & ~m) | ((unsigned long long)(carry + r) << ((so % 8) * 8));
        }
      }
      out64[(size_t)(carry + r) * WPR + q] = word;
      if (slot_of && q == 0) slot_of[carry + r] = src;
    }
    if (!out && slot_of)  // the list as indices only (LmArgs::match_index)
      for (u32 r = threadIdx.x; r < tot; r += SCAN_SB) slot_of[carry + r] = src_slot[r];
    __syncthreads();  // src_slot is reused by the next tile
    carry += tot;
  }
  if (threadIdx.x == 0 && total) *total = carry;
  // latency mode (api_map.hip): the tick's counter row -- complete with this kernel's total -- goes to the pinned host row from
  // here instead of through a copy operation behind the launch (row_src: the device row `total` belongs to)
  if (row_host && threadIdx.x < row_n) {
    row_host[threadIdx.x] = (row_src + threadIdx.x == total) ? carry : row_src[threadIdx.x];
    __threadfence_system();
  }
}
// One workgroup copying the records pays up to ~10 000 flags (the reference's PROCESS_EVENT_NUM); at 20 000 the parallel
// compaction launch wins again (346x260 throughput tick: 57.4 against 54.3 M events/s).
static constexpr size_t SCAN_COMPACT_SMALL_MAX = 10240;
bool scan_compact_is_small(size_t n) { return n > 0 && n <= SCAN_COMPACT_SMALL_MAX; }
void launch_scan_compact_matches_small(const u32* flags, u32* prefix, u32* d_total, size_t n, const esvo_match_t* slots,
                                       esvo_match_t* out, u32* slot_of, hipStream_t s) {
  hipLaunchKernelGGL((scan_compact_small_kernel<esvo_match_t, false>), dim3(1), dim3(SCAN_SB), 0, s, flags, prefix, d_total, n, slots,
                     out, slot_of, (const u32*)nullptr, (u32*)nullptr, 0u);
}
void launch_scan_compact_points_small(const u32* flags, u32* prefix, u32* d_total, size_t n, const DevPoint* slots, DevPoint* out,
                                      hipStream_t s, const u32* row_src, u32* row_host, u32 row_n) {
  hipLaunchKernelGGL((scan_compact_small_kernel<DevPoint, true>), dim3(1), dim3(SCAN_SB), 0, s, flags, prefix, d_total, n, slots, out,
                     (u32*)nullptr, row_src, row_host, row_n);
}

// Small host -> device uploads of the tick path (pose table, frame table) as a KERNEL that reads the pinned host buffer:
// a kernel launch never blocks the host, whereas hipMemcpyAsync of a few KB was measured to stall its caller for 6-11 ms
// once per process when a third stream of the handle is busy (the copy engine's queue is shared between the streams).
// zero (nullable, n_zero <= 256 words): a few counters cleared by the same launch (the tick's counter row)
__global__ void __launch_bounds__(256) upload_words_kernel(const u32* __restrict__ src, u32* __restrict__ dst, size_t n,
                                                           u32* __restrict__ zero, u32 n_zero) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
  if (blockIdx.x == 0 && threadIdx.x < n_zero) zero[threadIdx.x] = 0u;
}
void launch_upload_words(const void* pinned_src, void* d_dst, size_t bytes, hipStream_t s, u32* d_zero, u32 n_zero) {
  const size_t n = bytes / 4;  // callers pass multiples of 4 bytes
  if (n == 0) {
    if (d_zero && n_zero) hipMemsetAsync(d_zero, 0, sizeof(u32) * n_zero, s);
    return;
  }
  size_t blocks = (n + 255) / 256;
  if (blocks > 256) blocks = 256;
  hipLaunchKernelGGL(upload_words_kernel, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<const u32*>(pinned_src),
                     reinterpret_cast<u32*>(d_dst), n, d_zero, d_zero ? n_zero : 0u);
}

// Latency mode (api_map.hip, tick_phase2): what opens a tick's back stage -- the frame's points from their staging buffer into the
// window ring, the tick's pose table into the frame's slot, the frame table from pinned host memory -- as ONE launch instead of two
// copies and an upload (three dependent operations of 3-4 us each with ~6 us of queue latency between them).
// Gather mode (a_flags != nullptr): copy A is the stable COMPACTION of the refinement's solver slots itself -- a_src = the slot
// records (a_slots of them), kept where a_flags is set, record i to position a_prefix[i] of a_dst with that position in its `seq`
// field -- what scan_compact_small_kernel's copy loop does in one workgroup (20 us for DSEC's 2000 slots), here over the grid.
__global__ void __launch_bounds__(256) back_prologue_kernel(const u32* __restrict__ src, u32* __restrict__ dst, size_t n,
                                                            const unsigned long long* __restrict__ a_src, unsigned long long* __restrict__ a_dst,
                                                            size_t a_n, const unsigned long long* __restrict__ b_src,
                                                            unsigned long long* __restrict__ b_dst, size_t b_n,
                                                            const u32* __restrict__ a_flags, const u32* __restrict__ a_prefix, u32 a_slots) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, T = (size_t)gridDim.x * blockDim.x;
  if (a_flags) {
    constexpr u32 WPR = sizeof(DevPoint) / 8;
    constexpr size_t so = offsetof(DevPoint, seq);
    for (size_t i = t; i < (size_t)a_slots * WPR; i += T) {
      const u32 slot = (u32)(i / WPR), q = (u32)(i - (size_t)slot * WPR);
      if (!a_flags[slot]) continue;
      const u32 pos = a_prefix[slot];
      unsigned long long word = a_src[i];
      if (q == so / 8) {
        const unsigned long long m = 0xffffffffull << ((so % 8) * 8);
        word = (word & ~m) | ((unsigned long long)pos << ((so % 8) * 8));
      }
      a_dst[(size_t)pos * WPR + q] = word;
    }
  } else {
    for (size_t i = t; i < a_n; i += T) a_dst[i] = a_src[i];
  }
  for (size_t i = t; i < b_n; i += T) b_dst[i] = b_src[i];
  for (size_t i = t; i < n; i += T) dst[i] = src[i];
}
void launch_back_prologue(const void* pinned_src, void* d_dst, size_t bytes, const void* a_src, void* a_dst, size_t a_bytes,
                          const void* b_src, void* b_dst, size_t b_bytes, hipStream_t s, const u32* a_flags, const u32* a_prefix,
                          u32 a_slots) {
  static_assert(sizeof(DevPoint) % 8 == 0 && alignof(DevPoint) == 8, "records are copied as 64-bit words");
  const size_t n = bytes / 4, a_n = a_flags ? (size_t)a_slots * (sizeof(DevPoint) / 8) : a_bytes / 8, b_n = b_bytes / 8;
  const size_t most = std::max(n, std::max(a_n, b_n));
  if (most == 0) return;
  size_t blocks = (most + 255) / 256;
  if (blocks > 512) blocks = 512;
  hipLaunchKernelGGL(back_prologue_kernel, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<const u32*>(pinned_src),
                     reinterpret_cast<u32*>(d_dst), n, reinterpret_cast<const unsigned long long*>(a_src),
                     reinterpret_cast<unsigned long long*>(a_dst), a_n, reinterpret_cast<const unsigned long long*>(b_src),
                     reinterpret_cast<unsigned long long*>(b_dst), b_n, a_flags, a_prefix, a_slots);
}

size_t scan_scratch_elems(size_t n) { return (n + SCAN_TILE - 1) / SCAN_TILE + 1; }

template <class In>
static void exclusive_scan(In in, u32* d_out, u32* d_total, u32* d_block_sums, size_t n, u32* d_zero, hipStream_t s) {
  if (n == 0) {
    if (d_total) hipMemsetAsync(d_total, 0, sizeof(u32), s);
    return;
  }
  if (n <= SCAN_SMALL_MAX) {
    hipLaunchKernelGGL(scan_small_kernel<In>, dim3(1), dim3(SCAN_SB), 0, s, in, d_out, d_total, n, d_zero);
    return;
  }
  const u32 nb = (u32)((n + SCAN_TILE - 1) / SCAN_TILE);
  hipLaunchKernelGGL(scan_reduce_kernel<In>, dim3(nb), dim3(SCAN_B), 0, s, in, d_block_sums, n);
  hipLaunchKernelGGL(scan_down_kernel<In>, dim3(nb), dim3(SCAN_B), 0, s, in, d_out, d_block_sums, n, d_total, d_zero);
}
// d_out may alias d_in.  d_total (nullable) receives the sum.  d_block_sums: scan_scratch_elems(n).
void launch_exclusive_scan_u32(const u32* d_in, u32* d_out, u32* d_total, u32* d_block_sums, size_t n,
                               hipStream_t s) {
  exclusive_scan(ScanInU32{d_in}, d_out, d_total, d_block_sums, n, nullptr, s);
}
// The band mode's frame order (api_map.hip, tick_phase1_enqueue) has the tile sums of its first scan produced by the kernel in
// front of it (SCAN_TILE slots per sum), so that scan is its down-sweep alone -- for n above the single-workgroup bound only.
static_assert(SCAN_TILE == (int)SCAN_TILE_SLOTS, "common.hpp's SCAN_TILE_SLOTS is this file's tile");
bool scan_is_small(size_t n) { return n <= SCAN_SMALL_MAX; }
u32 scan_tiles(size_t n) { return (u32)((n + SCAN_TILE - 1) / SCAN_TILE); }
void launch_scan_down_code_bit0(const uint8_t* d_codes, u32* d_out, u32* d_total, const u32* d_tile_sums, size_t n, u32* d_zero, hipStream_t s) {
  hipLaunchKernelGGL(scan_down_kernel<ScanInCodeBit0>, dim3(scan_tiles(n)), dim3(SCAN_B), 0, s, ScanInCodeBit0{d_codes}, d_out, d_tile_sums, n, d_total,
                     d_zero);
}
// the same over bit 0 of one byte per element; d_zero (nullable): n words cleared by the same launches
void launch_exclusive_scan_code_bit0(const uint8_t* d_codes, u32* d_out, u32* d_total, u32* d_block_sums, size_t n, u32* d_zero,
                                     hipStream_t s) {
  exclusive_scan(ScanInCodeBit0{d_codes}, d_out, d_total, d_block_sums, n, d_zero, s);
}

}  // namespace esvo
