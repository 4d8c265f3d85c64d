// kernels_shard.hip — multi-GPU tick: putting the ranks' depth points in the reference's order.
//
// Per-event work is dealt round-robin: rank r block-matches and refines the slots w with
// w % N == r (kernels_bm.hip) and keeps its results in a dense local list, in increasing w.  The
// order of the tick's frame (vdp after pointCulling) is a function of two bits per slot only:
//   bit 0  the event was matched             -> position j of the match in vEMP (prefix over w)
//   bit 1  its point survived LM + culling   -> solver slot s = stride_slot(j, M, T)
//                                               (DepthProblemSolver.cpp:75-90), final index = prefix over s
// Both exchanges of a tick are ALL-GATHERS of fixed-size blocks:
//   1. the (matched, kept) byte of every own slot -- block r holds rank r's slots r, r + N, r + 2N ... back to back (ceil(n / N)
//      bytes rounded up to 8); every rank then derives the same order AND every rank's kept count;
//   2. the kept points, each already carrying its final index (seq) -- block r = [count (8 B) | points], the block length sized
//      by the largest kept count among the ranks (known to every rank from exchange 1), so no zero padding travels beyond
//      the imbalance between ranks.  Every rank copies the points of all blocks to frame[seq].
// Nothing here touches a point's payload.
#include "common.hpp"

namespace esvo {

// own block of exchange 1 (zeroed by the caller): byte k = slot r + k N
__global__ void __launch_bounds__(256) shard_codes_kernel(const u32* __restrict__ own_w, const u32* __restrict__ keep,
                                                          const u32* __restrict__ n_local, u32 max_local, u32 N,
                                                          uint8_t* __restrict__ block) {
  const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  u32 n = *n_local;
  if (n > max_local) n = max_local;
  if (k >= n) return;
  block[own_w[k] / N] = (uint8_t)(1u | (keep[k] ? 2u : 0u));
}
void launch_shard_codes(const u32* own_w, const u32* keep, const u32* n_local, u32 max_local, u32 N, uint8_t* block, hipStream_t s) {
  if (max_local == 0) return;
  hipLaunchKernelGGL(shard_codes_kernel, dim3((max_local + 255) / 256), dim3(256), 0, s, own_w, keep, n_local, max_local, N, block);
}

// ---- routed band mode: per-event work belongs to the rank that owns floor(y_rect) of the event, so a rank's own slots are
// scattered over the tick.  Exchange 1 then carries TWO BITS per slot of the WHOLE tick (own slots set, the others zero):
// block = ceil(n / 16) 32-bit words rounded up to 8 bytes, the same on every rank.
// own block (zeroed by the caller): the slot of each own match comes from its walk position (esvo_match_t::event_idx)
__global__ void __launch_bounds__(256) shard_codes_routed_kernel(const esvo_match_t* __restrict__ own_matches, const u32* __restrict__ keep,
                                                                 const u32* __restrict__ n_local, u32 max_local, u32 n, u32 T,
                                                                 u32* __restrict__ own_w, u32* __restrict__ block) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  u32 m = *n_local;
  if (m > max_local) m = max_local;
  if (i >= m) return;
  const u32 w = stride_slot(own_matches[i].event_idx, n, T);
  own_w[i] = w;
  if (w < n) atomicOr(&block[w >> 4], (1u | (keep[i] ? 2u : 0u)) << (2u * (w & 15u)));
}
void launch_shard_codes_routed(const esvo_match_t* own_matches, const u32* keep, const u32* n_local, u32 max_local, u32 n, u32 T, u32* own_w,
                               u32* block, hipStream_t s) {
  if (max_local == 0) return;
  hipLaunchKernelGGL(shard_codes_routed_kernel, dim3((max_local + 255) / 256), dim3(256), 0, s, own_matches, keep, n_local, max_local, n, T,
                     own_w, block);
}
// after exchange 1: the gathered blocks [N][block_words] into one byte per slot, and every rank's kept count.  A slot has at most
// one owner, so the OR of the N ranks' words IS the tick's codes: one thread per word reads its N copies and writes all sixteen
// bytes, zeros included (round 6: no memset of the codes beforehand, one launch over the words instead of N).
// tile_sums (nullable): the workgroup is then 128 threads = 128 words = 2048 slots = one tile of the scan that follows, and leaves
// the tile's number of matched slots there -- that scan is its down-sweep alone.
__global__ void __launch_bounds__(256) shard_unpack_routed_kernel(const u32* __restrict__ blocks, u32 block_words, u32 n_words, u32 N, u32 n,
                                                                  uint8_t* __restrict__ codes, u32* __restrict__ rank_kept,
                                                                  u32* __restrict__ tile_sums) {
  __shared__ u32 part[1024];   // kept points per rank in this workgroup (N <= SHARD_MAX_RANKS)
  __shared__ u32 matched;
  if (threadIdx.x == 0) matched = 0u;
  for (u32 r = threadIdx.x; r < N; r += blockDim.x) part[r] = 0u;
  __syncthreads();
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  u32 all = 0;
  for (u32 r = 0; r < N; ++r) {
    const u32 word = i < n_words ? blocks[(size_t)r * block_words + i] : 0u;
    all |= word;
    u32 cnt = (u32)__popc(word & 0xaaaaaaaau);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) cnt += __shfl_xor(cnt, d);
    if ((threadIdx.x & 63u) == 0 && cnt) atomicAdd(&part[r], cnt);
  }
  if (i < n_words) {
    const u32 w0 = 16u * i;
    if (w0 + 16u <= n) {  // sixteen codes = one 16-byte store (the array is 16-byte aligned at a multiple of 16 slots)
      u32 q4[4];
#pragma unroll
      for (u32 g = 0; g < 4; ++g) {
        const u32 b = (all >> (8u * g)) & 0xffu;   // four 2-bit codes -> four bytes
        q4[g] = (b & 3u) | (((b >> 2) & 3u) << 8) | (((b >> 4) & 3u) << 16) | (((b >> 6) & 3u) << 24);
      }
      *reinterpret_cast<uint4*>(codes + w0) = make_uint4(q4[0], q4[1], q4[2], q4[3]);
    } else {
      for (u32 q = 0; q < 16 && w0 + q < n; ++q) codes[w0 + q] = (uint8_t)((all >> (2u * q)) & 3u);
    }
  }
  if (tile_sums) {
    u32 m = (u32)__popc(all & 0x55555555u);   // (slots at or beyond n carry no bits: the blocks are zero there)
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m += __shfl_xor(m, d);
    if ((threadIdx.x & 63u) == 0 && m) atomicAdd(&matched, m);
  }
  __syncthreads();
  for (u32 r = threadIdx.x; r < N; r += blockDim.x)
    if (part[r]) atomicAdd(&rank_kept[r], part[r]);
  if (tile_sums && threadIdx.x == 0) tile_sums[blockIdx.x] = matched;
}
void launch_shard_unpack_routed(const u32* blocks, u32 block_words, u32 N, u32 n, uint8_t* codes, u32* rank_kept, u32* tile_sums, hipStream_t s) {
  if (n == 0) return;
  const u32 n_words = (n + 15) / 16;
  static_assert(SCAN_TILE_SLOTS == 128 * 16, "a 128-thread workgroup of the unpack covers one scan tile");
  const u32 threads = tile_sums ? 128u : 256u;
  hipLaunchKernelGGL(shard_unpack_routed_kernel, dim3((n_words + threads - 1) / threads), dim3(threads), 0, s, blocks, block_words, n_words, N, n,
                     codes, rank_kept, tile_sums);
}

// after exchange 1: the gathered blocks [N][block_bytes] back into one byte per slot, and every rank's kept count
// (rank_kept[r], zero on entry: one atomic per workgroup)
__global__ void __launch_bounds__(256) shard_unpack_codes_kernel(const uint8_t* __restrict__ blocks, u32 block_bytes, u32 N, u32 n,
                                                                 uint8_t* __restrict__ codes, u32* __restrict__ rank_kept) {
  __shared__ u32 part[4];
  const u32 k = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
  const u64 w = (u64)r + (u64)k * N;
  u32 kept = 0;
  if (w < n) {
    const u32 c = blocks[(size_t)r * block_bytes + k];
    codes[w] = (uint8_t)c;
    kept = (c >> 1) & 1u;
  }
  const u32 cnt = (u32)__popcll(__ballot(kept));
  if ((threadIdx.x & 63u) == 0) part[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) {
    const u32 t = part[0] + part[1] + part[2] + part[3];
    if (t) atomicAdd(&rank_kept[r], t);
  }
}
void launch_shard_unpack_codes(const uint8_t* blocks, u32 block_bytes, u32 N, u32 n, uint8_t* codes, u32* rank_kept, hipStream_t s) {
  if (n == 0) return;
  const u32 per = (n + N - 1) / N;
  hipLaunchKernelGGL(shard_unpack_codes_kernel, dim3((per + 255) / 256, N), dim3(256), 0, s, blocks, block_bytes, N, n, codes,
                     rank_kept);
}

// keep_by_slot must be zero on entry (the scan before this clears it: launch_exclusive_scan_code_bit0).  cursor (nullable): the count
// word of the exchange-2 block, the append cursor of shard_pack_kernel, cleared here instead of by a memset of its own.
__global__ void __launch_bounds__(256) shard_keep_flags_kernel(const uint8_t* __restrict__ codes, const u32* __restrict__ prefix_f,
                                                               const u32* __restrict__ n_matches, u32 n, u32 T,
                                                               u32* __restrict__ keep_by_slot, unsigned long long* __restrict__ cursor) {
  const u32 w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w == 0 && cursor) *cursor = 0ull;
  const u32 c = w < n ? codes[w] : 0u;
  u32 s = 0xffffffffu;
  if (c & 1u) s = stride_slot(prefix_f[w], *n_matches, T);
  const bool hit = s < n;
  if (hit) keep_by_slot[s] = (c >> 1) & 1u;
}
// (Counting the kept slots per scan tile here as well -- atomics into ~100 words -- was tried in round 6 to save the second scan's
//  reduce launch: consecutive matches land in the same few tiles, the atomics of all eight XCDs meet on the same words and the
//  kernel takes 116-152 us instead of 5.)
void launch_shard_keep_flags(const uint8_t* codes, const u32* prefix_f, const u32* n_matches, u32 n, u32 T, u32* keep_by_slot,
                             unsigned long long* cursor, hipStream_t s) {
  if (n == 0) return;
  hipLaunchKernelGGL(shard_keep_flags_kernel, dim3((n + 255) / 256), dim3(256), 0, s, codes, prefix_f, n_matches, n, T,
                     keep_by_slot, cursor);
}

// own block of exchange 2: [count (u64) | kept points in any order], each point with its final index in seq (as
// compact_points_kernel numbers them).  The count word is the append cursor (zero on entry; one atomic per wave).  Thread 0 also
// publishes the block length of the exchange -- the largest kept count among the ranks -- and clears the per-rank counts for
// the next tick (every workgroup of shard_unpack_codes_kernel finished long ago: same stream).
__global__ void __launch_bounds__(256) shard_pack_kernel(const u32* __restrict__ own_w, const u32* __restrict__ keep,
                                                         const DevPoint* __restrict__ local_pts, const u32* __restrict__ n_local,
                                                         u32 max_local, const u32* __restrict__ prefix_f,
                                                         const u32* __restrict__ n_matches, const u32* __restrict__ prefix_g, u32 T,
                                                         unsigned long long* __restrict__ block, u32 block_cap, u32 frame_cap,
                                                         u32* __restrict__ rank_kept, u32 N, u32* __restrict__ max_kept_out,
                                                         const u32* __restrict__ halo_viol) {
  const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k == 0) {
    u32 mx = 0;
    for (u32 r = 0; r < N; ++r) { mx = max(mx, rank_kept[r]); rank_kept[r] = 0u; }
    *max_kept_out = mx;
    // routed band mode: this rank's count of matches whose refinement read outside its Time-Surface rows travels in the
    // high half of the block's count word, so that every rank learns of it at the same tick (shard_scatter_kernel)
    if (halo_viol && *halo_viol) atomicAdd(block, (unsigned long long)*halo_viol << 32);
  }
  u32 n = *n_local;
  if (n > max_local) n = max_local;
  const bool mine = k < n && keep[k];
  u32 idx = 0;
  if (mine) {
    const u32 s = stride_slot(prefix_f[own_w[k]], *n_matches, T);
    idx = prefix_g[s];
  }
  const bool put = mine && idx < frame_cap;
  const u64 m = __ballot(put);
  if (m == 0) return;
  const u32 lane = threadIdx.x & 63u;
  u32 base = 0;
  if (lane == (u32)__ffsll((long long)m) - 1u) base = (u32)atomicAdd(block, (unsigned long long)__popcll(m));  // (low half: the cursor)
  base = __shfl(base, __ffsll((long long)m) - 1);
  if (!put) return;
  const u32 pos = base + (u32)__popcll(m & ((1ull << lane) - 1ull));
  if (pos >= block_cap) return;
  DevPoint o = local_pts[k];
  o.seq = idx;
  reinterpret_cast<DevPoint*>(block + 1)[pos] = o;
}
void launch_shard_pack(const u32* own_w, const u32* keep, const DevPoint* local_pts, const u32* n_local, u32 max_local,
                       const u32* prefix_f, const u32* n_matches, const u32* prefix_g, u32 T, unsigned long long* block, u32 block_cap,
                       u32 frame_cap, u32* rank_kept, u32 N, u32* max_kept_out, hipStream_t s, const u32* halo_viol) {
  static_assert(sizeof(DevPoint) % 8 == 0, "blocks are exchanged as 64-bit words");
  hipLaunchKernelGGL(shard_pack_kernel, dim3(max_local ? (max_local + 255) / 256 : 1), dim3(256), 0, s, own_w, keep, local_pts, n_local,
                     max_local, prefix_f, n_matches, prefix_g, T, block, block_cap, frame_cap, rank_kept, N, max_kept_out, halo_viol);
}

// after exchange 2: every block's points to frame[seq]; one thread per 64-bit word (13 per point)
__global__ void __launch_bounds__(256) shard_scatter_kernel(const unsigned long long* __restrict__ blocks, size_t block_words, u32 max_kept,
                                                            DevPoint* __restrict__ frame, u32 frame_cap, u32* __restrict__ viol_total) {
  constexpr u32 WP = sizeof(DevPoint) / 8;
  const unsigned long long* blk = blocks + (size_t)blockIdx.y * block_words;
  u32 cnt = (u32)blk[0];
  if (viol_total && blockIdx.x == 0 && threadIdx.x == 0 && (blk[0] >> 32)) atomicAdd(viol_total, (u32)(blk[0] >> 32));
  if (cnt > max_kept) cnt = max_kept;
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  const u32 kk = t / WP, wd = t % WP;
  if (kk >= cnt) return;
  const DevPoint* p = reinterpret_cast<const DevPoint*>(blk + 1) + kk;
  const u32 idx = p->seq;
  if (idx >= frame_cap) return;
  reinterpret_cast<unsigned long long*>(frame + idx)[wd] = reinterpret_cast<const unsigned long long*>(p)[wd];
}
void launch_shard_scatter(const unsigned long long* blocks, size_t block_words, u32 N, u32 max_kept, DevPoint* frame, u32 frame_cap,
                          hipStream_t s, u32* viol_total) {
  // (no kept point anywhere: unrouted, the exchange did not take place; routed, it carried the count words alone -- the ranks'
  //  halo violations are summed from them)
  if ((max_kept == 0 && !viol_total) || frame_cap == 0) return;
  constexpr u32 WP = sizeof(DevPoint) / 8;
  hipLaunchKernelGGL(shard_scatter_kernel, dim3(max_kept ? (max_kept * WP + 255) / 256 : 1, N), dim3(256), 0, s, blocks, block_words,
                     max_kept, frame, frame_cap, viol_total);
}

}  // namespace esvo
