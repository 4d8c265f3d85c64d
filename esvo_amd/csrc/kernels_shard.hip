// kernels_shard.hip — multi-GPU tick: putting the ranks' depth points in the reference's order.
//
// Per-event work is dealt round-robin: rank r block-matches and refines the slots w with
// w % N == r (kernels_bm.hip) and keeps its results in a dense local list, in increasing w.  The
// order of the tick's frame (vdp after pointCulling) is a function of two bits per slot only:
//   bit 0  the event was matched             -> position j of the match in vEMP (prefix over w)
//   bit 1  its point survived LM + culling   -> solver slot s = stride_slot(j, M, T)
//                                               (DepthProblemSolver.cpp:75-90), final index = prefix over s
// so the ranks exchange these bits (one byte per slot, summed: foreign bytes are zero), every rank
// derives the same order, writes its own points at their final indices into a zeroed frame, and a second
// sum over the frame completes it everywhere.  Nothing here touches a point's payload.
#include "common.hpp"

namespace esvo {

__global__ void __launch_bounds__(256) shard_codes_kernel(const u32* __restrict__ own_w, const u32* __restrict__ keep,
                                                          const u32* __restrict__ n_local, u32 max_local,
                                                          uint8_t* __restrict__ codes) {
  const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  u32 n = *n_local;
  if (n > max_local) n = max_local;
  if (k >= n) return;
  codes[own_w[k]] = (uint8_t)(1u | (keep[k] ? 2u : 0u));
}
void launch_shard_codes(const u32* own_w, const u32* keep, const u32* n_local, u32 max_local, uint8_t* codes, hipStream_t s) {
  if (max_local == 0) return;
  hipLaunchKernelGGL(shard_codes_kernel, dim3((max_local + 255) / 256), dim3(256), 0, s, own_w, keep, n_local, max_local, codes);
}

__global__ void __launch_bounds__(256) shard_match_flags_kernel(const uint8_t* __restrict__ codes, u32 n, u32* __restrict__ flags) {
  const u32 w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w < n) flags[w] = codes[w] & 1u;
}
void launch_shard_match_flags(const uint8_t* codes, u32 n, u32* flags, hipStream_t s) {
  if (n == 0) return;
  hipLaunchKernelGGL(shard_match_flags_kernel, dim3((n + 255) / 256), dim3(256), 0, s, codes, n, flags);
}

// keep_by_slot must be zero on entry
__global__ void __launch_bounds__(256) shard_keep_flags_kernel(const uint8_t* __restrict__ codes, const u32* __restrict__ prefix_f,
                                                               const u32* __restrict__ n_matches, u32 n, u32 T,
                                                               u32* __restrict__ keep_by_slot) {
  const u32 w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n) return;
  const u32 c = codes[w];
  if (!(c & 1u)) return;
  const u32 s = stride_slot(prefix_f[w], *n_matches, T);
  if (s < n) keep_by_slot[s] = (c >> 1) & 1u;
}
void launch_shard_keep_flags(const uint8_t* codes, const u32* prefix_f, const u32* n_matches, u32 n, u32 T, u32* keep_by_slot,
                             hipStream_t s) {
  if (n == 0) return;
  hipLaunchKernelGGL(shard_keep_flags_kernel, dim3((n + 255) / 256), dim3(256), 0, s, codes, prefix_f, n_matches, n, T,
                     keep_by_slot);
}

// frame[0, K) <- 0, then the own points at their final indices: two kernels, because the zeroing must be complete
// before any point is placed (K is only known on the device, so a hipMemsetAsync cannot be sized by the host)
__global__ void __launch_bounds__(256) shard_zero_frame_kernel(unsigned long long* __restrict__ words, const u32* __restrict__ n_points,
                                                               u32 frame_cap) {
  u32 K = *n_points;
  if (K > frame_cap) K = frame_cap;
  const size_t total = (size_t)K * (sizeof(DevPoint) / 8);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) words[i] = 0ull;
}
__global__ void __launch_bounds__(256) shard_place_kernel(const u32* __restrict__ own_w, const u32* __restrict__ keep,
                                                          const DevPoint* __restrict__ local_pts, const u32* __restrict__ n_local,
                                                          u32 max_local, const u32* __restrict__ prefix_f,
                                                          const u32* __restrict__ n_matches, const u32* __restrict__ prefix_g, u32 T,
                                                          DevPoint* __restrict__ frame, u32 frame_cap) {
  const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  u32 n = *n_local;
  if (n > max_local) n = max_local;
  if (k >= n || !keep[k]) return;
  const u32 s = stride_slot(prefix_f[own_w[k]], *n_matches, T);
  const u32 idx = prefix_g[s];
  if (idx >= frame_cap) return;
  DevPoint o = local_pts[k];
  o.seq = idx;  // as compact_points_kernel
  frame[idx] = o;
}
void launch_shard_place(const u32* own_w, const u32* keep, const DevPoint* local_pts, const u32* n_local, u32 max_local,
                        const u32* prefix_f, const u32* n_matches, const u32* prefix_g, const u32* n_points, u32 T, DevPoint* frame,
                        u32 frame_cap, hipStream_t s) {
  static_assert(sizeof(DevPoint) % 8 == 0, "frame is exchanged as 64-bit words");
  if (max_local == 0 || frame_cap == 0) return;
  hipLaunchKernelGGL(shard_zero_frame_kernel, dim3(512), dim3(256), 0, s, reinterpret_cast<unsigned long long*>(frame), n_points,
                     frame_cap);
  hipLaunchKernelGGL(shard_place_kernel, dim3((max_local + 255) / 256), dim3(256), 0, s, own_w, keep, local_pts, n_local, max_local,
                     prefix_f, n_matches, prefix_g, T, frame, frame_cap);
}

}  // namespace esvo
