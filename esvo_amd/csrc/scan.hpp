// scan.hpp — block-level building blocks of the exclusive prefix sums (scan.hip, and the fused cell scan of kernels_fuse.hip)
#pragma once
#include "common.hpp"

namespace esvo {

static constexpr int SCAN_B = 256;            // threads per block (4 waves)
static constexpr int SCAN_V = 8;              // items per thread
static constexpr int SCAN_TILE = SCAN_B * SCAN_V;

__device__ inline u32 wave_incl_scan(u32 v, int lane) {
#pragma unroll
  for (int d = 1; d < ESVO_WAVE; d <<= 1) {
    u32 t = __shfl_up(v, d, ESVO_WAVE);
    if (lane >= d) v += t;
  }
  return v;
}

// exclusive scan of one value per thread across the block (NW waves); returns block total in *total
template <int NW = SCAN_B / ESVO_WAVE>
__device__ inline u32 block_excl_scan(u32 v, u32* total, u32* lds /*>= NW*/) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  u32 incl = wave_incl_scan(v, lane);
  if (lane == 63) lds[wave] = incl;
  __syncthreads();
  u32 base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    u32 s = lds[w];
    if (w < wave) base += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return base + incl - v;
}

}  // namespace esvo
