// kernels_viz.hip — the mapper's debug images.
//
// Replaces Visualization::plot_map + DrawPoint (esvo_core/src/tools/Visualization.cpp:13-94) as
// esvo_Mapping::publishMappingResults calls them (esvo_core/src/esvo_Mapping.cpp:868-884): every element of the DepthMap
// that passes the image's filter paints a filled radius-1 circle (cv::circle(..., 1, color, cv::FILLED): the 5-pixel
// plus) in the jet colour of its value at (int)x.  The reference walks its element list, so where circles overlap the
// LAST element wins; on the device every pixel first learns the largest list position (seq) that covers it
// (atomicMax), then exactly that element paints it -- the same image, whatever order the threads run in.
#include "common.hpp"

namespace esvo {

struct VizSel {
  int type;  // 0 InvDepthMap, 1 StdVarMap, 2 CostMap, 3 AgeMap (Visualization.h:11-17)
  double max_range, min_range, thr1, thr2;
};

// value of the element for the image, or false if the image's filter drops it (Visualization.cpp:28-64)
__device__ inline bool viz_value(const MapCell& c, const VizSel& s, double& val) {
  if (!(c.inv_depth > -1e-6)) return false;  // it->valid() (the caller has checked that the cell holds an element)
  switch (s.type) {
    case 0: val = c.inv_depth; return c.variance < s.thr1 * s.thr1 && (double)c.age >= (double)(int)s.thr2;
    case 1: val = sqrt(c.variance); return c.variance < s.thr1 * s.thr1;
    case 2: val = c.residual; return c.residual < s.thr1;
    default: val = (double)c.age; return (double)c.age >= (double)(int)s.thr1;
  }
}
template <bool PAINT>
__global__ void __launch_bounds__(256) viz_kernel(const MapCell* __restrict__ map, u32* __restrict__ owner,
                                                  uint8_t* __restrict__ bgr, const uint8_t* __restrict__ jet, VizSel s, int W, int H,
                                                  int band0, int band1) {
  const int cell = blockIdx.x * blockDim.x + threadIdx.x;
  if (cell >= W * H) return;
  { const int row = cell / W; if (row < band0 || row >= band1) return; }
  if (!(map_flags(map, W * H)[cell] & CELL_ALIVE)) return;
  const MapCell c = map[cell];
  double val;
  if (!viz_value(c, s, val)) return;
  int index = (int)floor((val - s.min_range) / (s.max_range - s.min_range) * 255.0);  // DrawPoint, Visualization.cpp:82
  index = index > 255 ? 255 : (index < 0 ? 0 : index);
  const int cx = (int)c.x[0], cy = (int)c.x[1];  // cv::Point from doubles
  const int dx[5] = {0, -1, 1, 0, 0}, dy[5] = {0, 0, 0, -1, 1};
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const int px = cx + dx[k], py = cy + dy[k];
    if (px < 0 || px >= W || py < 0 || py >= H) continue;
    const int pix = py * W + px;
    if (!PAINT) {
      atomicMax(&owner[pix], c.seq + 1u);
    } else if (owner[pix] == c.seq + 1u) {
      bgr[3 * pix + 0] = jet[3 * index + 0];
      bgr[3 * pix + 1] = jet[3 * index + 1];
      bgr[3 * pix + 2] = jet[3 * index + 2];
    }
  }
}

void launch_debug_image(const MapCell* map, u32* owner, uint8_t* bgr, const uint8_t* jet, int type, double max_range,
                        double min_range, double thr1, double thr2, const DevParams& p, hipStream_t s) {
  const int ncell = p.W * p.H;
  VizSel sel{type, max_range, min_range, thr1, thr2};
  hipMemsetAsync(owner, 0, sizeof(u32) * ncell, s);
  hipMemsetAsync(bgr, 0, (size_t)ncell * 3, s);
  const dim3 grid((ncell + 255) / 256), block(256);
  hipLaunchKernelGGL(viz_kernel<false>, grid, block, 0, s, map, owner, bgr, jet, sel, p.W, p.H, p.band_y0, p.band_y1);
  hipLaunchKernelGGL(viz_kernel<true>, grid, block, 0, s, map, owner, bgr, jet, sel, p.W, p.H, p.band_y0, p.band_y1);
}

}  // namespace esvo
