// Microbenchmark of the Student-t scale iteration of lm_refine (one iteration = 7 residuals per lane,
// 16-lane DPP reduction, s2 = sum/N, convergence test).  Variants isolate where the cycles go.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I esvo_amd/csrc -I include tools/lm_loop_bench.hip -o /tmp/lmb && /tmp/lmb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "fdiv.hpp"
using namespace esvo;

template <int CTRL> __device__ inline double dpp_f64(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  return __hiloint2double(__builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true), __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true));
}
__device__ inline double grp_sum(double v) {
  v = v + dpp_f64<0xB1>(v); v = v + dpp_f64<0x4E>(v); v = v + dpp_f64<0x141>(v); v = v + dpp_f64<0x140>(v);
  return v;
}

// VARIANT 0: literal (2 divisions per element) 1: shared reciprocal for r2/s1  2: no inner division (mul, wrong numerics)
// 3: no divisions at all in the element loop  4: literal without the reduction  5: literal without sum/N and test divisions
template <int VARIANT>
__global__ void __launch_bounds__(256) k(const double* in, double* out, int iters) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  double r2[7], a[7];
  const double nu = 2.182;
  for (int y = 0; y < 7; ++y) { r2[y] = in[(t * 7 + y) & 4095] + 1.0; a[y] = r2[y] * (nu + 1); }
  double s1 = 298.5, acc = 0;
  for (int it = 0; it < iters; ++it) {
    double tt[7];
    if (VARIANT == 1) {
      const Recip rs = make_recip(s1);
#pragma unroll
      for (int y = 0; y < 7; ++y) tt[y] = a[y] / (nu + div_fast(r2[y], rs));
    } else if (VARIANT == 2) {
      const double inv = 1.0 / s1;
#pragma unroll
      for (int y = 0; y < 7; ++y) tt[y] = a[y] / (nu + r2[y] * inv);
    } else if (VARIANT == 3) {
#pragma unroll
      for (int y = 0; y < 7; ++y) tt[y] = a[y] * (nu + r2[y] * s1);
    } else {
#pragma unroll
      for (int y = 0; y < 7; ++y) tt[y] = a[y] / (nu + r2[y] / s1);
    }
    double c = tt[0];
#pragma unroll
    for (int y = 1; y < 7; ++y) c = c + tt[y];
    const double sum = (VARIANT == 4) ? c : grp_sum(c);
    double s2 = (VARIANT == 5) ? sum * 0.0095 : sum / 105.0;
    const bool go = (VARIANT == 5) ? (fabs(s2 - s1) > 0.05 * s1) : (fabs(s2 - s1) / s1 > 0.05);
    acc += go ? 1.0 : 0.5;
    s1 = 250.0 + 1e-9 * s2;  // keeps the dependence on s2 without converging
  }
  out[t] = acc + s1;
}

template <int V> void run(const double* din, double* dout, int blocks, int iters, const char* name) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, din, dout, 10);
  hipEventRecord(e0); hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, din, dout, iters); hipEventRecord(e1);
  hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
  const double waves = blocks * 4.0, simds = 1024.0;
  printf("%-44s %8.3f ms  %7.1f ns/iter/wave-slot  (%.0f cycles of a SIMD per wave-iteration at 2.4 GHz)\n", name, ms,
         ms * 1e6 / iters / (waves / simds), ms * 1e6 / iters / (waves / simds) * 2.4);
}

int main() {
  std::vector<double> h(4096); for (int i = 0; i < 4096; ++i) h[i] = (i * 37 % 1000) * 0.37;
  double *din, *dout; const int blocks = 256 * 12;  // 12 blocks (48 waves) per CU: 3 waves per SIMD resident at 168 VGPRs is not modelled
  hipMalloc((void**)&din, 4096 * 8); hipMalloc((void**)&dout, (size_t)blocks * 256 * 8);
  hipMemcpy(din, h.data(), 4096 * 8, hipMemcpyHostToDevice);
  const int iters = 2000;
  run<0>(din, dout, blocks, iters, "0 literal (2 div/elem)");
  run<1>(din, dout, blocks, iters, "1 shared reciprocal for r2/s1");
  run<2>(din, dout, blocks, iters, "2 inner division -> multiply");
  run<3>(din, dout, blocks, iters, "3 no division in the element loop");
  run<4>(din, dout, blocks, iters, "4 literal, no DPP reduction");
  run<5>(din, dout, blocks, iters, "5 literal, no sum/N + test divisions");
  return 0;
}
