// tools/reg_step_bench.hip -- latency of ONE Student-t fusion step of the regulariser (DepthRegularization.cpp:72-86 as
// reg_apply_kernel executes it: kernels_fuse.hip, fuse_step): a single wave runs N dependent steps, time / N.
//   hipcc --offload-arch=gfx950 -O3 -I esvo_amd/csrc -I include tools/reg_step_bench.hip -o /tmp/reg_step_bench && /tmp/reg_step_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include "fdiv.hpp"
using namespace esvo;
__global__ void steps(double* io, int n) {
  double nu_post = io[0], inv_post = io[1], s2_post = io[2], nu_div = nu_post + 1;
  Recip rnu = make_recip(nu_div);
  const double nu_obs = io[3], s2_obs = io[5];
  double inv_obs = io[4];
  for (int i = 0; i < n; ++i) {
    const double nu_prior = nu_post, inv_prior = inv_post, s2_prior = s2_post;
    nu_post = (nu_obs < nu_prior) ? nu_obs : nu_prior;
    if (nu_post + 1 != nu_div) { nu_div = nu_post + 1; rnu = make_recip(nu_div); }
    const double ssum = s2_obs + s2_prior;
    const Recip rsum = make_recip(ssum);
    const double a1 = s2_obs * inv_prior + s2_prior * inv_obs;
    const double dd = inv_prior - inv_obs;
    const double a2 = dd * dd;
    const double pp = s2_prior * s2_obs;
    const double q1 = div_fast(a1, rsum);
    const double a3 = nu_post + div_fast(a2, rsum);
    const double a4 = div_fast(a3, rnu) * pp;
    const double q4 = div_fast(a4, rsum);
    const bool ok = (int)rnu.fast & (int)fdiv_ok_b4(ssum, a1, a2, a3, a4);
    if (ok) { inv_post = q1; s2_post = q4; } else { inv_post = a1 / ssum; s2_post = ((nu_post + a2 / ssum) / (nu_post + 1) * pp) / ssum; }
    inv_obs += 1e-9;
  }
  io[6 + threadIdx.x % 2] = inv_post + s2_post + nu_post;
}
int main() {
  double h[8] = {2.2, 0.5, 1e-3, 2.2, 0.5001, 1.1e-3, 0, 0};
  double* d;
  hipMalloc(&d, sizeof(h));
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int waves : {1, 2048}) {
    const int n = 20000;
    hipLaunchKernelGGL(steps, dim3(waves), dim3(64), 0, 0, d, n);
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(steps, dim3(waves), dim3(64), 0, 0, d, n);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    printf("%d wave(s): %.1f ns per dependent Student-t step\n", waves, ms * 1e6 / n);
  }
  return 0;
}
