// tools/valu_rates.hip -- issue cost of the instructions the LM kernel is made of, on gfx950: cycles of a SIMD per wave64
// instruction at 1, 2 and 4 resident waves per SIMD, for independent chains (ILP 8) and one dependent chain (latency), and the
// cost of ONE tight t-scale iteration of lm_refine_kernel (kernels_lm.hip: residual_eval's tight loop, 7 rows per lane) as the
// kernel executes it, with its reciprocals replaced by multiplies, and without its reduction.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I esvo_amd/csrc -I include tools/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "fdiv.hpp"
using namespace esvo;

enum Op { FMA64, MUL64, ADD64, RCP64, RSQ64, FMA32, RCP32, DPPMOV, SWAP16, CVT6432 };
template <Op OP> __device__ inline void op1(double& x, double c) {
  if constexpr (OP == FMA64) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x) : "v"(c));
  if constexpr (OP == MUL64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(c));
  if constexpr (OP == ADD64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(c));
  if constexpr (OP == RCP64) asm volatile("v_rcp_f64 %0, %0" : "+v"(x));
  if constexpr (OP == RSQ64) asm volatile("v_rsq_f64 %0, %0" : "+v"(x));
  if constexpr (OP == FMA32) { float& f = reinterpret_cast<float&>(x); float cf = (float)c; asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f) : "v"(cf)); }
  if constexpr (OP == RCP32) { float& f = reinterpret_cast<float&>(x); asm volatile("v_rcp_f32 %0, %0" : "+v"(f)); }
  if constexpr (OP == DPPMOV) { int& f = reinterpret_cast<int&>(x); asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(f)); }
  if constexpr (OP == SWAP16) { int* f = reinterpret_cast<int*>(&x); asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(f[0]), "+v"(f[1])); }
  if constexpr (OP == CVT6432) { float t; asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(t) : "v"(x)); asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(x) : "v"(t)); }
}
__device__ unsigned long long g_clk[2];  // shader cycles, 100 MHz reference ticks of wave 0 of the last launch
template <Op OP, int ILP> __global__ void __launch_bounds__(64) rate(double* io, int n) {
  const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  double x[ILP];
  for (int i = 0; i < ILP; ++i) x[i] = io[i] + threadIdx.x * 1e-3;
  const double c = io[9];
  for (int it = 0; it < n; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < ILP; ++i) op1<OP>(x[i], c);
  }
  double s = 0;
  for (int i = 0; i < ILP; ++i) s += x[i];
  if (s == 12345.678) io[10] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { g_clk[0] = __builtin_readcyclecounter() - c0; g_clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }
}

template <int CTRL> __device__ inline double dpp_f64(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  return __hiloint2double(__builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true), __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true));
}
__device__ inline double grp_sum(double v) {
  v = v + dpp_f64<0xB1>(v); v = v + dpp_f64<0x4E>(v); v = v + dpp_f64<0x141>(v); v = v + dpp_f64<0x140>(v);
  return v;
}
// the tight t-scale iteration.  V 0: as shipped  1: v_rcp_f64 of the per-row divisors -> a multiply (wrong numbers: what the seven
// reciprocals cost)  2: no DPP butterfly  3: per-row reciprocal seeded by v_rcp_f32 + one more Newton step (numbers may differ)
template <int V> __global__ void __launch_bounds__(64) scale_iter(const double* in, double* out, int iters) {
  const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  const int t = blockIdx.x * 64 + threadIdx.x;
  double r2[7], r2n[7];
  const double nu = in[4095] * 0 + 2.182;
  for (int y = 0; y < 7; ++y) { r2[y] = in[(t * 7 + y) & 4095] + 1.0; r2n[y] = r2[y] * (nu + 1); }
  double s1 = 298.5, acc = 0;
  const Recip rN = make_recip(105.0);
  for (int it = 0; it < iters; ++it) {
    Recip rs1; rs1.b = s1; rs1.y = recip_refined(s1);
    double tt[7];
#pragma unroll
    for (int y = 0; y < 7; ++y) {
      Recip rd; rd.b = nu + div_fast(r2[y], rs1);
      if (V == 1) { double yy = rd.b * 0.37; yy = __builtin_fma(yy, __builtin_fma(-rd.b, yy, 1.0), yy); rd.y = __builtin_fma(yy, __builtin_fma(-rd.b, yy, 1.0), yy); }
      else if (V == 3) {
        double yy = (double)__builtin_amdgcn_rcpf((float)rd.b);
        yy = __builtin_fma(yy, __builtin_fma(-rd.b, yy, 1.0), yy);
        yy = __builtin_fma(yy, __builtin_fma(-rd.b, yy, 1.0), yy);
        rd.y = __builtin_fma(yy, __builtin_fma(-rd.b, yy, 1.0), yy);
      } else rd.y = recip_refined(rd.b);
      tt[y] = div_fast(r2n[y], rd);
    }
    double c = tt[0];
#pragma unroll
    for (int y = 1; y < 7; ++y) c = c + tt[y];
    const double sum = (V == 2) ? c : grp_sum(c);
    const double s2 = div_fast(sum, rN);
    const double rel = div_fast(fabs(s2 - s1), rs1);
    acc += (rel > 0.05) ? 1.0 : 0.5;
    s1 = 250.0 + 1e-9 * s2;  // keeps the dependence on s2 without converging
  }
  out[t] = acc + s1;
  if (blockIdx.x == 0 && threadIdx.x == 0) { g_clk[0] = __builtin_readcyclecounter() - c0; g_clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }
}

static hipEvent_t e0, e1;
template <class F> float timed(F&& launch) {
  launch(); hipDeviceSynchronize();
  hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
static double clk_ghz = 2.4;
// shader clock of the last launch, GHz (s_memrealtime ticks at 100 MHz)
static double last_clk() {
  unsigned long long h[2];
  hipMemcpyFromSymbol(h, HIP_SYMBOL(g_clk), sizeof(h));
  return h[1] ? (double)h[0] / (double)h[1] * 0.1 : 0.0;
}
template <Op OP> void rates(double* d, const char* name) {
  printf("%-22s", name);
  const int n = 4000;
  for (int wps : {1, 2, 4}) {
    const int waves = 1024 * wps;
    // cycles of a SIMD per wave instruction: time * clk / (instructions per wave * waves per SIMD)
    const float a = timed([&] { hipLaunchKernelGGL((rate<OP, 8>), dim3(waves), dim3(64), 0, 0, d, n); });
    const double ca = last_clk();
    const float b = timed([&] { hipLaunchKernelGGL((rate<OP, 1>), dim3(waves), dim3(64), 0, 0, d, n); });
    const double cb = last_clk();
    printf("  %dw: ilp8 %5.2f @%.2f dep %5.2f @%.2f", wps, a * 1e6 * ca / ((double)n * 64 * wps), ca, b * 1e6 * cb / ((double)n * 8 * wps), cb);
  }
  printf("   (cycles of a SIMD per wave64 instruction; dep: x waves interleaving one chain each)\n");
}
template <int V> void iter(const double* din, double* dout, const char* name) {
  printf("%-58s", name);
  const int n = 3000;
  for (int wps : {1, 2, 3, 4}) {
    const int waves = 1024 * wps;
    const float a = timed([&] { hipLaunchKernelGGL(scale_iter<V>, dim3(waves), dim3(64), 0, 0, din, dout, n); });
    const double c = last_clk();
    printf("  %dw %6.1f @%.2f", wps, a * 1e6 * c / ((double)n * wps), c);
  }
  printf("   (SIMD cycles per wave-iteration)\n");
}
int main(int argc, char** argv) {
  if (argc > 1) clk_ghz = atof(argv[1]);
  hipEventCreate(&e0); hipEventCreate(&e1);
  std::vector<double> h(4096); for (int i = 0; i < 4096; ++i) h[i] = 1.0 + (i * 37 % 1000) * 0.37;
  h[9] = 1.0000001;
  double *din, *dout;
  hipMalloc((void**)&din, 4096 * 8); hipMalloc((void**)&dout, (size_t)4096 * 64 * 8);
  hipMemcpy(din, h.data(), 4096 * 8, hipMemcpyHostToDevice);
  printf("cycles at the shader clock measured inside each launch (@GHz: s_memtime / s_memrealtime of one wave)\n");
  rates<FMA64>(din, "v_fma_f64"); rates<MUL64>(din, "v_mul_f64"); rates<ADD64>(din, "v_add_f64");
  rates<RCP64>(din, "v_rcp_f64"); rates<RSQ64>(din, "v_rsq_f64"); rates<FMA32>(din, "v_fma_f32"); rates<RCP32>(din, "v_rcp_f32");
  rates<DPPMOV>(din, "v_mov_b32_dpp"); rates<SWAP16>(din, "v_permlane16_swap"); rates<CVT6432>(din, "cvt f64->f32->f64 (2)");
  iter<0>(din, dout, "t-scale iteration as shipped (tight loop)");
  iter<1>(din, dout, "  per-row v_rcp_f64 -> multiply");
  iter<2>(din, dout, "  no DPP butterfly");
  iter<3>(din, dout, "  per-row seed v_rcp_f32 + 3 Newton steps");
  return 0;
}
