// tools/dev_hooks/hooks_impl.hpp -- bodies of the instrumentation hooks of esvo_amd/csrc/dev_hooks.hpp.  NOT product code: compiled
// only into the tools' variants of the library (tools/ab_build.py <name> -DLM_STATS | -DREG_STATS) and into the parity twin
// (esvo_amd/lib.py build(perturbed=True): -DESVO_PERTURB_ONE_ULP).
#pragma once
#include <hip/hip_runtime.h>

#ifdef LM_STATS
// in-kernel counters of lm_refine_kernel (tools/lm_stats.py, tools/lm_attribution.py)
#ifdef DEV_HOOKS_LM_TU
__device__ unsigned long long g_lm_dbg[8];
__device__ unsigned int g_lm_slot[3][1 << 18];  // per solver slot: evaluations, t-scale iterations, those of the first evaluation
extern "C" void esvo_debug_lm_counters(unsigned long long out[8]) { hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lm_dbg), 64); }
extern "C" void esvo_debug_lm_slots(unsigned int* out, int clear) {  // out[3][1 << 18]
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lm_slot), sizeof(unsigned int) * 3 * (1 << 18));
  if (clear) { void* p = nullptr; hipGetSymbolAddress(&p, HIP_SYMBOL(g_lm_slot)); hipMemset(p, 0, sizeof(unsigned int) * 3 * (1 << 18)); }
}
#endif
#define LM_COUNT(i, cond) do { if (cond) atomicAdd(&g_lm_dbg[i], 1ull); } while (0)
#define LM_SLOT(k, s, cond, v) do { if ((cond) && (s) < (1u << 18)) g_lm_slot[k][s] += (v); } while (0)
#define DEV_LM_EVAL(pr) do { LM_COUNT(0, (pr).c == 0); LM_SLOT(0, (pr).dbg_slot, (pr).c == 0, 1u); LM_COUNT(1, __lane_id() == __ffsll(__ballot(1)) - 1); } while (0)
#define DEV_LM_SHORTCUT(pr) LM_COUNT(4, (pr).c == 0)   /* the provable outcome of the uncapped scale loop was taken */
#define DEV_LM_PROBLEM_FIELDS unsigned int dbg_slot;
#define DEV_LM_SET_SLOT(pr, s) do { (pr).dbg_slot = (s); } while (0)
// the first evaluation of a match: how many of its residuals are non-zero
#define DEV_LM_FIRST_EVAL_KNZ(pr, knz) do { if ((pr).c == 0 && (pr).dbg_slot < (1u << 18) && g_lm_slot[0][(pr).dbg_slot] == 1u) g_lm_slot[2][(pr).dbg_slot] = 1000u * (unsigned int)(knz); } while (0)
#define DEV_LM_SCALE_ITER(pr) do { \
    LM_COUNT(2, (pr).c == 0); \
    LM_SLOT(1, (pr).dbg_slot, (pr).c == 0, 1u); \
    LM_SLOT(2, (pr).dbg_slot, (pr).c == 0 && g_lm_slot[0][(pr).dbg_slot < (1u << 18) ? (pr).dbg_slot : 0] == 1u, 1u); \
    LM_COUNT(3, __lane_id() == __ffsll(__ballot(1)) - 1); } while (0)
#define DEV_LM_JAC_DECL double dbg_xjac = __longlong_as_double(0x7ff8000000000000ll);
#define DEV_LM_JAC_PASS(pr, active, x) do { LM_COUNT(5, (pr).c == 0 && (active)); LM_COUNT(6, (pr).c == 0 && (active) && (x) == dbg_xjac); dbg_xjac = (x); } while (0)
#endif  // LM_STATS

#ifdef ESVO_PERTURB_ONE_ULP
// libesvo_hip_perturbed.so only: the depth of every eighth solver slot's point one unit in the last place off (p_cam is what the
// fusion propagates) -- the deliberate defect tests/test_gpu_bench_parity.py uses to show that bench.py's parity.oracle_equal has teeth
#define DEV_PERTURB_POINT(o, s) do { if (((s) & 7u) == 7u) (o).p_cam[2] = __longlong_as_double(__double_as_longlong((o).p_cam[2]) ^ 1ll); } while (0)
#endif

#ifdef REG_STATS
// tools/reg_floor.py: [0] the longest chain of a WORKGROUP of reg_apply_kernel -- per block of staged rows the slowest wave's sum over
// the rows of its busiest lane's Student-t steps, summed over the blocks -- [1] the longest chain of a single element (its close
// taps), [2] Student-t steps executed by waves (lockstep: a wave steps through a row as often as its busiest lane), [3] close taps;
// maxima / sums over the launches since the last read-out
#ifdef DEV_HOOKS_FUSE_TU
__device__ unsigned long long g_reg_stats[8];
extern "C" void esvo_debug_reg_stats(unsigned long long out[8], int clear) {
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(out, HIP_SYMBOL(g_reg_stats), 64);
  if (clear) { unsigned long long z[8] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_reg_stats), z, 64); }
}
#endif
#define DEV_REG_DECL __shared__ unsigned int s_blk_chain[REG_TY]; unsigned int st_wg_chain = 0, st_wave_blk = 0, st_wave_steps = 0;
#define DEV_REG_BLOCK_BEGIN() do { st_wave_blk = 0; } while (0)
#define DEV_REG_ROW_BEGIN(nclose) const unsigned int st_before = (nclose);
#define DEV_REG_ROW_END(nclose) do { \
    unsigned int mx = (nclose) - st_before; \
    for (int d = 32; d >= 1; d >>= 1) mx = max(mx, (unsigned int)__shfl_xor((int)mx, d)); \
    st_wave_blk += mx; st_wave_steps += mx; } while (0)
#define DEV_REG_BLOCK_END(lane, wv) do { \
    if ((lane) == 0) s_blk_chain[wv] = st_wave_blk; \
    __syncthreads(); \
    unsigned int wg = 0; \
    for (int w = 0; w < REG_TY; ++w) wg = max(wg, s_blk_chain[w]); \
    st_wg_chain += wg; \
    __syncthreads(); } while (0)
#define DEV_REG_DONE(t, lane, nclose) do { \
    if ((t) == 0) atomicMax(&g_reg_stats[0], (unsigned long long)st_wg_chain); \
    atomicMax(&g_reg_stats[1], (unsigned long long)(nclose)); \
    atomicAdd(&g_reg_stats[3], (unsigned long long)(nclose)); \
    if ((lane) == 0) atomicAdd(&g_reg_stats[2], (unsigned long long)st_wave_steps); } while (0)
#endif  // REG_STATS
