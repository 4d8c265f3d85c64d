// dev_hooks.hpp -- where the tools' instrumentation attaches to the kernels.  In the product build every hook below is EMPTY;
// a build with -DLM_STATS / -DREG_STATS (tools/README.md: in-kernel counters of the refinement and of the regulariser) or
// -DESVO_PERTURB_ONE_ULP (the deliberately wrong twin library tests/test_gpu_bench_parity.py loads) pulls their bodies in from
// tools/dev_hooks/hooks_impl.hpp -- no instrumentation code lives in the shipping sources.
#pragma once
#if defined(LM_STATS) || defined(REG_STATS) || defined(ESVO_PERTURB_ONE_ULP)
#include "../../tools/dev_hooks/hooks_impl.hpp"
#endif
// ---- refinement (kernels_lm.hip) ----
#ifndef LM_STATS
#define DEV_LM_EVAL(pr) do {} while (0)
#define DEV_LM_SHORTCUT(pr) do {} while (0)
#define DEV_LM_PROBLEM_FIELDS
#define DEV_LM_SET_SLOT(pr, s) do {} while (0)
#define DEV_LM_FIRST_EVAL_KNZ(pr, knz) do {} while (0)
#define DEV_LM_SCALE_ITER(pr) do {} while (0)
#define DEV_LM_JAC_DECL
#define DEV_LM_JAC_PASS(pr, active, x) do {} while (0)
#endif
#ifndef ESVO_PERTURB_ONE_ULP
#define DEV_PERTURB_POINT(o, s) do {} while (0)
#endif
// ---- regulariser (kernels_fuse.hip, reg_apply_kernel) ----
#ifndef REG_STATS
#define DEV_REG_DECL
#define DEV_REG_BLOCK_BEGIN() do {} while (0)
#define DEV_REG_ROW_BEGIN(nclose)
#define DEV_REG_ROW_END(nclose) do {} while (0)
#define DEV_REG_BLOCK_END(lane, wv) do {} while (0)
#define DEV_REG_DONE(t, lane, nclose) do {} while (0)
#endif
