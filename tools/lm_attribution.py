#!/usr/bin/env python
"""Where the LM kernel's vector instructions go (verdict round 4, item 2a).

  static   python tools/lm_attribution.py static            (no GPU) compiles kernels_lm.hip to gfx950 assembly with line tables
           (-gline-tables-only: same registers, same instruction count as the shipped build -- checked) and counts the VALU
           instructions of the NARROW lm_refine_kernel per source region; the t-scale loop body is taken from the compiler's
           loop annotation.  -> profiles/<tag>_lm_static.json
  dynamic  python tools/lm_attribution.py dynamic           (GPU, ESVO_HIP_LIB = a -DLM_STATS build: tools/ab_build.py lmstats
           -DLM_STATS) runs the headline workload and reads the kernel's own counters: evaluations and t-scale iterations per 16-lane
           group (= per match) and per wave (= what is executed), shortcuts, Jacobian evaluations.  -> gpurun_out/<tag>_lm_dynamic.json
  table    python tools/lm_attribution.py table STATIC.json DYNAMIC.json [SQ_INSTS_VALU per launch] [SQ_WAVES per launch]
           static counts x dynamic multiplicities, against the hardware counter.
Regions are line ranges of esvo_amd/csrc/kernels_lm.hip found by their opening statements (so the tool follows edits); an
instruction inlined from fdiv.hpp / lm_common.hpp / the HIP headers belongs to the region of the last kernels_lm.hip line before it."""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "esvo_amd", "csrc", "kernels_lm.hip")
KERNEL = "lm_refine_kernelILb0ELb0ELi0ELb0ELb0EEE"   # narrow layout, Student-t, single launch, no band guard

MARKS = [  # (region, first line = the line holding this text); a region lasts until the next mark
    ("interpolation (geometry, loads, bilinear)", "__device__ inline PatchGeom interp_geom"),
    ("projection (cam2World, T, world2Cam, 4 quotients)", "__device__ bool lm_eval("),
    ("interpolation (geometry, loads, bilinear)", "const PatchGeom g1 = interp_geom(p, x1u, x1v)"),
    ("failure fill", "if (!okw) {  // failure fill"),
    ("residuals, moments, range tests", "  int knz = 0;"),
    ("t-scale: shortcut test + set-up", "if ((double)knz * (nu + 1) / (double)N <= 0.94"),
    ("t-scale loop (tight)", "while ((unsigned)(((__double2hiint(s1) >> 20) & 0xfff) - 923) <= 200u) {"),
    ("t-scale loop (general, rare)", "    while (!done) {"),
    ("weights sqrt((nu+1)/(nu+r^2/s2)) r (tight)", "// The weights sqrt((nu + 1) / (nu + r^2 / s2))"),
    ("weights (general, rare)", "  const Recip rs2 = make_recip(s2);"),
    ("kernel prologue (match, pose, set-up)", "lm_refine_kernel(LmArgs a, DevParams p, u32* n_solved, LmSplit sp) {"),
    ("solver: lmpar + trial point", "    if (need_step) {  // determine the LM parameter"),
    ("solver: evaluator call site / pair exchange", "    bool out_tight;"),
    ("solver: phase 0 (minimizeInit)", "    if (phase == 0) {  // minimizeInit"),
    ("solver: phase 1 (forward difference, J, qtf)", "    } else if (phase == 1) {"),
    ("solver: phase 2 (trust-region test)", "    } else {  // phase 2: trust-region trial at xnew"),
    ("solver: outer loop (DepthProblemSolver.cpp:161-188)", "    if (phase == 0) {\n"),
    ("kernel epilogue (point, culling, store)", "  if (!active || !lead) return;\n  if constexpr (BAND) { if (viol) atomicAdd(a.halo_viol, 1u); }"),
    ("(other kernels / host)", "// ---- the persistent narrow layout (round 5)"),
]


def regions():
    text = open(SRC).read()
    out = []
    for name, mark in MARKS:
        i = text.find(mark)
        assert i >= 0, mark
        out.append((text.count("\n", 0, i) + 1, name))
    out.sort()
    return out


def region_of(line, regs):
    name = "(before the first region)"
    for first, n in regs:
        if line >= first:
            name = n
        else:
            break
    return name


def static(tag):
    asm = "/tmp/lm_attr.s"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-gline-tables-only",
                           "-I", os.path.join(ROOT, "include"), "-S", "--cuda-device-only", "-o", asm, SRC], stderr=subprocess.DEVNULL)
    lines = open(asm).read().split("\n")
    start = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and KERNEL in l.split(":")[0]][0]
    end = [i for i, l in enumerate(lines) if i > start and l.strip().startswith(".Lfunc_end")][0]
    regs = regions()
    helpers_end = [a for a, n in regs if n.startswith("interpolation")][0]
    cur_line, loop = 0, None
    count, f64, loops, quarter, loops_q = {}, {}, {}, {}, {}
    QUARTER = ("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32")
    for l in lines[start + 1:end]:
        t = l.strip()
        if t.startswith(".loc"):
            # the trailing comment holds the whole inlining chain, innermost first: file:line:col @[ caller:line:col @[ ... ] ].
            # The instruction belongs to the innermost kernels_lm.hip line OUTSIDE the generic helpers at the top of the file
            # (DPP sums, patch_sum: lines below interp_geom), i.e. to the statement of lm_eval / the kernel that needed it.
            chain = [int(x) for x in re.findall(r"kernels_lm\.hip:(\d+)", t)]
            pick = [x for x in chain if x >= helpers_end]
            if pick:
                cur_line = pick[0]
            elif chain:
                cur_line = chain[-1]
            continue
        m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", l) or re.match(r"^; %(bb\.\d+):\s*(;.*)?$", l)
        if m:
            c = m.group(2) or ""
            d = re.search(r"Depth=(\d+)", c)
            hd = re.search(r"Header=(BB\d+_\d+)", c)
            if "Parent Loop" in c:      # the header block of a nested loop is annotated with its parent (and the parent's depth)
                loop = (int(d.group(1)) + 1 if d else 2, m.group(1).replace(".L", ""))
            elif d and hd:
                loop = (int(d.group(1)), hd.group(1))
            else:
                loop = None
            continue
        if not t or t[0] in ";.":
            continue
        op = t.split()[0]
        if not op.startswith("v_"):
            continue
        r = region_of(cur_line, regs)
        count[r] = count.get(r, 0) + 1
        if "_f64" in op:
            f64[r] = f64.get(r, 0) + 1
        q = op.split("_e64")[0].split("_e32")[0] in QUARTER   # quarter-rate (transcendental unit): 16.3 cycles per wave64, tools/valu_rates.hip
        if q:
            quarter[r] = quarter.get(r, 0) + 1
        if loop and loop[0] >= 2:
            key = f"{r} @ inner loop {loop[1]}"
            loops[key] = loops.get(key, 0) + 1
            if q:
                loops_q[key] = loops_q.get(key, 0) + 1
    meta = next(l for l in lines[end:] if ".vgpr_count" in l) if any(".vgpr_count" in l for l in lines[end:]) else ""
    out = {"kernel": KERNEL, "valu_static_total": sum(count.values()), "valu_by_region": count, "f64_by_region": f64, "quarter_rate_by_region": quarter, "valu_in_inner_loops": loops,
           "quarter_rate_in_inner_loops": loops_q,
           "regions_first_line": {n: a for a, n in regs}}
    path = os.path.join(ROOT, "profiles", f"{tag}_lm_static.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out, indent=1))
    print("->", path, meta.strip())


def dynamic(tag, n_ticks=8):
    import ctypes
    sys.path.insert(0, ROOT)
    import bench
    from esvo_amd import lib
    rig, stream, p, ticks = bench.make_workload("dsec640x480", 40)
    dev = lib.Esvo(p, rig)
    dev.ts_push_events(0, stream.ev_left)
    dev.ts_push_events(1, stream.ev_right)
    L = lib.load()
    if not hasattr(L, "esvo_debug_lm_counters"):
        raise SystemExit("ESVO_HIP_LIB must name a -DLM_STATS build (python tools/ab_build.py lmstats -DLM_STATS)")
    out = (ctypes.c_ulonglong * 8)()
    bench.run_single(dev, stream, ticks, 0, 6)
    dev.synchronize()
    L.esvo_debug_lm_counters(out)
    base = list(out)
    b = dev.stats()
    bench.run_single(dev, stream, ticks, 6, 6 + n_ticks)
    dev.synchronize()
    s = dev.stats()
    L.esvo_debug_lm_counters(out)
    o = [int(x) - int(y) for x, y in zip(out, base)]
    m = int(s.total_matches - b.total_matches)
    res = {"ticks": n_ticks, "matches_per_tick": m / n_ticks, "events_per_tick": int(s.total_events_in - b.total_events_in) / n_ticks,
           "evals_per_group_per_tick": o[0] / n_ticks, "evals_per_wave_per_tick": o[1] / n_ticks,
           "tscale_iters_per_group_per_tick": o[2] / n_ticks, "tscale_iters_per_wave_per_tick": o[3] / n_ticks,
           "shortcuts_per_tick": o[4] / n_ticks, "jacobian_evals_per_tick": o[5] / n_ticks,
           "evals_per_match": o[0] / m, "tscale_iters_per_eval_group": o[2] / max(o[0], 1), "tscale_iters_per_eval_wave": o[3] / max(o[1], 1)}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    path = os.path.join(ROOT, "gpurun_out", f"{tag}_lm_dynamic.json")
    json.dump(res, open(path, "w"), indent=1)
    print(json.dumps(res, indent=1))


def table(static_json, dynamic_json, measured=None, sq_waves=None):
    S, D = json.load(open(static_json)), json.load(open(dynamic_json))
    R = S["valu_by_region"]
    Q = S.get("quarter_rate_by_region", {})
    loops = S["valu_in_inner_loops"]
    tight_key = max([k for k in loops if k.startswith("t-scale loop (tight)")] or [""], key=lambda k: loops.get(k, 0))
    tight_body = loops.get(tight_key, 0)
    tight_q = S.get("quarter_rate_in_inner_loops", {}).get(tight_key, 0)
    # measured issue cost per wave64 instruction at the kernel's two waves per SIMD (tools/valu_rates.hip, profiles/r06_valu_rates.txt)
    C_PLAIN, C_QUARTER = 4.5, 16.5
    W = D["matches_per_tick"] / 4.0                      # working waves (four matches each)
    E, Lw = D["evals_per_wave_per_tick"], D["tscale_iters_per_wave_per_tick"]
    Eg, Lg = D["evals_per_group_per_tick"] / 4.0, D["tscale_iters_per_group_per_tick"] / 4.0   # the same in wave units without lockstep loss
    jac = D["jacobian_evals_per_tick"] / 4.0 * (E / max(Eg, 1))   # phase-1 passes per wave (scaled like the evaluations)
    rows = []

    def add(name, per, mult, what, q=0):
        rows.append((name, per, mult, per * mult, what, ((per - q) * C_PLAIN + q * C_QUARTER) * mult))
    g = lambda k: R.get(k, 0)
    gq = lambda *ks: sum(Q.get(k, 0) for k in ks)
    add("prologue + epilogue", g("kernel prologue (match, pose, set-up)") + g("kernel epilogue (point, culling, store)"), W, "per working wave",
        gq("kernel prologue (match, pose, set-up)", "kernel epilogue (point, culling, store)"))
    add("projection", g("projection (cam2World, T, world2Cam, 4 quotients)"), E, "per evaluation (wave)", gq("projection (cam2World, T, world2Cam, 4 quotients)"))
    add("interpolation", g("interpolation (geometry, loads, bilinear)"), E, "per evaluation (wave)", gq("interpolation (geometry, loads, bilinear)"))
    add("residuals + range tests + shortcut test", g("residuals, moments, range tests") + g("t-scale: shortcut test + set-up"), E, "per evaluation (wave)",
        gq("residuals, moments, range tests", "t-scale: shortcut test + set-up"))
    add("t-scale loop", tight_body, Lw, "per iteration (wave); body = the tight loop's", tight_q)
    add("weights", g("weights sqrt((nu+1)/(nu+r^2/s2)) r (tight)"), E, "per evaluation (wave); the tight variant", gq("weights sqrt((nu+1)/(nu+r^2/s2)) r (tight)"))
    add("solver: phase 1 (J, qtf) + lmpar", g("solver: phase 1 (forward difference, J, qtf)") + g("solver: lmpar + trial point"), jac, "per Jacobian pass (wave)",
        gq("solver: phase 1 (forward difference, J, qtf)", "solver: lmpar + trial point"))
    add("solver: phase 2 + outer loop", g("solver: phase 2 (trust-region test)") + g("solver: outer loop (DepthProblemSolver.cpp:161-188)") +
        g("solver: evaluator call site / pair exchange"), max(E - jac - W, 0), "per trial point (wave)",
        gq("solver: phase 2 (trust-region test)", "solver: outer loop (DepthProblemSolver.cpp:161-188)", "solver: evaluator call site / pair exchange"))
    add("solver: phase 0", g("solver: phase 0 (minimizeInit)"), W, "per working wave", gq("solver: phase 0 (minimizeInit)"))
    model = sum(r[3] for r in rows)
    cyc = sum(r[5] for r in rows)
    lock_e = (E - Eg) / E if E else 0
    lock_l = (Lw - Lg) / Lw if Lw else 0
    print(f"narrow lm_refine_kernel, headline workload: {D['matches_per_tick']:.0f} matches per launch = {W:.0f} working waves; "
          f"{D['evals_per_match']:.1f} evaluations per match, {D['tscale_iters_per_eval_group']:.2f} t-scale iterations per evaluation "
          f"({D['tscale_iters_per_eval_wave']:.2f} executed per wave-evaluation: four matches in lockstep)")
    print(f"{'region':44s} {'VALU/unit':>9s} {'units/launch':>13s} {'VALU/launch':>12s} {'share':>6s} {'issue cycles':>13s} {'share':>6s}  unit")
    for name, per, mult, tot, what, cy in rows:
        print(f"{name:44s} {per:9.0f} {mult:13.0f} {tot:12.3e} {100 * tot / model:5.1f}% {cy:13.3e} {100 * cy / cyc:5.1f}%  {what}")
    print(f"{'model total':44s} {'':9s} {'':13s} {model:12.3e} {'':6s} {cyc:13.3e}")
    print(f"issue cycles: every instruction at {C_PLAIN} cycles of its SIMD, the quarter-rate ones (v_rcp_f64, v_rsq_f64, v_sqrt_f64) at {C_QUARTER} -- the costs "
          f"measured at two waves per SIMD (tools/valu_rates.hip).  On 1024 SIMDs at 2.37 GHz the model's total is {cyc / 1024 / 2.37e9 * 1e3:.3f} ms of issue time per launch")
    if measured:
        print(f"{'SQ_INSTS_VALU (hardware counter, per launch)':44s} {'':9s} {'':13s} {float(measured):12.3e}   model / measured = {model / float(measured):.2f}")
    print(f"lockstep: {100 * lock_e:.0f} % of the executed evaluations and {100 * lock_l:.0f} % of the executed t-scale iterations are re-runs for "
          f"groups that did not need them (a wave runs the longest of its four matches)")
    if sq_waves:
        empty = float(sq_waves) - W
        print(f"empty waves: {empty:.0f} of {float(sq_waves):.0f} launched leave after the match-count test (~12 VALU each: {12 * empty:.2e}, "
              f"{100 * 12 * empty / model:.2f} % of the model total)")


if __name__ == "__main__":
    cmd = sys.argv[1] if len(sys.argv) > 1 else "static"
    tag = os.environ.get("ESVO_TAG", "r05")
    if cmd == "static":
        static(tag)
    elif cmd == "dynamic":
        dynamic(tag)
    else:
        table(*sys.argv[2:])
