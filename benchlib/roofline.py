"""Roofline accounting of a bench line: SURVEY 8(d)'s algorithmic bytes, the committed rocprofv3 counters, VALU issue rates."""
import csv
import glob
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from esvo_amd import calib, lib, params, rostime, synth  # noqa: E402,F401
from .workload import HBM_PEAK_GBS, KERNEL_NAMES, KERNEL_SYMBOLS, VALU_PEAK_INST_S  # noqa: E402


def algorithmic_bytes(kernel, st, W, H, nd, fusion_radius=1, events=None, matches=None):
    """SURVEY.md §8(d) per-unit byte model x the units one launch processed (averages over the timed ticks)."""
    events = st.last_events_in if events is None else events
    matches = st.last_matches if matches is None else matches
    if kernel == "bm_match":  # 16 B event + 8 B LUT + 1 B mask + 105 B left + 7*(15+Nd-1) B strip + 48 B out
        return events * (16 + 8 + 1 + 105 + 7 * (15 + nd - 1) + 48)
    if kernel == "lm_refine":  # 48 B match + 2 x (16x8) B TS blocks + 64 B point
        return matches * (48 + 2 * 128 + 64)
    if kernel == "fuse":  # 64 B read + K cells x (52 B read + 52 B write) per window point
        k = 9 if fusion_radius else 4
        return st.last_window_points * (64 + k * 104)
    if kernel == "regularize":  # 52 B per valid pixel (+ cached taps)
        return st.last_map_size * 52
    if kernel == "ts_render":
        return W * H * 9
    return 0


TRAFFIC_NOTE = ("traffic = 2 x FETCH_SIZE + WRITE_SIZE of separate rocprofv3 --pmc passes (committed CSV named in traffic_source). The x2 on "
                "FETCH_SIZE is MI355X_MICROARCH.md's gfx950 correction, calibrated for streaming 16-byte-per-lane loads only; for byte "
                "gathers and atomics (ts_scatter, scatter_records, reg_apply's LDS staging) it is an uncalibrated upper bound")


def roofline_rows(st, kavg, rig, nd, p, prof, events, matches, scattered):
    """per-kernel roofline entries: Time-Surface scatter and render (the HBM-bound stage), block matching, LM"""
    rows = []
    ts_bytes = {"ts_scatter": scattered * 24.0,                    # 16 B event read + 8 B SAE atomic
                "ts_render": 2.0 * rig.width * rig.height * 9.0}   # both cameras: 8 B SAE read + 1 B mono8 write per pixel
    for slot in (0, 1, 2, 3):
        name = KERNEL_NAMES[slot]
        if slot < 2:
            nbytes = ts_bytes[name]
        else:
            nbytes = algorithmic_bytes(name, st, rig.width, rig.height, nd, p.fusion_radius, events=events, matches=matches)
        gbs = (nbytes / (kavg[slot] * 1e-3)) / 1e9 if kavg[slot] > 0 else 0.0
        tr, vl = profile_figures(prof, name, float(kavg[slot]))
        rows.append({"kernel": name, "avg_launch_ms": float(kavg[slot]), "algorithmic_bytes_per_launch": nbytes,
                     "achieved": gbs, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": tr,
                     "valu_frac": None if vl is None else vl["frac"]})
    return rows


def committed_profile(workload):
    """The newest committed rocprofv3 round (profiles/<tag>_meta.json names the command and workload it ran): per-kernel
    HBM bytes (separate FETCH_SIZE / WRITE_SIZE passes; FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 correction) and
    SQ counters per launch.  Counter values are properties of (build, workload): they are only attached to a bench line of
    the same workload, and the line says which files they came from."""
    metas = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_meta.json")))
    for path in reversed(metas):
        try:
            meta = json.load(open(path))
        except Exception:
            continue
        if meta.get("workload") != workload:
            continue
        tag = os.path.basename(path)[:-len("_meta.json")]
        out = {"tag": tag, "hbm": {}, "sq": {}, "files": []}
        for kind, fn in (("hbm", f"{tag}_hbm_traffic.csv"), ("sq", f"{tag}_sq_counters.csv")):
            fp = os.path.join(ROOT, "profiles", fn)
            if not os.path.exists(fp):
                continue
            out["files"].append("profiles/" + fn)
            with open(fp) as f:
                for row in csv.DictReader(f):
                    k = row["kernel"]
                    if kind == "hbm":
                        out["hbm"].setdefault(k, {})[row["counter"]] = float(row["avg_value_per_dispatch_KB"]) * 1024.0
                    else:
                        out["sq"].setdefault(k, {})[row["counter"]] = float(row["avg_value_per_dispatch"])
                        out.setdefault("sq_dispatches", {})[k] = int(row.get("dispatches") or 0)
        return out
    return None


def profile_figures(prof, kernel, launch_ms):
    """(traffic bytes per launch, VALU block) of `kernel` from a committed profile, or (None, None)"""
    if prof is None:
        return None, None
    sym = KERNEL_SYMBOLS.get(kernel)
    traffic = valu = None
    for k, v in prof["hbm"].items():
        if sym and sym in k and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            traffic = 2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]
    for k, v in prof["sq"].items():
        if sym and sym in k and "SQ_INSTS_VALU" in v:
            insts = v["SQ_INSTS_VALU"]
            rate = insts / (launch_ms * 1e-3) if launch_ms > 0 else 0.0
            valu = {"wave_insts_per_launch": insts, "achieved": rate, "peak": VALU_PEAK_INST_S, "unit": "wave64 VALU instructions/s",
                    "frac": rate / VALU_PEAK_INST_S,
                    "note": "instruction count per launch from the committed SQ_INSTS_VALU pass of this workload (deterministic for "
                            "a build), divided by this run's HIP-event launch time; peak = 1024 SIMDs x 2.4 GHz / 4 cycles per "
                            "f64 instruction"}
            if "SQ_ACTIVE_INST_VALU" in v and "SQ_BUSY_CYCLES" in v:
                valu["busy_frac_profiled"] = (v["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0) / (v["SQ_BUSY_CYCLES"] / 32.0)
    return traffic, valu


def whole_tick_valu(prof, tick_ms):
    """VALU wave-instructions of EVERY kernel of a tick (committed SQ_INSTS_VALU pass: per-launch average x launches per tick,
    the LM kernel's dispatch count = the ticks of that pass) against what the chip can issue in this run's tick time.  The
    three stages overlap on three queues, so this -- not one kernel's rate -- is the figure that bounds the tick."""
    if prof is None or not prof.get("sq_dispatches"):
        return None
    ticks = max([n for k, n in prof["sq_dispatches"].items() if "lm_refine_kernel" in k] or [0])
    if ticks <= 0 or tick_ms <= 0:
        return None
    total = sum(v["SQ_INSTS_VALU"] * prof["sq_dispatches"].get(k, 0) for k, v in prof["sq"].items() if "SQ_INSTS_VALU" in v) / ticks
    rate = total / (tick_ms * 1e-3)
    return {"wave_insts_per_tick": total, "achieved": rate, "peak": VALU_PEAK_INST_S, "frac": rate / VALU_PEAK_INST_S,
            "note": "all kernels of a tick; peak as above (nominal 2.4 GHz); `frac_at_measured_clock` re-prices it at the shader clock "
                    "measured inside this run (DESIGN.md section 5: 2.3 GHz -- round 3's 1.75 GHz estimate was wrong)"}


def attach_measured_clock(v, sclk_mhz):
    """the same instruction rate priced against the issue peak at the clock measured inside this run"""
    peak = 256 * 4 * sclk_mhz * 1e6 / 4.0
    v["sclk_mhz_measured"] = sclk_mhz
    v["peak_at_measured_clock"] = peak
    v["frac_at_measured_clock"] = v["achieved"] / peak


