"""Randomised Time-Surface ingest against the oracle's literal restatement of the reference's TimeSurface class (eventsCallback's
insertion sort and per-pixel queues, createTimeSurfaceAtTime; oracle/esvo_oracle.cpp, pinned to the class compiled from source by
tests/test_ref_pin.py): a seeded stream is delivered OUT OF ORDER (bundles of 0.1-1 ms arriving up to 0.6 ms off their time), cut
into packets of random size, each handed over through a random ingest call -- esvo_ts_push_events, the ROS wire format
(esvo_ts_push_event_array) or the asynchronous pinned path -- with renders at random (non-decreasing) times in between, on a random
rig (identity remap or a dataset's rectification), decay, median, queue length (0 = one stamp per pixel, compared with an unbounded
queue; 3; 20) and raster mode (backward / FORWARD).  Every rendered image must equal the oracle's byte for byte.
usage: python tools/fuzz_ts.py [cases] [first seed]      (GPU; exits 1 on any difference)"""
import os
import sys
import time

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np  # noqa: E402

from esvo_amd import calib, lib, params, synth  # noqa: E402
from esvo_amd.abi import EVENT_DTYPE, event_ns, serialize_event_array  # noqa: E402
from oracle import oracle  # noqa: E402


def jitter_arrival(ns, rng, bundle_ns, jitter_ns):
    """arrival order of a stream whose bundles are each delivered up to +-jitter off their time (stable inside a bundle)"""
    ns = ns.astype(np.int64)
    bundle = (ns - ns[0]) // bundle_ns
    off = rng.integers(-jitter_ns, jitter_ns + 1, int(bundle.max()) + 1)
    key = ns - (ns - ns[0]) % bundle_ns + off[bundle]
    return np.argsort(key, kind="stable")


def run_case(seed):
    rng = np.random.default_rng(seed)
    pick = lambda xs: xs[int(rng.integers(len(xs)))]  # noqa: E731
    rig_name = pick(["ideal", "upenn", "rpg", "hkust", "dsec"])
    rig = calib.ideal_rig(pick([64, 240, 346]), pick([48, 180, 260]), 200.0, 0.1) if rig_name == "ideal" else calib.dataset_rig(rig_name)
    ql = pick([0, 0, 20, 3])
    decay, median = pick([10.0, 30.0, 100.0]), pick([1, 1, 3])
    forward = bool(rng.integers(4) == 0)
    bundle_ns, jitter_ns = pick([100_000, 250_000, 1_000_000]), pick([0, 100_000, 200_000, 600_000])
    preset = {"dsec": "mapping_dsec", "rpg": "mapping_rpg", "hkust": "mapping_hkust"}.get(rig_name, "mvstereo_upenn")
    p, _ = params.make_params(params.PRESETS[preset], rig, decay_ms=decay, median_blur_kernel_size=median, max_event_queue_len=ql)
    st = synth.make_stream(rig, pick([1000, 4000, 12000]), float(rng.uniform(0.03, 0.12)), 0.2, 1.0, seed=int(rng.integers(1 << 30)), speed=1.0)
    ev = st.ev_left
    arr = np.ascontiguousarray(ev[jitter_arrival(event_ns(ev), rng, bundle_ns, jitter_ns)])
    ans = event_ns(arr).astype(np.int64)
    dev = lib.Esvo(p, rig)
    ots = oracle.OracleTS(rig.width, rig.height, queue_len=ql if ql else 1 << 20)
    bad, n_img = [], 0
    pos, t_last, seen_max = 0, 0, 0
    pinned = []
    while pos < len(arr):
        n = int(rng.integers(20, 4000))
        pk = arr[pos:pos + n]
        how = pick(["plain", "plain", "wire", "async"])
        if how == "wire":
            dev.ts_push_event_array(0, serialize_event_array(pk, rig.width, rig.height))
        elif how == "async":
            buf = np.ascontiguousarray(pk.copy(), dtype=EVENT_DTYPE)
            pinned.append(buf)
            dev.ts_push_events_async(0, buf)
            if rng.integers(2):
                dev.ts_push_wait(0)
        else:
            dev.ts_push_events(0, pk)
        ots.push(pk)
        seen_max = max(seen_max, int(ans[pos:pos + n].max()))
        pos += n
        if rng.integers(3) == 0 or pos >= len(arr):
            t = max(t_last, seen_max - int(pick([0, 0, 100_000, 1_000_000])) + 1)
            t_last = t
            if forward:
                g = dev.ts_render_forward(0, t)
                o = ots.render_forward(t, rig.left.rect_lut, decay_ms=decay, ignore_polarity=True, median_k=median)
            else:
                g = dev.ts_render(0, t)
                o = ots.render(t, decay_ms=decay, ignore_polarity=True, median_k=median, map_x=rig.left.map_x, map_y=rig.left.map_y)
            n_img += 1
            if not np.array_equal(g, o):
                bad.append((n_img, f"{int(np.count_nonzero(g != o))} pixels differ at render {n_img}"))
    dev.ts_push_wait(0)
    late = int(dev.stats().late_events[0])   # (events whose stamp precedes one already staged: withheld from the raster as the reference does)
    dev.close()
    brief = (f"{rig_name} {rig.width}x{rig.height} q{ql} decay {decay:g} median {median} {'FORWARD' if forward else 'backward'} bundles {bundle_ns // 1000} us "
             f"+-{jitter_ns // 1000} us: {len(arr)} events, {late} late, {n_img} renders")
    return bad, brief


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
    failed = 0
    t0 = time.time()
    for seed in range(s0, s0 + n):
        try:
            bad, brief = run_case(seed)
        except Exception as e:  # noqa: BLE001
            bad, brief = [("-", f"{type(e).__name__}: {e}")], ""
        failed += bool(bad)
        print(f"seed {seed}: {'EQUAL' if not bad else 'DIFFERENT ' + str(bad[:4])}  {brief}", flush=True)
    print(f"{n} cases, {failed} with a difference, {time.time() - t0:.0f} s")
    sys.exit(1 if failed else 0)


if __name__ == "__main__":
    main()
