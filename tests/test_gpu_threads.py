"""The handle's threading contract (include/esvo_hip.h, "Threads"): an INGEST thread, a MAPPER thread and a TRACKER thread
work on ONE handle at the same time -- the reference's eventsCallback on the ROS spinner (esvo_Mapping.cpp:669-703,
TimeSurface.cpp:403-425), MappingAtTime on the MappingLoop worker (esvo_Mapping.cpp:179-247) and the tracker's loop -- and
every DepthMap equals the one of the serial run; the tracker's evaluations equal the CPU oracle's."""
import threading

import numpy as np
import pytest

from esvo_amd import params, rostime

pytestmark = pytest.mark.gpu
F64 = ["inv_depth", "scale2", "nu", "variance", "residual", "x", "p_cam"]


def _same(a, b):
    assert len(a) == len(b), (len(a), len(b))
    for f in ("row", "col", "age"):
        assert np.array_equal(a[f], b[f]), f
    for f in F64:
        assert np.array_equal(a[f], b[f]), f


def _tick_times(stream, n, t_first=0.05, dt=0.008):
    return [stream.t0_ns + int((t_first + k * dt) * 1e9) for k in range(n)]


def _serial(p, rig, stream, times):
    from esvo_amd import lib
    dev = lib.Esvo(p, rig)
    dev.ts_push_events(0, stream.ev_left)
    dev.ts_push_events(1, stream.ev_right)
    maps, surfaces = [], []
    for t in times:
        surfaces.append(dev.ts_render(0, t))
        dev.ts_render(1, t, download=False)
        stamps, poses = rostime.pose_table(stream.pose, t, p.bm_half_slice_thickness)
        dev.set_observation(t, None, None, stream.pose(t))
        dev.tick(t, stamps, poses)
        maps.append(dev.get_map())
    cloud = dev.get_pointcloud()
    dev.close()
    return maps, surfaces, cloud


@pytest.mark.parametrize("preset,rig_fix,stream_fix,over,block", [
    ("mapping_dsec", "dsec_rig", "dsec_stream", dict(process_event_num=4000), 1500),
    ("mapping_upenn", "upenn_rig", "upenn_stream", dict(event_ring_capacity=16384), 700),   # the ring wraps under the ticks
])
def test_pusher_ticker_and_tracker_threads(request, preset, rig_fix, stream_fix, over, block):
    from esvo_amd import lib
    from oracle import oracle as O
    rig, stream = request.getfixturevalue(rig_fix), request.getfixturevalue(stream_fix)
    p, _ = params.make_params(params.PRESETS[preset], rig, **over)
    times = _tick_times(stream, 8)
    big = dict(over)
    big.pop("event_ring_capacity", None)
    p_serial, _ = params.make_params(params.PRESETS[preset], rig, **big)
    want_maps, surfaces, cloud = _serial(p_serial, rig, stream, times)
    assert len(want_maps[-1]) > 100 and len(cloud) > 100

    dev = lib.Esvo(p, rig)
    errors = []
    staged = [0, 0]                                 # per camera: stamp of the newest staged event
    rendered = [stream.t0_ns]                       # stamp of the newest render (the pusher of a small ring must not run away)
    # the Time-Surface node renders at 100 Hz whether the mapper ticks or not: the renders before the first tick
    pre_times = list(range(stream.t0_ns + 8_000_000, times[0] - 1_000_000, 8_000_000))
    cv = threading.Condition()
    t_end = times[-1] + 2_000_000
    small_ring = "event_ring_capacity" in over

    def pusher(cam):
        try:
            ev = stream.ev_left if cam == 0 else stream.ev_right
            ns = stream.ns_left if cam == 0 else stream.ns_right
            hi_all = int(np.searchsorted(ns, t_end))
            at = 0
            while at < hi_all:
                nxt = min(at + block, hi_all)
                if small_ring:   # flow control: a slot is reused once its event is in the SAE, i.e. after the render behind it
                    with cv:
                        cv.wait_for(lambda: ns[nxt - 1] < rendered[0] + 15_000_000 or errors, timeout=60)
                while True:
                    try:
                        dev.ts_push_events(cam, ev[at:nxt])
                        break
                    except lib.EsvoError as e:   # "event ring full: render (scatter) before staging more": wait for a render
                        if "ring full" not in str(e):
                            raise
                        with cv:
                            cv.wait(timeout=0.002)
                at = nxt
                with cv:
                    staged[cam] = int(ns[at - 1])
                    cv.notify_all()
        except Exception as e:  # noqa: BLE001
            errors.append(("pusher", cam, repr(e)))
            with cv:
                cv.notify_all()

    got_maps = []

    def ticker():
        try:
            for t in pre_times + times:
                with cv:   # causality stays with the caller: the events before t, and one at or after it (Appendix A-3), are staged first
                    ok = cv.wait_for(lambda: (staged[0] >= t and staged[1] >= t) or errors, timeout=120)
                assert ok and not errors, errors
                dev.ts_render(0, t, download=False)
                dev.ts_render(1, t, download=False)
                with cv:
                    rendered[0] = t
                    cv.notify_all()
                if t < times[0]:
                    continue
                stamps, poses = rostime.pose_table(stream.pose, t, p.bm_half_slice_thickness)
                dev.set_observation(t, None, None, stream.pose(t))
                dev.tick(t, stamps, poses)
                got_maps.append(dev.get_map())
        except Exception as e:  # noqa: BLE001
            errors.append(("ticker", repr(e)))
            with cv:
                cv.notify_all()

    trk_checks = [0]
    stop = threading.Event()

    def tracker():
        try:
            trk = O.OracleTracker(rig)
            pts = cloud[:1500]
            T_ref = stream.pose(times[2])
            k = 0
            while not stop.is_set():
                img = surfaces[k % len(surfaces)]
                trk.set_current(img, 5)
                dev.track_set_current(img, 5)                 # host image: deterministic whatever the mapper renders meanwhile
                trk.set_reference(pts, T_ref)
                dev.track_set_reference(pts, T_ref)
                T = np.linalg.inv(T_ref) @ stream.pose(times[2] + 4_000_000)
                Tw = np.linalg.inv(T)
                g = dev.track_residuals(Tw, 0, 400, huber=True, huber_threshold=50.0)
                o = trk.residuals(Tw, 0, 400, huber=True, huber_threshold=50.0)
                assert np.array_equal(g, o)
                gj, oj = dev.track_jacobian(T[:3, :3].copy(), T[:3, 3].copy(), 0, 400), trk.jacobian(T[:3, :3].copy(), T[:3, 3].copy(), 0, 400)
                assert np.array_equal(gj, oj)
                try:   # the resident surface, racing the mapper's renders: whichever frame it catches, it is a whole one
                    dev.track_set_current(None, 5)
                    neg = dev.track_images()[0]
                    assert neg.shape == (rig.height, rig.width)
                except lib.EsvoError as e:
                    assert "esvo_ts_render" in str(e)         # nothing rendered yet
                trk_checks[0] += 1
                k += 1
        except Exception as e:  # noqa: BLE001
            errors.append(("tracker", repr(e)))

    threads = [threading.Thread(target=pusher, args=(0,)), threading.Thread(target=pusher, args=(1,)),
               threading.Thread(target=ticker), threading.Thread(target=tracker)]
    for th in threads:
        th.start()
    threads[2].join(timeout=300)
    stop.set()
    for th in threads:
        th.join(timeout=120)
    assert not errors, errors
    assert len(got_maps) == len(want_maps) and trk_checks[0] >= 1
    for a, b in zip(got_maps, want_maps):
        _same(a, b)
    s = dev.stats()
    assert s.ticks == len(times) and s.events_staged[0] > 0 and s.events_staged[1] > 0


def test_last_error_is_per_thread(upenn_rig, upenn_stream):
    """esvo_last_error returns the calling thread's message: a failure on the ingest thread does not rewrite what the
    mapper thread reads."""
    from esvo_amd import lib
    p, _ = params.make_params(params.PRESETS["mvstereo_upenn"], upenn_rig)
    dev = lib.Esvo(p, upenn_rig)
    seen = {}

    def bad_push():
        try:
            dev.ts_push_event_array(0, b"\x00" * 8)   # a message shorter than its header
        except lib.EsvoError as e:
            seen["push"] = str(e)

    with pytest.raises(lib.EsvoError, match="set_observation"):
        dev.match(upenn_stream.ev_left[:10], *rostime.pose_table(upenn_stream.pose, upenn_stream.t0_ns, 0.001))
    th = threading.Thread(target=bad_push)
    th.start()
    th.join()
    assert "shorter" in seen["push"]
    assert b"set_observation" in dev.lib.esvo_last_error(dev.h)   # this thread's own message is still there
