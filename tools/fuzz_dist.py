"""Randomised multi-GPU parity on ONE GPU: the configurations of tools/fuzz_parity.py (random rig, patch, norm, window, regulariser,
denoising, Time-Surface options ...) run once on a single handle and once split over 2 / 3 / 4 / 8 ranks -- threads, one handle
each, the all-gather an in-process transport (esvo_comm_init_callbacks: the code a node with one process per GPU runs) -- either
tick-interleaved (esvo_comm_tick[_resident], two rounds in flight, a read-out in the middle of a round) or with every tick split by
image row band (esvo_comm_shard_tick, events routed by rectified row or broadcast, the band maps gathered every tick).  Every map
a rank sees must equal the single handle's bit for bit.  A routed band run may refuse a tick with ESVO_ERR_HALO (a refinement left
the rendered rows): reported as such, not a difference.
usage: python tools/fuzz_dist.py [cases] [first seed]      (GPU; exits 1 on any difference)"""
import os
import sys
import threading
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np  # noqa: E402
import torch  # noqa: E402

torch.cuda.init()   # before the first handle and in the main thread: the in-process transport copies with torch from callback threads

import fuzz_parity  # noqa: E402
from benchlib.workload import map_sha1  # noqa: E402
from esvo_amd import calib, lib, params, rostime, synth  # noqa: E402
from esvo_amd import dist as edist  # noqa: E402


def run_ranks(world, body):
    errs, outs = [], [None] * world

    def main(r):
        try:
            outs[r] = body(r)
        except BaseException as e:  # noqa: BLE001
            errs.append((r, e))

    th = [threading.Thread(target=main, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    return outs, errs


def run_case(seed):
    rig_name, cfg, over, sc = fuzz_parity.draw(seed)
    rng = np.random.default_rng(seed + 77)
    pick = lambda xs: xs[int(rng.integers(len(xs)))]  # noqa: E731
    mode = pick(["tick", "band", "band"])
    world = pick([2, 3, 4, 8])
    rig = calib.dataset_rig(rig_name)
    p, _ = params.make_params(cfg, rig, **over)
    n_ticks = sc["ticks"] + (world if mode == "tick" else 0)      # tick mode: at least one full round + a tail
    dur = 0.06 + (n_ticks + 1) * sc["tick_ms"] * 1e-3
    st = synth.make_stream(rig, sc["points"], dur, sc["rho"][0], sc["rho"][1], seed=sc["seed"], speed=sc["speed"])
    ticks = []
    for k in range(n_ticks):
        t = st.t0_ns + int((0.06 + (k + 1) * sc["tick_ms"] * 1e-3) * 1e9)
        stamps, poses = rostime.pose_table(st.pose, t, p.bm_half_slice_thickness)
        ticks.append((t, stamps, poses, st.pose(t)))
    # the single handle
    dev = lib.Esvo(p, rig)
    dev.ts_push_events(0, st.ev_left)
    dev.ts_push_events(1, st.ev_right)
    ref = []
    for t, stamps, poses, T in ticks:
        dev.tick_resident(t, T, stamps, poses)
        mp = dev.get_map()
        ref.append((len(mp), map_sha1(mp)))
    dev.close()
    if max(n for n, _ in ref) == 0:
        return None, "nothing mapped on one GPU: skipped"
    tr = edist.LocalTransport(world)
    bad = []
    what = f"{mode} world {world}"
    if mode == "tick":
        resident = bool(rng.integers(2))
        k_mid = int(rng.integers(n_ticks - 1))
        what += f" {'resident' if resident else 'four calls'} read-out after tick {k_mid}"

        def body(r):
            d = lib.Esvo(p, rig)
            d.comm_init_callbacks(r, world, lambda s, dd, n, stt: tr.all_gather(r, s, dd, n))
            d.ts_push_events(0, st.ev_left)
            d.ts_push_events(1, st.ev_right)
            got = []
            for k, (t, stamps, poses, T) in enumerate(ticks):
                if resident:
                    d.comm_tick_resident(t, T, stamps, poses)
                else:
                    if d.comm_owns_next_tick():
                        d.ts_render(0, t, download=False)
                        d.ts_render(1, t, download=False)
                    d.comm_tick(t, T, stamps, poses)
                if k == k_mid or k == n_ticks - 1:
                    mp, idx = d.comm_newest_map()
                    got.append((k, idx, len(mp), map_sha1(mp)))
            d.close()
            return got
        outs, errs = run_ranks(world, body)
        if errs:
            bad.append(("error", str(errs[0][1])[:200]))
        else:
            for r, got in enumerate(outs):
                for k, idx, n, sha in got:
                    if idx != k or (n, sha) != ref[k]:
                        bad.append((r, f"newest map after tick {k}: index {idx}, {n} vs {ref[k][0]} elements"))
    else:
        routing = pick(["y_rect", "y_rect", "broadcast"])
        if p.max_event_queue_len or p.bm_updown:
            routing = "broadcast"   # (what esvo_shard_set_routing refuses to route)
        what += f" {routing}"

        def body(r):
            d = lib.Esvo(p, rig)
            y0, y1 = edist.band_of(r, world, rig.height)
            d.set_band(y0, y1, r, world, routing=routing)
            d.comm_init_callbacks(r, world, lambda s, dd, n, stt: tr.all_gather(r, s, dd, n))
            d.ts_push_events(0, st.ev_left)
            d.ts_push_events(1, st.ev_right)
            got = []
            try:
                for t, stamps, poses, T in ticks:
                    d.ts_render(0, t, download=False)
                    d.ts_render(1, t, download=False)
                    d.set_observation(t, None, None, T)
                    d.comm_shard_tick(t, stamps, poses)
                    mp = d.comm_gather_map()
                    got.append((len(mp), map_sha1(mp)))
            except lib.EsvoError as e:
                if getattr(e, "code", None) != lib.ERR_HALO:
                    raise
                got.append("halo")
            d.close()
            return got
        outs, errs = run_ranks(world, body)
        if errs:
            bad.append(("error", str(errs[0][1])[:200]))
        else:
            halo = [g for g in outs if g and g[-1] == "halo"]
            if halo:
                if len(halo) != world or len({len(g) for g in outs}) != 1:
                    bad.append(("halo", "not every rank refused the same tick"))
                what += f" -- ESVO_ERR_HALO at tick {len(outs[0]) - 1}"
            for r, got in enumerate(outs):
                for k, g in enumerate(got):
                    if g != "halo" and g != ref[k]:
                        bad.append((r, f"tick {k}: {g[0]} vs {ref[k][0]} elements"))
    brief = (f"{rig_name} patch {cfg['patch_size_X']}x{cfg['patch_size_Y']} {cfg['LSnorm']} ev{cfg['PROCESS_EVENT_NUM']} fr{cfg['fusion_radius']} "
             f"{cfg['FUSION_STRATEGY'][6:]} reg{int(cfg['Regularization'])}/{cfg['RegularizationRadius']} dn{int(cfg['Denoising'])} sm{int(cfg['SmoothTimeSurface'])} "
             f"q{p.max_event_queue_len} ud{int(cfg['BM_bUpDownConfiguration'])} {cfg['node']} | {what} | maps {[n for n, _ in ref]}")
    return bad, brief


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 9000
    failed = ran = 0
    t0 = time.time()
    for seed in range(s0, s0 + n):
        try:
            bad, brief = run_case(seed)
        except Exception as e:  # noqa: BLE001
            bad, brief = [("-", f"{type(e).__name__}: {e}")], str(fuzz_parity.draw(seed)[:3])
        if bad is None:
            continue
        ran += 1
        failed += bool(bad)
        print(f"seed {seed}: {'EQUAL' if not bad else 'DIFFERENT ' + str(bad[:4])}  {brief}", flush=True)
    print(f"{n} seeds, {ran} cases that map something, {failed} with a difference, {time.time() - t0:.0f} s")
    sys.exit(1 if failed else 0)


if __name__ == "__main__":
    main()
