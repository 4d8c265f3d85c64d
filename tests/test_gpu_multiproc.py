"""The real multi-process drivers (dist.TickShardedEsvo / dist.ShardedEsvo under torch.distributed) on ONE GPU: two ranks
share the device and exchange through gloo, so the collectives, the device-pointer views and the round / phase logic run
exactly as they do over RCCL; only the transport differs.  The final DepthMap must have the single-process SHA-1."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(extra_env, launcher, n, strong=True, check=True, extras=False):
    env = dict(os.environ, **extra_env)
    cmd = launcher + [os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--workload", "upenn346x260", "--steps", "5", "--warmup", "2",
                      "--no-cpu-baseline"] + ([] if extras else ["--no-extras"]) + (["--check"] if check else []) + (["--strong"] if strong else [])
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


def test_two_ranks_give_the_single_process_map():
    single = _bench({}, [sys.executable], 1)
    assert single["check"]["final"]["map_size"] > 100
    assert single["check"]["oracle"]["equal"], single["check"]["oracle"]   # bench.py --check: the first timed tick equals the oracle's
    torchrun = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", "29533"]
    for mode in ("tick", "band"):
        two = _bench({"ESVO_SHARED_GPU": "1", "ESVO_DIST_BACKEND": "gloo", "ESVO_SHARD_MODE": mode}, torchrun, 2)
        assert two["n_gpus"] == 2 and two["check"]["final"] == single["check"]["final"], (mode, two["check"], single["check"])
        assert two["config"]["events_per_tick"] == single["config"]["events_per_tick"]
    # the default N-GPU mode is weak scaling: 5 timed ticks per rank
    weak = _bench({"ESVO_SHARED_GPU": "1", "ESVO_DIST_BACKEND": "gloo"}, torchrun, 2, strong=False)
    assert weak["scaling"] == "weak" and weak["config"]["ticks_timed"] == 10 and weak["steps"] == 5
    assert abs(weak["config"]["events_per_tick"] - single["config"]["events_per_tick"]) < 0.05 * single["config"]["events_per_tick"]


def _device_count():
    import torch
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it spawns the two ranks itself (here: both on the one GPU, gloo);
    without enough devices -- and without the shared-GPU test switch -- it refuses instead of reporting a one-rank number."""
    two = _bench({"ESVO_SHARED_GPU": "1", "ESVO_DIST_BACKEND": "gloo"}, [sys.executable], 2, strong=False)
    assert two["n_gpus"] == 2 and two["scaling"] == "weak" and two["config"]["ticks_timed"] == 10
    assert two["launcher"].startswith("self") and len(two["ranks_seen"]) == 2 and {r["rank"] for r in two["ranks_seen"]} == {0, 1}
    # the line of a multi-GPU run carries BOTH modes: tick-interleaved (weak, `value`) and one tick split by slot / row band
    both = _bench({"ESVO_SHARED_GPU": "1", "ESVO_DIST_BACKEND": "gloo"}, [sys.executable], 2, strong=False, check=False, extras=True)
    assert both["band_mode"]["scaling"] == "strong" and both["band_mode"]["value"] > 0 and both["band_mode"]["map_size_after_gather"] > 100
    if _device_count() < 2:
        env = {k: v for k, v in os.environ.items() if k not in ("ESVO_SHARED_GPU", "WORLD_SIZE", "RANK", "LOCAL_RANK")}
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"], env=env, cwd=ROOT,
                             capture_output=True, text=True, timeout=300)
        assert out.returncode != 0 and "GPU(s) visible" in out.stderr and not [l for l in out.stdout.splitlines() if l.startswith("{")]


def _selftest(extra_env, n):
    env = dict(os.environ, **extra_env)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--selftest"], env=env, cwd=ROOT,
                         capture_output=True, text=True, timeout=300)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    return out.returncode, (json.loads(line[-1]) if line else None), out.stderr[-1500:]


def test_selftest_reports_ranks_exchange_and_map_equality():
    """`python bench.py --gpus N --selftest`: the plumbing check a scaling run starts with.  Here two ranks on the one GPU over
    gloo (the torch.distributed drivers; on a multi-GPU node the same command runs esvo_comm_* over RCCL), and one rank alone."""
    rc, one, err = _selftest({}, 1)
    assert rc == 0 and one["ok"] and one["n_gpus"] == 1 and one["depth_map"]["one_gpu"]["map_size"] > 100, (rc, one, err)
    rc, two, err = _selftest({"ESVO_SHARED_GPU": "1", "ESVO_DIST_BACKEND": "gloo"}, 2)
    assert rc == 0 and two["ok"], (rc, two, err)
    assert {r["rank"] for r in two["ranks_seen"]} == {0, 1} and two["all_gather"]["verified"] and two["all_gather"]["us_min"] > 0
    assert two["depth_map_equal_to_one_gpu"] and set(two["depth_map"]) == {"tick", "tick_torch", "band", "band_broadcast", "one_gpu"}
    # tick_torch: the C round logic (two rounds in flight, count-sized blocks) with torch.distributed as its transport
    ex = two["depth_map"]["tick_torch"]["exchange"]
    assert ex["rounds"] == 3 and ex["regrows"] == 0 and ex["bytes_sent"] > 0
    assert two["depth_map"]["one_gpu"]["sha1"] == one["depth_map"]["one_gpu"]["sha1"]
    assert two["seconds"] < 60


def test_selftest_at_world_8_on_a_shared_gpu():
    """BASELINE.json's headline world size, functionally: eight ranks (processes) on the ONE GPU over gloo through
    `python bench.py --gpus 8 --selftest` -- tick-interleaved, row-routed bands (260 rows / 8: ragged 33- and 29-row bands) and
    the broadcast switch all reproduce the one-GPU DepthMap; a routed rank stages a fraction of the stream.  The first real
    8-GPU run executes exactly this command (then over RCCL) before it measures anything: it must finish well inside 120 s."""
    rc, r8, err = _selftest({"ESVO_SHARED_GPU": "1", "ESVO_DIST_BACKEND": "gloo"}, 8)
    assert rc == 0 and r8 and r8["ok"], (rc, r8, err)
    assert {r["rank"] for r in r8["ranks_seen"]} == set(range(8)) and r8["all_gather"]["verified"]
    assert r8["depth_map_equal_to_one_gpu"] and set(r8["depth_map"]) == {"tick", "tick_torch", "band", "band_broadcast", "one_gpu"}
    band = r8["depth_map"]["band"]
    assert band["halo_violations"] == 0
    assert band["events_staged_rank0"][0] < 0.6 * band["events_in_stream"][0], band
    assert r8["seconds"] < 120, r8["seconds"]


@pytest.mark.skipif(_device_count() < 2, reason="needs two MI355X on the node (real RCCL between two processes)")
def test_two_gpus_over_real_rccl():
    """Self-enabling on a multi-GPU node: one process per GPU, esvo_comm_init at world 2 over real RCCL (ncclCommInitRank,
    ncclAllGather of the frames / of the slot bits and kept points inside libesvo_hip.so), no launcher.  Both ways of using the
    GPUs must reproduce the single-GPU DepthMap."""
    single = _bench({}, [sys.executable], 1)
    for mode in ("tick", "band"):
        two = _bench({"ESVO_SHARD_MODE": mode}, [sys.executable], 2)
        assert two["n_gpus"] == 2 and "esvo_comm_*" in two["config"]["parallelism"], two["config"]["parallelism"]
        assert "failed" not in two["config"]["parallelism"]                       # no fallback to the torch.distributed driver
        assert two["rccl"]["version_code"] > 0 and two["rccl"]["library"]
        assert len({r["device"] for r in two["ranks_seen"]}) == 2                 # really two devices
        assert two["check"]["final"] == single["check"]["final"], (mode, two["check"], single["check"])
    both = _bench({}, [sys.executable], 2, strong=False, check=False, extras=True)  # the default line: weak + the band-mode block
    assert both["scaling"] == "weak" and both["n_gpus"] == 2
    assert both["band_mode"]["scaling"] == "strong" and both["band_mode"]["value"] > 0 and both["band_mode"]["map_size_after_gather"] > 100
