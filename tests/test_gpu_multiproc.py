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


def _bench(extra_env, launcher, n, strong=True):
    env = dict(os.environ, **extra_env)
    cmd = launcher + [os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--workload", "upenn346x260", "--steps", "5", "--warmup", "2",
                      "--no-cpu-baseline", "--no-extras", "--check"] + (["--strong"] if strong else [])
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
