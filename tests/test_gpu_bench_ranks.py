"""bench.py's N > 1 entry point on a one-GPU box: `python bench.py --gpus 2` must launch two ranks itself (gloo, the ranks share the
device: a plumbing check of the path the driver runs with RCCL on eight GPUs), report n_gpus 2 and twice the per-GPU batch, shard ONE
list in the ragged workload, and keep `verified` true against the real reference decoder.  (The promotion of benchmarks/r03_j.sh.)"""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _bench(*argv, env=None):
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None); e.pop("RANK", None); e.pop("LOCAL_RANK", None)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]          # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_gpus_2_means_two_ranks_and_a_global_batch_of_128():
    r = _bench("--gpus", "2", "--steps", "4", "--warmup", "2", "--no-extras", "--no-cpu-baseline", "--scorer", "fixture")
    assert r["n_gpus"] == 2 and r["config"]["global_batch"] == 128 and r["steps"] == 4
    assert r["verified"] is True and r["verified_against"] in ("reference", "reference+blocking", "blocking")
    import torch
    if torch.cuda.device_count() < 2:
        assert "gloo" in r["config"]["backend"] and r["config"]["rccl_ranks"] == 0      # flagged: not a measurement
    else:
        assert r["config"]["rccl_ranks"] == 2
    assert r["value"] > 0 and r["scaling"] == "weak"


def test_ragged_shards_one_list_over_the_ranks():
    r = _bench("--gpus", "2", "--workload", "ragged", "--utterances", "80", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--scorer", "fixture")
    assert r["n_gpus"] == 2 and r["config"]["global_batch"] == 160            # ONE list of 2 x 80 utterances
    assert r["verified"] is True
    assert "one LibriSpeech-shaped list of 160 utterances" in r["config"]["workload"]


def test_a_launcher_that_disagrees_with_gpus_is_refused():
    e = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "--gpus 2" in (r.stderr + r.stdout)
