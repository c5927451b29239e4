"""The N>1 path on CPU: world_size-2 gloo run of the utterance sharding and the transcript gather (no GPU)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from stt_amd import dist as sdist

WORKER = r'''
import os, sys, json
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from stt_amd import dist as sdist
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%%s" %% os.environ["PORT"], rank=int(os.environ["RANK"]), world_size=2)
rank = dist.get_rank()
lengths = [16000 * (1 + (7 * i) %% 15) for i in range(23)]
shards = sdist.shard_utterances(lengths, 2)
mine = ["utt%%d-len%%d-é" %% (i, lengths[i]) if i %% 5 else "" for i in shards[rank]]
allr = sdist.gather_transcripts(mine, device=torch.device("cpu"))
# equal batches declared (bench.py's weak scaling): single collective; one long transcript forces the second round
sdist.assume_equal_batches()
eq = ["r%%d-%%d" %% (rank, i) for i in range(7)]
eq[3] = "x" * (5000 if rank == 1 else 3)
alleq = sdist.gather_transcripts(eq, device=torch.device("cpu"), bytes_per_utterance=16)
if rank == 0:
    print(json.dumps({"shards": shards, "all": allr, "eq": alleq}))
dist.destroy_process_group()
'''


def test_lpt_sharding_balances_and_covers():
    rng = np.random.default_rng(2)
    lengths = rng.integers(16000, 240000, 1000)
    for w in (1, 2, 4, 8):
        sh = sdist.shard_utterances(lengths, w)
        assert sorted(i for s in sh for i in s) == list(range(1000))
        loads = [int(lengths[s].sum()) for s in sh]
        assert max(loads) - min(loads) <= lengths.max()
        for s in sh:
            assert all(lengths[s[k]] >= lengths[s[k + 1]] for k in range(len(s) - 1))


def test_gather_transcripts_two_ranks_gloo(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "w.py"
    script.write_text(WORKER % ROOT)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=240) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    import json
    res = json.loads(outs[0][0].strip().splitlines()[-1])
    lengths = [16000 * (1 + (7 * i) % 15) for i in range(23)]
    for rank in range(2):
        want = ["utt%d-len%d-é" % (i, lengths[i]) if i % 5 else "" for i in res["shards"][rank]]
        assert res["all"][rank] == want
        weq = ["r%d-%d" % (rank, i) for i in range(7)]
        weq[3] = "x" * (5000 if rank == 1 else 3)
        assert res["eq"][rank] == weq


def test_gather_without_process_group_is_identity():
    assert sdist.gather_transcripts(["a", "b"]) == [["a", "b"]]


def test_native_sharding_rule_equals_the_python_twin():
    """include/stt_amd.h: STTX_ShardUtterances (the dealing rule of STTX_FleetSpeechToTextBatch, stt_amd/csrc/fleet.cpp) against
    stt_amd.dist.shard_utterances on random length sets, ties included.  Host only."""
    import numpy as np

    from stt_amd import dist as sd
    from stt_amd import model as M
    rng = np.random.RandomState(3)
    for case in range(40):
        n = int(rng.randint(0, 200))
        lens = rng.randint(1, 20, size=n) * 8000 if case % 3 == 0 else rng.randint(0, 240000, size=n)
        for shards in (1, 2, 3, 8):
            want = sd.shard_utterances(lens, shards)
            got = M.shard_utterances_native(lens, shards)
            assert len(got) == n
            for r, idx in enumerate(want):
                assert sorted(i for i in range(n) if got[i] == r) == sorted(idx), (case, shards, r)


@pytest.mark.parametrize("shards", [1, 2, 4, 8])
def test_fleet_record_pack_and_unpack(shards):
    """stt_amd/csrc/fleet.cpp: the transcript records exactly as the two all-gathers move them (per-rank [index, length, bytes]
    records, padded to the longest rank, concatenated), packed and unpacked on the host -- empty strings, empty shards, UTF-8."""
    import ctypes as C

    from stt_amd import model as M
    from stt_amd import native
    if not os.path.exists(native.LIB_PATH):
        pytest.skip("libstt.so not built")
    L = native.lib()
    rng = np.random.RandomState(shards)
    texts = ["", "a", "she had your dark suit", "naïve café 北京", " ", "x" * 3000] + ["w%d " % i * int(rng.randint(0, 9)) for i in range(40)]
    for layout in ("lpt", "one_rank", "round_robin"):
        n = len(texts)
        if layout == "lpt":
            shard_of = M.shard_utterances_native([len(t) + 1 for t in texts], shards)
        elif layout == "one_rank":
            shard_of = [shards - 1] * n               # every other shard sends an empty record
        else:
            shard_of = [i % shards for i in range(n)]
        arr = (C.c_char_p * n)(*[t.encode("utf-8") for t in texts])
        so = (C.c_uint * n)(*shard_of)
        r = L.STTX_TestFleetRecords(arr, so, n, shards)
        assert r, layout
        got = [C.string_at(r[i]).decode("utf-8") for i in range(n)]
        L.STTX_FreeStrings(r, n)
        assert got == texts, layout
    assert not L.STTX_TestFleetRecords((C.c_char_p * 1)(b"x"), (C.c_uint * 1)(shards), 1, shards)   # a shard index out of range
