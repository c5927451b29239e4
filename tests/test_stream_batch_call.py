"""stt_amd/model.py: StreamBatchCall -- the preallocated argument tables a server's hop loop fills by index.  Host only: a stand-in library
records what the C entry points would have been handed (addresses are dereferenced and compared with the slices they should name)."""
import ctypes as C

import numpy as np

from stt_amd import model as M
from stt_amd import native


class _FakeStream(object):
    def __init__(self, handle):
        self._impl = C.c_void_p(handle)

    def _check(self):
        assert self._impl


class _FakeLib(object):
    def __init__(self):
        self.fed, self.decoded, self.freed = [], [], 0
        self._keep = []

    def STTX_FeedAudioContentBatchEx(self, streams, audio, sizes, last, n):
        rows = []
        for i in range(n):
            a = np.ctypeslib.as_array(C.cast(audio[i], C.POINTER(C.c_short)), shape=(sizes[i],)).copy() if sizes[i] else np.zeros(0, np.int16)
            rows.append((streams[i], a, int(last[i])))
        self.fed.append(rows)

    def STTX_DecodeStreamsBatch(self, streams, finish, n):
        self.decoded.append([(streams[i], int(finish[i])) for i in range(n)])
        arr = (C.c_char_p * n)(*[("s%d" % streams[i]).encode() for i in range(n)])
        self._keep.append(arr)
        return arr

    def STTX_FreeStrings(self, r, n):
        self.freed += 1


def test_rows_name_the_right_audio_and_flags(monkeypatch):
    fake = _FakeLib()
    monkeypatch.setattr(native, "lib", lambda: fake)
    utts = [np.arange(100 * (u + 1), 100 * (u + 1) + 7000 + 3000 * u, dtype=np.int16) for u in range(3)]     # 7000, 10000, 13000 samples
    streams = [_FakeStream(1000 + u) for u in range(3)]
    call = M.StreamBatchCall(4)
    base = [a.ctypes.data for a in utts]
    k = 5120
    for i, a in enumerate(utts):                        # second hop: utterance 0 ends in it, the others do not
        left = len(a) - k
        call.set(i, streams[i], base[i] + 2 * k, min(5120, left), last=2 if left <= 5120 else 0)
    call.feed(3)
    rows = fake.fed[0]
    assert [r[0] for r in rows] == [1000, 1001, 1002]
    assert [r[2] for r in rows] == [2, 2, 0]
    for (_, got, _), a in zip(rows, utts):
        assert np.array_equal(got, a[k:k + 5120])
    # third hop: utterance 2 live, the two that ended ride along with empty buffers and are finished by the decode
    call.set(0, streams[2], base[2] + 2 * 2 * k, len(utts[2]) - 2 * k, last=2)
    call.set(1, streams[0], 0, 0, finish=1)
    call.set(2, streams[1], None, 0, finish=1)
    call.feed(3)
    rows = fake.fed[1]
    assert np.array_equal(rows[0][1], utts[2][2 * k:]) and rows[0][2] == 2
    assert rows[1][1].size == 0 and rows[2][1].size == 0 and rows[1][2] == 0 and rows[2][2] == 0      # (flags of an earlier hop do not linger)
    out = call.decode(3, [streams[0], streams[1]])
    assert out == ["s1002", "s1000", "s1001"] and fake.decoded[0] == [(1002, 0), (1000, 1), (1001, 1)] and fake.freed == 1
    assert streams[0]._impl is None and streams[1]._impl is None and streams[2]._impl
    assert call.decode(0) == []


def test_bench_stream_pass_feeds_every_utterance_once_and_keeps_the_cohort_size(monkeypatch):
    """bench.py: stream_pass (configs[2]'s hop loop) against the stand-in library: every utterance's audio goes in exactly once and in
    order, no hop has more rows than --streams (live + draining), every utterance comes back with a transcript."""
    import argparse
    import sys

    from conftest import ROOT
    sys.path.insert(0, ROOT)
    import bench

    class Lib(_FakeLib):
        def STTX_SetTuning(self, *a):
            return 0

    fake = Lib()
    monkeypatch.setattr(native, "lib", lambda: fake)
    monkeypatch.setattr(native, "set_tuning", lambda *a: None)

    class FakeModel(object):
        n = 0

        def createStream(self):
            FakeModel.n += 1
            return _FakeStream(FakeModel.n)

    class Cx(object):
        pass
    cx = Cx()
    cx.stream_models = [FakeModel(), FakeModel()]
    rng = np.random.RandomState(1)
    utts = [np.arange(int(rng.uniform(1, 4) * 16000), dtype=np.int16) + 7 * u for u in range(40)]
    utts[5] = utts[5][:5120 * 3].copy()                     # an exact multiple of the hop
    lat = []
    texts = bench.stream_pass(cx, argparse.Namespace(streams=8, cohorts=2), utts, lat)
    assert all(t is not None for t in texts)
    assert max(len(rows) for rows in fake.fed) <= 8 and len(lat) == len(fake.fed)
    by_stream = {}
    for rows in fake.fed:
        for h, a, last in rows:
            by_stream.setdefault(h, []).append((a, last))
    got = sorted((np.concatenate([a for a, _ in v]).tobytes() for v in by_stream.values()), key=len)
    want = sorted((u.tobytes() for u in utts), key=len)
    assert sorted(got) == sorted(want)
    for v in by_stream.values():                            # the last audio carries the flag, once; what follows is the empty ride-along row
        flags = [l for _, l in v]
        assert flags.count(2) == 1 and all(a.size == 0 for a, _ in v[flags.index(2) + 1:])
