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
