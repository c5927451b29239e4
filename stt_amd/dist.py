"""stt_amd/dist.py -- multi-GPU plumbing (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" on CPU tests).

The path shards by utterance (the reference does the same with one process per GPU,
training/coqui_stt_training/transcribe.py:40-56,136-148): every rank holds a full replica of the weights and the
scorer, there is no communication during compute, and ONE exchange at the end gathers the variable-length
transcripts: all_gather of int32 byte counts, then all_gather of a padded uint8 buffer.
"""
import numpy as np


def shard_utterances(lengths, world_size):
    """Longest-processing-time-first assignment of utterances to ranks.  Returns world_size lists of indices
    (each sorted by descending length, so a rank's batches are length-homogeneous)."""
    order = np.argsort(-np.asarray(lengths, dtype=np.int64), kind="stable")
    loads = [0] * world_size
    shards = [[] for _ in range(world_size)]
    for i in order:
        r = int(np.argmin(loads))
        shards[r].append(int(i))
        loads[r] += int(lengths[i])
    return shards


def gather_transcripts(texts, device=None, group=None, bytes_per_utterance=256):
    """All ranks call this with their local list of str; every rank returns the list of all ranks' lists (rank order).

    One collective in the common case: every rank contributes one fixed-capacity uint8 record
    ([int64 total bytes][int32 n][int32 byte length x n][utf-8 bytes ...], capacity = 64 + n_max_guess * bytes_per_utterance,
    identical on all ranks because they decode equally sized batches).  If some rank's record does not fit (its true size
    travels in the header, so every rank sees it), a second all_gather with the needed capacity follows.  The record is
    packed on the host and moved with a single copy each way: the exchange is latency-bound (a few KB)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return [list(texts)]
    world = dist.get_world_size(group)
    enc = [t.encode("utf-8") for t in texts]
    head = np.array([len(enc)] + [len(e) for e in enc], dtype=np.int32)
    body = np.concatenate([head.view(np.uint8), np.frombuffer(b"".join(enc), dtype=np.uint8)])
    rec = np.concatenate([np.array([body.size], dtype=np.int64).view(np.uint8), body])

    def exchange(cap):
        buf = np.zeros(cap, dtype=np.uint8)
        buf[:min(cap, rec.size)] = rec[:cap]
        mine = torch.from_numpy(buf).to(device) if device is not None else torch.from_numpy(buf)
        allb = torch.zeros(world * cap, dtype=torch.uint8, device=device)
        dist.all_gather_into_tensor(allb, mine, group=group)
        return allb.cpu().numpy().reshape(world, cap)

    # the capacity must be the same on every rank without talking: derive it from a quantity they share (the largest
    # batch any rank can hold is not known, so round the local count up generously; a mismatch is caught below)
    n_guess = 64 * ((len(enc) + 63) // 64 or 1)
    cap = 64 + n_guess * (8 + int(bytes_per_utterance))
    if not _same_batch_hint(group):
        # ranks may hold different counts (LPT shards): agree on the capacity with one small reduction
        t = torch.tensor([cap], dtype=torch.int64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        cap = int(t.item())
    flat = exchange(cap)
    need = max(8 + int(flat[r, :8].view(np.int64)[0]) for r in range(world))
    if need > cap:
        flat = exchange(need)
    out = []
    for r in range(world):
        row = flat[r, 8:]
        n = int(row[:4].view(np.int32)[0])
        lens = row[4:4 + 4 * n].view(np.int32)
        off = 4 + 4 * n
        items = []
        for ln in lens:
            items.append(bytes(row[off:off + int(ln)]).decode("utf-8", "replace"))
            off += int(ln)
        out.append(items)
    return out


_SAME_BATCH = set()


def assume_equal_batches(group=None):
    """Callers whose ranks always decode the same number of utterances (bench.py: weak scaling) may declare it once; the
    capacity agreement (one small all_reduce) is then skipped and the gather is a single collective."""
    _SAME_BATCH.add(id(group) if group is not None else None)


def _same_batch_hint(group):
    return (id(group) if group is not None else None) in _SAME_BATCH
