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


def gather_transcripts(texts, device=None, group=None):
    """All ranks call this with their local list of str; every rank returns the list of all ranks' lists (rank order).

    Two collectives: all_gather of the packed size, then all_gather of one padded uint8 record per rank
    ([int32 n][int32 byte length x n][utf-8 bytes ...]).  The record is packed on the host and moved with a single copy
    each way: the exchange is latency-bound (a few KB), so the number of device round trips is what matters."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return [list(texts)]
    world = dist.get_world_size(group)
    enc = [t.encode("utf-8") for t in texts]
    head = np.array([len(enc)] + [len(e) for e in enc], dtype=np.int32)
    rec = np.concatenate([head.view(np.uint8), np.frombuffer(b"".join(enc), dtype=np.uint8)])
    size = torch.tensor([rec.size], dtype=torch.int64, device=device)
    sizes = torch.zeros(world, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(sizes, size, group=group)
    cap = int(sizes.max().item())
    buf = np.zeros(cap, dtype=np.uint8)
    buf[:rec.size] = rec
    mine = torch.from_numpy(buf).to(device) if device is not None else torch.from_numpy(buf)
    allb = torch.zeros(world * cap, dtype=torch.uint8, device=device)
    dist.all_gather_into_tensor(allb, mine, group=group)
    flat = allb.cpu().numpy().reshape(world, cap)
    out = []
    for r in range(world):
        row = flat[r]
        n = int(row[:4].view(np.int32)[0])
        lens = row[4:4 + 4 * n].view(np.int32)
        off = 4 + 4 * n
        items = []
        for ln in lens:
            items.append(bytes(row[off:off + int(ln)]).decode("utf-8", "replace"))
            off += int(ln)
        out.append(items)
    return out
