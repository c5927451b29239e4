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
    """All ranks call this with their local list of str; every rank returns the list of all ranks' lists (rank order)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return [list(texts)]
    world = dist.get_world_size(group)
    enc = [t.encode("utf-8") for t in texts]
    lens = torch.tensor([len(e) for e in enc] or [0], dtype=torch.int32, device=device)
    meta = torch.tensor([len(enc), int(lens.max().item()) if len(enc) else 0], dtype=torch.int32, device=device)
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    n_max = max(int(m[0]) for m in metas)
    l_max = max(1, max(int(m[1]) for m in metas))
    lens_p = torch.zeros(max(1, n_max), dtype=torch.int32, device=device)
    lens_p[:len(enc)] = lens[:len(enc)]
    buf = torch.zeros((max(1, n_max), l_max), dtype=torch.uint8, device=device)
    for i, e in enumerate(enc):
        if e:
            buf[i, :len(e)] = torch.frombuffer(bytearray(e), dtype=torch.uint8).to(buf.device)
    all_lens = [torch.zeros_like(lens_p) for _ in range(world)]
    all_buf = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(all_lens, lens_p, group=group)
    dist.all_gather(all_buf, buf, group=group)
    out = []
    for r in range(world):
        n = int(metas[r][0])
        ls = all_lens[r].cpu().numpy()
        b = all_buf[r].cpu().numpy()
        out.append([bytes(b[i, :ls[i]]).decode("utf-8", "replace") for i in range(n)])
    return out
