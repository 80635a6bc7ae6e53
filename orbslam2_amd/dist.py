"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" == RCCL over xGMI).

Frame pairs are independent units: ranks shard them round-robin and never exchange
pixels, keypoints or matches.  The only collectives on the path are ONE-TIME broadcasts from rank 0 -- the
extractor parameters, the rBRIEF pattern table itself (4 KiB as int32, `broadcast_pattern`; every rank hands
it to its context with `orbfe_set_pattern`, as ORBextractor copies the table per object,
src/ORBextractor.cc:442-444) and the vocabulary -- plus a MAX all-reduce of the elapsed time for reporting.
CPU tests run the same code over gloo.
"""
from __future__ import annotations

import hashlib
import os
import struct

import torch
import torch.distributed as dist

PARAM_FMT = "<ifiiiiii5f"  # nfeatures, scaleFactor, nlevels, iniTh, minTh, patch, halfPatch, edge, fx, fy, cx, cy, bf
PATTERN_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "orb_pattern_31.inc")


def world_info():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend: str, device=None, force: bool = False):
    """Process group over `backend` when more than one rank runs (or `force`: a one-rank group, which lets a one-GPU box exercise the
    very RCCL calls of the multi-GPU path -- every helper below runs its collective whenever a group exists)."""
    rank, _, world = world_info()
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        import datetime
        # a rank that never arrives fails the rendezvous of the others after this long instead of the default half hour
        kw["timeout"] = datetime.timedelta(seconds=int(os.environ.get("ORBFE_DIST_TIMEOUT_S", "300")))
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world


def pack_params(nfeatures, scale_factor, nlevels, ini_th, min_th, patch, half_patch, edge, fx, fy, cx, cy, bf) -> bytes:
    blob = struct.pack(PARAM_FMT, nfeatures, scale_factor, nlevels, ini_th, min_th, patch, half_patch, edge, fx, fy, cx, cy, bf)
    return blob + hashlib.sha256(open(PATTERN_PATH, "rb").read()).digest()


def unpack_params(blob: bytes):
    return struct.unpack(PARAM_FMT, blob[: struct.calcsize(PARAM_FMT)])


def broadcast_params(blob: bytes, device) -> bytes:
    """Rank 0's parameter blob on every rank; raises if this rank's pattern table differs from rank 0's."""
    t = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(device)
    if dist.is_initialized():
        mine = t.clone()
        dist.broadcast(t, src=0)
        n = struct.calcsize(PARAM_FMT)
        if not torch.equal(mine[n:], t[n:]):
            raise RuntimeError("rank %d: rBRIEF pattern differs from rank 0" % dist.get_rank())
    return bytes(t.cpu().numpy().tobytes())


def compiled_pattern():
    """bit_pattern_31_ as the library was compiled with it (orb_pattern_31.inc): int32 [256][4] = x0 y0 x1 y1 per test."""
    import re
    import numpy as np
    text = re.sub(r"/\*.*?\*/", "", open(PATTERN_PATH).read(), flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    v = np.array([int(t) for t in re.findall(r"-?\d+", text)], dtype=np.int32)
    if v.size != 1024:
        raise RuntimeError("orb_pattern_31.inc: expected 1024 integers, found %d" % v.size)
    return v.reshape(256, 4)


def broadcast_pattern(pattern, device):
    """The rBRIEF test table of rank 0 on every rank (BASELINE north_star: "RCCL broadcast of the ORB pattern"): 1024 int32
    + their sha256 in one collective; `pattern` is ignored on the other ranks (None allowed).  Returns int32 [256][4]; the
    caller passes it to Context.set_pattern.  Without a process group it is a checked copy."""
    import numpy as np
    rank = dist.get_rank() if dist.is_initialized() else 0
    if rank == 0:
        if pattern is None:
            raise ValueError("broadcast_pattern: rank 0 must supply the table")
        body = np.ascontiguousarray(np.asarray(pattern, dtype="<i4").reshape(1024)).tobytes()
        payload = body + hashlib.sha256(body).digest()
        t = torch.frombuffer(bytearray(payload), dtype=torch.uint8).to(device)
    else:
        t = torch.empty(4096 + 32, dtype=torch.uint8, device=device)
    if dist.is_initialized():
        dist.broadcast(t, src=0)
    raw = t.cpu().numpy().tobytes()
    if hashlib.sha256(raw[:4096]).digest() != raw[4096:]:
        raise RuntimeError("rank %d: pattern table corrupted in the broadcast" % rank)
    return np.frombuffer(raw[:4096], dtype="<i4").reshape(256, 4).copy()


def broadcast_blob(blob, device) -> bytes:
    """A byte string held by rank 0 (the fbow vocabulary file, Thirdparty/fbow/src/fbow.cpp:172-191 format; `blob` is
    ignored on the other ranks and may be None) on every rank: one 8-byte length broadcast, then the payload in one
    collective (backend "nccl" = RCCL over xGMI, device tensors; gloo: CPU tensors).  Every rank then hands the bytes to
    orbfe_vocab_load, which keeps the tree in its GPU's HBM.  A sha256 travels with the payload and is checked."""
    if not dist.is_initialized():
        if blob is None:
            raise ValueError("broadcast_blob: rank 0 must supply the blob")
        return bytes(blob)
    rank = dist.get_rank()
    if rank == 0 and blob is None:
        raise ValueError("broadcast_blob: rank 0 must supply the blob")
    n = torch.tensor([len(blob) if rank == 0 else 0], dtype=torch.int64, device=device)
    dist.broadcast(n, src=0)
    size = int(n.item())
    if rank == 0:
        payload = bytes(blob) + hashlib.sha256(bytes(blob)).digest()
        t = torch.frombuffer(bytearray(payload), dtype=torch.uint8).to(device)
    else:
        t = torch.empty(size + 32, dtype=torch.uint8, device=device)
    dist.broadcast(t, src=0)
    raw = t.cpu().numpy().tobytes()
    body, digest = raw[:size], raw[size:]
    if hashlib.sha256(body).digest() != digest:
        raise RuntimeError("rank %d: vocabulary blob corrupted in the broadcast" % rank)
    return body


def shard_pairs(total_pairs: int, rank: int, world: int):
    """Round-robin by frame index: pair i goes to rank i % world."""
    return list(range(rank, total_pairs, world))


def chain_schedule(steps: int, chains: int):
    """Steps 0 .. steps - 1 with `chains` step chains in flight: step k is enqueued on context / stream k % chains, once."""
    chains = max(1, int(chains))
    return [(k, k % chains) for k in range(int(steps))]


def max_over_ranks(value: float, device) -> float:
    if dist.is_initialized():
        t = torch.tensor([value], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    return value


def count_ranks(device) -> int:
    """Ranks that actually joined the group: a SUM all-reduce of 1 (1 without a group)."""
    if dist.is_initialized():
        t = torch.ones(1, dtype=torch.int64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return int(t.item())
    return 1


def barrier():
    if dist.is_initialized():
        dist.barrier()
