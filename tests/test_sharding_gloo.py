"""N>1 path on CPU: two gloo ranks shard frame pairs exactly like bench.py's ranks do over RCCL.
The per-pair compute here is the oracle (test infrastructure); what is under test is the sharding, the one-time
parameter broadcast and the max-over-ranks reduction -- results must not depend on the world size."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOTAL_PAIRS = 5
W, H, NF, FX, BF = 200, 140, 150, 250.0, 100.0


def _pair_result(i):
    from oracle import oracle as O
    from orbslam2_amd import synth
    left, right = synth.stereo_pair(W, H, seed=900 + i)
    exl, exr = O.Extractor(nfeatures=NF), O.Extractor(nfeatures=NF)
    kl, dl = exl.extract(left); kr, dr = exr.extract(right)
    ur, dp, m = O.stereo_matches(exl, exr, kl, dl, kr, dr, BF, FX)
    return (len(kl), len(kr), m, int(dl.astype(np.int64).sum()), float(ur.sum()))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    from orbslam2_amd import dist as D
    D.init("gloo")
    dev = torch.device("cpu")
    # rank 0 owns the parameters; other ranks start from garbage and must receive rank 0's
    mine = D.pack_params(NF, 1.2, 8, 20, 7, 31, 15, 19, FX, FX, W / 2, H / 2, BF) if rank == 0 else \
        D.pack_params(1, 2.0, 3, 40, 30, 31, 15, 19, 1.0, 1.0, 0.0, 0.0, 1.0)
    blob = D.broadcast_params(mine, dev)
    p = D.unpack_params(blob)
    assert p[0] == NF and p[2] == 8 and abs(p[8] - FX) < 1e-6, p
    # vocabulary: only rank 0 holds the fbow blob; every rank must end up with the same bytes (what orbfe_vocab_load gets),
    # and the oracle's loader must accept them on every rank
    import hashlib
    voc = None
    if rank == 0:
        from orbslam2_amd import bow as B
        rng = np.random.default_rng(3)
        voc = B.build_vocabulary(rng.integers(0, 256, (1500, 32), dtype=np.uint8), k=10, levels=3)
    voc = D.broadcast_blob(voc, dev)
    digests = [None] * world
    dist.all_gather_object(digests, (len(voc), hashlib.sha256(voc).hexdigest()))
    assert len(set(digests)) == 1 and digests[0][0] > 10000, digests
    import ctypes as C
    from oracle import oracle as O
    OL = O.lib()
    OL.orc_vocab_from_blob.restype = C.c_void_p; OL.orc_vocab_from_blob.argtypes = [C.c_void_p, C.c_size_t]
    OL.orc_vocab_destroy.argtypes = [C.c_void_p]; OL.orc_vocab_destroy.restype = None
    buf = np.frombuffer(voc, np.uint8)
    hv = OL.orc_vocab_from_blob(buf.ctypes.data_as(C.c_void_p), len(buf))
    assert hv
    OL.orc_vocab_destroy(hv)
    # the rBRIEF test table itself travels (north_star: "broadcast of the ORB pattern"): rank 0 owns a PERMUTED table, the other
    # ranks start from nothing; every rank must end up with rank 0's 1024 integers, and an extractor given them must differ from
    # the compiled table's descriptors in the same way on every rank
    pat0 = None
    if rank == 0:
        pat0 = D.compiled_pattern()[np.random.default_rng(11).permutation(256)]
    pat = D.broadcast_pattern(pat0, dev)
    pd = [None] * world
    dist.all_gather_object(pd, hashlib.sha256(pat.tobytes()).hexdigest())
    assert len(set(pd)) == 1, pd
    assert pat.shape == (256, 4) and not np.array_equal(pat, D.compiled_pattern())
    assert sorted(map(tuple, pat.tolist())) == sorted(map(tuple, D.compiled_pattern().tolist()))
    from orbslam2_amd import synth
    img = synth.stereo_pair(W, H, seed=900)[0]
    ex_def, ex_pat = O.Extractor(nfeatures=NF), O.Extractor(nfeatures=NF)
    ex_pat.set_pattern(pat)
    assert np.array_equal(ex_pat.pattern(), pat)
    (k0, d0), (k1, d1) = ex_def.extract(img), ex_pat.extract(img)
    assert np.array_equal(k0, k1) and not np.array_equal(d0, d1) # the table only enters the descriptors
    dd = [None] * world
    dist.all_gather_object(dd, hashlib.sha256(d1.tobytes()).hexdigest())
    assert len(set(dd)) == 1, dd
    my_pairs = D.shard_pairs(TOTAL_PAIRS, rank, world)
    res = {i: _pair_result(i) for i in my_pairs}
    gathered = [None] * world
    dist.all_gather_object(gathered, res)
    t = D.max_over_ranks(1.0 + rank, dev)
    assert t == float(world)
    D.barrier()
    if rank == 0:
        merged = {}
        for g in gathered:
            assert not (set(g) & set(merged)), "a pair was processed twice"
            merged.update(g)
        q.put(merged)
    dist.destroy_process_group()


def test_two_ranks_partition_the_batch_and_match_single_process():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    merged = q.get()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert sorted(merged) == list(range(TOTAL_PAIRS))
    for i in range(TOTAL_PAIRS):
        assert merged[i] == _pair_result(i)


def test_shard_pairs_round_robin():
    from orbslam2_amd import dist as D
    for world in (1, 2, 4, 8):
        seen = sorted(i for r in range(world) for i in D.shard_pairs(64, r, world))
        assert seen == list(range(64))
        assert all(len(D.shard_pairs(64, r, world)) == 64 // world for r in range(world))
