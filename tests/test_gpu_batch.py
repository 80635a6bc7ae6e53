"""Device-resident batched path (orbfe_enqueue_stereo / orbfe_fetch_image) and stream groups:
every pair of a batch must equal the single-frame result (and hence the oracle), whatever the grouping."""
import os

import numpy as np
import pytest

from orbslam2_amd import synth

pytestmark = pytest.mark.gpu

CFG = dict(width=400, height=200, nfeatures=400, fx=350.0, fy=350.0, cx=200.0, cy=100.0, bf=140.0)


@pytest.fixture(scope="module")
def batch():
    import torch
    from orbslam2_amd import api
    pairs = [synth.stereo_pair(CFG["width"], CFG["height"], seed=300 + i) for i in range(5)]
    single = api.Context(max_images=2, **CFG)
    ref = [single.stereo_frame(l, r) for l, r in pairs]
    single.close()
    host = np.stack([im for p in pairs for im in p])
    return dict(torch=torch, api=api, pairs=pairs, ref=ref, dev=torch.from_numpy(host).cuda())


@pytest.mark.parametrize("groups", [1, 2, 3, 5, 8])
def test_batched_stereo_equals_single_frames(batch, groups):
    api, torch = batch["api"], batch["torch"]
    ctx = api.Context(max_images=10, **CFG)
    ctx.set_streams(groups)
    for rep in range(2):  # second pass reuses every buffer
        ctx.enqueue_stereo(batch["dev"].data_ptr(), 5, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        counts = ctx.fetch_counts(10)
        for i, ref in enumerate(batch["ref"]):
            left = ctx.fetch_image(2 * i, stereo=True)
            right = ctx.fetch_image(2 * i + 1)
            assert counts[2 * i] == len(ref["kps_left"]) and counts[2 * i + 1] == len(ref["kps_right"])
            assert np.array_equal(left["kps"], ref["kps_left"]) and np.array_equal(left["desc"], ref["desc_left"])
            assert np.array_equal(right["kps"], ref["kps_right"]) and np.array_equal(right["desc"], ref["desc_right"])
            assert np.array_equal(left["u_right"], ref["u_right"]) and np.array_equal(left["depth"], ref["depth"])
    ctx.close()


def test_batched_extract_and_profiling(batch):
    api, torch = batch["api"], batch["torch"]
    ctx = api.Context(max_images=10, **CFG)
    ctx.set_streams(2)
    ctx.set_profiling(True)
    ctx.enqueue_extract(batch["dev"].data_ptr(), 7, 0)  # odd count, library's own stream
    ctx.synchronize()
    ms, calls = ctx.stage_times()
    assert calls == 1 and ms["fast"] > 0 and ms["stereo_match"] == 0
    for i in range(7):
        ref = batch["ref"][i // 2]
        k = "left" if i % 2 == 0 else "right"
        got = ctx.fetch_image(i)
        assert np.array_equal(got["kps"], ref["kps_" + k]) and np.array_equal(got["desc"], ref["desc_" + k])
    with pytest.raises(api.OrbfeError):
        ctx.enqueue_extract(batch["dev"].data_ptr(), 11, 0)  # more than max_images
    ctx.close()


def test_fetch_batch_async_into_pinned_buffers(batch):
    """orbfe_fetch_batch_async: the whole batch's results land in caller (pinned) buffers laid out like the device arrays, on
    the caller's stream; equal to the per-image fetches."""
    api, torch = batch["api"], batch["torch"]
    ctx = api.Context(max_images=10, **CFG)
    cap = ctx.capacity
    st = torch.cuda.Stream()
    h_k = torch.empty(10 * cap * 28, dtype=torch.uint8).pin_memory()
    h_d = torch.empty(10 * cap * 32, dtype=torch.uint8).pin_memory()
    h_c = torch.empty(10, dtype=torch.int32).pin_memory()
    h_u = torch.empty(10 * cap, dtype=torch.float32).pin_memory()
    h_z = torch.empty(10 * cap, dtype=torch.float32).pin_memory()
    torch.cuda.synchronize()
    ctx.enqueue_stereo(batch["dev"].data_ptr(), 5, st.cuda_stream)
    ctx.fetch_batch_async(10, h_k.data_ptr(), h_d.data_ptr(), h_c.data_ptr(), h_u.data_ptr(), h_z.data_ptr(), st.cuda_stream)
    st.synchronize()
    kps = np.frombuffer(h_k.numpy().tobytes(), api.KP_DTYPE).reshape(10, cap)
    desc = h_d.numpy().reshape(10, cap, 32)
    cnt = h_c.numpy()
    ur = h_u.numpy().reshape(10, cap); dp = h_z.numpy().reshape(10, cap)
    for i, ref in enumerate(batch["ref"]):
        nl, nr = len(ref["kps_left"]), len(ref["kps_right"])
        assert cnt[2 * i] == nl and cnt[2 * i + 1] == nr
        assert np.array_equal(kps[2 * i, :nl], ref["kps_left"]) and np.array_equal(kps[2 * i + 1, :nr], ref["kps_right"])
        assert np.array_equal(desc[2 * i, :nl], ref["desc_left"]) and np.array_equal(desc[2 * i + 1, :nr], ref["desc_right"])
        assert np.array_equal(ur[2 * i, :nl], ref["u_right"]) and np.array_equal(dp[2 * i, :nl], ref["depth"])
    ctx.close()


KITTI = dict(width=1241, height=376, nfeatures=2000, fx=718.856, fy=718.856, cx=607.1928, cy=185.2157, bf=386.1448)


def test_kitti_batch_of_64_pairs_every_distinct_pair_vs_oracle():
    """BASELINE.json config 4 at its stated size: 64 KITTI-geometry stereo pairs in flight in ONE orbfe_enqueue_stereo call
    (what bench.py times), built from 16 distinct seeds; EVERY slot's keypoints (all fields), descriptors, uRight and depth
    are compared with the CPU oracle of its pair, bit for bit (uRight / depth: north_star's 1e-4 is met with margin 0)."""
    import torch
    from oracle import oracle as O
    from orbslam2_amd import api
    P, ND = 64, 16
    distinct = [synth.stereo_pair(KITTI["width"], KITTI["height"], seed=7100 + 13 * i) for i in range(ND)]
    refs = []
    for l, r in distinct:
        exl, exr = O.Extractor(nfeatures=KITTI["nfeatures"]), O.Extractor(nfeatures=KITTI["nfeatures"])
        kl, dl = exl.extract(l); kr, dr = exr.extract(r)
        ur, dp, m = O.stereo_matches(exl, exr, kl, dl, kr, dr, KITTI["bf"], KITTI["fx"])
        assert m > 100
        refs.append((kl, dl, kr, dr, ur, dp))
    order = [(5 * i + 3) % ND for i in range(P)]  # neighbouring slots hold different pairs
    host = np.empty((2 * P, KITTI["height"], KITTI["width"]), np.uint8)
    for i, k in enumerate(order):
        host[2 * i], host[2 * i + 1] = distinct[k]
    dev = torch.from_numpy(host).cuda()
    ctx = api.Context(max_images=2 * P, **KITTI)
    for groups in (1, 2):
        ctx.set_streams(groups)
        ctx.enqueue_stereo(dev.data_ptr(), P, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        counts = ctx.fetch_counts(2 * P)
        for i, k in enumerate(order):
            kl, dl, kr, dr, ur, dp = refs[k]
            left = ctx.fetch_image(2 * i, stereo=True)
            right = ctx.fetch_image(2 * i + 1)
            what = "groups %d slot %d (pair %d)" % (groups, i, k)
            assert counts[2 * i] == len(kl) and counts[2 * i + 1] == len(kr), what
            for f in ("x", "y", "size", "angle", "response", "octave", "class_id"):
                assert np.array_equal(left["kps"][f], kl[f]), what + " left " + f
                assert np.array_equal(right["kps"][f], kr[f]), what + " right " + f
            assert np.array_equal(left["desc"], dl) and np.array_equal(right["desc"], dr), what
            assert np.array_equal(left["u_right"], ur) and np.array_equal(left["depth"], dp), what
    ctx.close()


def test_sharded_batch_on_two_ranks_covers_every_pair_once(batch):
    """Strong-scaling layout of bench.py --mode strong on one GPU: two contexts play ranks 0 and 1 of world size 2, each
    extracts dist.shard_pairs(5, rank, 2) of one global list; the union is every pair exactly once, equal to the
    single-frame results."""
    from orbslam2_amd import dist as D
    api, torch = batch["api"], batch["torch"]
    seen = {}
    for rank in range(2):
        mine = D.shard_pairs(5, rank, 2)
        host = np.stack([im for i in mine for im in batch["pairs"][i]])
        dev = torch.from_numpy(host).cuda()
        ctx = api.Context(max_images=2 * len(mine), **CFG)
        ctx.enqueue_stereo(dev.data_ptr(), len(mine), 0)
        for j, i in enumerate(mine):
            assert i not in seen
            seen[i] = (ctx.fetch_image(2 * j, stereo=True), ctx.fetch_image(2 * j + 1))  # fetch waits for the enqueue stream
        ctx.close()
    assert sorted(seen) == list(range(5))
    for i, ref in enumerate(batch["ref"]):
        left, right = seen[i]
        assert np.array_equal(left["kps"], ref["kps_left"]) and np.array_equal(left["desc"], ref["desc_left"])
        assert np.array_equal(right["kps"], ref["kps_right"]) and np.array_equal(left["u_right"], ref["u_right"])


def test_rccl_collectives_of_the_sharded_bench_on_one_rank(tmp_path):
    """The multi-GPU path's collectives -- parameter + pattern-checksum broadcast, vocabulary broadcast, MAX all-reduce of the elapsed
    time, barrier -- through the REAL backend (torch.distributed "nccl" = RCCL, device tensors) as a one-rank group: the same calls
    bench.py makes on 8 GPUs, runnable on a one-GPU box.  (Two gloo ranks cover the N > 1 logic: tests/test_sharding_gloo.py.)"""
    import subprocess
    import sys
    script = tmp_path / "rccl_one_rank.py"
    script.write_text('''
import os, sys, hashlib
sys.path.insert(0, %r)
os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
import numpy as np, torch
from orbslam2_amd import dist as D
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
D.init("nccl", dev, force=True)
import torch.distributed as dist
assert dist.is_initialized() and dist.get_backend() == "nccl"
blob = D.pack_params(2000, 1.2, 8, 20, 7, 31, 15, 19, 718.856, 718.856, 607.19, 185.22, 386.14)
assert D.broadcast_params(blob, dev) == blob
voc = np.random.default_rng(1).integers(0, 256, 3_000_000, dtype=np.uint8).tobytes()
got = D.broadcast_blob(voc, dev)
assert hashlib.sha256(got).digest() == hashlib.sha256(voc).digest()
assert D.max_over_ranks(1.25, dev) == 1.25
D.barrier()
torch.cuda.synchronize()
dist.destroy_process_group()
print("rccl one-rank ok")
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "rccl one-rank ok" in r.stdout, r.stdout + r.stderr[-2000:]
