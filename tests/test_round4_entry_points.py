"""Round-4 entry points and launch plans: the packed result block (orbfe_fetch_batch_packed + orbfe_expand_packed must equal
orbfe_fetch_batch_async / orbfe_fetch_image bit for bit), the batched RGB-D chain (orbfe_enqueue_rgbd == N orbfe_rgbd_frame calls),
and pyramid level 0 read in place from the caller's packed images (== the ingest copy, at every row alignment, up to the very last
byte of the caller's buffer)."""
import ctypes as C
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
    host = np.stack([im for p in pairs for im in p])
    return dict(torch=torch, api=api, pairs=pairs, dev=torch.from_numpy(host).cuda())


@pytest.mark.parametrize("flags", [0, 1, 2, 3])
def test_packed_block_expands_to_the_unpacked_fetch(batch, flags):
    api, torch = batch["api"], batch["torch"]
    ctx = api.Context(max_images=10, **CFG)
    st = torch.cuda.Stream()
    ctx.enqueue_stereo(batch["dev"].data_ptr(), 5, st.cuda_stream)
    lay = ctx.packed_layout(10, flags)
    pinned = torch.empty(lay.bytes, dtype=torch.uint8).pin_memory()
    ctx.fetch_batch_packed(10, flags, pinned.data_ptr(), lay.bytes, st.cuda_stream)
    st.synchronize()
    block = pinned.numpy()
    assert lay.n_images_out == (5 if flags & api.PACK_LEFT_ONLY else 10) and lay.n_pairs == (5 if flags & api.PACK_STEREO else 0)
    total = 0
    for o in range(lay.n_images_out):
        slot = o * (2 if flags & api.PACK_LEFT_ONLY else 1)
        ref = ctx.fetch_image(slot, stereo=slot % 2 == 0)
        got = ctx.expand_packed(block, lay, o)
        assert got["kps"].tobytes() == ref["kps"].tobytes()  # every field of every record, bit for bit (angle incl.)
        assert np.array_equal(got["desc"], ref["desc"])
        if flags & api.PACK_STEREO and slot % 2 == 0:
            assert got["u_right"].tobytes() == ref["u_right"].tobytes() and got["depth"].tobytes() == ref["depth"].tobytes()
        else:
            assert "u_right" not in got
        total += len(ref["kps"])
    assert total > 1000
    # fewer bytes than the unpacked fetch of the same slots: 28 + 32 + 8 B per slot of every image
    assert lay.bytes < 10 * ctx.capacity * (28 + 32 + 8) * (0.75 if flags == api.PACK_STEREO else 1.0)
    with pytest.raises(api.OrbfeError):
        ctx.fetch_batch_packed(10, flags, pinned.data_ptr(), lay.bytes - 64, st.cuda_stream)  # block too small
    with pytest.raises(api.OrbfeError):
        ctx.packed_layout(9, api.PACK_STEREO)  # pairs need an even number of slots
    ctx.close()


def test_packed_block_on_goldens_and_natural_pairs():
    """The expansion on every committed stereo golden and natural pair: records equal the single-frame entry point's."""
    import torch
    from orbslam2_amd import api
    from tests import natural as N
    cases = []
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    for name in ("stereo_320x240_f500.npz", "stereo_400x160_f300.npz"):
        w, h, nf, fx, bf, seed = np.load(os.path.join(gdir, name))["params"]
        l, r = synth.stereo_pair(int(w), int(h), seed=int(seed))
        cases.append((l, r, int(nf)))
    for nm in N.PAIRS:
        l, r, d, nf = N.pair(nm)
        cases.append((l, r, nf))
    assert cases
    for left, right, nf in cases:
        h, w = left.shape
        ctx = api.Context(width=w, height=h, nfeatures=nf, fx=300.0, fy=300.0, cx=w / 2, cy=h / 2, bf=100.0, max_images=2)
        ref = ctx.stereo_frame(left, right)
        dev = torch.from_numpy(np.stack([left, right])).cuda()
        ctx.enqueue_stereo(dev.data_ptr(), 1, 0)
        block, lay = ctx.fetch_packed(2, api.PACK_STEREO)
        gl, gr = ctx.expand_packed(block, lay, 0), ctx.expand_packed(block, lay, 1)
        assert gl["kps"].tobytes() == ref["kps_left"].tobytes() and gr["kps"].tobytes() == ref["kps_right"].tobytes()
        assert np.array_equal(gl["desc"], ref["desc_left"]) and np.array_equal(gr["desc"], ref["desc_right"])
        assert gl["u_right"].tobytes() == ref["u_right"].tobytes() and gl["depth"].tobytes() == ref["depth"].tobytes()
        ctx.close()


@pytest.mark.parametrize("u16", [False, True])
def test_enqueue_rgbd_equals_single_rgbd_frames(u16):
    import torch
    from orbslam2_amd import api
    W, H, NF, n = 424, 240, 500, 5
    cfg = dict(width=W, height=H, nfeatures=NF, fx=300.0, fy=300.0, cx=W / 2, cy=H / 2, bf=15.0)
    frames = []
    for i in range(n):
        img, _, depth = synth.stereo_pair(W, H, seed=700 + i, with_depth=True, bf=cfg["bf"])
        if u16:
            depth = np.clip(np.round(depth * 1000.0), 0, 65535).astype(np.uint16)
        frames.append((img, depth))
    single = api.Context(max_images=1, **cfg)
    factor = 1.0 / 1000.0
    ref = [single.rgbd_frame(g, d, factor) for g, d in frames]
    single.close()
    ctx = api.Context(max_images=n, **cfg)
    d_gray = torch.from_numpy(np.stack([g for g, _ in frames])).cuda()
    dstack = np.stack([d for _, d in frames])
    d_depth = torch.from_numpy(dstack.view(np.int16) if u16 else dstack).cuda()  # torch has no uint16 on every build: same bytes
    st = torch.cuda.Stream()
    for rep in range(2):
        ctx.enqueue_rgbd(d_gray.data_ptr(), d_depth.data_ptr(), n, depth_is_u16=u16, depth_map_factor=factor, stream=st.cuda_stream)
        st.synchronize()
        for i in range(n):
            got = ctx.fetch_image(i, stereo=True)
            assert got["kps"].tobytes() == ref[i]["kps"].tobytes() and np.array_equal(got["desc"], ref[i]["desc"])
            assert got["u_right"].tobytes() == ref[i]["u_right"].tobytes() and got["depth"].tobytes() == ref[i]["depth"].tobytes()
    assert sum((r["depth"] > 0).sum() for r in ref) > 200
    ctx.close()


@pytest.mark.parametrize("w,h", [(401, 203), (402, 201), (403, 202), (404, 200), (1241, 376), (640, 480)])
def test_level0_in_place_equals_the_ingest_copy_at_every_row_alignment(w, h, monkeypatch):
    """Widths of every residue mod 4 (rows of a packed image then start at every byte alignment), through the batched path where
    the library reads the caller's buffer in place; the copy plan (ORBFE_NO_INPLACE=1) is the round-3 path."""
    import torch
    from orbslam2_amd import api
    nf = 600
    cfg = dict(width=w, height=h, nfeatures=nf, fx=350.0, fy=350.0, cx=w / 2, cy=h / 2, bf=140.0, max_images=4)
    pairs = [synth.stereo_pair(w, h, seed=40 + i) for i in range(2)]
    host = np.stack([im for p in pairs for im in p])
    # 3 spare bytes in front: the images then start at an odd address
    raw = torch.zeros(host.size + 3, dtype=torch.uint8).cuda()
    for lead in (0, 1, 3):
        dev = raw[lead:lead + host.size]
        dev.copy_(torch.from_numpy(host.reshape(-1)))
        res = {}
        for mode in ("1", "0"):
            monkeypatch.setenv("ORBFE_NO_INPLACE", mode)
            ctx = api.Context(**cfg)
            ctx.enqueue_stereo(dev.data_ptr(), 2, 0)
            ctx.synchronize()
            res[mode] = [ctx.fetch_image(i, stereo=i % 2 == 0) for i in range(4)]
            # read in place, level 0 is the caller's buffer: the library only follows that pointer for callers that keep it valid
            # (round-4 advisor finding); the copy plans (ORBFE_NO_INPLACE=1, or ORBFE_PYR_LDS=1 whose level 1 cannot read in place) own level 0
            in_place = mode == "0" and os.environ.get("ORBFE_PYR_LDS") != "1"
            try:
                ctx.fetch_pyramid(1, 0)
                assert not in_place
            except api.OrbfeError as e:
                assert in_place and e.code == -5 and "orbfe_set_input_retained" in str(e)
                ctx.set_input_retained(True)
            lv0 = ctx.fetch_pyramid(1, 0)
            assert np.array_equal(lv0, host[1])
            blur = [ctx.fetch_pyramid(3, l, blurred=True) for l in range(2)]
            res[mode + "b"] = blur
            ctx.close()
        for a, b in zip(res["1"], res["0"]):
            assert a["kps"].tobytes() == b["kps"].tobytes() and np.array_equal(a["desc"], b["desc"])
            if "u_right" in a:
                assert a["u_right"].tobytes() == b["u_right"].tobytes()
        for a, b in zip(res["1b"], res["0b"]):
            assert np.array_equal(a, b)
        assert len(res["0"][0]["kps"]) > 100


def test_level0_in_place_never_reads_past_the_callers_buffer():
    """1280 x 720 images are a whole number of 4 KiB pages: a batch allocated exactly ends on a page boundary, so a read past the
    last image's last byte would be a read of unmapped memory.  hipMalloc pads to its own granularity, so the images are placed
    at the very END of a larger allocation whose size is a multiple of 2 MiB."""
    import torch
    from orbslam2_amd import api
    from oracle import oracle as O
    W, H, NF = 1280, 720, 1000
    img = synth.stereo_pair(W, H, seed=5)[0]
    total = 2 * 1024 * 1024
    raw = torch.zeros(total, dtype=torch.uint8).cuda()
    dev = raw[total - W * H:]
    dev.copy_(torch.from_numpy(img.reshape(-1)))
    ctx = api.Context(width=W, height=H, nfeatures=NF, fx=900.0, fy=900.0, cx=W / 2, cy=H / 2, bf=45.0, max_images=1)
    ctx.enqueue_extract(dev.data_ptr(), 1, 0)
    ctx.synchronize()
    got = ctx.fetch_image(0)
    k, d = O.Extractor(nfeatures=NF).extract(img)
    assert got["kps"].tobytes() == k.astype(api.KP_DTYPE).tobytes() and np.array_equal(got["desc"], d)
    ctx.close()


def test_packed_block_stored_directly_into_pinned_host_memory(batch):
    """ORBFE_PACK_DIRECT: the gather kernel writes the block into pinned host memory itself (no copy queued); same bytes as the copied
    block; ordinary (pageable) memory is refused, not written to."""
    api, torch = batch["api"], batch["torch"]
    ctx = api.Context(max_images=10, **CFG)
    st = torch.cuda.Stream()
    ctx.enqueue_stereo(batch["dev"].data_ptr(), 5, st.cuda_stream)
    flags = api.PACK_STEREO
    lay = ctx.packed_layout(10, flags)
    copied = torch.zeros(lay.bytes, dtype=torch.uint8).pin_memory()
    direct = torch.zeros(lay.bytes, dtype=torch.uint8).pin_memory()
    ctx.fetch_batch_packed(10, flags, copied.data_ptr(), lay.bytes, st.cuda_stream)
    ctx.fetch_batch_packed(10, flags | api.PACK_DIRECT, direct.data_ptr(), lay.bytes, st.cuda_stream)
    st.synchronize()
    a, b = copied.numpy(), direct.numpy()
    for o in range(10):
        ga, gb = ctx.expand_packed(a, lay, o), ctx.expand_packed(b, lay, o)
        assert ga["kps"].tobytes() == gb["kps"].tobytes() and np.array_equal(ga["desc"], gb["desc"])
        if o % 2 == 0:
            assert ga["u_right"].tobytes() == gb["u_right"].tobytes() and ga["depth"].tobytes() == gb["depth"].tobytes()
    pageable = np.zeros(lay.bytes, np.uint8)
    with pytest.raises(api.OrbfeError):
        ctx.fetch_batch_packed(10, flags | api.PACK_DIRECT, pageable.ctypes.data, lay.bytes, st.cuda_stream)
    assert not pageable.any()
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("knob,off", [("ORBFE_NO_PROC_ORDER", "1"), ("ORBFE_BLUR_RIDE_FROM", "0"), ("ORBFE_BLUR_RIDE_FROM", "3")])
@pytest.mark.parametrize("w,h,nf,n_pairs", [(1241, 376, 2000, 3), (640, 480, 1000, 1), (403, 202, 300, 2)])
def test_launch_plan_knobs_change_no_result(knob, off, w, h, nf, n_pairs, monkeypatch):
    """describe_kernel walking the quadtree kernel's spatial processing order (default) or the slots (ORBFE_NO_PROC_ORDER=1), and the
    blur of every level / of levels 3.. riding in FAST's launch (ORBFE_BLUR_RIDE_FROM: the default of 64-image batches, forced here on small
    ones) instead of beside the resize launches: the same bytes out, also for the blurred pyramid of every level (which the knob moves between launches)."""
    import torch
    from orbslam2_amd import api
    cfg = dict(width=w, height=h, nfeatures=nf, fx=350.0, fy=350.0, cx=w / 2, cy=h / 2, bf=140.0, max_images=2 * n_pairs)
    pairs = [synth.stereo_pair(w, h, seed=90 + i) for i in range(n_pairs)]
    host = np.stack([im for p in pairs for im in p])
    dev = torch.from_numpy(host).cuda()
    res = {}
    for mode in ("default", "off"):
        if mode == "off":
            monkeypatch.setenv(knob, off)
        ctx = api.Context(**cfg)
        ctx.enqueue_stereo(dev.data_ptr(), n_pairs, 0)
        ctx.synchronize()
        res[mode] = [ctx.fetch_image(i, stereo=i % 2 == 0) for i in range(2 * n_pairs)]
        res[mode + "b"] = [ctx.fetch_pyramid(2 * n_pairs - 1, l, blurred=True) for l in range(8)]
        ctx.close()
    for a, b in zip(res["default"], res["off"]):
        assert a["kps"].tobytes() == b["kps"].tobytes() and np.array_equal(a["desc"], b["desc"])
        if "u_right" in a:
            assert a["u_right"].tobytes() == b["u_right"].tobytes() and a["depth"].tobytes() == b["depth"].tobytes()
    for a, b in zip(res["defaultb"], res["offb"]):
        assert np.array_equal(a, b)
    assert len(res["default"][0]["kps"]) > 100
