"""Device-resident batched path (orbfe_enqueue_stereo / orbfe_fetch_image) and stream groups:
every pair of a batch must equal the single-frame result (and hence the oracle), whatever the grouping."""
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
