"""Input preparation: oracle against hand-computed cv2-convention cases (CPU), HIP kernels against the oracle (GPU)."""
import numpy as np
import pytest
import torch

from oracle import input_ref as I


def test_linear_matches_cv2_convention_by_hand():
    # 2x2 -> 4x4 with pixel-centre mapping: positions -0.25, 0.25, 0.75, 1.25 -> clamped weights 0, .25, .75, 1
    a = np.array([[0.0, 1.0], [2.0, 3.0]])
    r = I.resize_linear(a, 4, 4)
    np.testing.assert_allclose(r[0], [0.0, 0.25, 0.75, 1.0], atol=1e-12)
    np.testing.assert_allclose(r[:, 0], [0.0, 0.5, 1.5, 2.0], atol=1e-12)
    np.testing.assert_allclose(r[3, 3], 3.0)
    # same size = identity; 2x decimation of a ramp samples pixel pairs' means
    b = np.random.RandomState(0).rand(7, 5, 3)
    np.testing.assert_allclose(I.resize_linear(b, 7, 5), b, atol=1e-12)
    ramp = np.arange(8.0)[None, :].repeat(2, 0)
    np.testing.assert_allclose(I.resize_linear(ramp, 2, 4)[0], [0.5, 2.5, 4.5, 6.5], atol=1e-12)


def test_cubic_matches_cv2_convention_by_hand():
    # weights at x=0.5 are (-3/32, 19/32, 19/32, -3/32) for A=-0.75, at x=0.25 (-27, 225, 67, -9)/256; identity at equal size;
    # ramp 0..15 upsampled 2x: dst 3 -> src 1.25 -> taps 0,1,2,3 -> 225/256 + 2*67/256 - 3*9/256 = 1.296875 (A=-0.75 does
    # not reproduce a ramp exactly, unlike A=-0.5); dst 0 -> src -0.25 -> taps clamp to (0,0,0,1) at x=0.75
    np.testing.assert_allclose(I._cubic_w(np.array([0.5]))[:, 0], [-0.09375, 0.59375, 0.59375, -0.09375], atol=1e-12)
    np.testing.assert_allclose(I._cubic_w(np.array([0.25]))[:, 0], np.array([-27, 225, 67, -9]) / 256.0, atol=1e-12)
    b = np.random.RandomState(1).rand(6, 6, 2)
    np.testing.assert_allclose(I.resize_cubic(b, 6, 6), b, atol=1e-12)
    ramp = np.arange(16.0)[None, :, None].repeat(4, 0)
    r = I.resize_cubic(ramp, 4, 32)[0, :, 0]
    np.testing.assert_allclose(r[3], 1.296875, atol=1e-12)
    np.testing.assert_allclose(r[0], I._cubic_w(np.array([0.75]))[3, 0] * 1.0, atol=1e-12)


def test_resizer_geometry_and_thermal_stretch():
    assert I.resized_hw(600, 800, 512) == (384, 512) and I.resized_hw(1080, 1440, 512) == (384, 512)
    assert I.resized_hw(901, 700, 512) == (512, int(700 * (512 / 901)))
    rgb = np.random.RandomState(2).randint(0, 256, (30, 40, 3), dtype=np.uint8)
    out = I.prepare_rgb(rgb, 32)
    assert out.shape == (3, 32, 32) and out.dtype == np.float32
    assert np.all(out[:, 24:, :] == 0)                      # letterbox below the 24 resized rows
    t = np.array([[100, 21000], [26000, 65000]], dtype=np.uint16)
    o = I.prepare_thermal(t, 2)
    np.testing.assert_allclose(o[0], np.rint((np.clip(t, 20800, 27000) - 20800.0) * 255 / 6200) / 255.0, atol=1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("H,W,S", [(30, 40, 32), (61, 37, 48), (16, 16, 64), (270, 360, 128)])
def test_device_image_letterbox_matches_oracle(H, W, S):
    from mm_distillnet_amd import _lib
    call = _lib.call
    dev = "cuda:0"
    rs = np.random.RandomState(H * 7 + W)
    rgb = rs.randint(0, 256, (H, W, 3), dtype=np.uint8)
    depth = rs.randint(0, 256, (H, W, 3), dtype=np.uint8)
    th = rs.randint(15000, 30000, (H, W)).astype(np.uint16)
    mean = torch.tensor(I.IMAGENET_MEAN, dtype=torch.float32, device=dev)
    std = torch.tensor(I.IMAGENET_STD, dtype=torch.float32, device=dev)
    out = torch.full((3, S, S), 7.0, device=dev)
    call("mmd_image_letterbox", torch.from_numpy(rgb).to(dev), 0, H, W, 3, 1.0 / 255.0, mean, std, 0, 0.0, 0.0, None, S, out)
    np.testing.assert_allclose(out.cpu().numpy(), I.prepare_rgb(rgb, S), rtol=1e-5, atol=2e-5)
    call("mmd_image_letterbox", torch.from_numpy(depth).to(dev), 0, H, W, 3, 1.0 / 255.0, None, None, 0, 0.0, 0.0, None, S, out)
    np.testing.assert_allclose(out.cpu().numpy(), I.prepare_depth(depth, S), rtol=1e-5, atol=2e-6)
    tdev = torch.from_numpy(th.view(np.int16)).to(dev)       # torch has no uint16 arithmetic; the bytes are what travels
    mm = torch.zeros(2, device=dev)
    call("mmd_image_minmax", tdev, 1, H * W, 20800.0, 27000.0, mm)
    c = np.clip(th, 20800, 27000)
    assert mm.cpu().tolist() == [float(c.min()), float(c.max())]
    out1 = torch.full((1, S, S), 7.0, device=dev)
    call("mmd_image_letterbox", tdev, 1, H, W, 1, 1.0 / 255.0, None, None, 1, 20800.0, 27000.0, mm, S, out1)
    np.testing.assert_allclose(out1.cpu().numpy(), I.prepare_thermal(th, S), rtol=1e-5, atol=2e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("h,w,S", [(128, 128, 512), (64, 173, 96), (128, 128, 128)])
def test_device_cubic_resize_matches_oracle(h, w, S):
    from mm_distillnet_amd import _lib
    dev = "cuda:0"
    spec = (np.random.RandomState(h + w).randn(h, w, 8) * 15 - 40).astype(np.float32)
    out = torch.empty(8, S, S, device=dev)
    _lib.call("mmd_resize_cubic", torch.from_numpy(spec).to(dev), h, w, 8, S, out)
    np.testing.assert_allclose(out.cpu().numpy(), I.prepare_audio(spec, S), rtol=1e-5, atol=2e-4)


@pytest.mark.gpu
def test_device_pipeline_batch_matches_oracle():
    """Raw frames -> device batch through the copy-stream pipeline == the reference's transform chain per sample."""
    from mm_distillnet_amd.data import RawSyntheticMultimodalDetection, DeviceInputPipeline
    ds = RawSyntheticMultimodalDetection({"seed": 24, "image_size": 96}, length=3, frame_hw=(54, 72), mel_hw=(32, 32))
    samples = [ds[i] for i in range(3)]
    pipe = DeviceInputPipeline(96, "cuda:0")
    for _ in range(2):          # second submit reuses the pinned staging buffers
        batch = pipe.submit(samples).wait()
    torch.cuda.synchronize()
    assert batch["rgb"].shape == (3, 3, 96, 96) and batch["audio"].shape == (3, 8, 96, 96)
    for b, s in enumerate(samples):
        np.testing.assert_allclose(batch["rgb"][b].cpu().numpy(), I.prepare_rgb(s["rgb"].numpy(), 96), rtol=1e-5, atol=2e-5)
        np.testing.assert_allclose(batch["depth"][b].cpu().numpy(), I.prepare_depth(s["depth"].numpy(), 96), rtol=1e-5, atol=2e-6)
        th = s["thermal"].numpy().view(np.uint16)
        np.testing.assert_allclose(batch["thermal"][b].cpu().numpy(), I.prepare_thermal(th, 96), rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(batch["audio"][b].cpu().numpy(), I.prepare_audio(s["audio"].numpy(), 96), rtol=1e-5, atol=2e-4)


def test_collate_raw_stacks_equal_frames_only():
    from mm_distillnet_amd.data import RawSyntheticMultimodalDetection, collate_raw
    ds = RawSyntheticMultimodalDetection({"seed": 3}, length=3, frame_hw=(20, 24), mel_hw=(8, 8))
    samples = [ds[i] for i in range(3)]
    st = collate_raw(samples)
    assert isinstance(st, dict) and st["rgb"].shape == (3, 20, 24, 3) and st["thermal"].dtype == torch.int16 and st["id"] == [0, 1, 2]
    assert torch.equal(st["audio"][2], samples[2]["audio"])
    odd = dict(samples[1], rgb=samples[1]["rgb"][:10])
    out = collate_raw([samples[0], odd])          # mixed frame sizes: the per-sample list, untouched
    assert isinstance(out, list) and out[0] is samples[0] and out[1] is odd


@pytest.mark.gpu
def test_device_pipeline_stacked_and_pinned_paths_match_list_path():
    """The stacked batch `collate_raw` makes (one H2D copy per modality), pageable or already pinned as DataLoader(pin_memory=True)
    hands it over, gives the same device batch, bit for bit, as the list of per-sample dicts."""
    from mm_distillnet_amd.data import RawSyntheticMultimodalDetection, DeviceInputPipeline, collate_raw
    ds = RawSyntheticMultimodalDetection({"seed": 24, "image_size": 96}, length=3, frame_hw=(54, 72), mel_hw=(32, 32))
    samples = [ds[i] for i in range(3)]
    pipe = DeviceInputPipeline(96, "cuda:0")
    ref = {k: v.clone() for k, v in pipe.submit(samples).wait().items()}
    st = collate_raw(samples)
    got = {k: v.clone() for k, v in pipe.submit(st).wait().items()}
    pinned = {k: (v.pin_memory() if torch.is_tensor(v) else v) for k, v in st.items()}
    got_p = {k: v.clone() for k, v in pipe.submit(pinned).wait().items()}
    pipe.submit(st).wait()          # (the pinned batch's copies are consumed before its tensors may go)
    torch.cuda.synchronize()
    for k in ref:
        assert torch.equal(ref[k], got[k]) and torch.equal(ref[k], got_p[k]), k
