"""GPU: device-pointer entry points (the path bench.py drives) through torch tensors on a side stream."""
import numpy as np
import pytest

from minivideo_amd.synth import synth_packed
from oracle import loader

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("n_frames", [3, 400])          # 3 -> 16-wave workgroups, 400 -> 8-wave workgroups
def test_recon_dev_matches_oracle(hot, torch_cuda, fused, n_frames):
    torch = torch_cuda
    W, H = 13, 9
    distinct = 5
    params, rec = synth_packed(W, H, distinct, seed=n_frames + fused, profile="high")
    idx = np.arange(n_frames) % distinct
    packed = np.ascontiguousarray(rec[idx])
    dev = torch.device("cuda", 0)
    d_packed = torch.from_numpy(packed.reshape(-1)).to(dev)
    d_yuv = torch.zeros(n_frames * params.yuv_bytes, dtype=torch.uint8, device=dev)
    d_rgb = torch.zeros(n_frames * params.rgb_bytes, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize(dev)
    st = torch.cuda.Stream(device=dev)
    hot.set_fused_color(fused)
    try:
        hot.recon_dev(params, d_packed.data_ptr(), n_frames, d_yuv.data_ptr(), d_rgb.data_ptr(), st.cuda_stream)
        hot.sync_check(st.cuda_stream)
    finally:
        hot.set_fused_color(True)
    yuv_o, rgb_o = loader.recon(params, rec, distinct, want_rgb=True)
    yuv_g = d_yuv.cpu().numpy().reshape(n_frames, -1)
    rgb_g = d_rgb.cpu().numpy().reshape(n_frames, -1)
    yuv_o = yuv_o.reshape(distinct, -1)
    rgb_o = rgb_o.reshape(distinct, -1)
    for f in range(n_frames):
        assert np.array_equal(yuv_g[f], yuv_o[idx[f]]), f
        assert np.array_equal(rgb_g[f], rgb_o[idx[f]]), f


def test_yuv_only_leaves_rgb_untouched(hot, torch_cuda):
    torch = torch_cuda
    params, rec = synth_packed(6, 4, 2, seed=3)
    dev = torch.device("cuda", 0)
    d_packed = torch.from_numpy(rec.reshape(-1)).to(dev)
    d_yuv = torch.zeros(2 * params.yuv_bytes, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize(dev)
    st = torch.cuda.Stream(device=dev)
    hot.recon_dev(params, d_packed.data_ptr(), 2, d_yuv.data_ptr(), None, st.cuda_stream)
    hot.sync_check(st.cuda_stream)
    yuv_o, _ = loader.recon(params, rec, 2)
    assert np.array_equal(d_yuv.cpu().numpy(), yuv_o)


def test_invalid_arguments_fail_loudly(hot):
    from minivideo_amd import MiniVideoError
    from minivideo_amd.hotpath import StreamParams
    with pytest.raises(MiniVideoError):
        hot.recon_dev(StreamParams(0, 4, 0, 0, 0), 1, 1, 1, None, None)
    with pytest.raises(MiniVideoError):
        hot.recon_dev(StreamParams(4, 4, 40, 0, 0), 1, 1, 1, None, None)
