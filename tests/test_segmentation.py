"""ht / lt producer (ra-slam_amd/ratsdf/segmentation.py): size handling, ones when absent, and that
its device-resident maps integrate exactly like the same maps passed from the host."""
import numpy as np
import pytest
import torch

from ratsdf.segmentation import InferenceEngine


class Toy(torch.nn.Module):
    """Stands in for the reference's TorchScript network: 3 -> 2 channels, output in (0, 1)."""

    def __init__(self):
        super().__init__()
        self.conv = torch.nn.Conv2d(3, 2, 3, padding=1)

    def forward(self, x):
        return torch.sigmoid(self.conv(x))


def scripted(tmp_path):
    torch.manual_seed(3)
    path = tmp_path / "toy.pt"
    torch.jit.script(Toy()).save(str(path))
    return path


def test_absent_model_means_ones():
    eng = InferenceEngine("", 64, 48, device="cpu")
    assert eng.infer_one(np.zeros((48, 64, 3), dtype=np.uint8)) == (None, None)


def test_sizes_follow_the_reference(tmp_path):
    eng = InferenceEngine(scripted(tmp_path), 640, 480, device="cpu")
    assert (eng.whole_width, eng.whole_height) == (672, 512)      # ((int)(w / 32) + 1) * 32
    eng = InferenceEngine(scripted(tmp_path), 80, 60, device="cpu")
    assert (eng.whole_width, eng.whole_height) == (96, 64)
    rgb = np.random.default_rng(0).integers(0, 256, (60, 80, 3), dtype=np.uint8)
    ht, lt = eng.infer_one(rgb)
    assert ht.shape == (60, 80) and lt.shape == (60, 80) and ht.dtype == torch.float32
    assert float(ht.min()) > 0 and float(ht.max()) < 1
    # a constant image gives a map that is constant away from the borders (resize x conv x resize)
    ht2, _ = eng.infer_one(np.full((60, 80, 3), 128, dtype=np.uint8))
    inner = ht2[8:-8, 8:-8]
    assert float(inner.max() - inner.min()) < 1e-5


def _toy_numpy(path):
    """the scripted toy network in numpy (3x3 correlation, zero padding, sigmoid)"""
    from scipy.signal import correlate2d
    m = torch.jit.load(str(path), map_location="cpu")
    wgt = m.conv.weight.detach().numpy().astype(np.float64)
    bias = m.conv.bias.detach().numpy().astype(np.float64)

    def net(x):
        x = x[0].astype(np.float64)
        out = np.stack([sum(correlate2d(x[c], wgt[o, c], mode="same") for c in range(3)) + bias[o]
                        for o in range(2)])
        return (1.0 / (1.0 + np.exp(-out))).astype(np.float32)
    return net


def test_resize_u8_matches_the_restated_opencv_kernel():
    """device (torch, integer arithmetic) == numpy restatement of cv::resize INTER_LINEAR for 8-bit
    images, bit for bit: up- and down-scaling, odd sizes, the ScanNet colour size."""
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "oracle"))
    import segmentation_oracle as O
    from ratsdf.segmentation import resize_u8_linear
    rng = np.random.default_rng(5)
    for (h, w), (oh, ow) in [((48, 64), (64, 96)), ((61, 83), (64, 96)), ((97, 130), (48, 64)),
                             ((968, 1296), (480, 640)), ((5, 7), (5, 7)), ((2, 2), (7, 5))]:
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        want = O.resize_u8_linear(img, oh, ow)
        got = resize_u8_linear(torch.from_numpy(img), oh, ow).numpy()
        assert np.array_equal(got, want), ((h, w), (oh, ow), int(np.abs(got.astype(int) - want).max()))
    # a constant image stays constant, an identity resize is the identity
    img = np.full((10, 12, 3), 200, np.uint8)
    assert np.all(O.resize_u8_linear(img, 32, 32) == 200)
    img = rng.integers(0, 256, (9, 11, 3), dtype=np.uint8)
    assert np.array_equal(O.resize_u8_linear(img, 9, 11), img)


def test_producer_matches_the_numpy_restatement(tmp_path):
    """The whole producer (8-bit resize -> [0,1] -> network -> float resize back; and the uint8
    output path) against oracle/segmentation_oracle.py with the same toy network in numpy."""
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "oracle"))
    import segmentation_oracle as O
    path = scripted(tmp_path)
    eng = InferenceEngine(path, 80, 60, device="cpu")
    net = _toy_numpy(path)
    rgb = np.random.default_rng(11).integers(0, 256, (60, 80, 3), dtype=np.uint8)
    ht, lt = eng.infer_one(rgb)
    want = O.infer_one(rgb, 80, 60, net)
    assert np.max(np.abs(ht.numpy() - want[0])) < 2e-6 and np.max(np.abs(lt.numpy() - want[1])) < 2e-6
    u = eng.infer_one(rgb, ret_uint8=True)
    wu = O.infer_one(rgb, 80, 60, net, ret_uint8=True)
    assert u[0].shape == (64, 96) and u[0].dtype == torch.uint8          # network resolution
    for a, b in zip(u, wu):
        assert np.max(np.abs(a.numpy().astype(int) - b.astype(int))) <= 1   # float -> u8 truncation edge


@pytest.mark.gpu
def test_producer_on_the_gpu_matches_the_numpy_restatement(tmp_path):
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "oracle"))
    import segmentation_oracle as O
    path = scripted(tmp_path)
    eng = InferenceEngine(path, 160, 120, device="cuda:0")
    net = _toy_numpy(path)
    rgb = np.random.default_rng(12).integers(0, 256, (120, 160, 3), dtype=np.uint8)
    ht, lt = eng.infer_one(rgb)
    assert ht.is_cuda
    want = O.infer_one(rgb, 160, 120, net)
    assert np.max(np.abs(ht.cpu().numpy() - want[0])) < 5e-6
    assert np.max(np.abs(lt.cpu().numpy() - want[1])) < 5e-6


@pytest.mark.gpu
def test_device_maps_integrate_like_host_maps(tmp_path, make_engine):
    from parity import assert_maps_equal
    from ratsdf import synthetic
    frames = synthetic.stream("room", 3, scale=0.25)
    h, w = frames[0]["depth"].shape
    seg = InferenceEngine(scripted(tmp_path), w, h, device="cuda:0")
    a, b = make_engine(0.02, 0.12), make_engine(0.02, 0.12)
    dev = torch.device("cuda", 0)
    for f in frames:
        ht, lt = seg.infer_one(f["rgb"])                 # stay on the device
        torch.cuda.synchronize()
        rgb, depth = torch.from_numpy(f["rgb"]).to(dev), torch.from_numpy(f["depth"]).to(dev)
        a.integrate_device(rgb.data_ptr(), depth.data_ptr(), ht.data_ptr(), lt.data_ptr(), h, w, 4.0,
                           f["intrinsics"], f["pose"])
        a.synchronize()
        b.integrate(f["rgb"], f["depth"], ht.cpu().numpy(), lt.cpu().numpy(), 4.0, f["intrinsics"],
                    f["pose"])
    assert_maps_equal(a, b, tol=0.0)
