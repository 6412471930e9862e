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
