"""ht / lt producer: the step in front of TSDFSystem::Integrate (segmentation/inference.cc:43-99).

The reference loads a TorchScript network, resizes the colour image to the next multiple of 32 in
both directions (``((int)(w / 32) + 1) * 32``, inference.cc:47-48), scales it to [0, 1], runs the
network (output: 2 x H' x W', high-touch / low-touch probability), copies the two maps to the host,
resizes them back with cv::resize and hands them to Integrate -- which uploads them again.  Here the
maps stay on the device: ``infer_one`` returns two float32 CUDA (ROCm) tensors of the frame's size
that go straight into ``ratsdf_integrate_device``.

Without a model path the reference returns all-ones images (inference.cc:63-68); so does this class
(as ``None, None``: the engine treats missing ht / lt as ones, modules/tsdf_module.cc:27-31).

cv::resize's default INTER_LINEAR samples at half-pixel centres without antialiasing, which is
``torch.nn.functional.interpolate(mode="bilinear", align_corners=False)``; results agree with
OpenCV's to float rounding (OpenCV is not available here: parity unpinned, see tests).
"""
import torch
import torch.nn.functional as F


class InferenceEngine:
    def __init__(self, compiled_engine_path, width, height, device=None):
        self.width, self.height = int(width), int(height)
        self.whole_width = (self.width // 32 + 1) * 32      # inference.cc:47
        self.whole_height = (self.height // 32 + 1) * 32    # inference.cc:48
        self.device = torch.device(device if device is not None
                                   else ("cuda" if torch.cuda.is_available() else "cpu"))
        self.running = bool(compiled_engine_path)
        self.engine = None
        if self.running:
            self.engine = torch.jit.load(str(compiled_engine_path), map_location=self.device).eval()

    @torch.no_grad()
    def infer_one(self, rgb):
        """rgb: H x W x 3 uint8 (numpy array or tensor, RGB).  Returns (ht, lt): float32 H x W tensors
        on ``self.device``, or (None, None) when no model is loaded."""
        if not self.running:
            return None, None
        x = torch.as_tensor(rgb)
        if x.shape[0] != self.height or x.shape[1] != self.width or x.shape[2] != 3:
            raise ValueError("image size does not match the engine's")
        x = x.to(self.device).permute(2, 0, 1).unsqueeze(0).to(torch.float32)      # 1 x 3 x H x W
        x = F.interpolate(x, size=(self.whole_height, self.whole_width), mode="bilinear",
                          align_corners=False)                                      # inference.cc:74
        x = x * (1.0 / 255.0)                                                       # inference.cc:13
        y = self.engine(x).squeeze().detach()                                       # inference.cc:83-84
        if y.dim() != 3 or y.shape[0] < 2:
            raise RuntimeError("the network must return a 2 x H x W probability map")
        y = F.interpolate(y[:2].unsqueeze(0).to(torch.float32), size=(self.height, self.width),
                          mode="bilinear", align_corners=False)[0]                  # inference.cc:29
        return y[0].contiguous(), y[1].contiguous()
