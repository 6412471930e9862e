"""ht / lt producer: the step in front of TSDFSystem::Integrate (segmentation/inference.cc:43-99).

The reference loads a TorchScript network, resizes the colour image to the next multiple of 32 in
both directions (``((int)(w / 32) + 1) * 32``, inference.cc:47-48), scales it to [0, 1], runs the
network (output: 2 x H' x W', high-touch / low-touch probability), copies the two maps to the host,
resizes them back with cv::resize and hands them to Integrate -- which uploads them again.  Here the
maps stay on the device: ``infer_one`` returns two float32 CUDA (ROCm) tensors of the frame's size
that go straight into ``ratsdf_integrate_device``.

Without a model path the reference returns all-ones images (inference.cc:63-68); so does this class
(as ``None, None``: the engine treats missing ht / lt as ones, modules/tsdf_module.cc:27-31).

The two resizes are cv::resize with its default INTER_LINEAR:
* colour image, CV_8UC3 (inference.cc:74): an 8-bit result computed by OpenCV's fixed-point bilinear
  kernel (11-bit coefficients); ``resize_u8_linear`` restates it with integer tensor arithmetic on
  the device, so the network sees the very bytes the reference's network sees;
* probability maps, CV_32FC1 (inference.cc:29): half-pixel sample positions, no antialiasing, which is
  ``torch.nn.functional.interpolate(mode="bilinear", align_corners=False)`` up to the order of the
  float operations (<= 1e-6 on probabilities).
OpenCV is not available here: parity with the library itself is unpinned; tests compare with the numpy
restatement in oracle/segmentation_oracle.py.
"""
import torch
import torch.nn.functional as F

_COEF_SCALE = 2048.0   # INTER_RESIZE_COEF_SCALE, 11 bits


def _axis_table(src, dst, device):
    """first tap (clamped) and fixed-point weights of the two taps for every output position"""
    d = torch.arange(dst, dtype=torch.float64, device=device)
    f = ((d + 0.5) * (float(src) / float(dst)) - 0.5).to(torch.float32)
    s = torch.floor(f)
    f = f - s
    s = s.to(torch.int64)
    f = torch.where((s < 0) | (s >= src - 1), torch.zeros_like(f), f)
    s = s.clamp(0, src - 1)
    a1 = torch.round(f * _COEF_SCALE).to(torch.int32)                 # cvRound: half to even, like torch.round
    a0 = torch.round((1.0 - f) * _COEF_SCALE).to(torch.int32)
    return s, (s + 1).clamp(max=src - 1), a0, a1


def resize_u8_linear(img, out_h, out_w):
    """cv::resize(CV_8UC3 -> out_h x out_w), INTER_LINEAR, as OpenCV computes it for 8-bit images.
    img: H x W x C uint8 tensor (any device); returns out_h x out_w x C uint8."""
    h, w = img.shape[0], img.shape[1]
    dev = img.device
    x0, x1, a0, a1 = _axis_table(w, out_w, dev)
    y0, y1, b0, b1 = _axis_table(h, out_h, dev)
    src = img.to(torch.int32)
    hor = src[:, x0] * a0.view(1, -1, 1) + src[:, x1] * a1.view(1, -1, 1)
    r0, r1 = hor[y0], hor[y1]
    out = (((b0.view(-1, 1, 1) * (r0 >> 4)) >> 16) + ((b1.view(-1, 1, 1) * (r1 >> 4)) >> 16) + 2) >> 2
    return out.clamp(0, 255).to(torch.uint8)


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
    def infer_one(self, rgb, ret_uint8=False):
        """rgb: H x W x 3 uint8 (numpy array or tensor, RGB).  Returns (ht, lt): float32 H x W tensors
        on ``self.device``, or (None, None) when no model is loaded.  ret_uint8 (the reference's
        ret_uint8_flag, inference.cc:90-92): uint8 maps at the NETWORK's resolution instead."""
        if not self.running:
            return None, None
        x = torch.as_tensor(rgb)
        if x.shape[0] != self.height or x.shape[1] != self.width or x.shape[2] != 3:
            raise ValueError("image size does not match the engine's")
        x = resize_u8_linear(x.to(self.device), self.whole_height, self.whole_width)  # inference.cc:74
        x = x.to(torch.float32) * (1.0 / 255.0)                                     # inference.cc:13
        x = x.permute(2, 0, 1).unsqueeze(0).contiguous()                            # 1 x 3 x H' x W'
        y = self.engine(x).squeeze().detach()                                       # inference.cc:83-84
        if y.dim() != 3 or y.shape[0] < 2:
            raise RuntimeError("the network must return a 2 x H x W probability map")
        if ret_uint8:                                                               # inference.cc:33-41
            u = y[:2].mul(255).clamp(0, 255).to(torch.uint8)
            return u[0].contiguous(), u[1].contiguous()
        y = F.interpolate(y[:2].unsqueeze(0).to(torch.float32), size=(self.height, self.width),
                          mode="bilinear", align_corners=False)[0]                  # inference.cc:29
        return y[0].contiguous(), y[1].contiguous()
