"""3x3 convolution time, NCHW vs channels_last activations/weights, with MIOpen find mode (cudnn.benchmark)."""
import os
for k in ("FWD", "BWD", "WRW"):
    os.environ.setdefault("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_" + k, "0")
import torch
import torch.nn.functional as F
torch.backends.cudnn.benchmark = True
dev, dt = "cuda", torch.bfloat16
shapes = [(16, 320, 320, 64, 3), (16, 640, 320, 64, 3), (16, 960, 320, 64, 3), (16, 640, 640, 32, 3), (16, 1280, 640, 32, 3),
          (16, 1920, 640, 32, 3), (16, 1280, 1280, 16, 3), (16, 2560, 1280, 16, 3), (16, 1280, 1280, 8, 3), (16, 2560, 1280, 8, 3),
          (16, 320, 320, 32, 3), (16, 640, 320, 64, 1)]
tot = {False: 0.0, True: 0.0}
for (B, ci, co, hw, k) in shapes:
    row = []
    for cl in (False, True):
        x = torch.randn(B, ci, hw, hw, device=dev, dtype=dt)
        w = torch.randn(co, ci, k, k, device=dev, dtype=dt) * 0.02
        if cl:
            x = x.contiguous(memory_format=torch.channels_last)
            w = w.contiguous(memory_format=torch.channels_last)
        for _ in range(3):
            y = F.conv2d(x, w, None, 1, k // 2)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            y = F.conv2d(x, w, None, 1, k // 2)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        tot[cl] += us
        row.append(us)
    fl = 2.0 * B * hw * hw * ci * co * k * k
    print("B%d %4d->%4d @%2d k%d: NCHW %7.1f us (%4.0f TF)  NHWC %7.1f us (%4.0f TF)" % (B, ci, co, hw, k, row[0], fl / row[0] / 1e6, row[1], fl / row[1] / 1e6))
print("sum NCHW %.0f us, NHWC %.0f us" % (tot[False], tot[True]))
