#!/usr/bin/env python
"""Time the deformable-convolution operator (BASELINE config 4) at the shape SURVEY 8d names: Z folded into the batch,
2-D DCNv1 3x3 on [B*16, C, 64, 160], deformable_groups 4, im2col_step 64 -- forward and backward through the C ABI
(rt_pose_amd/dcn.py), plus torch's conv2d on the same shape for scale.  Prints per-kernel effective bandwidth."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def t(f, it=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(it):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / it * 1e3


def main():
    from rt_pose_amd.dcn import deform_conv
    b, c, h, w, co = 128, int(sys.argv[1]) if len(sys.argv) > 1 else 32, 64, 160, 32
    x = torch.randn(b, c, h, w, device="cuda", requires_grad=True)
    off_scale = float(os.environ.get("RTP_BENCH_DCN_OFF_SCALE", "0.5"))   # pixels (1 sigma); 0.2: every sample within a pixel
    off = (torch.randn(b, 4 * 18, h, w, device="cuda") * off_scale).requires_grad_(True)
    wt = (torch.randn(co, c, 3, 3, device="cuda") * 0.05).requires_grad_(True)
    fwd = lambda: deform_conv(x, off, wt, 1, 1, 1, 1, 4, 64)
    ms_f = t(fwd)
    y = fwd()
    if "fwd" in sys.argv:
        print("forward %.3f ms" % ms_f)
        return
    g = torch.randn_like(y)

    def fb():
        for v in (x, off, wt):
            v.grad = None
        deform_conv(x, off, wt, 1, 1, 1, 1, 4, 64).backward(g)
    ms_fb = t(fb)
    ref = t(lambda: torch.nn.functional.conv2d(x, wt, None, 1, 1))
    alg = (x.numel() + off.numel() + y.numel()) * 4
    print("DCNv1 3x3 [%d,%d,%d,%d] -> %d, dg=4, offsets %.2f px: forward %.2f ms, forward+backward %.2f ms; torch conv2d forward %.2f ms" % (b, c, h, w, co, off_scale, ms_f, ms_fb, ref))
    print("  forward algorithmic bytes (input %.0f MB + offsets %.0f MB + output %.0f MB) -> %.0f GB/s; %.1f TFLOP/s fp32; "
          "%.1f G corner samples/s" % (x.numel() * 4 / 1e6, off.numel() * 4 / 1e6, y.numel() * 4 / 1e6, alg / ms_f / 1e6,
                                     2.0 * b * h * w * co * c * 9 / ms_f / 1e9, 4.0 * b * h * w * c * 9 / ms_f / 1e6))


if __name__ == "__main__":
    main()
