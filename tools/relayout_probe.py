"""Times opus_msmv_relayout and its inverse per pyramid level (B=1, T=8, N=6, G=4, C=64; levels of a
640 x 1600 input at strides 8..64, 1.5 x the maps of the benchmark's 256 x 704 input): µs and TB/s of read + written bytes per launch."""
import ctypes
import sys

import torch

sys.path.insert(0, ".")
from opus_amd import _lib  # noqa: E402


def main():
    lib = _lib.load()
    B, T, N, G, C = 1, 8, 6, 4, 64
    dev = torch.device("cuda:0")
    tot = {0: 0.0, 1: 0.0}
    for H, W in ((80, 200), (40, 100), (20, 50), (10, 25)):
        chw = torch.randn(B, T * N, G * C, H, W, device=dev)
        hwc = torch.empty(B * T * G, N, H, W, C, device=dev)
        nbytes = 2 * chw.numel() * 4
        for inverse in (0, 1):
            src, dst = (chw, hwc) if not inverse else (hwc, chw)
            st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

            def run():
                _lib.check(lib.opus_msmv_relayout(ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(dst.data_ptr()),
                                                  B, T, N, G, C, H, W, inverse, st), "relayout")
            for _ in range(5):
                run()
            ts = []
            for _ in range(30):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); run(); e1.record(); e1.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            ts.sort()
            us = ts[len(ts) // 2]
            tot[inverse] += us
            print(f"{H}x{W} inverse={inverse}: {us:7.1f} us  {nbytes / us / 1e6:5.2f} TB/s", flush=True)
    print(f"sum forward {tot[0]:.1f} us, inverse {tot[1]:.1f} us", flush=True)


if __name__ == "__main__":
    main()
