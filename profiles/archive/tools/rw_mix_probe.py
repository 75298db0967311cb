#!/usr/bin/env python3
"""What the part gives two perfectly linear 16 B-per-lane streams (tools/stream_mix.hip) as a function of the read : write
mix -- from write-only to read-mostly -- at a fixed number of bytes per point: the constitutive kernels are write-heavy
(J2 1 : 3.8, elastic 1 : 7, FeFp 1 : 6.4), and the 8 TB/s of the data sheet is not what a write-heavy stream sees."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    import torch

    n = 10_000_000 // 64 * 64
    dev = torch.device("cuda:0")
    lib = C.CDLL(os.path.join(ROOT, "tools", "libstreammix.so"))
    lib.stream_mix_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.stream_mix_nt_launch.argtypes = lib.stream_mix_launch.argtypes
    st = torch.cuda.current_stream().cuda_stream
    total = 960
    rbuf = torch.randn(n * total // 8, dtype=torch.float64, device=dev)
    wbuf = torch.empty(n * total // 8, dtype=torch.float64, device=dev)
    mixes = [(0, 960), (64, 896), (96, 864), (128, 832), (192, 768), (320, 640), (480, 480), (640, 320), (832, 128), (944, 16)]
    res = []
    for rb, wb in mixes:
        best = {}
        for name, fn in (("plain", lib.stream_mix_launch), ("nt", lib.stream_mix_nt_launch)):
            for blocks in (1024, 2048, 4096):
                ts = []
                for r in range(10):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    fn(rbuf.data_ptr(), wbuf.data_ptr(), n, rb, wb, blocks, st or None)
                    e1.record()
                    torch.cuda.synchronize()
                    if r >= 2:
                        ts.append(e0.elapsed_time(e1))
                t = float(np.median(ts))
                best[name] = min(best.get(name, 1e9), t)
        ms = min(best.values())
        res.append({"read_B": rb, "write_B": wb, "ms": round(ms, 4), "ms_plain": round(best["plain"], 4), "ms_nt_stores": round(best["nt"], 4),
                    "GBs_total": round(total * n / ms / 1e6, 1), "GBs_written": round(wb * n / ms / 1e6, 1), "GBs_read": round(rb * n / ms / 1e6, 1)})
        print(json.dumps(res[-1]), flush=True)


if __name__ == "__main__":
    main()
