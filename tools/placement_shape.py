#!/usr/bin/env python3
"""Where does the bimodal kernel time (0.82 vs 0.94 ms at 1e7 points) come from?  Uses the J2-shaped
streaming probe (tools/stream_mix.hip: strain + 7 SoA slots in, stress + 7 SoA slots + tangent out)
with explicit state pointers:
  A. the same layout in K separate allocations          -> physical placement
  B. different offsets / slot strides inside ONE allocation -> virtual layout
"""
import ctypes as C
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    import torch

    n = 10_000_000 // 64 * 64
    dev = torch.device("cuda:0")
    lib = C.CDLL(os.path.join(ROOT, "tools", "libstreammix.so"))
    lib.stream_mix_j2_shape_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
    st = torch.cuda.current_stream().cuda_stream
    eps = torch.randn((n, 6), dtype=torch.float64, device=dev)
    sig = torch.empty((n, 6), dtype=torch.float64, device=dev)
    ct = torch.empty((n, 36), dtype=torch.float64, device=dev)
    blocks = 1024

    def run(s0_ptr, s1_ptr, ld, reps=12):
        for _ in range(2):
            lib.stream_mix_j2_shape_launch(eps.data_ptr(), s0_ptr, s1_ptr, ld, sig.data_ptr(), ct.data_ptr(), n, blocks, st or None)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for e0, e1 in ev:
            e0.record()
            lib.stream_mix_j2_shape_launch(eps.data_ptr(), s0_ptr, s1_ptr, ld, sig.data_ptr(), ct.data_ptr(), n, blocks, st or None)
            e1.record()
        torch.cuda.synchronize()
        return round(float(np.median([a.elapsed_time(b) for a, b in ev])), 4)

    ld = n + 32
    bufs = [torch.zeros(2 * 7 * ld + (64 << 17), dtype=torch.float64, device=dev) for _ in range(8)]   # + 64 MiB slack
    for k, b in enumerate(bufs):
        p = b.data_ptr()
        print(json.dumps({"part": "A", "alloc": k, "base": hex(p), "ms": run(p, p + 7 * ld * 8, ld), "ms_swapped": run(p + 7 * ld * 8, p, ld)}), flush=True)
    for k in (0, len(bufs) - 1):
        p = bufs[k].data_ptr()
        for off in (0, 256, 4096, 65536, 1 << 20, 2 << 20, 16 << 20, 48 << 20):
            print(json.dumps({"part": "B-offset", "alloc": k, "offset": off, "ms": run(p + off, p + off + 7 * ld * 8, ld)}), flush=True)
        for pad in (0, 16, 32, 64, 96, 128, 256, 512, 1024, 4096):
            l2 = n + pad
            print(json.dumps({"part": "B-ld", "alloc": k, "ld_pad_doubles": pad, "ms": run(p, p + 7 * l2 * 8, l2)}), flush=True)
        for gap in (0, 256, 1024, 4096, 1 << 20):
            print(json.dumps({"part": "B-s1gap", "alloc": k, "gap": gap, "ms": run(p, p + 7 * ld * 8 + gap, ld)}), flush=True)
    # state in one allocation vs s0 and s1 in different allocations
    print(json.dumps({"part": "C", "s0_alloc": 0, "s1_alloc": 1, "ms": run(bufs[0].data_ptr(), bufs[1].data_ptr(), ld)}), flush=True)
    print(json.dumps({"part": "C", "s0_alloc": 2, "s1_alloc": 3, "ms": run(bufs[2].data_ptr(), bufs[3].data_ptr(), ld)}), flush=True)


if __name__ == "__main__":
    main()
