#!/usr/bin/env python3
"""EXPERIMENT: is there a class structure in device memory?  Copy 2 GiB from allocation i to
allocation j for all pairs (torch copy_, fp64) and print the matrix of effective GB/s (read+write)."""
import json
import sys

import numpy as np


def main():
    import torch

    dev = torch.device("cuda:0")
    npools = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    gib = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    pools = [torch.zeros((gib << 30) // 8, dtype=torch.float64, device=dev) for _ in range(npools)]
    half = pools[0].numel() // 2

    def tm(fn, reps=8):
        for _ in range(2):
            fn()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for e0, e1 in ev:
            e0.record()
            fn()
            e1.record()
        torch.cuda.synchronize()
        return float(np.median([e0.elapsed_time(e1) for e0, e1 in ev]))

    print(json.dumps({"bases": [hex(p.data_ptr()) for p in pools]}))
    for i in range(npools):
        row = []
        for j in range(npools):
            src = pools[i][:half]
            dst = pools[j][half:]
            ms = tm(lambda: dst.copy_(src))
            row.append(round(2 * half * 8 / ms / 1e6))
        print(json.dumps({"src": i, "GBs_to_dst": row}), flush=True)


if __name__ == "__main__":
    main()
