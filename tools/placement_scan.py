#!/usr/bin/env python3
"""EXPERIMENT: class of each 1 GiB chunk of one big allocation, measured by the copy rate from two
reference chunks (chunk 0 and the chunk with the lowest rate from chunk 0)."""
import json
import sys

import numpy as np


def main():
    import torch

    dev = torch.device("cuda:0")
    gib = int(sys.argv[1]) if len(sys.argv) > 1 else 48
    chunk = (1 << 30) // 8
    big = torch.zeros(gib * chunk, dtype=torch.float64, device=dev)

    def tm(fn, reps=6):
        for _ in range(2):
            fn()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for e0, e1 in ev:
            e0.record()
            fn()
            e1.record()
        torch.cuda.synchronize()
        return float(np.median([e0.elapsed_time(e1) for e0, e1 in ev]))

    def scan(ref):
        src = big[ref * chunk:(ref + 1) * chunk]
        out = []
        for k in range(gib):
            dst = big[k * chunk:(k + 1) * chunk]
            if k == ref:
                out.append(0)
                continue
            ms = tm(lambda: dst.copy_(src))
            out.append(round(2 * chunk * 8 / ms / 1e6))
        return out

    print(json.dumps({"base": hex(big.data_ptr()), "GiB": gib}))
    a = scan(0)
    print(json.dumps({"ref_chunk": 0, "GBs": a}), flush=True)
    lo = int(np.argmin([x if x else 10**9 for x in a]))
    print(json.dumps({"ref_chunk": lo, "GBs": scan(lo)}), flush=True)


if __name__ == "__main__":
    main()
