"""PCIe device-to-host rate of this box for the CLI's copy-out: one 2 GB copy into pinned memory on one stream, the same bytes as
two / four concurrent copies on as many streams, and with a compute kernel running beside them.  Needs a GPU.  python tools/d2h_probe.py"""
import json
import time

import torch


def main():
    dev = torch.device('cuda', 0)
    n = 2 << 30
    src = torch.empty(n, dtype=torch.uint8, device=dev)
    dst = torch.empty(n, dtype=torch.uint8, pin_memory=True)
    res = {}
    for parts in (1, 2, 4):
        streams = [torch.cuda.Stream(device=dev) for _ in range(parts)]
        step = n // parts
        best = 0.0
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i, s in enumerate(streams):
                with torch.cuda.stream(s):
                    dst[i * step:(i + 1) * step].copy_(src[i * step:(i + 1) * step], non_blocking=True)
            for s in streams:
                s.synchronize()
            best = max(best, n / (time.perf_counter() - t0) / 1e9)
        res[f'{parts}_streams_GBps'] = round(best, 1)
    # the copy engine's own path, blocking (what round 4's consumer did)
    t0 = time.perf_counter()
    dst.copy_(src, non_blocking=False)
    res['blocking_GBps'] = round(n / (time.perf_counter() - t0) / 1e9, 1)
    print(json.dumps(res))


if __name__ == '__main__':
    main()
