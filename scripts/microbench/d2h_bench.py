#!/usr/bin/env python3
"""How should a finished frame reach pageable host memory (HdrImage::pixels)?  Times, for a 1280x960 and a 1920x1080 float3 frame:
device -> pageable, hipHostRegister + copy + hipHostUnregister, device -> pinned staging + memcpy, device -> pinned only."""
import ctypes as C
import time

import numpy as np
import torch

rt = torch.cuda.cudart()
for w, h in ((1280, 960), (1920, 1080)):
    n = w * h * 3
    dev = torch.rand(n, device="cuda")
    host = torch.zeros(n, dtype=torch.float32)
    pinned = torch.empty(n, dtype=torch.float32).pin_memory()
    torch.cuda.synchronize()
    res = {}
    for name in ("pageable", "pinned+memcpy", "pinned only"):  # hipHostRegister of user memory is refused on the box (memlock limit)
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            if name == "pageable":
                host.copy_(dev)
            elif name.startswith("register"):
                err = rt.cudaHostRegister(host.data_ptr(), 4 * n, 0)
                if int(err) != 0:  # page-locking user memory can be refused (RLIMIT_MEMLOCK): not something to build on
                    best = float("nan")
                    break
                if name == "register":
                    host.copy_(dev)
                rt.cudaHostUnregister(host.data_ptr())
            elif name == "pinned+memcpy":
                pinned.copy_(dev)
                C.memmove(host.data_ptr(), pinned.data_ptr(), 4 * n)
            else:
                pinned.copy_(dev)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        res[name] = round(best * 1e3, 3)
    print(f"{w}x{h} ({4 * n / 1e6:.1f} MB): ms", res)
