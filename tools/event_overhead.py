#!/usr/bin/env python3
"""What a pair of HIP events adds to the launch it brackets: elapsed time of event pairs with NOTHING between them, and around
a kernel that does nothing, on an idle GPU."""
import torch
dev = torch.device("cuda:0")
s = torch.cuda.Stream(device=dev)
x = torch.zeros(1, device=dev)
torch.cuda.synchronize()
for what in ("nothing", "tiny kernel"):
    vals = []
    for _ in range(200):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(s):
            a.record()
            if what != "nothing":
                x.add_(1.0)
            b.record()
        torch.cuda.synchronize()
        vals.append(a.elapsed_time(b) * 1e3)
    vals = sorted(vals[20:])
    print("%-12s us: min %.2f  median %.2f  p90 %.2f" % (what, vals[0], vals[len(vals) // 2], vals[len(vals) * 9 // 10]))
