"""What a plain device fill and a plain device copy reach on this box, at the sizes of K1's output (the stage-blocked QP of
B = 8 192 / 65 536 instances, N = 30: 27 fields x B x 32 doubles) - the practical write roofline K1's figures are read against:
    /usr/local/graft/bin/gpurun --timeout 300 -- 'python profiles/fill_rate.py'
(torch is used for the buffers and the events only.)"""
import numpy as np
import torch

dev = torch.device("cuda:0")
for B in (8192, 65536):
    n = 27 * B * 32
    a = torch.empty(n, dtype=torch.float64, device=dev)
    b = torch.empty(n, dtype=torch.float64, device=dev)
    big = torch.empty(512 * 1024 * 1024 // 8, dtype=torch.float64, device=dev)      # evicts the 256 MB cache between samples
    for name, fn, byts in (("fill", lambda: a.zero_(), 8 * n), ("copy", lambda: b.copy_(a), 16 * n)):
        for cold in (False, True):
            ts = []
            for _ in range(40):
                if cold:
                    big.add_(1.0)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); fn(); e1.record(); e1.synchronize()
                ts.append(e0.elapsed_time(e1))
            ts = np.array(ts[5:])
            print("B %6d %s (%4.0f MB%s): min %.4f med %.4f ms -> %.3f / %.3f of 8 TB/s" % (
                B, name, byts / 1e6, ", cache evicted before each" if cold else "", ts.min(), np.median(ts), byts / ts.min() / 1e-3 / 8e12,
                byts / np.median(ts) / 1e-3 / 8e12))
