#!/usr/bin/env python3
"""What a plain device-to-device copy / fill of the rebalances' byte counts achieves on this GPU (torch's copy kernel): the
practical ceiling the rebalance passes are held against in DESIGN §6 / §9.2.  usage: python tools/copy_rate.py"""
import torch

for mb in (201, 402, 805):
    n = mb * 1000 * 1000 // 4
    a = torch.empty(n, dtype=torch.int32, device="cuda").random_()
    b = torch.empty_like(a)
    for _ in range(3):
        b.copy_(a)
        b.zero_()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        b.copy_(a)
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) / 20 * 1e3
    print(f"D2D copy of {mb} MB: {us:.1f} us = {2 * mb / us:.2f} TB/s read + write")
    s.record()
    for _ in range(20):
        b.zero_()
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) / 20 * 1e3
    print(f"fill of {mb} MB: {us:.1f} us = {mb / us:.2f} TB/s write")
