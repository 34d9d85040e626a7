"""Times K2 (k_snow_threshold) alone: python tools/bench_k2.py [rows]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rsplash_amd.api as api
H = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
W, nt = 4096, 365
n = H * W
dev = torch.device("cuda:0")
tc = (torch.rand(nt, n, dtype=torch.float64, device=dev) - 0.5) * 40
par = torch.rand(api.NPAR, n, dtype=torch.float64, device=dev) * 50
Tt = torch.empty(n, dtype=torch.float64, device=dev)
s = torch.cuda.current_stream().cuda_stream
for _ in range(2):
    api.dev_snow_threshold(n, nt, tc, n, par, n, Tt, s)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    api.dev_snow_threshold(n, nt, tc, n, par, n, Tt, s)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print(f"k_snow_threshold {H}x{W}x{nt}: {ms:.3f} ms  {n*nt*8/ms/1e6:.1f} GB/s  checksum {float(Tt.sum()):.6e}")
