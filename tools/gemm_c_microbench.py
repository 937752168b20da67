"""time vra_wna16_gemm at M rows for the four Llama-3-8B projection shapes (kernel selection as in production)"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.getcwd())
from vllm_rs_amd import ops
L = ops.lib()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 32
shapes = [("o", 4096, 4096), ("down", 14336, 4096), ("qkv_q", 4096, 6144)]
for name, K, N in shapes:
    nlayer = 6
    ws = [ops.DevBuf(K * N // 2) for _ in range(nlayer)]
    sc = [ops.DevBuf(K // 128 * N * 2) for _ in range(nlayer)]
    for w in ws: L.vra_fill_hash_u32(w.ptr, K * N // 8, 1, 0)
    for s in sc: L.vra_fill_uniform(s.ptr, K // 128 * N, 2, 0.002, 0.02, 0, 0)
    x = ops.DevBuf(M * K * 2); L.vra_fill_normal(x.ptr, M * K, 3, 0.0, 1.0, 0, 0)
    out = ops.DevBuf(M * N * 2)
    e0, e1 = L.vra_event_create(), L.vra_event_create()
    def run(n):
        for i in range(n):
            L.vra_wna16_gemm(x.ptr, ws[i % nlayer].ptr, sc[i % nlayer].ptr, None, None, None, out.ptr, M, K, N, 128, 0, 0, 0, 0)
    run(10); L.vra_device_sync()
    L.vra_event_record(e0, 0); run(100); L.vra_event_record(e1, 0)
    ms = L.vra_event_elapsed_ms(e0, e1)
    print(f"M={M} {name:6s} K={K} N={N}: {ms*10:.2f} us  ({(K*N/2)/(ms/100*1e-3)/1e9:.0f} GB/s)  err={ops.last_error() if hasattr(ops,'last_error') else ''}")
