"""Large-M int4 GEMM microbenchmark (kernel D vs kernel B): python tools/gemm_big_microbench.py [M ...]
VRA_NO_KERNEL_D=1 routes everything to kernel B; VRA_GD_MB=2|4 forces the m-tiles per wave of kernel D."""
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
from vllm_rs_amd import ops

L = ops.lib()
Ms = [int(v) for v in sys.argv[1:]] or [128, 512, 4096]
SHAPES = {"gate_up": (4096, 14336), "o": (4096, 4096), "down": (14336, 4096)}
e0, e1 = L.vra_event_create(), L.vra_event_create()
for M in Ms:
    for which, (K, N) in SHAPES.items():
        nl = 3
        ws = [ops.DevBuf(K * N // 2) for _ in range(2 * nl)]
        sc = [ops.DevBuf(K // 128 * N * 2) for _ in range(2 * nl)]
        for w in ws:
            L.vra_fill_hash_u32(w.ptr, K * N // 8, 1, 0)
        for s in sc:
            L.vra_fill_uniform(s.ptr, K // 128 * N, 2, 0.002, 0.02, 0, 0)
        x = ops.DevBuf(M * K * 2)
        L.vra_fill_normal(x.ptr, M * K, 3, 0.0, 1.0, 0, 0)
        out = ops.DevBuf(M * N * 2)

        def run(i):
            if which == "gate_up":
                L.vra_wna16_gate_up_silu(x.ptr, ws[2 * i].ptr, sc[2 * i].ptr, None, ws[2 * i + 1].ptr, sc[2 * i + 1].ptr, None, out.ptr, M, K, N, 128, 0, 0, 0, 0)
            else:
                L.vra_wna16_gemm(x.ptr, ws[i].ptr, sc[i].ptr, None, None, None, out.ptr, M, K, N, 128, 0, 0, 0, 0)

        for i in range(3):
            run(i % nl)
        L.vra_device_sync()
        reps = 20 if M <= 512 else 6
        L.vra_event_record(e0, 0)
        for i in range(reps):
            run(i % nl)
        L.vra_event_record(e1, 0)
        ms = L.vra_event_elapsed_ms(e0, e1) / reps
        flops = 2.0 * M * K * N * (2 if which == "gate_up" else 1)
        print(f"M={M:5d} {which:8s} {ms * 1e3:9.1f} us  {flops / ms / 1e9:8.1f} TFLOP/s", flush=True)
        del ws, sc, x, out
