"""us per call of vra_dense_gemm at the Llama-3 lm_head shape (f32 logits) by row count; VRA_NO_GEMV_DW=1 for kernel B"""
import os, sys
sys.path.insert(0, os.getcwd())
from vllm_rs_amd import ops
L = ops.lib()
K, N = 4096, 128256
w = ops.DevBuf(N * K * 2); L.vra_fill_normal(w.ptr, N * K, 1, 0.0, 0.02, 0, 0)
for M in [int(a) for a in sys.argv[1:]] or [8, 9, 16, 17, 32]:
    x = ops.DevBuf(M * K * 2); L.vra_fill_normal(x.ptr, M * K, 3, 0.0, 1.0, 0, 0)
    out = ops.DevBuf(M * N * 4)
    run = lambda: L.vra_dense_gemm(x.ptr, w.ptr, None, out.ptr, M, K, N, 0, 2, 0)
    e0, e1 = L.vra_event_create(), L.vra_event_create()
    for _ in range(3): run()
    L.vra_device_sync(); L.vra_event_record(e0, 0)
    for _ in range(20): run()
    L.vra_event_record(e1, 0)
    us = L.vra_event_elapsed_ms(e0, e1) * 50
    print(f"M {M:2d}: {us:7.1f} us  {N * K * 2 / us / 1e6:5.2f} TB/s")
