"""us per lm_head + greedy argmax at the Llama-3 shape by row count: two launches (vra_dense_gemm, vra_argmax_f32) against the one
launch of vra_dense_gemm_argmax (dense W kernel with the last-arriver reduction, gemv_dw.cuh), alternating, HIP events"""
import os, sys
sys.path.insert(0, os.getcwd())
from vllm_rs_amd import ops
L = ops.lib()
K, N = 4096, 128256
w = ops.DevBuf(N * K * 2); L.vra_fill_normal(w.ptr, N * K, 1, 0.0, 0.02, 0, 0)
ws = ops.DevBuf(L.vra_dense_gemm_argmax_workspace_bytes()).zero()
for M in [int(a) for a in sys.argv[1:]] or [9, 16, 32]:
    x = ops.DevBuf(M * K * 2); L.vra_fill_normal(x.ptr, M * K, 3, 0.0, 1.0, 0, 0)
    out, tok = ops.DevBuf(M * N * 4), ops.DevBuf(M * 4)
    def two():
        L.vra_dense_gemm(x.ptr, w.ptr, None, out.ptr, M, K, N, 0, 2, 0)
        L.vra_argmax_f32(out.ptr, tok.ptr, M, N, 0)
    def one():
        L.vra_dense_gemm_argmax(x.ptr, w.ptr, None, out.ptr, tok.ptr, ws.ptr, M, K, N, 0, 0)
    def t(run, n=40):
        e0, e1 = L.vra_event_create(), L.vra_event_create()
        for _ in range(3): run()
        L.vra_device_sync(); L.vra_event_record(e0, 0)
        for _ in range(n): run()
        L.vra_event_record(e1, 0)
        return L.vra_event_elapsed_ms(e0, e1) * 1e3 / n
    r = [(t(two), t(one)) for _ in range(3)]
    print(f"M {M:2d}: two launches " + " ".join(f"{a:6.1f}" for a, _ in r) + "  one launch " + " ".join(f"{b:6.1f}" for _, b in r) + " us")
