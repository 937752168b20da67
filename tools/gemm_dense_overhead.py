import os, sys
sys.path.insert(0, os.getcwd())
from vllm_rs_amd import ops
L = ops.lib()
e0, e1 = L.vra_event_create(), L.vra_event_create()
M, N = 4096, 4096
for K in (128, 256, 512, 1024, 2048, 4096):
    x = ops.DevBuf(M * K * 2); L.vra_fill_normal(x.ptr, M * K, 3, 0.0, 1.0, 0, 0)
    wd = ops.DevBuf(K * N * 2); L.vra_fill_normal(wd.ptr, K * N, 4, 0.0, 0.02, 0, 0)
    res = ops.DevBuf(M * N * 2); L.vra_fill_normal(res.ptr, M * N, 5, 0.0, 1.0, 0, 0)
    out = ops.DevBuf(M * N * 2)
    for use_res in (0, 1):
        def run():
            L.vra_dense_frag_gemm(x.ptr, wd.ptr, None, res.ptr if use_res else None, out.ptr, M, K, N, 0, 0, 256, 0)
        for _ in range(3): run()
        L.vra_device_sync(); L.vra_event_record(e0, 0)
        for _ in range(20): run()
        L.vra_event_record(e1, 0)
        ms = L.vra_event_elapsed_ms(e0, e1) / 20
        print(f"K={K:5d} residual={use_res}: {ms*1e3:7.1f} us  ({K//64} K-steps)", flush=True)
