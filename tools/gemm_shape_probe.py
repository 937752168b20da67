"""One int4 GEMM shape through vra_wna16_gemm under the dispatch's choice and with kernel D forced (VRA_GD_MB=2|4 in a child process):
    python tools/gemm_shape_probe.py K N gptq|awq[+dual] M ...        e.g. 3584 3584 awq 128 256 512   (Qwen2-7B's o_proj)
+dual: the gate/up pair with SiLU*mul (vra_wna16_gate_up_silu; N = intermediate size).  VRA_GEMV_W_MAX_ROWS=32 keeps kernel W to decode steps."""
import os
import subprocess
import sys

if os.environ.get("_PROBE_CHILD"):
    sys.path.insert(0, os.getcwd())
    from vllm_rs_amd import ops
    L = ops.lib()
    K, N, awq, dual = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3].startswith("awq"), sys.argv[3].endswith("+dual")
    Ms = [int(v) for v in sys.argv[4:]]
    e0, e1 = L.vra_event_create(), L.vra_event_create()
    w, sc, qz = ops.DevBuf(K * N // 2), ops.DevBuf(K // 128 * N * 2), ops.DevBuf(K // 128 * N // 8 * 4)
    w2 = ops.DevBuf(K * N // 2)
    L.vra_fill_hash_u32(w2.ptr, K * N // 8, 7, 0)
    L.vra_fill_hash_u32(w.ptr, K * N // 8, 1, 0)
    L.vra_fill_hash_u32(qz.ptr, K // 128 * N // 8, 2, 0)
    L.vra_fill_uniform(sc.ptr, K // 128 * N, 2, 0.002, 0.02, 0, 0)
    out = {}
    for M in Ms:
        x, res, o = ops.DevBuf(M * K * 2), ops.DevBuf(M * N * 2), ops.DevBuf(M * N * 2)
        L.vra_fill_normal(x.ptr, M * K, 3, 0.0, 1.0, 0, 0)
        L.vra_fill_normal(res.ptr, M * N, 4, 0.0, 1.0, 0, 0)

        def run():
            if dual:
                L.vra_wna16_gate_up_silu(x.ptr, w.ptr, sc.ptr, qz.ptr if awq else None, w2.ptr, sc.ptr, qz.ptr if awq else None, o.ptr, M, K, N, 128, int(awq), 0, 0, 0)
            else:
                L.vra_wna16_gemm(x.ptr, w.ptr, sc.ptr, qz.ptr if awq else None, None, res.ptr, o.ptr, M, K, N, 128, int(awq), 0, 0, 0)
        for _ in range(3):
            run()
        ops.check_error()
        L.vra_device_sync()
        L.vra_event_record(e0, 0)
        for _ in range(20):
            run()
        L.vra_event_record(e1, 0)
        us = L.vra_event_elapsed_ms(e0, e1) / 20 * 1e3
        out[M] = f"{us:7.1f} us {2.0 * M * K * N * (2 if dual else 1) / us / 1e6:6.1f} TF"
    print(os.environ.get("_PROBE_TAG"), out, flush=True)
    sys.exit(0)

NOW = {"VRA_GEMV_W_MAX_ROWS": "32"}  # kernel W out of the way for the forced variants
for tag, env in (("dispatch", {}), ("kernel D mb 2", dict(NOW, VRA_GD_MB="2")), ("kernel D mb 4", dict(NOW, VRA_GD_MB="4")), ("kernel B", dict(NOW, VRA_NO_KERNEL_D="1"))):
    e = dict(os.environ, _PROBE_CHILD="1", _PROBE_TAG=tag, **env)
    subprocess.run([sys.executable, __file__] + sys.argv[1:], env=e)
