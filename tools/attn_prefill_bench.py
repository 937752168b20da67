"""Prefill attention alone: one sequence of T tokens (Llama-3-8B heads: 32 q / 8 kv, D 128) over the paged cache, timed over
N launches; prints ms and TFLOP/s (4*D*causal pairs*Hq).  A/B: VRA_NO_PREFILL_TILED=1 selects the round-1 kernel."""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import numpy as np

from vllm_rs_amd import ops

Hq, Hkv, D, BS = 32, 8, 128, 64
fp8 = bool(int(os.environ.get("FP8", "0")))
att = ops.PagedAttention(Hq, D, D ** -0.5, Hkv, BS, ops.BF16, fp8_kvcache=fp8)
L = ops.lib()
for T in [int(a) for a in sys.argv[1:]] or [4096]:
    nb = (T + BS - 1) // BS
    r = np.random.default_rng(0)
    q = ops.dev((r.standard_normal((T, Hq, D)).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16))
    esz = 1 if fp8 else 2
    kc = ops.DevBuf(nb * Hkv * BS * D * esz).fill_bytes(0x3c if fp8 else 0x3c)
    vc = ops.DevBuf(nb * Hkv * BS * D * esz).fill_bytes(0x3c)
    bt = ops.dev(r.permutation(nb).astype(np.uint32))
    cl = ops.dev(np.array([T], np.uint32))
    cu = ops.dev(np.array([0, T], np.uint32))
    n = 20
    o = ops.DevBuf(T * Hq * D * 2)  # (one output buffer: forward_prefill allocates and frees one per call — a device sync per launch)
    if not fp8:  # random K / V (fp8: the constant fill stays)
        L.vra_fill_normal(kc.ptr, nb * Hkv * BS * D, 32, 0.0, 1.0, 0, 0)
        L.vra_fill_normal(vc.ptr, nb * Hkv * BS * D, 33, 0.0, 1.0, 0, 0)
    for it in range(2):
        L.vra_device_sync()
        t0 = time.perf_counter()
        for _ in range(n):
            L.vra_paged_attention_prefill_sw(o.ptr, q.ptr, None, None, kc.ptr, vc.ptr, bt.ptr, cl.ptr, cu.ptr, None, 1, T, T, Hq, Hkv, D, BS, nb, att.scale,
                                             0.0, 0, ops.BF16, att.kv_dtype, 0)
        L.vra_device_sync()
        ms = (time.perf_counter() - t0) / n * 1e3
    fl = 4.0 * D * Hq * (T * (T + 1) / 2)
    print(f"T={T} fp8={int(fp8)} tiled={0 if os.environ.get('VRA_NO_PREFILL_TILED') else 1}: {ms:.3f} ms  {fl / ms / 1e9:.1f} TFLOP/s")
