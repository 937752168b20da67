"""Phase cycle sums of the LDS-tiled prefill attention kernel (needs a -DVRA_GEMV_TS build: make B=build_ts EXTRA=-DVRA_GEMV_TS
OUT=../libvra_ts.so; run with VRA_LIB=.../libvra_ts.so).  Per wave and KV tile: barrier 1 | K/V registers -> LDS | barrier 2 |
next tile's loads issued | S = K.Q^T MFMAs | softmax VALU | O += V.P MFMAs."""
import ctypes
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np

from vllm_rs_amd import ops

Hq, Hkv, D, BS = 32, 8, 128, 64
att = ops.PagedAttention(Hq, D, D ** -0.5, Hkv, BS, ops.BF16)
L = ops.lib()
T = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
nb = (T + BS - 1) // BS
r = np.random.default_rng(0)
q = ops.dev((r.standard_normal((T, Hq, D)).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16))
kc = ops.dev((r.standard_normal((nb * Hkv * BS * D)).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16))
vc = ops.dev((r.standard_normal((nb * Hkv * BS * D)).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16))
bt = ops.dev(r.permutation(nb).astype(np.uint32))
cl = ops.dev(np.array([T], np.uint32))
cu = ops.dev(np.array([0, T], np.uint32))
for _ in range(3):
    o = att.forward_prefill(q, T, T, cu, 1, k_cache=kc, v_cache=vc, block_tables=bt, context_lens=cl, max_blocks=nb)
L.vra_device_sync()
n = 4096 * 32
buf = (ctypes.c_ulonglong * n)()
L.vra_debug_attn_pf_ts.argtypes = [ctypes.c_void_p, ctypes.c_int]
L.vra_debug_attn_pf_ts(buf, n)
t = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 4, 8).astype(np.float64)
g = int((t[:, 0, 7] != 0).sum())
t = t[:g]
nt = t[:, :, 7:8]
per = t[:, :, :7] / nt
names = ["barrier 1", "regs -> LDS", "barrier 2", "issue loads", "S mfma", "softmax", "PV mfma"]
print(f"T={T}: {g} workgroups; mean cycles per KV tile (all waves), and for the workgroups with the most tiles")
big = t[:, 0, 7] >= np.percentile(t[:, 0, 7], 90)
for i, nm in enumerate(names):
    print(f"  {nm:12s} {per[:, :, i].mean():8.0f}   heavy {per[big][:, :, i].mean():8.0f}")
print(f"  total        {per.sum(axis=2).mean():8.0f}   heavy {per[big].sum(axis=2).mean():8.0f}")
