"""Per-workgroup timeline of decode_attn_fused_kernel (needs a -DVRA_ATTN_TS build:
make B=build_ats EXTRA=-DVRA_ATTN_TS OUT=.../libvra_ats.so; run with VRA_LIB=.../libvra_ats.so).
usage: attn_ts.py [batch] [ctx]"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from vllm_rs_amd import ops
L = ops.lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ctx = int(sys.argv[2]) if len(sys.argv) > 2 else 150
Hq, Hkv, D, BS = 32, 8, 128, 64
nblk = (ctx + BS - 1) // BS
NB = B * nblk + 4
pa = ops.PagedAttention(Hq, D, 1.0 / np.sqrt(D), Hkv, BS)
kc = ops.DevBuf(NB * Hkv * BS * D * 2); vc = ops.DevBuf(NB * Hkv * BS * D * 2)
L.vra_fill_normal(kc.ptr, NB * Hkv * BS * D, 1, 0.0, 1.0, 0, 0); L.vra_fill_normal(vc.ptr, NB * Hkv * BS * D, 2, 0.0, 1.0, 0, 0)
q = ops.DevBuf(B * Hq * D * 2); k = ops.DevBuf(B * Hkv * D * 2); v = ops.DevBuf(B * Hkv * D * 2)
for buf, n, sd in ((q, B * Hq * D, 3), (k, B * Hkv * D, 4), (v, B * Hkv * D, 5)):
    L.vra_fill_normal(buf.ptr, n, sd, 0.0, 1.0, 0, 0)
cos = ops.DevBuf(8192 * 64 * 2); sin = ops.DevBuf(8192 * 64 * 2)
L.vra_fill_uniform(cos.ptr, 8192 * 64, 6, -1.0, 1.0, 0, 0); L.vra_fill_uniform(sin.ptr, 8192 * 64, 7, -1.0, 1.0, 0, 0)
bt = np.arange(B * nblk, dtype=np.uint32).reshape(B, nblk)
pos = np.full(B, ctx - 1, np.int64); slots = np.array([int(bt[b, (ctx - 1) // BS]) * BS + (ctx - 1) % BS for b in range(B)], np.int64)
cl = np.full(B, ctx, np.uint32)
d = [ops.dev(x) for x in (pos, slots, bt, cl)]
ws = ops.DevBuf(L.vra_paged_attention_decode_workspace_bytes(B, Hq, D, ctx))
e0, e1 = L.vra_event_create(), L.vra_event_create()
run = lambda: pa.rope_cache_decode(q, k, v, kc, vc, cos, sin, d[0], d[1], d[2], d[3], B, nblk, ctx, ws)
for _ in range(5): run()
L.vra_device_sync(); L.vra_event_record(e0, 0)
for _ in range(50): run()
L.vra_event_record(e1, 0); print("avg us per call (incl. merge if split)", L.vra_event_elapsed_ms(e0, e1) * 20)
n = 4096 * 16
buf = (ctypes.c_ulonglong * n)()
L.vra_debug_attn_ts.argtypes = [ctypes.c_void_p, ctypes.c_int]; L.vra_debug_attn_ts(buf, n)
t = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 16).astype(np.int64)
g = int((t[:, 0] != 0).sum()); t = t[:g]; base = t[:, 0].min()
names = ["start", "ctx/pos/slot known", "new k/v staged (thread 0)", "q fragments", "barrier passed", "K/V loads issued (first tile)", "QK done", "softmax done", "PV done (loop end)", "lds_o written", "merge barrier passed", "end"]
print("workgroups", g)
for i, nm in enumerate(names):
    c = t[:, [3, 4, 5, 6, 7, 8, 9, 10, 11][i - 3]] if False else t[:, i if i < 3 else i]
    ok = c != 0
    if ok.sum() == 0: continue
    r = (c[ok] - base) / 100.0
    print(f"{i:2d} {nm:32s} n={ok.sum():4d} min {r.min():6.2f} p50 {np.median(r):6.2f} max {r.max():6.2f}")
