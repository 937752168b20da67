"""Per-workgroup timeline of gemm_q4_kernel (kernel C); needs the -DVRA_GEMV_TS build (VRA_LIB=.../libvra_ts.so).
usage: gemm_c_ts.py M which   (which: gate_up | o | down | qkv)"""
import sys, os, ctypes
import numpy as np
sys.path.insert(0, os.getcwd())
from vllm_rs_amd import ops
L = ops.lib()
M = int(sys.argv[1]); which = sys.argv[2]
K, N = {"gate_up": (4096, 14336), "o": (4096, 4096), "down": (14336, 4096), "qkv": (4096, 6144)}[which]
nl = 4
ws = [ops.DevBuf(K * N // 2) for _ in range(2 * nl)]; sc = [ops.DevBuf(K // 128 * N * 2) for _ in range(2 * nl)]
for w in ws: L.vra_fill_hash_u32(w.ptr, K * N // 8, 1, 0)
for s in sc: L.vra_fill_uniform(s.ptr, K // 128 * N, 2, 0.002, 0.02, 0, 0)
x = ops.DevBuf(M * K * 2); L.vra_fill_normal(x.ptr, M * K, 3, 0.0, 1.0, 0, 0)
out = ops.DevBuf(M * N * 2)
def run(i):
    if which == "gate_up":
        L.vra_wna16_gate_up_silu(x.ptr, ws[2*i].ptr, sc[2*i].ptr, None, ws[2*i+1].ptr, sc[2*i+1].ptr, None, out.ptr, M, K, N, 128, 0, 0, 0, 0)
    else:
        L.vra_wna16_gemm(x.ptr, ws[i].ptr, sc[i].ptr, None, None, None, out.ptr, M, K, N, 128, 0, 0, 0, 0)
e0, e1 = L.vra_event_create(), L.vra_event_create()
for i in range(8): run(i % nl)
L.vra_device_sync(); L.vra_event_record(e0, 0)
for i in range(40): run(i % nl)
L.vra_event_record(e1, 0); print("avg us", L.vra_event_elapsed_ms(e0, e1) * 25)
n = 4096 * 32
buf = (ctypes.c_ulonglong * n)()
L.vra_debug_ts.argtypes = [ctypes.c_void_p, ctypes.c_int]; L.vra_debug_ts(buf, n)
t = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 32).astype(np.int64)
g = int((t[:, 0] != 0).sum()); t = t[:g]; base = t[:, 0].min()
names = {0: "c:start", 1: "c:ring", 2: "c:b0 arrive", 3: "c:b0 pass", 4: "c:b1 arrive", 5: "c:b1 pass", 6: "c:b2 arr", 7: "c:b2 pass", 8: "c:b3 arr", 9: "c:b3 pass",
         14: "c:loop end", 15: "c:x free", 16: "p:start", 17: "p:stage0", 18: "p:stage1", 19: "p:stage2", 20: "p:stage3", 24: "p:partials", 25: "p:epilogue"}
print("workgroups", g)
for i in sorted(names):
    col = t[:, i]; ok = col != 0
    if ok.sum() == 0: continue
    r = (col[ok] - base) / 100.0
    print(f"{names[i]:12s} n={ok.sum():4d} min {r.min():7.2f} p50 {np.median(r):7.2f} p90 {np.percentile(r,90):7.2f} max {r.max():7.2f}")
items = {"gate_up": None, "o": 32, "down": 32, "qkv": 48}[which]
if items and g % items == 0 and g > items:
    kz = g // items
    for z in (0, kz // 2, kz - 1):
        r = (t[z * items:(z + 1) * items, 25] - base) / 100.0
        s0 = (t[z * items:(z + 1) * items, 16] - base) / 100.0
        for st, nm in ((26, "handshake done"), (27, "slabs summed (last unit)")):
            c = t[z * items:(z + 1) * items, st]
            if (c != 0).all(): print(f"   z={z} {nm}: p50 {np.median((c - base) / 100.0):6.2f} max {((c - base) / 100.0).max():6.2f}")
        print(f"slice z={z}: p:start p50 {np.median(s0):6.2f} max {s0.max():6.2f}   p:epilogue min {r.min():6.2f} p50 {np.median(r):6.2f} max {r.max():6.2f}" + ("   <- owner (sums the slabs)" if z == kz - 1 else ""))

# per-wave phase cycle sums of the compute waves (shader clock): chunk barriers | ring wait + dequant + MFMAs | fix-up + refill
ph = np.frombuffer(buf, dtype=np.uint64)[2048 * 32:2048 * 32 + g * 8 * 4].reshape(g, 8, 4).astype(np.float64)
ok = (ph[:, :, 3] > 0).all(axis=1)
print("phase sums present for", int(ok.sum()), "of", g, "workgroups")
if ok.any():
    ph = ph[ok]
    steps = ph[:, :, 3]
    print("cycles per tile-step, mean over workgroups, by compute wave (barrier | wait+dequant+mfma | fix-up+refill); steps per wave", int(steps[0, 0]))
    for w in range(8):
        print(f"  wave {w}: {(ph[:, w, 0] / steps[:, w]).mean():7.0f} | {(ph[:, w, 1] / steps[:, w]).mean():7.0f} | {(ph[:, w, 2] / steps[:, w]).mean():7.0f}")
