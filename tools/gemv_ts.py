"""Per-workgroup timeline of gemv_kernel (needs a -DVRA_GEMV_TS build: make B=build_ts EXTRA=-DVRA_GEMV_TS OUT=.../libvra_ts.so)."""
import sys, os, ctypes
import numpy as np
sys.path.insert(0, os.getcwd())
from vllm_rs_amd import engine as E
cfg = dict(E.LLAMA3_8B); cfg["num_layers"] = int(os.environ.get("TS_LAYERS", "8"))
eng = E.Engine(cfg, max_num_seqs=8, max_model_len=2048, num_gpu_blocks=64, use_graph=False).init_synthetic()
which = int(sys.argv[1]) if len(sys.argv) > 1 else 2
ms = eng.bench_gemm(which, 1, 50)
print("avg us", ms * 1e3)
n = 4096 * 32
buf = (ctypes.c_ulonglong * n)()
eng.L.vra_debug_ts.argtypes = [ctypes.c_void_p, ctypes.c_int]
eng.L.vra_debug_ts(buf, n)
t = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 32).astype(np.int64)
g = int((t[:, 0] != 0).sum())
t = t[:g]
base = t[:, 0].min()
print("grid", g, "span(us) first start -> last end", (t[:, 15].max() - base) / 100.0)
names = ["start", "issued", "prologue", "c0", "e0", "c1", "e1", "c2", "e2", "c3", "e3", "c4", "e4", "c5+", "e5+", "end", "p:xissued", "p:ss", "p:bar1", "p:bar2", "p:staged"]
for i, nm in enumerate(names):
    col = t[:, i]
    ok = col != 0
    if ok.sum() == 0: continue
    r = (col[ok] - base) / 100.0
    print(f"{nm:9s} n={ok.sum():4d}  min {r.min():7.2f}  p50 {np.median(r):7.2f}  p90 {np.percentile(r,90):7.2f}  max {r.max():7.2f}")
for w in (0, 1, g // 2, g - 1):
    print("wg", w, [(int(v - base) if v else None) for v in t[w]])
