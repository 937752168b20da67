"""Per-workgroup timeline of kernel W (needs a -DVRA_GEMV_TS build: make B=build_ts EXTRA=-DVRA_GEMV_TS OUT=../libvra_ts.so;
run with VRA_LIB=.../libvra_ts.so).  Stamps are wall-clock (100 MHz) values of wave 0 of every workgroup.
usage: gemv_w_ts.py [rows] [which ...]   (which: 0 = norm + q/k/v, 1 = o_proj + residual, 2 = norm + gate/up, 3 = down_proj in K slices)"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
from vllm_rs_amd import engine as E

cfg = dict(E.LLAMA3_8B)
cfg["num_layers"] = int(os.environ.get("TS_LAYERS", "8"))
eng = E.Engine(cfg, max_num_seqs=32, max_model_len=2048, num_gpu_blocks=64, use_graph=False).init_synthetic()
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 32
names = {0: "start", 1: "epi staged", 2: "x+ring issued", 3: "sumsq done", 4: "norm barrier", 5: "normalised", 6: "xsum done", 7: "kz flags seen", 8: "unit0 mfma", 9: "unit0 barrier",
         10: "unit0 reduced", 11: "unitN mfma", 12: "unitN barrier", 13: "unitN reduced", 14: "final barrier", 15: "end"}
for which in [int(a) for a in sys.argv[2:]] or [0, 1]:
    ms = eng.bench_gemm(which, rows, 50)
    n = 4096 * 32
    buf = (ctypes.c_ulonglong * n)()
    eng.L.vra_debug_ts.argtypes = [ctypes.c_void_p, ctypes.c_int]
    eng.L.vra_debug_ts(buf, n)
    t = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 32).astype(np.int64)
    t[t < t[:, 0].max() - 5000] = 0  # stamps older than 50 us before the last launch's starts: an earlier launch (another kernel shape)
    g = int((t[:2048, 0] != 0).sum())
    t = t[:g]
    base = t[:, 0].min()
    print(f"== W {which} at {rows} rows: avg {ms * 1e3:.2f} us per launch; grid {g}; first start -> last end {(t[:, 15].max() - base) / 100.0:.2f} us")
    for i in sorted(names):
        col = t[:, i]
        ok = col != 0
        if ok.sum() == 0:
            continue
        r = (col[ok] - base) / 100.0
        print(f"  {names[i]:14s} n={ok.sum():4d}  min {r.min():6.2f}  p50 {np.median(r):6.2f}  p90 {np.percentile(r, 90):6.2f}  max {r.max():6.2f}")
    for w in (0, g // 2, g - 1):
        print("  wg", w, {names[i]: round((int(t[w, i]) - int(base)) / 100.0, 2) for i in sorted(names) if t[w, i]})
