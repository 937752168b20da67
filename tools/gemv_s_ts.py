"""Per-workgroup timeline of kernel E (needs a -DVRA_GEMV_TS build: make B=build_ts EXTRA=-DVRA_GEMV_TS OUT=../libvra_ts.so;
run with VRA_LIB=.../libvra_ts.so).  Stamps are wall-clock (100 MHz) values of wave 0 of every workgroup."""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
from vllm_rs_amd import engine as E

cfg = dict(E.LLAMA3_8B)
cfg["num_layers"] = int(os.environ.get("TS_LAYERS", "8"))
eng = E.Engine(cfg, max_num_seqs=8, max_model_len=2048, num_gpu_blocks=64, use_graph=False).init_synthetic()
names = {0: "start", 16: "x staged", 1: "ring issued", 18: "norm barrier", 2: "staged", 3: "step0 done", 5: "step1 done", 7: "step2 done", 9: "step3 done",
         11: "step4 done", 13: "step5+ done", 15: "loop end", 17: "final barrier", 14: "end"}
for which in [int(a) for a in sys.argv[1:]] or [1, 2]:
    ms = eng.bench_gemm(which, 1, 50)
    n = 4096 * 32
    buf = (ctypes.c_ulonglong * n)()
    eng.L.vra_debug_ts.argtypes = [ctypes.c_void_p, ctypes.c_int]
    eng.L.vra_debug_ts(buf, n)
    t = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 32).astype(np.int64)
    g = int((t[:2048, 0] != 0).sum())
    t = t[:g]
    base = t[:, 0].min()
    print(f"== GEMV {which}: avg {ms * 1e3:.2f} us per launch; grid {g}; first start -> last end {(t[:, 14].max() - base) / 100.0:.2f} us")
    for i in (0, 16, 1, 18, 2, 3, 5, 7, 9, 11, 13, 15, 17, 14):
        col = t[:, i]
        ok = col != 0
        if ok.sum() == 0:
            continue
        r = (col[ok] - base) / 100.0
        print(f"  {names[i]:13s} n={ok.sum():4d}  min {r.min():6.2f}  p50 {np.median(r):6.2f}  p90 {np.percentile(r, 90):6.2f}  max {r.max():6.2f}")
    # every wave's loop end (slots behind the per-workgroup stamps): how far apart do the 16 waves of a workgroup finish?
    pw = np.frombuffer(buf, dtype=np.uint64)[2048 * 32:2048 * 32 + g * 16].reshape(g, 16).astype(np.int64)
    if (pw != 0).all():
        rel = (pw - base) / 100.0
        spread = rel.max(axis=1) - rel.min(axis=1)
        print(f"  per-wave loop end: first wave p50 {np.median(rel.min(axis=1)):6.2f}  last wave p50 {np.median(rel.max(axis=1)):6.2f}  spread p50 {np.median(spread):5.2f} p90 {np.percentile(spread, 90):5.2f} max {spread.max():5.2f}")
        print("  per-wave loop end, median over workgroups, wave 0..15:", " ".join(f"{v:5.2f}" for v in np.median(rel, axis=0)))
    for w in (0, g // 2, g - 1):
        print("  wg", w, {names[i]: round((int(t[w, i]) - int(base)) / 100.0, 2) for i in names if t[w, i]})
