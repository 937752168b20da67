"""Timeline of the two-phase decode launches (gemv_q4s2_kernel; needs a -DVRA_GEMV_TS build, VRA_LIB=.../libvra_ts.so):
phase A stamps in rows 0.., phase B stamps in rows 1024.. of the stamp buffer; times relative to the first workgroup's start."""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
from vllm_rs_amd import engine as E

cfg = dict(E.LLAMA3_8B)
cfg["num_layers"] = int(os.environ.get("TS_LAYERS", "32"))
eng = E.Engine(cfg, max_num_seqs=8, max_model_len=2048, num_gpu_blocks=64, use_graph=False).init_synthetic()
names = {0: "start", 16: "x in LDS", 1: "ring issued", 19: "at barrier", 20: "barrier passed", 18: "norm barrier", 2: "staged", 3: "step0 done", 15: "loop end",
         17: "final barrier", 14: "end"}
order = (0, 19, 20, 16, 1, 18, 2, 3, 15, 17, 14)
for which, label in ((4, "o_proj -> gate/up"), (5, "down -> next q/k/v")):
    sep = eng.bench_gemm(1 if which == 4 else 3, 1, 50) + eng.bench_gemm(2 if which == 4 else 0, 1, 50)
    ms = eng.bench_gemm(which, 1, 50)
    n = 4096 * 32
    buf = (ctypes.c_ulonglong * n)()
    eng.L.vra_debug_ts.argtypes = [ctypes.c_void_p, ctypes.c_int]
    eng.L.vra_debug_ts(buf, n)
    t = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 32).astype(np.int64)
    A, B = t[:1024], t[1024:2048]
    g = int((B[:, 14] != 0).sum())
    base = A[:g, 0].min()
    print(f"== {label}: two-phase {ms * 1e3:.2f} us per launch, the two launches {sep * 1e3:.2f} us; grid {g}")
    for ph, T in (("A", A[:g]), ("B", B[:g])):
        for i in order:
            if ph == "A" and i in (19, 20):
                continue
            col = T[:, i]
            ok = col != 0
            if ok.sum() == 0:
                continue
            r = (col[ok] - base) / 100.0
            r = r[(r > -1) & (r < 200)]
            if len(r):
                print(f"  {ph} {names[i]:15s} n={len(r):4d}  min {r.min():6.2f}  p50 {np.median(r):6.2f}  max {r.max():6.2f}")
