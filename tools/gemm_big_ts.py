"""Phase cycle sums of kernel D (needs a -DVRA_GEMV_TS build: make B=build_ts EXTRA=-DVRA_GEMV_TS OUT=../libvra_ts.so; run with
VRA_LIB=.../libvra_ts.so).  Per wave and k-tile: dequant + load issue | MFMA steps | x tile -> LDS (waits for the x loads) | barrier."""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
from vllm_rs_amd import engine as E

cfg = dict(E.LLAMA3_8B)
cfg["num_layers"] = 2
eng = E.Engine(cfg, max_num_seqs=8, max_model_len=2048, num_gpu_blocks=64, use_graph=False).init_synthetic()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for which, name, K in [(0, "qkv", 4096), (1, "o", 4096), (2, "gate_up", 4096), (3, "down", 14336)]:
    ms = eng.bench_gemm(which, M, 3)
    n = 4096 * 32
    buf = (ctypes.c_ulonglong * n)()
    eng.L.vra_debug_ts.argtypes = [ctypes.c_void_p, ctypes.c_int]
    eng.L.vra_debug_ts(buf, n)
    t = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 8, 4).astype(np.float64)
    g = int((t[:, 0, 1] != 0).sum())
    t = t[:g] / (K / 128)
    print(f"== {name} M={M}: {ms * 1e3:.1f} us per launch; {g} workgroups sampled; cycles per k-tile (mean over workgroups)")
    for w in range(8):
        print(f"  wave {w}: dequant+issue {t[:, w, 0].mean():7.0f} | mfma {t[:, w, 1].mean():7.0f} | x store {t[:, w, 2].mean():7.0f} | barrier {t[:, w, 3].mean():7.0f} | total {t[:, w].sum(axis=1).mean():7.0f}")
