"""Per-launch time of the four decode GEMVs of a Llama-3-8B layer at M rows (HIP events, rotating layers), for A/B runs:
VRA_LIB=<other .so>, VRA_EXP=2 (no tail prefetch), VRA_GS_GRID=all, VRA_NO_GEMV_S=1 (round-1 kernel A)."""
import os
import sys

sys.path.insert(0, os.getcwd())
from vllm_rs_amd import engine as E

cfg = dict(E.LLAMA3_8B)
cfg["num_layers"] = 8
eng = E.Engine(cfg, max_num_seqs=8, max_model_len=2048, num_gpu_blocks=64, use_graph=False).init_synthetic()
tag = " ".join(f"{k}={os.environ[k]}" for k in ("VRA_LIB", "VRA_EXP", "VRA_GS_GRID", "VRA_NO_GEMV_S", "VRA_NO_GEMV_W") if k in os.environ) or "default"
for M in [int(a) for a in sys.argv[1:]] or [1]:
    tot_ms = tot_b = 0
    parts = []
    for w, name in [(0, "qkv"), (1, "o"), (2, "gate_up"), (3, "down")]:
        ms = eng.bench_gemm(w, M, 400)
        b = eng.gemm_bytes(w, M)
        tot_ms += ms
        tot_b += b
        parts.append(f"{name} {ms * 1e3:6.2f}us {b / ms / 1e6:5.0f}GB/s")
    print(f"[{tag[-60:]}] M={M}: " + " | ".join(parts) + f" || layer {tot_ms * 1e3:6.2f}us family {tot_b / tot_ms / 1e6:5.0f} GB/s = {tot_b / tot_ms / 8e9:.3f}")
