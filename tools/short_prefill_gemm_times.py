"""launch times of the four GEMMs of a layer at short-prefill row counts (bench_gemm: HIP events, rotating over the layers' weights)
    VRA_GEMV_W_MAX_ROWS=32 python tools/short_prefill_gemm_times.py   # kernels B / D
    python tools/short_prefill_gemm_times.py                          # kernel W in row blocks for q/k/v and o_proj"""
import os
import sys
sys.path.insert(0, ".")
from vllm_rs_amd import engine as E
cfg = dict(E.LLAMA3_8B, num_layers=8)
eng = E.Engine(cfg, max_num_seqs=32, max_model_len=4096, num_gpu_blocks=256, use_graph=False, seed=1, cpu_mem_fold=0.0).init_synthetic()
print("VRA_GEMV_W_MAX_ROWS =", os.environ.get("VRA_GEMV_W_MAX_ROWS", "default (256)"))
for M in (32, 64, 96, 128, 160, 200, 256, 512):
    row = []
    for w, nm in ((0, "norm+qkv"), (1, "o_proj"), (2, "norm+gate_up"), (3, "down")):
        row.append(f"{nm} {eng.bench_gemm(w, M, 40) * 1e3:7.2f}")
    print(f"M={M:4d}  " + "   ".join(row))
eng.close()
