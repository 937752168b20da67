import os, sys
sys.path.insert(0, ".")
from vllm_rs_amd import engine as E
cfg = dict(E.LLAMA3_8B, num_layers=4)
eng = E.Engine(cfg, max_num_seqs=32, max_model_len=4096, num_gpu_blocks=256, use_graph=False, seed=1, cpu_mem_fold=0.0).init_synthetic()
print("VRA_GD_MB =", os.environ.get("VRA_GD_MB", "cost model"), " VRA_GD_SPLITK =", os.environ.get("VRA_GD_SPLITK", "cost model"))
for M in (160, 200, 224, 256):
    print(f"M={M:4d}  gate_up {eng.bench_gemm(2, M, 30) * 1e3:7.2f}   down {eng.bench_gemm(3, M, 30) * 1e3:7.2f}")
