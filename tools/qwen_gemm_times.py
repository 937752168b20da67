"""per-GEMM launch times of the Qwen2-7B AWQ shape (bench_gemm: HIP events, rotating over the layers' weights)"""
import sys
sys.path.insert(0, ".")
from vllm_rs_amd import engine as E
eng = E.Engine(E.QWEN2_7B, max_num_seqs=32, max_model_len=2048, num_gpu_blocks=256, use_graph=False, seed=1, cpu_mem_fold=0.0).init_synthetic()
for M in (1, 32):
    tot = 0.0
    for w, name in ((0, "norm+qkv"), (1, "o_proj"), (2, "norm+gate_up"), (3, "down")):
        ms = eng.bench_gemm(w, M, 160)
        b = eng.gemm_bytes(w, M)
        tot += ms
        print(f"M={M:2d} {name:14s} {ms * 1e3:7.2f} us  {b / ms / 1e9:6.2f} TB/s")
    print(f"M={M:2d} family {tot * 1e3:.2f} us per layer = {tot * 28:.3f} ms per token")
eng.close()
