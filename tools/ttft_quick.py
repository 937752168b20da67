"""p50 TTFT of short prompts (bench.py's ttft_p50 on the Llama-3-8B shape) without the other bench legs: A/B aid for the short-prefill
kernels (VRA_GEMV_W_MAX_ROWS=32 puts the 33..256-row GEMMs back on kernels B / D)."""
import os
import sys

sys.path.insert(0, os.getcwd())
import bench
from vllm_rs_amd import engine as E

cfg = dict(E.LLAMA3_8B)
eng = E.Engine(cfg, max_num_seqs=32, max_model_len=8192, num_gpu_blocks=2048, use_graph=True, seed=1234, cpu_mem_fold=0.0).init_synthetic()
V = cfg["vocab_size"]
bench.ttft_p50(eng, 128, V, 1, reps=2)  # warm
out = {}
for plen, bs in [(64, 1), (128, 1), (200, 1), (256, 1), (128, 2), (128, 32)]:
    out[f"bs{bs}_prompt{plen}"] = round(bench.ttft_p50(eng, plen, V, bs, reps=7), 3)
print(os.environ.get("VRA_GEMV_W_MAX_ROWS", "default"), out)
eng.close()
