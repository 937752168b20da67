"""p50 TTFT of the long prompts of bench.py (Llama-3-8B shape): 2048 and 4096 tokens, 32 768 in 8192-token chunks — A/B aid for kernel D
and the prefill attention kernel.  VRA_LIB picks the library."""
import os
import sys

sys.path.insert(0, os.getcwd())
import bench
from vllm_rs_amd import engine as E

cfg = dict(E.LLAMA31_8B)
eng = E.Engine(cfg, max_num_seqs=8, max_model_len=40960, num_gpu_blocks=2048, use_graph=True, seed=1234, cpu_mem_fold=0.0).init_synthetic()
V = cfg["vocab_size"]
bench.ttft_p50(eng, 128, V, 1, reps=2)  # warm
out = {}
for plen, reps in [(2048, 5), (4096, 3), (32768, 2)]:
    out[f"bs1_prompt{plen}"] = round(bench.ttft_p50(eng, plen, V, 1, reps=reps), 2)
print(os.path.basename(os.environ.get("VRA_LIB", "default")), out)
eng.close()
