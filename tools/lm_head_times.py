"""us per launch of the decode lm_head (final norm + lm_head from the tile-major copy + greedy tokens) by row count, in the engine
(vra_engine_bench_gemm which = 4): python tools/lm_head_times.py [rows ...]; VRA_LIB selects an experiment build"""
import os, sys
sys.path.insert(0, os.getcwd())
from vllm_rs_amd import engine as E
cfg = dict(E.LLAMA3_8B)
cfg["num_layers"] = 1
eng = E.Engine(cfg, max_num_seqs=32, max_model_len=1024, num_gpu_blocks=64, use_graph=False).init_synthetic()
rows = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8, 16, 32]
for rep in range(2):
    print(os.environ.get("VRA_LIB", "default").split("/")[-1], " ".join(f"M{m}: {eng.bench_gemm(4, m, 100) * 1e3:6.1f}" for m in rows), "us", flush=True)
