import sys, os
sys.path.insert(0, os.getcwd())
from vllm_rs_amd import engine as E
cfg = dict(E.LLAMA3_8B); cfg["num_layers"] = 8
eng = E.Engine(cfg, max_num_seqs=8, max_model_len=2048, num_gpu_blocks=64, use_graph=False).init_synthetic()
for w, name in [(2, "gate_up"), (3, "down"), (0, "qkv"), (1, "o")]:
    ms = eng.bench_gemm(w, 1, 200); b = eng.gemm_bytes(w, 1)
    print(f"{os.environ.get('VRA_LIB','base')[-20:]:20s} {name:8s} {ms*1e3:8.2f} us  {b/ms/1e6:8.1f} GB/s")
