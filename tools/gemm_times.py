"""per-GEMM launch times of a model shape (bench_gemm: HIP events, rotating over the layers' weights) + whole decode step
    python tools/gemm_times.py <qwen2-7b-awq|llama3-8b-gptq|llama3-70b-tp8-rank|tinyllama> [layers]"""
import sys
import time
sys.path.insert(0, ".")
import numpy as np
from vllm_rs_amd import engine as E
name = sys.argv[1] if len(sys.argv) > 1 else "qwen2-7b-awq"
cfg = dict({"qwen2-7b-awq": E.QWEN2_7B, "llama3-8b-gptq": E.LLAMA3_8B, "llama3-70b-tp8-rank": E.LLAMA3_70B_TP8_RANK, "tinyllama": E.TINYLLAMA}[name])
if len(sys.argv) > 2:
    cfg["num_layers"] = int(sys.argv[2])
eng = E.Engine(cfg, max_num_seqs=32, max_model_len=4096, num_gpu_blocks=512, use_graph=True, seed=1, cpu_mem_fold=0.0).init_synthetic()
L = cfg["num_layers"]
if cfg.get("quant_method"):
    for M in (1, 16, 32, 128, 2048):
        tot = 0.0
        for w, nm in ((0, "norm+qkv"), (1, "o_proj"), (2, "norm+gate_up"), (3, "down")):
            ms = eng.bench_gemm(w, M, 160 if M <= 32 else 12)
            b = eng.gemm_bytes(w, M)
            tot += ms
            H, I, D = cfg["hidden_size"], cfg["intermediate_size"], cfg["head_dim"]
            params = [H * (cfg["num_heads"] + 2 * cfg["num_kv_heads"]) * D, cfg["num_heads"] * D * H, 2 * H * I, I * H][w]
            print(f"M={M:4d} {nm:14s} {ms * 1e3:8.2f} us  {b / ms / 1e9:6.2f} TB/s  {2.0 * M * params / ms / 1e9:7.0f} TFLOP/s")
        print(f"M={M:4d} family {tot * 1e3:.2f} us per layer = {tot * L:.3f} ms per token")
for bs in (1, 32):
    r = np.random.default_rng(bs)
    rids = [eng.add_request(r.integers(0, cfg["vocab_size"], size=128).astype(np.uint32), max_tokens=200, ignore_eos=True) for _ in range(bs)]
    for _ in range(12):
        eng.step()
    eng.L.vra_device_sync()
    t0 = time.perf_counter()
    for _ in range(128):
        eng.step()
    eng.L.vra_device_sync()
    dt = time.perf_counter() - t0
    print(f"decode bs={bs}: {dt / 128 * 1e3:.3f} ms/step = {bs * 128 / dt:.0f} tok/s")
    while eng.has_unfinished():
        eng.step()
eng.close()
