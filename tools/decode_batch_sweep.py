"""ms per decode step of the Llama-3-8B-shape engine by batch size (graph replay), for A/B runs (VRA_NO_GEMV_W=1: kernel C / A)."""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import numpy as np

from vllm_rs_amd import engine as E

cfg = dict(E.LLAMA3_8B)
eng = E.Engine(cfg, max_num_seqs=32, max_model_len=2048, num_gpu_blocks=1024, use_graph=True).init_synthetic()
r = np.random.default_rng(0)
tag = " ".join(f"{k}={os.environ[k]}" for k in ("VRA_NO_GEMV_W", "VRA_NO_GEMV_S") if k in os.environ) or "default"
for bs in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 5, 8, 12, 16, 24, 32]:
    rids = [eng.add_request(r.integers(1000, 100000, size=128).astype(np.uint32), max_tokens=40, ignore_eos=True) for _ in range(bs)]
    while True:
        n, pf = eng.step()
        if not pf and n == bs:
            break
    for _ in range(4):
        eng.step()
    ms = eng.timed_decode(24) / 24
    while eng.has_unfinished():
        eng.step()
    print(f"[{tag}] bs {bs:2d}: {ms:6.3f} ms/step  {bs / ms * 1e3:8.0f} tok/s")
