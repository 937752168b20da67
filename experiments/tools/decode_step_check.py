"""GPU check of the persistent decode step (csrc/decode_step.hip) against the launch-per-op decode path of the same
engine: logits of decode steps must be BIT-IDENTICAL (same GEMV arithmetic; same attention tile split while the
launch path does not split the KV range).  Usage: python tools/decode_step_check.py [variant ...]"""
import ctypes
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from oracle import model as om  # noqa: E402  (test infrastructure: only for the random checkpoint)
from vllm_rs_amd import _lib  # noqa: E402
from vllm_rs_amd.engine import Engine  # noqa: E402

BF16, F16 = 0, 1


def small_cfg(**kw):
    cfg = dict(arch="llama", hidden_size=256, intermediate_size=512, num_layers=2, num_heads=4, num_kv_heads=2, head_dim=64,
               vocab_size=512, max_position_embeddings=2048, rms_norm_eps=1e-5, rope_theta=10000.0, quant_method="gptq",
               group_size=128, dtype=BF16)
    cfg.update(kw)
    return cfg


VARIANTS = {
    "small": small_cfg(),
    "small_f16": small_cfg(dtype=F16),
    "small_awq": small_cfg(arch="qwen2", quant_method="awq", attention_bias=True, num_heads=8, num_kv_heads=2, head_dim=64, hidden_size=512),
    "llama3_8b_1l": small_cfg(hidden_size=4096, intermediate_size=14336, num_layers=1, num_heads=32, num_kv_heads=8, head_dim=128,
                              vocab_size=2048, rope_theta=500000.0),
    "llama3_8b_3l": small_cfg(hidden_size=4096, intermediate_size=14336, num_layers=3, num_heads=32, num_kv_heads=8, head_dim=128,
                              vocab_size=2048, rope_theta=500000.0),
    "qwen2_7b_1l": small_cfg(arch="qwen2", attention_bias=True, hidden_size=3584, intermediate_size=18944, num_layers=1, num_heads=28,
                             num_kv_heads=4, head_dim=128, vocab_size=2048, quant_method="awq", rope_theta=1e6, rms_norm_eps=1e-6),
    "tinyllama_q": small_cfg(hidden_size=2048, intermediate_size=5632, num_layers=2, num_heads=32, num_kv_heads=4, head_dim=64,
                             vocab_size=2048),
}


def run(name, lens=(37,), steps=3):
    cfg = VARIANTS[name]
    lib = _lib.load()
    w = om.make_random_checkpoint(cfg, 3)
    eng = Engine(cfg, num_gpu_blocks=64, max_num_seqs=8, max_model_len=2048, use_graph=False).load_weights(w)
    r = np.random.default_rng(1)
    prompts = [r.integers(0, cfg["vocab_size"], size=n).tolist() for n in lens]
    BS = 64
    nb = [(len(p) + steps + BS) // BS for p in prompts]
    bt = np.zeros((len(prompts), max(nb)), np.uint32)
    nxt = 0
    for i, n in enumerate(nb):
        bt[i, :n] = np.arange(nxt, nxt + n)
        nxt += n
    ids, pos, slots, cu = [], [], [], [0]
    for b, p in enumerate(prompts):
        for j in range(len(p)):
            ids.append(p[j]); pos.append(j); slots.append(int(bt[b, j // BS]) * BS + j % BS)
        cu.append(len(ids))
    ctx = np.array([len(p) for p in prompts], np.uint32)
    lib.vra_debug_set_decode_step(0)
    logits = eng.forward_raw(np.array(ids, np.uint32), np.array(pos, np.int64), np.array(slots, np.int64), bt, ctx, np.array(cu, np.uint32))
    seqs = [list(p) for p in prompts]
    tok = logits.argmax(-1)
    ok = True
    for step in range(steps):
        for s, t in zip(seqs, tok):
            s.append(int(t))
        ids = np.array([s[-1] for s in seqs], np.uint32)
        pos = np.array([len(s) - 1 for s in seqs], np.int64)
        slots = np.array([int(bt[b, (len(s) - 1) // BS]) * BS + (len(s) - 1) % BS for b, s in enumerate(seqs)], np.int64)
        ctx = np.array([len(s) for s in seqs], np.uint32)
        lib.vra_debug_set_decode_step(0)
        ref = eng.forward_raw(ids, pos, slots, bt, ctx, None)
        lib.vra_debug_set_decode_step(1)
        t0 = time.time()
        got = eng.forward_raw(ids, pos, slots, bt, ctx, None)
        dt = time.time() - t0
        same = np.array_equal(got.view(np.uint32), ref.view(np.uint32))
        d = np.abs(got - ref)
        print(f"[{name} lens={lens}] step {step}: bitwise {'EQUAL' if same else 'DIFFERENT'}; max |d| {d.max():.5f} (scale {np.abs(ref).max():.2f}); "
              f"finite {np.isfinite(got).all()}; {dt * 1e3:.1f} ms", flush=True)
        ok = ok and same
        tok = ref.argmax(-1)
    eng.close()
    return ok


if __name__ == "__main__":
    names = sys.argv[1:] or ["small", "small_f16", "small_awq", "tinyllama_q", "llama3_8b_1l", "qwen2_7b_1l", "llama3_8b_3l"]
    allok = True
    for n in names:
        allok = run(n, (37,)) and allok
        allok = run(n, (70, 5)) and allok
    print("ALL EQUAL" if allok else "MISMATCH")
