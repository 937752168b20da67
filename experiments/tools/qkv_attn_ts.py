"""Timeline of the fused q/k/v + attention decode launch (csrc/qkv_attn.hip).  Needs a -DVRA_GEMV_TS build:
  make -C vllm_rs_amd/csrc B=build_ts EXTRA=-DVRA_GEMV_TS OUT=$PWD/vllm_rs_amd/libvra_ts.so RUNNER=/tmp/vra_runner_ts
run with VRA_LIB=.../libvra_ts.so.  Stamps are wall-clock (100 MHz) values of thread 0 of every workgroup; the attention stamps
exist only on the attention workgroups (one per sequence and kv head)."""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
from vllm_rs_amd import engine as E

cfg = dict(E.LLAMA3_8B)
cfg["num_layers"] = 1  # ONE layer: the stamps of the fused launch are not overwritten by a later layer's
ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 200
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 1
eng = E.Engine(cfg, max_num_seqs=8, max_model_len=4096, num_gpu_blocks=256, use_graph=False).init_synthetic()
r = np.random.default_rng(0)
rids = [eng.add_request(r.integers(1000, 100000, size=ctx).astype(np.uint32), max_tokens=40, ignore_eos=True) for _ in range(bs)]
for _ in range(30):
    eng.step()
n = 4096 * 32
buf = (ctypes.c_ulonglong * n)()
eng.L.vra_debug_ts.argtypes = [ctypes.c_void_p, ctypes.c_int]
eng.L.vra_debug_ts(buf, n)
t = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 32).astype(np.int64)
g = int((t[:2048, 31] != 0).sum())
t = t[:g]
base = t[:, 31].min()
names = {31: "kernel entry", 30: "gemv part done (incl. granule stores issued)", 19: "attention start", 20: "ctx/pos/slot known", 21: "K/V tile loads issued",
         22: "rope rows staged", 23: "granules swept (wave 0)", 24: "barrier 1", 25: "q/k rotated, barrier 2", 26: "tiles done", 27: "merge barrier", 28: "end"}
print(f"# tools/qkv_attn_ts.py {ctx} {bs}: fused q/k/v + attention launch, context {ctx}+, {bs} sequence(s); grid {g}")
for i in (31, 30, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28):
    col = t[:, i]
    ok = col != 0
    if ok.sum() == 0:
        continue
    v = (col[ok] - base) / 100.0
    print(f"  {names[i]:48s} n={ok.sum():4d}  min {v.min():6.2f}  p50 {np.median(v):6.2f}  max {v.max():6.2f}")
att = np.nonzero(t[:, 28] != 0)[0]
for w in att[:3]:
    print("  wg", int(w), {names[i].split(" (")[0]: round((int(t[w, i]) - int(base)) / 100.0, 2) for i in (31, 30, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28) if t[w, i]})
eng.close()
