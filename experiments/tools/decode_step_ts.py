"""Per-workgroup / per-phase timeline of the persistent decode step (kernel P).  Needs a -DVRA_GEMV_TS build:
  make -C vllm_rs_amd/csrc B=build_ts EXTRA=-DVRA_GEMV_TS OUT=$PWD/vllm_rs_amd/libvra_ts.so RUNNER=/tmp/vra_runner_ts
run with VRA_LIB=vllm_rs_amd/libvra_ts.so.  Stamps are wall-clock (100 MHz) values of consumer 0 of every workgroup:
0 phase start | 1 grid barrier passed | 2 x staged (GEMV) / new token staged (attention) | 3 loop end | 4 partials met |
5 outputs stored, arrived."""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
from vllm_rs_amd import engine as E  # noqa: E402

L = int(os.environ.get("TS_LAYERS", "4"))
cfg = dict(E.LLAMA3_8B)
cfg["num_layers"] = L
eng = E.Engine(cfg, max_num_seqs=4, max_model_len=2048, num_gpu_blocks=64, use_graph=False).init_synthetic()
prompt = list(range(1000, 1000 + int(os.environ.get("TS_PROMPT", "150"))))
eng.generate([prompt], max_tokens=int(os.environ.get("TS_TOKENS", "12")), ignore_eos=True)
nph = 5 * L
n = 256 * 256 * 8 + 256 * 64 * 4 + 8 * 16 * 8
buf = (ctypes.c_ulonglong * n)()
eng.L.vra_debug_decode_step_ts.argtypes = [ctypes.c_void_p, ctypes.c_int]
eng.L.vra_debug_decode_step_ts(buf, n)
allts = np.frombuffer(buf, dtype=np.uint64).astype(np.int64)
raw = allts[:256 * 256 * 8].reshape(256, 256, 8)[:, :nph, :]
slot_ts = allts[256 * 256 * 8:256 * 256 * 8 + 256 * 64 * 4].reshape(256, 64, 4)
cyc_ts = allts[256 * 256 * 8 + 256 * 64 * 4:].reshape(8, 16, 8)
t = raw[:, :, :6]
# shader clock over the main loop of the gate/up phases: cycles (s_memtime) per wall-clock microsecond
gu = [5 * l + 3 for l in range(L)]
cyc = (raw[:, gu, 7] - raw[:, gu, 6]).astype(np.float64)
wall = (raw[:, gu, 3] - raw[:, gu, 2]).astype(np.float64) / 100.0
print(f"shader clock during the gate/up main loops: {np.median(cyc / np.maximum(wall, 1e-9)):.0f} cycles per us (median over workgroups and layers)")
if not t.any():
    sys.exit("no stamps: not a VRA_GEMV_TS build, or the persistent step did not run")
base = t[:, 0, 0].min()
r = (t - base) / 100.0  # us
names = ["qkv", "attn", "o", "gate/up", "down"]
print(f"launch: first start -> last end {r[:, nph - 1, 5].max():.2f} us for {L} layers = {r[:, nph - 1, 5].max() / L:.2f} us per layer")
print("phase                start(min/max)   grid passed(p50/max)   staged(p50)   loop end(p50/max)   met(p50)   arrived(p50/max)   | span of the phase (first start -> last arrive)")
for ph in range(nph):
    x = r[:, ph, :]
    act = x[:, 3] > 0 if ph % 5 != 1 else x[:, 5] > 0
    st = x[:, 0]
    row = f"L{ph // 5} {names[ph % 5]:8s}"
    g = x[:, 1]
    print(f"{row:14s} {st.min():8.2f} {st.max():7.2f}   {np.median(g):8.2f} {g.max():7.2f}   {np.median(x[:, 2][x[:, 2] > 0]) if (x[:, 2] > 0).any() else 0:8.2f}   "
          f"{np.median(x[:, 3][x[:, 3] > 0]) if (x[:, 3] > 0).any() else 0:8.2f} {x[:, 3].max():7.2f}   {np.median(x[:, 4][x[:, 4] > 0]) if (x[:, 4] > 0).any() else 0:8.2f}   "
          f"{np.median(x[:, 5]):8.2f} {x[:, 5].max():7.2f}   | {x[:, 5].max() - st.min():6.2f}")
# per-phase durations averaged over layers >= 1 (layer 0 starts with an empty ring)
if L > 1:
    print("\nmean over layers 1..: last arrive of the previous phase -> last arrive of this phase (what the phase adds to the critical path)")
    for k in range(5):
        d = []
        for l in range(1, L):
            ph = 5 * l + k
            d.append(r[:, ph, 5].max() - r[:, ph - 1, 5].max())
        seg = []
        for l in range(1, L):
            ph = 5 * l + k
            x = r[:, ph, :]
            prev_end = r[:, ph - 1, 5].max()
            seg.append([np.median(x[:, 1]) - prev_end, np.median(x[:, 2][x[:, 2] > 0]) - np.median(x[:, 1]) if (x[:, 2] > 0).any() else 0,
                        x[:, 3].max() - np.median(x[:, 2][x[:, 2] > 0]) if (x[:, 2] > 0).any() else 0, x[:, 5].max() - x[:, 3].max()])
        seg = np.array(seg).mean(axis=0)
        print(f"  {names[k]:8s} {np.mean(d):6.2f} us   (barrier seen {seg[0]:5.2f} | staging {seg[1]:5.2f} | stream until the LAST workgroup's loop end {seg[2]:5.2f} | reduce + store + arrive {seg[3]:5.2f})")
eng.close()

# per-slot timeline of phase VRA_TS_PHASE (default 8 = layer 1 gate/up) for a few workgroups
tsp = int(os.environ.get("VRA_TS_PHASE", "8"))
if tsp < nph:
    p0 = t[:, tsp, 0].min()
    print(f"\nper-slot stamps of phase {tsp} (us after the phase's first start): issued | published | consumer 0 requests | consumer 0 done")
    for wg in (0, 100, 200):
        rows = []
        for k in range(64):
            v = slot_ts[wg, k]
            if not v.any():
                break
            rows.append(" ".join(f"{(x - p0) / 100.0:6.2f}" if x else "   -  " for x in v))
        print(f" wg {wg}: grid passed {(t[wg, tsp, 1] - p0) / 100.0:.2f}, staged {(t[wg, tsp, 2] - p0) / 100.0:.2f}, loop end {(t[wg, tsp, 3] - p0) / 100.0:.2f}")
        for k, r_ in enumerate(rows):
            print(f"   slot {k:2d}: {r_}")

print("\nshader-clock stamps inside the main loop, workgroup 100, per consumer and k-step: cycles from the step's start: landed seen | reads back | MFMAs + drain | fix-up; then gap to the next step")
for cc in range(8):
    for k in range(16):
        v = cyc_ts[cc, k]
        if not v[0]:
            break
        nxt = cyc_ts[cc, k + 1, 0] if k + 1 < 16 and cyc_ts[cc, k + 1, 0] else 0
        print(f"  consumer {cc} step {k:2d}: " + " ".join(f"{int(v[i] - v[0]):6d}" for i in range(1, 5)) + (f"   next step starts at {int(nxt - v[0]):6d}" if nxt else ""))
