#!/usr/bin/env python3
"""gpurun_out/prof_* (tools/collect_profiles.sh) -> profiles/r06_*.txt and profiles/r06_pmc.json.

r06_pmc.json carries the .so hash the counters were taken with: bench.py reports `roofline.traffic` only when the library it
runs is that very library (a kernel change can never leave a stale number in the driver's line).
HBM bytes per launch = FETCH_SIZE [KiB] * 1024 * 2 — the gfx950 correction of MI355X_MICROARCH.md (HBM section): FETCH_SIZE reports
half of the bytes of a wide coalesced read.  WRITE_SIZE is reported as measured (uncalibrated on this part)."""
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")


def parse_pmc(path):
    """rocpd_pmc.py summary -> {(kernel, counter): (calls, avg_value, avg_us)}"""
    out = {}
    if not os.path.exists(path):
        return out
    for line in open(path):
        m = re.match(r"^(.{64}) (\S+)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s*$", line.rstrip("\n"))
        if m:
            out[(m.group(1).strip(), m.group(2))] = (int(m.group(3)), float(m.group(4)), float(m.group(5)))
    return out


def main():
    os.makedirs(P, exist_ok=True)
    copies = {"prof_trace_bs1.summary.txt": "r06_decode_bs1_kernel_trace.txt", "prof_trace_bs32.summary.txt": "r06_decode_bs32_kernel_trace.txt",
              "prof_trace_prefill.summary.txt": "r06_prefill_4096_kernel_trace.txt", "prof_pmc_fetch_bs1.summary.txt": "r06_pmc_fetch_size_decode_bs1.txt",
              "prof_pmc_write_bs1.summary.txt": "r06_pmc_write_size_decode_bs1.txt", "prof_pmc_mfma_bs1.summary.txt": "r06_pmc_mfma_decode_bs1.txt",
              "prof_pmc_mfma_bs32.summary.txt": "r06_pmc_mfma_decode_bs32.txt", "prof_pmc_fetch_bs32.summary.txt": "r06_pmc_fetch_size_decode_bs32.txt",
              "prof_pmc_mfma_prefill.summary.txt": "r06_pmc_mfma_prefill_4096.txt", "prof_counters_available.txt": "r06_counters_available.txt",
              "prof_trace_prefill128.summary.txt": "r06_prefill_128_kernel_trace.txt", "prof_trace_qwen2_bs1.summary.txt": "r06_decode_bs1_qwen2_7b_awq_kernel_trace.txt",
              "prof_trace_qwen2_bs32.summary.txt": "r06_decode_bs32_qwen2_7b_awq_kernel_trace.txt",
              "prof_pmc_fetch_qwen2_bs1.summary.txt": "r06_pmc_fetch_size_decode_bs1_qwen2_7b_awq.txt"}
    notes = {"trace": "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 32 --warmup 4 --batch {B} --no-graph --no-extras (eager launches; includes the "
                      "one-time weight-fill and prefill kernels), summarised by tools/rocpd_stats.py\n",
             "pmc": "# rocprofv3 --pmc <counters> -- python bench.py --steps 32 --warmup 4 --batch {B} --no-graph --no-extras (counter pass on its own: no trace "
                    "domain), summarised by tools/rocpd_pmc.py\n"}
    for src, dst in copies.items():
        sp = os.path.join(G, src)
        if os.path.exists(sp) and os.path.getsize(sp) > 0:
            txt = open(sp).read()
            kind = "trace" if "trace" in src else ("pmc" if "pmc" in src else None)
            with open(os.path.join(P, dst), "w") as f:
                f.write(txt)
                if kind:
                    note = notes[kind].replace("{B}", "32" if "bs32" in src else "1")
                    if "prefill" in src:  # the prefill runs profile tools/prefill_once.py (two 4096- or 128-token prompts), not bench.py
                        note = note.replace("python bench.py --steps 32 --warmup 4 --batch 1 --no-graph --no-extras", "python tools/prefill_once.py" + (" 128" if "128" in src else ""))
                    if "qwen2" in src:
                        note = note.replace("python bench.py", "python bench.py --model qwen2-7b-awq")
                    f.write(note)
            print("wrote", dst)
    shas = open(os.path.join(G, "prof_lib_sha16.txt")).read().split() if os.path.exists(os.path.join(G, "prof_lib_sha16.txt")) else []
    sha = shas[0] if shas else None
    src_sha = shas[1] if len(shas) > 1 else None
    fetch = parse_pmc(os.path.join(G, "prof_pmc_fetch_bs1.summary.txt"))
    write = parse_pmc(os.path.join(G, "prof_pmc_write_bs1.summary.txt"))
    mfma = parse_pmc(os.path.join(G, "prof_pmc_mfma_bs1.summary.txt"))
    kernels = {}
    for (k, c), (calls, val, us) in fetch.items():
        if c == "FETCH_SIZE":
            kernels.setdefault(k, {}).update(FETCH_SIZE_KiB_avg=val, hbm_bytes_per_launch=int(val * 1024 * 2), launches=calls, avg_us=us)
    for (k, c), (calls, val, us) in write.items():
        if c == "WRITE_SIZE":
            kernels.setdefault(k, {}).update(WRITE_SIZE_KiB_avg=val)
    for (k, c), (calls, val, us) in mfma.items():
        kernels.setdefault(k, {})[c + "_avg"] = val
    # in-situ durations: the average launch time of every kernel INSIDE the eager bs-1 decode step (kernel trace), next to the
    # isolated-launch timings bench.py measures itself
    in_situ = {}
    tp = os.path.join(G, "prof_trace_bs1.summary.txt")
    if os.path.exists(tp):
        for line in open(tp):
            m = re.match(r"^(.{70,}?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", line.rstrip("\n"))
            if m:
                in_situ[re.sub(r",\s+", ",", m.group(1).strip())] = {"calls": int(m.group(2)), "avg_us": float(m.group(4))}
    # kernel names as bench.py spells them (no spaces inside the template list)
    kernels = {re.sub(r",\s+", ",", k): v for k, v in kernels.items()}
    json.dump({"source": "profiles/r06_pmc_*_decode_bs1.txt (rocprofv3 --pmc, one counter set per pass)", "lib_sha16": sha, "src_sha16": src_sha,
               "correction": "hbm_bytes_per_launch = FETCH_SIZE[KiB] * 1024 * 2 (gfx950 reports half of a wide coalesced read: MI355X_MICROARCH.md HBM section)",
               "kernels": kernels, "in_situ_decode_bs1_kernel_trace": in_situ}, open(os.path.join(P, "r06_pmc.json"), "w"), indent=1)
    print("wrote r06_pmc.json with", len(kernels), "kernels; lib", sha)


if __name__ == "__main__":
    main()
