"""The two drop-in-path legs of bench.py on their own (ffi_step, runner_ipc_step): quick check / A-B aid."""
import json
import os
import sys

sys.path.insert(0, os.getcwd())
import bench
from vllm_rs_amd import _lib, engine as E

L = _lib.load()
cfg = dict(E.LLAMA3_8B)
which = sys.argv[1:] or ["ffi", "ipc"]
if "ffi" in which:
    pass
    print(json.dumps({"ffi_step": bench.ffi_step(L, cfg)}))
if "ipc" in which:
    print(json.dumps({"runner_ipc_step": bench.runner_ipc_step(cfg)}))
