"""A `runner` process that speaks the reference's engine <-> runner protocol (vllm_rs_amd/wire.py; src/runner/runner.rs
main loop, src/runner/mod.rs): connect to the engine's local socket, announce `ready`, take `Init` (JSON), load the model shard,
answer `InitAck`, take `UsableMemoryLeft(EngineConfig)` (JSON: the engine's KV plan, src/core/engine.rs:355-378), size the cache
from it, answer `InitAck` again (src/core/runner.rs:443-455, src/runner/runner.rs:214-236), then serve `RunPrefill` / `RunDecode`
with `RunResponse` token ids until `Shutdown`.

    python -m vllm_rs_amd.runner_ipc --sock <name> [--uuid <id>]

`--sock` is the name the reference passes to its runner binary; `interprocess`'s GenericNamespaced names are Linux
abstract-namespace Unix sockets, i.e. the address "\\0<name>".

In the reference the ENGINE process owns the scheduler and the block manager; the runner only builds the step's
InputMetadata from the sequences it is handed (ModelRunner::prepare_prefill / prepare_decode, src/core/runner.rs:978-1388)
and runs the forward pass + sampling.  The same split here: `step_inputs_*` restate that metadata arithmetic on the wire
structs, the forward pass is `Engine.forward_raw` (the native runtime, no scheduler involved).
"""
import argparse
import os
import socket

import numpy as np

from . import wire

CHUNK = 8192  # scheduler.rs:203 / runner.rs:984


def step_inputs_prefill(seqs, block_size):
    """ModelRunner::prepare_prefill (runner.rs:978-1241) on wire `Sequence`s -> (ids, positions, slot_mapping, block_tables,
    context_lens, cu_seqlens_q)"""
    ids, pos, slots, cu, ctx = [], [], [], [0], []
    max_bt = max(len(s["block_table"]) for s in seqs)
    bt = np.zeros((len(seqs), max_bt), np.uint32)
    for b, s in enumerate(seqs):
        cached, toks = s["num_cached_tokens"], s["token_ids"]
        n = min(CHUNK, len(toks) - cached)
        ids += toks[cached:cached + n]
        pos += range(cached, cached + n)
        slots += [int(s["block_table"][p // block_size]) * block_size + p % block_size for p in range(cached, cached + n)]
        cu.append(len(ids))
        ctx.append(cached + n)
        bt[b, :len(s["block_table"])] = s["block_table"]
    return (np.array(ids, np.uint32), np.array(pos, np.int64), np.array(slots, np.int64), bt, np.array(ctx, np.uint32), np.array(cu, np.uint32))


def step_inputs_decode(seqs, block_size):
    """ModelRunner::prepare_decode (runner.rs:1243-1388) on wire `DecodeSequence`s; slot = block_table_last * BS +
    last_block_tokens - 1 (runner.rs:1259-1262)"""
    max_bt = max(len(s["block_tables"]) for s in seqs)
    bt = np.zeros((len(seqs), max_bt), np.uint32)
    for b, s in enumerate(seqs):
        bt[b, :len(s["block_tables"])] = s["block_tables"]
    return (np.array([s["last_token"] for s in seqs], np.uint32), np.array([s["len"] - 1 for s in seqs], np.int64),
            np.array([int(s["block_table_last"]) * block_size + s["last_block_tokens"] - 1 for s in seqs], np.int64), bt,
            np.array([s["len"] for s in seqs], np.uint32), None)


def strategy_of(sp):
    """LogitsProcessor::get_strategy + the runner's defaults (runner.rs:1436-1497): None => greedy"""
    sp = sp or {}
    t, k, p = sp.get("temperature"), sp.get("top_k"), sp.get("top_p")
    if t is not None and t == 0.0:
        return None
    has_user = t is not None or (k or 0) > 0 or (p is not None and 0.0 < p < 1.0)
    if not has_user:
        return dict(k=32, p=0.95, t=0.7)  # no generation config: the reference's default (A4)
    if t is None or t < 1e-7:
        return None
    return dict(k=k if (k or 0) > 0 else 0, p=p if p is not None else -1.0, t=t)


class RunnerServer:
    """the message loop of runner.rs:246-430 around a forward function `forward(ids, pos, slots, bt, ctx, cu_q) -> f32 logits [B, V]`
    and a sampler `sample(logits, strategy) -> token ids`"""

    def __init__(self, sock, forward, sample, block_size=64, swap=None):
        self.sock, self.forward, self.sample, self.BS, self.swap = sock, forward, sample, block_size, swap
        self.cached_strategy = "unset"

    def _run(self, seqs, is_prefill):
        try:
            return self._run_checked(seqs, is_prefill)
        except Exception as e:  # a runner error is answered with an empty RunResponse (runner.rs:246-292)
            print(f"runner_ipc: step failed: {e}", flush=True)
            return []

    def _run_checked(self, seqs, is_prefill):
        if is_prefill:
            inp = step_inputs_prefill(seqs, self.BS)
            self.cached_strategy = strategy_of(seqs[0]["sampling_params"])  # cached for the decode steps (A3)
        else:
            inp = step_inputs_decode(seqs, self.BS)
        logits = self.forward(*inp)
        strat = self.cached_strategy if self.cached_strategy != "unset" else dict(k=32, p=0.95, t=0.7)
        return [int(t) for t in self.sample(logits, strat)]

    def serve(self):
        while True:
            try:
                name, p = wire.decode(wire.recv_frame(self.sock))
            except wire.WireError as e:  # logged, not answered (runner.rs:246-430)
                print(f"runner_ipc: undecodable or unserved frame ({e}): ignored", flush=True)
                continue
            if name == "Shutdown":
                return
            if name == "RunPrefill":
                wire.send_frame(self.sock, wire.encode(("RunResponse", self._run(p[0], True))))
            elif name == "RunDecode":
                wire.send_frame(self.sock, wire.encode(("RunResponse", self._run(p[0], False))))
            elif name in ("FinishDecode", "LoadingProgress", "Heartbeat"):
                pass  # runner.rs:294-315: bookkeeping only, no reply
            elif name == "ClearBlocks":
                wire.send_frame(self.sock, wire.encode(("ClearBlocksResponse", True)))
            elif name == "KVCacheSwap":  # runner.rs:297-312 -> ModelRunner::swap_kvcache
                ok = bool(self.swap(p[0], p[1])) if self.swap else False
                wire.send_frame(self.sock, wire.encode(("KVCacheSwapResponse", ok)))
            else:  # the reference only logs what it does not serve (a reply would be read as the ack of the engine's next frame)
                print(f"runner_ipc: message {name} is not served on this path: ignored", flush=True)


def connect(sock_name):
    s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
    s.connect("\0" + sock_name)
    s.sendall(b"ready\n")  # runner.rs:57
    return s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sock", required=True)
    ap.add_argument("--uuid", default="")
    a = ap.parse_args()
    sock = connect(a.sock)
    req = wire.decode_init_json(wire.recv_frame(sock))
    import ctypes as C

    from . import _lib
    from .engine import Engine
    L = _lib.load()
    cfg = wire.model_cfg_from_init(req)
    ec = req.get("econfig") or {}
    rank, world, dev = req["rank"], req["num_shards"], req["dev_id"]
    L.vra_set_device(dev)
    comm = None
    if world > 1:
        idb = (C.c_uint8 * 128).from_buffer_copy(req["nccl_id"])
        comm = L.vra_comm_create(idb, rank, world, dev)  # Comm::from_rank (runner.rs:80-89)
        if not comm:
            raise RuntimeError("vra_comm_create: " + L.vra_last_error().decode())
    kw = dict(block_size=ec.get("block_size", 64), max_num_seqs=ec.get("max_num_seqs", 32), max_model_len=ec.get("max_model_len") or 0,
              num_gpu_blocks=ec.get("num_blocks", 0), enable_prefix_cache=False, use_graph=False, tp_rank=rank, tp_world_size=world, device=dev,
              seed=ec.get("seed") or 1234, comm=comm, fp8_kvcache=bool(ec.get("fp8_kvcache")),
              cpu_mem_fold=ec.get("cpu_mem_fold") if ec.get("cpu_mem_fold") is not None else 0.2)  # kvcache_allocator.rs:317
    paths = req.get("model_pathes") or {}
    cfg_file = paths.get("config_filename")
    if cfg_file and os.path.exists(cfg_file):
        eng = Engine.from_pretrained(os.path.dirname(cfg_file), finalize=False, **kw)
    else:  # no checkpoint on this box: synthetic weights of the configured shape (bench mode)
        eng = Engine(cfg, **kw).init_synthetic(finalize=False)
    eng.finalize_model()
    # the model is loaded: InitAck #1, then the engine's KV plan (JSON), then the cache, then InitAck #2
    wire.send_frame(sock, wire.encode(("InitAck", True)))
    neg = wire.decode_usable_memory_left_json(wire.recv_frame(sock))
    if neg is not None:
        eng.update_config(num_gpu_blocks=neg.get("num_blocks", 0), max_num_seqs=neg.get("max_num_seqs", 0), max_model_len=neg.get("max_model_len") or 0,
                          cpu_mem_fold=neg.get("cpu_mem_fold") if neg.get("cpu_mem_fold") is not None else 0.2)
    eng.finalize()
    calls = [0]

    def sample(logits, strat):
        if strat is None:
            return np.argmax(logits, axis=-1)  # first maximal index, as candle's argmax
        B, V = logits.shape
        lg = np.ascontiguousarray(logits, np.float32)
        d_l, d_o = L.vra_malloc(lg.nbytes), L.vra_malloc(B * 4)
        L.vra_memcpy_h2d(d_l, lg.ctypes.data_as(C.c_void_p), lg.nbytes, 0)
        calls[0] += 1
        L.vra_sample(d_l, d_o, B, V, strat["k"], strat["p"], strat["t"], (kw["seed"] << 20) + calls[0], None, None, 0)
        out = np.empty(B, np.uint32)
        L.vra_memcpy_d2h(out.ctypes.data_as(C.c_void_p), d_o, B * 4, 0)
        L.vra_device_sync()
        L.vra_free(d_l), L.vra_free(d_o)
        return out
    wire.send_frame(sock, wire.encode(("InitAck", True)))  # InitAck #2: the cache exists

    def swap(mapping, swap_in):
        pairs = np.array([[k, v] for k, v in mapping.items()], np.int64).reshape(-1)
        return L.vra_engine_swap_blocks(eng.h, pairs.ctypes.data_as(C.c_void_p), len(mapping), int(bool(swap_in))) == 0
    RunnerServer(sock, eng.forward_raw, sample, kw["block_size"], swap).serve()
    eng.close()
    if comm:
        L.vra_comm_destroy(comm)


if __name__ == "__main__":
    main()
