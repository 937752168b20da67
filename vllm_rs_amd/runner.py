"""Tensor-parallel launcher: one runner process per rank, driven by the engine process.

Restates the reference's engine <-> runner split for this path (src/runner/mod.rs:25-121 `RunnerInitRequest` /
`MessageType`, src/runner/runner.rs:20-125 runner main, src/core/engine.rs:187-330 spawn + Init + InitAck):
  * the engine process creates the 128-byte RCCL unique id and ships it in `Init` together with rank, device,
    world size and the configs; every runner builds its communicator from it (`Comm::from_rank`, runner.rs:80-89);
  * for the one-shot transport every runner exports a 64-byte IPC handle of its exchange region; the engine process
    gathers the W handles and sends the table back (same hand-off, one more round);
  * every runner reports its KV plan, the engine process takes the minimum (`UsableMemoryLeft`, mod.rs:277) so that
    all ranks allocate the same number of blocks, and the runners answer `InitAck`;
  * afterwards the engine process broadcasts work (`RunPrefill` / `RunDecode` in the reference; here the requests
    themselves: every rank runs the same deterministic scheduler in lock step, the all-reduces keep them together)
    and collects `RunResponse` from every rank — at temperature 0 all ranks must agree (Appendix A21).

Transport "auto": RCCL + one-shot when every rank has its own GPU, one-shot only when ranks share a GPU (RCCL refuses
two ranks on one device) — the latter is how the TP product path runs on a single-GPU box.

Messages travel over multiprocessing pipes as (tag, payload) tuples; the reference's byte-level wire format
(u32-LE length + bincode + 1-byte ack) is the subject of vllm_rs_amd/wire.py.
"""
import multiprocessing as mp
import os
import traceback

import numpy as np


def _runner_main(conn, rank):
    """one runner process (runner.rs main): Init -> communicator -> model shard -> InitAck -> serve requests"""
    try:
        import ctypes as C

        from . import _lib
        from .engine import Engine
        tag, init = conn.recv()
        assert tag == "Init"
        if init.get("shared_device"):
            # Ranks that SHARE a GPU (the single-GPU stand-in of the TP tests): every HIP process opens up to four hardware queues;
            # nine processes oversubscribe the device's queue slots, the scheduler then has to rotate queues — and a queue whose
            # kernel spins in the one-shot exchange waiting for a peer does not yield: the peer's queue can stay unmapped until the
            # bounded wait expires (tools/tp8_stress.py: "expected epoch 26, its flag read 25" after 62 clean forwards).  One
            # hardware queue per runner keeps all ranks' queues mapped at once.  Real TP (one rank per GPU) is not affected.
            os.environ.setdefault("GPU_MAX_HW_QUEUES", init.get("hw_queues", "1"))
        L = _lib.load()
        dev, world = init["device"], init["world"]
        L.vra_set_device(dev)
        comm = None
        if init["transport"] in ("rccl", "both"):
            idb = (C.c_uint8 * 128).from_buffer_copy(init["nccl_id"])
            comm = L.vra_comm_create(idb, rank, world, dev)
            if not comm:
                raise RuntimeError("vra_comm_create: " + L.vra_last_error().decode())
        if init["transport"] in ("ipc", "both"):
            h = (C.c_uint8 * 64)()
            comm = L.vra_comm_ipc_begin(comm, rank, world, dev, h)
            if not comm:
                raise RuntimeError("vra_comm_ipc_begin: " + L.vra_last_error().decode())
            conn.send(("IpcHandle", bytes(h)))
            tag, table = conn.recv()
            assert tag == "IpcTable" and len(table) == 64 * world
            tb = (C.c_uint8 * len(table)).from_buffer_copy(table)
            if L.vra_comm_ipc_connect(comm, tb, init.get("oneshot_max_bytes", 0)) != 0:
                raise RuntimeError("vra_comm_ipc_connect: " + L.vra_last_error().decode())
        kw = dict(init["engine_kw"])
        eng = Engine(init["cfg"], tp_rank=rank, tp_world_size=world, device=dev, comm=comm, **kw)
        if init["tensors"] is None:
            eng.init_synthetic(finalize=False)
        else:
            eng.load_weights(init["tensors"], finalize=False)
        conn.send(("UsableBlocks", int(eng.plan_kv_blocks())))
        tag, nb = conn.recv()
        assert tag == "NumBlocks"
        eng.set_num_gpu_blocks(nb).finalize()
        if init.get("snapshots"):
            eng.tp_snapshots(True)
        conn.send(("InitAck", True))
        while True:
            tag, payload = conn.recv()
            if tag == "Shutdown":
                break
            try:
                if tag == "ForwardRaw":
                    conn.send(("Logits", eng.forward_raw(*payload)))
                elif tag == "Generate":
                    prompts, kwargs = payload
                    conn.send(("Tokens", [o.tolist() for o in eng.generate(prompts, **kwargs)]))
                elif tag == "TimedDecode":
                    prompts, warm, steps = payload
                    rids = [eng.add_request(p, max_tokens=warm + steps + 8, ignore_eos=True) for p in prompts]
                    while True:
                        n, is_prefill = eng.step()
                        if not is_prefill and n == len(prompts):
                            break
                    for _ in range(warm):
                        eng.step()
                    ms = eng.timed_decode(steps)
                    outs = [eng.output(r).tolist() for r in rids]
                    while eng.has_unfinished():
                        eng.step()
                    conn.send(("Timed", (ms, outs)))
                elif tag == "AllReduce":  # the collective on its own (parity tests of comm.hip), on the engine's stream
                    conn.send(("Reduced", _all_reduce_once(L, comm, eng, rank, payload)))
                elif tag == "Snapshots":  # layer 0's stages of the last forward (parity instrumentation)
                    conn.send(("Stages", eng.read_tp_snapshots()))
                elif tag == "NumBlocksQuery":
                    conn.send(("NumBlocks", eng.num_gpu_blocks))
                else:
                    conn.send(("Error", f"unknown message {tag}"))
            except Exception:  # noqa: BLE001 - reported to the engine process (MessageType::Error)
                conn.send(("Error", traceback.format_exc()))
        eng.close()
        L.vra_comm_destroy(comm)
        conn.send(("Bye", rank))
    except Exception:  # noqa: BLE001
        try:
            conn.send(("Error", traceback.format_exc()))
        except Exception:  # noqa: BLE001
            pass


def _all_reduce_once(L, comm, eng, rank, p):
    """p: dict(data=[array per rank] (uint16 bit patterns or float32), dtype, bias, residual, reps) -> this rank's result"""
    import ctypes as C
    x = np.ascontiguousarray(p["data"][rank])
    st = L.vra_engine_stream(eng.h)
    nbytes = x.nbytes
    d_src, d_dst = L.vra_malloc(nbytes), L.vra_malloc(nbytes)
    d_bias = d_res = None
    if p.get("bias") is not None:
        b = np.ascontiguousarray(p["bias"])
        d_bias = L.vra_malloc(b.nbytes)
        L.vra_memcpy_h2d(d_bias, b.ctypes.data_as(C.c_void_p), b.nbytes, st)
    if p.get("residual") is not None:
        r = np.ascontiguousarray(p["residual"])
        d_res = L.vra_malloc(r.nbytes)
        L.vra_memcpy_h2d(d_res, r.ctypes.data_as(C.c_void_p), r.nbytes, st)
    out = np.empty_like(x)
    for _ in range(p.get("reps", 1)):  # repeated launches walk the double-buffered slots and the per-slice epochs
        L.vra_memcpy_h2d(d_src, x.ctypes.data_as(C.c_void_p), nbytes, st)
        if d_bias or d_res:
            rows, cols = x.shape
            L.vra_all_reduce_fused(comm, d_src, d_dst, d_bias, d_res, rows, cols, p["dtype"], st)
        else:
            L.vra_all_reduce(comm, d_src, d_dst, x.size, p["dtype"], st)
        L.vra_memcpy_d2h(out.ctypes.data_as(C.c_void_p), d_dst, nbytes, st)
        L.vra_stream_sync(st)
    err = L.vra_last_error().decode()
    timed_out = L.vra_comm_take_error(comm)
    for d in (d_src, d_dst, d_bias, d_res):
        if d:
            L.vra_free(d)
    if err or timed_out:
        raise RuntimeError(f"all_reduce: {err or 'one-shot exchange timed out'}")
    return out


class TPEngine:
    """engine-process side: spawns `world` runners and talks to them (engine.rs:187-330, 844-892)"""

    def __init__(self, cfg, world, *, devices=None, transport="auto", tensors=None, oneshot_max_bytes=0, timeout=600, snapshots=False,
                 **engine_kw):
        from . import _lib
        L = _lib.load()
        ndev = L.vra_device_count()
        if ndev <= 0:
            raise RuntimeError("TPEngine needs a GPU (there is no CPU fallback)")
        if devices is None:
            devices = [r if ndev >= world else 0 for r in range(world)]
        shared = len(set(devices)) < world
        if transport == "auto":
            transport = "ipc" if shared else "both"
        if shared and transport != "ipc":
            raise ValueError("ranks sharing a GPU can only use the one-shot (ipc) transport: RCCL refuses duplicate devices")
        self.world, self.timeout, self.transport = world, timeout, transport
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        nccl_id = None
        if transport in ("rccl", "both"):
            import ctypes as C
            idb = (C.c_uint8 * 128)()
            if L.vra_comm_unique_id(idb) != 0:
                raise RuntimeError("vra_comm_unique_id: " + L.vra_last_error().decode())
            nccl_id = bytes(idb)
        ctx = mp.get_context("spawn")
        self.conns, self.procs = [], []
        for r in range(world):
            a, b = ctx.Pipe()
            p = ctx.Process(target=_runner_main, args=(b, r), daemon=True)
            p.start()
            self.conns.append(a)
            self.procs.append(p)
        for r, c in enumerate(self.conns):
            c.send(("Init", dict(rank=r, device=devices[r], world=world, transport=transport, nccl_id=nccl_id, cfg=cfg,
                                 engine_kw=engine_kw, tensors=tensors, oneshot_max_bytes=oneshot_max_bytes, snapshots=snapshots,
                                 shared_device=shared, hw_queues=os.environ.get("VRA_TP_SHARED_HW_QUEUES", "1"))))
        if transport in ("ipc", "both"):
            table = b"".join(self._expect(c, "IpcHandle") for c in self.conns)
            for c in self.conns:
                c.send(("IpcTable", table))
        plans = [self._expect(c, "UsableBlocks") for c in self.conns]
        nb = engine_kw.get("num_gpu_blocks") or min(plans)
        self.plans = plans
        for c in self.conns:
            c.send(("NumBlocks", int(nb)))
        for c in self.conns:
            self._expect(c, "InitAck")

    def _expect(self, c, tag):
        if not c.poll(self.timeout):
            raise TimeoutError(f"runner did not answer with {tag} within {self.timeout}s")
        t, payload = c.recv()
        if t == "Error":
            raise RuntimeError("runner error:\n" + payload)
        if t != tag:
            raise RuntimeError(f"expected {tag}, got {t}")
        return payload

    def _all(self, tag, payload, reply):
        """broadcast, then collect EVERY rank's answer before judging them: when a step fails, what each rank saw (which peer it
        waited for, at which epoch) is the evidence — the first rank's error alone hides the waiting graph"""
        for c in self.conns:
            c.send((tag, payload))
        got, errs = [], []
        for r, c in enumerate(self.conns):
            if not c.poll(self.timeout):
                errs.append(f"rank {r}: no answer within {self.timeout}s")
                got.append(None)
                continue
            t, pl = c.recv()
            if t != reply:
                errs.append(f"rank {r}: {pl.strip().splitlines()[-1] if t == 'Error' and isinstance(pl, str) else f'expected {reply}, got {t}'}")
                got.append(None)
            else:
                got.append(pl)
        if errs:
            ok = [r for r, g in enumerate(got) if g is not None]
            raise RuntimeError("runner error(s) in " + tag + ":\n" + "\n".join(errs) + (f"\n(ranks {ok} answered normally)" if ok else ""))
        return got

    def forward_raw(self, *args):
        """-> the f32 logits of every rank (A21: they must be identical)"""
        return self._all("ForwardRaw", args, "Logits")

    def generate(self, prompts, **kw):
        return self._all("Generate", ([np.asarray(p, np.uint32) for p in prompts], kw), "Tokens")

    def timed_decode(self, prompts, warmup, steps):
        """-> [(gpu ms of `steps` decode steps, outputs)] per rank"""
        return self._all("TimedDecode", ([np.asarray(p, np.uint32) for p in prompts], warmup, steps), "Timed")

    def all_reduce(self, data, dtype, bias=None, residual=None, reps=1):
        """data: one array per rank -> every rank's reduced array"""
        return self._all("AllReduce", dict(data=data, dtype=dtype, bias=bias, residual=residual, reps=reps), "Reduced")

    def snapshots(self):
        """per rank: stage name -> bit patterns of layer 0's stages of the last forward (needs snapshots=True)"""
        return self._all("Snapshots", None, "Stages")

    def num_gpu_blocks(self):
        return self._all("NumBlocksQuery", None, "NumBlocks")

    def close(self):
        for c in self.conns:
            try:
                c.send(("Shutdown", None))
            except Exception:  # noqa: BLE001
                pass
        for p in self.procs:
            p.join(timeout=30)
            if p.is_alive():
                p.terminate()
        self.conns, self.procs = [], []

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
