"""GPU test of the drop-in `runner` process: this test plays the reference's ENGINE process in the reference's
ORDER (src/core/engine.rs:300-378: accept, `ready`, Init as JSON -> InitAck, KV plan, UsableMemoryLeft(EngineConfig) as JSON ->
InitAck; src/utils/heartbeat.rs: the command channel; :844-892 RunPrefill / RunDecode) over the reference's wire format —
abstract-namespace Unix socket, `ready` line, JSON `Init`, bincode afterwards, 1-byte acks (vllm_rs_amd/wire.py) — against
the native `vra_runner` binary (vllm_rs_amd/host/runner_main.cpp: C++ over the C ABI, what the reference's engine would
spawn) and against its Python twin `python -m vllm_rs_amd.runner_ipc`, both loading an HF checkpoint directory from disk
(f16 scales in a bf16 model, as AutoGPTQ writes them).  Tokens must be the oracle's greedy tokens."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from oracle import model as om
from oracle import oracle as orc
from tests.test_checkpoint import write_safetensors
from tests.test_gpu_engine import small_cfg
from vllm_rs_amd import runner_ipc, wire

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


RUNNER_BIN = os.path.join(ROOT, "vllm_rs_amd", "vra_runner")


@pytest.mark.parametrize("which", ["native", "python"])
def test_reference_engine_protocol_drives_the_runner_process(tmp_path, which):
    cfg = small_cfg(quant_method="gptq")
    w = om.make_random_checkpoint(cfg, 13)
    # scales travel as F16 (checkpoints store them so even for bf16 models, wna16.rs:97-109); the synthetic bf16 scales lie in
    # f16's normal range, so the f16 copy is exact and the runner's f16 -> bf16 cast must give back the oracle's bits
    def disk(k, a):
        if a.dtype != np.uint16:
            return (a.view(np.int32), "i32")
        if k.endswith(".scales"):
            f = (a.astype(np.uint32) << 16).view(np.float32)
            h = f.astype(np.float16)
            assert (h.astype(np.float32) == f).all()
            return (h.view(np.uint16), "f16")
        return (a, "bf16")
    on_disk = {k: disk(k, a) for k, a in w.items()}
    write_safetensors(tmp_path / "model.safetensors", on_disk)
    hf = dict(architectures=["LlamaForCausalLM"], hidden_size=cfg["hidden_size"], intermediate_size=cfg["intermediate_size"],
              num_hidden_layers=cfg["num_layers"], num_attention_heads=cfg["num_heads"], num_key_value_heads=cfg["num_kv_heads"],
              head_dim=cfg["head_dim"], vocab_size=cfg["vocab_size"], max_position_embeddings=cfg["max_position_embeddings"],
              rms_norm_eps=cfg["rms_norm_eps"], rope_theta=cfg["rope_theta"], torch_dtype="bfloat16", tie_word_embeddings=False,
              quantization_config=dict(quant_method="gptq", bits=4, group_size=128, desc_act=False, sym=True))
    json.dump(hf, open(tmp_path / "config.json", "w"))
    name = f"vra-ipc-{os.getpid()}"
    srv = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
    srv.bind("\0" + name)
    srv.listen(1)
    srv.settimeout(180)
    env = dict(os.environ, PYTHONPATH=ROOT)
    name += "-" + which
    srv.close()
    srv = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
    srv.bind("\0" + name)
    srv.listen(1)
    srv.settimeout(180)
    cmd = [RUNNER_BIN] if which == "native" else [sys.executable, "-m", "vllm_rs_amd.runner_ipc"]
    uuid = f"t{os.getpid()}{which}"
    # the engine's heartbeat command channel (heartbeat.rs:8-78, command.rs:91-172): "command_{uuid}@vllm-rs-runner-heartbeat.sock"
    hb_srv = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
    hb_srv.bind("\0command_" + uuid + "@vllm-rs-runner-heartbeat.sock")
    hb_srv.listen(1)
    hb_srv.settimeout(30)
    proc = subprocess.Popen(cmd + ["--sock", name, "--uuid", uuid], cwd=ROOT, env=env)
    hb = None
    try:
        conn, _ = srv.accept()
        conn.settimeout(180)
        assert wire._recv_exact(conn, 6) == b"ready\n"
        if which == "native":  # the C++ runner runs the reference's heartbeat worker; the Python twin (test scaffolding) does not
            hb, _ = hb_srv.accept()
            hb.settimeout(30)
            assert wire._recv_exact(hb, 6) == b"ready\n"
            heartbeat = bytes([4, 0, 0, 0, 14, 0, 0, 0])  # golden frame: length 4 | bincode variant 14 = MessageType::Heartbeat
            assert wire.encode(("Heartbeat", None)) == heartbeat[4:]
            hb.sendall(heartbeat)
            assert wire._recv_exact(hb, 1) == b"\x01"
        # Init: num_blocks is the reference's placeholder (config.rs:458) — the cache must NOT be sized from it
        init = dict(rank=0, dev_id=0, num_shards=1, model_type="LLaMa", dtype="BF16", is_gguf=False, is_rope_i=False,
                    config=dict(hf, num_hidden_layers=cfg["num_layers"]), econfig=dict(block_size=64, max_num_seqs=8, num_blocks=128, max_model_len=512, seed=5),
                    model_pathes=dict(config_filename=str(tmp_path / "config.json"), filenames=[str(tmp_path / "model.safetensors")]))
        wire.send_frame(conn, wire.encode_init_json(init))
        # InitAck #1 (model loaded), hand-assembled: u32 length 5 | variant 1 | bool true; the ack byte goes back by hand
        assert wire._recv_exact(conn, 9) == bytes([5, 0, 0, 0, 1, 0, 0, 0, 1])
        conn.sendall(b"\x01")
        # the engine's KV plan: MessageType::UsableMemoryLeft(EngineConfig) as JSON, fields as serde writes them (config.rs:285-328)
        ecfg = dict(model_id=None, weight_path=str(tmp_path), weight_file=None, enforce_parser=None, hf_token=None, hf_token_path=None, num_blocks=32,
                    kv_fraction=0.5, mamba_fraction=None, cpu_mem_fold=0.5, kvcache_memory_bytes=32 * 2 * 2 * 64 * 64 * 2 * 2, mamba_memory_bytes=0,
                    mamba_slot_bytes=0, mamba_cache_capacity=None, block_size=64, max_num_seqs=8, max_num_batched_tokens=2048, config_model_len=512,
                    max_model_len=512, max_tokens=None, isq=None, num_shards=1, device_ids=[0], generation_cfg=None, seed=5, prefix_cache=False,
                    prefix_cache_max_tokens=None, fp8_kvcache=False, server_mode=False, pd_config=None, mcp_command=None, mcp_config=None, mcp_args=None,
                    tool_prompt_template=None, pd_server_prefix_cache_ratio=None, pd_client_prefix_cache_ratio=None, yarn_scaling_factor=None,
                    disable_reasoning=False)
        frame = wire.encode_usable_memory_left_json(ecfg)
        assert frame.startswith(b'{"UsableMemoryLeft": {"model_id": null')
        wire.send_frame(conn, frame)
        assert wire._recv_exact(conn, 9) == bytes([5, 0, 0, 0, 1, 0, 0, 0, 1])  # InitAck #2: cache sized, (graphs captured)
        conn.sendall(b"\x01")
        oracle = om.OracleModel(cfg, w, num_blocks=32)
        greedy = dict(temperature=0.0)
        a = dict(id=1, token_ids=list(range(5, 75)), block_table=[3, 4], num_cached_tokens=0, sampling_params=greedy, status="Running")
        b = dict(id=2, token_ids=list(range(100, 130)), block_table=[7], num_cached_tokens=0, sampling_params=greedy, status="Running")
        wire.send_frame(conn, wire.encode(("RunPrefill", ([a, b], True))))
        _, toks = wire.decode(wire.recv_frame(conn))
        ref = oracle.forward(*runner_ipc.step_inputs_prefill([a, b], 64))
        want = orc.argmax_f32(ref).tolist()
        gaps = [np.sort(r)[-1] - np.sort(r)[-2] for r in ref]
        assert all(t == wv or g < 0.04 for t, wv, g in zip(toks, want, gaps)), (toks, want, gaps)
        seqs, tables = [a["token_ids"] + [want[0]], b["token_ids"] + [want[1]]], [[3, 4], [7]]
        for _ in range(3):
            ds = [dict(id=i + 1, last_token=s[-1], len=len(s), last_block_tokens=len(s) - (len(t) - 1) * 64, block_table_last=t[-1], block_tables=t,
                       sampling_params=greedy) for i, (s, t) in enumerate(zip(seqs, tables))]
            wire.send_frame(conn, wire.encode(("RunDecode", (ds, False))))
            _, toks = wire.decode(wire.recv_frame(conn))
            ref = oracle.forward(*runner_ipc.step_inputs_decode(ds, 64))
            want = orc.argmax_f32(ref).tolist()
            gaps = [np.sort(r)[-1] - np.sort(r)[-2] for r in ref]
            assert all(t == wv or g < 0.04 for t, wv, g in zip(toks, want, gaps)), (toks, want, gaps)
            for s, t in zip(seqs, want):
                s.append(t)
        # MessageType::KVCacheSwap (runner.rs:297-312, ModelRunner::swap_kvcache): sequence b's block 7 goes out to CPU block 0
        # and comes back into GPU block 9; decoding continues from the moved block exactly as the oracle does from the old one
        wire.send_frame(conn, wire.encode(("KVCacheSwap", ({7: 0}, False))))
        assert wire.decode(wire.recv_frame(conn)) == ("KVCacheSwapResponse", True)
        wire.send_frame(conn, wire.encode(("KVCacheSwap", ({0: 9}, True))))
        assert wire.decode(wire.recv_frame(conn)) == ("KVCacheSwapResponse", True)
        moved = [[3, 4], [9]]
        ds = [dict(id=i + 1, last_token=s[-1], len=len(s), last_block_tokens=len(s) - (len(t) - 1) * 64, block_table_last=t[-1], block_tables=t,
                   sampling_params=greedy) for i, (s, t) in enumerate(zip(seqs, moved))]
        wire.send_frame(conn, wire.encode(("RunDecode", (ds, False))))
        _, toks = wire.decode(wire.recv_frame(conn))
        ds_ref = [dict(d, block_table_last=t[-1], block_tables=t) for d, t in zip(ds, tables)]
        ref = oracle.forward(*runner_ipc.step_inputs_decode(ds_ref, 64))
        want = orc.argmax_f32(ref).tolist()
        gaps = [np.sort(r)[-1] - np.sort(r)[-2] for r in ref]
        assert all(t == wv or g < 0.04 for t, wv, g in zip(toks, want, gaps)), ("after the swap", toks, want, gaps)
        wire.send_frame(conn, wire.encode(("KVCacheSwap", ({31: 99}, False))))   # CPU block 99 does not exist: refused, not executed
        assert wire.decode(wire.recv_frame(conn)) == ("KVCacheSwapResponse", False)
        # the cache has the NEGOTIATED 32 blocks (cpu_mem_fold 0.5 -> 16 CPU blocks), not Init's placeholder 128: a block id past it
        # never reaches the device — the step is answered with an empty RunResponse, as a failed step in the reference
        bad = [dict(ds[0], block_table_last=40, block_tables=[40])]
        wire.send_frame(conn, wire.encode(("RunDecode", (bad, False))))
        assert wire.decode(wire.recv_frame(conn)) == ("RunResponse", [])
        wire.send_frame(conn, wire.encode(("KVCacheSwap", ({5: 15}, False))))   # CPU block 15 exists (32 x 0.5 = 16), 16 does not
        assert wire.decode(wire.recv_frame(conn)) == ("KVCacheSwapResponse", True)
        wire.send_frame(conn, wire.encode(("KVCacheSwap", ({5: 16}, False))))
        assert wire.decode(wire.recv_frame(conn)) == ("KVCacheSwapResponse", False)
        # a message this path does not serve is logged and NOT answered (runner.rs:246-430): the stream stays in step
        import struct
        wire.send_frame(conn, struct.pack("<IQ", wire.VIDX["CheckPrefillStatus"], 3))
        wire.send_frame(conn, wire.encode(("ClearBlocks", [1, 2])))
        assert wire.decode(wire.recv_frame(conn)) == ("ClearBlocksResponse", True)
        wire.send_frame(conn, wire.encode(("FinishDecode", 1)))
        wire.send_frame(conn, wire.encode(("Shutdown", None)))
        assert proc.wait(60) == 0
    finally:
        if proc.poll() is None:
            proc.kill()
        srv.close()
        hb_srv.close()
        if hb:
            hb.close()
