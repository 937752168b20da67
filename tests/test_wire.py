"""CPU tests of the runner IPC wire format (src/runner/mod.rs:169-295, src/core/sequence.rs:32-62,
src/utils/config.rs:505-537).  The golden frames below are assembled BY HAND from the Rust struct definitions and bincode 1.x's
rules (little endian, fixed-width integers, u32 variant index, 1-byte Option tags, u64 lengths) with struct.pack — independently
of vllm_rs_amd/wire.py — and must match its output byte for byte; the message loop runs over a real abstract-namespace Unix
socket (what interprocess's GenericNamespaced is on Linux) against the CPU oracle as the model."""
import socket
import struct
import threading

import numpy as np
import pytest

from oracle import model as om
from oracle import oracle as orc
from vllm_rs_amd import runner_ipc, wire

u8 = lambda v: struct.pack("<B", v)
u32 = lambda v: struct.pack("<I", v)
u64 = lambda v: struct.pack("<Q", v)
i64 = lambda v: struct.pack("<q", v)
f32 = lambda v: struct.pack("<f", v)
none = u8(0)
some = lambda b: u8(1) + b
string = lambda s: u64(len(s)) + s.encode()
vec_u32 = lambda xs: u64(len(xs)) + b"".join(u32(x) for x in xs)


def golden_sampling_params(temperature, top_k, top_p, freq):
    """SamplingParams { temperature, max_tokens, ignore_eos, top_k, top_p, session_id, frequency_penalty, presence_penalty,
    stop_sequences, [stop_token_ids: serde(skip)], thinking, mcp_mode, grammar, grammar_json, reasoning_effort }"""
    return (some(f32(temperature)) + some(u64(77)) + u8(1) + some(i64(top_k)) + some(f32(top_p)) + none + some(f32(freq)) + none +
            some(u64(1) + string("</s>")) + none + none + none + none + none)


def test_variant_indices_follow_the_enum_declaration():
    assert wire.VARIANTS.index("Init") == 0 and wire.VARIANTS.index("InitAck") == 1 and wire.VARIANTS.index("RunPrefill") == 3
    assert wire.VARIANTS.index("RunDecode") == 4 and wire.VARIANTS.index("RunResponse") == 5 and wire.VARIANTS.index("FinishDecode") == 8
    assert wire.VARIANTS.index("Error") == 13 and wire.VARIANTS.index("KVCacheSwap") == 21 and wire.VARIANTS.index("ClearBlocks") == 31
    assert wire.VARIANTS.index("Shutdown") == 34 and len(wire.VARIANTS) == 35


def test_small_messages_golden_bytes():
    assert wire.encode(("InitAck", True)) == u32(1) + u8(1)
    assert wire.encode(("RunResponse", [5, 70000])) == u32(5) + u64(2) + u32(5) + u32(70000)
    assert wire.encode(("FinishDecode", 42)) == u32(8) + u64(42)
    assert wire.encode(("Shutdown", None)) == u32(34)
    assert wire.encode(("Heartbeat", None)) == u32(14)
    assert wire.encode(("Error", "boom")) == u32(13) + u64(4) + b"boom"
    assert wire.encode(("ClearBlocks", [1, 2, 3])) == u32(31) + vec_u32([1, 2, 3])
    assert wire.encode(("LoadingProgress", (3, 32))) == u32(2) + u64(3) + u64(32)
    assert wire.encode(("KVCacheSwap", ({7: 9}, True))) == u32(21) + u64(1) + u64(7) + u64(9) + u8(1)
    for m in [("InitAck", False), ("RunResponse", []), ("FinishDecode", 2 ** 40), ("Error", "x"), ("ClearBlocksResponse", True)]:
        assert wire.decode(wire.encode(m)) == m


def test_run_decode_golden_frame():
    sp = dict(temperature=0.5, max_tokens=77, ignore_eos=True, top_k=-1, top_p=0.9, frequency_penalty=1.5, stop_sequences=["</s>"])
    seq = dict(id=9, last_token=1234, len=130, last_block_tokens=2, block_table_last=17, block_tables=[4, 5, 17], sampling_params=sp)
    golden = (u32(4) + u64(1) + u64(9) + u32(1234) + u64(130) + u64(2) + u32(17) + vec_u32([4, 5, 17]) +
              golden_sampling_params(0.5, -1, 0.9, 1.5) + u8(0))
    assert wire.encode(("RunDecode", ([seq], False))) == golden
    name, (seqs, flag) = wire.decode(golden)
    assert name == "RunDecode" and flag is False and seqs[0]["block_tables"] == [4, 5, 17] and seqs[0]["sampling_params"]["top_k"] == -1
    assert abs(seqs[0]["sampling_params"]["top_p"] - 0.9) < 1e-7 and seqs[0]["sampling_params"]["stop_sequences"] == ["</s>"]


def test_run_prefill_golden_frame():
    sp = dict(temperature=0.0, max_tokens=77, ignore_eos=True, top_k=40, top_p=1.0, frequency_penalty=0.0, stop_sequences=["</s>"])
    seq = dict(id=3, created_time=1700000000000, swapped_time=None, status="Running", token_ids=[1, 2, 3, 4, 5], output_ids=[5], block_table=[8],
               num_cached_tokens=0, mamba_prefix_hash=None, last_token=5, block_size=64, sampling_params=sp, pd_first_token=None, images=None,
               is_tool_call_end=False, hit_stop_sequence=False, stop_sequence=None)
    golden = (u32(3) + u64(1) +
              u64(3) + u64(1700000000000) + none + u32(1) + vec_u32([1, 2, 3, 4, 5]) + vec_u32([5]) + vec_u32([8]) + u64(0) + none + u32(5) + u64(64) +
              golden_sampling_params(0.0, 40, 1.0, 0.0) + none + none + u8(0) + u8(0) + none +
              u8(1))
    assert wire.encode(("RunPrefill", ([seq], True))) == golden
    name, (seqs, flag) = wire.decode(golden)
    assert name == "RunPrefill" and flag is True and seqs[0]["token_ids"] == [1, 2, 3, 4, 5] and seqs[0]["status"] == "Running"


def test_malformed_frames_are_rejected():
    with pytest.raises(wire.WireError):
        wire.decode(u32(99))
    with pytest.raises(wire.WireError):
        wire.decode(u32(5) + u64(3) + u32(1))            # Vec<u32> shorter than its length
    with pytest.raises(wire.WireError):
        wire.decode(u32(1) + u8(1) + u8(0))              # trailing byte
    with pytest.raises(wire.WireError):
        wire.decode(u32(1) + u8(7))                      # invalid bool
    with pytest.raises(wire.WireError):
        wire.encode(("TransferPrefill", None))


def test_init_json_and_nccl_id():
    nid = bytes(range(128))
    req = dict(rank=1, dev_id=1, num_shards=2, model_type="LLaMa", dtype="BF16", is_gguf=False, is_rope_i=False, nccl_id=nid,
               config=dict(architectures=["LlamaForCausalLM"], head_dim=None, num_attention_heads=4, num_key_value_heads=2, max_position_embeddings=512,
                           hidden_size=256, num_hidden_layers=2, intermediate_size=512, rms_norm_eps=1e-5, vocab_size=512, rope_theta=10000.0,
                           quantization_config=dict(quant_method="gptq", bits=4, group_size=128)),
               econfig=dict(block_size=64, max_num_seqs=8, num_blocks=32, max_model_len=512, seed=7), model_pathes=dict(config_filename="/nowhere/config.json"))
    b = wire.encode_init_json(req)
    assert b.startswith(b'{"Init": {') and b"=" not in b      # externally tagged enum; base64 without padding (mod.rs:44)
    back = wire.decode_init_json(b)
    assert back["nccl_id"] == nid and back["rank"] == 1 and back["num_shards"] == 2
    cfg = wire.model_cfg_from_init(back)
    assert cfg["hidden_size"] == 256 and cfg["head_dim"] == 64 and cfg["quant_method"] == "gptq" and cfg["dtype"] == 0 and cfg["arch"] == "llama"


def test_message_loop_over_an_abstract_unix_socket_with_the_oracle_as_model():
    """engine side (this test) <-> runner side (RunnerServer in a thread) over "\\0<name>": ready line, framing with acks,
    RunPrefill (two sequences, one with a cached prefix chunk) and RunDecode steps; tokens must be the oracle's greedy tokens for
    the metadata the runner derives from the wire structs (runner.rs:978-1388)"""
    cfg = dict(arch="llama", hidden_size=128, intermediate_size=256, num_layers=1, num_heads=4, num_kv_heads=2, head_dim=32, vocab_size=300,
               max_position_embeddings=512, rms_norm_eps=1e-5, rope_theta=10000.0, quant_method="gptq", group_size=128, dtype=0)
    w = om.make_random_checkpoint(cfg, 1)
    name = f"vra-test-{np.random.default_rng().integers(1 << 30)}"
    srv = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
    srv.bind("\0" + name)
    srv.listen(1)

    def runner():
        s = runner_ipc.connect(name)
        model = om.OracleModel(cfg, w, num_blocks=16)
        RunnerServer = runner_ipc.RunnerServer
        RunnerServer(s, model.forward, lambda lg, strat: np.argmax(lg, -1), 64).serve()
        s.close()
    th = threading.Thread(target=runner, daemon=True)
    th.start()
    conn, _ = srv.accept()
    assert wire._recv_exact(conn, 6) == b"ready\n"
    greedy = dict(temperature=0.0)
    a = dict(id=1, token_ids=list(range(5, 75)), block_table=[3, 4], num_cached_tokens=0, sampling_params=greedy, status="Running")
    b = dict(id=2, token_ids=list(range(100, 130)), block_table=[7], num_cached_tokens=0, sampling_params=greedy, status="Running")
    wire.send_frame(conn, wire.encode(("RunPrefill", ([a, b], True))))
    name_, toks = wire.decode(wire.recv_frame(conn))
    assert name_ == "RunResponse" and len(toks) == 2
    ref = om.OracleModel(cfg, w, num_blocks=16)
    inp = runner_ipc.step_inputs_prefill([a, b], 64)
    assert inp[2].tolist()[:3] == [3 * 64, 3 * 64 + 1, 3 * 64 + 2] and inp[5].tolist() == [0, 70, 100]
    want = orc.argmax_f32(ref.forward(*inp)).tolist()
    assert toks == want
    seqs = [a["token_ids"] + [toks[0]], b["token_ids"] + [toks[1]]]
    tables = [[3, 4], [7]]
    for _ in range(3):
        ds = [dict(id=i + 1, last_token=s[-1], len=len(s), last_block_tokens=len(s) - (len(t) - 1) * 64, block_table_last=t[-1], block_tables=t,
                   sampling_params=greedy) for i, (s, t) in enumerate(zip(seqs, tables))]
        wire.send_frame(conn, wire.encode(("RunDecode", (ds, False))))
        _, toks = wire.decode(wire.recv_frame(conn))
        want = orc.argmax_f32(ref.forward(*runner_ipc.step_inputs_decode(ds, 64))).tolist()
        assert toks == want
        for s, t in zip(seqs, toks):
            s.append(t)
    wire.send_frame(conn, wire.encode(("FinishDecode", 1)))
    wire.send_frame(conn, wire.encode(("ClearBlocks", [3, 4])))
    assert wire.decode(wire.recv_frame(conn)) == ("ClearBlocksResponse", True)
    wire.send_frame(conn, wire.encode(("Shutdown", None)))
    th.join(10)
    assert not th.is_alive()
    conn.close()
    srv.close()


# ------------------------------------------------------------------------------------------------ the C++ codec (vra_runner)
import os
import subprocess

RUNNER = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vllm_rs_amd", "vra_runner")
needs_runner = pytest.mark.skipif(not os.path.exists(RUNNER), reason="vllm_rs_amd/vra_runner is not built (make -C vllm_rs_amd/csrc)")


def _echo(payload, *mode):
    out = subprocess.run([RUNNER, *mode], input=payload, capture_output=True, timeout=60)
    assert out.returncode == 0, out.stderr.decode()
    return out.stdout


@needs_runner
def test_cpp_codec_reproduces_the_hand_built_frames_and_every_variant():
    """host/wire.h (the codec of the native `vra_runner` binary) decodes and re-encodes each frame to the same bytes: the
    hand-built golden frames above, and one message of every variant the runner exchanges with all optional fields set"""
    sp = dict(temperature=0.5, top_k=7, top_p=0.9, stop_sequences=["a", "bc"], grammar='{"x":1}', grammar_json="{}", reasoning_effort="High",
              max_tokens=99, session_id="s", thinking=True, mcp_mode=False, frequency_penalty=0.25, presence_penalty=-0.5, ignore_eos=True)
    seq = dict(id=3, created_time=11, token_ids=[1, 2, 3], block_table=[9, 8], num_cached_tokens=1, sampling_params=sp, status="Swapped", swapped_time=5,
               stop_sequence="zz", pd_first_token=4, mamba_prefix_hash=77, output_ids=[3], is_tool_call_end=True, hit_stop_sequence=True, block_size=32)
    msgs = [("InitAck", True), ("LoadingProgress", (3, 9)), ("RunPrefill", ([seq, dict(id=4, token_ids=[5], block_table=[1], sampling_params=None)], True)),
            ("RunDecode", ([dict(id=1, last_token=5, len=70, last_block_tokens=6, block_table_last=4, block_tables=[3, 4], sampling_params=sp)], False)),
            ("RunResponse", [1, 2, 3]), ("RunResponse", []), ("FinishDecode", 12), ("Error", "boom é"), ("Heartbeat", None), ("Shutdown", None),
            ("KVCacheSwap", ({1: 2, 3: 4}, True)), ("KVCacheSwapResponse", False), ("ClearBlocks", [4, 5]), ("ClearBlocksResponse", True)]
    for m in msgs:
        b = wire.encode(m)
        assert _echo(b, "--wire-echo") == b, m[0]
    gold = u32(4) + u64(1) + (u64(9) + u32(123) + u64(130) + u64(2) + u32(17) + vec_u32([5, 6, 17]) + golden_sampling_params(0.7, 32, 0.95, 0.5)) + u8(0)
    assert _echo(gold, "--wire-echo") == gold
    # malformed input is refused, not guessed at
    bad = subprocess.run([RUNNER, "--wire-echo"], input=wire.encode(("RunResponse", [1, 2]))[:-1], capture_output=True, timeout=60)
    assert bad.returncode != 0 and b"truncated" in bad.stderr


@needs_runner
def test_cpp_init_json_maps_to_the_same_model_config():
    """MessageType::Init as serde_json writes it -> the vra_model_config / vra_engine_config the Python runner derives"""
    init = dict(rank=1, dev_id=0, num_shards=2, model_type="LLaMa", dtype="F16", is_gguf=False,
                config=dict(architectures=["Qwen2ForCausalLM"], hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                            num_key_value_heads=2, vocab_size=1000, max_position_embeddings=2048, rms_norm_eps=1e-6, rope_theta=1e6,
                            rope_scaling=dict(rope_type="llama3", factor=8.0, low_freq_factor=1.0, high_freq_factor=4.0, original_max_position_embeddings=8192),
                            quantization_config=dict(quant_method="AWQ", bits=4, group_size=64), tie_word_embeddings=None, head_dim=None),
                econfig=dict(block_size=32, max_num_seqs=4, num_blocks=99, seed=7, fp8_kvcache=True, max_model_len=None),
                nccl_id=bytes(range(128)), model_pathes=dict(config_filename="/x/config.json", filenames=["a", "b"]))
    line = _echo(wire.encode_init_json(init), "--init-echo").decode().split()
    kv = dict(zip(line[0::2], line[1::2]))
    cfg = wire.model_cfg_from_init(wire.decode_init_json(wire.encode_init_json(init)))
    assert (kv["rank"], kv["world"], kv["arch"], kv["H"], kv["I"], kv["L"], kv["Hq"], kv["Hkv"], kv["D"], kv["V"]) == ("1", "2", "1", "256", "512", "2", "4", "2", "64", "1000")
    assert cfg["arch"] == "qwen2" and cfg["head_dim"] == 64 and cfg["quant_method"] == "awq" and cfg["group_size"] == 64 and cfg["dtype"] == 1
    assert (kv["quant"], kv["g"], kv["dtype"], kv["bias"], kv["rope"], kv["maxpos"]) == ("2", "64", "1", "1", "2", "2048")
    assert float(kv["theta"]) == 1e6 and abs(float(kv["eps"]) - 1e-6) < 1e-12
    assert (kv["bs"], kv["seqs"], kv["blocks"], kv["seed"], kv["fp8"], kv["nccl"], kv["files"]) == ("32", "4", "99", "7", "1", "128", "2")
    for broken in (b'{"Init": {"config": {}}}', b'{"RunPrefill": 1}', b'{"Init": '):
        assert subprocess.run([RUNNER, "--init-echo"], input=broken, capture_output=True, timeout=60).returncode != 0


@needs_runner
def test_cpp_checkpoint_dtype_conversions_match_numpy():
    """the safetensors loader of vra_runner converts f16 <-> bf16 <-> f32 in software (scales and biases are f16 on disk even
    for bf16 models, wna16.rs:97-109): exhaustive over all 65536 half patterns, and round-to-nearest-even from f32"""
    from vllm_rs_amd.checkpoint import f32_to_bf16_bits
    allh = np.arange(65536, dtype=np.uint16)
    wide = np.frombuffer(_echo(allh.tobytes(), "--cvt", "f16-f32"), np.float32)
    ref = allh.view(np.float16).astype(np.float32)
    assert (wide.view(np.uint32) == ref.view(np.uint32))[~np.isnan(ref)].all() and np.isnan(wide[np.isnan(ref)]).all()
    widb = np.frombuffer(_echo(allh.tobytes(), "--cvt", "bf16-f32"), np.float32)
    assert (widb.view(np.uint32) == (allh.astype(np.uint32) << 16)).all()
    r = np.random.default_rng(0)
    x = np.concatenate([r.standard_normal(200000).astype(np.float32) * np.float32(10.0) ** r.integers(-9, 6, 200000).astype(np.float32),
                        ref[~np.isnan(ref)], np.array([65504.0, 65519.9, 65520.0, 1e9, -1e9, 5.96e-8, 2.98e-8, 2.9e-8, 0.0, -0.0, np.inf, -np.inf], np.float32),
                        (ref[~np.isnan(ref)].astype(np.float64) * (1 + 2.0 ** -12)).astype(np.float32)])
    with np.errstate(over="ignore"):
        want16 = x.astype(np.float16).view(np.uint16)
    got16 = np.frombuffer(_echo(x.tobytes(), "--cvt", "f32-f16"), np.uint16)
    assert (got16 == want16).all(), np.flatnonzero(got16 != want16)[:5]
    gotb = np.frombuffer(_echo(x.tobytes(), "--cvt", "f32-bf16"), np.uint16)
    assert (gotb == f32_to_bf16_bits(x)).all()
