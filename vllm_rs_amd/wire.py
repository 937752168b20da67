"""The reference's engine <-> runner wire format, byte for byte, so that this runtime can be spawned as the
`runner` process of an unmodified vllm.rs engine.

  framing   src/runner/mod.rs:246-295 — u32 little-endian payload length, payload, then the RECEIVER writes one ack byte 0x01
            (send_local waits for it, receive_local sends it).
  payload   `MessageType` (src/runner/mod.rs:169-244) serialised with serde:
              * JSON (serde_json, externally tagged enums) for `Init` — the runner reads it with use_json = true
                (src/runner/runner.rs:73) and the engine sends it so (send_and_expect_ack, mod.rs:297-312);
              * bincode 1.x default options for everything after it (runner.rs:250, use_json = false): little endian,
                fixed-width integers (usize / isize / u64 = 8 bytes, u32 / enum variant index = 4 bytes), bool = 1 byte,
                Option = 1 tag byte + value, Vec / String = u64 length + elements, f32 = 4 bytes, tuples / structs = fields
                in declaration order, HashMap = u64 length + (key, value) pairs.
  structs   Sequence / DecodeSequence (src/core/sequence.rs:32-60), SamplingParams without the `python` feature
            (src/utils/config.rs:505-537; `stop_token_ids` is #[serde(skip)], `grammar` travels as Option<String> of JSON).

Only the variants the forward-pass runner exchanges are encoded/decoded; others raise WireError with the variant name.
"""
import base64
import json
import struct

VARIANTS = ["Init", "InitAck", "LoadingProgress", "RunPrefill", "RunDecode", "RunResponse", "RunEmbed", "RunResponseEmbed", "FinishDecode",
            "CaptureMambaPrefixState", "CaptureMambaPrefixStateResponse", "HasMambaPrefixState", "HasMambaPrefixStateResponse", "Error",
            "Heartbeat", "TransferPrefill", "TransferPrefillResponse", "ReceivePrefill", "ReceivePrefillResponse", "CheckPrefillStatus",
            "CheckPrefillStatusResponse", "KVCacheSwap", "KVCacheSwapResponse", "KvCacheSend", "KvCacheSendResponse", "KvCacheReceive",
            "KvCacheReceiveResponse", "KvCacheRelease", "KvCacheReleaseResponse", "CheckKvCacheRelease", "CheckKvCacheReleaseResponse",
            "ClearBlocks", "ClearBlocksResponse", "UsableMemoryLeft", "Shutdown"]  # declaration order = bincode variant index
VIDX = {n: i for i, n in enumerate(VARIANTS)}
SEQ_STATUS = ["Waiting", "Running", "Finished", "Cached", "Swapped", "FinishSwapped"]  # sequence.rs:7-15
REASONING_EFFORT = ["Low", "Medium", "High"]


class WireError(ValueError):
    pass


# ---------------------------------------------------------------- bincode primitives
class W:
    def __init__(self):
        self.b = bytearray()

    def u8(self, v): self.b += struct.pack("<B", v)
    def bool(self, v): self.u8(1 if v else 0)
    def u32(self, v): self.b += struct.pack("<I", v)
    def u64(self, v): self.b += struct.pack("<Q", v)
    def i64(self, v): self.b += struct.pack("<q", v)
    def f32(self, v): self.b += struct.pack("<f", v)

    def string(self, s):
        e = s.encode()
        self.u64(len(e))
        self.b += e

    def opt(self, v, put):
        if v is None:
            self.u8(0)
        else:
            self.u8(1)
            put(v)

    def vec(self, xs, put):
        self.u64(len(xs))
        for x in xs:
            put(x)

    def vec_u32(self, xs):
        self.u64(len(xs))
        self.b += struct.pack(f"<{len(xs)}I", *xs)


class R:
    def __init__(self, b):
        self.b, self.i = memoryview(bytes(b)), 0

    def _take(self, fmt, n):
        if self.i + n > len(self.b):
            raise WireError("truncated bincode payload")
        v = struct.unpack_from(fmt, self.b, self.i)[0]
        self.i += n
        return v

    def u8(self): return self._take("<B", 1)

    def bool(self):
        v = self.u8()
        if v > 1:
            raise WireError(f"invalid bool byte {v}")
        return bool(v)

    def u32(self): return self._take("<I", 4)
    def u64(self): return self._take("<Q", 8)
    def i64(self): return self._take("<q", 8)
    def f32(self): return self._take("<f", 4)

    def string(self):
        n = self.u64()
        if self.i + n > len(self.b):
            raise WireError("truncated string")
        s = bytes(self.b[self.i:self.i + n]).decode()
        self.i += n
        return s

    def opt(self, get):
        t = self.u8()
        if t > 1:
            raise WireError(f"invalid Option tag {t}")
        return get() if t else None

    def vec(self, get): return [get() for _ in range(self.u64())]

    def vec_u32(self):
        n = self.u64()
        if self.i + 4 * n > len(self.b):
            raise WireError("truncated Vec<u32>")
        v = list(struct.unpack_from(f"<{n}I", self.b, self.i))
        self.i += 4 * n
        return v

    def done(self):
        if self.i != len(self.b):
            raise WireError(f"{len(self.b) - self.i} trailing bytes")


# ---------------------------------------------------------------- structs
SP_DEFAULT = dict(temperature=None, max_tokens=None, ignore_eos=False, top_k=None, top_p=None, session_id=None, frequency_penalty=None,
                  presence_penalty=None, stop_sequences=None, thinking=None, mcp_mode=None, grammar=None, grammar_json=None, reasoning_effort=None)


def put_sampling_params(w, sp):
    """SamplingParams, config.rs:505-537 (non-python layout), field order as declared"""
    sp = dict(SP_DEFAULT, **(sp or {}))
    w.opt(sp["temperature"], w.f32)
    w.opt(sp["max_tokens"], w.u64)
    w.bool(sp["ignore_eos"])
    w.opt(sp["top_k"], w.i64)
    w.opt(sp["top_p"], w.f32)
    w.opt(sp["session_id"], w.string)
    w.opt(sp["frequency_penalty"], w.f32)
    w.opt(sp["presence_penalty"], w.f32)
    w.opt(sp["stop_sequences"], lambda xs: w.vec(xs, w.string))
    w.opt(sp["thinking"], w.bool)
    w.opt(sp["mcp_mode"], w.bool)
    w.opt(sp["grammar"], w.string)        # serialize_optional_grammar: Option<String> holding the grammar's JSON (config.rs:128-145)
    w.opt(sp["grammar_json"], w.string)
    w.opt(sp["reasoning_effort"], lambda e: w.u32(REASONING_EFFORT.index(e)))


def get_sampling_params(r):
    return dict(temperature=r.opt(r.f32), max_tokens=r.opt(r.u64), ignore_eos=r.bool(), top_k=r.opt(r.i64), top_p=r.opt(r.f32),
                session_id=r.opt(r.string), frequency_penalty=r.opt(r.f32), presence_penalty=r.opt(r.f32),
                stop_sequences=r.opt(lambda: r.vec(r.string)), thinking=r.opt(r.bool), mcp_mode=r.opt(r.bool), grammar=r.opt(r.string),
                grammar_json=r.opt(r.string), reasoning_effort=r.opt(lambda: REASONING_EFFORT[r.u32()]))


def put_sequence(w, s):
    """Sequence, sequence.rs:32-51"""
    w.u64(s["id"])
    w.u64(s.get("created_time", 0))
    w.opt(s.get("swapped_time"), w.u64)
    w.u32(SEQ_STATUS.index(s.get("status", "Waiting")))
    w.vec_u32(s["token_ids"])
    w.vec_u32(s.get("output_ids", []))
    w.vec_u32(s["block_table"])
    w.u64(s.get("num_cached_tokens", 0))
    w.opt(s.get("mamba_prefix_hash"), w.u64)
    w.u32(s.get("last_token", s["token_ids"][-1] if s["token_ids"] else 0))
    w.u64(s.get("block_size", 64))
    put_sampling_params(w, s.get("sampling_params"))
    w.opt(s.get("pd_first_token"), w.u32)
    if s.get("images") is not None:
        raise WireError("Sequence.images (multimodal) is outside this path")
    w.u8(0)
    w.bool(s.get("is_tool_call_end", False))
    w.bool(s.get("hit_stop_sequence", False))
    w.opt(s.get("stop_sequence"), w.string)


def get_sequence(r):
    s = dict(id=r.u64(), created_time=r.u64(), swapped_time=r.opt(r.u64), status=SEQ_STATUS[r.u32()], token_ids=r.vec_u32(), output_ids=r.vec_u32(),
             block_table=r.vec_u32(), num_cached_tokens=r.u64(), mamba_prefix_hash=r.opt(r.u64), last_token=r.u32(), block_size=r.u64(),
             sampling_params=get_sampling_params(r), pd_first_token=r.opt(r.u32))
    if r.u8() != 0:
        raise WireError("Sequence.images (multimodal) is outside this path")
    s.update(images=None, is_tool_call_end=r.bool(), hit_stop_sequence=r.bool(), stop_sequence=r.opt(r.string))
    return s


def put_decode_sequence(w, s):
    """DecodeSequence, sequence.rs:53-62"""
    w.u64(s["id"])
    w.u32(s["last_token"])
    w.u64(s["len"])
    w.u64(s["last_block_tokens"])
    w.u32(s["block_table_last"])
    w.vec_u32(s["block_tables"])
    put_sampling_params(w, s.get("sampling_params"))


def get_decode_sequence(r):
    return dict(id=r.u64(), last_token=r.u32(), len=r.u64(), last_block_tokens=r.u64(), block_table_last=r.u32(), block_tables=r.vec_u32(),
                sampling_params=get_sampling_params(r))


# ---------------------------------------------------------------- MessageType
def encode(msg):
    """msg: (variant name, payload) -> bincode bytes.  Payload shapes: InitAck bool; LoadingProgress (usize, usize); RunPrefill
    ([Sequence], bool); RunDecode ([DecodeSequence], bool); RunResponse [u32]; FinishDecode usize; Error str; Heartbeat/Shutdown None;
    KVCacheSwap ({src: dst}, bool); *Response bool; ClearBlocks [u32]"""
    name, p = msg
    w = W()
    if name not in VIDX:
        raise WireError(f"unknown MessageType variant {name}")
    w.u32(VIDX[name])
    if name in ("InitAck", "KVCacheSwapResponse", "ClearBlocksResponse"):
        w.bool(p)
    elif name == "LoadingProgress":
        w.u64(p[0]), w.u64(p[1])
    elif name == "RunPrefill":
        w.vec(p[0], lambda s: put_sequence(w, s)), w.bool(p[1])
    elif name == "RunDecode":
        w.vec(p[0], lambda s: put_decode_sequence(w, s)), w.bool(p[1])
    elif name in ("RunResponse", "ClearBlocks"):
        w.vec_u32(p)
    elif name == "FinishDecode":
        w.u64(p)
    elif name == "Error":
        w.string(p)
    elif name in ("Heartbeat", "Shutdown"):
        pass
    elif name == "KVCacheSwap":
        w.u64(len(p[0]))
        for k, v in p[0].items():
            w.u64(k), w.u64(v)
        w.bool(p[1])
    else:
        raise WireError(f"MessageType::{name} is not part of the forward-pass runner protocol")
    return bytes(w.b)


def decode(b):
    r = R(b)
    i = r.u32()
    if i >= len(VARIANTS):
        raise WireError(f"variant index {i} out of range")
    name = VARIANTS[i]
    if name in ("InitAck", "KVCacheSwapResponse", "ClearBlocksResponse"):
        p = r.bool()
    elif name == "LoadingProgress":
        p = (r.u64(), r.u64())
    elif name == "RunPrefill":
        p = (r.vec(lambda: get_sequence(r)), r.bool())
    elif name == "RunDecode":
        p = (r.vec(lambda: get_decode_sequence(r)), r.bool())
    elif name in ("RunResponse", "ClearBlocks"):
        p = r.vec_u32()
    elif name == "FinishDecode":
        p = r.u64()
    elif name == "Error":
        p = r.string()
    elif name in ("Heartbeat", "Shutdown"):
        p = None
    elif name == "KVCacheSwap":
        n = r.u64()
        m = {}
        for _ in range(n):
            k = r.u64()
            m[k] = r.u64()
        p = (m, r.bool())
    else:
        raise WireError(f"MessageType::{name} is not part of the forward-pass runner protocol")
    r.done()
    return name, p


# ---------------------------------------------------------------- Init (JSON)
def encode_init_json(req):
    """MessageType::Init(RunnerInitRequest) as serde_json writes it: {"Init": {...}}; nccl_id is base64 without padding of the
    128 id bytes (mod.rs:31-57)"""
    body = dict(req)
    if isinstance(body.get("nccl_id"), (bytes, bytearray)):
        body["nccl_id"] = base64.b64encode(bytes(body["nccl_id"])).decode().rstrip("=")
    return json.dumps({"Init": body}).encode()


def encode_usable_memory_left_json(econfig):
    """MessageType::UsableMemoryLeft(EngineConfig) as the engine sends it after the first InitAck (send_and_expect_ack -> JSON,
    src/runner/mod.rs:300-312, src/core/engine.rs:372-377)"""
    return json.dumps({"UsableMemoryLeft": econfig}).encode()


def decode_usable_memory_left_json(b):
    """-> the EngineConfig dict, or None when the frame is something else (the reference then keeps Init.econfig)"""
    try:
        d = json.loads(bytes(b).decode())
    except (UnicodeDecodeError, json.JSONDecodeError):
        return None
    if isinstance(d, dict) and isinstance(d.get("UsableMemoryLeft"), dict):
        return d["UsableMemoryLeft"]
    return None


def decode_init_json(b):
    """-> dict(rank, dev_id, num_shards, model_type, config (HF-style keys as the reference's Config serialises them), econfig,
    model_pathes, is_gguf, dtype, is_rope_i, nccl_id bytes or None)"""
    d = json.loads(bytes(b).decode())
    if not isinstance(d, dict) or "Init" not in d:
        raise WireError("expected MessageType::Init as JSON")
    req = dict(d["Init"])
    nid = req.get("nccl_id")
    if isinstance(nid, str):
        raw = base64.b64decode(nid + "=" * (-len(nid) % 4))
        if len(raw) != 128:
            raise WireError(f"Expected 128 bytes but got {len(raw)}")
        req["nccl_id"] = raw
    return req


def model_cfg_from_init(req):
    """the reference's `Config` (config.rs:218-255) + dtype -> the cfg dict of vllm_rs_amd.engine.model_config"""
    c = req["config"]
    qc = c.get("quantization_config") or {}
    arch = (c.get("architectures") or ["LlamaForCausalLM"])[0]
    rs = c.get("rope_scaling") or None
    sw = c.get("sliding_window")
    sw = int(sw) if sw and c.get("use_sliding_window", True) and int(sw) < int(c["max_position_embeddings"]) else 0
    return dict(sliding_window=sw, arch="qwen2" if arch.startswith("Qwen2") else ("qwen3" if arch.startswith("Qwen3") else "llama"), hidden_size=c["hidden_size"], intermediate_size=c["intermediate_size"],
                num_layers=c["num_hidden_layers"], num_heads=c["num_attention_heads"], num_kv_heads=c["num_key_value_heads"],
                head_dim=c.get("head_dim") or c["hidden_size"] // c["num_attention_heads"], vocab_size=c["vocab_size"],
                max_position_embeddings=c["max_position_embeddings"], rms_norm_eps=c["rms_norm_eps"], rope_theta=c.get("rope_theta") or 10000.0,
                rope_scaling=rs, attention_bias=bool(c.get("attention_bias") or c.get("qkv_bias") or (arch.startswith("Qwen2"))),
                quant_method=(qc.get("quant_method") or None) and str(qc.get("quant_method")).lower(), group_size=qc.get("group_size", 128), dtype={"BF16": 0, "F16": 1}[req.get("dtype", "BF16")],
                tie_word_embeddings=bool(c.get("tie_word_embeddings")))


# ---------------------------------------------------------------- framing
def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(n - len(buf))
        if not chunk:
            raise ConnectionError("peer closed the stream")
        buf += chunk
    return bytes(buf)


def send_frame(sock, payload):
    """send_local (mod.rs:246-275): length, payload, then wait for the 1-byte acknowledgment"""
    sock.sendall(struct.pack("<I", len(payload)) + payload)
    ack = _recv_exact(sock, 1)
    if ack != b"\x01":
        raise WireError(f"unexpected acknowledgment byte {ack!r}")


def recv_frame(sock):
    """receive_local (mod.rs:277-295): length, payload, then acknowledge with 0x01"""
    n = struct.unpack("<I", _recv_exact(sock, 4))[0]
    payload = _recv_exact(sock, n)
    sock.sendall(b"\x01")
    return payload
