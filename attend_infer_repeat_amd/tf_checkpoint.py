"""Reader / writer of TensorFlow-1 `model.ckpt-N` checkpoints (the TensorBundle "V2" format) without TensorFlow, and the
importer that turns one into the fused engine's parameters ("optional importer for TF1 model.ckpt").

The reference only ever SAVES (scripts/multi_mnist.py:116 `tf.train.Saver()`, :145-146 `saver.save(sess, '.../model.ckpt',
global_step=train_itr)`); what it leaves on disk is `model.ckpt-<step>.index` + `model.ckpt-<step>.data-00000-of-00001`.

UNPINNED: there is no TensorFlow in this image, so nothing here has seen a file TensorFlow wrote.  The format below is restated
from its published description (tensorflow/core/util/tensor_bundle, tensorflow/core/lib/io/{table,block,format}: a LevelDB-style
sorted table whose values are BundleHeaderProto / BundleEntryProto messages).  What the tests pin: the CRC-32C known answers of
RFC 3720 B.4, the LevelDB CRC mask, reader(writer(x)) == x over every dtype handled, prefix-compressed keys written by an
independent block builder, and rejection of corrupted blocks.  The VARIABLE NAMES of the reference's graph (Sonnet 1.1 module scopes)
cannot be checked here either: `default_name_map` matches by scope grouping + shapes and lists what it found when in doubt;
pass `name_map=` to override.

Layout of `<prefix>.index` (all integers little-endian):
  data block(s) | metaindex block | index block | footer (48 bytes: two BlockHandles as varint64 pairs, zero padded to 40, then
  the magic 0xdb4775248b80fb57).  Every block is followed by a 5-byte trailer: compression type (0 = none, 1 = snappy) and the
  masked CRC-32C of block + type.  A block is a run of entries (varint32 shared, varint32 non_shared, varint32 value_len, key
  suffix, value) followed by the uint32 restart offsets and their count.  Index-block values are BlockHandles of data blocks.
  Key "" holds the BundleHeaderProto (fields: num_shards = 1, endianness = 2, version = 3); every other key is a tensor name whose
  BundleEntryProto gives (field numbers) dtype = 1, shape = 2, shard_id = 3, offset = 4, size = 5, crc32c = 6 (fixed32, masked) in the data shard.
"""
import os
import re
import struct
from typing import Callable, Dict, Iterable, List, Optional, Tuple

import numpy as np

_MAGIC = 0xdb4775248b80fb57
_FOOTER = 48
_MASK_DELTA = 0xa282ead8

# tensorflow/core/framework/types.proto
_DTYPES = {1: np.dtype("<f4"), 2: np.dtype("<f8"), 3: np.dtype("<i4"), 4: np.dtype("u1"), 5: np.dtype("<i2"), 6: np.dtype("i1"),
           9: np.dtype("<i8"), 10: np.dtype("?"), 17: np.dtype("<u2"), 19: np.dtype("<f2"), 22: np.dtype("<u4"), 23: np.dtype("<u8")}
_DTYPE_IDS = {v: k for k, v in _DTYPES.items()}


class TFCheckpointError(ValueError):
    pass


# ---- CRC-32C (Castagnoli), as LevelDB masks it ---------------------------------------------------------------------------------
def _make_tables():
    t0 = np.zeros(256, dtype=np.uint32)
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ (0x82f63b78 if c & 1 else 0)
        t0[i] = c
    tabs = [t0]
    for _ in range(7):                                   # slicing-by-8: table k advances a byte through k more zero bytes
        prev = tabs[-1]
        tabs.append((prev >> np.uint32(8)) ^ t0[prev & np.uint32(0xff)])
    return [[int(x) for x in t] for t in tabs]


_T = _make_tables()


def crc32c(data: bytes, crc: int = 0) -> int:
    """CRC-32C of `data` (RFC 3720 B.4: crc32c(b'123456789') == 0xE3069283), slicing-by-8 over plain ints."""
    c = crc ^ 0xffffffff
    mv = memoryview(data).cast("B")
    n8 = len(mv) // 8 * 8
    t0, t1, t2, t3, t4, t5, t6, t7 = _T
    if n8:
        words = struct.unpack("<%dQ" % (n8 // 8), mv[:n8])
        for w in words:
            w ^= c
            c = (t7[w & 0xff] ^ t6[(w >> 8) & 0xff] ^ t5[(w >> 16) & 0xff] ^ t4[(w >> 24) & 0xff] ^
                 t3[(w >> 32) & 0xff] ^ t2[(w >> 40) & 0xff] ^ t1[(w >> 48) & 0xff] ^ t0[w >> 56])
    for b in mv[n8:]:
        c = t0[(c ^ b) & 0xff] ^ (c >> 8)
    return c ^ 0xffffffff


def mask_crc(crc: int) -> int:
    return ((((crc >> 15) | (crc << 17)) & 0xffffffff) + _MASK_DELTA) & 0xffffffff


def unmask_crc(masked: int) -> int:
    rot = (masked - _MASK_DELTA) & 0xffffffff
    return ((rot >> 17) | (rot << 15)) & 0xffffffff


# ---- varints and the three protobuf messages the bundle uses -------------------------------------------------------------------
def _get_varint(buf, pos: int) -> Tuple[int, int]:
    out, shift = 0, 0
    while True:
        if pos >= len(buf):
            raise TFCheckpointError("truncated varint")
        b = buf[pos]; pos += 1
        out |= (b & 0x7f) << shift
        if b < 0x80:
            return out, pos
        shift += 7
        if shift > 63:
            raise TFCheckpointError("varint longer than 64 bits")


def _put_varint(v: int) -> bytes:
    if v < 0:
        v += 1 << 64
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7f) | 0x80); v >>= 7
    out.append(v)
    return bytes(out)


def _fields(buf) -> Iterable[Tuple[int, int, object]]:
    """(field number, wire type, value) of one protobuf message; length-delimited values as memoryview."""
    pos, mv = 0, memoryview(buf)
    while pos < len(mv):
        key, pos = _get_varint(mv, pos)
        f, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _get_varint(mv, pos)
        elif wt == 1:
            v = struct.unpack_from("<Q", mv, pos)[0]; pos += 8
        elif wt == 5:
            v = struct.unpack_from("<I", mv, pos)[0]; pos += 4
        elif wt == 2:
            n, pos = _get_varint(mv, pos)
            if pos + n > len(mv):
                raise TFCheckpointError("truncated length-delimited field")
            v = mv[pos:pos + n]; pos += n
        else:
            raise TFCheckpointError(f"unsupported protobuf wire type {wt}")
        yield f, wt, v


def _signed(v: int) -> int:
    return v - (1 << 64) if v >= (1 << 63) else v


def _parse_shape(buf) -> Tuple[int, ...]:
    dims = []
    for f, _, v in _fields(buf):
        if f == 2:                                       # TensorShapeProto.dim
            size = 0
            for g, _, w in _fields(v):
                if g == 1:
                    size = _signed(w)
            dims.append(size)
        elif f == 3 and v:                               # unknown_rank
            raise TFCheckpointError("tensor of unknown rank in a checkpoint")
    return tuple(dims)


def _parse_entry(buf) -> dict:
    e = dict(dtype=0, shape=(), shard_id=0, offset=0, size=0, crc32c=None, sliced=False)
    for f, _, v in _fields(buf):
        if f == 1: e["dtype"] = v
        elif f == 2: e["shape"] = _parse_shape(v)
        elif f == 3: e["shard_id"] = v
        elif f == 4: e["offset"] = v
        elif f == 5: e["size"] = v
        elif f == 6: e["crc32c"] = v
        elif f == 7: e["sliced"] = True
    return e


def _parse_header(buf) -> dict:
    h = dict(num_shards=0, endianness=0, producer=0)
    for f, _, v in _fields(buf):
        if f == 1: h["num_shards"] = v
        elif f == 2: h["endianness"] = v                 # 0 = little
        elif f == 3:
            for g, _, w in _fields(v):
                if g == 1: h["producer"] = w
    return h


def _ld(field: int, payload: bytes) -> bytes:
    return _put_varint((field << 3) | 2) + _put_varint(len(payload)) + payload


def _vi(field: int, v: int) -> bytes:
    return _put_varint(field << 3) + _put_varint(v)


def _entry_bytes(dtype_id: int, shape, offset: int, size: int, crc_masked: int) -> bytes:
    shp = b"".join(_ld(2, _vi(1, int(d))) for d in shape)
    out = _vi(1, dtype_id) + _ld(2, shp)
    if offset:
        out += _vi(4, offset)                            # proto3: zero-valued scalars are not serialised
    out += _vi(5, size) + _put_varint((6 << 3) | 5) + struct.pack("<I", crc_masked)
    return out


# ---- table blocks ----------------------------------------------------------------------------------------------------------------
def _read_block(buf, offset: int, size: int, verify: bool) -> memoryview:
    if offset + size + 5 > len(buf):
        raise TFCheckpointError("block handle points past the end of the index file")
    body = memoryview(buf)[offset:offset + size]
    ctype = buf[offset + size]
    stored = struct.unpack_from("<I", buf, offset + size + 1)[0]
    if verify and unmask_crc(stored) != crc32c(bytes(memoryview(buf)[offset:offset + size + 1])):
        raise TFCheckpointError(f"index block at {offset}: checksum mismatch")
    if ctype != 0:
        raise TFCheckpointError("compressed index block (type %d): the bundle writer never compresses its index; "
                                "re-save with a stock tf.train.Saver" % ctype)
    return body


def _block_entries(block) -> List[Tuple[bytes, memoryview]]:
    if len(block) < 4:
        raise TFCheckpointError("block shorter than its restart count")
    n_restarts = struct.unpack_from("<I", block, len(block) - 4)[0]
    limit = len(block) - 4 - 4 * n_restarts
    if limit < 0:
        raise TFCheckpointError("restart array larger than its block")
    out, pos, key = [], 0, b""
    while pos < limit:
        shared, pos = _get_varint(block, pos)
        non_shared, pos = _get_varint(block, pos)
        vlen, pos = _get_varint(block, pos)
        if shared > len(key) or pos + non_shared + vlen > limit:
            raise TFCheckpointError("corrupt block entry")
        key = key[:shared] + bytes(block[pos:pos + non_shared]); pos += non_shared
        out.append((key, block[pos:pos + vlen])); pos += vlen
    return out


def _handle(buf, pos: int) -> Tuple[int, int, int]:
    off, pos = _get_varint(buf, pos)
    size, pos = _get_varint(buf, pos)
    return off, size, pos


def read_index(prefix: str, verify: bool = True) -> Tuple[dict, Dict[str, dict]]:
    """Header and name -> entry (dtype id, shape, shard_id, offset, size, masked crc32c) of `<prefix>.index`."""
    path = prefix + ".index"
    with open(path, "rb") as f:
        buf = f.read()
    if len(buf) < _FOOTER or struct.unpack_from("<Q", buf, len(buf) - 8)[0] != _MAGIC:
        raise TFCheckpointError(f"{path}: not a TensorBundle index (bad magic); a TF-0.x 'V1' checkpoint is a single file and is not read here")
    foot = memoryview(buf)[len(buf) - _FOOTER:]
    _, _, p = _handle(foot, 0)                           # metaindex: unused by the bundle
    ioff, isize, _ = _handle(foot, p)
    header, entries = None, {}
    for _, hv in _block_entries(_read_block(buf, ioff, isize, verify)):
        doff, dsize, _ = _handle(hv, 0)
        for key, val in _block_entries(_read_block(buf, doff, dsize, verify)):
            if key == b"":
                header = _parse_header(val)
            else:
                entries[key.decode("utf-8")] = _parse_entry(val)
    if header is None:
        raise TFCheckpointError(f"{path}: no bundle header under the empty key")
    if header["endianness"] != 0:
        raise TFCheckpointError("big-endian bundle")
    return header, entries


def read_bundle(prefix: str, names: Optional[Iterable[str]] = None, verify: str = "all") -> Dict[str, np.ndarray]:
    """Tensors of a TF-1 checkpoint `<prefix>.index` + `<prefix>.data-*-of-*` as numpy arrays.
    verify: "all" (index blocks and tensor bytes), "index", or "none"."""
    if verify not in ("all", "index", "none"):
        raise ValueError("verify must be 'all', 'index' or 'none'")
    header, entries = read_index(prefix, verify != "none")
    want = list(entries) if names is None else list(names)
    shards: Dict[int, np.memmap] = {}
    out = {}
    for name in want:
        if name not in entries:
            raise KeyError(f"{name!r} is not in {prefix}.index ({len(entries)} tensors)")
        e = entries[name]
        if e["sliced"]:
            raise TFCheckpointError(f"{name}: partitioned variable (slices); not produced by the reference's Saver")
        if e["dtype"] not in _DTYPES:
            raise TFCheckpointError(f"{name}: dtype id {e['dtype']} is not a fixed-width numeric type")
        dt = _DTYPES[e["dtype"]]
        count = int(np.prod(e["shape"], dtype=np.int64)) if e["shape"] else 1
        if count * dt.itemsize != e["size"]:
            raise TFCheckpointError(f"{name}: {e['size']} bytes for shape {e['shape']} of {dt}")
        sid = e["shard_id"]
        if sid not in shards:
            p = "%s.data-%05d-of-%05d" % (prefix, sid, header["num_shards"])
            if not os.path.exists(p):
                raise TFCheckpointError(f"missing data shard {p}")
            shards[sid] = np.memmap(p, dtype=np.uint8, mode="r") if os.path.getsize(p) else np.zeros(0, np.uint8)
        raw = shards[sid][e["offset"]:e["offset"] + e["size"]]
        if len(raw) != e["size"]:
            raise TFCheckpointError(f"{name}: data shard ends inside the tensor")
        data = raw.tobytes()
        if verify == "all" and e["crc32c"] is not None and unmask_crc(e["crc32c"]) != crc32c(data):
            raise TFCheckpointError(f"{name}: checksum mismatch in the data shard")
        out[name] = np.frombuffer(data, dtype=dt).reshape(e["shape"]).copy()
    return out


# ---- writer (export to the reference; round trips in the tests) -------------------------------------------------------------------
def _build_block(items: List[Tuple[bytes, bytes]], restart_interval: int = 16) -> bytes:
    out, restarts, prev = bytearray(), [], b""
    for i, (k, v) in enumerate(items):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(out))
        else:
            while shared < min(len(prev), len(k)) and prev[shared] == k[shared]:
                shared += 1
        out += _put_varint(shared) + _put_varint(len(k) - shared) + _put_varint(len(v)) + k[shared:] + v
        prev = k
    if not restarts:
        restarts = [0]
    out += b"".join(struct.pack("<I", r) for r in restarts) + struct.pack("<I", len(restarts))
    return bytes(out)


def _with_trailer(block: bytes) -> bytes:
    return block + b"\x00" + struct.pack("<I", mask_crc(crc32c(block + b"\x00")))


def write_bundle(prefix: str, tensors: Dict[str, np.ndarray], block_size: int = 4096) -> None:
    """`<prefix>.index` + `<prefix>.data-00000-of-00001` holding `tensors` (names sorted bytewise, as the table requires): what
    `tf.train.Saver().restore(sess, prefix)` of a TF-1 graph with those variable names reads."""
    names = sorted(tensors, key=lambda s: s.encode("utf-8"))
    if any(n == "" for n in names):
        raise ValueError("the empty name is the bundle header's key")
    items: List[Tuple[bytes, bytes]] = [(b"", _vi(1, 1) + _ld(3, _vi(1, 1)))]      # num_shards 1, little endian, producer 1
    offset = 0
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        for n in names:
            a = np.asarray(tensors[n], order="C")                # (ascontiguousarray would turn a scalar into shape (1,))
            dt = a.dtype.newbyteorder("<") if a.dtype.byteorder == ">" else a.dtype
            if np.dtype(dt) not in _DTYPE_IDS:
                raise TFCheckpointError(f"{n}: dtype {a.dtype} has no fixed-width TensorFlow counterpart here")
            raw = a.astype(dt, copy=False).tobytes()
            f.write(raw)
            items.append((n.encode("utf-8"), _entry_bytes(_DTYPE_IDS[np.dtype(dt)], a.shape, offset, len(raw), mask_crc(crc32c(raw)))))
            offset += len(raw)
    out, index_items, cur, cur_bytes = bytearray(), [], [], 0
    def flush():
        nonlocal cur, cur_bytes
        if not cur:
            return
        blk = _build_block(cur)
        index_items.append((cur[-1][0], _put_varint(len(out)) + _put_varint(len(blk))))
        out.extend(_with_trailer(blk)); cur, cur_bytes = [], 0
    for kv in items:
        cur.append(kv); cur_bytes += len(kv[0]) + len(kv[1]) + 3
        if cur_bytes >= block_size:
            flush()
    flush()
    meta = _build_block([]); meta_h = _put_varint(len(out)) + _put_varint(len(meta)); out.extend(_with_trailer(meta))
    idx = _build_block(index_items, restart_interval=1); idx_h = _put_varint(len(out)) + _put_varint(len(idx)); out.extend(_with_trailer(idx))
    foot = meta_h + idx_h
    out.extend(foot + b"\x00" * (40 - len(foot)) + struct.pack("<Q", _MAGIC))
    with open(prefix + ".index", "wb") as f:
        f.write(bytes(out))


# ---- names: the reference's graph -> the engine's flat layout ------------------------------------------------------------------
_SLOT = re.compile(r"/(RMSProp(_\d+)?|Adam(_\d+)?|Momentum|ExponentialMovingAverage)$")
# (engine group, how many Affine layers) in the order cell.py:116-171 / model.py:218-231 first connect them
_HINTS = {"input_encoder": ("encoder",), "glimpse_encoder": ("encoder",), "glimpse_decoder": ("decoder",),
          "transform": ("transform",), "steps": ("steps",), "baseline": ("baseline",), "what": ("gaussian",)}


def _natural(s: str):
    return [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", s)]


def default_name_map(shapes: "Dict[str, Tuple[int, ...]]", tf_shapes: "Dict[str, Tuple[int, ...]]") -> Dict[str, str]:
    """engine name -> TF variable name.  `shapes` is engine_config.param_shapes(cfg); `tf_shapes` name -> shape of the checkpoint's
    tensors.  Optimiser slots and non-trainable scalars are ignored.  Matching: the `w` / `b` pairs of one TF scope form chains
    (a layer's output width is the next one's input width); an engine MLP takes the unused chain of its own shapes whose scope
    carries its hint (case-insensitive), else the only unused chain of those shapes; anything else raises with the candidates."""
    model_vars = {n: tuple(s) for n, s in tf_shapes.items() if not _SLOT.search(n) and len(s) >= 1}
    groups: Dict[str, List[str]] = {}
    for n in shapes:
        groups.setdefault(n.split("/")[0], []).append(n)
    # TF side: layers = scopes holding exactly a 2-D `w` and a 1-D `b`
    layers = {}
    for n, s in model_vars.items():
        scope, _, leaf = n.rpartition("/")
        if leaf in ("w", "b"):
            layers.setdefault(scope, {})[leaf] = n
    layers = {sc: d for sc, d in layers.items() if set(d) == {"w", "b"} and len(model_vars[d["w"]]) == 2}
    used, out = set(), {}

    def chains_for(want: List[Tuple[int, int]]) -> List[List[str]]:
        by_parent: Dict[str, List[str]] = {}
        for sc in layers:
            if sc not in used:
                by_parent.setdefault(sc.rpartition("/")[0], []).append(sc)
        found = []
        for parent, scs in by_parent.items():
            scs = sorted(scs, key=_natural)
            for i in range(len(scs) - len(want) + 1):
                cand = scs[i:i + len(want)]
                if all(model_vars[layers[c]["w"]] == w for c, w in zip(cand, want)):
                    found.append(cand); break
        return found

    for g, names in groups.items():
        if g == "lstm":
            continue
        ws = [n for n in names if n.endswith("/w")]
        want = [tuple(shapes[n]) for n in ws]
        cands = chains_for(want)
        hinted = [c for c in cands if any(h in c[0].lower() for h in _HINTS.get(g, (g,)))]
        pick = hinted if len(hinted) == 1 else (cands if len(cands) == 1 else None)
        if pick is None:
            # two modules of one class and one shape (none in the reference's architecture): creation order = scope numbering
            pick = sorted(hinted or cands, key=lambda c: _natural(c[0]))[:1]
        if not pick and g == "baseline":                  # built lazily by the first REINFORCE call (model.py:218-231): may be absent
            continue
        if not pick:
            raise TFCheckpointError(f"no chain of layers with shapes {want} for {g!r}; layers found: "
                                    + ", ".join(f"{sc} {model_vars[d['w']]}" for sc, d in sorted(layers.items())))
        for wn, sc in zip(ws, pick[0]):
            out[wn] = layers[sc]["w"]; out[wn[:-1] + "b"] = layers[sc]["b"]; used.add(sc)
    if "lstm" in groups:
        def only(pred, what):
            c = [n for n in model_vars if n not in out.values() and pred(n)]
            if len(c) != 1:
                raise TFCheckpointError(f"{what}: expected one candidate, found {sorted(c)}")
            return c[0]
        out["lstm/w_gates"] = only(lambda n: n.endswith("w_gates") and model_vars[n] == tuple(shapes["lstm/w_gates"]), "lstm/w_gates")
        out["lstm/b_gates"] = only(lambda n: n.endswith("b_gates") and model_vars[n] == tuple(shapes["lstm/b_gates"]), "lstm/b_gates")
        init = sorted((n for n in model_vars if "initial_state" in n and model_vars[n] == tuple(shapes["lstm/h0"])), key=_natural)
        if len(init) == 2:                                # snt.LSTM.initial_state(trainable=True): (hidden, cell) in that order
            out["lstm/h0"], out["lstm/c0"] = init
        elif init:
            raise TFCheckpointError(f"trainable initial state: expected two tensors, found {init}")
    return out


def import_tf_checkpoint(prefix: str, shapes: "Dict[str, Tuple[int, ...]]",
                         name_map: "Optional[Dict[str, str] | Callable[[str], str]]" = None, verify: str = "all") -> Dict[str, np.ndarray]:
    """Parameters of a reference checkpoint under the engine's names, ready for `Engine.load_parameters`
    (scripts/multi_mnist.py --init-from-tf-ckpt): Sonnet's Linear `w [in, out]`, `b [out]` and LSTM `w_gates [in + hidden, 4 hidden]` (gate order
    i, j, f, o; the forget bias 1 is added at run time, not stored) are the engine's own layouts -- no transposition.
    Engine names absent from the map (e.g. the baseline before it was built, an untrained initial state) are left out."""
    _, entries = read_index(prefix, verify != "none")
    tf_shapes = {n: e["shape"] for n, e in entries.items()}
    if name_map is None:
        m = default_name_map(shapes, tf_shapes)
    elif callable(name_map):
        m = {k: name_map(k) for k in shapes}
    else:
        m = dict(name_map)
    for k, tfn in m.items():
        if k not in shapes:
            raise KeyError(f"{k!r} is not an engine parameter")
        if tfn not in entries:
            raise KeyError(f"{tfn!r} (for {k}) is not in the checkpoint")
        if tuple(entries[tfn]["shape"]) != tuple(shapes[k]):
            raise TFCheckpointError(f"{k}: engine shape {tuple(shapes[k])}, checkpoint {tfn} has {tuple(entries[tfn]['shape'])}")
    raw = read_bundle(prefix, sorted(set(m.values())), verify)
    return {k: raw[tfn].astype(np.float32) for k, tfn in m.items()}


def mapping_report(prefix: str, shapes: "Dict[str, Tuple[int, ...]]", name_map: "Optional[Dict[str, str]]" = None) -> "Dict[str, list]":
    """What an import leaves behind: engine parameters the name map does not cover (they keep their random initial values)
    and trainable-looking checkpoint variables nobody uses (rank >= 1, not an optimiser slot, not global_step) -- the two silent ways a
    wrong guess about Sonnet's variable names can go unnoticed.  scripts/multi_mnist.py prints both lists."""
    _, entries = read_index(prefix, False)
    tf_shapes = {n: e["shape"] for n, e in entries.items()}
    m = dict(name_map) if name_map is not None else default_name_map(shapes, tf_shapes)
    used = set(m.values())
    is_slot = lambda n: "/RMSProp" in n or n.rsplit("/", 1)[-1].startswith("RMSProp")
    unused = sorted(n for n, sh in tf_shapes.items() if n not in used and len(sh) >= 1 and not is_slot(n) and "global_step" not in n)
    return {"unmapped_engine_parameters": sorted(k for k in shapes if k not in m), "unused_checkpoint_variables": unused}


def import_tf_optimizer_slots(prefix: str, shapes: "Dict[str, Tuple[int, ...]]", name_map: "Optional[Dict[str, str]]" = None,
                              verify: str = "all") -> "Dict[str, Dict[str, np.ndarray]]":
    """The RMSProp slots a Saver wrote beside the variables (model.py:265: `tf.train.RMSPropOptimizer(momentum=.9, centered=True)` for the
    model AND the baseline variables): TensorFlow names a slot `<variable>/<optimizer name>` and numbers repeats, and RMSProp creates its
    slots in the order rms, mg (centred only), momentum -- `<var>/RMSProp`, `<var>/RMSProp_1`, `<var>/RMSProp_2` (uncentred:
    `RMSProp` = rms, `RMSProp_1` = momentum).  Returns {"ms": {...}, "mg": {...}, "mom": {...}} under the engine's names for every
    variable whose slots are ALL in the file (`AIREngine.load_optimizer_slots`); same caveat as the module: unvalidated against TensorFlow."""
    _, entries = read_index(prefix, verify != "none")
    m = dict(name_map) if name_map is not None else default_name_map(shapes, {n: e["shape"] for n, e in entries.items()})
    out = {"ms": {}, "mg": {}, "mom": {}}
    want = {}
    for k, tfn in m.items():
        have = [s for s in ("RMSProp", "RMSProp_1", "RMSProp_2") if tfn + "/" + s in entries]
        if len(have) == 3:
            want[k] = dict(ms=tfn + "/RMSProp", mg=tfn + "/RMSProp_1", mom=tfn + "/RMSProp_2")
        elif len(have) == 2:
            want[k] = dict(ms=tfn + "/RMSProp", mom=tfn + "/RMSProp_1")
    names = sorted({n for d in want.values() for n in d.values()})
    for n in names:
        k = next(k for k, d in want.items() if n in d.values())
        if tuple(entries[n]["shape"]) != tuple(shapes[k]):
            raise TFCheckpointError(f"slot {n}: shape {tuple(entries[n]['shape'])}, variable {k} has {tuple(shapes[k])}")
    raw = read_bundle(prefix, names, verify)
    for k, d in want.items():
        for slot, n in d.items():
            out[slot][k] = raw[n].astype(np.float32)
    return out


def global_step_of(prefix: str) -> Optional[int]:
    """The Saver's `global_step` tensor if the checkpoint holds one (multi_mnist.py:128 would honour it)."""
    _, entries = read_index(prefix)
    for n in entries:
        if n.rsplit("/", 1)[-1] == "global_step":
            return int(read_bundle(prefix, [n])[n])
    return None
