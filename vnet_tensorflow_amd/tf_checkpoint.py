"""Reader / writer of TensorFlow checkpoints (`tf.train.Saver`, format V2: "tensor bundle") without TensorFlow.

The reference saves and restores its training state with `tf.train.Saver` (model.py:689-699, 758-764, 806-808, 1138-1139): files
`checkpoint-<step>.index` + `checkpoint-<step>.data-00000-of-00001`, the state file `checkpoint-latest`.  This module lets a user of the
reference bring those checkpoints along (`read(prefix)` -> {variable name: ndarray}; `image2label.load_tf_checkpoint`) and take
weights back (`write(prefix, tensors)`).  TensorFlow is not installable here, so the format is restated from its published
definition -- tensorflow/core/util/tensor_bundle (BundleHeaderProto / BundleEntryProto, tensor_bundle.proto) on top of
tensorflow/core/lib/io/table (the LevelDB table format: prefix-compressed blocks with restart points, 5-byte block trailers with a
masked CRC-32C, a 48-byte footer ending in the magic 0xdb4775248b80fb57; TF writes the bundle index uncompressed,
tensor_bundle.cc `options.compression = table::kNoCompression`) -- and pinned only by its own round trip, by the published CRC-32C
check value and by hand-assembled known-answer bytes (tests/test_tf_checkpoint.py): UNPINNED by a real TensorFlow file, like the rest
of the parity story.  Snappy-compressed index blocks (never written by TF's bundle writer) are refused by name.

Variable names are the reference's own (`vnet/input_layer/...`); optimiser slots follow tf.train.AdamOptimizer
(`<variable>/Adam` = m, `<variable>/Adam_1` = v, `beta1_power`, `beta2_power`) and tf.train.MomentumOptimizer (`<variable>/Momentum`);
`global_step` (int64 scalar) and `start_epoch` (int32 [1], model.py:668)."""
import os
import struct

import numpy as np

MAGIC = 0xdb4775248b80fb57
_MASK_DELTA = 0xa282ead8
# tensorflow/core/framework/types.proto: DataType
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64, 10: np.bool_,
           17: np.uint16, 19: np.float16, 22: np.uint32, 23: np.uint64}
_DTYPE_IDS = {np.dtype(v): k for k, v in _DTYPES.items()}


def _crc_table():
    t = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82f63b78 if c & 1 else c >> 1
        t.append(c)
    return t


_TABLE = _crc_table()


def crc32c(data, crc=0):
    """CRC-32C (Castagnoli), the checksum of the table format and of every tensor in a bundle (check value: crc32c(b"123456789") = 0xe3069283)."""
    c = crc ^ 0xffffffff
    t = _TABLE
    for b in bytes(data):
        c = t[(c ^ b) & 0xff] ^ (c >> 8)
    return c ^ 0xffffffff


def mask_crc(c):
    """leveldb / TF crc32c::Mask: stored CRCs are rotated and offset so that a CRC of data that embeds CRCs stays well distributed."""
    return ((((c >> 15) | (c << 17)) & 0xffffffff) + _MASK_DELTA) & 0xffffffff


def unmask_crc(m):
    r = (m - _MASK_DELTA) & 0xffffffff
    return ((r >> 17) | (r << 15)) & 0xffffffff


# ---- varints / the handful of protobuf fields a bundle uses ----------------------------------------------------------------------
def _put_varint(v):
    out = bytearray()
    v &= (1 << 64) - 1
    while v >= 0x80:
        out.append((v & 0x7f) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def _get_varint(buf, pos):
    shift = v = 0
    while True:
        b = buf[pos]
        pos += 1
        v |= (b & 0x7f) << shift
        if not b & 0x80:
            return v, pos
        shift += 7


def _pb_fields(buf):
    """[(field number, wire type, value)] of one protobuf message (varint, 64-bit, length-delimited and 32-bit wire types)."""
    pos, out = 0, []
    while pos < len(buf):
        key, pos = _get_varint(buf, pos)
        f, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _get_varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wt == 2:
            n, pos = _get_varint(buf, pos)
            v = bytes(buf[pos:pos + n])
            pos += n
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        else:
            raise ValueError("tensor bundle: unsupported protobuf wire type %d" % wt)
        out.append((f, wt, v))
    return out


def _entry_proto(dtype_id, shape, offset, size, crc_masked):
    """BundleEntryProto {dtype = 1, shape = 2 (TensorShapeProto: repeated Dim dim = 2 {size = 1}), shard_id = 3, offset = 4, size = 5,
    crc32c = 6 (fixed32)}; proto3: zero-valued scalars are omitted."""
    shp = b"".join(b"\x12" + _put_varint(len(d)) + d for d in ((b"\x08" + _put_varint(int(s))) if s else b"" for s in shape))
    out = b"\x08" + _put_varint(dtype_id) + b"\x12" + _put_varint(len(shp)) + shp
    if offset:
        out += b"\x20" + _put_varint(offset)
    if size:
        out += b"\x28" + _put_varint(size)
    return out + b"\x35" + struct.pack("<I", crc_masked)


def _parse_entry(buf):
    e = {"dtype": 0, "shape": [], "shard_id": 0, "offset": 0, "size": 0, "crc32c": 0, "slices": 0}
    for f, _wt, v in _pb_fields(buf):
        if f == 1:
            e["dtype"] = v
        elif f == 2:
            for f2, _w2, dim in _pb_fields(v):
                if f2 == 2:
                    size = 0
                    for f3, _w3, sv in _pb_fields(dim):
                        if f3 == 1:
                            size = sv - (1 << 64) if sv >> 63 else sv
                    e["shape"].append(size)
                elif f2 == 3 and dim:
                    raise ValueError("tensor bundle: tensor of unknown rank")
        elif f == 3:
            e["shard_id"] = v
        elif f == 4:
            e["offset"] = v
        elif f == 5:
            e["size"] = v
        elif f == 6:
            e["crc32c"] = v
        elif f == 7:
            e["slices"] += 1
    return e


# ---- the table (tensorflow/core/lib/io: block.cc, format.cc, table_builder.cc) -------------------------------------------------
def _block_entries(block):
    """(key, value) pairs of one block: entries [shared][non_shared][value_len] key-suffix value ..., restart offsets, restart count."""
    nrestart = struct.unpack_from("<I", block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * nrestart
    pos, key = 0, b""
    while pos < end:
        shared, pos = _get_varint(block, pos)
        non_shared, pos = _get_varint(block, pos)
        vlen, pos = _get_varint(block, pos)
        key = key[:shared] + bytes(block[pos:pos + non_shared])
        pos += non_shared
        yield key, bytes(block[pos:pos + vlen])
        pos += vlen


def _read_block(buf, offset, size, verify):
    body, ctype = buf[offset:offset + size], buf[offset + size]
    if verify:
        want = unmask_crc(struct.unpack_from("<I", buf, offset + size + 1)[0])
        if crc32c(buf[offset:offset + size + 1]) != want:
            raise ValueError("tensor bundle index: block checksum mismatch at offset %d" % offset)
    if ctype == 1:
        raise ValueError("tensor bundle index: snappy-compressed block (tf's BundleWriter writes kNoCompression; this reader has no snappy)")
    if ctype != 0:
        raise ValueError("tensor bundle index: unknown block compression type %d" % ctype)
    return body


def _build_block(items, restart_interval):
    out, restarts, last, n = bytearray(), [], b"", 0
    for k, v in items:
        shared = 0
        if n % restart_interval == 0:
            restarts.append(len(out))
        else:
            m = min(len(k), len(last))
            while shared < m and k[shared] == last[shared]:
                shared += 1
        out += _put_varint(shared) + _put_varint(len(k) - shared) + _put_varint(len(v)) + k[shared:] + v
        last, n = k, n + 1
    if not restarts:
        restarts.append(0)
    for r in restarts:
        out += struct.pack("<I", r)
    out += struct.pack("<I", len(restarts))
    return bytes(out)


def _emit_block(f, body):
    off = f.tell()
    f.write(body + b"\x00" + struct.pack("<I", mask_crc(crc32c(body + b"\x00"))))
    return off, len(body)


# ---- public API ------------------------------------------------------------------------------------------------------------------
def list_variables(prefix, verify_index=True):
    """{name: (dtype, shape)} of the bundle `prefix`(.index / .data-00000-of-00001), in the file's (sorted) order."""
    return {k: (_DTYPES[e["dtype"]], tuple(e["shape"])) for k, e in _entries(prefix, verify_index).items()}


def _entries(prefix, verify_index=True):
    buf = open(prefix + ".index", "rb").read()
    if len(buf) < 48 or struct.unpack_from("<Q", buf, len(buf) - 8)[0] != MAGIC:
        raise ValueError("%s.index is not a TensorFlow tensor-bundle index (bad table magic)" % prefix)
    foot = buf[len(buf) - 48:]
    _mo, p = _get_varint(foot, 0)
    _ms, p = _get_varint(foot, p)
    io, p = _get_varint(foot, p)
    isz, p = _get_varint(foot, p)
    entries, header = {}, None
    for _k, handle in _block_entries(_read_block(buf, io, isz, verify_index)):
        off, q = _get_varint(handle, 0)
        size, q = _get_varint(handle, q)
        for key, val in _block_entries(_read_block(buf, off, size, verify_index)):
            if key == b"":
                header = val
                continue
            entries[key.decode("utf-8")] = _parse_entry(val)
    if header is None:
        raise ValueError("%s.index has no bundle header entry" % prefix)
    hd = {"num_shards": 0, "endianness": 0}
    for f, _wt, v in _pb_fields(header):
        if f == 1:
            hd["num_shards"] = v
        elif f == 2:
            hd["endianness"] = v
    if hd["endianness"] != 0:
        raise ValueError("big-endian tensor bundle: not supported")
    for k, e in entries.items():
        e["num_shards"] = hd["num_shards"]
    return entries


def read(prefix, names=None, verify="small"):
    """{variable name: ndarray} from a TF V2 checkpoint.  names: restrict to these.  verify: "all" checks every tensor's CRC-32C (pure
    Python: ~1 MB/s), "small" (default) only tensors up to 1 MiB (plus every index block), False none."""
    entries = _entries(prefix, verify_index=bool(verify))
    out, files = {}, {}
    for name, e in entries.items():
        if names is not None and name not in names:
            continue
        if e["slices"]:
            raise ValueError("tensor bundle: %s is stored as slices of a partitioned variable: not supported" % name)
        if e["dtype"] not in _DTYPES:
            raise ValueError("tensor bundle: %s has DataType %d (strings / variants / quantised types are not supported)" % (name, e["dtype"]))
        shard = "%s.data-%05d-of-%05d" % (prefix, e["shard_id"], max(e["num_shards"], 1))
        if shard not in files:
            files[shard] = open(shard, "rb")
        f = files[shard]
        f.seek(e["offset"])
        raw = f.read(e["size"])
        if len(raw) != e["size"]:
            raise ValueError("tensor bundle: %s truncated in %s" % (name, shard))
        if verify == "all" or (verify == "small" and e["size"] <= (1 << 20)):
            if crc32c(raw) != unmask_crc(e["crc32c"]):
                raise ValueError("tensor bundle: checksum mismatch for %s" % name)
        dt = np.dtype(_DTYPES[e["dtype"]])
        n = int(np.prod(e["shape"])) if e["shape"] else 1
        if n * dt.itemsize != e["size"]:
            raise ValueError("tensor bundle: %s: %d bytes for shape %s of %s" % (name, e["size"], e["shape"], dt))
        out[name] = np.frombuffer(raw, dtype=dt).reshape(e["shape"]).copy()
    for f in files.values():
        f.close()
    if names is not None:
        missing = [n for n in names if n not in out]
        if missing:
            raise KeyError("not in checkpoint %s: %s" % (prefix, ", ".join(missing[:5])))
    return out


def write(prefix, tensors, block_size=4096, restart_interval=16):
    """Write {name: array} as a TF V2 checkpoint (one shard): what `tf.train.Saver.restore` / `tf.train.load_checkpoint` read.  Keys are
    stored sorted (the table format requires it), tensors back to back in that order; every tensor and block carries its masked CRC-32C."""
    names = sorted(tensors, key=lambda s: s.encode("utf-8"))
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    items = [(b"", b"\x08\x01\x1a\x02\x08\x01")]      # BundleHeaderProto {num_shards = 1, endianness = LITTLE (0, omitted), version {producer = 1}}
    with open(prefix + ".data-00000-of-00001", "wb") as fd:
        for name in names:
            a = np.asarray(tensors[name])                                  # (not ascontiguousarray: it turns a scalar into shape (1,))
            if a.dtype not in _DTYPE_IDS:
                raise ValueError("tensor bundle: dtype %s of %s is not supported" % (a.dtype, name))
            raw = a.astype(a.dtype.newbyteorder("<"), copy=False).tobytes(order="C")
            off = fd.tell()
            fd.write(raw)
            items.append((name.encode("utf-8"), _entry_proto(_DTYPE_IDS[a.dtype], a.shape, off, len(raw), mask_crc(crc32c(raw)))))
    with open(prefix + ".index", "wb") as f:
        index, chunk, nbytes = [], [], 0
        for k, v in items + [(None, None)]:
            if k is None or (chunk and nbytes + len(k) + len(v) > block_size):
                off, size = _emit_block(f, _build_block(chunk, restart_interval))
                # index key: any string >= the block's last key and < the next block's first key: the last key itself
                index.append((chunk[-1][0], _put_varint(off) + _put_varint(size)))
                chunk, nbytes = [], 0
            if k is None:
                break
            chunk.append((k, v))
            nbytes += len(k) + len(v) + 3
        moff, msize = _emit_block(f, _build_block([], 1))                           # (empty) metaindex block
        ioff, isize = _emit_block(f, _build_block(index, 1))
        foot = _put_varint(moff) + _put_varint(msize) + _put_varint(ioff) + _put_varint(isize)
        f.write(foot + b"\x00" * (40 - len(foot)) + struct.pack("<Q", MAGIC))


# ---- the reference's training state <-> this package's ---------------------------------------------------------------------------
def split_training_state(tensors, variable_names, beta1=0.9):
    """Sort the tensors of a reference checkpoint into (variables, optimiser state, global_step, start_epoch).
    variable_names: the network's variables (trainable and moving statistics, SURVEY B.1 names).  Optimiser state: for
    tf.train.AdamOptimizer {"kind": "adam", "m": {name: array}, "v": {...}, "t": steps taken (= global_step; `training/beta1_power` = beta1^(t+1) cross-checks it)}; for
    tf.train.MomentumOptimizer {"kind": "momentum", "acc": {...}}; None if the checkpoint holds no slots."""
    names = set(variable_names)
    missing = [n for n in variable_names if n not in tensors]
    if missing:
        raise KeyError("checkpoint lacks %d of the network's %d variables, e.g. %s" % (len(missing), len(names), ", ".join(missing[:3])))
    variables = {n: tensors[n] for n in variable_names}
    m = {n[:-len("/Adam")]: a for n, a in tensors.items() if n.endswith("/Adam") and n[:-len("/Adam")] in names}
    v = {n[:-len("/Adam_1")]: a for n, a in tensors.items() if n.endswith("/Adam_1") and n[:-len("/Adam_1")] in names}
    acc = {n[:-len("/Momentum")]: a for n, a in tensors.items() if n.endswith("/Momentum") and n[:-len("/Momentum")] in names}
    opt = None
    gs = int(np.asarray(tensors["global_step"]).reshape(-1)[0]) if "global_step" in tensors else 0
    if m and v:
        # Adam's step count.  The reference calls optimizer.minimize(loss, global_step=...) inside tf.name_scope("training")
        # (model.py:647-662): minimize increments global_step once per apply, so t = global_step; the non-slot accumulators are
        # tf.Variables, which honour the name scope: `training/beta1_power` (a bare `beta1_power` from an un-scoped graph is accepted
        # too).  beta1_power = beta1^(t+1) is only a CROSS-CHECK, and only while it is a normal float32: 0.9^(t+1) goes denormal near
        # t ~ 830 and is exactly 0 past t ~ 987, where log(beta1_power) says nothing.
        t = gs
        key = next((k for k in ("training/beta1_power", "beta1_power") if k in tensors), None)
        if key is None:
            key = next((k for k in sorted(tensors) if k.endswith("/beta1_power")), None)
        if key is not None:
            p = float(np.asarray(tensors[key]).reshape(-1)[0])
            if 1e-30 < p < 1.0:
                tp = max(int(round(np.log(p) / np.log(beta1))) - 1, 0)
                if "global_step" not in tensors:
                    t = tp
                elif abs(tp - gs) > max(2, gs // 100):
                    raise ValueError("checkpoint: %s = %g says %d Adam steps, global_step says %d" % (key, p, tp, gs))
        opt = {"kind": "adam", "m": m, "v": v, "t": t}
    elif acc:
        opt = {"kind": "momentum", "acc": acc}
    ep = int(np.asarray(tensors["start_epoch"]).reshape(-1)[0]) if "start_epoch" in tensors else 0
    return variables, opt, gs, ep
