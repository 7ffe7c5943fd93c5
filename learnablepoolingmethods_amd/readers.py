"""YT8M frame-level reader (reference: readers.py:134-271, utils.py:28-43): TFRecord files of tf.train.SequenceExample
protos -> quantised frame matrices, frame counts, dense labels.  TensorFlow is not needed: the TFRecord framing
(length, masked CRC-32C, payload, masked CRC-32C) and the four protobuf messages involved are decoded by hand from the
public wire formats.

Unlike the reference, the reader hands the frames on QUANTISED (uint8, 1 byte per feature): dequantisation
(``utils.Dequantize``), the zero padding past ``num_frames`` and the input L2 normalisation of the training step are one
HIP kernel on the device (``ops.dequantize_l2_normalize``), so the host->device copy and the first HBM read carry 4x fewer
bytes.  ``dequantize=True`` reproduces the reference's float32 ``[max_frames, sum(feature_sizes)]`` matrix on the host.

There are no TFRecord fixtures in the reference; ``write_tfrecord`` / ``make_sequence_example`` produce files in the same
format for the round-trip tests and for synthetic data."""
from __future__ import annotations

import struct
from typing import Dict, Iterable, Iterator, List, Sequence, Tuple

import numpy as np
import torch

from . import utils

# ---- CRC-32C (Castagnoli), table driven, and the TFRecord mask ---------------------------------------------------------
_CRC_TABLE = None


def _crc_table():
    global _CRC_TABLE
    if _CRC_TABLE is None:
        poly, tab = 0x82F63B78, []
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ poly if c & 1 else c >> 1
            tab.append(c)
        _CRC_TABLE = tab
    return _CRC_TABLE


def crc32c(data: bytes) -> int:
    tab, c = _crc_table(), 0xFFFFFFFF
    for b in data:
        c = tab[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc32c(data: bytes) -> int:
    c = crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def read_tfrecord(path: str, verify_crc: bool = False) -> Iterator[bytes]:
    """Yields the payload of every record of a TFRecord file."""
    with open(path, "rb") as f:
        while True:
            head = f.read(12)
            if not head:
                return
            if len(head) < 12:
                raise IOError(f"{path}: truncated record header")
            (length,), (lcrc,) = struct.unpack("<Q", head[:8]), struct.unpack("<I", head[8:])
            if verify_crc and masked_crc32c(head[:8]) != lcrc:
                raise IOError(f"{path}: corrupt record length")
            data = f.read(length)
            tail = f.read(4)
            if len(data) < length or len(tail) < 4:
                raise IOError(f"{path}: truncated record")
            if verify_crc and masked_crc32c(data) != struct.unpack("<I", tail)[0]:
                raise IOError(f"{path}: corrupt record payload")
            yield data


def write_tfrecord(path: str, records: Iterable[bytes]) -> None:
    with open(path, "wb") as f:
        for data in records:
            head = struct.pack("<Q", len(data))
            f.write(head + struct.pack("<I", masked_crc32c(head)) + data + struct.pack("<I", masked_crc32c(data)))


# ---- protobuf wire format (only what tf.train.SequenceExample needs) ----------------------------------------------------
def _varint(buf: bytes, i: int) -> Tuple[int, int]:
    shift = val = 0
    while True:
        b = buf[i]
        i += 1
        val |= (b & 0x7F) << shift
        if not b & 0x80:
            return val, i
        shift += 7


def _fields(buf: bytes) -> Iterator[Tuple[int, int, object]]:
    """(field number, wire type, value) for every field of a message; length-delimited values come as memoryviews."""
    i, n = 0, len(buf)
    mv = memoryview(buf)
    while i < n:
        key, i = _varint(buf, i)
        num, wt = key >> 3, key & 7
        if wt == 0:
            v, i = _varint(buf, i)
        elif wt == 2:
            ln, i = _varint(buf, i)
            v = mv[i:i + ln]
            i += ln
        elif wt == 5:
            v = mv[i:i + 4]
            i += 4
        elif wt == 1:
            v = mv[i:i + 8]
            i += 8
        else:
            raise ValueError(f"unsupported protobuf wire type {wt}")
        yield num, wt, v


def _parse_feature(buf) -> Tuple[str, list]:
    """tf.train.Feature: oneof bytes_list = 1 / float_list = 2 / int64_list = 3."""
    for num, _, v in _fields(bytes(buf)):
        if num == 1:
            return "bytes", [bytes(x) for n2, _, x in _fields(bytes(v)) if n2 == 1]
        if num == 3:
            vals = []
            for n2, wt, x in _fields(bytes(v)):
                if n2 != 1:
                    continue
                if wt == 0:                                   # unpacked
                    vals.append(x)
                else:                                         # packed varints
                    xb, j = bytes(x), 0
                    while j < len(xb):
                        val, j = _varint(xb, j)
                        vals.append(val)
            return "int64", [val - (1 << 64) if val >= (1 << 63) else val for val in vals]
        if num == 2:
            vals = []
            for n2, wt, x in _fields(bytes(v)):
                if n2 == 1:
                    vals.extend(np.frombuffer(bytes(x), dtype="<f4").tolist())
            return "float", vals
    return "empty", []


def _parse_map(buf, parse_value) -> Dict[str, object]:
    out = {}
    for num, _, entry in _fields(bytes(buf)):
        if num != 1:
            continue
        key, val = None, None
        for n2, _, x in _fields(bytes(entry)):
            if n2 == 1:
                key = bytes(x).decode("utf-8")
            elif n2 == 2:
                val = parse_value(x)
        out[key] = val
    return out


def parse_sequence_example(serialized: bytes):
    """-> (context {name: (kind, values)}, feature_lists {name: [(kind, values), ...]})."""
    context, lists = {}, {}
    for num, _, v in _fields(serialized):
        if num == 1:
            context = _parse_map(v, _parse_feature)
        elif num == 2:
            lists = _parse_map(v, lambda fl: [_parse_feature(x) for n2, _, x in _fields(bytes(fl)) if n2 == 1])
    return context, lists


# ---- encoder (tests, synthetic data) ----------------------------------------------------------------------------------
def _enc_varint(v: int) -> bytes:
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _enc_ld(num: int, payload: bytes) -> bytes:
    return _enc_varint((num << 3) | 2) + _enc_varint(len(payload)) + payload


def _enc_bytes_feature(values: Sequence[bytes]) -> bytes:
    return _enc_ld(1, b"".join(_enc_ld(1, v) for v in values))


def _enc_int64_feature(values: Sequence[int]) -> bytes:
    return _enc_ld(3, _enc_ld(1, b"".join(_enc_varint(int(v)) for v in values)))


def make_sequence_example(video_id: str, labels: Sequence[int], features: Dict[str, np.ndarray]) -> bytes:
    """features: {name: uint8 [num_frames, feature_size]} -> serialized tf.train.SequenceExample (one bytes value per frame)."""
    ctx = _enc_ld(1, _enc_ld(1, b"id") + _enc_ld(2, _enc_bytes_feature([video_id.encode("utf-8")])))
    ctx += _enc_ld(1, _enc_ld(1, b"labels") + _enc_ld(2, _enc_int64_feature(labels)))
    fl = b""
    for name, mat in features.items():
        mat = np.ascontiguousarray(mat, dtype=np.uint8)
        flist = b"".join(_enc_ld(1, _enc_bytes_feature([row.tobytes()])) for row in mat)
        fl += _enc_ld(1, _enc_ld(1, name.encode("utf-8")) + _enc_ld(2, flist))
    return _enc_ld(1, ctx) + _enc_ld(2, fl)


# ---- the reader ---------------------------------------------------------------------------------------------------------
class BaseReader(object):
    """readers.py:59-66."""

    def prepare_reader(self, unused_filename_queue):
        raise NotImplementedError()


class YT8MFrameFeatureReader(BaseReader):
    """readers.py:134-271.  Same constructor; records come from files instead of a TF filename queue."""

    def __init__(self, num_classes=3862, feature_sizes=(1024, 128), feature_names=("rgb", "audio"), max_frames=300):
        assert len(feature_names) == len(feature_sizes), \
            "length of feature_names (={}) != length of feature_sizes (={})".format(len(feature_names), len(feature_sizes))
        assert len(feature_names) > 0, "No feature selected: feature_names is empty!"
        self.num_classes = num_classes
        self.feature_sizes = list(feature_sizes)
        self.feature_names = list(feature_names)
        self.max_frames = max_frames

    def prepare_serialized_examples(self, serialized_example: bytes, max_quantized_value=2, min_quantized_value=-2,
                                    dequantize=False):
        """-> (video_id, frames [max_frames, sum(feature_sizes)], labels bool [num_classes], num_frames).  frames is uint8
        (quantised, zero beyond num_frames) or, with dequantize=True, the reference's float32 matrix (readers.py:176-193)."""
        context, lists = parse_sequence_example(serialized_example)
        video_id = context["id"][1][0].decode("utf-8") if "id" in context else ""
        labels = np.zeros(self.num_classes, dtype=bool)
        for v in context.get("labels", ("int64", []))[1]:
            if 0 <= v < self.num_classes:                      # sparse_to_dense(validate_indices=False)
                labels[v] = True
        num_frames, mats = -1, []
        for name, size in zip(self.feature_names, self.feature_sizes):
            rows = [np.frombuffer(vals[0], dtype=np.uint8) for _, vals in lists[name]]
            mat = np.stack(rows).reshape(-1, size) if rows else np.zeros((0, size), dtype=np.uint8)
            n = min(mat.shape[0], self.max_frames)
            if num_frames == -1:
                num_frames = n
            elif n != num_frames:
                raise ValueError(f"{video_id}: feature '{name}' has {n} frames, expected {num_frames}")
            mats.append(mat[:n])
        q = np.zeros((self.max_frames, sum(self.feature_sizes)), dtype=np.uint8)
        q[:num_frames] = np.concatenate(mats, axis=1)
        if dequantize:
            f = np.zeros(q.shape, dtype=np.float32)
            f[:num_frames] = utils.Dequantize(q[:num_frames].astype(np.float32), max_quantized_value, min_quantized_value)
            return video_id, f, labels, num_frames
        return video_id, q, labels, num_frames

    def batches(self, files: Sequence[str], batch_size: int, drop_remainder: bool = False, verify_crc: bool = False):
        """Yields (ids, frames uint8 [B, max_frames, F], labels bool [B, V], num_frames int32 [B]) as torch tensors."""
        ids: List[str] = []
        q, y, nf = [], [], []

        def flush():
            out = (list(ids), torch.from_numpy(np.stack(q)), torch.from_numpy(np.stack(y)), torch.tensor(nf, dtype=torch.int32))
            ids.clear(); q.clear(); y.clear(); nf.clear()
            return out
        for path in files:
            for rec in read_tfrecord(path, verify_crc=verify_crc):
                vid, frames, labels, n = self.prepare_serialized_examples(rec)
                ids.append(vid); q.append(frames); y.append(labels); nf.append(n)
                if len(ids) == batch_size:
                    yield flush()
        if ids and not drop_remainder:
            yield flush()
