"""`th.save(tensor)` of a float64 tensor of a FIXED shape, byte for byte, without running the serializer per sample.

`create_dataset` (/root/reference/music_gan/create_dataset.py:52-62) writes every sample with `th.save(magn_phase.to(th.float64))`:
a zip container (records data.pkl, byteorder, data/0, version, .data/serialization_id, ...) in which only three things depend on the
sample: the payload of `data/0`, its CRC-32 (in the data descriptor behind the payload and in the central directory), and the
`serialization_id` record (20 digits of a hash over the records' CRCs, plus that record's own CRC).  `PtTemplate` saves ONE dummy
tensor through torch, locates those fields, and from then on a file is `prefix + payload + patched suffix` -- with the payload's
CRC-32 supplied by the caller (ops.crc32_of_float64 computes it on the GPU, where the sample already is).  Whatever torch version
is installed defines the container: the template is checked at construction against `th.save` of a second tensor, and `ok` is
False if the bytes differ (the caller then keeps calling `th.save`)."""
from __future__ import annotations

import io
import struct
import zipfile
import zlib

import numpy as np
import torch as th

_M64 = (1 << 64) - 1


def _hash_combine(seed: int, v: int) -> int:
    """c10::hash_combine on size_t."""
    return (seed ^ ((v + 0x9E3779B9 + ((seed << 6) & _M64) + (seed >> 2)) & _M64)) & _M64


def _save_bytes(t: th.Tensor) -> bytes:
    b = io.BytesIO()
    th.save(t, b)
    return b.getvalue()


class PtTemplate:
    def __init__(self, shape, dtype=th.float64):
        self.shape, self.dtype = tuple(shape), dtype
        self.ok = False
        try:
            self._build()
            self.ok = self._self_check()
        except Exception:  # noqa: BLE001  (an unexpected container layout: the caller falls back to th.save)
            self.ok = False

    # ------------------------------------------------------------------ layout discovery
    def _build(self) -> None:
        zero = th.zeros(self.shape, dtype=self.dtype)
        raw = _save_bytes(zero)
        zf = zipfile.ZipFile(io.BytesIO(raw))
        infos = zf.infolist()
        data = [i for i in infos if i.filename.endswith("data/0")]
        sid = [i for i in infos if i.filename.endswith(".data/serialization_id")]
        assert len(data) == 1 and len(sid) == 1 and infos[-1] is sid[0], "unexpected records"
        data, sid = data[0], sid[0]
        self.nbytes = data.file_size
        assert self.nbytes == zero.numel() * zero.element_size() and data.compress_type == 0 and (data.flag_bits & 8)

        def payload_offset(info):
            nlen, elen = struct.unpack("<HH", raw[info.header_offset + 26:info.header_offset + 30])
            return info.header_offset + 30 + nlen + elen

        self.pay = payload_offset(data)
        assert raw[self.pay:self.pay + self.nbytes] == bytes(self.nbytes)
        self.prefix = raw[:self.pay]
        self.suffix0 = bytearray(raw[self.pay + self.nbytes:])
        base = self.pay + self.nbytes  # file offset of suffix[0]

        def descriptor_crc_pos(info):  # 'PK\x07\x08' crc32 sizes, right behind the record's data
            p = payload_offset(info) + info.file_size
            assert raw[p:p + 4] == b"PK\x07\x08"
            return p + 4 - base

        def central_crc_pos(info):
            # walk the central directory: signature, 12 bytes, crc32 at +16, name length at +28
            p = raw.rfind(b"PK\x05\x06")
            cd_off = struct.unpack("<I", raw[p + 16:p + 20])[0]
            if cd_off == 0xFFFFFFFF:  # zip64: the real offset sits in the zip64 end-of-central-directory record
                q = raw.rfind(b"PK\x06\x06")
                cd_off = struct.unpack("<Q", raw[q + 48:q + 56])[0]
            q = cd_off
            while raw[q:q + 4] == b"PK\x01\x02":
                nlen, elen, clen = struct.unpack("<HHH", raw[q + 28:q + 34])
                name = raw[q + 46:q + 46 + nlen].decode()
                if name == info.filename:
                    return q + 16 - base
                q += 46 + nlen + elen + clen
            raise AssertionError("record not in the central directory")

        self.pos_data_desc, self.pos_data_cd = descriptor_crc_pos(data), central_crc_pos(data)
        self.pos_sid = payload_offset(sid) - base
        self.pos_sid_desc, self.pos_sid_cd = descriptor_crc_pos(sid), central_crc_pos(sid)
        self.sid_len = sid.file_size
        assert self.sid_len == 40
        self.sid_head = raw[payload_offset(sid):payload_offset(sid) + 20]  # hash of the record names: the same for every file
        # CRCs of the records in write order, the sample-dependent one marked
        self.rec_crcs = [None if i is data else i.CRC for i in infos if i is not sid]
        assert min(self.pos_data_desc, self.pos_data_cd, self.pos_sid, self.pos_sid_desc, self.pos_sid_cd) >= 0

    def _self_check(self) -> bool:
        t = th.from_numpy(np.random.default_rng(1).random(self.shape)).to(self.dtype)
        want = _save_bytes(t)
        payload = t.numpy().tobytes()
        got = self.prefix + payload + self.suffix(zlib.crc32(payload) & 0xFFFFFFFF)
        return got == want

    # ------------------------------------------------------------------ per sample
    def suffix(self, payload_crc: int) -> bytes:
        """The bytes behind the payload for a sample whose float64 payload has this CRC-32."""
        s = bytearray(self.suffix0)
        comb = 0
        for c in self.rec_crcs:
            comb = _hash_combine(comb, payload_crc if c is None else c)
        sid = self.sid_head + str(comb).zfill(20).encode()
        sid_crc = zlib.crc32(sid) & 0xFFFFFFFF
        struct.pack_into("<I", s, self.pos_data_desc, payload_crc)
        struct.pack_into("<I", s, self.pos_data_cd, payload_crc)
        s[self.pos_sid:self.pos_sid + self.sid_len] = sid
        struct.pack_into("<I", s, self.pos_sid_desc, sid_crc)
        struct.pack_into("<I", s, self.pos_sid_cd, sid_crc)
        return bytes(s)

    def write(self, path: str, payload, payload_crc: int) -> None:
        """payload: a C-contiguous buffer of exactly `nbytes` bytes (the float64 sample)."""
        import os
        view = memoryview(payload).cast("B")
        assert len(view) == self.nbytes
        fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
        try:
            parts = [memoryview(self.prefix), view, memoryview(self.suffix(payload_crc))]
            total, done = sum(len(p) for p in parts), os.writev(fd, parts)
            while done < total:  # (short write: finish piecewise)
                off = done
                for p in parts:
                    if off < len(p):
                        done += os.write(fd, p[off:])
                        break
                    off -= len(p)
        finally:
            os.close(fd)
