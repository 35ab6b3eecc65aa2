"""A reader of unaligned BAM files written from the SAM specification alone (sections 4.1 BGZF, 4.2 the BAM format) — test
infrastructure: what the library writes without htslib is read back here without htslib, block by block and field by field.
records(path) -> (header text, [record dict]); every structural rule the specification states is asserted on the way."""
import struct
import zlib

BGZF_EOF = bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0])
LETTERS = "=ACMGRSVTWYHKDBN"


def blocks(data):
    """the uncompressed data of a BGZF file; every block a gzip member with the 'BC' subfield holding its size - 1"""
    out, at, count = [], 0, 0
    while at < len(data):
        assert data[at:at + 4] == b"\x1f\x8b\x08\x04", "no BGZF block header at %d" % at
        xlen = struct.unpack_from("<H", data, at + 10)[0]
        extra, size = data[at + 12:at + 12 + xlen], None
        k = 0
        while k + 4 <= len(extra):
            tag, length = extra[k:k + 2], struct.unpack_from("<H", extra, k + 2)[0]
            if tag == b"BC":
                assert length == 2
                size = struct.unpack_from("<H", extra, k + 4)[0] + 1
            k += 4 + length
        assert size is not None and size <= 65536 and at + size <= len(data), "block size"
        payload = data[at + 12 + xlen:at + size - 8]
        piece = zlib.decompress(payload, -15)
        crc, isize = struct.unpack_from("<II", data, at + size - 8)
        assert isize == len(piece) <= 65536 and crc == zlib.crc32(piece)
        out.append(piece)
        at += size
        count += 1
    assert data.endswith(BGZF_EOF), "no end of file block"
    return b"".join(out), count


def records(path):
    data, count = blocks(open(path, "rb").read())
    assert data[:4] == b"BAM\x01"
    l_text = struct.unpack_from("<i", data, 4)[0]
    text = data[8:8 + l_text].decode()
    n_ref = struct.unpack_from("<i", data, 8 + l_text)[0]
    assert n_ref == 0
    at = 12 + l_text
    out = []
    while at < len(data):
        block_size = struct.unpack_from("<i", data, at)[0]
        ref, pos, l_name, mapq, bin_, n_cigar, flag, l_seq, next_ref, next_pos, tlen = struct.unpack_from("<iiBBHHHiiii", data, at + 4)
        p = at + 36
        name = data[p:p + l_name]
        assert name.endswith(b"\0") and name.count(b"\0") == 1
        p += l_name + 4 * n_cigar
        packed = data[p:p + (l_seq + 1) // 2]
        sequence = "".join(LETTERS[b >> 4] + LETTERS[b & 15] for b in packed)[:l_seq]
        p += (l_seq + 1) // 2
        quality = data[p:p + l_seq]
        p += l_seq
        tags, end = [], at + 4 + block_size
        while p < end:
            tag, kind = data[p:p + 2].decode(), chr(data[p + 2])
            p += 3
            if kind == "Z":
                stop = data.index(b"\0", p)
                tags.append((tag, kind, data[p:stop].decode()))
                p = stop + 1
            elif kind == "i":
                tags.append((tag, kind, struct.unpack_from("<i", data, p)[0]))
                p += 4
            elif kind == "f":
                tags.append((tag, kind, struct.unpack_from("<f", data, p)[0]))
                p += 4
            else:
                raise AssertionError("tag type " + kind)
        assert p == end
        out.append({"name": name[:-1].decode(), "flag": flag, "ref": ref, "pos": pos, "mapq": mapq, "bin": bin_, "n_cigar": n_cigar, "next_ref": next_ref, "next_pos": next_pos, "tlen": tlen,
                    "sequence": sequence, "quality": bytes(quality), "tags": tags})
        at = end
    assert at == len(data)
    return text, out, count
