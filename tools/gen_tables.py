#!/usr/bin/env python3
"""Derive the Brotli constant tables from the format specification text and
write them as binary data blobs under brotli-rs_amd/tables/.

Input  : the format spec shipped with the reference,
         docs/draft-alakuijala-brotli-07.txt  (NOT the reference's .rs files).
Output : brotli-rs_amd/tables/{dictionary,context_lut,transforms}.bin
Every table carries a CRC-32 published in the spec text itself; the script
refuses to write anything whose CRC does not match.

  DICT        spec Appendix A hex dump          122784 B  CRC 0x5136cb04
  Lut0/1/2    spec section 7.1 tables           3 x 256 B CRC 0x8e91efb7 / 0xd01a32f4 / 0x0dd7a0d6
  transforms  spec Appendix B (prefix\\0 op suffix\\0) x 121 = 648 B  CRC 0x3d965f81

This runs only in the build container (the spec text lives in /root/reference);
the blobs it writes are committed, so nothing at build/run time needs the spec.
The insert/copy, block-count and NDBITS/DOFFSET tables are tiny closed-form
tables and are written directly in the C sources (with their spec sections cited).
"""
import os
import re
import sys
import zlib

SPEC = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/docs/draft-alakuijala-brotli-07.txt"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "brotli-rs_amd", "tables")

PAGE_NOISE = re.compile(r"^(Alakuijala & Szabadka|Internet-Draft)")


def spec_lines():
    with open(SPEC, "r", encoding="latin-1") as f:
        return [l.rstrip("\n") for l in f]


def gen_dictionary(lines):
    start = next(i for i, l in enumerate(lines) if l.startswith("Appendix A. Static dictionary data"))
    end = next(i for i, l in enumerate(lines) if l.startswith("Appendix B. List of word transformations"))
    hexrun = []
    for l in lines[start:end]:
        s = l.strip()
        if re.fullmatch(r"[0-9a-f]+", s) and len(s) % 2 == 0 and len(s) >= 2:
            hexrun.append(s)
    data = bytes.fromhex("".join(hexrun))
    assert len(data) == 122784, len(data)
    assert zlib.crc32(data) == 0x5136CB04, hex(zlib.crc32(data))
    return data


def gen_luts(lines):
    out = []
    for name, crc in (("Lut0", 0x8E91EFB7), ("Lut1", 0xD01A32F4), ("Lut2", 0x0DD7A0D6)):
        i = next(k for k, l in enumerate(lines) if l.strip() == name + " :=")
        vals = []
        k = i + 1
        while len(vals) < 256:
            l = lines[k]
            k += 1
            if not l.strip() or PAGE_NOISE.match(l.strip()):
                continue
            if re.fullmatch(r"[\d,\s]+", l):
                vals += [int(x) for x in l.replace(",", " ").split()]
        assert len(vals) == 256
        b = bytes(vals)
        assert zlib.crc32(b) == crc, (name, hex(zlib.crc32(b)))
        out.append(b)
    return b"".join(out)


OPS = {"Identity": 0, "UppercaseFirst": 1, "UppercaseAll": 2}
for n in range(1, 10):
    OPS["OmitFirst%d" % n] = 2 + n
    OPS["OmitLast%d" % n] = 11 + n


def c_unescape(s):
    return s.encode("latin-1").decode("unicode_escape").encode("latin-1")


def gen_transforms(lines):
    start = next(i for i, l in enumerate(lines) if l.startswith("Appendix B. List of word transformations"))
    row = re.compile(r'^\s*(\d+)\s+"((?:[^"\\]|\\.)*)"\s+(\w+)\s+"((?:[^"\\]|\\.)*)"\s*$')
    rows = {}
    for l in lines[start:]:
        m = row.match(l)
        if m:
            rows[int(m.group(1))] = (c_unescape(m.group(2)), OPS[m.group(3)], c_unescape(m.group(4)))
    assert sorted(rows) == list(range(121)), len(rows)
    blob = b"".join(p + b"\0" + bytes([op]) + s + b"\0" for p, op, s in (rows[i] for i in range(121)))
    assert len(blob) == 648, len(blob)
    assert zlib.crc32(blob) == 0x3D965F81, hex(zlib.crc32(blob))
    return blob


def main():
    lines = spec_lines()
    os.makedirs(OUT, exist_ok=True)
    for name, data in (("dictionary.bin", gen_dictionary(lines)),
                       ("context_lut.bin", gen_luts(lines)),
                       ("transforms.bin", gen_transforms(lines))):
        with open(os.path.join(OUT, name), "wb") as f:
            f.write(data)
        print("%-18s %7d B  crc32 %08x" % (name, len(data), zlib.crc32(data)))


if __name__ == "__main__":
    main()
