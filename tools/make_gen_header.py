#!/usr/bin/env python3
"""The constant part of every compressed meta-block the on-device stream generator (brotli-rs_amd/csrc/brx_gen.hip)
writes: one block type per category, NPOSTFIX = NDIRECT = 0, one literal tree, one distance tree, and three STATIC
prefix codes in their complex-code transmission form (RFC 7932 section 3.5) -- literals: all 256 symbols, 8 bits;
insert&copy: all 704 symbols, 9 bits for symbols 0..319, 10 bits above; distance: 64 symbols, 6 bits.  Built with the
test suite's bit-level assembler (tests/craft.py) and committed as brotli-rs_amd/tables/gen_header.bin: u32 number of bits
(little endian), then the bits, LSB first.  Re-run after changing anything here; tools/bin2h.py checks the CRC."""
import os
import struct
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from craft import Bits, complex_code, uniform_lengths  # noqa: E402

b = Bits()
b.put(0, 1); b.put(0, 1); b.put(0, 1)   # NBLTYPESL, NBLTYPESI, NBLTYPESD = 1
b.put(0, 2)                              # NPOSTFIX = 0
b.put(0, 4)                              # NDIRECT >> NPOSTFIX = 0
b.put(0, 2)                              # context mode of the one literal block type (irrelevant: one tree)
b.put(0, 1)                              # NTREESL = 1
b.put(0, 1)                              # NTREESD = 1
complex_code(b, [8] * 256)
complex_code(b, uniform_lengths(704))
complex_code(b, uniform_lengths(64))
blob = struct.pack("<I", b.n) + b.bytes()
out = os.path.join(ROOT, "brotli-rs_amd", "tables", "gen_header.bin")
open(out, "wb").write(blob)
print("%s: %d bits, %d bytes, crc32 %#010x" % (out, b.n, len(blob), zlib.crc32(blob)))
