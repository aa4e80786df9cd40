#!/usr/bin/env python3
"""The constant part of every compressed meta-block the on-device stream generator (brotli-rs_amd/csrc/brx_gen.hip)
writes: one block type per category, NPOSTFIX = NDIRECT = 0, one literal tree, one distance tree, and three STATIC
prefix codes in their complex-code transmission form (RFC 7932 section 3.5) -- literals: all 256 symbols, 8 bits;
insert&copy: all 704 symbols, 9 bits for symbols 0..319, 10 bits above; distance: 64 symbols, 6 bits.  Built with the
test suite's bit-level assembler (tests/craft.py) and committed as brotli-rs_amd/tables/gen_header.bin: two records
(A: one literal block type; B: two, switching every 100 literals), each u32 number of bits (little endian), then the
bits, LSB first, padded to a multiple of 4 bytes.  Re-run after changing anything here; tools/bin2h.py checks the CRC."""
import os
import struct
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from craft import Bits, complex_code, simple_code, uniform_lengths  # noqa: E402

def tail(b):
    b.put(0, 1); b.put(0, 1)                 # NBLTYPESI, NBLTYPESD = 1
    b.put(0, 2)                              # NPOSTFIX = 0
    b.put(0, 4)                              # NDIRECT >> NPOSTFIX = 0


def codes(b):
    b.put(0, 1)                              # NTREESL = 1
    b.put(0, 1)                              # NTREESD = 1
    complex_code(b, [8] * 256)
    complex_code(b, uniform_lengths(704))
    complex_code(b, uniform_lengths(64))


# A: one literal block type
a = Bits()
a.put(0, 1)                                  # NBLTYPESL = 1
tail(a)
a.put(0, 2)                                  # context mode of the one literal block type (irrelevant: one tree)
codes(a)
# B: two literal block types taking turns every 100 literals (both use the one literal tree): the block-switch machinery
# with the cheapest possible codes -- the type code has ONE symbol (1 = "the next type"), the count code has ONE symbol
# (11 = 97 + 4 extra bits), so a switch costs 4 bits in the stream: the extra bits of the count, value 3
b2 = Bits()
b2.put(1, 1); b2.put(0, 3)                   # NBLTYPESL = 2
simple_code(b2, [1], 2)                      # block-type code over NBLTYPES + 2 = 4 symbols
simple_code(b2, [11], 5)                     # block-count code over 26 symbols
b2.put(3, 4)                                 # first block count: 97 + 3
tail(b2)
b2.put(0, 2); b2.put(0, 2)                   # context modes of the two literal block types
codes(b2)
blob = b""
for x in (a, b2):
    body = x.bytes()
    body += b"\0" * ((-len(body)) % 4)
    blob += struct.pack("<I", x.n) + body
out = os.path.join(ROOT, "brotli-rs_amd", "tables", "gen_header.bin")
open(out, "wb").write(blob)
print("%s: %d + %d bits, %d bytes, crc32 %#010x" % (out, a.n, b2.n, len(blob), zlib.crc32(blob)))
