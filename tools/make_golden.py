#!/usr/bin/env python3
"""Extract the reference's golden vectors into tests/golden/ (data only, no source).

Runs only in the build container (needs /root/reference).  Produces:

  tests/golden/data/*                 the reference's data/ fixtures (compressed streams and
                                      their expected outputs) -- data files its tests hold
  tests/golden/manifest.json          per stream: sizes, sha256, expected status
  tests/golden/inline_vectors.json    the byte vectors written inline in tests/lib.rs with the
                                      expected output, or the error-message substring of the
                                      #[should_panic(expected=...)] attribute
  tests/golden/transform_vectors.json the 121 (id, word, expected) triples of the transformation
                                      unit tests (src/transformation/mod.rs:211-1302)

Only vectors are extracted: literal inputs / expected values.  No code is copied.
"""
import hashlib
import json
import os
import re
import shutil

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "..", "tests", "golden")

# Expected error kind (status code) for the reject streams in data/.
# These are the kinds whose message contains the substring demanded by the matching
# should_panic test in tests/lib.rs (same bytes), see SURVEY Appendix C/D.
REJECT_STATUS = {
    "frewsxcv_01.compressed": 24, "frewsxcv_02.compressed": 8, "frewsxcv_03.compressed": 12,
    "frewsxcv_04.compressed": 1, "frewsxcv_05.compressed": 24, "frewsxcv_06.compressed": 23,
    "frewsxcv_07.compressed": 1, "frewsxcv_08.compressed": 24, "frewsxcv_09.compressed": 10,
}


def sha(b):
    return hashlib.sha256(b).hexdigest()


def rust_str(s):
    """Unescape a Rust string literal body (only the escapes the test files use)."""
    out = bytearray()
    i = 0
    b = s.encode("utf-8")
    while i < len(b):
        c = b[i]
        if c == 0x5C:  # backslash
            n = chr(b[i + 1])
            i += 2
            if n == "n":
                out.append(10)
            elif n == "t":
                out.append(9)
            elif n == "r":
                out.append(13)
            elif n == "0":
                out.append(0)
            elif n in "\"'\\":
                out.append(ord(n))
            elif n == "x":
                out.append(int(b[i:i + 2].decode(), 16))
                i += 2
            elif n == "u":
                j = b.index(b"}", i)
                out += chr(int(b[i + 1:j].decode(), 16)).encode("utf-8")
                i = j + 1
            elif n == "\n":  # line continuation
                while i < len(b) and chr(b[i]) in " \t\n":
                    i += 1
            else:
                raise ValueError("escape \\" + n)
        else:
            out.append(c)
            i += 1
    return bytes(out)


def data_fixtures():
    dst = os.path.join(GOLD, "data")
    os.makedirs(dst, exist_ok=True)
    src = os.path.join(REF, "data")
    manifest = []
    for name in sorted(os.listdir(src)):
        if ".compressed" not in name:
            continue
        base = name.split(".compressed")[0]
        comp = open(os.path.join(src, name), "rb").read()
        shutil.copyfile(os.path.join(src, name), os.path.join(dst, name))
        os.chmod(os.path.join(dst, name), 0o644)
        entry = {"stream": name, "in_bytes": len(comp), "in_sha256": sha(comp)}
        if name in REJECT_STATUS:
            entry["status"] = REJECT_STATUS[name]
        else:
            exp = open(os.path.join(src, base), "rb").read()
            if not os.path.exists(os.path.join(dst, base)):
                shutil.copyfile(os.path.join(src, base), os.path.join(dst, base))
                os.chmod(os.path.join(dst, base), 0o644)
            entry.update({"status": 0, "expected": base, "out_bytes": len(exp), "out_sha256": sha(exp)})
        manifest.append(entry)
    with open(os.path.join(GOLD, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1)
    return manifest


def inline_vectors():
    src = open(os.path.join(REF, "tests", "lib.rs"), encoding="utf-8").read()
    tests = re.split(r"\n#\[test\]\n", src)[1:]
    out = []
    for t in tests:
        m = re.search(r"fn (\w+)\(\)", t)
        name = m.group(1)
        if "Decompressor" not in t:
            continue
        line = src[:src.index("fn " + name + "()")].count("\n") + 1
        pm = re.search(r'should_panic\(expected\s*=\s*"([^"]*)"\)', t)
        vm = next((x for x in re.finditer(r"vec!\[([^\]]*)\]", t) if "0x" in x.group(1)), None)
        fm = re.search(r'File::open\("(data/[^"]+\.compressed[^"]*)"\)', t)
        bm = re.search(r'Decompressor::new\(&b"((?:[^"\\]|\\.)*)"', t)
        entry = {"test": name, "line": line}
        if bm:
            entry["input_hex"] = rust_str(bm.group(1)).hex()
        elif fm and vm is None:
            entry["input_file"] = os.path.basename(fm.group(1))
        else:
            entry["input_hex"] = bytes(int(x, 16) for x in re.findall(r"0x([0-9a-fA-F]{2})", vm.group(1))).hex()
        if pm:
            entry["expect_error_substring"] = pm.group(1)
        else:
            em = re.search(r'assert_eq!\(\s*"((?:[^"\\]|\\.)*)"\s*,\s*decompressed\)', t, re.S)
            ef = re.search(r'File::open\("(data/[^"]+)"\)\.unwrap\(\)\.read_to_(?:string|end)\(&mut expected\)', t)
            if em:
                entry["expected_hex"] = rust_str(em.group(1)).hex()
            elif ef:
                entry["expected_file"] = os.path.basename(ef.group(1))
            else:
                raise ValueError("no expectation in " + name)
        out.append(entry)
    with open(os.path.join(GOLD, "inline_vectors.json"), "w") as f:
        json.dump(out, f, indent=1)
    return out


def transform_vectors():
    src = open(os.path.join(REF, "src", "transformation", "mod.rs"), encoding="utf-8").read()
    out = []
    for m in re.finditer(r'fn should_transform_(\d+)\s*\(\)\s*\{\s*let base_word = String::from\("((?:[^"\\]|\\.)*)"\)'
                         r'(?:(?!#\[test\]).)*?let expected = "((?:[^"\\]|\\.)*)";', src, re.S):
        out.append({"id": int(m.group(1)), "word_hex": rust_str(m.group(2)).hex(),
                    "expected_hex": rust_str(m.group(3)).hex()})
    # id 102's test states its expectation as [0xc2, 0xa0] ++ base_word instead of a string literal
    m = re.search(r'fn should_transform_102\s*\(\)\s*\{\s*let base_word = String::from\("((?:[^"\\]|\\.)*)"\)'
                  r'.*?let expected = \[vec!\[0xc2, 0xa0\], base_word\.clone\(\)\]\.concat\(\);', src, re.S)
    w = rust_str(m.group(1))
    out.append({"id": 102, "word_hex": w.hex(), "expected_hex": (b"\xc2\xa0" + w).hex()})
    out.sort(key=lambda v: v["id"])
    assert [v["id"] for v in out] == list(range(121)), len(out)
    with open(os.path.join(GOLD, "transform_vectors.json"), "w") as f:
        json.dump(out, f, indent=1)
    return out


def main():
    os.makedirs(GOLD, exist_ok=True)
    m = data_fixtures()
    i = inline_vectors()
    t = transform_vectors()
    print("data streams: %d (valid %d, reject %d)" % (len(m), sum(e["status"] == 0 for e in m),
                                                     sum(e["status"] != 0 for e in m)))
    print("inline decode tests: %d; transform vectors: %d" % (len(i), len(t)))


if __name__ == "__main__":
    main()
