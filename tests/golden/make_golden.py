#!/usr/bin/env python3
"""Regenerates the golden fixtures under tests/golden/ from the read-only reference tree.

Run in the authoring container only (`/root/reference` does not exist on the GPU box):

    python tests/golden/make_golden.py

What it extracts is DATA, never source text:
  * constants.json   -- numeric constant arrays the reference's tests use as golden vectors
                        (src/tables.rs: FIXED_LITLEN_TABLE, FIXED_DIST_TABLE, HUFFMAN_LENGTHS,
                        LENGTH_TO_SYMBOL, LENGTH_TO_LEN_EXTRA; src/compress/ultrafast.rs HEADER)
  * vectors/*.zz     -- the three regression inputs of src/decompress.rs:1344-1384
  * vectors/corpus/* -- the 66 fuzz/corpus/inflate inputs (CI replays them, rust.yml:81-85)
  * manifest.json    -- expected (length, adler32, status) per vector.  Expected outputs of
                        the corpus come from Python's zlib (an independent inflate, the role
                        miniz_oxide plays in fuzz_targets/inflate.rs); the .zz expectations are
                        the literal asserts of the reference tests.
"""
import json
import os
import re
import shutil
import zlib

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def rust_array(text, name):
    m = re.search(r"const\s+" + name + r"\s*:\s*\[[^\]]+\]\s*=\s*\[(.*?)\];", text, re.S)
    if not m:
        raise SystemExit("array %s not found" % name)
    return [int(x) for x in re.findall(r"\d+", m.group(1))]


def main():
    tables = open(os.path.join(REF, "src/tables.rs")).read()
    ultrafast = open(os.path.join(REF, "src/compress/ultrafast.rs")).read()
    constants = {
        "source": "image-rs/fdeflate src/tables.rs, src/compress/ultrafast.rs (numeric data only)",
        "FIXED_LITLEN_TABLE": rust_array(tables, "FIXED_LITLEN_TABLE"),
        "FIXED_DIST_TABLE": rust_array(tables, "FIXED_DIST_TABLE"),
        "HUFFMAN_LENGTHS": rust_array(tables, "HUFFMAN_LENGTHS"),
        "LENGTH_TO_SYMBOL": rust_array(tables, "LENGTH_TO_SYMBOL"),
        "LENGTH_TO_LEN_EXTRA": rust_array(tables, "LENGTH_TO_LEN_EXTRA"),
        "ULTRAFAST_HEADER": rust_array(ultrafast, "HEADER"),
    }
    assert len(constants["FIXED_LITLEN_TABLE"]) == 512
    assert len(constants["FIXED_DIST_TABLE"]) == 32
    assert len(constants["HUFFMAN_LENGTHS"]) == 286
    assert len(constants["LENGTH_TO_SYMBOL"]) == 256
    assert len(constants["LENGTH_TO_LEN_EXTRA"]) == 256
    assert len(constants["ULTRAFAST_HEADER"]) == 54
    with open(os.path.join(HERE, "constants.json"), "w") as f:
        json.dump(constants, f)

    vec_dir = os.path.join(HERE, "vectors")
    corpus_dir = os.path.join(vec_dir, "corpus")
    os.makedirs(corpus_dir, exist_ok=True)

    manifest = {"zz": {}, "corpus": {}}
    # src/decompress.rs:1344-1384 -- literal asserts of the reference tests
    expect_zz = {
        "input-chunking-sensitivity-example1.zz": {
            "status_ignore_adler": 0, "length": 281, "adler32": 751299,
            "ref": "src/decompress.rs:1345-1352"},
        "input-chunking-sensitivity-example2.zz": {
            "status_ignore_adler": 9, "ref": "src/decompress.rs:1359-1368 BadLiteralLengthHuffmanTree"},
        "input-chunking-sensitivity-example3.zz": {
            "status_ignore_adler": 9, "ref": "src/decompress.rs:1375-1384 BadLiteralLengthHuffmanTree"},
    }
    for name, exp in expect_zz.items():
        shutil.copyfile(os.path.join(REF, "tests", name), os.path.join(vec_dir, name))
        manifest["zz"][name] = exp

    src_corpus = os.path.join(REF, "fuzz/corpus/inflate")
    for name in sorted(os.listdir(src_corpus)):
        data = open(os.path.join(src_corpus, name), "rb").read()
        shutil.copyfile(os.path.join(src_corpus, name), os.path.join(corpus_dir, name))
        entry = {"in_len": len(data)}
        try:
            out = zlib.decompress(data)
            entry.update(zlib_ok=True, length=len(out), adler32=zlib.adler32(out))
        except zlib.error as e:  # pragma: no cover - all 66 are valid today
            entry.update(zlib_ok=False, error=str(e))
        manifest["corpus"][name] = entry
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print("constants.json, manifest.json, %d corpus vectors, %d zz vectors written"
          % (len(manifest["corpus"]), len(manifest["zz"])))


if __name__ == "__main__":
    main()
