#!/usr/bin/env python3
"""Build-container script (never shipped to the GPU box, reads /root/reference): a net file WRITTEN by librecur_amd's
rnn_save_net (recur_amd/csrc/rnn_io.c + cdb.c) is opened with the REFERENCE's own CDB reader -- scripts/pycdb.py, the
module its tooling reads .net files with -- and every record is looked up THROUGH THE HASH TABLES (pycdb.py:107-139,
`Reader.gets`, which `Reader.get` wraps: the same two-level walk tinycdb's cdb_seek makes for rnn_load_net,
recur-nn-io.c:168-184), not by a sequential walk over the records.

pycdb.py is Python 2 (`xrange`, `.next()`, `ord()` over a str): it is imported as it lies, with `xrange` provided and its
own `py_djb_hash` fed the key as a latin-1 str; `Reader.get` itself ends in `.next()`, so the generator it wraps is used.

Output: tests/golden/cdb_written_by_rnn_save_net.json -- for each of two files the sha256 of the file as written and,
per record, the key, the value's length and sha1 AS THE REFERENCE'S READER FOUND IT.  tests/test_abi.py writes the same
files again (on any box) and holds them to these bytes: what the library writes is what the reference's reader read.
"""
import builtins
import hashlib
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
builtins.xrange = range
sys.path.insert(0, "/root/reference/scripts")
import pycdb  # noqa: E402  (the reference's reader, imported where it lies)

import cdb_cases  # noqa: E402


def read_with_the_references_reader(path):
    data = open(path, "rb").read()
    r = pycdb.Reader(data, hashfn=lambda k: pycdb.py_djb_hash(k.decode("latin-1")))
    records = []
    for key, value in r.iteritems():  # the records in file order ...
        found = list(r.gets(key))     # ... each one looked up through the hash tables
        assert found and found[0] == value, "hash lookup of %r does not find the record" % key
        assert len(found) == 1, "%r is there %d times" % (key, len(found))
        records.append({"key": key.decode(), "len": len(value), "sha1": hashlib.sha1(value).hexdigest()})
    assert len(r) == len(records), (len(r), len(records))  # slots / 2 over the 256 tables == records
    assert next(r.gets(b"no such key"), None) is None
    return {"sha256": hashlib.sha256(data).hexdigest(), "bytes": len(data), "records": records}


def main():
    out = {}
    with tempfile.TemporaryDirectory() as d:
        for name in cdb_cases.CASES:
            path = cdb_cases.write_case(name, d)
            out[name] = read_with_the_references_reader(path)
            print("%s: %d bytes, %d records, every key found by hash lookup" % (name, out[name]["bytes"], len(out[name]["records"])))
    with open(os.path.join(HERE, "cdb_written_by_rnn_save_net.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
