"""Alphabet discovery, the char -> symbol table and text encoding
(recur_amd/csrc/charmodel.c; host-only integer code, bit-exact bar).
Pinned by the expected alphabets of the reference's own test table
(test/test_charmodel_alphabet.c, extracted as data into
tests/golden/alphabet_cases.json) over the reference's erewhon.txt."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import recur_ctypes as rc

LIB = rc.bind_char(rc.load_amd())
CASES = json.load(open(os.path.join(rc.ROOT, "tests", "golden", "alphabet_cases.json")))
TEXT = open(rc.EREWHON, "rb").read()


def codepoints(s, utf8):
    return sorted(ord(ch) for ch in s) if utf8 else sorted(s.encode("latin-1"))


@pytest.mark.parametrize("case", CASES, ids=lambda c: "thr%g_case%d_utf%d_sp%d_dig%g" % (
    c["threshold"], c["ignore_case"], c["utf8"], c["collapse_space"], c["digit_adjust"]))
def test_alphabet_matches_reference_table(case):
    a = LIB.rnn_char_new_alphabet()
    LIB.rnn_char_alphabet_set_flags(a, bool(case["ignore_case"]), bool(case["utf8"]),
                                    bool(case["collapse_space"]))
    err = LIB.rnn_char_find_alphabet_s(TEXT, len(TEXT), a, case["threshold"], case["digit_adjust"],
                                       case["alpha_adjust"])
    assert err == 0
    al = a.contents
    got = sorted(al.points[i] for i in range(al.len))
    got_c = sorted(al.collapsed_points[i] for i in range(al.collapsed_len))
    assert got == codepoints(case["alphabet"], case["utf8"])
    assert got_c == codepoints(case["collapse"], case["utf8"])
    if case["collapse"]:
        # the representative of the collapsed characters leads the alphabet; it is the
        # one alphabet member that is out of code point order (charmodel-init.c:115-120)
        rest = [al.points[i] for i in range(1, al.len)]
        assert rest == sorted(rest)
    LIB.rnn_char_free_alphabet(a)


def test_file_and_string_entry_points_agree_and_errors():
    a = LIB.rnn_char_new_alphabet()
    b = LIB.rnn_char_new_alphabet()
    LIB.rnn_char_alphabet_set_flags(a, True, False, True)
    LIB.rnn_char_alphabet_set_flags(b, True, False, True)
    assert LIB.rnn_char_find_alphabet_f(rc.EREWHON.encode(), a, 1e-4, 1.0, 1.0) == 0
    assert LIB.rnn_char_find_alphabet_s(TEXT, len(TEXT), b, 1e-4, 1.0, 1.0) == 0
    assert [a.contents.points[i] for i in range(a.contents.len)] == \
           [b.contents.points[i] for i in range(b.contents.len)]
    assert LIB.rnn_char_find_alphabet_f(b"/no/such/file", a, 1e-4, 1.0, 1.0) == -1
    assert LIB.rnn_char_find_alphabet_s(b"", 0, a, 1e-4, 1.0, 1.0) == -1 and a.contents.len == 0
    LIB.rnn_char_free_alphabet(a)
    LIB.rnn_char_free_alphabet(b)


def python_encode(text, charset, collapse_space=True):
    """independent restatement of the LUT + encoder (charmodel-init.c:238-329) for
    text-predict's default alphabet"""
    space = charset.index(b" ")
    lut = [space] * 257
    for i, c in enumerate(charset):
        lut[c] = i
        if chr(c).islower():
            lut[ord(chr(c).upper())] = i
    out = []
    prev = space
    for ch in text:
        if ch == 0:
            break
        c = lut[ch if ch < 128 else ch - 256]   # text[i] is a signed char in the reference
        if not collapse_space or c != space or prev != space:
            prev = c
            out.append(c)
    return np.array(out, np.uint8)


def test_default_alphabet_lut_and_encoding():
    a = rc.default_text_alphabet(LIB)
    lut = LIB.rnn_char_new_char_lut(a)
    table = [lut[i] for i in range(257)]
    space = rc.DEFAULT_CHARSET.index(b" ")
    assert table[ord("e")] == table[ord("E")] == rc.DEFAULT_CHARSET.index(b"e")
    assert table[ord("\n")] == space and table[ord("1")] == space     # quirk 9: collapse chars -> space
    assert table[ord("8")] == 0
    enc = rc.encode_erewhon(LIB)
    want = python_encode(TEXT, rc.DEFAULT_CHARSET)
    assert len(enc) == len(want) and np.array_equal(enc, want)
    assert enc.max() < len(rc.DEFAULT_CHARSET) and len(enc) < len(TEXT)
    # no two consecutive spaces survive
    sp = enc == space
    assert not (sp[1:] & sp[:-1]).any()
    LIB.rnn_char_free_alphabet(a)


def test_utf8_decoding_path():
    s = "nā te wai ā ō".encode("utf-8") * 50
    a = LIB.rnn_char_new_alphabet()
    LIB.rnn_char_alphabet_set_flags(a, True, True, True)
    assert LIB.rnn_char_find_alphabet_s(s, len(s), a, 1e-3, 1.0, 1.0) == 0
    pts = [a.contents.points[i] for i in range(a.contents.len)]
    assert 0x101 in pts and 0x14d in pts and 0x304 in pts
    n = C.c_int(0)
    enc = LIB.rnn_char_alloc_encoded_text(a, s, len(s), C.byref(n), None, False)
    assert n.value == len(s.decode("utf-8")) - 0  # one symbol per code point (no double spaces here)
    assert max(enc[i] for i in range(n.value)) < a.contents.len
    LIB.rnn_char_free_alphabet(a)
