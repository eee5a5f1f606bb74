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


# ------------------------------------------------------------------------------------
# metadata, file names, text <-> symbols (recur_amd/csrc/charmodel_meta.c), pinned by
# values the reference's own headers produced (tests/golden/make_golden.py: rnn_hash32 of
# recur-common.h, utf8.h both ways) and the formats of charmodel-init.c:534-670.

Z = np.load(os.path.join(rc.ROOT, "tests", "golden", "ref_vectors.npz"))
LIBC = C.CDLL(None)
LIBC.free.argtypes = [C.c_void_p]


def take_string(ptr):
    s = C.string_at(ptr)
    LIBC.free(ptr)
    return s


def default_metadata():
    return rc.CharMetadata(rc.DEFAULT_CHARSET, rc.DEFAULT_COLLAPSE_CHARS, False, True, True)


DEFAULT_METADATA_STRING = (b"alphabet 8%20etaonihsrdlucmwfygpb,v.k-;x\"qj'?:z)(_!*&\n"
                           b"collapse_chars 10872}{659/34][@\nutf8 0\ncollapse_space 1\ncase_insensitive 1\n")


def test_metadata_string_round_trip_and_filename_hash():
    m = default_metadata()
    s = take_string(LIB.rnn_char_construct_metadata(C.byref(m)))
    assert s == DEFAULT_METADATA_STRING
    # the reference's hash of exactly this string (hash32_values[2])
    lens = Z["hash32_input_lens"]
    blob = Z["hash32_inputs"].tobytes()
    strings, pos = [], 0
    for n in lens:
        strings.append(blob[pos:pos + n])
        pos += n + 1
    assert strings[2] == s
    want = int(Z["hash32_values"][2])
    name = take_string(LIB.rnn_char_construct_net_filename(C.byref(m), b"text", 42, 0, 99, 42))
    assert name == b"text-s%x-i42-h99-o42.net" % want
    name = take_string(LIB.rnn_char_construct_net_filename(C.byref(m), b"text", 42, 16, 99, 42))
    assert name == b"text-s%x-i42-b16-h99-o42.net" % want
    # every other reference hash through the same function: an alphabet string chosen so that the
    # metadata text is known is not possible for arbitrary bytes, so restate the hash and pin
    # the restatement on all of the reference's values
    def hash32(b):
        sig = 0
        for t in b:
            x = (sig - t) & 0xffffffff
            sig ^= (((x << 13) | (x >> 19)) + t) & 0xffffffff
        return sig
    assert [hash32(x) for x in strings] == [int(v) for v in Z["hash32_values"]]
    # load: percent decoding, the five keys in order, flags
    m2 = rc.CharMetadata()
    assert LIB.rnn_char_load_metadata(s, C.byref(m2)) == 0
    assert m2.alphabet == rc.DEFAULT_CHARSET and m2.collapse_chars == rc.DEFAULT_COLLAPSE_CHARS
    assert (m2.utf8, m2.case_insensitive, m2.collapse_space) == (False, True, True)
    odd = rc.CharMetadata(bytes([1, 37, 200, 65, 127, 32]), b"%%", True, False, False)
    so = take_string(LIB.rnn_char_construct_metadata(C.byref(odd)))
    assert so == b"alphabet %01%25%c8A%7f%20\ncollapse_chars %25%25\nutf8 1\ncollapse_space 0\ncase_insensitive 0\n"
    m3 = rc.CharMetadata()
    assert LIB.rnn_char_load_metadata(so, C.byref(m3)) == 0
    assert m3.alphabet == bytes([1, 37, 200, 65, 127, 32]) and m3.collapse_chars == b"%%" and m3.utf8
    # wrong key order or a missing line is an error
    assert LIB.rnn_char_load_metadata(b"collapse_chars x\nalphabet y\n", C.byref(rc.CharMetadata())) == -1
    assert LIB.rnn_char_load_metadata(b"alphabet abc\ncollapse_chars d\nutf8 0\n", C.byref(rc.CharMetadata())) == -1


def test_check_metadata_policies():
    lib = rc.load_amd()
    net = lib.rnn_new(42, 8, 42, rc.FLAG_STANDARD, 1, None, 4, 1e-3, 0.9, 0.0, rc.RELU)
    m = default_metadata()
    assert LIB.rnn_char_check_metadata(None, C.byref(m), False, False) == -1
    assert LIB.rnn_char_check_metadata(net, C.byref(m), False, False) == 0        # no metadata in the net
    other = rc.CharMetadata(b"abc ", b"xyz", False, False, True)
    s_other = take_string(LIB.rnn_char_construct_metadata(C.byref(other)))
    strdup = LIBC.strdup
    strdup.restype, strdup.argtypes = C.c_void_p, [C.c_char_p]
    C.cast(C.byref(net.contents, rc.RecurNN.metadata.offset), C.POINTER(C.c_void_p))[0] = strdup(s_other)
    assert LIB.rnn_char_check_metadata(net, C.byref(other), False, False) == 0    # equal
    assert LIB.rnn_char_check_metadata(net, C.byref(m), False, False) == -2       # differs, no policy
    # an alphabet straight from the net's metadata
    a = LIB.rnn_char_new_alphabet_from_net(net)
    assert [a.contents.points[i] for i in range(a.contents.len)] == list(b"abc ")
    assert [a.contents.collapsed_points[i] for i in range(a.contents.collapsed_len)] == list(b"xyz")
    assert a.contents.flags == rc.CHAR_COLLAPSE_SPACE
    assert LIB.rnn_char_get_codepoint(a, b"c") == 2 and LIB.rnn_char_get_codepoint(a, b"q") == -1
    LIB.rnn_char_free_alphabet(a)
    trusting = default_metadata()
    # the caller's strings are replaced with heap copies (charmodel-init.c:636-650 frees the old ones)
    for f in ("alphabet", "collapse_chars"):
        C.cast(C.byref(trusting, getattr(rc.CharMetadata, f).offset), C.POINTER(C.c_void_p))[0] = \
            strdup(getattr(trusting, f))
    assert LIB.rnn_char_check_metadata(net, C.byref(trusting), True, False) == 0  # adopts the net's
    assert trusting.alphabet == b"abc " and trusting.collapse_chars == b"xyz" and not trusting.case_insensitive
    assert LIB.rnn_char_check_metadata(net, C.byref(m), False, True) == 0         # forces the caller's
    assert net.contents.metadata == DEFAULT_METADATA_STRING
    lib.rnn_delete_net(net)


def test_utf8_both_ways_match_reference_header():
    a = LIB.rnn_char_new_alphabet()
    LIB.rnn_char_alphabet_set_flags(a, False, True, False)
    codes, enc = Z["utf8_codes"], Z["utf8_encoded"]
    for code, row in zip(codes, enc):
        n, want = int(row[0]), bytes(int(b) & 0xff for b in row[1:1 + int(row[0])])
        a.contents.points[1] = int(code)
        a.contents.len = 2
        got_len = C.c_int(-1)
        sym = np.array([1], np.uint8)
        p = LIB.rnn_char_uncollapse_text(a, rc.u8ptr(sym), 1, C.byref(got_len))
        got = take_string(p)
        assert got == want and got_len.value == n, hex(int(code))   # n == 0: unrepresentable, nothing written
    # decoding: the symbol whose code point the reference's reader returns
    a.contents.points[0] = ord("z")  # a failed read leaves code point 0, which is then not found
    for seq, (want, used) in zip(Z["utf8_sequences"], Z["utf8_decoded"]):
        q = bytes(seq).rstrip(b"\0")
        a.contents.points[1] = int(want) if want > 0 else 0x7fffffff
        assert LIB.rnn_char_get_codepoint(a, q) == (1 if want > 0 else -1), q
    # bytes mode: latin-1 straight through, stops at a zero code point
    LIB.rnn_char_alphabet_set_flags(a, False, False, False)
    for i, c in enumerate(b"\x00xy\xe9"):
        a.contents.points[i] = c
    a.contents.len = 4
    got_len = C.c_int(0)
    sym = np.array([1, 3, 2, 0, 1], np.uint8)
    assert take_string(LIB.rnn_char_uncollapse_text(a, rc.u8ptr(sym), 5, C.byref(got_len))) == b"x\xe9y"
    assert got_len.value == 3
    LIB.rnn_char_free_alphabet(a)


def test_learn_rate_schedule_cuts_when_validation_stalls():
    """eval_simple (charmodel-predict.c:82-118): a third of the remembered scores must all be
    beaten, with a refractory period of recent_len calls; host logic only."""
    lib = rc.load_amd()
    net = lib.rnn_new(42, 8, 42, rc.FLAG_STANDARD, 3, None, 4, 1e-2, 0.9, 0.25, rc.RELU)
    model = rc.CharModel()
    model.net = net
    model.periodic_weight_noise = 0.5
    LIB.rnn_char_init_schedule(C.byref(model.schedule), 9, 1e-3, 0.5, 1)
    s = model.schedule
    assert s.recent_len == 9 and s.timeout == 9 and all(s.recent[i] == np.float32(1e10) for i in range(9))
    # the slot each call overwrites comes from the net's generator: mirror it
    orc = rc.load_oracle()
    g = rc.OrcRng()
    n = net.contents
    g.a, g.b, g.c, g.d = n.rng.a, n.rng.b, n.rng.c, n.rng.d
    recent = [1e10] * 9
    timeout, lr, noise, pwn = 9, np.float32(1e-2), np.float32(0.25), np.float32(0.5)
    scores = [3.0, 2.9, 2.8] * 3 + [2.7, 5.0, 5.0, 5.0, 5.0, 6.0, 6.0, 6.0, 6.0, 6.0, 6.0, 6.0, 7.0, 7.0,
                                    7.0, 7.0, 7.0, 7.0, 7.0, 7.0, 7.0, 7.0, 7.0, 7.0, 7.0, 7.0, 7.0, 7.0]
    cuts = 0
    for score in scores:
        s.eval(C.byref(model), score, 0)
        if lr > np.float32(1e-3):
            i = orc.orc_rand_small_int(C.byref(g), 9)
            recent[i] = score
            if timeout:
                timeout -= 1
            else:
                i += 1
                beaten = False
                for _ in range(3):
                    if i >= 9:
                        i = 0
                    if score < recent[i]:
                        beaten = True
                        break
                    i += 1
                if not beaten:
                    timeout = 9
                    lr = max(np.float32(1e-3), np.float32(lr * np.float32(0.5)))
                    noise = np.float32(noise * np.float32(0.5))
                    pwn = np.float32(pwn * np.float32(0.5))
                    cuts += 1
        assert n.bptt.contents.learn_rate == lr and s.timeout == timeout
        assert n.presynaptic_noise == noise and model.periodic_weight_noise == pwn
        assert [s.recent[k] for k in range(9)] == [np.float32(x) for x in recent]
    assert cuts >= 2
    assert (n.rng.a, n.rng.d) == (g.a, g.d)
    lib.rnn_delete_net(net)
