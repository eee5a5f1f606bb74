"""The strings around an audio classifier net and the balanced-training rule (recur_amd/csrc/classify_host.c,
include/recur_amd_classify.h; gstclassify.c:689-748, 841-929, 1141-1150, 2190-2215).  Host only."""
import ctypes as C

import numpy as np

import recur_ctypes as rc

LIB = rc.bind_classify(rc.load_amd())
LIBC = C.CDLL(None)
LIBC.free.argtypes = [C.c_void_p]


def take_string(p):
    s = C.string_at(p)
    LIBC.free(p)
    return s


def defaults():
    # the plugin's defaults (gstclassify.h / gstclassify.c property table): classes "01", 100 .. 1600 Hz ...
    return rc.ClassifyMetadata(b"01", 100.0, 1600.0, 700.0, 0, 256, b"classify", 0, 600.0, 0.0, 0, 0.0, None, None)


def test_metadata_string_is_the_references_fourteen_lines():
    m = defaults()
    got = take_string(LIB.rnn_amd_classify_construct_metadata(C.byref(m)))
    want = (b"classes 01\nmin-frequency 100.000000\nmax-frequency 1600.000000\nknee-frequency 700.000000\nmfccs 0\n"
            b"window-size 256\nbasename classify\ndelta-features 0\nfocus-frequency 600.000000\nlag 0.000000\n"
            b"intensity-feature 0\nconfirmation-lag 0.000000\nfeatures-offset (null)\nfeatures-scale (null)\n")
    assert got == want
    m.features_offset, m.features_scale, m.lag, m.intensity_feature = b"0.5:-1", b"2:2", 0.064, 1
    got = take_string(LIB.rnn_amd_classify_construct_metadata(C.byref(m)))
    assert b"lag 0.064000\nintensity-feature 1\n" in got and got.endswith(b"features-offset 0.5:-1\nfeatures-scale 2:2\n")


def test_metadata_loads_back_as_far_as_the_reference_reads_it():
    m = defaults()
    m.classes, m.basename, m.mfccs, m.window_size, m.delta_features = b"MFn", b"birds", 20, 512, 2
    m.min_freq, m.lag, m.intensity_feature, m.confirmation_lag = 220.5, 0.128, 1, 0.75
    s = take_string(LIB.rnn_amd_classify_construct_metadata(C.byref(m)))
    back = rc.ClassifyMetadata()
    back.confirmation_lag = -3.0  # the reference's template never reaches this line: the value stays
    assert LIB.rnn_amd_classify_load_metadata(s, C.byref(back)) == 0
    assert (back.classes, back.basename, back.mfccs, back.window_size, back.delta_features) == (b"MFn", b"birds", 20, 512, 2)
    assert abs(back.min_freq - 220.5) < 1e-6 and abs(back.lag - 0.128) < 1e-6 and back.intensity_feature == 1
    assert back.confirmation_lag == -3.0 and back.features_offset is None
    LIB.rnn_amd_classify_free_metadata_items(C.byref(back))
    # an older net: fewer items.  The count of missing ones comes back, what was there is set
    short = b"classes 01\nmin-frequency 100.000000\nmax-frequency 1600.000000\nknee-frequency 700.000000\nmfccs 0\n"
    old = rc.ClassifyMetadata()
    assert LIB.rnn_amd_classify_load_metadata(short, C.byref(old)) == 6
    assert old.classes == b"01" and old.max_freq == 1600.0
    LIB.rnn_amd_classify_free_metadata_items(C.byref(old))
    assert LIB.rnn_amd_classify_load_metadata(None, C.byref(old)) == -1


def test_net_file_name_embeds_the_metadata_hash():
    m = defaults()
    meta = take_string(LIB.rnn_amd_classify_construct_metadata(C.byref(m)))
    # recur-common.h:207-216, restated independently here
    sig = 0
    for t in meta:
        x = (sig - t) & 0xffffffff
        sig ^= ((((x << 13) | (x >> 19)) & 0xffffffff) + t) & 0xffffffff
    name = take_string(LIB.rnn_amd_classify_net_filename(b"classify", meta, 32, 0, 512, 2, 8000, 256))
    assert name == b"classify-%x-i32-h512-o2-8000Hz-w256.net" % sig
    name = take_string(LIB.rnn_amd_classify_net_filename(b"classify", meta, 40, 24, 199, 5, 16000, 512))
    assert name == b"classify-%x-i40-b24-h199-o5-16000Hz-w512.net" % sig


def test_class_group_strings():
    off, size = (C.c_int * 8)(), (C.c_int * 8)()
    n_out, slen = C.c_int(0), C.c_int(0)
    assert LIB.rnn_amd_classify_parse_classes(b"01", off, size, 8, C.byref(n_out), C.byref(slen)) == 1
    assert (off[0], size[0], n_out.value, slen.value) == (0, 2, 2, 2)
    # several groups: the offsets are positions in the STRING (what the reference stores), the outputs the letters
    assert LIB.rnn_amd_classify_parse_classes(b"01,abc,Z", off, size, 8, C.byref(n_out), C.byref(slen)) == 3
    assert list(off[:3]) == [0, 3, 7] and list(size[:3]) == [2, 3, 1] and n_out.value == 6 and slen.value == 8
    assert LIB.rnn_amd_classify_parse_classes(b"ab,,c", off, size, 2, None, None) == 3  # counted, two written
    assert list(size[:2]) == [2, 0]


def test_loaded_net_is_checked_against_the_configuration():
    net = LIB.rnn_new(32, 64, 2, rc.FLAG_STANDARD, 1, None, 10, 1e-3, 0.9, 0.0, rc.RELU)
    m = defaults()
    meta = take_string(LIB.rnn_amd_classify_construct_metadata(C.byref(m)))
    assert LIB.rnn_amd_classify_check_net(net, meta, 64, 0, 2, 0) == 0      # no metadata in the net yet
    keep = C.create_string_buffer(meta)
    net.contents.metadata = C.cast(keep, C.c_char_p)
    assert LIB.rnn_amd_classify_check_net(net, meta, 64, 0, 2, 0) == 0
    assert LIB.rnn_amd_classify_check_net(net, meta, 99, 0, 2, 0) == -1     # another hidden size
    assert LIB.rnn_amd_classify_check_net(net, meta, 64, 0, 3, 0) == -1     # another class count
    other = meta.replace(b"window-size 256", b"window-size 512")
    assert LIB.rnn_amd_classify_check_net(net, other, 64, 0, 2, 0) == -1    # other audio settings
    assert LIB.rnn_amd_classify_check_net(net, other, 64, 0, 2, 1) == 0     # ... unless forced
    net.contents.metadata = None
    LIB.rnn_delete_net(net)


def test_balanced_training_probabilities():
    b = LIB.rnn_amd_balanced_new(3, 2.0)
    LIB.rnn_amd_balanced_begin(b)
    assert list(b.contents.train_p[:3]) == [1.0, 1.0, 1.0]  # nothing seen yet: every labelled window trains
    for c, n in enumerate((90, 9, 0)):
        b.contents.seen[c] = n
    LIB.rnn_amd_balanced_begin(b)
    share = np.float32(1.0) / np.float32(100.0)
    want = [float(np.float32(np.float32(1.0) - np.float32(n) * share) ** 2) for n in (90, 9, 0)]
    assert np.allclose(list(b.contents.train_p[:3]), want, rtol=1e-6)
    LIB.rnn_amd_balanced_free(b)
