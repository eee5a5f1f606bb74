"""Net files written by librecur_amd's rnn_save_net for the container checks (tests/golden/make_cdb_reader_check.py in the
build container, tests/test_abi.py anywhere): host-only calls, seeded, no GPU."""
import os

import recur_ctypes as rc

CASES = ("plain_relu_37", "bottom_layer_resqrt_metadata")


def write_case(name, directory):
    amd = rc.load_amd()
    if name == "plain_relu_37":
        net = amd.rnn_new(11, 37, 5, rc.FLAG_STANDARD | rc.FLAG_ADAPTIVE_MIN_ERROR, 77, None, 9, 3e-3, 0.93, 0.0, rc.RELU)
        amd.rnn_randomise_weights_auto(net)
    elif name == "bottom_layer_resqrt_metadata":
        net = amd.rnn_new_with_bottom_layer(19, 8, 64, 12, rc.FLAG_STANDARD, 5, None, 6, 1e-2, 0.9, 0.05, rc.RESQRT, 0)
        amd.rnn_randomise_weights_auto(net)
        net.contents.metadata = b'{"alphabet": "abc", "note": "written by librecur_amd"}'
        net.contents.generation = 4242
    else:
        raise KeyError(name)
    path = os.path.join(directory, name + ".net")
    cwd = os.getcwd()
    os.chdir(directory)  # rnn_save_net makes its temp file in the cwd (recur-nn-io.c:17-21)
    try:
        assert amd.rnn_save_net(net, path.encode(), 0) == 0
    finally:
        os.chdir(cwd)
    return path
