"""Host-side logic of the sharded (multi-GPU) training set, no GPU needed: the clone
generators of a shard are those the reference gives the same global streams
(recur-nn-init.c:232-241, 300-305), whatever the number of ranks."""
import ctypes as C

import numpy as np
import pytest

import recur_ctypes as rc
import replay

AMD = rc.load_amd()


def _rngs(nets, n):
    return [(nets[j].contents.rng.a, nets[j].contents.rng.b, nets[j].contents.rng.c, nets[j].contents.rng.d)
            for j in range(n)]


def _proto(seed, noise=0.0):
    return AMD.rnn_new(42, 39, 42, rc.FLAG_STANDARD | rc.FLAG_ADAPTIVE_MIN_ERROR, seed, None, 8, 1e-2, 0.9,
                       noise, rc.RELU)


@pytest.mark.parametrize("world,total", [(2, 6), (3, 12), (4, 4), (1, 5)])
def test_shard_generators_equal_the_single_set(world, total):
    full = AMD.rnn_new_training_set(_proto(3), total)
    want = _rngs(full, total)
    per = total // world
    got = []
    for r in range(world):
        nets = AMD.rnn_amd_new_training_set_shard(_proto(3), per, r * per, total)
        got += _rngs(nets, per)
        n0, n1 = nets[0].contents, nets[per - 1].contents
        if per > 1:  # clones share the prototype's weights and deltas like rnn_new_training_set's
            addr = lambda p: C.addressof(p.contents)
            assert addr(n0.ih_weights) == addr(n1.ih_weights)
            assert addr(n0.bptt.contents.ih_delta) == addr(n1.bptt.contents.ih_delta)
            assert addr(n0.bptt.contents.history) != addr(n1.bptt.contents.history)
        AMD.rnn_delete_training_set(nets, per, 0)
    assert got == want
    AMD.rnn_delete_training_set(full, total, 0)


def test_shard_generators_equal_the_reference_golden():
    """relu_weighted's generator snapshot comes from the REAL reference (4 streams, seed 3,
    no noise: the states are those right after rnn_new_training_set)."""
    c = replay.golden_case("relu_weighted")
    got = []
    for r in range(2):
        net = _proto(3)
        p = rc.InitParams()  # the scenario initialises the weights first (scenarios.ApiSet)
        AMD.rnn_init_default_weight_parameters(net, C.byref(p))
        p.method, p.flat_shape, p.flat_perforation = rc.INIT_FLAT, rc.DIST_SEMICIRCLE, 0.0
        AMD.rnn_randomise_weights_clever(net, C.byref(p))
        nets = AMD.rnn_amd_new_training_set_shard(net, 2, 2 * r, 4)
        got += _rngs(nets, 2)
    assert got == [tuple(int(x) for x in c["rng"][j]) for j in range(4)]


def test_bad_shards_are_refused():
    p = _proto(5)
    assert not AMD.rnn_amd_new_training_set_shard(p, 0, 0, 4)
    assert not AMD.rnn_amd_new_training_set_shard(p, 3, 2, 4)
    assert not AMD.rnn_amd_new_training_set_shard(p, 2, -1, 4)
    assert AMD.rnn_amd_dist_world() == 1 and AMD.rnn_amd_dist_rank() == 0  # no group joined


def test_bench_gpus_n_spawns_n_ranks_and_propagates_failure():
    """`bench.py --gpus 2` without a launcher starts two ranks itself (before touching a
    GPU).  Here there is no GPU, so both ranks must fail loudly and the launcher must
    return non-zero instead of hanging or printing a metric line."""
    import os
    import subprocess
    import sys
    if AMD.rnn_amd_device_count() > 0:
        pytest.skip("a GPU is present: the real run is the driver's")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(rc.ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, env=env,
                       timeout=300)
    assert r.returncode != 0
    assert r.stderr.count("needs a GPU") == 2 and "ranks failed" in r.stderr
    assert "stream-timesteps" not in r.stdout


# ---- bench.py's launcher side (round 6): the parent's watchdog, the ranks' host-side meeting place

def test_bench_parent_ends_a_run_that_hangs():
    """bench.py --gpus N without a launcher starts the ranks itself; ranks that never finish (stuck in a collective, a
    barrier nobody else reaches) are ended by the parent -- the exact children it started -- after --rank-timeout
    seconds, exit status 124.  RECUR_BENCH_TEST_HANG=all makes every rank sleep before it touches anything."""
    import os
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rank-timeout", "2"],
                       env=dict(os.environ, RECUR_BENCH_TEST_HANG="all"), capture_output=True, text=True, timeout=120)
    assert r.returncode == 124, (r.returncode, r.stderr[-1000:])
    assert "ending them" in r.stderr and time.time() - t0 < 60
    assert r.stdout.strip() == ""  # no JSON line from a run that did not finish


GROUP_SCRIPT = r"""
import os, sys, json
sys.path.insert(0, %(root)r)
import bench
rank, world = int(sys.argv[1]), int(sys.argv[2])
g = bench.HostGroup(rank, world, timeout_s=60.0)
got = []
for k in range(50):
    got.append(g.gather(1000 * k + rank))
d = g.gather(0.5 + rank, "<d")
g.barrier()
g.close()
print("RESULT " + json.dumps({"got": got, "d": d}))
"""


def test_bench_host_group_gathers_in_rank_order():
    """bench.HostGroup: the ranks of one node meet in a file in /dev/shm -- a barrier and an all-gather of 8-byte values on
    the HOST, so that what the JSON line says about the ranks (checksums, times, which exchange) does not depend on the
    collective library it checks.  Three processes, fifty gathers: everybody sees everybody's value of that round."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_PORT="0", RECUR_BENCH_RUN_ID="hostgroup_test_%d" % os.getpid())
    procs = [subprocess.Popen([sys.executable, "-c", GROUP_SCRIPT % {"root": root}, str(r), "3"], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(3)]
    outs = [p.communicate(timeout=120) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0 and "RESULT " in so, so[-500:] + se[-1500:]
        res = json.loads(so.split("RESULT ", 1)[1])
        assert res["got"] == [[1000 * k + r for r in range(3)] for k in range(50)]
        assert res["d"] == [0.5, 1.5, 2.5]
    assert not os.path.exists("/dev/shm/recur_amd_bench_0_hostgroup_test_%d" % os.getpid())
