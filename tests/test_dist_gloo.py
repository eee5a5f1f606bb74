"""The N > 1 path on CPU: two gloo ranks, each holding half of the streams of one
training set (computed by the oracle), one all-reduce of ih_delta||ho_delta per
generation, replicated update -- must equal the single-process run over all the
streams.  On the GPUs the same three phases run inside librecur_amd (rnn_amd_set_char_step with
a group joined: deltas, RCCL all-reduce, update); here the arithmetic is the oracle's, the
sharding and the clone seeding are the product's host code."""
import ctypes as C
import os
import socket
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _worker(rank, world, port, S_local, steps, q):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import ctypes as C
    import torch
    import torch.distributed as dist
    import recur_ctypes as rc
    import scenarios as sc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    text = sc.synthetic_text(4000)
    first, n, total = rank * S_local, S_local, world * S_local  # contiguous blocks of streams per rank
    o = sc.OracleSet(input_size=42, hidden_size=45, output_size=42, S=n, D=7, learn_rate=5e-3, seed=2,
                     noise=0.03)
    o.z.contents.global_first, o.z.contents.global_count = first, total
    # the shard's generators come from the PRODUCT's host logic (no GPU needed for it):
    # rnn_amd_new_training_set_shard replays the reference's clone seeding for the global set
    amd = rc.load_amd()
    proto = amd.rnn_new(42, 45, 42, rc.FLAG_STANDARD | rc.FLAG_ADAPTIVE_MIN_ERROR, 2, None, 7, 5e-3, 0.9, 0.03,
                        rc.RELU)
    p = rc.InitParams()
    amd.rnn_init_default_weight_parameters(proto, C.byref(p))
    p.method, p.flat_shape, p.flat_perforation = rc.INIT_FLAT, rc.DIST_SEMICIRCLE, 0.0
    amd.rnn_randomise_weights_clever(proto, C.byref(p))
    nets = amd.rnn_amd_new_training_set_shard(proto, n, first, total)
    for j in range(n):
        r = nets[j].contents.rng
        o.z.contents.rng[j].a, o.z.contents.rng[j].b, o.z.contents.rng[j].c, o.z.contents.rng[j].d = r.a, r.b, r.c, r.d
    a = o.arrays()
    ih, ho = a["ih_delta"], a["ho_delta"]

    def reduce():
        flat = torch.from_numpy(np.concatenate([ih.reshape(-1), ho.reshape(-1)]))
        dist.all_reduce(flat)
        ih.reshape(-1)[:] = flat[:ih.size].numpy()
        ho.reshape(-1)[:] = flat[ih.size:].numpy()

    def step(i):  # one generation = local deltas -> all-reduce(sum) -> replicated update
        o.char_step_deltas(text, i)
        if world > 1:
            reduce()
        o.orc.orc_apply_learning(o.z, rc.WEIGHTED, 0.9)

    for i in range(steps):
        step(i)
    q.put((rank, a["ih_w"].copy(), a["ho_w"].copy(), a["ih_m"].copy(), a["hidden"].copy(), o.snapshot()["rng"]))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(300)
def test_two_rank_sharding_matches_single_process():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    S_local, steps = 3, 12
    results = {}
    for world in (1, 2):
        q = ctx.Queue()
        port = _free_port()
        streams = S_local * 2 // world  # same global set of 6 streams both times
        procs = [ctx.Process(target=_worker, args=(r, world, port, streams, steps, q)) for r in range(world)]
        for p in procs:
            p.start()
        got = [q.get(timeout=240) for _ in range(world)]
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        results[world] = sorted(got, key=lambda t: t[0])
    single = results[1][0]
    r0, r1 = results[2]
    # replicas stay identical (gloo's sum is the same on both ranks) ...
    for k in (1, 2, 3):
        assert np.array_equal(r0[k], r1[k])
    # ... and equal the single-process run up to the order of the float sum
    for k in (1, 2, 3):
        err = np.linalg.norm(r0[k].astype(np.float64) - single[k]) / np.linalg.norm(single[k])
        assert err < 1e-5, err
    hidden = np.concatenate([r0[4], r1[4]])
    assert np.linalg.norm(hidden - single[4]) / np.linalg.norm(single[4]) < 1e-5
    # every global stream drew its noise from the same generator wherever it lived
    assert np.array_equal(np.concatenate([r0[5], r1[5]]), single[5])
