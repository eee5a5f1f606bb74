"""The sharded (multi-GPU) product path on ONE GPU.

* Two ranks emulated one after the other in one process: two training-set shards
  (rnn_amd_new_training_set_shard + rnn_amd_set_shard), each computing its deltas into its
  own external device buffer (rnn_amd_set_external_delta), the two buffers summed by hand
  (what the all-reduce does), rnn_apply_learning on each replica -- against the single set
  of all the streams on the device AND against the oracle, generator states bit-exact
  (so the clone seeding by global stream number, recur-nn-init.c:300-305, has to be right,
  also with presynaptic noise).
* The library's own exchange step through RCCL with one rank (a subprocess: a joined group
  is process-wide state): rnn_amd_set_char_step with a group joined == without.

The sum being distributed is recur-nn.c:724-739 over the sharing of recur-nn-init.c:232-241.
"""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import recur_ctypes as rc
import replay
import scenarios as sc

pytestmark = pytest.mark.gpu
RTOL = 1e-4


@pytest.fixture(scope="module")
def amd():
    lib = rc.load_amd()
    assert lib.rnn_amd_device_count() >= 1, "no HIP device: the product has no CPU fallback"
    return lib


@pytest.fixture(scope="module")
def hip():
    lib = C.CDLL("libamdhip64.so")
    lib.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    lib.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    lib.hipFree.argtypes = [C.c_void_p]
    return lib


D2H, H2D = 2, 1


@pytest.mark.parametrize("label,kw,noise", [
    # the LDS-DMA delta kernel + the fused forward / top kernels (32 streams per rank, hidden 128)
    ("dma_path", dict(input_size=42, hidden_size=128, output_size=42, D=6, learn_rate=2e-3, seed=21), 0.0),
    # presynaptic noise: every global stream must draw from the reference's generator for it
    ("noisy", dict(input_size=42, hidden_size=64, output_size=42, D=5, learn_rate=2e-3, seed=22), 0.05),
    # ragged: the generic GEMM kernels
    ("ragged", dict(input_size=42, hidden_size=45, output_size=42, D=7, learn_rate=5e-3, seed=23), 0.0),
])
def test_two_emulated_ranks_equal_the_single_set_and_the_oracle(amd, hip, label, kw, noise):
    S = 32 if label != "ragged" else 3
    steps = 12
    text = sc.synthetic_text(6000)
    ranks = [sc.AmdBatchedSet(amd, S=S, noise=noise, shard=(r * S, 2 * S), **kw) for r in range(2)]
    n_delta = ranks[0].I * ranks[0].H + ranks[0].H * ranks[0].O
    bufs = []
    for r, g in enumerate(ranks):
        g.load_text(text)
        amd.rnn_amd_set_shard(g.handle, r * S, 2 * S)
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), 4 * n_delta) == 0
        amd.rnn_amd_set_external_delta(g.handle, p)
        bufs.append(p)
    single = sc.AmdBatchedSet(amd, S=2 * S, noise=noise, **kw)
    single.load_text(text)
    o = sc.OracleSet(S=2 * S, noise=noise, **kw)
    h = [np.zeros(n_delta, np.float32) for _ in range(2)]
    for i in range(steps):
        for g in ranks:
            amd.rnn_amd_set_char_step_deltas(g.handle, i)
        amd.rnn_amd_synchronize()
        for r in range(2):
            assert hip.hipMemcpy(h[r].ctypes.data, bufs[r], 4 * n_delta, D2H) == 0
        total = h[0] + h[1]  # what the all-reduce leaves on every rank
        for r, g in enumerate(ranks):
            assert hip.hipMemcpy(bufs[r], total.ctypes.data, 4 * n_delta, H2D) == 0
            amd.rnn_apply_learning(g.net, rc.WEIGHTED, 0.9)
        single.char_step(text, i, rc.WEIGHTED, 0.9)
        o.char_step(text, i, rc.WEIGHTED, 0.9)
    snaps = [g.snapshot() for g in ranks]
    ss, so = single.snapshot(), o.snapshot()
    # replicas are identical ...
    for k in ("ih_w", "ho_w", "ih_m", "ho_m"):
        assert np.array_equal(snaps[0][k], snaps[1][k]), k
    # ... and equal the single device set and the oracle (the order of the sum over streams differs)
    for want in (ss, so):
        replay.check(snaps[0], want, RTOL, keys=["ih_w", "ho_w", "ih_m", "ho_m"], exact=())
        # the external buffer holds the global sum; ih_delta of the snapshot comes from it
        replay.check(snaps[1], want, RTOL, keys=["ih_delta", "ho_delta"], exact=())
        both = {k: np.concatenate([snaps[0][k], snaps[1][k]], axis=1 if k == "hist" else 0)
                for k in ("hist", "hidden", "output", "o_error", "min_error_factor", "ih_scale", "index",
                          "generation", "rng")}
        replay.check(both, want, RTOL, keys=["hist", "hidden", "output", "o_error", "min_error_factor",
                                             "ih_scale"], exact=("index", "generation", "rng"))
    for r, g in enumerate(ranks):
        amd.rnn_amd_set_external_delta(g.handle, None)
        g.close()
        hip.hipFree(bufs[r])
    single.close()
    o.close()


@pytest.mark.parametrize("world,label,kw,method", [
    (2, "dma_path", dict(input_size=42, hidden_size=128, output_size=42, D=6, learn_rate=2e-3, seed=21), rc.WEIGHTED),
    (4, "dma_path_4", dict(input_size=42, hidden_size=128, output_size=42, D=6, learn_rate=2e-3, seed=24), rc.NESTEROV),
    (8, "one_launch_chain_8", dict(input_size=42, hidden_size=256, output_size=42, D=5, learn_rate=1e-3, seed=25), rc.WEIGHTED),
    (3, "ragged_3", dict(input_size=42, hidden_size=45, output_size=42, D=7, learn_rate=5e-3, seed=23), rc.ADAGRAD),
    # a rule with a second accumulator (aux arrays: they too live in the owner's range only)
    (2, "rprop_aux", dict(input_size=42, hidden_size=64, output_size=42, D=5, learn_rate=1e-3, seed=26,
                          flags=rc.FLAG_STANDARD | rc.FLAG_ADAPTIVE_MIN_ERROR | rc.FLAG_AUX_ARRAYS), rc.RPROP),
])
def test_kernel_issued_exchange_with_emulated_ranks(amd, world, label, kw, method):
    """The exchange step without a collective library (include/recur_amd.h: rnn_amd_set_exchange_*; kernels_apply.hip:
    k_apply_xchg): every rank adds the ranks' local delta sums for ITS range of the weight arrays in rank order through
    peer pointers, updates weights and momentum there and stores the new weights into every rank's arrays.  `world`
    virtual ranks = `world` engines of this process in lock step (deltas on every rank, then the update on every rank):
    after 10 generations the replicas' weights are bit-identical, they and the momentum assembled from the ranks' own
    ranges equal the single set of all the streams and the oracle at 1e-4, the generator states are exact."""
    S = 16 if "ragged" not in label else 3
    steps = 10
    text = sc.synthetic_text(6000)
    ranks = [sc.AmdBatchedSet(amd, S=S, shard=(r * S, world * S), **kw) for r in range(world)]
    blobs = C.create_string_buffer(rc.RNN_AMD_EXCHANGE_BLOB_BYTES * world)
    for r, g in enumerate(ranks):
        g.load_text(text)
        amd.rnn_amd_set_shard(g.handle, r * S, world * S)
        if method == rc.ADAGRAD:
            amd.rnn_set_momentum_values(g.net, 0.1)
        if method == rc.RPROP:
            amd.rnn_set_aux_values(g.net, 1e-4)
        amd.rnn_amd_set_exchange_export(g.handle, C.byref(blobs, r * rc.RNN_AMD_EXCHANGE_BLOB_BYTES))
    for r, g in enumerate(ranks):
        assert amd.rnn_amd_set_exchange_join(g.handle, r, world, blobs, None, 1) == 0
    single = sc.AmdBatchedSet(amd, S=world * S, **kw)
    single.load_text(text)
    o = sc.OracleSet(S=world * S, **kw)
    if method == rc.ADAGRAD:
        amd.rnn_set_momentum_values(single.net, 0.1)
        o.arrays()["ih_m"][:] = 0.1
        o.arrays()["ho_m"][:] = 0.1
    if method == rc.RPROP:
        amd.rnn_set_aux_values(single.net, 1e-4)
        o.arrays()["ih_aux"][:] = 1e-4
        o.arrays()["ho_aux"][:] = 1e-4
    for i in range(steps):
        for g in ranks:
            amd.rnn_amd_set_char_step_deltas(g.handle, i)
        for g in ranks:
            amd.rnn_amd_set_apply_exchange(g.handle, method, 0.9)
        single.char_step(text, i, method, 0.9)
        o.char_step(text, i, method, 0.9)
    snaps = [g.snapshot() for g in ranks]
    ss, so = single.snapshot(), o.snapshot()
    for r in range(1, world):  # replicas: identical weights
        for k in ("ih_w", "ho_w"):
            assert np.array_equal(snaps[0][k], snaps[r][k]), (k, r)
    # momentum and summed deltas live in each rank's own range: put them together
    joined = {k: np.zeros_like(ss[k]).reshape(-1) for k in ("ih_m", "ho_m", "ih_delta", "ho_delta")}
    first, count = C.c_size_t(), C.c_size_t()
    for r, g in enumerate(ranks):
        for which, names in ((0, ("ih_m", "ih_delta")), (1, ("ho_m", "ho_delta"))):
            amd.rnn_amd_set_exchange_range(g.handle, which, C.byref(first), C.byref(count))
            for k in names:
                joined[k][first.value:first.value + count.value] = snaps[r][k].reshape(-1)[first.value:first.value + count.value]
    got = dict(snaps[0])
    for k in joined:
        got[k] = joined[k].reshape(ss[k].shape)
    for want in (ss, so):
        if method == rc.RPROP and want is so:
            continue  # (RPROP steps by the SIGN of a delta: against another summation order a delta within rounding of zero
            #           flips a step; the single device set shares the streams' order within a rank's sum)
        replay.check(got, want, RTOL if method != rc.RPROP else 2e-3, keys=["ih_w", "ho_w", "ih_m", "ho_m", "ih_delta", "ho_delta"], exact=())
        both = {k: np.concatenate([sn[k] for sn in snaps], axis=1 if k == "hist" else 0)
                for k in ("hist", "hidden", "min_error_factor", "ih_scale", "index", "generation", "rng")}
        replay.check(both, want, RTOL, keys=["hist", "hidden", "min_error_factor", "ih_scale"],
                     exact=("index", "generation", "rng"))
    for g in ranks:
        amd.rnn_amd_set_exchange_leave(g.handle)
        g.close()
    single.close()
    o.close()


XCHG_IPC_SCRIPT = r"""
import ctypes as C, mmap, os, sys, time, json
sys.path.insert(0, %(tests)r)
import numpy as np, recur_ctypes as rc, scenarios as sc
rank, world, shm_path = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
B = rc.RNN_AMD_EXCHANGE_BLOB_BYTES
fd = os.open(shm_path, os.O_RDWR)
mm = mmap.mmap(fd, 4096)          # [0, 64): the arrival counters; [256 ...): the blobs; [3072 + r]: "blob r is there"
amd = rc.load_amd()
amd.rnn_amd_use_device(0, None)
S = 16
kw = dict(input_size=42, hidden_size=128, output_size=42, D=6, learn_rate=2e-3, seed=21)
text = sc.synthetic_text(6000)
g = sc.AmdBatchedSet(amd, S=S, shard=(rank * S, world * S), **kw)
g.load_text(text)
amd.rnn_amd_set_shard(g.handle, rank * S, world * S)
blob = C.create_string_buffer(B)
amd.rnn_amd_set_exchange_export(g.handle, blob)
mm[256 + rank * B:256 + (rank + 1) * B] = blob.raw
mm[3072 + rank] = 1
t0 = time.time()
while not all(mm[3072 + r] for r in range(world)):
    assert time.time() - t0 < 120, "the other rank never exported"
    time.sleep(0.01)
blobs = C.create_string_buffer(bytes(mm[256:256 + world * B]), world * B)
counters = (C.c_char * 64).from_buffer(mm, 0)
assert amd.rnn_amd_set_exchange_join(g.handle, rank, world, blobs, counters, 0) == 0
if os.environ.get("XCHG_TEST_RANK1_STALLS") and rank == 1:
    time.sleep(300)  # (joined, never steps: the other rank's first barrier has to give up; the test ends this process)
for i in range(8):
    amd.rnn_amd_set_char_step(g.handle, i, rc.WEIGHTED, 0.9)
s = g.snapshot()
np.savez(shm_path + ".rank%%d.npz" %% rank, ih_w=s["ih_w"], ho_w=s["ho_w"])
amd.rnn_amd_set_exchange_leave(g.handle)
del counters
print("RESULT ok")
"""


def test_kernel_issued_exchange_between_two_processes_on_one_gpu(amd, tmp_path):
    """The same exchange between two PROCESSES that share this GPU (RCCL refuses duplicate GPUs; this exchange does not):
    the ranks' delta and weight arrays cross the process boundary as hipIpcMemHandles inside the 256-byte blobs, the
    arrival counters are a file in /dev/shm that both map, every generation is rnn_amd_set_char_step = deltas -> barrier
    kernel -> k_apply_xchg -> barrier kernel.  Both replicas must end bit-identical and equal the single set of all the
    streams at 1e-4."""
    shm = "/dev/shm/recur_amd_xchg_%d" % os.getpid()
    with open(shm, "wb") as f:
        f.write(b"\0" * 4096)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    script = XCHG_IPC_SCRIPT % {"tests": os.path.dirname(os.path.abspath(__file__))}
    try:
        procs = [subprocess.Popen([sys.executable, "-c", script, str(r), "2", shm], stdout=subprocess.PIPE,
                                  stderr=subprocess.PIPE, text=True, env=env) for r in range(2)]
        outs = [p.communicate(timeout=600) for p in procs]
        for p, (so_, se_) in zip(procs, outs):
            assert p.returncode == 0 and "RESULT ok" in so_, so_[-1500:] + se_[-3000:]
        a, b = (np.load(shm + ".rank%d.npz" % r) for r in range(2))
        for k in ("ih_w", "ho_w"):
            assert np.array_equal(a[k], b[k]), k
        kw = dict(input_size=42, hidden_size=128, output_size=42, D=6, learn_rate=2e-3, seed=21)
        text = sc.synthetic_text(6000)
        single = sc.AmdBatchedSet(amd, S=32, **kw)
        single.load_text(text)
        for i in range(8):
            single.char_step(text, i, rc.WEIGHTED, 0.9)
        ss = single.snapshot()
        for k in ("ih_w", "ho_w"):
            assert rc.rel_err(a[k], ss[k]) < RTOL and rc.max_err(a[k], ss[k]) < RTOL, k
        single.close()
    finally:
        for f in (shm, shm + ".rank0.npz", shm + ".rank1.npz"):
            if os.path.exists(f):
                os.unlink(f)


def test_regrow_keeps_the_text_and_the_external_delta_buffer(amd, hip):
    """A clone made after rnn_amd_set_open (a validation net, say) regrows the device image;
    the registered text and the external delta buffer must survive it."""
    kw = dict(input_size=42, hidden_size=64, output_size=42, S=4, D=5, learn_rate=1e-3, seed=4)
    text = sc.synthetic_text(3000)
    g = sc.AmdBatchedSet(amd, **kw)
    g.load_text(text)
    n_delta = g.I * g.H + g.H * g.O
    p = C.c_void_p()
    assert hip.hipMalloc(C.byref(p), 4 * n_delta) == 0
    amd.rnn_amd_set_external_delta(g.handle, p)
    o = sc.OracleSet(**kw)
    for i in range(3):
        amd.rnn_amd_set_char_step_deltas(g.handle, i)
        amd.rnn_apply_learning(g.net, rc.WEIGHTED, 0.9)
        o.char_step(text, i, rc.WEIGHTED, 0.9)
    fl = g.net.contents.flags & ~(rc.FLAG_OWN_WEIGHTS | rc.FLAG_OWN_BPTT)
    extra = [amd.rnn_clone(g.net, fl, rc.SUBSEED, None) for _ in range(3)]  # Fcap 1 -> 3: a regrow
    rc.view(extra[2].contents.real_inputs, 42)[:] = 0
    amd.rnn_opinion(extra[2], None, 0.0)
    o.orc.orc_rand64(C.byref(o.z.contents.rng[0]))  # the three SUBSEED draws from the prototype
    o.orc.orc_rand64(C.byref(o.z.contents.rng[0]))
    o.orc.orc_rand64(C.byref(o.z.contents.rng[0]))
    for i in range(3, 6):
        amd.rnn_amd_set_char_step_deltas(g.handle, i)
        amd.rnn_amd_synchronize()
        amd.rnn_apply_learning(g.net, rc.WEIGHTED, 0.9)
        o.char_step(text, i, rc.WEIGHTED, 0.9)
    h = np.zeros(n_delta, np.float32)
    assert hip.hipMemcpy(h.ctypes.data, p, 4 * n_delta, D2H) == 0
    so = o.snapshot()
    assert rc.rel_err(h[:g.I * g.H], so["ih_delta"].reshape(-1)) < RTOL  # still landing in the caller's buffer
    replay.check(g.snapshot(), so, RTOL, keys=["ih_w", "ho_w", "ih_m", "hidden", "hist"],
                 exact=("index", "generation", "rng"))
    for e in extra:
        amd.rnn_delete_net(e)
    amd.rnn_amd_set_external_delta(g.handle, None)
    g.close()
    hip.hipFree(p)
    o.close()


RCCL_SCRIPT = r"""
import ctypes as C, sys, json
sys.path.insert(0, %(tests)r)
import numpy as np, recur_ctypes as rc, scenarios as sc
amd = rc.load_amd()
amd.rnn_amd_use_device(0, None)
text = sc.synthetic_text(6000)
kw = dict(input_size=42, hidden_size=128, output_size=42, S=32, D=6, learn_rate=2e-3, seed=21)
def run(n):
    g = sc.AmdBatchedSet(amd, **kw)
    g.load_text(text)
    for i in range(n):
        g.char_step(text, i, rc.WEIGHTED, 0.9)
    s = g.snapshot()
    g.close()
    return s
plain = run(10)
buf = C.create_string_buffer(128)
assert amd.rnn_amd_dist_get_id(buf) == 0
assert amd.rnn_amd_dist_init(0, 1, buf) == 0
assert amd.rnn_amd_dist_world() == 1 and amd.rnn_amd_dist_rank() == 0
assert amd.rnn_amd_dist_max(3.5) == 3.5
amd.rnn_amd_dist_barrier()
joined = run(10)
amd.rnn_amd_dist_finalize()
after = run(10)
out = {}
for k in ("ih_w", "ho_w", "ih_m", "ho_m", "ih_delta", "ho_delta", "hidden", "hist"):
    out[k] = [rc.rel_err(joined[k], plain[k]), rc.rel_err(after[k], plain[k])]
out["rng"] = bool(np.array_equal(joined["rng"], plain["rng"]))
print("RESULT " + json.dumps(out))
"""


def test_library_exchange_step_through_rccl_with_one_rank():
    """rnn_amd_dist_init(0, 1, id) -> the generation goes deltas -> k_delta_finalize -> RCCL
    all-reduce on the library's stream -> rnn_apply_learning; with one rank the sum is the
    identity, so the weights must equal the plain path's (to the slab summation, i.e. exactly
    or within rounding).  (RECUR_AMD_DIST_ONE_RANK_EXCHANGE=1: with one rank the library otherwise skips the
    exchange step altogether.)"""
    env = dict(os.environ, RECUR_AMD_DIST_ONE_RANK_EXCHANGE="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-c", RCCL_SCRIPT % {"tests": os.path.dirname(os.path.abspath(__file__))}],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    import json
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1]
    res = json.loads(line[7:])
    assert res.pop("rng")
    for k, (joined, after) in res.items():
        assert joined < 1e-6 and after == 0.0, (k, joined, after)


def test_two_halves_exchange_equals_the_one_launch_form():
    """RECUR_AMD_DIST_OVERLAP=1 (off by default: DESIGN.md section 6): the weight-delta GEMM as two row halves,
    each half's all-reduce on the library's second stream as soon as its deltas are complete, the update
    behind both.  Hidden 1024 (eight row tiles, rest rows riding with the upper half), one rank: the sum is
    the identity, so weights, momentum and deltas must equal the plain path's to the summation order of
    the K slabs."""
    env = dict(os.environ, RECUR_AMD_DIST_OVERLAP="1", RECUR_AMD_DIST_ONE_RANK_EXCHANGE="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    script = RCCL_SCRIPT.replace("hidden_size=128, output_size=42, S=32, D=6, learn_rate=2e-3",
                                 "hidden_size=1024, output_size=42, S=64, D=5, learn_rate=1e-4")
    r = subprocess.run([sys.executable, "-c", script % {"tests": os.path.dirname(os.path.abspath(__file__))}],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    import json
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
    assert res.pop("rng")
    for k, (joined, after) in res.items():
        assert joined < 2e-6 and after == 0.0, (k, joined, after)


def test_c_driver_with_G_1_goes_through_the_library_exchange_step():
    """tools/text_predict_amd -G 1: one forked-process-per-GPU launcher in plain C, the RCCL id
    handed over in shared memory, rnn_amd_dist_init, the shard constructor, the all-reduce
    inside rnn_amd_set_char_step.  With one rank the curve must be the plain run's."""
    import re
    exe = os.path.join(rc.ROOT, "build", "text_predict_amd")
    if not os.path.exists(exe):
        subprocess.run(["make", "-s", "-C", os.path.join(rc.ROOT, "recur_amd", "csrc")], check=True)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rows = {}
    for tag, extra in (("plain", []), ("G1", ["-G", "1"])):
        r = subprocess.run([exe, "-f", rc.EREWHON, "-H", "128", "-t", "32", "-d", "8", "-l", "1e-3", "-s", "150",
                            "-r", "50", "-V", "1200"] + extra, capture_output=True, text=True, timeout=600,
                           cwd=rc.ROOT, env=env)
        assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
        rows[tag] = re.findall(r"generation\s+(\d+) t_entropy ([\d.]+) v_entropy ([\d.]+) accuracy ([\d.]+)", r.stdout)
        assert [int(x[0]) for x in rows[tag]] == [50, 100, 150]
    for a, b in zip(rows["plain"], rows["G1"]):
        assert abs(float(a[1]) - float(b[1])) < 2e-4 and abs(float(a[2]) - float(b[2])) < 2e-3, (a, b)


def test_a_rank_that_never_steps_ends_the_other_with_a_message_not_with_half_a_sum():
    """Two processes join the kernel-issued exchange; rank 1 never steps.  Rank 0's first arrival barrier polls its time out
    (20 s by the device's clock; 3 s here), gives
    up, raises the abort word (code 6) -- and the device's own copy of it, which is what k_apply_xchg reads (round 6: it
    read the host's word, every thread, over PCIe: 50 us per launch) so that no weights are made from half a sum --, and
    rank 0's next synchronisation ends the process with the message that says what happened."""
    shm = "/dev/shm/recur_amd_xchg_stall_%d" % os.getpid()
    with open(shm, "wb") as f:
        f.write(b"\0" * 4096)
    env = dict(os.environ, XCHG_TEST_RANK1_STALLS="1", RECUR_AMD_XCHG_BARRIER_TIMEOUT_S="3")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    script = XCHG_IPC_SCRIPT % {"tests": os.path.dirname(os.path.abspath(__file__))}
    procs = [subprocess.Popen([sys.executable, "-c", script, str(r), "2", shm], stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True, env=env) for r in range(2)]
    try:
        so_, se_ = procs[0].communicate(timeout=240)
        assert procs[0].returncode != 0 and "RESULT ok" not in so_
        assert "at a barrier of the kernel-issued exchange" in se_, se_[-2000:]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()  # these exact children
            p.wait()
        for f in (shm, shm + ".rank0.npz", shm + ".rank1.npz"):
            if os.path.exists(f):
                os.unlink(f)


# ---- bench.py with more than one rank validates itself (round 6; the sum replaced is recur-nn.c:724-739)

def _bench(args, env_extra, timeout=900):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **env_extra)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=env, capture_output=True, text=True,
                       timeout=timeout)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None)


def test_bench_with_two_ranks_says_of_itself_whether_it_was_a_sharded_run():
    """`bench.py --gpus 2` as the driver would start it, the two ranks sharing this box's one GPU (RECUR_BENCH_SHARE_GPU=1:
    no RCCL group -- it refuses duplicate devices --, the kernel-issued exchange between the two processes): the start-up
    cross-check runs (one update through the exchange, replicas compared through a kernel's loads and over a copy), the
    JSON line carries every rank's checksum, replicas_identical, exchange_us and which exchange ran and why."""
    r, out = _bench(["--gpus", "2", "--steps", "10", "--warmup", "2", "--prefill", "30", "--streams", "64",
                     "--no-cpu-baseline"], {"RECUR_BENCH_SHARE_GPU": "1"})
    assert r.returncode == 0 and out, r.stdout[-2000:] + r.stderr[-3000:]
    assert out["n_gpus"] == 2 and out["config"]["global_streams"] == 128
    assert out["replicas_identical"] is True
    assert len(out["replica_checksums"]) == 2 and len(set(out["replica_checksums"])) == 1
    assert out["replica_checksums"] == out["replica_checksums_over_a_copy"]
    assert out["exchange_us"] > 0 and out["config"]["exchange"] == "kernel-issued"
    sel = out["config"]["exchange_selection"]
    assert sel["kernel_join"] and sel["kernel_replicas_identical"] and sel["kernel_us_per_generation"] > 0
    assert out["config"]["ranks_share_one_gpu"] is True and out["config"]["rccl_ranks"] == 0
    assert "chosen at start-up" in out["config"]["parallelism"]
    assert out["value"] > 0 and abs(out["value"] - 10 * 128 / (out["ms_per_step"] * 1e-3 * 10)) < 1e-6 * out["value"]


def test_bench_fails_when_the_replicas_differ():
    """The same run with ONE weight of the last rank's replica moved by one ulp after the timed region
    (RECUR_BENCH_INJECT_FAULT=replica): replicas_identical is false, the ranks' checksums differ, exit status 3."""
    r, out = _bench(["--gpus", "2", "--steps", "4", "--warmup", "1", "--prefill", "25", "--streams", "32",
                     "--no-cpu-baseline", "--no-roofline"], {"RECUR_BENCH_SHARE_GPU": "1", "RECUR_BENCH_INJECT_FAULT": "replica"})
    assert r.returncode == 3, (r.returncode, r.stderr[-2000:])
    assert out and out["replicas_identical"] is False and len(set(out["replica_checksums"])) == 2
    assert "DIFFER" in r.stderr


def test_replica_checksum_through_a_kernel_and_over_a_copy(amd):
    """rnn_amd_set_replica_checksum: sum of word_i (2 i + 1) mod 2^64 over ih_weights || ho_weights (|| the momentum
    arrays), by a kernel and over a copy -- equal to each other and to numpy's over a snapshot, and it moves when one
    weight moves by one ulp."""
    g = sc.AmdBatchedSet(amd, input_size=42, hidden_size=128, output_size=42, S=16, D=5, learn_rate=1e-3, seed=4)
    text = sc.synthetic_text(3000)
    for i in range(7):
        g.char_step(text, i, rc.WEIGHTED, 0.9)
    s = g.snapshot()

    def want(keys):
        words = np.concatenate([s[k].reshape(-1).view(np.uint32) for k in keys]).astype(np.uint64)
        idx = np.arange(words.size, dtype=np.uint64)
        with np.errstate(over="ignore"):
            return int((words * (np.uint64(2) * idx + np.uint64(1))).sum(dtype=np.uint64))

    for with_m, keys in ((0, ("ih_w", "ho_w")), (1, ("ih_w", "ho_w", "ih_m", "ho_m"))):
        dev = amd.rnn_amd_set_replica_checksum(g.handle, with_m, 1)
        host = amd.rnn_amd_set_replica_checksum(g.handle, with_m, 0)
        assert dev == host == want(keys), (with_m, hex(dev), hex(host), hex(want(keys)))
    before = amd.rnn_amd_set_replica_checksum(g.handle, 0, 1)
    amd.rnn_amd_sync_host(g.net, rc.RNN_AMD_WEIGHTS)
    w = rc.view(g.net.contents.ho_weights, g.H, g.O)
    w[3, 2] = np.nextafter(w[3, 2], np.float32(9.0))
    amd.rnn_amd_host_written(g.net, rc.RNN_AMD_WEIGHTS)
    assert amd.rnn_amd_set_replica_checksum(g.handle, 0, 1) != before
    g.close()


REJOIN_SCRIPT = r"""
import ctypes as C, sys
sys.path.insert(0, %(tests)r)
import numpy as np, recur_ctypes as rc, scenarios as sc
amd = rc.load_amd()
amd.rnn_amd_use_device(0, None)
kw = dict(input_size=42, hidden_size=64, output_size=42, S=16, D=5, learn_rate=1e-3, seed=8)
text = sc.synthetic_text(3000)
g = sc.AmdBatchedSet(amd, **kw)
g.load_text(text)
counters = (C.c_uint32 * 16)()
blob = C.create_string_buffer(rc.RNN_AMD_EXCHANGE_BLOB_BYTES)
i = 0
for session in range(2):
    amd.rnn_amd_set_exchange_export(g.handle, blob)
    assert amd.rnn_amd_set_exchange_join(g.handle, 0, 1, blob, counters, 0) == 0
    for _ in range(8):
        amd.rnn_amd_set_char_step(g.handle, i, rc.WEIGHTED, 0.9)
        i += 1
    amd.rnn_amd_synchronize()
    amd.rnn_amd_set_exchange_leave(g.handle)
    assert counters[0] == 16 * (session + 1), (session, counters[0])  # two barriers per generation, counted ON
plain = sc.AmdBatchedSet(amd, **kw)
for k in range(16):
    plain.char_step(text, k, rc.WEIGHTED, 0.9)
a, b = g.snapshot(), plain.snapshot()
for k in ("ih_w", "ho_w"):
    assert rc.rel_err(a[k], b[k]) < 1e-6, k
print("RESULT ok")
"""


def test_exchange_rejoins_on_counters_an_earlier_session_left():
    """rnn_amd_set_exchange_join counts its barriers on from where the shared counters stand (round 5 counted from 0 and
    trusted the launcher to have zeroed them: on counters an earlier session left, the first barriers would have passed
    at once).  One rank with real barriers (counters in host memory; RECUR_AMD_DIST_ONE_RANK_EXCHANGE=1 keeps the step
    in, read once per process: a subprocess), eight generations, leave, join AGAIN on the same counters, eight more: the
    counter stands at 16, then 32, and the weights equal a plain set's."""
    env = dict(os.environ, RECUR_AMD_DIST_ONE_RANK_EXCHANGE="1")
    r = subprocess.run([sys.executable, "-c", REJOIN_SCRIPT % {"tests": os.path.dirname(os.path.abspath(__file__))}],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "RESULT ok" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


FINEGRAINED_SCRIPT = r"""
import sys, json, hashlib
sys.path.insert(0, %(tests)r)
import numpy as np, recur_ctypes as rc, scenarios as sc
amd = rc.load_amd()
amd.rnn_amd_use_device(0, None)
g = sc.AmdBatchedSet(amd, input_size=42, hidden_size=1024, output_size=42, S=32, D=10, learn_rate=1e-4, seed=3)
text = sc.synthetic_text(5000)
for i in range(14):
    g.char_step(text, i, rc.WEIGHTED, 0.9)
s = g.snapshot()
print("RESULT " + json.dumps({k: hashlib.sha256(s[k].tobytes()).hexdigest() for k in ("ih_w", "ho_w", "ih_m", "ih_delta")}))
"""


def test_exchanged_arrays_as_fine_grained_allocations_train_the_same_bits():
    """RECUR_AMD_XCHG_FINEGRAINED=1 (for a node on which bench.py's start-up cross-check finds the
    kernel-issued exchange's replicas differing): weights and delta sums as hipExtMallocWithFlags(fine-grained) allocations
    -- the same fourteen generations at hidden 1024, the same bits in weights, momentum and deltas."""
    out = []
    for fine in ("0", "1"):
        r = subprocess.run([sys.executable, "-c", FINEGRAINED_SCRIPT % {"tests": os.path.dirname(os.path.abspath(__file__))}],
                           capture_output=True, text=True, env=dict(os.environ, RECUR_AMD_XCHG_FINEGRAINED=fine), timeout=600)
        assert r.returncode == 0 and "RESULT " in r.stdout, r.stdout[-1000:] + r.stderr[-2000:]
        out.append(json.loads(r.stdout.split("RESULT ", 1)[1]))
    assert out[0] == out[1]
