#!/usr/bin/env python3
"""bench.py -- BPTT timesteps*streams/sec of the RNN core on MI355X.

One "step" is one generation of the synchronic mini-batch: for every stream
advance + forward + softmax loss + full truncated-BPTT backward (depth 20) +
weight-delta accumulation, then one optimiser update -- the loop body of the
reference's rnn_char_epoch multi-tap branch (charmodel-predict.c:288-311),
driven through librecur_amd.so's C ABI with the text resident in HBM.

Workload (BASELINE.json configs / SURVEY.md section 8(d)): hidden 1024,
256 streams per GPU, BPTT depth 20, RELU, 42 inputs / 42 outputs
(i_size 1068, h_size 1028, o_size 44), flags STANDARD | ADAPTIVE_MIN_ERROR,
flat semicircle init seed 1, weighted momentum 0.95 / 0.5, learn rate 1e-5,
synthetic symbol stream (rand_small_int seed 7, 42 symbols).

Multi-GPU: one process per GPU, streams sharded over ranks (weak scaling: 256
streams per GPU), ONE RCCL all-reduce of ih_delta||ho_delta per generation INSIDE
the library (rnn_amd_dist_*; recur_amd/csrc/dist.c), replicated optimiser step.
`--gpus N` without a launcher's WORLD_SIZE starts the N ranks itself (children are
spawned before anything touches a GPU; a wall-clock watchdog ends them all and
exits 124 when a run hangs); under `python -m torch.distributed.run`
the ranks are the launcher's.  No PyTorch in the data path (nor imported).

A run with more than one rank validates itself (round 6): the JSON line carries
`rccl_ranks`, every rank's 64-bit replica checksum -- read by a kernel through the
caches AND over a copy -- with `replicas_identical`, and `exchange_us` (HIP events
around the exchange step, per generation); replicas that differ end the run with
exit status 3.  `--exchange auto` (the default with N > 1) makes one update through
RCCL and one through the kernel-issued exchange from the SAME deltas, times both,
and takes the kernel-issued form only if its replicas are identical, its result
is RCCL's to rounding, and it is faster; `config.parallelism` says which and why.

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import mmap
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
TESTS = os.path.join(ROOT, "tests")  # the checker's loaders (oracle, compiled reference): the cpu_baseline legs only

HIDDEN, STREAMS, DEPTH, ALPHABET = 1024, 256, 20, 42
LEARN_RATE, MOMENTUM = 1e-5, 0.95
PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_HBM_GBS = 8000.0
# HBM-side bytes per launch come from the PMC passes of THIS round's kernels, summarised by
# tools/pmc_summary.py into this file (rocprofv3 cannot run inside the timed process)
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r06_pmc_traffic.json")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--hidden", type=int, default=HIDDEN)
    ap.add_argument("--streams", type=int, default=STREAMS, help="streams per GPU")
    ap.add_argument("--depth", type=int, default=DEPTH)
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak (default, the driver's contract): --streams per GPU; strong: --streams-global "
                         "streams in all, split evenly over the ranks (BASELINE.json configs[3]: 256 over 8)")
    ap.add_argument("--streams-global", type=int, default=STREAMS, help="--scaling strong: streams of the whole job")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of EACH cpu leg")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--prefill", type=int, default=300, help="untimed generations before the warm-up")
    ap.add_argument("--dist", action="store_true",
                    help="join an RCCL group even with one rank (exercises the exchange step)")
    ap.add_argument("--exchange", choices=("auto", "rccl", "kernel"), default=os.environ.get("RECUR_AMD_DIST_EXCHANGE", "auto"),
                    help="the exchange step of a sharded run: one RCCL all-reduce per generation, the library's "
                         "kernel-issued reduce-scatter -> sharded update -> all-gather through peer pointers "
                         "(rnn_amd_set_exchange_*; also with one rank), or auto (default): with more than one rank both "
                         "are cross-checked and timed at the start and the kernel-issued form is taken only if its "
                         "replicas are identical, its result equals RCCL's to rounding and it is faster; one rank: rccl")
    ap.add_argument("--rank-timeout", type=float, default=float(os.environ.get("RECUR_BENCH_RANK_TIMEOUT", "1500")),
                    help="--gpus N without a launcher: seconds of wall clock after which the parent ends every rank "
                         "(the exact children it started) and exits 124")
    ap.add_argument("--all-cores-leg", action="store_true", help=argparse.SUPPRESS)
    return ap.parse_args()


# ------------------------------------------------------------------ launcher --

def spawn_ranks(n, limit_s):
    """--gpus N with no launcher: start N ranks of this script (one per GPU) as child
    processes, BEFORE this process touches a GPU, and relay rank 0's JSON line.  The parent
    never touches a GPU: it only watches -- a rank that dies takes the others with it, and a
    run that is still going after limit_s seconds of wall clock (a rank stuck in a collective,
    a barrier nobody else reaches) is ended child by child, exit status 124."""
    t_start = time.time()
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    out0 = tempfile.TemporaryFile()
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RECUR_BENCH_RUN_ID=str(os.getpid()))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL))
    # a rank that dies would leave the others waiting in a collective: watch them all
    rcs = [None] * n
    while any(rc is None for rc in rcs):
        for r, p in enumerate(procs):
            if rcs[r] is None:
                rcs[r] = p.poll()
        timed_out = time.time() - t_start > limit_s
        if timed_out or any(rc not in (None, 0) for rc in rcs):
            if timed_out:
                sys.stderr.write("bench.py: %d ranks still running after %.0f s (--rank-timeout): ending them\n"
                                 % (sum(rc is None for rc in rcs), limit_s))
            else:
                time.sleep(2.0)
            for r, p in enumerate(procs):
                if p.poll() is None:
                    p.kill()  # this exact child
                rcs[r] = p.wait()
            if timed_out:
                sys.exit(124)
            break
        time.sleep(0.05)
    out0.seek(0)
    sys.stdout.write(out0.read().decode())
    sys.stdout.flush()
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        sys.stderr.write("bench.py: ranks failed (rank, exit code): %s\n" % bad)
        sys.exit(max(1, min(255, max(abs(rc) for _, rc in bad))))
    sys.exit(0)


def exchange_id(amd, rank, world):
    """rank 0's RCCL id reaches the others through a file (single node): no PyTorch, and
    the launcher's own store on MASTER_PORT is left alone."""
    key = "%s_%s" % (os.environ.get("MASTER_PORT", "0"),
                     os.environ.get("RECUR_BENCH_RUN_ID") or os.getppid())
    path = os.path.join(tempfile.gettempdir(), "recur_amd_rccl_id_" + key)
    buf = C.create_string_buffer(128)
    if rank == 0:
        if amd.rnn_amd_dist_get_id(buf) != 0:
            raise SystemExit("bench.py: cannot get an RCCL id")
        with open(path + ".tmp", "wb") as f:
            f.write(buf.raw)
        os.replace(path + ".tmp", path)
    else:
        t0 = time.time()
        while True:
            try:
                if os.path.getsize(path) == 128 and os.path.getmtime(path) > t0 - 300:
                    break
            except OSError:
                pass
            if time.time() - t0 > 300:
                raise SystemExit("bench.py: rank %d never saw rank 0's RCCL id" % rank)
            time.sleep(0.02)
        buf.raw = open(path, "rb").read()
    if amd.rnn_amd_dist_init(rank, world, buf) != 0:
        raise SystemExit("bench.py: rnn_amd_dist_init failed on rank %d" % rank)
    if rank == 0:  # the init is collective: everybody has read the file
        os.unlink(path)


class HostGroup:
    """The ranks of ONE node meeting in a 4 KB file in /dev/shm that all of them map: the kernel-issued exchange's arrival
    counters and blobs, and -- host side only, nothing of the data path -- a barrier and an all-gather of 8-byte values
    (checksums, times, decisions), so that what the JSON line says about the ranks does not itself depend on the
    collective library it is checking.  Layout: [0, 64) the exchange's counters; [64, 128) one barrier word per rank;
    [128, 192) "blob r is there"; [256, 256 + 8 x 256) blobs; [3072, 3072 + 8 x 8) the gather's slots."""
    B = 256

    def __init__(self, rank, world, timeout_s=300.0):
        import struct
        self.struct = struct
        self.rank, self.world, self.timeout_s = rank, world, timeout_s
        key = "%s_%s" % (os.environ.get("MASTER_PORT", "0"), os.environ.get("RECUR_BENCH_RUN_ID") or
                         (os.getppid() if world > 1 else os.getpid()))
        self.path = "/dev/shm/recur_amd_bench_" + key
        t0 = time.time()
        if rank == 0:
            with open(self.path + ".tmp", "wb") as f:
                f.write(b"\0" * 4096)
            os.replace(self.path + ".tmp", self.path)
        while not (os.path.exists(self.path) and os.path.getmtime(self.path) > t0 - 300):
            if time.time() - t0 > timeout_s:
                raise SystemExit("bench.py: rank %d never saw the ranks' file %s" % (rank, self.path))
            time.sleep(0.02)
        self.fd = os.open(self.path, os.O_RDWR)
        self.mm = mmap.mmap(self.fd, 4096)
        self.seq = 0
        self.counters = None

    def barrier(self):
        self.seq += 1
        self.struct.pack_into("<I", self.mm, 64 + 4 * self.rank, self.seq)
        t0 = time.time()
        while True:
            if all(self.struct.unpack_from("<I", self.mm, 64 + 4 * r)[0] >= self.seq for r in range(self.world)):
                return
            if time.time() - t0 > self.timeout_s:
                raise SystemExit("bench.py: rank %d waited %.0f s at a host barrier: a rank is missing" % (self.rank, self.timeout_s))
            time.sleep(0.0002)

    def gather(self, value, fmt="<Q"):
        """every rank's 8-byte value, in rank order, on every rank"""
        self.struct.pack_into(fmt, self.mm, 3072 + 8 * self.rank, value)
        self.barrier()
        got = [self.struct.unpack_from(fmt, self.mm, 3072 + 8 * r)[0] for r in range(self.world)]
        self.barrier()
        return got

    def join_kernel_exchange(self, amd, handle):
        """every rank's 256-byte blob (IPC handles of its delta and weight arrays) to every rank, then the join on the
        shared counters (which need no zeroing: the library counts on from where they stand)"""
        B = self.B
        blob = C.create_string_buffer(B)
        amd.rnn_amd_set_exchange_export(handle, blob)
        self.mm[256 + self.rank * B:256 + (self.rank + 1) * B] = blob.raw
        self.barrier()
        blobs = C.create_string_buffer(bytes(self.mm[256:256 + self.world * B]), self.world * B)
        if self.counters is None:
            self.counters = (C.c_char * 64).from_buffer(self.mm, 0)
        rc_ = amd.rnn_amd_set_exchange_join(handle, self.rank, self.world, blobs, self.counters, 0)
        ok = min(self.gather(1 if rc_ == 0 else 0))  # (also the rendezvous: nobody steps before everybody has joined)
        if ok != 1 and rc_ == 0:
            amd.rnn_amd_set_exchange_leave(handle)  # (another rank could not join: nobody uses this exchange)
        return ok == 1

    def close(self):
        self.barrier()  # nobody maps the counters any more
        self.counters = None
        try:
            self.mm.close()
        except BufferError:
            pass
        os.close(self.fd)
        if self.rank == 0 and os.path.exists(self.path):
            os.unlink(self.path)


def replica_checksums(amd, gpu, group, with_momentum):
    """every rank's checksum of its replica, formed by a kernel (through the caches) and over a copy: lists in rank order"""
    dev = amd.rnn_amd_set_replica_checksum(gpu.handle, int(with_momentum), 1)
    host = amd.rnn_amd_set_replica_checksum(gpu.handle, int(with_momentum), 0)
    if group is None:
        return [dev], [host]
    return group.gather(dev), group.gather(host)


def select_exchange(amd, rc, gpu, group, rank, world, i, have_rccl, n_time=20):
    """--exchange auto with more than one rank: ONE update through each exchange from the same local deltas and the same
    weights, then n_time generations of each, timed.  The kernel-issued form is taken only if (1) after its update every
    rank's replica is the same, read through the caches and over a copy -- that is the visibility of peers' stores that
    include/recur_amd.h says is not established between devices --, (2) its weights are RCCL's to rounding (the two add
    the ranks' sums in different orders), (3) it is faster.  Every rank decides from the same gathered numbers.
    Returns (use_kernel, report, next i)."""
    import numpy as np
    W = rc.RNN_AMD_WEIGHTS | rc.RNN_AMD_MOMENTUMS | rc.RNN_AMD_DELTAS
    rep = {"generations_timed": n_time}
    net = gpu.net
    n0, b0 = net.contents, net.contents.bptt.contents
    views = lambda: {"ih_w": rc.view(n0.ih_weights, gpu.I, gpu.H), "ho_w": rc.view(n0.ho_weights, gpu.H, gpu.O),
                     "ih_m": rc.view(b0.ih_momentum, gpu.I, gpu.H), "ho_m": rc.view(b0.ho_momentum, gpu.H, gpu.O),
                     "ih_delta": rc.view(b0.ih_delta, gpu.I, gpu.H), "ho_delta": rc.view(b0.ho_delta, gpu.H, gpu.O)}

    def restore(saved):
        amd.rnn_amd_sync_host(net, W)
        for k, v in views().items():
            v[:] = saved[k]
        amd.rnn_amd_host_written(net, W)

    def timed(n, i):
        amd.rnn_amd_synchronize()
        group.barrier()
        t0 = time.perf_counter()
        for _ in range(n):
            amd.rnn_amd_set_char_step(gpu.handle, i, rc.WEIGHTED, MOMENTUM)
            i += 1
        amd.rnn_amd_synchronize()
        group.barrier()
        return max(group.gather(time.perf_counter() - t0, "<d")) / n * 1e6, i

    # the local deltas of one generation, and the state they are applied to
    amd.rnn_amd_set_char_step_deltas(gpu.handle, i)
    i += 1
    amd.rnn_amd_sync_host(net, W)
    saved = {k: v.copy() for k, v in views().items()}
    # -- the kernel-issued exchange
    kernel_ok = group.join_kernel_exchange(amd, gpu.handle)
    rep["kernel_join"] = bool(kernel_ok)
    wk = None
    if kernel_ok:
        amd.rnn_amd_set_apply_exchange(gpu.handle, rc.WEIGHTED, MOMENTUM)
        dev, host = replica_checksums(amd, gpu, group, False)
        rep["kernel_replicas_identical"] = len(set(dev)) == 1 and dev == host
        amd.rnn_amd_sync_host(net, rc.RNN_AMD_WEIGHTS)
        wk = {k: views()[k].copy() for k in ("ih_w", "ho_w")}
        rep["kernel_us_per_generation"], i = timed(n_time, i)
        amd.rnn_amd_set_exchange_leave(gpu.handle)
        group.barrier()
    restore(saved)  # replicated weights, whole momentum arrays, the SAME local deltas
    use_kernel = False
    if have_rccl:
        amd.rnn_amd_set_dist_all_reduce_deltas(gpu.handle)
        amd.rnn_apply_learning(net, rc.WEIGHTED, MOMENTUM)
        dev, host = replica_checksums(amd, gpu, group, True)
        rep["rccl_replicas_identical"] = len(set(dev)) == 1 and dev == host
        if wk is not None:
            amd.rnn_amd_sync_host(net, rc.RNN_AMD_WEIGHTS)
            err = max(float(np.abs(views()[k] - wk[k]).max() / max(np.abs(wk[k]).max(), 1e-30)) for k in wk)
            rep["kernel_vs_rccl_max_err"] = max(group.gather(err, "<d"))
        rep["rccl_us_per_generation"], i = timed(n_time, i)
        use_kernel = bool(kernel_ok and rep["kernel_replicas_identical"] and rep["kernel_vs_rccl_max_err"] <= 1e-6 and
                          rep["kernel_us_per_generation"] < rep["rccl_us_per_generation"])
        rep["why"] = ("kernel-issued: replicas identical, RCCL's result to %.1e, %.1f against %.1f us per generation"
                      % (rep["kernel_vs_rccl_max_err"], rep["kernel_us_per_generation"], rep["rccl_us_per_generation"])
                      ) if use_kernel else (
            "rccl: " + ("the peers' arrays could not be opened" if not kernel_ok else
                        "the kernel-issued exchange left replicas that differ" if not rep["kernel_replicas_identical"] else
                        "the kernel-issued update is %.1e from RCCL's" % rep["kernel_vs_rccl_max_err"]
                        if rep["kernel_vs_rccl_max_err"] > 1e-6 else
                        "%.1f us per generation against the kernel-issued %.1f" % (rep["rccl_us_per_generation"],
                                                                                  rep["kernel_us_per_generation"])))
    else:  # ranks that share a GPU (the one-GPU emulation): RCCL refuses duplicate devices
        use_kernel = bool(kernel_ok and rep["kernel_replicas_identical"])
        rep["why"] = "kernel-issued: the ranks share a GPU (no RCCL group); replicas identical: %s" % rep.get("kernel_replicas_identical")
        if not use_kernel:
            raise SystemExit("bench.py: ranks that share a GPU have only the kernel-issued exchange, and it failed: %r" % rep)
    if min(group.gather(1 if use_kernel else 0)) != max(group.gather(1 if use_kernel else 0)):
        raise SystemExit("bench.py: the ranks disagree about the exchange to use")
    return use_kernel, rep, i


# --------------------------------------------------------------- CPU baseline --

def cpu_one_core(rc, sc, gpu_set, text, i_next, budget_s):
    """The CPU path on ONE host core (the reference is single-threaded), continued from
    the GPU's own post-warm-up state (full history ring, realistic zero-row sparsity) on
    32 of the streams.  Uses oracle/_ref (the real reference, -Ofast) when it travelled
    here, else the repo's restatement."""
    S_cpu = min(32, gpu_set.S)
    kw = dict(input_size=ALPHABET, hidden_size=gpu_set.hidden_size, output_size=ALPHABET,
              S=S_cpu, D=gpu_set.D, learn_rate=LEARN_RATE, seed=1)
    snap = gpu_set.snapshot()
    if rc.have_ref() and os.path.exists(rc.REF_FAST_LIB):
        kind = "reference"
        ref = rc.load_ref(fast=True)
        cpu = sc.ApiSet(ref, softmax_best_guess=ref.ref_softmax_best_guess, **kw)
        n0 = cpu.net.contents
        rc.view(n0.ih_weights, cpu.I, cpu.H)[:] = snap["ih_w"]
        rc.view(n0.ho_weights, cpu.H, cpu.O)[:] = snap["ho_w"]
        rc.view(n0.bptt.contents.ih_momentum, cpu.I, cpu.H)[:] = snap["ih_m"]
        rc.view(n0.bptt.contents.ho_momentum, cpu.H, cpu.O)[:] = snap["ho_m"]
        for j in range(S_cpu):
            n = cpu.nets[j].contents
            b = n.bptt.contents
            rc.view(b.history, cpu.D, cpu.I)[:] = snap["hist"][:, j, :]
            rc.view(n.hidden_layer, cpu.H)[:] = snap["hidden"][j]
            while b.index != snap["index"][j]:
                ref.rnn_bptt_advance(cpu.nets[j])
            b.min_error_factor = float(snap["min_error_factor"][j])
    else:
        kind = "port"
        cpu = sc.OracleSet(fast=True, **kw)
        a = cpu.arrays()
        for k in ("ih_w", "ho_w", "ih_m", "ho_m"):
            a[k][:] = snap[k]
        a["hist"][:] = snap["hist"][:, :S_cpu, :]
        a["hidden"][:] = snap["hidden"][:S_cpu]
        a["index"][:] = snap["index"][:S_cpu]
        a["min_error_factor"][:] = snap["min_error_factor"][:S_cpu]
    gens = 0
    t0 = time.perf_counter()
    while True:
        cpu.char_step(text, i_next + gens, rc.WEIGHTED, MOMENTUM)
        gens += 1
        el = time.perf_counter() - t0
        if el >= budget_s or gens >= 600:
            break
    cpu.close()
    return {
        "value": gens * S_cpu / el, "unit": "stream-timesteps/s", "cores": 1, "kind": kind,
        "sample": "%d generations x %d streams (hidden %d, depth %d) continued from the GPU's "
                  "post-warm-up state, %.1f s" % (gens, S_cpu, gpu_set.hidden_size, gpu_set.D, el),
    }


def _all_cores_worker(p, P, s_p, hidden, depth, text, n_floats, slabs_mm, result_mm, flags_mm, barrier,
                      budget_s, q):
    """One of P processes of the best-effort CPU run: s_p of the streams, a full replica of
    the weights, per generation the deltas of its streams -> slab p; everybody sums one
    1/P-th of the range over all slabs -> result; everybody applies the summed deltas."""
    import numpy as np
    sys.path.insert(0, TESTS)
    import recur_ctypes as rc
    import scenarios as sc
    try:
        os.sched_setaffinity(0, {sorted(os.sched_getaffinity(0))[p % len(os.sched_getaffinity(0))]})
    except (AttributeError, OSError):
        pass
    kw = dict(input_size=ALPHABET, hidden_size=hidden, output_size=ALPHABET, S=s_p, D=depth,
              learn_rate=LEARN_RATE, seed=1)
    if rc.have_ref() and os.path.exists(rc.REF_FAST_LIB):
        kind = "reference"
        ref = rc.load_ref(fast=True)
        cpu = sc.ApiSet(ref, softmax_best_guess=ref.ref_softmax_best_guess, **kw)
        cpu.global_first, cpu.global_count = p * s_p, P * s_p
        b0 = cpu.net.contents.bptt.contents
        ih, ho = rc.view(b0.ih_delta, cpu.I * cpu.H), rc.view(b0.ho_delta, cpu.H * cpu.O)
        apply = lambda: ref.rnn_apply_learning(cpu.net, rc.WEIGHTED, MOMENTUM)
    else:
        kind = "port"
        cpu = sc.OracleSet(fast=True, **kw)
        cpu.z.contents.global_first, cpu.z.contents.global_count = p * s_p, P * s_p
        a = cpu.arrays()
        ih, ho = a["ih_delta"].reshape(-1), a["ho_delta"].reshape(-1)
        apply = lambda: cpu.orc.orc_apply_learning(cpu.z, rc.WEIGHTED, MOMENTUM)
    slabs = np.frombuffer(slabs_mm, dtype=np.float32).reshape(P, n_floats)
    result = np.frombuffer(result_mm, dtype=np.float32)
    flags = np.frombuffer(flags_mm, dtype=np.int32)  # [0]: stop
    lo, hi = (n_floats * p) // P, (n_floats * (p + 1)) // P
    warm = depth + 2
    gens, t0, el = 0, None, 0.0
    g = 0
    while True:
        if g == warm:
            t0 = time.perf_counter()
        cpu.char_step_deltas(text, g)
        slabs[p, :ih.size] = ih
        slabs[p, ih.size:] = ho
        barrier.wait(timeout=600)
        stop = int(flags[0])  # written before the barrier above, read after it: the same everywhere
        result[lo:hi] = slabs[:, lo:hi].sum(axis=0, dtype=np.float32)
        barrier.wait(timeout=600)
        ih[:] = result[:ih.size]
        ho[:] = result[ih.size:]
        apply()
        g += 1
        if t0 is not None:
            gens += 1
            el = time.perf_counter() - t0
        if stop:
            break
        if p == 0 and t0 is not None and (el >= budget_s or gens >= 400):
            flags[0] = 1
    q.put((p, kind, gens, el))


def _all_cores_run(args, text, P, budget_s):
    import multiprocessing as mp
    S = args.streams
    s_p = S // P
    I = (args.hidden + ALPHABET + 1 + 3) // 4 * 4
    H = (args.hidden + 1 + 3) // 4 * 4
    O = (ALPHABET + 3) // 4 * 4
    n_floats = I * H + H * O
    ctx = mp.get_context("fork")
    slabs_mm = mmap.mmap(-1, P * n_floats * 4)
    result_mm = mmap.mmap(-1, n_floats * 4)
    flags_mm = mmap.mmap(-1, 64)
    barrier = ctx.Barrier(P)
    q = ctx.Queue()
    procs = [ctx.Process(target=_all_cores_worker,
                         args=(p, P, s_p, args.hidden, args.depth, text, n_floats, slabs_mm, result_mm,
                               flags_mm, barrier, budget_s, q), daemon=True) for p in range(P)]
    for pr in procs:
        pr.start()
    try:
        got = [q.get(timeout=budget_s * 6 + 240) for _ in range(P)]
    except Exception as e:  # a worker died: report nothing rather than a wrong number
        for pr in procs:
            pr.terminate()
        return {"error": "run with %d processes failed: %r" % (P, e)}
    for pr in procs:
        pr.join(timeout=30)
    for mm in (slabs_mm, result_mm, flags_mm):
        mm.close()
    _, kind, gens, _ = sorted(got)[0]
    el = max(t[3] for t in got)
    return {
        "value": gens * S / el, "unit": "stream-timesteps/s", "cores": P, "kind": kind,
        "sample": "%d generations x %d streams as %d processes x %d streams (hidden %d, depth %d), "
                  "delta sum through shared memory every generation, replicated update, after %d "
                  "warm-up generations, %.1f s" % (gens, S, P, s_p, args.hidden, args.depth,
                                                   args.depth + 2, el),
    }


def cpu_all_cores(args, text):
    """Best-effort CPU figure: the SAME 256-stream
    workload sharded over the host's cores, one process per shard, per-generation delta sum
    through shared memory, replicated update.  More processes mean less arithmetic each but
    more copies of the 23 MB update, so a few process counts are tried and the best is
    reported (all of them listed).  Runs before this process touches the GPU (it forks)."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    S = args.streams
    cands = [P for P in (256, 128, 64, 32, 16, 8, 4, 2, 1) if P <= cores and P <= S and S % P == 0][:4]
    if not cands:
        cands = [1]
    each = max(2.5, args.cpu_seconds / len(cands))
    tried, best = {}, None
    for P in cands:
        r = _all_cores_run(args, text, P, each)
        tried[str(P)] = r.get("value", r.get("error"))
        if "value" in r and (best is None or r["value"] > best["value"]):
            best = r
    out = dict(best) if best else {"error": "no all-core run completed"}
    out["cores_available"] = cores
    out["tried_processes_to_rate"] = tried
    return out


# ------------------------------------------------------------------------ main --

def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus, args.rank_timeout)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    hang = os.environ.get("RECUR_BENCH_TEST_HANG")
    if "WORLD_SIZE" in os.environ and (hang == "all" or (hang == "1" and rank == world - 1)):
        time.sleep(3600)  # (tests/test_dist_host.py: a rank that never arrives -- the parent's watchdog has to end the run)
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import numpy as np
    from recur_amd import api as rc, drivers as sc  # the measured path's driver lives with the package

    if args.scaling == "strong":
        if args.streams_global % world or (args.streams_global // world) % 16:
            raise SystemExit("bench.py --scaling strong: %d streams over %d ranks is not whole 16-stream tiles per rank"
                             % (args.streams_global, world))
        args.streams = args.streams_global // world
    S, D, Hd = args.streams, args.depth, args.hidden
    # untimed, before the warm-up: the history ring is full whatever --warmup is (D + 5), and the
    # device has ramped to the clocks it holds under this load (measured: a 20-step timed region
    # right after 30 generations runs 2.4 % slower than after 300; sustained rate is the metric)
    prefill = max(D + 5, args.prefill)
    total_steps = prefill + args.warmup + args.steps + 400
    text = sc.synthetic_text_np(max(20000, S * world * 40 + total_steps + 16), ALPHABET, 7)

    if args.all_cores_leg:
        # the all-core CPU figure in a process of its own (it forks its workers and never touches
        # the GPU): started by the measuring process AFTER the GPU legs, see below
        print(json.dumps(cpu_all_cores(args, text)), flush=True)
        return

    if args.dist and world == 1:  # (with one rank the library skips the exchange step unless told to keep it in)
        os.environ.setdefault("RECUR_AMD_DIST_ONE_RANK_EXCHANGE", "1")
    if world > 1 and os.environ.get("RECUR_BENCH_SHARE_GPU", "0") == "1":
        # ranks that share a GPU: two one-launch chains (256 workgroups each, one per CU, waiting for one another) cannot
        # both be resident -- the emulation takes the launch-per-step chain (the library reads its switches once, at load)
        os.environ.setdefault("RECUR_AMD_CHAIN_PERSIST", "0")
    amd = rc.load_amd()
    ndev = amd.rnn_amd_device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs a GPU (librecur_amd has no CPU fallback)")
    # RECUR_BENCH_SHARE_GPU=1: every rank on device 0 -- the one-GPU emulation of a sharded run (tests/test_gpu_dist.py):
    # no RCCL group (it refuses duplicate devices), the kernel-issued exchange between the processes, host-side bracket
    share_gpu = world > 1 and os.environ.get("RECUR_BENCH_SHARE_GPU", "0") == "1"
    if share_gpu:
        local_rank = 0
    if local_rank >= ndev:
        raise SystemExit("bench.py: rank %d of %d but only %d GPU(s) are visible" % (rank, world, ndev))
    amd.rnn_amd_use_device(local_rank, None)
    dist = (world > 1 or args.dist or os.environ.get("RECUR_BENCH_FORCE_DIST", "0") == "1") and not share_gpu
    if dist:
        exchange_id(amd, rank, world)
        assert amd.rnn_amd_dist_world() == world and amd.rnn_amd_dist_rank() == rank
    group = HostGroup(rank, world) if (world > 1 or args.exchange == "kernel") else None

    gpu = sc.AmdBatchedSet(amd, input_size=ALPHABET, hidden_size=Hd, output_size=ALPHABET, S=S,
                           D=D, learn_rate=LEARN_RATE, seed=1, momentum=MOMENTUM,
                           shard=(rank * S, world * S))
    gpu.load_text(text)
    amd.rnn_amd_set_shard(gpu.handle, rank * S, world * S)

    def step(i):  # deltas -> (the exchange between the ranks, inside the library) -> update
        amd.rnn_amd_set_char_step(gpu.handle, i, rc.WEIGHTED, MOMENTUM)

    def fence():
        amd.rnn_amd_synchronize()
        if dist:
            amd.rnn_amd_dist_barrier()
        elif group is not None:
            group.barrier()
        amd.rnn_amd_synchronize()

    i = 0
    # which exchange (more than one rank): asked for, or cross-checked and timed here (select_exchange)
    xchg, selection = False, None
    if args.exchange == "kernel" and group is not None:
        if not group.join_kernel_exchange(amd, gpu.handle):
            raise SystemExit("bench.py: rnn_amd_set_exchange_join failed (rank %d)" % rank)
        xchg = True
    elif world > 1 and (args.exchange == "auto" or share_gpu):
        for _ in range(D + 5):  # a full history ring first: the deltas that are cross-checked are a real generation's
            step(i)
            i += 1
        fence()
        use_kernel, selection, i = select_exchange(amd, rc, gpu, group, rank, world, i, have_rccl=dist)
        if use_kernel:
            if not group.join_kernel_exchange(amd, gpu.handle):
                raise SystemExit("bench.py: the second rnn_amd_set_exchange_join failed (rank %d)" % rank)
            xchg = True

    for _ in range(prefill + args.warmup):
        step(i)
        i += 1
    fence()
    st = rc.AmdStats()
    amd.rnn_amd_set_read_stats(gpu.handle, C.byref(st), 1)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(i)
        i += 1
    fence()
    elapsed = time.perf_counter() - t0
    # max over ranks (identity for one rank)
    elapsed = amd.rnn_amd_dist_max(elapsed) if dist else max(group.gather(elapsed, "<d")) if group is not None else elapsed
    amd.rnn_amd_set_read_stats(gpu.handle, C.byref(st), 1)
    mean_depth = st.bptt_depth_sum / max(st.count, 1)
    zero_frac = st.hidden_zeros / max(st.count, 1)
    t_entropy = -st.entropy / max(st.count, 1)

    # roofline leg: HIP events around the GEMM launches, on the launch stream, over a further
    # short run of the same loop (kept out of the timed region: the event records cost time)
    roofline = None
    exchange_us = 0.0 if world == 1 and not (dist or xchg) else None
    if not args.no_roofline:
        I, H = gpu.I, gpu.H
        amd.rnn_amd_kernel_time_enable(1)
        amd.rnn_amd_kernel_time_ms(0, None, 1)
        n_roof = min(args.steps, 50)
        for _ in range(n_roof):
            step(i)
            i += 1
        fence()
        amd.rnn_amd_kernel_time_enable(0)
        cls = {}
        for which, name in ((0, "bptt_chain_gemm"), (1, "delta_gemm"), (2, "forward_gemm"),
                            (3, "optimiser")):
            n = C.c_long(0)
            ms = amd.rnn_amd_kernel_time_ms(which, C.byref(n), 0)
            cls[name] = (ms, n.value)
        n_x = C.c_long(0)
        exchange_us = 1e3 * amd.rnn_amd_kernel_time_ms(5, C.byref(n_x), 0) / n_roof  # (class 5: include/recur_amd.h)
        amd.rnn_amd_kernel_time_ms(0, None, 1)
        amd.rnn_amd_set_read_stats(gpu.handle, C.byref(st), 1)
        d_exec = st.bptt_depth_sum / max(st.count, 1)
        gI, gH, gO = (Hd + ALPHABET + 1 + 3) // 4 * 4, (Hd + 1 + 3) // 4 * 4, (ALPHABET + 3) // 4 * 4  # padded sizes
        # the kernels' OWN work per generation: the chain multiplies the hidden x hidden block
        # (the extras' columns are another kernel's), the delta GEMM every row of W
        fused_update = cls["optimiser"][1] == 0  # no optimiser launch: the delta GEMM's epilogue updated both layers
        per_gen_flops = {
            "bptt_chain_gemm": 2.0 * S * Hd * Hd * D,
            "delta_gemm": 2.0 * I * Hd * S * d_exec,
        }
        per_gen_bytes = {  # algorithmic: W ONCE PER LAUNCH (it stays in registers) + per step E in, X mask,
            # E out / X, E, dW once
            # (+ since round 4 the launch's tail: the extras and the control logic read every error plane once more)
            "bptt_chain_gemm": 4.0 * (Hd * Hd + 3 * S * Hd * D) + 4.0 * S * (D + 1) * gH,
            # X ([S D][I]) and E ([S D][hidden]) once; then, with the update in the launch's epilogue (one rank, the
            # momentum rule: round 5), weights and momentum of BOTH layers read and written and the deltas stored --
            # 5 passes over [I][hidden] and [H][O] --, otherwise the delta's one store
            "delta_gemm": 4.0 * (S * d_exec * (I + Hd) + (5 if fused_update else 1) * (I * Hd + gH * gO)),
        }
        dom = max(per_gen_flops, key=lambda k: cls[k][0])
        ms, n = cls[dom]
        launches_per_gen = max(n, 1) / n_roof
        avg_us = 1e3 * ms / max(n, 1)
        flop_launch = per_gen_flops[dom] / launches_per_gen
        bytes_launch = per_gen_bytes[dom] / launches_per_gen
        achieved = flop_launch / (avg_us * 1e-6) / 1e12 if n else 0.0
        frac_mfma = achieved / PEAK_FP32_MFMA_TFLOPS
        frac_hbm = bytes_launch / (avg_us * 1e-6) / 1e9 / PEAK_HBM_GBS if n else 0.0
        traffic, traffic_src = None, None
        if (Hd, S, D) == (HIDDEN, STREAMS, DEPTH) and os.path.exists(TRAFFIC_FILE):
            t = json.load(open(TRAFFIC_FILE)).get(dom)
            # the counters come from a committed profile of this command, not from this run: refuse them when
            # the kernel timed here is no longer the kernel that was profiled (its duration moved by > 20 %)
            was = (t or {}).get("avg_us_when_profiled")
            if t and was and abs(avg_us - was) > 0.2 * was:
                traffic_src = "profiles/%s is stale (kernel %.1f us when profiled, %.1f us now): not reported" % (
                    os.path.basename(TRAFFIC_FILE), was, avg_us)
                t = None
            if t:
                traffic = t["bytes_per_launch"] * t.get("launches_per_generation", launches_per_gen) / launches_per_gen
                traffic_src = "profiles/%s (rocprofv3 --pmc passes of this bench command, 2 x FETCH_SIZE + " \
                              "WRITE_SIZE)" % os.path.basename(TRAFFIC_FILE)
        roofline = {
            "bound": "mfma" if frac_mfma >= frac_hbm else "hbm", "kernel": dom, "achieved": achieved,
            "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": frac_mfma,
            "traffic": traffic, "traffic_source": traffic_src,
            "algorithmic_bytes": bytes_launch, "frac_of_hbm_peak": frac_hbm,
            "avg_launch_us": avg_us, "launches": n, "launches_per_generation": launches_per_gen,
            "flop_per_launch": flop_launch,
            "classes_ms_per_step": {k: v[0] / n_roof for k, v in cls.items()},
            "steps": n_roof,
            # both MFMA-bound kernels, not only the longer one
            "per_kernel": {
                k: {"avg_launch_us": 1e3 * cls[k][0] / max(cls[k][1], 1),
                    "achieved": per_gen_flops[k] / (max(cls[k][1], 1) / n_roof) / (1e3 * cls[k][0] / max(cls[k][1], 1) * 1e-6) / 1e12
                    if cls[k][1] else 0.0}
                for k in per_gen_flops},
            # the whole generation by SURVEY.md section 8(d)'s count (forward + top layer + D_exec BPTT steps with their
            # weight-delta accumulation), over the timed region's ms_per_step
            "generation": {"flop": S * (2.0 * gI * gH + 2.0 * gH * gO + 2.0 * (2.0 * gH * gO) + d_exec * 4.0 * gI * gH),
                           "note": "bptt_chain_gemm's launch includes the extras and the control logic of all steps "
                                   "(its tail, ~6 us of the launch) since round 4; delta_gemm's launch includes the "
                                   "update of both layers (weights, momentum) since round 5: no optimiser launch"},
        }
        for k, d in roofline["per_kernel"].items():
            d["frac"] = d["achieved"] / PEAK_FP32_MFMA_TFLOPS
            lpg = max(cls[k][1], 1) / n_roof
            d["algorithmic_bytes"] = per_gen_bytes[k] / lpg
            d["frac_of_hbm_peak"] = d["algorithmic_bytes"] / (d["avg_launch_us"] * 1e-6) / 1e9 / PEAK_HBM_GBS if cls[k][1] else 0.0
        gen = roofline["generation"]
        gen["achieved"] = gen["flop"] / (elapsed / args.steps) / 1e12
        gen["frac"] = gen["achieved"] / PEAK_FP32_MFMA_TFLOPS

    # the replicas after the timed region and the roofline leg (behind both: the copies and the host's sum would let the
    # device's clocks drop in front of the leg's 50 generations): every rank's checksum of its weights (and momentum, where every rank holds
    # all of it: the kernel-issued exchange keeps a range's momentum on its owner), read by a kernel and over a copy
    if os.environ.get("RECUR_BENCH_INJECT_FAULT") == "replica" and rank == world - 1:
        # (tests: one weight of the last rank's replica moves by one ulp -- the run must say so and fail)
        import numpy as np
        amd.rnn_amd_sync_host(gpu.net, rc.RNN_AMD_WEIGHTS)
        w = rc.view(gpu.net.contents.ih_weights, gpu.I, gpu.H)
        w[5, 5] = np.nextafter(w[5, 5], np.float32(2.0))
        amd.rnn_amd_host_written(gpu.net, rc.RNN_AMD_WEIGHTS)
    cks_dev, cks_host = replica_checksums(amd, gpu, group if world > 1 else None, with_momentum=not xchg)
    replicas_identical = len(set(cks_dev)) == 1 and cks_dev == cks_host

    out = {
        "metric": "BPTT timesteps*streams/sec at 1024-hidden/256-stream",
        "value": args.steps * S * world / elapsed,
        "unit": "stream-timesteps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": "text-predict multi-tap generation: hidden %d, %d streams/GPU, BPTT depth "
                        "%d, 42 symbols, RELU, weighted momentum, lr 1e-5" % (Hd, S, D),
            "streams_per_gpu": S, "global_streams": S * world, "scaling_mode": args.scaling,
            "parallelism": ("streams sharded x%d, kernel-issued reduce-scatter -> sharded update -> all-gather through "
                            "peer pointers per generation (in librecur_amd)" % world) if xchg else
                           ("streams sharded x%d, one RCCL all-reduce of the weight deltas per generation "
                            "(in librecur_amd)" % world) if dist else "single GPU",
            "rccl_ranks": amd.rnn_amd_dist_world() if dist else 0,
            "exchange": "kernel-issued" if xchg else "rccl" if dist else "none",
            "exchange_selection": selection,
            "ranks_share_one_gpu": share_gpu,
            "untimed_prefill_generations": prefill,
            "mean_bptt_depth": mean_depth, "hidden_zero_fraction": zero_frac,
            "training_entropy_bits": t_entropy,
            # how tests/ read north_star's "1e-4 relative" (DESIGN.md section 4): 2-norm AND largest element at 1e-4, and
            # element by element |a - b| <= 1e-4 |b| on the elements >= 1e-2 of an array's largest (1e-1 in the hot
            # regime) -- below that floor the reference's own strict and -Ofast builds are up to 3.9e-4 apart
            "parity_bar": {"rtol": 1e-4, "elementwise_floor_of_max": 1e-2, "elementwise_floor_hot_regime": 1e-1},
        },
    }
    if selection:
        out["config"]["parallelism"] += " -- chosen at start-up (--exchange auto): " + selection["why"]
    # a sharded run says of itself whether it was one: the group RCCL reports, the replicas after the timed region
    # (bit-identical by construction on both exchange paths), what the exchange step cost a rank per generation
    out["replicas_identical"] = replicas_identical
    out["replica_checksums"] = ["%016x" % c for c in cks_dev]
    out["replica_checksums_over_a_copy"] = ["%016x" % c for c in cks_host]
    out["replica_checksum_of"] = "ih_weights||ho_weights" + ("" if xchg else "||ih_momentum||ho_momentum")
    out["exchange_us"] = exchange_us
    if world > 1 and not share_gpu and out["config"]["rccl_ranks"] != world:
        raise SystemExit("bench.py: %d ranks but the RCCL group has %d" % (world, out["config"]["rccl_ranks"]))
    if roofline is not None:
        out["roofline"] = roofline
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, TESTS)  # the checker's loaders (compiled reference, oracle): this leg only
        import recur_ctypes as checker_rc
        import scenarios as checker_sc
        base = cpu_one_core(checker_rc, checker_sc, gpu, text, i, args.cpu_seconds)
        # The all-core leg comes LAST and in a child process: run before the GPU legs, its half
        # minute of load on every host core left the GPU measurably slower for the timed region
        # that followed (chain kernel 133.8 against 127.8 us, 879 k against 900 k stream-timesteps/s,
        # same box, alternating runs), which is a property of the bench's order, not of the path.
        all_cores = None
        try:
            import subprocess
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--all-cores-leg", "--streams", str(S),
                                "--depth", str(D), "--hidden", str(Hd), "--cpu-seconds", str(args.cpu_seconds),
                                "--steps", str(args.steps), "--warmup", str(args.warmup),
                                "--prefill", str(args.prefill)],
                               capture_output=True, text=True, timeout=600)
            all_cores = json.loads(r.stdout.strip().splitlines()[-1])
        except Exception as e:  # the baseline is a reported figure, not the measurement
            all_cores = {"error": "all-core leg failed: %r" % (e,)}
        base["cores_available"] = all_cores.get("cores_available") if all_cores else None
        base["all_cores"] = all_cores
        out["cpu_baseline"] = base
    if xchg:
        amd.rnn_amd_set_exchange_leave(gpu.handle)
    if group is not None:
        group.close()  # (a barrier first: every rank has left, nobody maps the counters any more)
    if dist:
        amd.rnn_amd_dist_barrier()
        amd.rnn_amd_dist_finalize()
    if rank == 0:
        # RCCL writes a version banner through C stdio; push it out first so that the JSON
        # line is the last thing on stdout
        sys.stdout.flush()
        C.CDLL(None).fflush(None)
        print(json.dumps(out), flush=True)
    if not replicas_identical:
        sys.stderr.write("bench.py: the ranks' replicas DIFFER after the timed region (checksums through a kernel %s, over "
                         "a copy %s): the number above is not a valid measurement\n" % (out["replica_checksums"],
                                                                                      out["replica_checksums_over_a_copy"]))
        sys.exit(3)


if __name__ == "__main__":
    main()
