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

Multi-GPU: one process per GPU (torchrun), streams sharded over ranks (weak
scaling: 256 streams per GPU), one RCCL all-reduce of ih_delta||ho_delta per
generation, replicated optimiser step.

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HIDDEN, STREAMS, DEPTH, ALPHABET = 1024, 256, 20, 42
LEARN_RATE, MOMENTUM = 1e-5, 0.95
PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_HBM_GBS = 8000.0
# HBM-side bytes per launch of the dominant kernel at the default shape, from the PMC passes
# kept in profiles/r01_pmc_fetch_write_per_kernel.txt (2 x FETCH_SIZE + WRITE_SIZE, the gfx950
# correction of MI355X_MICROARCH.md); it cannot be read live without the profiler.
TRAFFIC_BYTES_PER_LAUNCH = {"bptt_chain_gemm": (2 * 7398.5 + 1312.0) * 1024,   # k_chain_main
                            "delta_gemm": (2 * 33946.5 + 17410.2) * 1024}      # k_delta_dma


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--hidden", type=int, default=HIDDEN)
    ap.add_argument("--streams", type=int, default=STREAMS, help="streams per GPU")
    ap.add_argument("--depth", type=int, default=DEPTH)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-roofline", action="store_true")
    return ap.parse_args()


def cpu_baseline(rc, sc, amd, gpu_set, text, i_next, budget_s):
    """Times the CPU path on the GPU box's host cores from the GPU's own
    post-warm-up state (so the history ring is full and the zero-row skip sees
    realistic sparsity).  Uses oracle/_ref (the real reference, -Ofast) when it
    travelled here, else the repo's restatement.  Single-threaded like the
    reference."""
    import numpy as np
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
        "value": gens * S_cpu / el,
        "unit": "stream-timesteps/s",
        "cores": 1,
        "kind": kind,
        "sample": "%d generations x %d streams (hidden %d, depth %d) continued from the GPU's "
                  "post-warm-up state, %.1f s" % (gens, S_cpu, gpu_set.hidden_size, gpu_set.D, el),
    }


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import numpy as np
    import torch
    import recur_ctypes as rc
    import scenarios as sc

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (librecur_amd has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dist = None
    # RECUR_BENCH_FORCE_DIST=1 exercises the sharded path (RCCL group, external delta buffer,
    # all-reduce) even with one rank: a single-GPU check of the multi-GPU plumbing
    force_dist = os.environ.get("RECUR_BENCH_FORCE_DIST", "0") == "1"
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    amd = rc.load_amd()
    amd.rnn_amd_use_device(local_rank, C.c_void_p(torch.cuda.current_stream().cuda_stream))

    S, D, Hd = args.streams, args.depth, args.hidden
    total_steps = args.warmup + args.steps + 260
    text = sc.synthetic_text(max(20000, S * world * 40 + total_steps + 16), ALPHABET, 7)
    gpu = sc.AmdBatchedSet(amd, input_size=ALPHABET, hidden_size=Hd, output_size=ALPHABET, S=S,
                           D=D, learn_rate=LEARN_RATE, seed=1, momentum=MOMENTUM)
    gpu.load_text(text)
    from recur_amd.dist import shard_range
    first, _, total = shard_range(rank, world, S)
    amd.rnn_amd_set_shard(gpu.handle, first, total)
    delta = None
    if dist is not None:
        delta = torch.zeros(gpu.I * gpu.H + gpu.H * gpu.O, dtype=torch.float32, device="cuda")
        amd.rnn_amd_set_external_delta(gpu.handle, C.c_void_p(delta.data_ptr()))

    from recur_amd.dist import ShardedStep
    if dist is not None:
        step = ShardedStep(lambda i: amd.rnn_amd_set_char_step_deltas(gpu.handle, i),
                           lambda: dist.all_reduce(delta),
                           lambda: amd.rnn_apply_learning(gpu.net, rc.WEIGHTED, MOMENTUM))
    else:
        # one GPU: nothing happens between the deltas and the update, so the library's own
        # generation call is used (same kernels; the delta slabs go straight into the update)
        def step(i):
            amd.rnn_amd_set_char_step(gpu.handle, i, rc.WEIGHTED, MOMENTUM)

    def fence():
        amd.rnn_amd_synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    i = 0
    for _ in range(args.warmup):
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
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    amd.rnn_amd_set_read_stats(gpu.handle, C.byref(st), 1)
    mean_depth = st.bptt_depth_sum / max(st.count, 1)
    zero_frac = st.hidden_zeros / max(st.count, 1)
    t_entropy = -st.entropy / max(st.count, 1)

    # roofline leg: HIP events around every GEMM launch, on the launch stream,
    # over a further short run of the same loop (kept out of the timed region so
    # that the event records do not perturb `value`)
    roofline = None
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
        amd.rnn_amd_kernel_time_ms(0, None, 1)
        amd.rnn_amd_set_read_stats(gpu.handle, C.byref(st), 1)
        d_exec = st.bptt_depth_sum / max(st.count, 1)
        flops = {
            "bptt_chain_gemm": 2.0 * S * I * H,            # per launch: one BPTT step, all streams
            "delta_gemm": 2.0 * I * H * S * d_exec,         # per launch: all executed steps
            "forward_gemm": 2.0 * S * I * H,
        }
        dom = max(("bptt_chain_gemm", "delta_gemm"), key=lambda k: cls[k][0])
        ms, n = cls[dom]
        avg_us = 1e3 * ms / max(n, 1)
        achieved = flops[dom] / (avg_us * 1e-6) / 1e12 if n else 0.0
        roofline = {
            "bound": "mfma", "kernel": dom, "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS,
            "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_MFMA_TFLOPS,
            "traffic": (TRAFFIC_BYTES_PER_LAUNCH.get(dom)
                        if (Hd, S, D) == (HIDDEN, STREAMS, DEPTH) else None),
            "algorithmic_bytes": 4.0 * (I * H + 3 * S * I),  # W once, E in, X mask, E out
            "avg_launch_us": avg_us, "launches": n, "flop_per_launch": flops[dom],
            "classes_ms_per_step": {k: v[0] / n_roof for k, v in cls.items()},
            "steps": n_roof,
        }

    out = {
        "metric": "BPTT timesteps*streams/sec at 1024-hidden/256-stream",
        "value": args.steps * S * world / elapsed,
        "unit": "stream-timesteps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": "text-predict multi-tap generation: hidden %d, %d streams/GPU, BPTT depth "
                        "%d, 42 symbols, RELU, weighted momentum, lr 1e-5" % (Hd, S, D),
            "streams_per_gpu": S, "global_streams": S * world,
            "parallelism": "streams sharded x%d, delta all-reduce (RCCL)" % world if world > 1
                           else "single GPU",
            "mean_bptt_depth": mean_depth, "hidden_zero_fraction": zero_frac,
            "training_entropy_bits": t_entropy,
        },
    }
    if roofline is not None:
        out["roofline"] = roofline
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(rc, sc, amd, gpu, text, i, args.cpu_seconds)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # RCCL writes a version banner through C stdio; push it out first so that the JSON
        # line is the last thing on stdout
        sys.stdout.flush()
        C.CDLL(None).fflush(None)
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
