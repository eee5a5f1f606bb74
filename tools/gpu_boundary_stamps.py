"""What the launch boundaries of a text-step generation cost, WITHOUT a profiler between the launches (GPU box; build with
`tools/mkabl.sh bnd -DBND_STAMPS` and run with RECUR_AMD_LIB=build/dev/abl_bnd/librecur_amd.so).

Every workgroup of the four launches marks its first and its last instruction with s_memrealtime (the 100 MHz clock all CUs
share: 10 ns).  A boundary A -> B is then [last workgroup of A ended] -> [first workgroup of B started] and, the other way of
reading it, [last of A ended] -> [LAST workgroup of B started] (the launch is resident).  rocprofv3's kernel trace puts its
own packets between dispatches (profiles/r06_launch_boundaries_under_rocprof.txt: 5.6 - 10 us per boundary, the generation
249 us instead of 220): this is the measurement DESIGN.md section 8 quotes for north_star's "one fused kernel" sentence.

The marks are overwritten by every launch, so a sample is: 60 generations back to back, synchronise, read -- the last
generation's three inner boundaries; and once more with a fifth launch sequence behind it whose first kernel is the next
forward pass (rnn_amd_set_char_step_deltas: its weight-delta launch is the unmarked form), for delta -> next forward."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import recur_ctypes as rc, scenarios as sc

amd = rc.load_amd()
S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
samples = int(sys.argv[2]) if len(sys.argv) > 2 else 20
text = sc.synthetic_text(60000)
g = sc.AmdBatchedSet(amd, input_size=42, hidden_size=1024, output_size=42, S=S, D=20, learn_rate=1e-5, seed=1)
g.load_text(text)
readers = {k: getattr(amd, "ramd_bnd_%s_stamps" % k) for k in ("fwd", "top", "chain", "delta")}
nwg = {"fwd": (S // 32) * 32, "top": S, "chain": 256, "delta": 256}


late = {}  # kernel -> list of [12]: the workgroup numbers that ended last
per_xcd = {}  # kernel -> list of [8] (workgroups blockIdx % 8 == x: the dispatcher deals them to the XCDs in turn): latest end, relative to the launch's first start


def marks():
    out = {}
    for k, f in readers.items():
        buf = np.zeros((2, 1024), np.uint64)
        f(C.c_void_p(buf.ctypes.data))
        b = buf[:, :nwg[k]].astype(np.int64)
        b = b[:, (b[0] > 0) & (b[1] > 0)]  # (a launch with fewer workgroups than that: the marks that were written)
        late.setdefault(k, []).append(np.argsort(-(buf[1, :nwg[k]].astype(np.int64)))[:12])
        if b.shape[1] >= 8:
            per_xcd.setdefault(k, []).append([(b[1][x::8].max() - b[0].min()) / 100.0 for x in range(8)] +
                                             [(b[1][x::8].min() - b[0].min()) / 100.0 for x in range(8)])
        out[k] = dict(first_start=int(b[0].min()), last_start=int(b[0].max()), first_end=int(b[1].min()), last_end=int(b[1].max()))
    return out


i = 0
for _ in range(300):
    amd.rnn_amd_set_char_step(g.handle, i, rc.WEIGHTED, 0.95)
    i += 1
amd.rnn_amd_synchronize()
rows = {}
us = lambda a, b: (b - a) / 100.0
for _ in range(samples):
    for _ in range(60):
        amd.rnn_amd_set_char_step(g.handle, i, rc.WEIGHTED, 0.95)
        i += 1
    m = marks()
    for a, b in (("fwd", "top"), ("top", "chain"), ("chain", "delta")):
        rows.setdefault("%s -> %s" % (a, b), []).append((us(m[a]["last_end"], m[b]["first_start"]), us(m[a]["last_end"], m[b]["last_start"]),
                                                       us(m[a]["first_end"], m[a]["last_end"])))
    for k in m:
        rows.setdefault("(%s: first workgroup's start to last workgroup's end)" % k, []).append((us(m[k]["first_start"], m[k]["last_end"]), 0.0, 0.0))
    rows.setdefault("(generation: forward's first start to delta's last end)", []).append((us(m["fwd"]["first_start"], m["delta"]["last_end"]), 0.0, 0.0))
    for _ in range(60):
        amd.rnn_amd_set_char_step(g.handle, i, rc.WEIGHTED, 0.95)
        i += 1
    amd.rnn_amd_set_char_step_deltas(g.handle, i)  # forward of the NEXT generation behind the last delta launch
    i += 1
    m = marks()
    rows.setdefault("delta -> fwd (next generation)", []).append((us(m["delta"]["last_end"], m["fwd"]["first_start"]),
                                                                  us(m["delta"]["last_end"], m["fwd"]["last_start"]),
                                                                  us(m["delta"]["first_end"], m["delta"]["last_end"])))
    amd.rnn_apply_learning(g.net, rc.WEIGHTED, 0.95)
print("hidden 1024, %d streams, depth 20; %d samples; us (10 ns marks)" % (S, samples))
print("%-62s %28s %28s %30s" % ("boundary", "A's last end -> B's first start", "-> B's LAST start", "A's first end -> A's last end"))
tot = 0.0
for k, v in rows.items():
    a = np.array(v)
    if k.startswith("("):
        print("%-62s mean %7.2f  min %7.2f  max %7.2f" % (k, a[:, 0].mean(), a[:, 0].min(), a[:, 0].max()))
    else:
        tot += a[:, 0].mean()
        print("%-62s mean %5.2f min %5.2f max %5.2f   mean %5.2f min %5.2f max %5.2f   mean %5.2f min %5.2f max %5.2f" % (
            k, a[:, 0].mean(), a[:, 0].min(), a[:, 0].max(), a[:, 1].mean(), a[:, 1].min(), a[:, 1].max(), a[:, 2].mean(), a[:, 2].min(), a[:, 2].max()))
print("the four boundaries together: %.2f us" % tot)
print("per XCD (workgroups with blockIdx % 8 == x), us after the launch's first start: the LAST end among them | the FIRST end among them")
for k, v in per_xcd.items():
    a = np.array(v).mean(axis=0)
    print("%-6s last end  " % k + " ".join("%7.2f" % x for x in a[:8]))
    print("%-6s first end " % k + " ".join("%7.2f" % x for x in a[8:]))
print("the twelve workgroups that ended last, by how often over the samples (workgroup number: times)")
for k, v in late.items():
    cnt = {}
    for a in v:
        for w in a:
            cnt[int(w)] = cnt.get(int(w), 0) + 1
    print("%-6s " % k + " ".join("%d:%d" % (w, c) for w, c in sorted(cnt.items(), key=lambda t: -t[1])[:16]))
