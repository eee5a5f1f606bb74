"""CPU probe (build container, where oracle/_ref exists): by how much does the REFERENCE differ from ITSELF between two
builds of the same sources -- strict IEEE (-O2 -fno-fast-math -ffp-contract=off: oracle/_ref/librecur_ref.so, the one the
golden vectors come from) and its own flags (-Ofast -ffast-math, fma contraction and reassociation allowed:
librecur_ref_fast.so)?  The same seeded text step for N generations per update rule and shape; reported: the 2-norm and the
largest-element difference of weights and deltas after the run, and the first generation at which an element of the
weights is more than 1e-4 / 2e-4 of the largest off.  Context: VERDICT.md round 3, weak spot 1 (the fuzzer files an ADADELTA
weight element within 5e-4 as an outlier).  usage: python tools/ref_self_difference.py [generations]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import recur_ctypes as rc, scenarios as sc

N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
strict, fast = rc.load_ref(), rc.load_ref(fast=True)
text = sc.synthetic_text(6000)
names = {rc.WEIGHTED: "weighted", rc.NESTEROV: "nesterov", rc.ADAGRAD: "adagrad", rc.ADADELTA: "adadelta"}
print("%-9s %-22s %10s %10s %10s %10s  first gen > 1e-4 / 2e-4 (weights, largest element)" % (
    "rule", "shape (H/S/D, lr)", "w 2-norm", "w max", "delta 2n", "delta max"))
for method in (rc.WEIGHTED, rc.NESTEROV, rc.ADAGRAD, rc.ADADELTA):
    for (H, S, D, lr) in ((64, 7, 8, 1e-3), (99, 6, 10, 3e-3), (128, 16, 6, 1e-2)):
        kw = dict(input_size=42, hidden_size=H, output_size=42, S=S, D=D, learn_rate=lr, seed=11)
        if method == rc.ADADELTA:
            kw['flags'] = rc.FLAG_STANDARD | rc.FLAG_ADAPTIVE_MIN_ERROR | rc.FLAG_AUX_ARRAYS
        a = sc.ApiSet(strict, softmax_best_guess=strict.ref_softmax_best_guess, **kw)
        b = sc.ApiSet(fast, softmax_best_guess=fast.ref_softmax_best_guess, **kw)
        if method in (rc.ADADELTA, rc.ADAGRAD):  # their accumulators start from a ballast, as the callers set them
            for s_, lib in ((a, strict), (b, fast)):
                lib.rnn_set_momentum_values(s_.net, 1e-6 if method == rc.ADADELTA else 0.1)
                if method == rc.ADADELTA:
                    lib.rnn_set_aux_values(s_.net, 1e-6)
        first1 = first2 = None
        for i in range(N):
            a.char_step(text, i, method, 0.9)
            b.char_step(text, i, method, 0.9)
            if first2 is None:
                sa, sb = a.snapshot(), b.snapshot()
                m = max(rc.max_err(sb[k], sa[k]) for k in ("ih_w", "ho_w"))
                if m > 1e-4 and first1 is None:
                    first1 = i
                if m > 2e-4:
                    first2 = i
        sa, sb = a.snapshot(), b.snapshot()
        w2 = max(rc.rel_err(sb[k], sa[k]) for k in ("ih_w", "ho_w"))
        wm = max(rc.max_err(sb[k], sa[k]) for k in ("ih_w", "ho_w"))
        d2 = max(rc.rel_err(sb[k], sa[k]) for k in ("ih_delta", "ho_delta"))
        dm = max(rc.max_err(sb[k], sa[k]) for k in ("ih_delta", "ho_delta"))
        print("%-9s %-22s %10.2e %10.2e %10.2e %10.2e  %s / %s" % (
            names[method], "%d/%d/%d, %g" % (H, S, D, lr), w2, wm, d2, dm, first1, first2))
        a.close()
        b.close()
