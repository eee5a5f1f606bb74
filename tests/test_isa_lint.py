"""Build-time check of k_delta_direct's machine code (no GPU needed: hipcc cross-compiles).

The kernel's operand loads are inline-asm global_load_* with counted s_waitcnt vmcnt waits; hipcc does not know that
they are asynchronous.  While the kernel was written it twice produced code that read or re-used a register between a
load and the wait that covers it (copies of ring registers in front of a tied wait; accumulators moved into ring
registers whose surplus loads were still in flight) -- wrong results that depend on timing.  tools/isa_lint_async_loads.py
walks the assembly with a model of the wave's load queue; this test compiles the kernel (through its microbenchmark's
translation unit, which instantiates the same template as the library) and requires a clean walk, no scratch memory and
no accumulation registers shuffled through AGPRs."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_delta_direct_touches_no_register_with_a_load_outstanding(tmp_path):
    asm = str(tmp_path / "dd.s")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-I" + os.path.join(ROOT, "recur_amd", "csrc"),
                    os.path.join(ROOT, "tools", "delta_direct_microbench.hip"), "-S", "--cuda-device-only", "-o", asm],
                   check=True, capture_output=True)
    text = open(asm).read()
    for npw in (1, 2):  # one piece of the rest rows per workgroup, or two
        sym = "k_delta_directILi8ELi5ELi%dEE" % npw
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_lint_async_loads.py"), asm, sym],
                           capture_output=True, text=True)
        assert r.returncode == 0 and " 0 problems" in r.stdout, r.stdout[-3000:]
        body = text[text.index("_Z14" + sym + "v6DdArgs:"):]
        body = body[:body.index("s_endpgm")]
        assert "scratch_" not in body and "v_accvgpr" not in body
        # two loops (all-ones / coefficients) x ring x (16 + pieces)
        assert body.count("v_mfma_f32_16x16x4_f32") == 2 * 5 * (16 + npw)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_delta_direct_with_the_top_layer_in_its_launch(tmp_path):
    """The library's own instantiations, k_delta_direct_ho among them (the top layer's delta and update between the
    ring's first requests and the first wait): its register pressure made hipcc keep scalar bases in VGPR lanes, and a
    v_readlane directly in front of an inline-asm load is a hazard that hipcc does not pad (the third rule of the lint:
    five wait states between a vector-ALU write of an SGPR and a memory instruction that reads it) -- `Memory access
    fault ... address (nil)` on the GPU until the loads outside the loop got their own s_nop."""
    asm = str(tmp_path / "bptt.s")
    src = os.path.join(ROOT, "recur_amd", "csrc")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-I" + src, "-I" + os.path.join(ROOT, "include"),
                    os.path.join(src, "kernels_bptt.hip"), "-S", "--cuda-device-only", "-o", asm],
                   check=True, capture_output=True)
    text = open(asm).read()
    for sym in ("k_delta_direct_hoILi8ELi5ELi1EE", "k_delta_direct_hoILi8ELi5ELi2EE",
                "k_delta_directILi8ELi5ELi1EE", "k_delta_directILi8ELi5ELi2EE"):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_lint_async_loads.py"), asm, sym],
                           capture_output=True, text=True)
        assert r.returncode == 0 and " 0 problems" in r.stdout, r.stdout[-3000:]
        start = next(i for i, l in enumerate(text.split("\n")) if l.startswith("_Z") and sym in l and l.split(";")[0].rstrip().endswith(":"))
        body = "\n".join(text.split("\n")[start:])
        body = body[:body.index("s_endpgm")]
        assert "scratch_" not in body and "v_accvgpr" not in body


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_chain_kernel_machine_code_contract(tmp_path):
    """k_chain_persist's multiplying waves are written instruction by instruction (SGPR-base stores and gate loads in inline
    asm, counted lgkmcnt waits on the A fragments) around a 128-register weight panel; what every build has to keep, for
    every one of its 45 instantiations (activation x hidden size x ONE / PAD / dense tail):
      * the two wait-state rules of tools/isa_lint_async_loads.py that hipcc does not apply INSIDE inline asm: two wait
        states between an inline-asm vector-ALU write and the MFMA that reads the register, five between a vector-ALU
        write of an SGPR (v_readlane: spilled SGPRs live in VGPR lanes) and an inline-asm memory instruction that takes
        it as its base (--hazards-only: the load-queue walk models k_delta_direct's ring, not this kernel's mix of
        compiler-tracked loads and asm loads behind full waits);
      * no accumulators shuffled through AGPRs, and no more scratch memory than today (a handful of spilled values in the
        prologue / tail of the small-set and small-net instantiations: at most 13 instructions);
    and for the hidden-1024 full-set instantiations (the north star's and its RESQRT / RECLIP20 twins):
      * NO scratch memory at all (round 6: a tail prefetch that lived across the join of the fetching and the multiplying
        waves' paths pushed the kernel from 227 VGPRs to 256 with 640 spills -- and the launch from 100 to 294 us; this
        is the build-time form of that measurement);
      * the burst: 2 x 128 MFMAs for the unrolled pair of half-steps (whole pairs in the twins)."""
    asm = str(tmp_path / "chain.s")
    src = os.path.join(ROOT, "recur_amd", "csrc")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-I" + src, "-I" + os.path.join(ROOT, "include"),
                    os.path.join(src, "kernels_chain.hip"), "-S", "--cuda-device-only", "-o", asm],
                   check=True, capture_output=True)
    text = open(asm).read().split("\n")
    starts = [i for i, l in enumerate(text) if l.startswith("_Z15k_chain_persist") and l.split(";")[0].rstrip().endswith(":")]
    assert len(starts) >= 40, len(starts)
    full_sets = 0
    for st in starts:
        sym = text[st].split(":")[0]
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_lint_async_loads.py"), asm, sym, "--hazards-only"],
                           capture_output=True, text=True)
        assert r.returncode == 0 and " 0 problems" in r.stdout, sym + "\n" + r.stdout[-3000:]
        body = []
        for l in text[st:]:
            body.append(l)
            if "s_endpgm" in l:
                break
        body = "\n".join(body)
        assert "v_accvgpr" not in body, sym
        assert body.count("scratch_") <= 13, (sym, body.count("scratch_"))
        if "ELi1024ELb0ELb0ELb0E" in sym:
            full_sets += 1
            assert "scratch_" not in body, sym
            mfmas = sum(1 for l in body.split("\n") if l.strip().startswith("v_mfma_f32_16x16x4_f32"))
            # (the north star's own instantiation: exactly the unrolled pair; hipcc duplicates the pair in the twins)
            assert mfmas % 256 == 0 and mfmas >= 256 and (mfmas == 256 or "ILi1E" not in sym), (sym, mfmas)
    assert full_sets == 3
