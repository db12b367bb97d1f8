"""What the compiler made of the traversal kernels, read from the device assembly (make build/kernels.s).  Three things
that are silent - the results stay right and a pass gets slower - and that each shipped for a while:
  * a spill: the kernels sit within a few registers of the 128-VGPR budget that four waves a SIMD allow;
  * a stack pop through a FLAT load (the LDS read and the HBM read of a pop merged into one load of a selected address);
  * the pipelined walk's node fetch waited for at once: the register allocator lands the five loads in short-lived
    registers and copies them home right away, and that copy waits for the loads a dozen instructions after they were
    issued instead of a triangle phase later (round 4, AO passes +5-8 %, profiles/r04_ab_procs.log)."""
import os
import re
import shutil
import subprocess

import pytest

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tray_racing_amd", "csrc")


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_no_product_kernel_spills_or_exceeds_128_vgprs():
    subprocess.run(["make", "-C", CSRC, "build/kernels.s"], check=True, capture_output=True, timeout=600)
    text = open(os.path.join(CSRC, "build", "kernels.s")).read()
    seen = 0
    for m in re.finditer(r"\.name:\s+(\S*k_trace\S*)\n(.*?)\.wavefront_size", text, re.S):
        name, body = m.group(1), m.group(2)
        t = re.search(r"k_traceILi(\d)ELb(\d)ELi(\d)ELb(\d)ELb(\d)", name)
        assert t, name
        mode, tlas, node, pipe, count = (int(x) for x in t.groups())
        vgpr = int(re.search(r"\.vgpr_count:\s+(\d+)", body).group(1))
        scratch = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", body).group(1))
        seen += 1
        if count:
            continue  # the counting kernels (diagnostics, trx_count_*) may spill: they are never timed
        # (<= 128 VGPRs = four waves to a SIMD for EVERY variant: the persistent grid is sized from one variant per TLAS
        # flavour, trace_grid_size, and must be resident whichever variant is launched)
        assert vgpr <= 128, "k_trace<mode %d, tlas %d, node %d, pipe %d>: %d VGPRs" % (mode, tlas, node, pipe, vgpr)
        assert scratch == 0, "k_trace<mode %d, tlas %d, node %d, pipe %d> spills %d bytes" % (mode, tlas, node, pipe, scratch)
    assert seen >= 3 * 2 * 4, "kernel metadata not found (%d kernels)" % seen


def _kernel_bodies(text):
    for m in re.finditer(r"\n(_ZN3trx\S*k_traceILi(\d)ELb(\d)ELi(\d)ELb(\d)ELb(\d)E\S*):", text):
        end = text.index(".Lfunc_end", m.end())
        yield tuple(int(x) for x in m.groups()[1:]), text[m.end():end].splitlines()


def _is_inst(line):
    line = line.strip()
    return bool(line) and not line.startswith((";", ".")) and not line.endswith(":")


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_stack_pops_are_lds_reads_and_the_node_fetch_is_not_waited_for_at_once():
    subprocess.run(["make", "-C", CSRC, "build/kernels.s"], check=True, capture_output=True, timeout=600)
    text = open(os.path.join(CSRC, "build", "kernels.s")).read()
    pipes = 0
    for (mode, tlas, node, pipe, count), body in _kernel_bodies(text):
        if count:
            continue
        what = "k_trace<mode %d, tlas %d, node %d, pipe %d>" % (mode, tlas, node, pipe)
        assert not any("flat_load_dwordx2" in l or "flat_store_dwordx2" in l for l in body), what + ": a stack access through a flat pointer"
        if not pipe:
            continue
        # the fetch of the NEXT node: five 16-byte loads in a row (80-byte node record) ...
        at = [k for k, l in enumerate(body) if "global_load_dwordx4" in l and sum("global_load_dwordx4" in x for x in body[max(k - 6, 0):k]) >= 4
              and not any("global_load_dwordx4" in x for x in body[k + 1:k + 3])]
        assert at, what + ": node fetch not found"
        k = at[0]
        # ... and the instructions issued before anything waits on vector memory (the triangle phase runs in between)
        n = 0
        for l in body[k + 1:]:
            if "s_waitcnt" in l and "vmcnt" in l:
                break
            n += _is_inst(l)
        assert n >= 100, what + ": the node fetch is waited for after %d instructions" % n
        pipes += 1
    assert pipes == 3 * 4   # AO, explicit rays and the one-launch frame, four node-test semantics
