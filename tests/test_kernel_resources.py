"""Compile-time guard for the hot kernels (no GPU needed: hipcc cross-compiles for gfx950).

The MFMA filter GEMM lives at the edge of the 256-VGPR budget (2 waves per SIMD) and its K-loop must be fully
unrolled by the register-ring depth — if either slips (a spill to scratch, or the ring indexed at run time) the
kernel still computes the right answer, only several times slower.  This test reads hipcc's own resource report."""

import re
import subprocess

from codd_query_engine_amd import build as b


def test_hot_kernels_do_not_spill():
    cmd = [b._hipcc(), *[f for f in b.HIPCC_FLAGS if f != "-shared"], "-c", "-I", b.os.path.join(b._ROOT, "include"), "-I", b.CSRC,
           "-Rpass-analysis=kernel-resource-usage", "-o", "/dev/null", b.os.path.join(b.CSRC, b.SOURCES[0])]
    proc = subprocess.run(cmd, capture_output=True, text=True)
    assert proc.returncode == 0, proc.stderr[-2000:]
    report = b.resource_report(proc.stderr)
    rows = {}
    for line in report.splitlines()[1:]:
        m = re.match(r"(.+?)\s+(\d+)\s+(\S+)\s+(\d+)\s+(\d+)\s+(\d+)\s+(\d+)\s+(\d+)$", line)
        if m:
            rows[m.group(1).strip()] = {"vgpr": int(m.group(2)), "spill": int(m.group(4)), "scratch": int(m.group(5)), "occ": int(m.group(6)), "sspill": int(m.group(8))}
    hot = [name for name in rows if "gemm_filter_kernel" in name or "scan_topk_kernel<0, 1, 3" in name or "scan_topk_kernel<0, 8, 3" in name
           or "finalize_kernel<0, 3" in name]
    assert len(hot) >= 12, report
    for name in hot:
        assert rows[name]["scratch"] == 0 and rows[name]["spill"] == 0, f"{name} spills:\n{report}"
    # the second-generation int8 kernel (filter_i8.h): two waves per SIMD and no scratch (a reload inside the loop would wait
    # vmcnt(0) and drain the hand-counted prefetch)
    tile = [name for name in rows if "i8_tile_kernel" in name]
    assert len(tile) == 19, report  # filter (generic / 3-step-multiple / 6-step-multiple / exactly-6-step rows) and sample, for 16 and for 8 query blocks, staged and resident slices; the fp16 filter
    for name in tile:
        # NO instantiation may touch scratch: the kernel issues its corpus loads and LDS-DMA by inline asm that hipcc cannot see
        # as in flight, so a ring[] / b[] register spilled between its load and the counted s_waitcnt would store stale bytes —
        # neighbours silently dropped; and a reload inside the loop waits vmcnt(0), draining the prefetch
        assert rows[name]["occ"] >= 2 and rows[name]["scratch"] == 0 and rows[name]["spill"] == 0, f"{name}:\n{report}"
        # ... and none may park scalar registers in vector lanes: every reload is a VALU write of an SGPR right in front of a
        # hand-issued vector-memory instruction (the hazard below), and 16 of them per K-step cost the headline kernel 2 %
        assert rows[name]["sspill"] == 0, f"{name} spills SGPRs:\n{report}"
    full = next(name for name in rows if "gemm_filter_kernel<0, 8, 0, 0>" in name)
    int8 = next(name for name in rows if "gemm_filter_kernel<0, 1, 1, 0>" in name)  # the single-query latency kernel (int8 shadow)
    assert rows[int8]["occ"] >= 2, report
    assert rows[full]["vgpr"] <= 256 and rows[full]["occ"] == 2, report


def test_hand_issued_vector_memory_operations_carry_their_own_wait_states():
    """gfx9 hazard: 5 wait states between a VALU write of an SGPR (v_readfirstlane / v_readlane — hipcc's SGPR-spill reloads)
    and a vector-memory instruction reading it.  hipcc pads its own instructions, not the inside of inline asm; in round 3 a
    descriptor word reloaded right in front of a hand-issued load was read stale (non-deterministic lost neighbours).  Every
    inline-asm buffer_load of i8_tile_kernel must therefore sit behind >= 5 wait states INSIDE its own asm block."""
    import os
    import tempfile

    src = b.os.path.join(b._ROOT, "scripts", "dev", "one_kernel.hip")
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        cmd = [b._hipcc(), *[f for f in b.HIPCC_FLAGS if f not in ("-shared", "-fPIC")], "--cuda-device-only", "-S", "-I", b.os.path.join(b._ROOT, "include"),
               "-I", b.CSRC, "-DONEK_MODE=0", "-DONEK_S3=3", "-DONEK_NQB=16", "-DONEK_RES=false", "-o", out, src]
        proc = subprocess.run(cmd, capture_output=True, text=True)
        assert proc.returncode == 0, proc.stderr[-2000:]
        text = open(out).read()
    blocks = re.findall(r";;#ASMSTART\n(.*?);;#ASMEND", text, flags=re.S)
    loads = 0
    for blk in blocks:
        lines = [l.strip() for l in blk.strip().splitlines() if l.strip()]
        for i, l in enumerate(lines):
            if l.startswith("buffer_load"):
                loads += 1
                waits = 0
                for prev in lines[:i]:
                    m = re.match(r"s_nop (\d+)", prev)
                    waits += int(m.group(1)) + 1 if m else 1
                assert waits >= 5, f"hand-issued load without its wait states: {lines}"
    assert loads >= 60, loads  # 13 per interval x 6 unrolled intervals, prologue, ...
    # and the headline instantiation keeps every scalar in scalar registers.  Round 3 first shipped it 12 SGPRs over the budget:
    # no scratch, no VGPR spill — hipcc parks SGPRs in lanes of a spare vector register, which the resource report above does
    # not show — and the query slices' buffer descriptor was re-read with four v_readlane in front of each of the 24 slice DMA
    # of a tile (2,389 in the unrolled program): the kernel lost what the round's schedule changes had gained.
    m = re.findall(r"\.sgpr_spill_count:\s+(\d+)", text)
    assert m and all(int(v) == 0 for v in m), m
    assert len(re.findall(r"\bv_readlane_b32\b", text)) <= 16, "SGPR spill reloads inside i8_tile_kernel<FILTER, 3, 16>"
