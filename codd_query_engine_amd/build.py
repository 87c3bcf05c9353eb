"""
Build driver for the HIP extension (gfx950 only, in-tree output).

    python -m codd_query_engine_amd.build [--report]

Produces codd_query_engine_amd/csrc/libcodd_knn.so with plain hipcc: the library is a
C-ABI shared object (include/codd_knn.h), not a torch extension, so there is nothing for
torch.utils.cpp_extension to add.  hipcc cross-compiles without a GPU.
"""

from __future__ import annotations

import os
import re
import subprocess
import sys

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
CSRC = os.path.join(_PKG, "csrc")
LIB_PATH = os.path.join(CSRC, "libcodd_knn.so")
SOURCES = ["codd_knn.hip"]
HEADERS = ["wave_topk.h", "row_traits.h", "filter_gemm.h", "filter_i8.h", os.path.join(_ROOT, "include", "codd_knn.h")]

HIPCC_FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-shared",
    # IEEE sqrt / divide and no fma contraction: the canonical score is an
    # exact sequence of fp32 operations shared with oracle/knn_oracle.c
    "-fhip-fp32-correctly-rounded-divide-sqrt",
    "-ffp-contract=off",
    # the filter GEMM's K loop must unroll over its whole register ring (3 slots x the stage length); if the body
    # outgrows LLVM's default pragma-unroll budget the ring silently becomes a scratch array
    # (tests/test_kernel_resources.py guards the outcome)
    "-mllvm",
    "-pragma-unroll-threshold=65536",
    "-Wall",
    "-Wno-unused-function",
]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.sep not in cand or os.path.exists(cand)):
            return cand
    return "hipcc"


def _stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]
    deps.append(os.path.abspath(__file__))
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False, report: bool = False) -> str:
    """Compile if sources are newer than the .so (or force). Returns the library path."""
    if not force and not report and not _stale():
        return LIB_PATH
    cmd = [_hipcc(), *HIPCC_FLAGS, "-I", os.path.join(_ROOT, "include"), "-I", CSRC]
    if report:
        cmd.append("-Rpass-analysis=kernel-resource-usage")
    tmp = LIB_PATH + ".tmp"
    cmd += ["-o", tmp, *[os.path.join(CSRC, s) for s in SOURCES]]
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        sys.stderr.write(proc.stdout + proc.stderr)
        raise RuntimeError("hipcc failed building libcodd_knn.so")
    os.replace(tmp, LIB_PATH)
    if report:
        print(resource_report(proc.stderr))
    return LIB_PATH


def build_variant(name: str, defines: dict) -> str:
    """Compile the same sources with extra -D switches into csrc/libcodd_knn_<name>.so
    (kernel A/B experiments; load it with CODD_KNN_LIB=<path>)."""
    out = os.path.join(CSRC, f"libcodd_knn_{name}.so")
    cmd = [_hipcc(), *HIPCC_FLAGS, "-I", os.path.join(_ROOT, "include"), "-I", CSRC]
    cmd += [f"-D{k}={v}" for k, v in defines.items()]
    cmd += ["-o", out, *[os.path.join(CSRC, s) for s in SOURCES]]
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        sys.stderr.write(proc.stdout + proc.stderr)
        raise RuntimeError(f"hipcc failed building variant {name}")
    return out


def resource_report(remarks: str) -> str:
    """Condense -Rpass-analysis=kernel-resource-usage into one line per kernel."""
    rows, cur = [], {}
    for line in remarks.splitlines():
        m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|TotalSGPRs|SGPRs|VGPRs Spill|SGPRs Spill|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]):\s*(\S+)", line)
        if not m:
            continue
        key, val = m.group(1), m.group(2)
        if key == "Function Name":
            if cur:
                rows.append(cur)
            try:
                val = subprocess.run(["c++filt", val], capture_output=True, text=True).stdout.strip() or val
            except OSError:
                pass
            cur = {"name": re.sub(r"\(anonymous namespace\)::", "", val).split("(")[0]}
        else:
            cur[key] = val
    if cur:
        rows.append(cur)
    # (sspill: SGPRs hipcc parked in lanes of a vector register — no scratch, no VGPR spill, but a v_readlane in front of every use)
    out = ["%-64s %5s %5s %6s %6s %4s %6s %6s" % ("kernel", "vgpr", "sgpr", "vspill", "scratch", "occ", "lds", "sspill")]
    for r in rows:
        out.append("%-64s %5s %5s %6s %6s %4s %6s %6s" % (
            r["name"][-64:], r.get("VGPRs", "?"), r.get("TotalSGPRs", r.get("SGPRs", "?")), r.get("VGPRs Spill", "?"),
            r.get("ScratchSize [bytes/lane]", "?"), r.get("Occupancy [waves/SIMD]", "?"), r.get("LDS Size [bytes/block]", "?"), r.get("SGPRs Spill", "?")))
    return "\n".join(out)


if __name__ == "__main__":
    p = build(force=True, report="--report" in sys.argv)
    print("built", p)
