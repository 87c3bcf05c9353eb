#!/usr/bin/env python3
"""Build (here) or time (on the GPU box) -D variants of the int8 tile kernel (development tool).
    python scripts/ab_i8_variants.py build
    python scripts/ab_i8_variants.py run [rows]
"""
import json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VARIANTS = {
    "noepi": {"CODD_I8_EXP_NOEPI": 1},
    "nohits": {"CODD_I8_EXP_NOHITS": 1},
    "noappend": {"CODD_I8_EXP_NOAPPEND": 1},
    "noflush": {"CODD_I8_EXP_NOFLUSH": 1},
    "noglobal": {"CODD_I8_EXP_NOGLOBAL": 1},
    "latefrags": {"CODD_I8_EARLY_FRAGS": 0},
    "fuse": {"CODD_I8_FUSE_EPI": 1},        # round 3: tile i's pair tests inside the first K-step of tile i + 1 (measured 1.7-3 % slower)
    "spread1": {"CODD_I8_SPREAD_VM": 1},    # round 3: the pair program's vector-memory operations one at a time behind MFMA groups
    "spread2": {"CODD_I8_SPREAD_VM": 2},    # ... SIMD partners taking turns
}


def lib(name):
    return os.path.join(ROOT, "codd_query_engine_amd", "csrc", f"libcodd_knn_{name}.so")


if sys.argv[1] == "build":
    from codd_query_engine_amd import build as b
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(3) as ex:
        # the *_EXP_* switches break results on purpose: they only compile in experiment builds
        # every schedule switch is experiment-only (#error otherwise): only the shipped defaults are under test
        only = sys.argv[2:]
        list(ex.map(lambda kv: b.build_variant(kv[0], {**kv[1], 'CODD_EXPERIMENTS': 1}), [kv for kv in VARIANTS.items() if not only or kv[0] in only]))
    print("built", list(VARIANTS))
else:
    rows = sys.argv[2] if len(sys.argv) > 2 else "4000000"
    for name in ["default"] + [v for v in VARIANTS if len(sys.argv) < 4 or v in sys.argv[3:]]:
        env = dict(os.environ)
        if name != "default":
            if not os.path.exists(lib(name)):
                continue
            env["CODD_KNN_LIB"] = lib(name)
        p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "ab_i8.py"), rows], env=env, capture_output=True, text=True, timeout=280)
        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
        print(name, line[-1] if line else "FAILED " + p.stderr[-300:], flush=True)
