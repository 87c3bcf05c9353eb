#!/usr/bin/env python3
"""Where a wave of i8_tile_kernel<FILTER> spends its cycles (development tool; guide 'In-kernel stamps').

    python scripts/stamps.py build            # here: csrc/libcodd_knn_stamps.so (-DCODD_EXPERIMENTS=1 -DCODD_I8_EXP_STAMPS=1)
    python scripts/stamps.py run [rows] [key=value ...]   # on the GPU box

The diagnostic build stamps s_memtime around the phases of every interval (K-step) and sums them per wave: issue (fragment
prefetch + DMA + corpus loads), the wait for the corpus fragments, the MFMA phase (64 MFMAs + 32 LDS fragment reads), the
end-of-interval wait + barrier, and the epilogue.  The stamps cost time themselves (each drains the scalar/LDS queue): read the
SHARES, not the totals."""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "codd_query_engine_amd", "csrc", "libcodd_knn_stamps.so")

if sys.argv[1] == "build":
    from codd_query_engine_amd import build as b
    print(b.build_variant("stamps", {"CODD_EXPERIMENTS": 1, "CODD_I8_EXP_STAMPS": 1}))
    sys.exit(0)

os.environ["CODD_KNN_LIB"] = LIB
import numpy as np
import torch
from codd_query_engine_amd import native
from codd_query_engine_amd.knn_index import DeviceKnnIndex

rows = int(sys.argv[2]) if len(sys.argv) > 2 and "=" not in sys.argv[2] else 4_000_000
opts = [a for a in sys.argv[2:] if "=" in a]
d, B, k = 768, 256, 10
g = torch.Generator(device="cuda").manual_seed(1)
ix = DeviceKnnIndex(d)
ix.reserve(rows)
for c0 in range(0, rows, 250_000):
    m = min(250_000, rows - c0)
    ix.upsert_device(c0, torch.randn((m, d), generator=g, device="cuda"))
for o in opts:
    key, _, val = o.partition("=")
    ix.set_option(key, int(val))
ix.set_option("shadow8_cooldown", 0)
q = torch.randn((B, d), generator=g, device="cuda")
for _ in range(20):
    ix.search_tensors(q, k)
torch.cuda.synchronize()
lib = native.load()
fn = lib.codd_knn_exp_read_stamps
fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
fn.restype = ctypes.c_int
G = ix.stat("num_cus")
buf = np.zeros((G, 8, 8), dtype=np.uint64)
rc = fn(ix._h, buf.ctypes.data, buf.size, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
assert rc == 0, rc
s = buf.astype(np.float64)
iv = s[:, :, 5]
ok = iv > 0
names = ["issue", "corpus_wait", "mfma_phase", "sync", "epilogue"]
out = {"rows": rows, "options": opts, "workgroups": int(ok.any(axis=1).sum()), "intervals_per_wave": float(iv[ok].mean()), "tiles_per_wg": float(s[:, :, 7][ok].mean())}
tot = s[:, :, 6]
for i, nme in enumerate(names):
    per_iv = s[:, :, i] / np.maximum(iv, 1)
    out[nme + "_cyc_per_interval"] = round(float(per_iv[ok].mean()), 1)
    out[nme + "_share"] = round(float((s[:, :, i][ok] / tot[ok]).mean()), 4)
    out[nme + "_cyc_per_interval_waves0_3_vs_4_7"] = [round(float(per_iv[:, :4][ok[:, :4]].mean()), 1), round(float(per_iv[:, 4:][ok[:, 4:]].mean()), 1)]
out["total_cyc_per_interval"] = round(float((tot / np.maximum(iv, 1))[ok].mean()), 1)
out["total_cyc_per_interval_p10_p90_over_waves"] = [round(float(np.percentile((tot / np.maximum(iv, 1))[ok], p)), 1) for p in (10, 90)]
# spread of a phase between the waves of ONE workgroup (what a barrier turns into waiting)
wg_spread = (s[:, :, 2] / np.maximum(iv, 1)).max(axis=1) - (s[:, :, 2] / np.maximum(iv, 1)).min(axis=1)
out["mfma_phase_spread_within_wg_cyc"] = round(float(wg_spread[ok.any(axis=1)].mean()), 1)
print(json.dumps(out))
