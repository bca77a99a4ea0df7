#!/usr/bin/env python3
"""Bitwise reproducibility of the streamed SpMM across launches, copies and processes: prints a digest of Y = A H and
Yt = At G on the 17-instance synthetic batch (34 M nonzeros).  usage: python3 tools/determinism_stream.py [instances]"""
import hashlib, os, sys
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from mllp_amd.graph import synthetic_batch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 17
sb = synthetic_batch(n)
g = torch.Generator(device="cuda").manual_seed(5)
H = torch.randn(sb.N, 16, device="cuda", generator=g)
G = torch.randn(sb.M, 16, device="cuda", generator=g)
dig = lambda t: hashlib.sha1(t.cpu().numpy().tobytes()).hexdigest()[:12]
import numpy as np
print("matrix", [hashlib.sha1(np.ascontiguousarray(sb.export(k)).tobytes()).hexdigest()[:10] for k in range(6)], "x1/x2/labels", dig(sb.x1), dig(sb.x2), dig(sb.labels))
print("inputs", dig(H), dig(G), "generic", dig(sb.spmm(H)), dig(sb.spmm(G, transpose=True)))
for rep in range(3):
    sb.build_spmm_copy(False); sb.build_spmm_copy(True)
    for k in range(3):
        Y, Yt = sb.spmm(H), sb.spmm(G, transpose=True)
        lhs, rhs = float((Y.double() * G.double()).sum()), float((H.double() * Yt.double()).sum())
        print(rep, k, dig(Y), dig(Yt), f"{lhs:.9f} {rhs:.9f} rel={abs(lhs - rhs) / max(abs(lhs), abs(rhs)):.3e}")
    sb.drop_spmm_copy(False); sb.drop_spmm_copy(True)
