#!/usr/bin/env python3
"""Extract the cell-count call trees from the reference's committed flamegraphs.

The reference holds no literal golden vectors; its only in-repo pins are the inferno
flamegraphs under verifier/profile/*.svg (written by verifier/src/stark/mod.rs:453-455,
512-514).  Each frame is `<g><title>NAME (N cells, P%)</title><rect x= y= width= ...>`.
This script rebuilds the call tree from the frame geometry and writes, per SVG, a JSON
fixture {"total_samples": T, "frames": [{"path": "a;b;c", "cells": N}, ...]} under
tests/golden/.  The fixture is DATA (names + counts), not reference source.

Run (container only; /root/reference is absent on the GPU box):
    python tools/make_golden_svg.py
"""
import json, re, sys, os

REF = "/root/reference/verifier/profile"
OUT = os.path.join(os.path.dirname(__file__), "..", "tests", "golden")
FRAME = re.compile(
    r'<g><title>([^<(]+) \(([\d,]+) cells, [\d.]+%\)</title><rect x="([\d.]+)%" y="(\d+)" width="([\d.]+)%"')

def parse(path):
    txt = open(path).read()
    total = int(re.search(r'total_samples="(\d+)"', txt).group(1))
    frames = []
    for m in FRAME.finditer(txt):
        name, n, x, y, w = m.group(1).strip(), int(m.group(2).replace(",", "")), float(m.group(3)), int(m.group(4)), float(m.group(5))
        frames.append(dict(name=name, cells=n, x=x, y=y, w=w))
    # flamegraph (non-inverted): root has the largest y; child sits at y-16 within parent's [x, x+w]
    by_y = {}
    for f in frames:
        by_y.setdefault(f["y"], []).append(f)
    ys = sorted(by_y, reverse=True)
    step = ys[0] - ys[1] if len(ys) > 1 else 16
    eps = 1e-3
    for f in frames:
        cands = [p for p in by_y.get(f["y"] + step, []) if p["x"] - eps <= f["x"] and f["x"] + f["w"] <= p["x"] + p["w"] + eps]
        # choose the tightest container
        f["parent"] = min(cands, key=lambda p: p["w"]) if cands else None
    def path_of(f):
        parts = []
        while f is not None:
            parts.append(f["name"]); f = f["parent"]
        return ";".join(reversed(parts))
    out = {}
    for f in frames:
        p = path_of(f)
        out[p] = out.get(p, 0) + f["cells"]
    return dict(total_samples=total, frames=[dict(path=k, cells=v) for k, v in sorted(out.items())])

if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    for name in ("gl", "bn254"):
        d = parse(os.path.join(REF, name + ".svg"))
        with open(os.path.join(OUT, f"svg_frames_{name}.json"), "w") as fh:
            json.dump(d, fh, indent=0)
        print(name, d["total_samples"], len(d["frames"]))
