"""Diagnose the matrix-pipe broad phase: primary-hit maps with and without MIRT_FLAG_VALU_BROAD."""
import sys
import numpy as np
from raytracer_amd import abi, presets

w = int(sys.argv[1]) if len(sys.argv) > 1 else 400
sc = presets.bouncing_balls_scene(1)
cam = presets.bouncing_balls_camera(w, 0, 50)
a_ids, a_t = cam.primary_hits(sc, precision=abi.F32, accel="valu")
m_ids, m_t = cam.primary_hits(sc, precision=abi.F32)
bad = np.argwhere(a_ids != m_ids)
print("size", a_ids.shape, "mismatches", len(bad))
for y, x in bad[:40]:
    lane = (y % 8) * 8 + (x % 8)
    print(f"  y={y} x={x} lane={lane} valu id={a_ids[y, x]} t={a_t[y, x]:.5f}  mx id={m_ids[y, x]} t={m_t[y, x]:.5f}")
if len(bad):
    lanes = [(y % 8) * 8 + (x % 8) for y, x in bad]
    print("lanes hist", np.bincount(lanes, minlength=64))
    ids = a_ids[bad[:, 0], bad[:, 1]]
    print("missed ids (valu's answer)", np.unique(ids)[:50])
    flat = sc.flat
