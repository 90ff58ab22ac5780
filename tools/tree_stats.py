"""Slot statistics of the 4-wide tree of a scene: used slots per node, triangles per leaf slot.   python tools/tree_stats.py [bathroom|living-room] [detail]"""
import os, sys
import numpy as np
import torch  # noqa: F401
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from raytracer_amd import api, scenes
name = sys.argv[1] if len(sys.argv) > 1 else "bathroom"
s = scenes.interior(name, float(sys.argv[2]) if len(sys.argv) > 2 else 1.0, 1920, 1080)
sc = api.Scene(s)
w = sc.wide_nodes()
ref = w["child"]
used = ref != 0xFFFFFFFF
leaf = used & ((ref & 0x80000000) != 0)
cnt = ((ref >> 27) & 15) + 1
print(name, s.ntris, "triangles", len(w), "nodes; used slots per node:", np.bincount(used.sum(1), minlength=5).tolist(),
      "; leaf slots by triangle count:", np.bincount(cnt[leaf], minlength=8).tolist(), "; leaf slots per node:", np.bincount(leaf.sum(1), minlength=5).tolist())
full = used.sum(1) == 4
two = (leaf & (cnt == 2))
print("nodes with a free slot and a 2-triangle leaf:", int(((~full) & two.any(1)).sum()), "; 2-triangle leaf slots in full nodes:", int(two[full].sum()), "of", int(two.sum()))
