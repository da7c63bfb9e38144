"""Tree quality as an INPUT: the builder's leaf size `max_members` (the reference's module constant is 8,
src/constants.py:28) at 8 / 4 / 2 on the bench scenes -- what a ray costs (wide-node visits, triangle records, own bytes: device
tallies), ms per sample with 8 sample streams, in the reference's child order and nearest first.  The renderer and the oracle walk
whatever Box[] they are given, so every tree is rendered bit-exactly (tests/test_gpu_round6.py); only max_members = 8 is the
reference's tree, and every other figure here is a labelled experiment.
    python tools/exp_leaf_size.py [scenes=interior] [sizes=8,4,2] [W=1920] [H=1080]"""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tools.exp_order_ab import walk_cost

names = (sys.argv[1] if len(sys.argv) > 1 else "interior").split(",")
sizes = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "8,4,2").split(",")]
W = int(sys.argv[3]) if len(sys.argv) > 3 else 1920
H = int(sys.argv[4]) if len(sys.argv) > 4 else 1080
for name in names:
    for mm in sizes:
        scene, desc = bench.build_scene(name, W, H, max_members=None if mm == 8 else mm)
        print(f"== {desc} {W}x{H}", flush=True)
        for order in (0, 1):
            for rep in range(2):
                print("cost:", json.dumps(dict(walk_cost(scene, order), max_members=mm)), flush=True)
        bench._SCENES.clear()
