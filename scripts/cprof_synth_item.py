"""cProfile of the host side of one generator item (config-5 settings).  usage: python scripts/cprof_synth_item.py [items=3]"""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch

import config5_lib as C5
from brainfm_amd import generator as G

items = int(sys.argv[1]) if len(sys.argv) > 1 else 3
np.random.seed(100)
torch.manual_seed(100)
ga = C5.gen_args(160)
ds = G.build_datasets(ga, "cuda:0", cases=[C5.voronoi_case(7)])["all"]
for _ in range(2):
    ds[0]
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(items):
    ds[0]
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(35)
st.sort_stats("cumulative").print_stats(45)
