#!/usr/bin/env python3
"""Write an APLA index file {"block_i": [r ints], ...} (the reference's inds-*.json layout, apla_vit.py:20-24; mandatory for
multi-GPU runs, apla_vit.py:77): block i gets the first r entries of torch.randperm(dim) under torch.manual_seed(seed + i).
usage: tools/make_inds.py <dim> <depth> <r> <seed> <out.json>"""
import json, sys
import torch
dim, depth, r, seed, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
d = {}
for i in range(depth):
    torch.manual_seed(seed + i)
    d[f"block_{i}"] = torch.randperm(dim)[:r].tolist()
json.dump(d, open(out, "w"))
print(out, depth, "blocks x", r)
