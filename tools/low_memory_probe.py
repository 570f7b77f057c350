#!/usr/bin/env python3
"""GPU box: the default pass size follows the free device memory (here most of the HBM is held by a torch tensor),
and the frame does not depend on it."""
import sys, os
sys.path.insert(0, ".")
import torch
from raytracer_amd import capi
from tools import config5_stress as c
# occupy most of the HBM so that the default pass size cannot be allocated in one piece
free, total = torch.cuda.mem_get_info()
hog = torch.empty(int(free - 6e9), dtype=torch.uint8, device="cuda")
hs, _ = c.build_scene(300, 1920, 1080, 64, 3); hs.build()
dev = capi.device_scene(hs.desc)
a, st = dev.render(seed=1)
del hog; torch.cuda.empty_cache()
b, st2 = dev.render(seed=1)
import numpy as np
print("passes with 6 GB free:", st.passes, " passes with all memory:", st2.passes, " equal:", np.array_equal(a.view(np.uint32), b.view(np.uint32)))
