import sys
sys.path.insert(0, '.')
import torch
from semantic_superpoint_amd import lib as L
torch.zeros(1, device='cuda')
lib = L.load_library()
for w in (3, 4):
    print("occupancy", w, lib.ssp_debug_occupancy(w))
