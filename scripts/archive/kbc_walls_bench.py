#!/usr/bin/env python3
"""KBC on a wall-bounded channel (bounce-back columns, periodic rows), 3 steps per launch: split on / off."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import model_bench as mb
import pylbm
for split in (1, 0):
    mb.lib.set_tuning(b"sw_split", split)
    bc = pylbm.Bc.periodic()
    bc.col_lo = bc.col_hi = pylbm.EDGE_BOUNCE_BACK
    mb.bench_single(pylbm.MODEL_KBC, "KBC channel: bounce-back columns, 3 steps per launch, sw_split=%d" % split, 4096, 4096,
                    pylbm.KbcParams(1.0 / (0.5 + 3 * 1.70766666e-4)), bc=bc)
mb.lib.set_tuning(b"sw_split", -1)
