"""One-step and trajectory parity on eight latitude bands of the 4096 x 4096 benchmark grid (8 rows x 4096\ncells x 365 days each; GPU box).  python tools/parity_bands.py"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from rsplash_amd import api
from oracle.binding import Oracle
from oracle.parity import assert_one_step, trajectory_summary
from gpu_pipeline import device_pipeline
o=Oracle(); bad=0
off = int(sys.argv[1]) if len(sys.argv) > 1 else 0   # other rows of every band: python tools/parity_bands.py 137
for row in (512, 1024, 1536, 2048, 2560, 3072, 3584, 4088):
    row = row + off if row + off + 8 <= 4096 else row - off
    p=device_pipeline(api, 20221130, 4096, 4096, row*4096, 8*4096, 365, 1)
    msg=[]
    try: assert_one_step(o,p,f"row {row}")
    except AssertionError as e: msg.append("one-step: "+str(e)[:300])
    r2=trajectory_summary(o,p); msg+=["traj: "+v[:300] for v in r2["violations"]]
    bad+=len(msg)
    print(json.dumps({"row":row,"crit":r2["critical_event_cells"],"flip":r2["critical_event_cells_that_differ"],"set_aside":r2["set_aside_cell_days"],"problems":msg})[:800],flush=True)
print("problems",bad)
