#!/usr/bin/env python3
"""All envs execute the same op (rect selections); exactly 100 launches per op so that a rocprofv3 kernel
trace can be chunked per op (tools/prof_summary.py --chunk 100)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from arcle_amd import actions
from arcle_amd.engine import EnvBatch
from arcle_amd.envs import O2ARCv2Env
dev = torch.device("cuda:0"); n = 8192; K = 100
ops_list = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "0,10,20,24,26,28,30,31,32,34,-1").split(",")]
bbox_np, op_np = bench.make_actions(K, n, 5)
bbox = torch.from_numpy(bbox_np).to(dev)
names = ["".join(map(str.capitalize, o.__name__.split("_"))) for o in O2ARCv2Env.default_operations()]
sh = torch.cuda.current_stream(dev).cuda_stream
for o in ops_list:
    batch = EnvBatch(n, 30, 30, -1, "o2arc", dev)
    batch.set_op_table(actions.table_descs(O2ARCv2Env.default_operations()))
    batch.set_tasks_padded(*bench.make_tasks(n, 1)); batch.reset()
    FL = batch.elide_flag | bench.STEP_AUTORESET | int(os.environ.get("OPPROF_EXTRA_FLAGS", "0"))  # the flags ARCVecEnv / bench.py step with (+ e.g. 64 to force the feature instantiation)
    ops = torch.from_numpy(op_np).to(dev) if o < 0 else torch.full((K, n), o, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    for i in range(K):
        batch.step_bbox_ptr(bbox[i].data_ptr(), ops[i].data_ptr(), FL, sh)
    torch.cuda.synchronize()
    print("chunk", "mix" if o < 0 else names[o], flush=True)
