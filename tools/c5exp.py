import os, sys, torch, numpy as np
sys.path.insert(0, os.environ["R"])
import bench
from arcle_amd import actions
from arcle_amd.engine import EnvBatch
from arcle_amd.envs import ARCEnv
dev = torch.device("cuda:0"); K = 100; n = 4096
tasks, (bb, oo) = bench.make_tasks_c5(n, 1000, 30, 30), bench.make_actions_c5(K, n, 2000, 30, 30)
bbd, ood = torch.from_numpy(bb).to(dev), torch.from_numpy(oo).to(dev)
for FL in (1, 3):
    batch = EnvBatch(n, 30, 30, -1, "arc", dev)
    batch.set_op_table(actions.table_descs(ARCEnv.default_operations()))
    batch.set_tasks_padded(*tasks); batch.reset()
    def enqueue(sh):
        for i in range(K):
            batch.step_bbox_ptr(bbd[i].data_ptr(), ood[i].data_ptr(), FL, sh)
    sec, _ = bench.graph_time(dev, enqueue, K)
    print("c5 flags", FL, "policy", os.environ.get("ARCLE_STREAM_POLICY"), "wpw", os.environ.get("ARCLE_WPW"), "%.2f us" % (sec * 1e6), "status", batch.status())
