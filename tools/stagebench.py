import sys, os, torch, numpy as np
sys.path.insert(0, '/root/repo')
import bench
from arcle_amd import actions
from arcle_amd.engine import EnvBatch
from arcle_amd.envs import O2ARCv2Env
dev=torch.device("cuda:0"); n=8192; K=200
bbox_np, op_np = bench.make_actions(K, n, 5)
bbox=torch.from_numpy(bbox_np).to(dev); ops=torch.from_numpy(op_np).to(dev)
batch = EnvBatch(n, 30, 30, -1, "o2arc", dev)
batch.set_op_table(actions.table_descs(O2ARCv2Env.default_operations()))
batch.set_tasks_padded(*bench.make_tasks(n, 1)); batch.reset()
st=torch.cuda.current_stream(dev); sh=st.cuda_stream
for stage in (1,2,3,5,4,0):
    for rep in range(2):
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for i in range(K): batch.step_bbox_ptr(bbox[i].data_ptr(), ops[i].data_ptr(), stage<<8, sh)
        e1.record(st); torch.cuda.synchronize()
    print("stage",stage,"us/launch %.2f"%(e0.elapsed_time(e1)*1e3/K), flush=True)
