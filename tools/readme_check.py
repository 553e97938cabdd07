import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arcle_amd.envs import O2ARCv2Env, ARCVecEnv
from arcle_amd.loaders import SyntheticLoader
from arcle_amd.engine import EnvBatch
import bench
dev = torch.device("cuda:0")
venv = ARCVecEnv(O2ARCv2Env, num_envs=8192, data_loader=SyntheticLoader(n_tasks=100))
obs, info = venv.reset()
bb_np, op_np = bench.make_actions(8, 8192, 1)
bb, op = torch.from_numpy(bb_np).to(dev), torch.from_numpy(op_np).to(dev)
print(venv.batch.launch_info("bbox", venv.flags))
obs, r, t, tr, info = venv.step_bbox(bb[0], op[0])
act5 = torch.cat([bb[1], op[1][:, None]], 1).contiguous()
obs, r, t, tr, info = venv.step_bbox5(act5)
plans = venv.autotune(bb, op)
print("autotune:", plans[0])
cs = venv.capture(bb, op); o, rK, tK, trK = cs.replay()
rows = venv.state_rows()
rows2, rew, term = venv.transition(rows, {"bbox": bb[2], "operation": op[2]})
R = venv.batch.packed_obs_size()
pk = torch.empty((8, 8192, R), dtype=torch.uint8, device=dev)
o, rTN, tTN, info = venv.rollout_bbox(bb, op, packed=pk)
g, gd, rw, tm = EnvBatch.unpack_obs(pk[-1], 30, 30)
assert torch.equal(g, o["grid"]) and torch.equal(rw, rTN[-1]), "packed rows of the last step == final obs"
torch.cuda.synchronize(); venv.check_errors(); print("README snippet ok")
