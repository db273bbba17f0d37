import sys, os, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np, torch
import test_agent_parity as T
g = np.load(os.path.join(T.GOLDEN, "agent_trace_ppo.npz"))
import pfrl_amd as pfrl
from pfrl_amd import agents
from pfrl_amd.envs.synthetic import HostSyntheticAtariVectorEnv
from pfrl_amd.nn import Branched
from pfrl_amd.policies import SoftmaxCategoricalHead
pfrl.utils.set_random_seed(0)
env = HostSyntheticAtariVectorEnv(4, seed=5, frame_shape=(12, 12), p_done=0.06)
phi = lambda x: np.asarray(x, dtype=np.float32) / 255
torch.manual_seed(4321)
model = torch.nn.Sequential(torch.nn.Flatten(), torch.nn.Linear(4*144, 32), torch.nn.ReLU(),
    Branched(torch.nn.Sequential(torch.nn.Linear(32, 6), SoftmaxCategoricalHead()), torch.nn.Linear(32, 1)))
opt = torch.optim.SGD(model.parameters(), lr=1e-2)
ag = agents.PPO(model, opt, gpu=0, gamma=0.99, lambd=0.95, phi=phi, update_interval=64, minibatch_size=16, epochs=2, clip_eps=0.1, standardize_advantages=True, max_grad_norm=0.5)
step=[0]
def replay(d):
    a = torch.as_tensor(g["actions"][step[0]], device=ag.device); step[0]+=1; return a
ag._sample_action = replay
losses=[]
orig=ag._lossfun
def spy(*a, **kw):
    out=orig(*a, **kw); losses.append([float(out.detach()), float(ag.value_loss_record.values()[-1]), float(ag.policy_loss_record.values()[-1])]); return out
ag._lossfun=spy
pfrl.experiments.train_agent_batch(ag, env, 70, tempfile.mkdtemp())
got=np.asarray(losses)
print("mean_std", ag._last_dataset["mean_std"].cpu().numpy())
d0=g["dataset0"]; print("ref mean/std", d0[:,0].mean(), d0[:,0].std())
for i in range(len(got)):
    print(i, got[i], g["losses"][i], got[i]-g["losses"][i])
