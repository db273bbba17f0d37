"""us per forward pass of the Nature trunk + head at B = 32 (HIP-graph replay of 10 passes), and
fwd + bwd.  PFRL_FWD_EXPERIMENT=<bits> selects experimental tile programs (csrc/qnet.hip)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import qnet_check as q  # noqa: E402
dev = torch.device("cuda:0")
ref, dut = q.make_model(dev)
B = int(os.environ.get("B", "32"))
xg = torch.rand(B, 4, 84, 84, device=dev).contiguous(memory_format=torch.channels_last)
x = xg.cpu().contiguous()
def fwd():
    with torch.no_grad():
        return dut(xg)
def fb():
    for p in dut.parameters():
        p.grad = None
    dut(xg).sum().backward()
with torch.no_grad():
    print("max err fwd %.2e" % float((dut(xg).cpu() - ref(x)).abs().max()))
print("exp=%s  fwd %.1f us   fwd+bwd %.1f us" % (os.environ.get("PFRL_FWD_EXPERIMENT", "0"), q.graph_time(fwd), q.graph_time(fb)))
