import os, sys
sys.path.insert(0, '/root/repo')
import torch
from sda_amd.nn import ResMLP
from sda_amd.utils import ACTIVATIONS
dev = torch.device('cuda:0')
rows = 1024 * 61
net = ResMLP(47, 15, hidden_features=[128] * 5, activation=ACTIVATIONS['SiLU']).to(dev)
x = torch.randn(rows, 47, device=dev, requires_grad=True)
for _ in range(10): net(x)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): net(x)
e1.record(); torch.cuda.synchronize()
print('SDA_ML_DBG', os.environ.get('SDA_ML_DBG'), 'fwd+saves %.1f us' % (e0.elapsed_time(e1) / 20 * 1e3))
