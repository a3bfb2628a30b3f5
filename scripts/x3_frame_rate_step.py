import sys, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morgana_amd import data, models, synthetic, optim
from morgana_amd import functional as F_hip
dev = 'cuda:0'
feats = data.to_device(synthetic.make_batch(256, 1000, seed=1), dev)
model = models.F0Model(precision='bf16x3', phone_rate=False).to(dev)
opt = optim.Adam(model.parameters(), lr=1e-3)
for i in range(8):
    opt.zero_grad()
    loss, _ = model(feats)
    F_hip.backward(loss)
    opt.step()
torch.cuda.synchronize()
print(float(loss))
