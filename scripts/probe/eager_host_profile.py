"""Host time of the eager C2 training step (what ExperimentBuilder.train_epoch(use_graphs=False) issues per batch): cProfile over 200 steps."""
import sys, time, cProfile, pstats
import torch
sys.path.insert(0, '.')
from morgana_amd import data, models, optim, synthetic
from morgana_amd import functional as F_hip
dev = torch.device('cuda:0')
model = models.F0Model(precision='bf16').to(dev)
feats = data.to_device(synthetic.make_batch(256, 1000), dev, bf16_tables=model.bf16_table_features())
opt = optim.Adam(model.parameters(), lr=0.01, fused_loop=True)
def step():
    opt.zero_grad()
    loss, _ = model(feats)
    F_hip.backward(loss)
    opt.step()
for _ in range(20):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    step()
host = time.perf_counter() - t0
torch.cuda.synchronize()
print('host issue per step: %.1f us; with sync %.1f us' % (host / 200 * 1e6, (time.perf_counter() - t0) / 200 * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(28)
