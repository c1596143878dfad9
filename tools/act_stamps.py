"""(-DDZ_ACT_STAMPS build only) per-workgroup wall-clock stamps of the one-launch decision."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dqn_zoo_amd import learner as ll, networks
sup = np.linspace(-10, 10, 51).astype(np.float32)
ln = ll.RainbowLearner(networks.RainbowNetwork(6, sup), ll.AdamConfig(), 32)
ln.act_graphs = False
x = torch.randint(0, 256, (1, 84, 84, 4), dtype=torch.uint8, device='cuda')
if os.environ.get('PINNED'):
  xh = torch.randint(0, 256, (1, 84, 84, 4), dtype=torch.uint8).pin_memory()
  from dqn_zoo_amd import device_obs
  x = device_obs.ObservationCache(ln.device).upload(xh[0].numpy())
for _ in range(300):
  ln.apply(x)
torch.cuda.synchronize()
off = int(ln.network.layout(1).c.ws_dfeat_part)
raw = ln._act_ws[off: off + 261 * 32].cpu().numpy().view(np.int64).reshape(261, 16)
t0 = raw[:25, 0].min()
us = (raw - t0) / 100.0
def show(name, rows, labels):
  r = us[rows]
  print(name)
  for i, l in enumerate(labels):
    print('   %-26s mean %6.2f  (%6.2f .. %6.2f)' % (l, r[:, i].mean(), r[:, i].min(), r[:, i].max()))
show('torso, all 25', slice(0, 25), ['start', 'conv1 patch in LDS', 'conv1 stored'])
show('torso, first 24', slice(0, 24), ['start', 'conv1 patch in LDS', 'conv1 stored', 'conv1 outputs all seen', 'conv2 patch in LDS', 'conv2 stored'])
show('torso, first 16', slice(0, 16), ['start', 'p1', 's1', 'seen1', 'p2', 'conv2 stored', 'conv2 outputs all seen', 'conv3 patch in LDS', 'conv3 stored'])
show('fc1, 224', slice(25, 249), ['start', 'W_eff formed', 'features seen', 'partials in LDS', 'slab stored'])
show('tail, 12', slice(249, 261), ['start', 'weights requested, noise drawn', 'slabs seen', 'h1 in LDS', 'ticket taken'])
last = us[249:261, 5].max()
print('last tail workgroup done: %.2f' % last)
