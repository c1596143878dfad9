"""(-DDZ_HC_STAMPS build only: tools/build_variant.sh dz_rainbow) per-workgroup wall-clock stamps of the
multi-role head launch (csrc/dz_head_chain.h), us after the launch's first stamp."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench

class A: pass
args = A(); args.capacity = 100000; args.batch = 32; args.seed = 1; args.stored_gradients = False
dev = torch.device('cuda', 0)
replay, learner, _ = bench.build_workload(args, dev, 1)
torch.cuda.set_stream(torch.cuda.Stream())
learner.use_graphs = False
step = bench.make_step(replay, learner, 32, fused_next_sample=True)
for _ in range(200):
  step()
torch.cuda.synchronize()
lay = learner.layout.c
NA, K = 6 * 51, 51
nA, nB, nC, nD1 = 3 * 32 * 1, 4 * 3 * ((NA + 63) // 64 + 1), 32, 256
nD2 = ((NA + 63) // 64) * 8 * 2
nG = 84
n = nA + nB + nC + nD1 + nD2 + nG
off = int(lay.ws_dfeat_part)
raw = learner.ws[off: off + n * 16].cpu().numpy().view(np.int64).reshape(n, 8)
t0 = raw[:nA, 0].min()
us = (raw - t0) / 100.0
def show(name, lo, hi, labels):
  r = us[lo:hi]
  print('%s (%d workgroups)' % (name, hi - lo))
  for i, l in enumerate(labels):
    print('   %-34s mean %6.2f  (%6.2f .. %6.2f)' % (l, r[:, i].mean(), r[:, i].min(), r[:, i].max()))
o = 0
show('A fold', o, o + nA, ['start', 'slabs summed (LDS)', 'h1 stored']); o += nA
show('B fc2', o, o + nB, ['start', 'W_eff formed', 'h1 word seen (watch)', 'h1 tile seen', 'slab stored']); o += nB
show('C loss', o, o + nC, ['start', 'producers\' words seen', 'slabs folded (LDS)', 'done']); o += nC
show('D1 dh1 rows', o, o + nD1, ['start', 'samples\' words seen (wave 0)', 'dY in registers (wave 0)', 'done']); o += nD1
show('D2 dW2', o, o + nD2, ['start', 'operands seen', 'stored']); o += nD2
show('G gram', o, o + nG, ['start', 'done']); o += nG
print('launch span: %.2f us' % (us[:, :5].max()))
print('B payload rounds: mean %.2f max %d' % (raw[nA:nA + nB, 5].mean(), raw[nA:nA + nB, 5].max()))
