import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
a = np.load(sys.argv[1])['act1_0'].reshape(3, 32, 400, 32)
b = np.load(sys.argv[2])['act1_0'].reshape(3, 32, 400, 32)
bad = ~np.isclose(a, b, rtol=1e-4, atol=1e-5)
print('bad frac', bad.mean(), 'finite', np.isfinite(b).mean())
print('by group', bad.mean(axis=(1, 2, 3)))
print('by channel', np.round(bad.mean(axis=(0, 1, 2)), 2))
print('by pixel%32', np.round(bad.reshape(3, 32 * 400 // 32, 32, 32).mean(axis=(0, 1, 3)), 2))
print('max abs b', np.nanmax(np.abs(b[np.isfinite(b)])))
