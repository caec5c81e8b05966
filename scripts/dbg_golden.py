import sys, os, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import ibamd
from test_golden import _partition
pk = dict(np.load('tests/golden/advection_partition.npz'))
part = _partition(pk)
dpart = ibamd.to_backend(part, ibamd.hip)
print(dpart.info)
u, C = ibamd.hip(pk['u']), ibamd.hip(pk['C'])
exp = pk['res_adv']
dom = pk['domain']
for flags in (0, 16, 1):
    got = ibamd.to_host(ibamd.residual_advection(dpart, u, C, flags=flags))
    err = np.abs(got-exp)/np.abs(exp).max()
    bad = np.nonzero(err > 1e-5)[0]
    print("flags", flags, "nbad", bad.size, "max", err.max())
    for c in bad[:40]:
        g = dom[c]; print("   cell", c, "gid", g, "blk", g//64, "pos", (g%64)%8, (g%64)//8, "err", err[c], "h", part.spacing[c])
