import torch, sys
a, b = torch.load('/tmp/a.pt'), torch.load('/tmp/b.pt')
da, db = dict(a), dict(b)
for k in ['e.layer2.2.gout', 'e.layer2.2.out', 'e.layer2.1.gout']:
    x, y = da[k], db[k]
    d = (x - y).abs()
    nz = (d > 1e-6 * x.abs().max()).nonzero()
    print(k, 'n differing', len(nz), 'of', x.numel(), 'max', float(d.max()), 'xmax', float(x.abs().max()))
    print(nz[:12].tolist())
    if len(nz):
        i = tuple(nz[0].tolist()); print('vals', float(x[i]), float(y[i]))
names=[k for k,_ in a]
print([n for n in names if 'layer2.2' in n])
