import numpy as np, torch, sys
sys.path.insert(0,'.')
from multiviewhmr_amd import aggregation
from oracle import cport
d=np.load('tests/golden/unproj_tiny_b2v2c4.npz')
dev=torch.device('cuda:0')
f=torch.from_numpy(d['features']).to(dev); p=torch.from_numpy(d['proj']).to(dev); c=torch.from_numpy(d['coords']).to(dev)
for m in ('sum','softmax'):
    o=aggregation.unprojection(f,p,c,aggregation_method=m).cpu().numpy()
    r=d['out_'+m]
    print(m,'err',np.abs(o-r).max(), 'nan',np.isnan(o).sum(), 'zeros',(o==0).mean(),(r==0).mean())
    print(o[0,0,0,0], r[0,0,0,0])
    print(o[0,1,2,3], r[0,1,2,3])
    print(o[1,3,4,5], r[1,3,4,5])
