import json, numpy as np, sys
sys.path.insert(0,'/root/repo')
import btl_bloomfilter_amd as bf
g=json.load(open('tests/golden/hash_vectors.json'))['nthash']
bad=0
for case in g:
    s=case['seq'].encode('latin-1')
    if not s: continue
    hv,valid=bf.hash_seqs(s,case['h'],case['k'])
    v=bf.bits_to_bool(valid,len(s)); pos=np.flatnonzero(v)
    exp=np.array([int(x,16) for x in case['hashes']],dtype=np.uint64).reshape(-1,case['h'])
    got=hv[pos]
    if pos.tolist()!=case['pos'] or not (got==exp).all():
        bad+=1
        if bad<6:
            mism=np.argwhere(got!=exp) if got.shape==exp.shape else None
            print('k',case['k'],'h',case['h'],'len',len(s),'seq',case['seq'][:30],'npos',len(pos),'mismatch rows/cols',mism[:8].tolist() if mism is not None else 'shape')
print('bad cases',bad,'of',len(g))
