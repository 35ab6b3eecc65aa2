for room in 40960 81920 131072; do echo "room $room"; PHQ_PAIR_LISTS_LDS=$room python - <<'P'
import sys; sys.path.insert(0,'.'); sys.path.insert(0,'profiles/tools')
import torch, shape_grid
for fam,L,N,per,pri in shape_grid.shapes():
    if N in (768,1536) and fam in ('single','paired') and pri=='uniform' and ((fam=='single' and L in (12,16,20)) or (fam=='paired' and L in (10,12,16))):
        spec,b=shape_grid.spec_of(fam,L,N,per,pri)
        rate,ms,chain,_=shape_grid.run(torch,spec,16_000_000)
        print(fam,L,N,'%.1f Greads/s %.3f ms'%(rate,ms),' -> '.join('%s %.3f'%(k.replace('pamld_','').replace('_kernel',''),t) for k,t in chain))
P
done
